// Stem (1->64, 3x3, ReLU) and head (64->1, 3x3, + global residual) stencils: HBM-bound VALU kernels.
//   stem: self.relu(self.input(x)) / self.relu(self.input_c(y))   CODON_x4.py:68,71
//   head: torch.add(self.output(out), residual)                    CODON_x4.py:130-131
// fp32 activations (NCHW): both walk rows with VEC consecutive pixels per lane, so a wave touches 64*VEC*4
// contiguous bytes per load/store instruction (1 KiB at VEC=4).  VEC=4 needs W % 4 == 0 (every
// plane row is then 16-byte aligned); other widths take the VEC=1 instantiation.
// 16-bit activations are channel-blocked (c8.h) and take the kernels of ew_c8.hip.

#include <type_traits>

#include "codon_common.h"

namespace codon {

template <int VEC>
struct Vec;
template <>
struct Vec<4> { using T = float4; };
template <>
struct Vec<1> { using T = float; };

// element accessors (fp32 activations; math is always fp32)
__device__ __forceinline__ float ldx(const float* p) { return *p; }
__device__ __forceinline__ void stx(float* p, float v) { *p = v; }
__device__ __forceinline__ void ld4(const float* p, float (&o)[4]) {
  const float4 c = *reinterpret_cast<const float4*>(p);
  o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = c.w;
}
__device__ __forceinline__ void st4(float* p, const float (&o)[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
}

template <int VEC, typename T>
__device__ __forceinline__ void load_row(const T* __restrict__ plane, int gy, int gx0, int H, int W,
                                         float (&v)[VEC + 2]) {
  // v[0] = pixel gx0-1 ... v[VEC+1] = pixel gx0+VEC, zero outside the image
  if (gy < 0 || gy >= H) {
#pragma unroll
    for (int i = 0; i < VEC + 2; ++i) v[i] = 0.f;
    return;
  }
  const T* row = plane + (long)gy * W;
  if constexpr (VEC == 4) {
    float c[4];
    ld4(row + gx0, c);
    v[1] = c[0]; v[2] = c[1]; v[3] = c[2]; v[4] = c[3];
  } else {
    v[1] = ldx(row + gx0);
  }
  v[0] = gx0 > 0 ? ldx(row + gx0 - 1) : 0.f;
  v[VEC + 1] = gx0 + VEC < W ? ldx(row + gx0 + VEC) : 0.f;
}

// flags: 1 = ReLU on the output; 2 = use the spatially flipped taps (w[c][8-t]): with w = output.weight
// this is dL/dt of the head conv (y = sum_c conv3x3(t_c, w_c)) given dL/dy.
// mask (optional, 64 channels): out = mask > 0 ? out : 0 (backward through conv11's ReLU).
template <int VEC, typename T>
__device__ __forceinline__ void stem_body(const float* __restrict__ x, const float* __restrict__ w,
                                          T* __restrict__ y, int H, int W, long y_img,
                                          long y_base, long total, int flags,
                                          const T* __restrict__ mask, long m_img, long m_base, long blk) {
  __shared__ float wsh[64 * 9];
  for (int i = threadIdx.x; i < 576; i += 256) wsh[i] = (flags & 2) ? w[(i / 9) * 9 + 8 - (i % 9)] : w[i];
  __syncthreads();
  const long idx = blk * 256L + threadIdx.x;
  if (idx >= total) return;
  const int WV = W / VEC;
  const int gxv = (int)(idx % WV);
  const long t = idx / WV;
  const int gy = (int)(t % H);
  const int b = (int)(t / H);
  const int gx0 = gxv * VEC;
  const float* plane = x + (long)b * H * W;
  float r0[VEC + 2], r1[VEC + 2], r2[VEC + 2];
  load_row<VEC, float>(plane, gy - 1, gx0, H, W, r0);
  load_row<VEC, float>(plane, gy, gx0, H, W, r1);
  load_row<VEC, float>(plane, gy + 1, gx0, H, W, r2);
  const long HW = (long)H * W;
  T* yo = y + (long)b * y_img + y_base + (long)gy * W + gx0;
  const T* mo = mask ? mask + (long)b * m_img + m_base + (long)gy * W + gx0 : nullptr;
  const bool relu = flags & 1;
#pragma unroll 4
  for (int co = 0; co < 64; ++co) {
    const float* k = wsh + co * 9;
    float o[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float a = k[0] * r0[i];
      a = fmaf(k[1], r0[i + 1], a);
      a = fmaf(k[2], r0[i + 2], a);
      a = fmaf(k[3], r1[i], a);
      a = fmaf(k[4], r1[i + 1], a);
      a = fmaf(k[5], r1[i + 2], a);
      a = fmaf(k[6], r2[i], a);
      a = fmaf(k[7], r2[i + 1], a);
      a = fmaf(k[8], r2[i + 2], a);
      o[i] = relu ? fmaxf(a, 0.f) : a;
    }
    if (mo) {
      if constexpr (VEC == 4) {
        float m[4];
        ld4(mo + co * HW, m);
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = m[i] > 0.f ? o[i] : 0.f;
      } else {
        o[0] = ldx(mo + co * HW) > 0.f ? o[0] : 0.f;
      }
    }
    if constexpr (VEC == 4) st4(yo + co * HW, o);
    else stx(yo + co * HW, o[0]);
  }
}

template <int VEC, typename T>
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                   T* __restrict__ y, int H, int W, long y_img,
                                                   long y_base, long total, int flags,
                                                   const T* __restrict__ mask, long m_img, long m_base) {
  stem_body<VEC, T>(x, w, y, H, W, y_img, y_base, total, flags, mask, m_img, m_base, (long)blockIdx.x);
}

// The depth and the guidance stem of a forward (/root/reference/CODON_X4/CODON_x4.py:68,71) as ONE launch (round 6: at one
// image per call each was a launch of 10-17 us that covers a fraction of the chip): workgroups [0, nblk) run image set a,
// [nblk, 2 nblk) set b -- same code per workgroup, same bits.
template <int VEC, typename T>
__global__ __launch_bounds__(256) void stem_pair_kernel(const float* __restrict__ xa, const float* __restrict__ wa,
                                                        T* __restrict__ ya, long ya_img, long ya_base,
                                                        const float* __restrict__ xb, const float* __restrict__ wb,
                                                        T* __restrict__ yb, long yb_img, long yb_base, int H, int W,
                                                        long total, unsigned nblk) {
  if (blockIdx.x < nblk) stem_body<VEC, T>(xa, wa, ya, H, W, ya_img, ya_base, total, 1, nullptr, 0, 0, (long)blockIdx.x);
  else stem_body<VEC, T>(xb, wb, yb, H, W, yb_img, yb_base, total, 1, nullptr, 0, 0, (long)(blockIdx.x - nblk));
}

// head: y = sum_c conv3x3(x_c, w_c) + res.  One thread owns VEC consecutive pixels of R consecutive rows (a band): per
// channel it loads the band's R + 2 rows ONCE and produces all R output rows from registers, so an input element is
// fetched (R + 2) / R times instead of 3 (round 1: one row per thread, 6.08 GB moved for 2.60 GB algorithmic -- the
// vertically adjacent rows sat in other workgroups on other XCDs, whose L2s each fetched them again).
// VEC = 4 (16-byte accesses); VEC = 1 for widths that are not a multiple.  16-bit activations: head_c8_kernel (ew_c8.hip).
// DEPTH channels are fetched before the first of them is used: one 128 x 128 image is 32 waves, and a wave that waits for
// every channel's rows before it asks for the next one's is 64 memory round trips long (57 us; 4 in flight: 37 us).  The
// channels are still accumulated one after the other in the same order: same bits.
template <int VEC, int R, typename T, int DEPTH = 1>
__global__ __launch_bounds__(256) void head_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ res, float* __restrict__ y, int H,
                                                   int W, long x_img, long x_base, int nband, int nseg, int nwave,
                                                   int nblk) {
  __shared__ float wsh[64 * 9];
  for (int i = threadIdx.x; i < 576; i += 256) wsh[i] = w[i];
  __syncthreads();
  // a WAVE owns one (image, band of R rows, segment of 64 * VEC columns): lane neighbours are pixel neighbours, so the
  // two halo columns of a lane's vector come from the adjacent lanes (ds_bpermute), not from memory; only lanes 0 and 63
  // fetch theirs.  Consecutive waves walk a band, then the next band of the image: with the XCD remap vertically adjacent
  // bands (which share their 2 halo rows) sit in the same XCD's L2.
  const int lane = threadIdx.x & 63;
  // readfirstlane: the wave index is wave-uniform but hipcc cannot prove it (threadIdx.x >> 6); left as a VGPR value it
  // wraps every buffer load in a waterfall loop over "distinct" descriptors / scalar offsets + s_waitcnt vmcnt(0)
  const int wid = __builtin_amdgcn_readfirstlane((int)xcd_remap(blockIdx.x, (unsigned)nblk) * 4 + (int)(threadIdx.x >> 6));
  if (wid >= nwave) return;
  const int seg = wid % nseg;
  const int t = wid / nseg;
  const int band = t % nband, b = t / nband;
  const int WV = W / VEC;
  const int gxv = seg * 64 + lane;
  const bool act = gxv < WV;
  const int gx0 = act ? gxv * VEC : 0, gy0 = band * R;
  const long HW = (long)H * W;
  const T* xb = x + (long)b * x_img + x_base;
  const bool ldl = act && lane == 0 && gx0 > 0;                 // this lane's left / right halo column comes from memory
  const bool ldr = act && lane == 63 && gx0 + VEC < W;
  const bool shl = lane > 0, shr = lane < 63 && gx0 + VEC < W;  // ... or from the neighbouring lane
  float o[R][VEC];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int i = 0; i < VEC; ++i) o[r][i] = 0.f;

  // Every load is an unconditional buffer instruction: `cond ? load : 0` compiles to a branch and an s_waitcnt vmcnt(0)
  // PER LOAD (one memory round trip each).  Lanes that must not load carry an out-of-range offset (returns 0, no
  // traffic); rows outside the image take a zero-length descriptor (wave-uniform SGPR select); the (channel, row) term
  // of the address is a scalar offset.
  constexpr unsigned OOB = 0xFFFFFFF0u;
  constexpr int ES = (int)sizeof(T);
  const __amdgpu_buffer_rsrc_t rs_img = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (int)(64u * (unsigned)HW * ES), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_nil = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, 0, 0x00020000);
  const unsigned vo = act ? (unsigned)gx0 * ES : OOB;
  const unsigned vl = ldl ? (unsigned)(gx0 - 1) * ES : OOB;
  const unsigned vr = ldr ? (unsigned)(gx0 + VEC) * ES : OOB;
  auto load_rows = [&](int c, float (&rw)[R + 2][VEC + 2]) {
#pragma unroll
    for (int j = 0; j < R + 2; ++j) {
      const int gy = gy0 - 1 + j;
      const bool rowok = gy >= 0 && gy < H;                       // wave-uniform
      const __amdgpu_buffer_rsrc_t rs = rowok ? rs_img : rs_nil;
      const unsigned so = ((unsigned)c * (unsigned)HW + (unsigned)(rowok ? gy : 0) * (unsigned)W) * ES;
      static_assert(ES == 4, "head_kernel is the fp32 (NCHW) kernel; 16-bit tensors take head_c8_kernel");
      if constexpr (VEC == 4) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0);
        const float4 f = *reinterpret_cast<const float4*>(&v);
        rw[j][1] = f.x; rw[j][2] = f.y; rw[j][3] = f.z; rw[j][4] = f.w;
      } else if constexpr (VEC == 2) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b64(rs, vo, so, 0);
        const float2 f = *reinterpret_cast<const float2*>(&v);
        rw[j][1] = f.x; rw[j][2] = f.y;
      } else {
        rw[j][1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, vo, so, 0));
      }
      rw[j][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, vl, so, 0));
      rw[j][VEC + 1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, vr, so, 0));
    }
  };
  auto accumulate = [&](int c, float (&rw)[R + 2][VEC + 2]) {
    float k[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) k[j] = wsh[c * 9 + j];
#pragma unroll
    for (int j = 0; j < R + 2; ++j) {
      const float fl = __shfl_up(rw[j][VEC], 1, 64), fr = __shfl_down(rw[j][1], 1, 64);
      rw[j][0] = shl ? fl : rw[j][0];
      rw[j][VEC + 1] = shr ? fr : rw[j][VEC + 1];
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float a = o[r][i];           // same tap order as round 1: results unchanged bit for bit
        a = fmaf(k[0], rw[r][i], a);
        a = fmaf(k[1], rw[r][i + 1], a);
        a = fmaf(k[2], rw[r][i + 2], a);
        a = fmaf(k[3], rw[r + 1][i], a);
        a = fmaf(k[4], rw[r + 1][i + 1], a);
        a = fmaf(k[5], rw[r + 1][i + 2], a);
        a = fmaf(k[6], rw[r + 2][i], a);
        a = fmaf(k[7], rw[r + 2][i + 1], a);
        a = fmaf(k[8], rw[r + 2][i + 2], a);
        o[r][i] = a;
      }
  };
  static_assert(64 % DEPTH == 0, "whole groups of channels");
#pragma unroll 1
  for (int c = 0; c < 64; c += DEPTH) {
    float ra[DEPTH][R + 2][VEC + 2];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load_rows(c + d, ra[d]);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) accumulate(c + d, ra[d]);
  }
  if (!act) return;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int gy = gy0 + r;
    if (gy >= H) break;
    const long off = (long)b * HW + (long)gy * W + gx0;
    if constexpr (VEC % 4 == 0) {
#pragma unroll
      for (int q = 0; q < VEC / 4; ++q) {
        const float4 rr = *reinterpret_cast<const float4*>(res + off + 4 * q);
        *reinterpret_cast<float4*>(y + off + 4 * q) = make_float4(o[r][4 * q] + rr.x, o[r][4 * q + 1] + rr.y,
                                                                  o[r][4 * q + 2] + rr.z, o[r][4 * q + 3] + rr.w);
      }
    } else if constexpr (VEC == 2) {
      const float2 rr = *reinterpret_cast<const float2*>(res + off);
      *reinterpret_cast<float2*>(y + off) = make_float2(o[r][0] + rr.x, o[r][1] + rr.y);
    } else {
      y[off] = o[r][0] + res[off];
    }
  }
}

template <typename T>
static int stem_launch(int B, int H, int W, const float* x, const float* w, T* y, int y_ctotal, int y_coff, int flags,
                       const T* mask, int m_ctotal, int m_coff, hipStream_t stream) {
  const long HW = (long)H * W;
  // images of a few thousand pixels (one 128 x 128 pair per call, BASELINE configs[0]): the four-pixel form is 16 workgroups
  // on 256 CUs, each thread 64 dependent row stores long -- the one-pixel form has four times the threads.  Same fma chain
  // per output either way: same bits.
  const bool tiny = (long)B * H * W <= 65536;
  const bool v4 = !tiny && (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) |
                                             reinterpret_cast<uintptr_t>(mask)) % 16 == 0);
  const long total = (long)B * H * (v4 ? W / 4 : W);
  const long blocks = (total + 255) / 256;
  CODON_REQUIRE(blocks < (1L << 31), CODON_ERR_UNSUPPORTED, "stem_fwd: grid too large");
  if (v4)
    hipLaunchKernelGGL((stem_kernel<4, T>), dim3((unsigned)blocks), dim3(256), 0, stream, x, w, y, H, W,
                       y_ctotal * HW, y_coff * HW, total, flags, mask, m_ctotal * HW, m_coff * HW);
  else
    hipLaunchKernelGGL((stem_kernel<1, T>), dim3((unsigned)blocks), dim3(256), 0, stream, x, w, y, H, W,
                       y_ctotal * HW, y_coff * HW, total, flags, mask, m_ctotal * HW, m_coff * HW);
  return check_launch("stem_kernel");
}

// ew_c8.hip: the same stencils over channel-blocked 16-bit tensors
int stem_fwd_c8(int, int, int, const float*, const float*, void*, int, int, int, const void*, int, int, int, hipStream_t);
int head_fwd_c8(int, int, int, const void*, int, int, const float*, const float*, void*, bool, int, hipStream_t);
size_t conv1ch_wgrad_c8_workspace_bytes(int, int, int);
int conv1ch_wgrad_c8(int, int, int, const void*, int, int, const float*, float*, int, float*, size_t, int, hipStream_t);

int stem_fwd(int B, int H, int W, const float* x, const float* w, void* y, int y_ctotal, int y_coff, int flags,
             const void* mask, int m_ctotal, int m_coff, int dtype, hipStream_t stream) {
  if (dtype != CODON_F32)
    return stem_fwd_c8(B, H, W, x, w, y, y_ctotal, y_coff, flags, mask, m_ctotal, m_coff, dtype, stream);
  return stem_launch<float>(B, H, W, x, w, (float*)y, y_ctotal, y_coff, flags, (const float*)mask, m_ctotal, m_coff,
                            stream);
}

int stem_pair_fwd_c8(int, int, int, const float*, const float*, void*, int, int, const float*, const float*, void*, int, int,
                     int, hipStream_t);

int stem_pair_fwd(int B, int H, int W, const float* xa, const float* wa, void* ya, int ya_ctotal, int ya_coff,
                  const float* xb, const float* wb, void* yb, int yb_ctotal, int yb_coff, int dtype, hipStream_t stream) {
  if (dtype != CODON_F32)
    return stem_pair_fwd_c8(B, H, W, xa, wa, ya, ya_ctotal, ya_coff, xb, wb, yb, yb_ctotal, yb_coff, dtype, stream);
  const long HW = (long)H * W;
  const bool tiny = (long)B * H * W <= 65536;                    // as stem_launch: same form, same bits as the lone launches
  const bool v4 = !tiny && (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(xa) | reinterpret_cast<uintptr_t>(ya) |
                                             reinterpret_cast<uintptr_t>(xb) | reinterpret_cast<uintptr_t>(yb)) % 16 == 0);
  const long total = (long)B * H * (v4 ? W / 4 : W);
  const long blocks = (total + 255) / 256;
  CODON_REQUIRE(2 * blocks < (1L << 31), CODON_ERR_UNSUPPORTED, "stem_pair_fwd: grid too large");
  if (v4)
    hipLaunchKernelGGL((stem_pair_kernel<4, float>), dim3((unsigned)(2 * blocks)), dim3(256), 0, stream, xa, wa, (float*)ya,
                       ya_ctotal * HW, ya_coff * HW, xb, wb, (float*)yb, yb_ctotal * HW, yb_coff * HW, H, W, total, (unsigned)blocks);
  else
    hipLaunchKernelGGL((stem_pair_kernel<1, float>), dim3((unsigned)(2 * blocks)), dim3(256), 0, stream, xa, wa, (float*)ya,
                       ya_ctotal * HW, ya_coff * HW, xb, wb, (float*)yb, yb_ctotal * HW, yb_coff * HW, H, W, total, (unsigned)blocks);
  return check_launch("stem_pair_kernel");
}

template <int VEC, int R, typename T, int DEPTH = 1>
static int head_launch_v(int B, int H, int W, const T* x, int x_ctotal, int x_coff, const float* w, const float* res,
                         float* y, hipStream_t stream) {
  const long HW = (long)H * W;
  const int nband = (H + R - 1) / R;
  const int nseg = (W / VEC + 63) / 64;
  const long nwave = (long)B * nband * nseg;
  const long blocks = (nwave + 3) / 4;
  CODON_REQUIRE(nwave < (1L << 31), CODON_ERR_UNSUPPORTED, "head_fwd: grid too large");
  hipLaunchKernelGGL((head_kernel<VEC, R, T, DEPTH>), dim3((unsigned)blocks), dim3(256), 0, stream, x, w, res, y, H, W,
                     x_ctotal * HW, x_coff * HW, nband, nseg, (int)nwave, (int)blocks);
  return check_launch("head_kernel");
}

template <typename T>
static int head_launch(int B, int H, int W, const T* x, int x_ctotal, int x_coff, const float* w, const float* res,
                       float* y, hipStream_t stream) {
  const long HW = (long)H * W;
  CODON_REQUIRE(64 * HW * (long)sizeof(T) < 0xFFFFFFF0L, CODON_ERR_UNSUPPORTED,
                "head_fwd: %dx%d image: 64 channel planes exceed the 4 GiB buffer-descriptor range", H, W);
  // 16-byte (fp32) / 8-byte (16-bit) row accesses need every plane row start aligned to them
  const uintptr_t al = sizeof(T) == 4 ? 16 : 8;
  const bool v4 = (W % 4 == 0) && (reinterpret_cast<uintptr_t>(x) % al == 0) &&
                  ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(res)) % 16 == 0) &&
                  ((x_ctotal * HW * sizeof(T)) % al == 0) && ((x_coff * HW * sizeof(T)) % al == 0);
#ifdef CODON_TUNE
  if (const char* e = getenv("CODON_HEAD_BAND")) {
    const int r = atoi(e);
#define HV(v_, r_) if (r == v_ * 100 + r_) return head_launch_v<v_, r_, T>(B, H, W, x, x_ctotal, x_coff, w, res, y, stream);
    if constexpr (sizeof(T) == 2) { HV(8, 2) HV(8, 4) HV(8, 8) }
    HV(2, 4) HV(2, 8) HV(2, 16) HV(4, 2) HV(4, 4) HV(4, 8) HV(4, 16)
#undef HV
  }
#endif
  // band height R: an input row is fetched (R + 2) / R times.  Measured at 32 x 64 x 480 x 640 (tools/time_head.py, one box):
  // fp32 R = 4 / 8 / 16: 0.531 / 0.468 / 0.462 ms; bf16: 0.429 / 0.393 / 0.345 ms (round 1: 1.11 / 1.30 ms).  Small
  // images keep R = 4 (more waves).
  const bool big = (long)B * H * W >= (1L << 22);
  // a few thousand pixels (one 128 x 128 image: 32 waves of the 4-pixel form, each 16 memory round trips long): one row and
  // 64 pixels per wave, 16 channels' rows in flight -- 256 waves, 4 round trips.  Same accumulation order: same bits.
  if ((long)B * H * W <= 65536) return head_launch_v<1, 1, T, 16>(B, H, W, x, x_ctotal, x_coff, w, res, y, stream);
  if (v4)
    return big ? head_launch_v<4, 16, T>(B, H, W, x, x_ctotal, x_coff, w, res, y, stream)
               : head_launch_v<4, 4, T, 4>(B, H, W, x, x_ctotal, x_coff, w, res, y, stream);
  return big ? head_launch_v<1, 4, T>(B, H, W, x, x_ctotal, x_coff, w, res, y, stream)
             : head_launch_v<1, 4, T, 8>(B, H, W, x, x_ctotal, x_coff, w, res, y, stream);
}

// y16 (16-bit activations only): y is stored in the activations' type instead of fp32
int head_fwd(int B, int H, int W, const void* x, int x_ctotal, int x_coff, const float* w, const float* res, void* y,
             bool y16, int dtype, hipStream_t stream) {
  if (dtype != CODON_F32) return head_fwd_c8(B, H, W, x, x_ctotal, x_coff, w, res, y, y16, dtype, stream);
  return head_launch<float>(B, H, W, (const float*)x, x_ctotal, x_coff, w, res, (float*)y, stream);
}

// ---- weight gradient of the 1->64 / 64->1 3x3 convs ----------------------------------------------
//   R[c][t] = sum_{b,q} A[b,c,q] * s[b, q + (t/3 - 1, t%3 - 1)]        c < 64, t < 9
//   stem: A = dL/d(conv output, ReLU-masked), s = x      -> input.weight.grad[c][0][t]  = R[c][t]
//   head: A = t11 (the head's input),         s = dL/dy  -> output.weight.grad[0][c][t] = R[c][8-t]
// HBM-bound (one read of the 64-channel tensor A): lanes walk x so every A load is a coalesced row segment;
// the 3x3 window of the 1-channel map s is loaded once per (row, 64-pixel chunk) and reused by all channels.
// Workgroup = (image, band of W1_ROWS rows); wave w owns channels 16w .. 16w+15 and keeps their 16 x 9 partial
// sums in registers over the whole band; one wave reduction per sum at the end, per-block partials, fixed-order
// final sum (deterministic).
constexpr int W1_ROWS = 8;

template <typename T>
__global__ __launch_bounds__(256) void conv1ch_wgrad_kernel(const T* __restrict__ a, long a_img, long a_base,
                                                            const float* __restrict__ s, float* __restrict__ part,
                                                            int H, int W, int nrowblk) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / nrowblk, rb = blockIdx.x % nrowblk;
  const long HW = (long)H * W;
  const T* ab = a + b * a_img + a_base + (long)(wave * 16) * HW;
  const float* sb = s + (long)b * HW;
  float acc[16][9];
#pragma unroll
  for (int c = 0; c < 16; ++c)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[c][t] = 0.f;
  const int y0 = rb * W1_ROWS, y1 = min(y0 + W1_ROWS, H);
  for (int y = y0; y < y1; ++y) {
    for (int x0 = 0; x0 < W; x0 += 64) {
      const int x = x0 + lane;
      const bool xin = x < W;
      float sv[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        const bool ok = xin && yy >= 0 && yy < H && xx >= 0 && xx < W;
        const float v = sb[ok ? (long)yy * W + xx : 0];
        sv[t] = ok ? v : 0.f;
      }
      const long off = xin ? (long)y * W + x : 0;
      float av[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) av[c] = ldx(ab + c * HW + off);   // 16 coalesced loads in flight
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const float v = xin ? av[c] : 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[c][t] = fmaf(v, sv[t], acc[c][t]);
      }
    }
  }
  float* o = part + (long)blockIdx.x * 576 + wave * 144;
#pragma unroll
  for (int c = 0; c < 16; ++c)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float v = acc[c][t];
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
      if (lane == 0) o[c * 9 + t] = v;
    }
}

// Second, fixed-order stage: a workgroup owns 64 of the 576 sums; its 16 thread groups each add a contiguous range
// of the partial rows, then the 16 group sums are added in order (deterministic).  The round-2 version walked all the
// rows with 576 threads in three workgroups: 0.63 ms for a 576-value result, longer than the kernel that feeds it.
__global__ __launch_bounds__(1024) void conv1ch_wgrad_reduce_kernel(const float* __restrict__ part,
                                                                    float* __restrict__ dw, int nparts, int flags) {
  const bool flip = flags & CODON_W1_FLIP, accum = flags & CODON_W1_ACCUMULATE;
  __shared__ float red[16][64];
  const int li = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + li;               // 9 workgroups x 64 = 576
  const int per = (nparts + 15) / 16;
  const int p0 = g * per, p1 = min(p0 + per, nparts);
  float s = 0.f;
  for (int p = p0; p < p1; ++p) s += part[(long)p * 576 + i];
  red[g][li] = s;
  __syncthreads();
  if (g == 0) {
    float t = red[0][li];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += red[k][li];
    const int oi = flip ? (i / 9) * 9 + 8 - (i % 9) : i;
    dw[oi] = accum ? dw[oi] + t : t;
  }
}

int conv1ch_wgrad_reduce(const float* part, float* dw, int nparts, int flip, hipStream_t stream) {
  if (flip & CODON_W1_DEFER) return CODON_OK;      // the rows stay in the workspace (codon_reduce_multi)
  hipLaunchKernelGGL(conv1ch_wgrad_reduce_kernel, dim3(9), dim3(1024), 0, stream, part, dw, nparts, flip);
  return check_launch("conv1ch_wgrad_reduce_kernel");
}

size_t conv1ch_wgrad_workspace_bytes(int B, int H, int W) {
  const int nrowblk = (H + W1_ROWS - 1) / W1_ROWS;
  return (size_t)B * nrowblk * 576 * sizeof(float);
}

int conv1ch_wgrad(int B, int H, int W, const void* a, int a_ctotal, int a_coff, const float* s, float* dw, int flip,
                  float* ws, size_t ws_bytes, int dtype, hipStream_t stream) {
  if (dtype != CODON_F32)
    return conv1ch_wgrad_c8(B, H, W, a, a_ctotal, a_coff, s, dw, flip, ws, ws_bytes, dtype, stream);
  CODON_REQUIRE(ws_bytes >= conv1ch_wgrad_workspace_bytes(B, H, W), CODON_ERR_BAD_ARG,
                "conv1ch_wgrad: workspace too small");
  const long HW = (long)H * W;
  const int nrowblk = (H + W1_ROWS - 1) / W1_ROWS;
  hipLaunchKernelGGL(conv1ch_wgrad_kernel<float>, dim3(B * nrowblk), dim3(256), 0, stream, (const float*)a,
                     a_ctotal * HW, a_coff * HW, s, ws, H, W, nrowblk);
  const int st = check_launch("conv1ch_wgrad_kernel");
  if (st != CODON_OK) return st;
  return conv1ch_wgrad_reduce(ws, dw, B * nrowblk, flip, stream);
}

}  // namespace codon
