// HBM-bound kernels of the 16-bit path over CHANNEL-BLOCKED tensors (c8.h): stem / head stencils, the 1-channel weight
// gradients, the CAC gate statistics / apply and their backward, elementwise helpers.  Reference lines as in the fp32
// files (stencil.hip, cac.hip, cac_bwd.hip): CODON_x4.py:68,71,85-118,130-131; CAC_module.py:38-63,78-94.
// One 16-byte vector = 8 channels of one pixel, so a lane that walks pixels touches 16 B per plane and a wave 1 KiB of
// contiguous memory per instruction at ANY H, W (no W % 4 / HW % 8 fast-path conditions).  All arithmetic is fp32.

#include <limits.h>
#include <math.h>

#include "c8.h"

namespace codon {

constexpr int EW_NP = 8;                 // pixels per thread in the tile kernels: tile = 2048 pixels (== PX_TILE of px8.h)
constexpr int EW_TILE = 256 * EW_NP;

__device__ __forceinline__ float c8_wsum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ float c8_wmax(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}
__device__ __forceinline__ int c8_wmin(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = min(v, __shfl_xor(v, m, 64));
  return v;
}

struct C8Slice {          // a channel slice of a C8 buffer, in 16-byte vectors
  const uint4* p;         // buffer base + (coff/8) * HW
  long img;               // (ctotal/8) * HW
};
static C8Slice c8_mk(const void* data, int ctotal, int coff, long HW) {
  return C8Slice{(const uint4*)data + (long)(coff / 8) * HW, (long)(ctotal / 8) * HW};
}
static C8Slice c8_mk(const codon_tensor* t, long HW) { return c8_mk(t->data, t->ctotal, t->coff, HW); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t c8_rsrc(const C8Slice& s, int b, int planes, unsigned HW16) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)(s.p + (long)b * s.img), 0, (int)((unsigned)planes * HW16), C8_RSRC_FLAGS);
}

// ---- stem: y = [relu](conv3x3_{1->64}(x)) [masked] ---------------------------------------------------------------------
// A wave owns (image, row, segment of 64 * VEC columns); a lane owns the VEC pixels seg0 + lane + 64 v, so every store
// instruction covers 64 consecutive pixels of one plane.  Weights live in LDS as [plane][tap][8 ch]: two broadcast
// ds_read_b128 feed 8 * VEC FMAs.  flags: 1 = ReLU, 2 = spatially flipped taps (dL/dt11 of the head conv).
template <class E, int VEC>
__device__ __forceinline__ void stem_c8_body(const float* __restrict__ x, const float* __restrict__ w,
                                             C8Slice y, C8Slice mask, int has_mask, int H, int W, int nseg,
                                             long nwave, int flags, long blk) {
  __shared__ float wsh[8 * 9 * 8];
  for (int i = threadIdx.x; i < 576; i += 256) {
    const int co = i / 9, t = i % 9;
    wsh[((co >> 3) * 9 + t) * 8 + (co & 7)] = (flags & 2) ? w[co * 9 + 8 - t] : w[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const long wid = blk * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (wid >= nwave) return;
  const int seg = (int)(wid % nseg);
  const long t_ = wid / nseg;
  const int gy = (int)(t_ % H), b = (int)(t_ / H);
  const long HW = (long)H * W;
  const unsigned HW16 = 16u * (unsigned)HW;
  const float* xb = x + (long)b * HW;
  float xin[VEC][9];
  unsigned vo[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    const int gx = seg * 64 * VEC + v * 64 + lane;
    const bool in = gx < W;
    vo[v] = in ? 16u * (unsigned)(gy * W + gx) : C8_OOB;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int yy = gy + t / 3 - 1, xx = gx + t % 3 - 1;
      const bool ok = in && yy >= 0 && yy < H && xx >= 0 && xx < W;
      const float q = xb[ok ? (long)yy * W + xx : 0];
      xin[v][t] = ok ? q : 0.f;
    }
  }
  const __amdgpu_buffer_rsrc_t yr = c8_rsrc(y, b, 8, HW16);
  const __amdgpu_buffer_rsrc_t mr = c8_rsrc(has_mask ? mask : y, b, 8, HW16);
  const bool relu = flags & 1;
#pragma unroll 2
  for (int cb = 0; cb < 8; ++cb) {
    float o[VEC][8];
#pragma unroll
    for (int v = 0; v < VEC; ++v)
#pragma unroll
      for (int j = 0; j < 8; ++j) o[v][j] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float4 k0 = *reinterpret_cast<const float4*>(&wsh[(cb * 9 + t) * 8]);
      const float4 k1 = *reinterpret_cast<const float4*>(&wsh[(cb * 9 + t) * 8 + 4]);
      const float k[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
#pragma unroll
      for (int v = 0; v < VEC; ++v)
#pragma unroll
        for (int j = 0; j < 8; ++j) o[v][j] = fmaf(k[j], xin[v][t], o[v][j]);
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      if (relu) {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[v][j] = fmaxf(o[v][j], 0.f);
      }
      if (has_mask) {
        float m8[8];
        c8_unpack<E>(c8_ld(mr, vo[v], (unsigned)cb * HW16), m8);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[v][j] = m8[j] > 0.f ? o[v][j] : 0.f;
      }
      c8_st(c8_pack<E>(o[v]), yr, vo[v], (unsigned)cb * HW16);
    }
  }
}

template <class E, int VEC>
__global__ __launch_bounds__(256) void stem_c8_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      C8Slice y, C8Slice mask, int has_mask, int H, int W, int nseg,
                                                      long nwave, int flags) {
  stem_c8_body<E, VEC>(x, w, y, mask, has_mask, H, W, nseg, nwave, flags, (long)blockIdx.x);
}

// the depth and the guidance stem of a forward as ONE launch (round 6; see stem_pair_kernel in stencil.hip): workgroups
// [0, nblk) run set a, [nblk, 2 nblk) set b
template <class E, int VEC>
__global__ __launch_bounds__(256) void stem_c8_pair_kernel(const float* __restrict__ xa, const float* __restrict__ wa, C8Slice ya,
                                                           const float* __restrict__ xb, const float* __restrict__ wb, C8Slice yb,
                                                           int H, int W, int nseg, long nwave, unsigned nblk) {
  if (blockIdx.x < nblk) stem_c8_body<E, VEC>(xa, wa, ya, ya, 0, H, W, nseg, nwave, 1, (long)blockIdx.x);
  else stem_c8_body<E, VEC>(xb, wb, yb, yb, 0, H, W, nseg, nwave, 1, (long)(blockIdx.x - nblk));
}

template <class E>
static int stem_c8_pair_launch(int B, int H, int W, const float* xa, const float* wa, C8Slice ya, const float* xb,
                               const float* wb, C8Slice yb, hipStream_t stream) {
  constexpr int VEC = 2;
  const int nseg = (W + 64 * VEC - 1) / (64 * VEC);
  const long nwave = (long)B * H * nseg;
  const long blocks = (nwave + 3) / 4;
  CODON_REQUIRE(2 * blocks < (1L << 31), CODON_ERR_UNSUPPORTED, "stem_pair_fwd: grid too large");
  hipLaunchKernelGGL((stem_c8_pair_kernel<E, VEC>), dim3((unsigned)(2 * blocks)), dim3(256), 0, stream, xa, wa, ya, xb, wb, yb,
                     H, W, nseg, nwave, (unsigned)blocks);
  return check_launch("stem_c8_pair_kernel");
}

int stem_pair_fwd_c8(int B, int H, int W, const float* xa, const float* wa, void* ya, int ya_ctotal, int ya_coff,
                     const float* xb, const float* wb, void* yb, int yb_ctotal, int yb_coff, int dtype, hipStream_t stream) {
  CODON_REQUIRE(c8_slice_ok(ya_ctotal, ya_coff, 64) && c8_slice_ok(yb_ctotal, yb_coff, 64), CODON_ERR_BAD_ARG,
                "stem_pair: 16-bit tensors are channel-blocked: ctotal / coff multiples of 8");
  const long HW = (long)H * W;
  CODON_REQUIRE(HW * 2 * 64 < (long)C8_OOB, CODON_ERR_UNSUPPORTED, "stem_pair: image too large for 32-bit buffer offsets");
  const C8Slice sa = c8_mk(ya, ya_ctotal, ya_coff, HW), sb = c8_mk(yb, yb_ctotal, yb_coff, HW);
  return dtype == CODON_F16 ? stem_c8_pair_launch<C8F16>(B, H, W, xa, wa, sa, xb, wb, sb, stream)
                            : stem_c8_pair_launch<C8Bf16>(B, H, W, xa, wa, sa, xb, wb, sb, stream);
}

template <class E>
static int stem_c8_launch(int B, int H, int W, const float* x, const float* w, C8Slice y, C8Slice mask, int has_mask,
                          int flags, hipStream_t stream) {
  constexpr int VEC = 2;
  const int nseg = (W + 64 * VEC - 1) / (64 * VEC);
  const long nwave = (long)B * H * nseg;
  const long blocks = (nwave + 3) / 4;
  CODON_REQUIRE(blocks < (1L << 31), CODON_ERR_UNSUPPORTED, "stem_fwd: grid too large");
  hipLaunchKernelGGL((stem_c8_kernel<E, VEC>), dim3((unsigned)blocks), dim3(256), 0, stream, x, w, y, mask, has_mask, H,
                     W, nseg, nwave, flags);
  return check_launch("stem_c8_kernel");
}

int stem_fwd_c8(int B, int H, int W, const float* x, const float* w, void* y, int y_ctotal, int y_coff, int flags,
                const void* mask, int m_ctotal, int m_coff, int dtype, hipStream_t stream) {
  CODON_REQUIRE(c8_slice_ok(y_ctotal, y_coff, 64) && (!mask || c8_slice_ok(m_ctotal, m_coff, 64)), CODON_ERR_BAD_ARG,
                "stem: 16-bit tensors are channel-blocked: ctotal / coff multiples of 8");
  const long HW = (long)H * W;
  CODON_REQUIRE(HW * 2 * 64 < (long)C8_OOB, CODON_ERR_UNSUPPORTED, "stem: image too large for 32-bit buffer offsets");
  const C8Slice ys = c8_mk(y, y_ctotal, y_coff, HW);
  const C8Slice ms = mask ? c8_mk(mask, m_ctotal, m_coff, HW) : ys;
  return dtype == CODON_F16 ? stem_c8_launch<C8F16>(B, H, W, x, w, ys, ms, mask ? 1 : 0, flags, stream)
                            : stem_c8_launch<C8Bf16>(B, H, W, x, w, ys, ms, mask ? 1 : 0, flags, stream);
}

// ---- head: y = conv3x3_{64->1}(x) + res -------------------------------------------------------------------------------
//   y(r, c) = sum_dx V_dx(r, c + dx - 1),   V_dx(r, col) = sum_{dy, ch} w[ch][dy][dx] * x[ch][(r + dy - 1, col)]
// A wave owns (image, band of R rows, 64 input columns = 62 output columns + one halo column each side); a lane owns
// ONE input column: per plane it loads the band's R + 2 vectors once (an input element is fetched (R + 2) / R * 64 / 62
// times) and accumulates V[r][dx] with the plane's 72 weights held in registers; the three column terms of an output
// meet through two lane shuffles at the end.  Every load is an unconditional buffer instruction: off-image columns
// carry an out-of-range offset, off-image rows a zero-length descriptor (wave-uniform select).
// Y16: the output map is stored in the activations' 16-bit type (a model cast as a whole, test.py:52, returns 16 bits): the
// fp32 sum rounded once -- the value an fp32 store followed by a conversion pass gives, without the pass.
template <class E, int R, bool Y16 = false>
__global__ __launch_bounds__(256) void head_c8_kernel(C8Slice x, const float* __restrict__ w,
                                                      const float* __restrict__ res, void* __restrict__ yv, int H, int W,
                                                      int nband, int nseg, long nwave, int nblk) {
  __shared__ float wsh[8 * 9 * 8];                      // [plane][tap][8 ch]
  for (int i = threadIdx.x; i < 576; i += 256) {
    const int c = i / 9, t = i % 9;
    wsh[((c >> 3) * 9 + t) * 8 + (c & 7)] = w[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const long wid = (long)xcd_remap(blockIdx.x, (unsigned)nblk) * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (wid >= nwave) return;
  const int seg = (int)(wid % nseg);
  const long t_ = wid / nseg;
  const int band = (int)(t_ % nband), b = (int)(t_ / nband);
  const int col = seg * 62 - 1 + lane;                  // this lane's input column
  const int gy0 = band * R;
  const long HW = (long)H * W;
  const unsigned HW16 = 16u * (unsigned)HW;
  const __amdgpu_buffer_rsrc_t rs_img = c8_rsrc(x, b, 8, HW16);
  const __amdgpu_buffer_rsrc_t rs_nil = __builtin_amdgcn_make_buffer_rsrc((void*)(x.p + (long)b * x.img), 0, 0, C8_RSRC_FLAGS);
  const unsigned vo = (col >= 0 && col < W) ? 16u * (unsigned)col : C8_OOB;
  float V[R][3];
#pragma unroll
  for (int r = 0; r < R; ++r) V[r][0] = V[r][1] = V[r][2] = 0.f;

#pragma unroll 1
  for (int cb = 0; cb < 8; ++cb) {
    float k[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float4 k0 = *reinterpret_cast<const float4*>(&wsh[(cb * 9 + t) * 8]);
      const float4 k1 = *reinterpret_cast<const float4*>(&wsh[(cb * 9 + t) * 8 + 4]);
      k[t][0] = k0.x; k[t][1] = k0.y; k[t][2] = k0.z; k[t][3] = k0.w;
      k[t][4] = k1.x; k[t][5] = k1.y; k[t][6] = k1.z; k[t][7] = k1.w;
    }
    u32x4 q[R + 2];
#pragma unroll
    for (int j = 0; j < R + 2; ++j) {
      const int gy = gy0 - 1 + j;
      const bool rowok = gy >= 0 && gy < H;             // wave-uniform
      q[j] = c8_ld(rowok ? rs_img : rs_nil, vo, (unsigned)cb * HW16 + 16u * (unsigned)((rowok ? gy : 0) * W));
    }
#pragma unroll
    for (int j = 0; j < R + 2; ++j) {
      float xv[8];
      c8_unpack<E>(q[j], xv);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int r = j - dy;                           // input row gy0 - 1 + j feeds output row r through filter row dy
        if (r < 0 || r >= R) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          float a = V[r][dx];
#pragma unroll
          for (int c = 0; c < 8; ++c) a = fmaf(k[dy * 3 + dx][c], xv[c], a);
          V[r][dx] = a;
        }
      }
    }
  }
  const bool act = lane >= 1 && lane <= 62 && col < W;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float left = __shfl_up(V[r][0], 1, 64), right = __shfl_down(V[r][2], 1, 64);
    const int gy = gy0 + r;
    if (act && gy < H) {
      const long off = (long)b * HW + (long)gy * W + col;
      const float v = (left + V[r][1] + right) + res[off];
      if constexpr (Y16) reinterpret_cast<unsigned short*>(yv)[off] = (unsigned short)(E::pack2(v, 0.f) & 0xffffu);
      else reinterpret_cast<float*>(yv)[off] = v;
    }
  }
}

template <class E, int R>
static int head_c8_launch(int B, int H, int W, C8Slice x, const float* w, const float* res, void* y, bool y16, hipStream_t stream) {
  const int nband = (H + R - 1) / R;
  const int nseg = (W + 61) / 62;
  const long nwave = (long)B * nband * nseg;
  const long blocks = (nwave + 3) / 4;
  CODON_REQUIRE(blocks < (1L << 31), CODON_ERR_UNSUPPORTED, "head_fwd: grid too large");
  if (y16)
    hipLaunchKernelGGL((head_c8_kernel<E, R, true>), dim3((unsigned)blocks), dim3(256), 0, stream, x, w, res, y, H, W, nband,
                       nseg, nwave, (int)blocks);
  else
    hipLaunchKernelGGL((head_c8_kernel<E, R, false>), dim3((unsigned)blocks), dim3(256), 0, stream, x, w, res, y, H, W, nband,
                       nseg, nwave, (int)blocks);
  return check_launch("head_c8_kernel");
}

int head_fwd_c8(int B, int H, int W, const void* x, int x_ctotal, int x_coff, const float* w, const float* res, void* y,
                bool y16, int dtype, hipStream_t stream) {
  CODON_REQUIRE(c8_slice_ok(x_ctotal, x_coff, 64), CODON_ERR_BAD_ARG,
                "head_fwd: 16-bit tensors are channel-blocked: ctotal / coff multiples of 8");
  const long HW = (long)H * W;
  CODON_REQUIRE(HW * 2 * 64 < (long)C8_OOB, CODON_ERR_UNSUPPORTED, "head_fwd: image too large for 32-bit buffer offsets");
  const C8Slice xs = c8_mk(x, x_ctotal, x_coff, HW);
  const bool big = (long)B * H * W >= (1L << 22);
  if (dtype == CODON_F16)
    return big ? head_c8_launch<C8F16, 8>(B, H, W, xs, w, res, y, y16, stream) : head_c8_launch<C8F16, 4>(B, H, W, xs, w, res, y, y16, stream);
  return big ? head_c8_launch<C8Bf16, 8>(B, H, W, xs, w, res, y, y16, stream) : head_c8_launch<C8Bf16, 4>(B, H, W, xs, w, res, y, y16, stream);
}

// ---- weight gradient of the 1->64 / 64->1 3x3 convs ------------------------------------------------------------------
//   R[c][t] = sum_{b,q} A[b,c,q] * s[b, q + (t/3 - 1, t%3 - 1)]        c < 64, t < 9        (see stencil.hip)
// Workgroup = (image, band of rows); wave w owns planes 2w, 2w+1 (16 channels) and keeps their 16 x 9 partial sums in
// registers over the whole band: lanes walk the row, a load is one 16-byte vector = the 8 channels of a plane.
constexpr int W1C8_ROWS = 8;

// One wave per 8-channel plane (8 waves): 72 accumulators per lane instead of 144 -- 4 instead of 2 waves per SIMD keep
// twice the loads in flight (the kernel waits on memory: 61 % of its wave cycles parked, 2.7 TB/s with two planes per
// wave).  Every accumulator still sees its pixels in the same order: bit-identical to the two-plane form.
constexpr int W1C8_PLANES = 1;       // planes per wave: 1 (8 waves); 2 (4 waves, 208 VGPRs) measured slower in round 4
template <class E>
__global__ __launch_bounds__(512 / W1C8_PLANES) void conv1ch_wgrad_c8_kernel(C8Slice a, const float* __restrict__ s,
                                                                                  float* __restrict__ part, int H, int W,
                                                                                  int nrowblk) {
  constexpr int NP = W1C8_PLANES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / nrowblk, rb = blockIdx.x % nrowblk;
  const long HW = (long)H * W;
  const unsigned HW16 = 16u * (unsigned)HW;
  const __amdgpu_buffer_rsrc_t ar = c8_rsrc(a, b, 8, HW16);
  const float* sb = s + (long)b * HW;
  float acc[NP][8][9];
#pragma unroll
  for (int g = 0; g < NP; ++g)
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[g][c][t] = 0.f;
  const int y0 = rb * W1C8_ROWS, y1 = min(y0 + W1C8_ROWS, H);
  for (int y = y0; y < y1; ++y) {
    for (int x0 = 0; x0 < W; x0 += 64) {
      const int x = x0 + lane;
      const bool xin = x < W;
      const unsigned vo = xin ? 16u * (unsigned)(y * W + x) : C8_OOB;
      u32x4 q[NP];
#pragma unroll
      for (int g = 0; g < NP; ++g) q[g] = c8_ld(ar, vo, (unsigned)(NP * wave + g) * HW16);
      float sv[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        const bool ok = xin && yy >= 0 && yy < H && xx >= 0 && xx < W;
        const float v = sb[ok ? (long)yy * W + xx : 0];
        sv[t] = ok ? v : 0.f;
      }
#pragma unroll
      for (int g = 0; g < NP; ++g) {
        float av[8];
        c8_unpack<E>(q[g], av);
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
          for (int t = 0; t < 9; ++t) acc[g][c][t] = fmaf(av[c], sv[t], acc[g][c][t]);
      }
    }
  }
  float* o = part + (long)blockIdx.x * 576 + wave * (NP * 72);
#pragma unroll
  for (int g = 0; g < NP; ++g)
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float v = c8_wsum(acc[g][c][t]);
        if (lane == 0) o[(g * 8 + c) * 9 + t] = v;
      }
}

// defined in stencil.hip: dw[576] = fixed-order sum of the (nparts, 576) partials
int conv1ch_wgrad_reduce(const float* part, float* dw, int nparts, int flip, hipStream_t stream);

size_t conv1ch_wgrad_c8_workspace_bytes(int B, int H, int W) {
  return (size_t)B * ((H + W1C8_ROWS - 1) / W1C8_ROWS) * 576 * sizeof(float);
}

int conv1ch_wgrad_c8(int B, int H, int W, const void* a, int a_ctotal, int a_coff, const float* s, float* dw, int flip,
                     float* ws, size_t ws_bytes, int dtype, hipStream_t stream) {
  CODON_REQUIRE(c8_slice_ok(a_ctotal, a_coff, 64), CODON_ERR_BAD_ARG,
                "conv1ch_wgrad: 16-bit tensors are channel-blocked: ctotal / coff multiples of 8");
  CODON_REQUIRE(ws_bytes >= conv1ch_wgrad_c8_workspace_bytes(B, H, W), CODON_ERR_BAD_ARG, "conv1ch_wgrad: workspace too small");
  const long HW = (long)H * W;
  const int nrowblk = (H + W1C8_ROWS - 1) / W1C8_ROWS;
  const C8Slice as = c8_mk(a, a_ctotal, a_coff, HW);
  if (dtype == CODON_F16)
    hipLaunchKernelGGL(conv1ch_wgrad_c8_kernel<C8F16>, dim3(B * nrowblk), dim3(512 / W1C8_PLANES), 0, stream, as, s, ws, H,
                       W, nrowblk);
  else
    hipLaunchKernelGGL(conv1ch_wgrad_c8_kernel<C8Bf16>, dim3(B * nrowblk), dim3(512 / W1C8_PLANES), 0, stream, as, s, ws, H,
                       W, nrowblk);
  const int st = check_launch("conv1ch_wgrad_c8_kernel");
  if (st != CODON_OK) return st;
  return conv1ch_wgrad_reduce(ws, dw, B * nrowblk, flip, stream);
}

// ---- CAC statistics: one pass over Fcat = [pre_c | pre] -----------------------------------------------------------------
// grid = (ntiles, B); a thread owns the 8 pixels tile0 + 256 k + tid and walks the 16 planes in Fcat order (colour
// first): the per-pixel channel max / sum stay in two registers per pixel, the per-channel tile sums / maxima are
// reduced in registers over the thread's pixels, then over the wave (shuffles), then over the 4 waves (LDS).
// NP = pixels per thread: NP (2048-pixel tiles), or 1 for small images (256-pixel tiles, all 16 planes' loads in flight:
// cac.hip, STATS_SMALL_HW -- the tile rule is the fp32 pass's, so that codon_cac_stats_tiles holds for every dtype)
template <class E, int NP = EW_NP>
__global__ __launch_bounds__(256) void cac_stats_c8_kernel(C8Slice pre_c, C8Slice pre, float* __restrict__ pooled,
                                                           float* __restrict__ partials, long HW, int ntiles,
                                                           const float* __restrict__ chs) {
  __shared__ float red[128][4][2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x, b = blockIdx.y;
  const long tile0 = (long)tile * (256 * NP);
  const unsigned HW16 = 16u * (unsigned)HW;
  bool ok[NP];
  unsigned vo[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const long q = tile0 + k * 256 + tid;
    ok[k] = q < HW;
    vo[k] = ok[k] ? 16u * (unsigned)q : C8_OOB;
  }
  float pmax[NP], psum[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) { pmax[k] = -INFINITY; psum[k] = 0.f; }
  const __amdgpu_buffer_rsrc_t rc = c8_rsrc(pre_c, b, 8, HW16), rd = c8_rsrc(pre, b, 8, HW16);

  constexpr int PLU = NP == 1 ? 16 : 1;      // one pixel per thread: all 16 planes' loads in flight
#pragma unroll PLU
  for (int pl = 0; pl < 16; ++pl) {
    const __amdgpu_buffer_rsrc_t r = pl < 8 ? rc : rd;
    const unsigned so = (unsigned)(pl & 7) * HW16;
    u32x4 q[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) q[k] = c8_ld(r, vo[k], so);
    float g8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) g8[j] = chs ? chs[b * 64 + (pl & 7) * 8 + j] : 1.f;
    float s8[8], m8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s8[j] = 0.f; m8[j] = -INFINITY; }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      float v[8];
      c8_unpack<E>(q[k], v);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (chs) v[j] *= g8[j];
        pmax[k] = fmaxf(pmax[k], v[j]);
        psum[k] += v[j];
        s8[j] += v[j];                              // out-of-range pixels load as 0
        m8[j] = ok[k] ? fmaxf(m8[j], v[j]) : m8[j];
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float s = c8_wsum(s8[j]), m = c8_wmax(m8[j]);
      if (lane == 0) { red[pl * 8 + j][wave][0] = s; red[pl * 8 + j][wave][1] = m; }
    }
  }
  // per-pixel outputs: plane 0 = channel max, plane 1 = channel mean (max FIRST, CAC_module.py:81)
  float* pm = pooled + (long)b * 2 * HW;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    if (ok[k]) {
      const long q = tile0 + k * 256 + tid;
      pm[q] = pmax[k];
      pm[HW + q] = psum[k] * (1.f / 128.f);
    }
  }
  __syncthreads();
  if (tid < 128) {
    const float s = (red[tid][0][0] + red[tid][1][0]) + (red[tid][2][0] + red[tid][3][0]);
    const float m = fmaxf(fmaxf(red[tid][0][1], red[tid][1][1]), fmaxf(red[tid][2][1], red[tid][3][1]));
    float2* out = reinterpret_cast<float2*>(partials + (((long)b * ntiles + tile) * 128 + tid) * 2);
    *out = make_float2(s, m);
  }
}

int cac_stats_tiles(int H, int W);          // cac.hip
int cac_stats_fwd_c8(int B, int H, int W, const codon_tensor* pc, const codon_tensor* pd, float* pooled, float* partials,
                     int dtype, hipStream_t stream, const float* chs) {
  const long HW = (long)H * W;
  const int nt = cac_stats_tiles(H, W);
  CODON_REQUIRE(B <= 65535, CODON_ERR_UNSUPPORTED, "cac_stats_fwd: batch %d > 65535", B);
  CODON_REQUIRE(c8_slice_ok(pc->ctotal, pc->coff, 64) && c8_slice_ok(pd->ctotal, pd->coff, 64), CODON_ERR_BAD_ARG,
                "cac_stats_fwd: 16-bit tensors are channel-blocked: ctotal / coff multiples of 8");
  CODON_REQUIRE(HW * 2 * 64 < (long)C8_OOB, CODON_ERR_UNSUPPORTED, "cac_stats_fwd: image too large for 32-bit buffer offsets");
  if (nt != (int)((HW + EW_TILE - 1) / EW_TILE)) {          // small image: 256-pixel tiles (cac.hip)
    if (dtype == CODON_F16)
      hipLaunchKernelGGL((cac_stats_c8_kernel<C8F16, 1>), dim3(nt, B), dim3(256), 0, stream, c8_mk(pc, HW), c8_mk(pd, HW), pooled,
                         partials, HW, nt, chs);
    else
      hipLaunchKernelGGL((cac_stats_c8_kernel<C8Bf16, 1>), dim3(nt, B), dim3(256), 0, stream, c8_mk(pc, HW), c8_mk(pd, HW), pooled,
                         partials, HW, nt, chs);
    return check_launch("cac_stats_c8_kernel<small>");
  }
  if (dtype == CODON_F16)
    hipLaunchKernelGGL(cac_stats_c8_kernel<C8F16>, dim3(nt, B), dim3(256), 0, stream, c8_mk(pc, HW), c8_mk(pd, HW), pooled,
                       partials, HW, nt, chs);
  else
    hipLaunchKernelGGL(cac_stats_c8_kernel<C8Bf16>, dim3(nt, B), dim3(256), 0, stream, c8_mk(pc, HW), c8_mk(pd, HW), pooled,
                       partials, HW, nt, chs);
  return check_launch("cac_stats_c8_kernel");
}

// ---- gate apply: out = pre * (ch * sp) + inputs, both streams ---------------------------------------------------------
// grid = (pixel tiles, B); a thread owns NP pixels (tile0 + 256 k + tid) and walks the 16 planes [depth | colour].
struct ApplyC8 {
  C8Slice pre, in, out;
};
template <class E>
__global__ __launch_bounds__(256) void cac_apply_c8_kernel(const ApplyC8 sd, const ApplyC8 sc, const float* __restrict__ ch,
                                                           const float* __restrict__ sp, long HW) {
  constexpr int NP = 4;
  const int tid = threadIdx.x, b = blockIdx.y;
  const long tile0 = (long)blockIdx.x * (256 * NP);
  const unsigned HW16 = 16u * (unsigned)HW;
  unsigned vo[NP];
  float g[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const long q = tile0 + k * 256 + tid;
    const bool ok = q < HW;
    vo[k] = ok ? 16u * (unsigned)q : C8_OOB;
    g[k] = sp[(long)b * HW + (ok ? q : 0)];
  }
#pragma unroll 1
  for (int st = 0; st < 2; ++st) {
    const ApplyC8& s = st ? sc : sd;
    const __amdgpu_buffer_rsrc_t rp = c8_rsrc(s.pre, b, 8, HW16), ri = c8_rsrc(s.in, b, 8, HW16), ro = c8_rsrc(s.out, b, 8, HW16);
#pragma unroll 2
    for (int pl = 0; pl < 8; ++pl) {
      const unsigned so = (unsigned)pl * HW16;
      float gc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) gc[j] = ch[b * 64 + pl * 8 + j];
      u32x4 qp[NP], qi[NP];
#pragma unroll
      for (int k = 0; k < NP; ++k) { qp[k] = c8_ld(rp, vo[k], so); qi[k] = c8_ld(ri, vo[k], so); }
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        float v[8], q[8];
        c8_unpack<E>(qp[k], v);
        c8_unpack<E>(qi[k], q);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], gc[j] * g[k], q[j]);
        c8_st(c8_pack<E>(v), ro, vo[k], so);
      }
    }
  }
}

int cac_apply_fwd_c8(int B, int H, int W, const codon_tensor* pre, const codon_tensor* pre_c, const float* ch,
                     const float* sp, const codon_tensor* in, const codon_tensor* in_c, const codon_tensor* out,
                     const codon_tensor* out_c, int dtype, hipStream_t stream) {
  const long HW = (long)H * W;
  CODON_REQUIRE(B <= 65535, CODON_ERR_UNSUPPORTED, "cac_apply_fwd: batch %d too large", B);
  for (const codon_tensor* t : {pre, pre_c, in, in_c, out, out_c})
    CODON_REQUIRE(c8_slice_ok(t->ctotal, t->coff, 64), CODON_ERR_BAD_ARG,
                  "cac_apply_fwd: 16-bit tensors are channel-blocked: ctotal / coff multiples of 8");
  CODON_REQUIRE(HW * 2 * 64 < (long)C8_OOB, CODON_ERR_UNSUPPORTED, "cac_apply_fwd: image too large for 32-bit buffer offsets");
  const ApplyC8 sd{c8_mk(pre, HW), c8_mk(in, HW), c8_mk(out, HW)}, sc{c8_mk(pre_c, HW), c8_mk(in_c, HW), c8_mk(out_c, HW)};
  const unsigned nt = (unsigned)((HW + 1023) / 1024);
  if (dtype == CODON_F16)
    hipLaunchKernelGGL(cac_apply_c8_kernel<C8F16>, dim3(nt, B), dim3(256), 0, stream, sd, sc, ch, sp, HW);
  else
    hipLaunchKernelGGL(cac_apply_c8_kernel<C8Bf16>, dim3(nt, B), dim3(256), 0, stream, sd, sc, ch, sp, HW);
  return check_launch("cac_apply_c8_kernel");
}

// ---- elementwise helpers ------------------------------------------------------------------------------------------------
// y = x * x * ch[b][c] (64 channels)
template <class E>
__global__ __launch_bounds__(256) void ew_sq_scale_c8_kernel(C8Slice x, const float* __restrict__ ch, C8Slice y, long HW) {
  const int b = blockIdx.y;
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  const unsigned HW16 = 16u * (unsigned)HW;
  const unsigned vo = q < HW ? 16u * (unsigned)q : C8_OOB;
  const __amdgpu_buffer_rsrc_t rx = c8_rsrc(x, b, 8, HW16), ry = c8_rsrc(y, b, 8, HW16);
#pragma unroll
  for (int pl = 0; pl < 8; ++pl) {
    float v[8];
    c8_unpack<E>(c8_ld(rx, vo, (unsigned)pl * HW16), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = v[j] * v[j] * ch[b * 64 + pl * 8 + j];
    c8_st(c8_pack<E>(v), ry, vo, (unsigned)pl * HW16);
  }
}

int ew_sq_scale_c8(int B, int H, int W, const codon_tensor* x, const float* ch, const codon_tensor* y, int dtype,
                   hipStream_t stream) {
  const long HW = (long)H * W;
  CODON_REQUIRE(B <= 65535, CODON_ERR_UNSUPPORTED, "ew_sq_scale: batch %d too large", B);
  CODON_REQUIRE(c8_slice_ok(x->ctotal, x->coff, 64) && c8_slice_ok(y->ctotal, y->coff, 64), CODON_ERR_BAD_ARG,
                "ew_sq_scale: 16-bit tensors are channel-blocked: ctotal / coff multiples of 8");
  const unsigned nt = (unsigned)((HW + 255) / 256);
  if (dtype == CODON_F16)
    hipLaunchKernelGGL(ew_sq_scale_c8_kernel<C8F16>, dim3(nt, B), dim3(256), 0, stream, c8_mk(x, HW), ch, c8_mk(y, HW), HW);
  else
    hipLaunchKernelGGL(ew_sq_scale_c8_kernel<C8Bf16>, dim3(nt, B), dim3(256), 0, stream, c8_mk(x, HW), ch, c8_mk(y, HW), HW);
  return check_launch("ew_sq_scale_c8_kernel");
}

// dst = [dst +] src, then dst = mask > 0 ? dst : 0; grid = (pixel blocks, planes, B)
template <class E>
__global__ __launch_bounds__(256) void ew_add_mask_c8_kernel(C8Slice dst, C8Slice src, int has_src, C8Slice mask,
                                                             int has_mask, int planes, long HW, int accumulate) {
  const int b = blockIdx.z, pl = blockIdx.y;
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  const unsigned HW16 = 16u * (unsigned)HW;
  const unsigned vo = q < HW ? 16u * (unsigned)q : C8_OOB;
  const unsigned so = (unsigned)pl * HW16;
  const __amdgpu_buffer_rsrc_t rd = c8_rsrc(dst, b, planes, HW16);
  float v[8], t[8];
  if (accumulate || !has_src) c8_unpack<E>(c8_ld(rd, vo, so), v);
  else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
  }
  if (has_src) {
    c8_unpack<E>(c8_ld(c8_rsrc(src, b, planes, HW16), vo, so), t);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += t[j];
  }
  if (has_mask) {
    c8_unpack<E>(c8_ld(c8_rsrc(mask, b, planes, HW16), vo, so), t);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = t[j] > 0.f ? v[j] : 0.f;
  }
  c8_st(c8_pack<E>(v), rd, vo, so);
}

int ew_add_mask_c8(int B, int H, int W, int C, const codon_tensor* dst, const codon_tensor* src, const codon_tensor* mask,
                   int accumulate, int dtype, hipStream_t stream) {
  const long HW = (long)H * W;
  CODON_REQUIRE(B <= 65535 && C % 8 == 0, CODON_ERR_UNSUPPORTED, "ew_add_mask: batch %d / channels %d", B, C);
  CODON_REQUIRE(c8_slice_ok(dst->ctotal, dst->coff, C) && (!src || c8_slice_ok(src->ctotal, src->coff, C)) &&
                    (!mask || c8_slice_ok(mask->ctotal, mask->coff, C)),
                CODON_ERR_BAD_ARG, "ew_add_mask: 16-bit tensors are channel-blocked: ctotal / coff multiples of 8");
  CODON_REQUIRE(HW * 2 * C < (long)C8_OOB, CODON_ERR_UNSUPPORTED, "ew_add_mask: image too large for 32-bit buffer offsets");
  const C8Slice d = c8_mk(dst, HW), s = src ? c8_mk(src, HW) : d, m = mask ? c8_mk(mask, HW) : d;
  const dim3 grid((unsigned)((HW + 255) / 256), C / 8, B);
  if (dtype == CODON_F16)
    hipLaunchKernelGGL(ew_add_mask_c8_kernel<C8F16>, grid, dim3(256), 0, stream, d, s, src ? 1 : 0, m, mask ? 1 : 0, C / 8, HW,
                       accumulate);
  else
    hipLaunchKernelGGL(ew_add_mask_c8_kernel<C8Bf16>, grid, dim3(256), 0, stream, d, s, src ? 1 : 0, m, mask ? 1 : 0, C / 8, HW,
                       accumulate);
  return check_launch("ew_add_mask_c8_kernel");
}

// dst = mask > 0 ? s0 + s1 [+ s2 [+ s3]] : 0, summed in fp32 in that order, rounded once (round 4: dL/d(fuse) of the
// fusion trunk -- f_{i+1} = confuse_fuse(...) + fuse, CODON_x4.py:126-128 -- collected in ONE pass over the four dL/d(f_i)
// instead of a copy and three read-modify-write passes); grid = (pixel blocks, planes, B)
struct EwSum4 {
  C8Slice s[4];
};
static_assert(sizeof(EwSum4) <= CODON_KERNARG_LIMIT, "passed by value as a kernel argument");
template <class E>
__global__ __launch_bounds__(256) void ew_sum_mask_c8_kernel(C8Slice dst, EwSum4 src, int nsrc, C8Slice mask, int has_mask,
                                                             int planes, long HW) {
  const int b = blockIdx.z, pl = blockIdx.y;
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  const unsigned HW16 = 16u * (unsigned)HW;
  const unsigned vo = q < HW ? 16u * (unsigned)q : C8_OOB;
  const unsigned so = (unsigned)pl * HW16;
  u32x4 raw[4], rm;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < nsrc) raw[i] = c8_ld(c8_rsrc(src.s[i], b, planes, HW16), vo, so);        // nsrc is launch-uniform
  if (has_mask) rm = c8_ld(c8_rsrc(mask, b, planes, HW16), vo, so);
  float v[8], t[8];
  c8_unpack<E>(raw[0], v);
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (i < nsrc) {
      c8_unpack<E>(raw[i], t);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += t[j];
    }
  if (has_mask) {
    c8_unpack<E>(rm, t);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = t[j] > 0.f ? v[j] : 0.f;
  }
  c8_st(c8_pack<E>(v), c8_rsrc(dst, b, planes, HW16), vo, so);
}

int ew_sum_mask_c8(int B, int H, int W, int C, const codon_tensor* dst, int nsrc, const codon_tensor* const* srcs,
                   const codon_tensor* mask, int dtype, hipStream_t stream) {
  const long HW = (long)H * W;
  CODON_REQUIRE(B <= 65535 && C % 8 == 0 && nsrc >= 1 && nsrc <= 4, CODON_ERR_UNSUPPORTED,
                "ew_sum_mask: batch %d / channels %d / %d sources (1..4)", B, C, nsrc);
  CODON_REQUIRE(c8_slice_ok(dst->ctotal, dst->coff, C) && (!mask || c8_slice_ok(mask->ctotal, mask->coff, C)), CODON_ERR_BAD_ARG,
                "ew_sum_mask: 16-bit tensors are channel-blocked: ctotal / coff multiples of 8");
  CODON_REQUIRE(HW * 2 * C < (long)C8_OOB, CODON_ERR_UNSUPPORTED, "ew_sum_mask: image too large for 32-bit buffer offsets");
  const C8Slice d = c8_mk(dst, HW), m = mask ? c8_mk(mask, HW) : d;
  EwSum4 s4;
  for (int i = 0; i < 4; ++i) {
    const codon_tensor* t = i < nsrc ? srcs[i] : srcs[0];
    CODON_REQUIRE(t && t->data && c8_slice_ok(t->ctotal, t->coff, C), CODON_ERR_BAD_ARG, "ew_sum_mask: source %d", i);
    s4.s[i] = c8_mk(t, HW);
  }
  const dim3 grid((unsigned)((HW + 255) / 256), C / 8, B);
  if (dtype == CODON_F16)
    hipLaunchKernelGGL(ew_sum_mask_c8_kernel<C8F16>, grid, dim3(256), 0, stream, d, s4, nsrc, m, mask ? 1 : 0, C / 8, HW);
  else
    hipLaunchKernelGGL(ew_sum_mask_c8_kernel<C8Bf16>, grid, dim3(256), 0, stream, d, s4, nsrc, m, mask ? 1 : 0, C / 8, HW);
  return check_launch("ew_sum_mask_c8_kernel");
}

// ---- CAC backward, the two passes over the 64-channel tensors (math: cac_bwd.hip) --------------------------------------
// A: per-(b,c) partial sums of dL/dch, dL/dz per pixel, first-arg-max pixel candidates of both global max-pools
template <class E>
__global__ __launch_bounds__(256) void cac_bwd_reduce_c8_kernel(C8Slice g_out, C8Slice g_outc, C8Slice pre, C8Slice pre_c,
                                                                const float* __restrict__ ch, const float* __restrict__ sp,
                                                                const float* __restrict__ pools, float* __restrict__ g_z,
                                                                float* __restrict__ part_gch, int* __restrict__ part_arg,
                                                                long HW, int ntiles) {
  __shared__ float red_s[64][4];
  __shared__ int red_a[128][4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x, b = blockIdx.y;
  const long tile0 = (long)tile * EW_TILE;
  const unsigned HW16 = 16u * (unsigned)HW;
  bool ok[EW_NP];
  int pidx[EW_NP];
  unsigned vo[EW_NP];
  float spv[EW_NP], gsp[EW_NP];
#pragma unroll
  for (int k = 0; k < EW_NP; ++k) {
    const long q = tile0 + k * 256 + tid;
    ok[k] = q < HW;
    pidx[k] = (int)q;
    vo[k] = ok[k] ? 16u * (unsigned)q : C8_OOB;
    spv[k] = ok[k] ? sp[(long)b * HW + q] : 0.f;
    gsp[k] = 0.f;
  }
  const float* mx = pools + ((long)b * 2 + 1) * 128;
  const __amdgpu_buffer_rsrc_t r_go = c8_rsrc(g_out, b, 8, HW16), r_gc = c8_rsrc(g_outc, b, 8, HW16),
                               r_p = c8_rsrc(pre, b, 8, HW16), r_pc = c8_rsrc(pre_c, b, 8, HW16);
#pragma unroll 1
  for (int pl = 0; pl < 8; ++pl) {
    const unsigned so = (unsigned)pl * HW16;
    float chc[8], mxc[8], mxd[8], s8[8];
    int ad[8], ac[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      chc[j] = ch[b * 64 + pl * 8 + j];
      mxc[j] = mx[pl * 8 + j];            // Fcat order: colour c, depth 64 + c
      mxd[j] = mx[64 + pl * 8 + j];
      s8[j] = 0.f; ad[j] = INT_MAX; ac[j] = INT_MAX;
    }
#pragma unroll
    for (int k = 0; k < EW_NP; ++k) {
      float go[8], gc[8], p[8], pc[8];
      c8_unpack<E>(c8_ld(r_go, vo[k], so), go);
      c8_unpack<E>(c8_ld(r_gc, vo[k], so), gc);
      c8_unpack<E>(c8_ld(r_p, vo[k], so), p);
      c8_unpack<E>(c8_ld(r_pc, vo[k], so), pc);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float gg = go[j] * p[j] + gc[j] * pc[j];
        s8[j] = fmaf(gg, spv[k], s8[j]);
        gsp[k] = fmaf(gg, chc[j], gsp[k]);
        if (ok[k] && p[j] == mxd[j]) ad[j] = min(ad[j], pidx[k]);
        if (ok[k] && pc[j] == mxc[j]) ac[j] = min(ac[j], pidx[k]);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float s = c8_wsum(s8[j]);
      const int a_d = c8_wmin(ad[j]), a_c = c8_wmin(ac[j]);
      if (lane == 0) { red_s[pl * 8 + j][wave] = s; red_a[pl * 8 + j][wave] = a_c; red_a[64 + pl * 8 + j][wave] = a_d; }
    }
  }
  // dL/dz = dL/dsp * sp * (1 - sp)
#pragma unroll
  for (int k = 0; k < EW_NP; ++k)
    if (ok[k]) g_z[(long)b * HW + pidx[k]] = gsp[k] * spv[k] * (1.f - spv[k]);
  __syncthreads();
  if (tid < 64)
    part_gch[((long)b * ntiles + tile) * 64 + tid] = (red_s[tid][0] + red_s[tid][1]) + (red_s[tid][2] + red_s[tid][3]);
  if (tid < 128)
    part_arg[((long)b * ntiles + tile) * 128 + tid] =
        min(min(red_a[tid][0], red_a[tid][1]), min(red_a[tid][2], red_a[tid][3]));
}

// A, fused form (round 4, the CAC backward of a training step): the same pass ALSO (1) records, per pixel, the first
// arg-max channel of the channel max-pool in Fcat order (colour 0..63, depth 64..127 -- what torch.max(dim=1) routes its
// gradient to, CAC_module.py:81) and (2) folds dL/d(out) into the running dL/d(inputs) (`inputs` feeds every block:
// CODON_x4.py:90-91,117-118).  With those two, what is left of cac_bwd_apply -- forming dL/d(pre) -- moves into the
// staging of the 1x1 conv's backward that consumes it (conv_wgrad_c8.hip, GB) and the 12.6 GB apply pass is gone.
// Same tiles, same per-thread pixels, same summation order as cac_bwd_reduce_c8_kernel: g_z / part_gch / part_arg are
// bit-identical.  Six 16-byte vectors per (plane, pixel): the 8-pixel unroll of the plain pass would need 351 VGPRs (one
// wave per SIMD), so the pixels are walked four at a time and the per-pixel state lives in LDS.
template <class E>
__global__ __launch_bounds__(256, 2) void cac_bwd_reduce_acc_c8_kernel(
    C8Slice g_out, C8Slice g_outc, C8Slice pre, C8Slice pre_c, C8Slice g_in, C8Slice g_in_c, const float* __restrict__ ch,
    const float* __restrict__ sp, const float* __restrict__ pools, const float* __restrict__ pooled, float* __restrict__ g_z,
    float* __restrict__ part_gch, int* __restrict__ part_arg, int* __restrict__ argch, long HW, int ntiles, int accumulate_in) {
  constexpr int KH = 4;                               // pixels in flight per thread (8 needed 351 VGPRs)
  __shared__ float red_s[64][4];
  __shared__ int red_a[128][4];
  __shared__ float st_sp[EW_NP][256], st_gsp[EW_NP][256], st_pmx[EW_NP][256];
  __shared__ unsigned st_acd[EW_NP][256];             // (first colour channel | first depth channel << 8) equal to the channel max
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x, b = blockIdx.y;
  const long tile0 = (long)tile * EW_TILE;
  const unsigned HW16 = 16u * (unsigned)HW;
#pragma unroll
  for (int k = 0; k < EW_NP; ++k) {
    const long q = tile0 + k * 256 + tid;
    const bool in = q < HW;
    st_sp[k][tid] = in ? sp[(long)b * HW + q] : 0.f;
    st_gsp[k][tid] = 0.f;
    st_pmx[k][tid] = in ? pooled[(long)b * 2 * HW + q] : 0.f;
    st_acd[k][tid] = 0xFFFFu;
  }
  const float* mx = pools + ((long)b * 2 + 1) * 128;
  const __amdgpu_buffer_rsrc_t r_go = c8_rsrc(g_out, b, 8, HW16), r_gc = c8_rsrc(g_outc, b, 8, HW16),
                               r_p = c8_rsrc(pre, b, 8, HW16), r_pc = c8_rsrc(pre_c, b, 8, HW16),
                               r_gi = c8_rsrc(g_in, b, 8, HW16), r_gic = c8_rsrc(g_in_c, b, 8, HW16);
#pragma unroll 1
  for (int pl = 0; pl < 8; ++pl) {
    const unsigned so = (unsigned)pl * HW16;
    float chc[8], mxc[8], mxd[8], s8[8];
    int ad[8], ac[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      chc[j] = ch[b * 64 + pl * 8 + j];
      mxc[j] = mx[pl * 8 + j];            // Fcat order: colour c, depth 64 + c
      mxd[j] = mx[64 + pl * 8 + j];
      s8[j] = 0.f; ad[j] = INT_MAX; ac[j] = INT_MAX;
    }
#pragma unroll 1
    for (int k0 = 0; k0 < EW_NP; k0 += KH) {
      u32x4 qgo[KH], qgc[KH], qp[KH], qpc[KH], qi[KH], qic[KH];
      unsigned vo[KH];
#pragma unroll
      for (int kk = 0; kk < KH; ++kk) {
        const long q = tile0 + (k0 + kk) * 256 + tid;
        vo[kk] = q < HW ? 16u * (unsigned)q : C8_OOB;
        qgo[kk] = c8_ld(r_go, vo[kk], so);
        qgc[kk] = c8_ld(r_gc, vo[kk], so);
        qp[kk] = c8_ld(r_p, vo[kk], so);
        qpc[kk] = c8_ld(r_pc, vo[kk], so);
        if (accumulate_in == 1) { qi[kk] = c8_ld(r_gi, vo[kk], so); qic[kk] = c8_ld(r_gic, vo[kk], so); }
      }
#pragma unroll
      for (int kk = 0; kk < KH; ++kk) {
        const int k = k0 + kk;
        const int pix = (int)tile0 + k * 256 + tid;
        const bool in = vo[kk] != C8_OOB;
        const float spv = st_sp[k][tid], pmx = st_pmx[k][tid];
        float gsp = st_gsp[k][tid];
        unsigned acd = st_acd[k][tid];
        float go[8], gc[8], p[8], pc[8];
        c8_unpack<E>(qgo[kk], go);
        c8_unpack<E>(qgc[kk], gc);
        c8_unpack<E>(qp[kk], p);
        c8_unpack<E>(qpc[kk], pc);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float gg = go[j] * p[j] + gc[j] * pc[j];
          s8[j] = fmaf(gg, spv, s8[j]);
          gsp = fmaf(gg, chc[j], gsp);
          if (in && p[j] == mxd[j]) ad[j] = min(ad[j], pix);
          if (in && pc[j] == mxc[j]) ac[j] = min(ac[j], pix);
          // planes are walked in ascending channel order: the first match of each stream is kept
          if (pc[j] == pmx && (acd & 0xFFu) == 0xFFu) acd = (acd & 0xFF00u) | (unsigned)(pl * 8 + j);
          if (p[j] == pmx && (acd >> 8) == 0xFFu) acd = (acd & 0xFFu) | ((unsigned)(pl * 8 + j) << 8);
        }
        st_gsp[k][tid] = gsp;
        st_acd[k][tid] = acd;
        if (accumulate_in == 1) {          // g_inputs (+)= g_out, the arithmetic of cac_bwd_apply_c8_kernel
          float gi[8], gic[8];
          c8_unpack<E>(qi[kk], gi);
          c8_unpack<E>(qic[kk], gic);
#pragma unroll
          for (int j = 0; j < 8; ++j) { gi[j] = gi[j] + go[j]; gic[j] = gic[j] + gc[j]; }
          c8_st(c8_pack<E>(gi), r_gi, vo[kk], so);
          c8_st(c8_pack<E>(gic), r_gic, vo[kk], so);
        } else if (accumulate_in == 0) {
          c8_st(qgo[kk], r_gi, vo[kk], so);
          c8_st(qgc[kk], r_gic, vo[kk], so);
        }                                  // 2: dL/d(inputs) already holds this block's dL/d(out) (codon_conv2d_sum_into_fwd)
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float s = c8_wsum(s8[j]);
      const int a_d = c8_wmin(ad[j]), a_c = c8_wmin(ac[j]);
      if (lane == 0) { red_s[pl * 8 + j][wave] = s; red_a[pl * 8 + j][wave] = a_c; red_a[64 + pl * 8 + j][wave] = a_d; }
    }
  }
  // dL/dz = dL/dsp * sp * (1 - sp); the pixel's arg-max channel
#pragma unroll
  for (int k = 0; k < EW_NP; ++k) {
    const long q = tile0 + k * 256 + tid;
    if (q < HW) {
      const float spv = st_sp[k][tid];
      g_z[(long)b * HW + q] = st_gsp[k][tid] * spv * (1.f - spv);
      const unsigned acd = st_acd[k][tid];
      const int ac_ = (int)(acd & 0xFFu), ad_ = (int)(acd >> 8);
      argch[(long)b * HW + q] = ac_ != 255 ? ac_ : (ad_ != 255 ? 64 + ad_ : 255);
    }
  }
  __syncthreads();
  if (tid < 64)
    part_gch[((long)b * ntiles + tile) * 64 + tid] = (red_s[tid][0] + red_s[tid][1]) + (red_s[tid][2] + red_s[tid][3]);
  if (tid < 128)
    part_arg[((long)b * ntiles + tile) * 128 + tid] =
        min(min(red_a[tid][0], red_a[tid][1]), min(red_a[tid][2], red_a[tid][3]));
}

int cac_bwd_reduce_c8(int B, int H, int W, const codon_tensor* g_out, const codon_tensor* g_outc, const codon_tensor* pre,
                      const codon_tensor* pre_c, const float* ch, const float* sp, const float* pools, float* g_z,
                      float* part_gch, int* part_arg, int dtype, hipStream_t stream, const float* pooled, int* argch,
                      const codon_tensor* g_in, const codon_tensor* g_in_c, int accumulate_in) {
  const long HW = (long)H * W;
  const int nt = (int)((HW + EW_TILE - 1) / EW_TILE);
  for (const codon_tensor* t : {g_out, g_outc, pre, pre_c})
    CODON_REQUIRE(c8_slice_ok(t->ctotal, t->coff, 64), CODON_ERR_BAD_ARG,
                  "cac_bwd_reduce: 16-bit tensors are channel-blocked: ctotal / coff multiples of 8");
  CODON_REQUIRE(HW * 2 * 64 < (long)C8_OOB, CODON_ERR_UNSUPPORTED, "cac_bwd_reduce: image too large for 32-bit buffer offsets");
  if (argch != nullptr) {
    CODON_REQUIRE(pooled && g_in && g_in_c && c8_slice_ok(g_in->ctotal, g_in->coff, 64) &&
                      c8_slice_ok(g_in_c->ctotal, g_in_c->coff, 64),
                  CODON_ERR_BAD_ARG, "cac_bwd_reduce_acc: pooled / g_in / g_in_c");
    if (dtype == CODON_F16)
      hipLaunchKernelGGL(cac_bwd_reduce_acc_c8_kernel<C8F16>, dim3(nt, B), dim3(256), 0, stream, c8_mk(g_out, HW),
                         c8_mk(g_outc, HW), c8_mk(pre, HW), c8_mk(pre_c, HW), c8_mk(g_in, HW), c8_mk(g_in_c, HW), ch, sp, pools,
                         pooled, g_z, part_gch, part_arg, argch, HW, nt, accumulate_in);
    else
      hipLaunchKernelGGL(cac_bwd_reduce_acc_c8_kernel<C8Bf16>, dim3(nt, B), dim3(256), 0, stream, c8_mk(g_out, HW),
                         c8_mk(g_outc, HW), c8_mk(pre, HW), c8_mk(pre_c, HW), c8_mk(g_in, HW), c8_mk(g_in_c, HW), ch, sp, pools,
                         pooled, g_z, part_gch, part_arg, argch, HW, nt, accumulate_in);
    return check_launch("cac_bwd_reduce_acc_c8_kernel");
  }
  if (dtype == CODON_F16)
    hipLaunchKernelGGL(cac_bwd_reduce_c8_kernel<C8F16>, dim3(nt, B), dim3(256), 0, stream, c8_mk(g_out, HW), c8_mk(g_outc, HW),
                       c8_mk(pre, HW), c8_mk(pre_c, HW), ch, sp, pools, g_z, part_gch, part_arg, HW, nt);
  else
    hipLaunchKernelGGL(cac_bwd_reduce_c8_kernel<C8Bf16>, dim3(nt, B), dim3(256), 0, stream, c8_mk(g_out, HW), c8_mk(g_outc, HW),
                       c8_mk(pre, HW), c8_mk(pre_c, HW), ch, sp, pools, g_z, part_gch, part_arg, HW, nt);
  return check_launch("cac_bwd_reduce_c8_kernel");
}

// D: g_pre / g_pre_c (direct term + avg-pool broadcast + max-pool routing + channel-mean broadcast + channel-max
// routing to the FIRST arg-max channel in Fcat order, as torch.max does) and g_inputs (+)= g_out
struct BwdApplyC8 {
  C8Slice g_out, pre, g_pre, g_in;
};
template <class E>
__global__ __launch_bounds__(256) void cac_bwd_apply_c8_kernel(const BwdApplyC8 sd, const BwdApplyC8 sc,
                                                               const float* __restrict__ ch, const float* __restrict__ sp,
                                                               const float* __restrict__ pooled,
                                                               const float* __restrict__ g_pooled,
                                                               const float* __restrict__ g_pools,
                                                               const int* __restrict__ argpix, int accumulate_in, long HW,
                                                               float inv_hw) {
  constexpr int NP = 2;
  const int tid = threadIdx.x, b = blockIdx.y;
  const long tile0 = (long)blockIdx.x * (256 * NP);
  const unsigned HW16 = 16u * (unsigned)HW;
  int pidx[NP];
  unsigned vo[NP];
  float spv[NP], pmax[NP], gpmax[NP], gpmean[NP];
  bool done[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const long q = tile0 + k * 256 + tid;
    const bool ok = q < HW;
    const long qq = ok ? q : 0;
    pidx[k] = ok ? (int)q : -1;
    vo[k] = ok ? 16u * (unsigned)q : C8_OOB;
    spv[k] = sp[(long)b * HW + qq];
    pmax[k] = pooled[(long)b * 2 * HW + qq];
    gpmax[k] = g_pooled[(long)b * 2 * HW + qq];
    gpmean[k] = g_pooled[(long)b * 2 * HW + HW + qq] * (1.f / 128.f);
    done[k] = false;
  }
  const float* gavg = g_pools + ((long)b * 2 + 0) * 128;
  const float* gmax = g_pools + ((long)b * 2 + 1) * 128;
  const int* ap = argpix + (long)b * 128;
#pragma unroll 1
  for (int st = 0; st < 2; ++st) {            // Fcat order: colour first, so ties route like torch.max(dim=1)
    const BwdApplyC8& s = st ? sd : sc;
    const int fbase = st ? 64 : 0;
    const __amdgpu_buffer_rsrc_t r_go = c8_rsrc(s.g_out, b, 8, HW16), r_p = c8_rsrc(s.pre, b, 8, HW16),
                                 r_gp = c8_rsrc(s.g_pre, b, 8, HW16), r_gi = c8_rsrc(s.g_in, b, 8, HW16);
#pragma unroll 1
    for (int pl = 0; pl < 8; ++pl) {
      const unsigned so = (unsigned)pl * HW16;
      float chc[8], ga[8], gm[8];
      int apx[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        chc[j] = ch[b * 64 + pl * 8 + j];
        ga[j] = gavg[fbase + pl * 8 + j] * inv_hw;
        gm[j] = gmax[fbase + pl * 8 + j];
        apx[j] = ap[fbase + pl * 8 + j];
      }
      u32x4 qgo[NP], qp[NP], qgi[NP];
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        qgo[k] = c8_ld(r_go, vo[k], so);
        qp[k] = c8_ld(r_p, vo[k], so);
        if (accumulate_in) qgi[k] = c8_ld(r_gi, vo[k], so);
      }
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        float go[8], p[8], gi[8], o[8];
        c8_unpack<E>(qgo[k], go);
        c8_unpack<E>(qp[k], p);
        if (accumulate_in) c8_unpack<E>(qgi[k], gi);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float v = go[j] * (chc[j] * spv[k]) + ga[j] + gpmean[k];
          if (pidx[k] == apx[j]) v += gm[j];
          const bool hit = !done[k] && p[j] == pmax[k];
          if (hit) v += gpmax[k];
          done[k] = done[k] || hit;
          o[j] = v;
          gi[j] = accumulate_in ? gi[j] + go[j] : go[j];
        }
        c8_st(c8_pack<E>(o), r_gp, vo[k], so);
        c8_st(c8_pack<E>(gi), r_gi, vo[k], so);
      }
    }
  }
}

int cac_bwd_apply_c8(int B, int H, int W, const codon_tensor* g_out, const codon_tensor* g_outc, const codon_tensor* pre,
                     const codon_tensor* pre_c, const float* ch, const float* sp, const float* pooled, const float* g_pooled,
                     const float* g_pools, const int* argpix, const codon_tensor* g_pre, const codon_tensor* g_pre_c,
                     const codon_tensor* g_in, const codon_tensor* g_in_c, int accumulate_in, int dtype, hipStream_t stream) {
  const long HW = (long)H * W;
  for (const codon_tensor* t : {g_out, g_outc, pre, pre_c, g_pre, g_pre_c, g_in, g_in_c})
    CODON_REQUIRE(c8_slice_ok(t->ctotal, t->coff, 64), CODON_ERR_BAD_ARG,
                  "cac_bwd_apply: 16-bit tensors are channel-blocked: ctotal / coff multiples of 8");
  CODON_REQUIRE(HW * 2 * 64 < (long)C8_OOB, CODON_ERR_UNSUPPORTED, "cac_bwd_apply: image too large for 32-bit buffer offsets");
  const float inv = (float)(1.0 / (double)HW);
  const BwdApplyC8 sd{c8_mk(g_out, HW), c8_mk(pre, HW), c8_mk(g_pre, HW), c8_mk(g_in, HW)};
  const BwdApplyC8 sc{c8_mk(g_outc, HW), c8_mk(pre_c, HW), c8_mk(g_pre_c, HW), c8_mk(g_in_c, HW)};
  const dim3 grid((unsigned)((HW + 511) / 512), B);
  if (dtype == CODON_F16)
    hipLaunchKernelGGL(cac_bwd_apply_c8_kernel<C8F16>, grid, dim3(256), 0, stream, sd, sc, ch, sp, pooled, g_pooled, g_pools,
                       argpix, accumulate_in, HW, inv);
  else
    hipLaunchKernelGGL(cac_bwd_apply_c8_kernel<C8Bf16>, grid, dim3(256), 0, stream, sd, sc, ch, sp, pooled, g_pooled, g_pools,
                       argpix, accumulate_in, HW, inv);
  return check_launch("cac_bwd_apply_c8_kernel");
}

}  // namespace codon
