// fp32 weight gradient, second generation: v_mfma_f32_16x16x4_f32 tiles, 8 balanced waves, double-buffered LDS.
//
//   dW[co][ci][dy][dx] = sum_{b,h,w} gy[b,co,h,w] * x[b,ci,h+dy-p,w+dx-p]        (autograd of nn.Conv2d,
//   /root/reference/CODON_X4/CODON_x4.py:24-47; the reference has no explicit backward -- SURVEY.md 3.4)
//
// conv_wgrad_f32.hip (round 1, 74 TF = 47 % of the fp32 MFMA peak) lost its time in three places: the 25 taps of a
// 5x5 filter as 32x32 tiles spread over 4 waves (7 slots each, 3 of 28 wasted), a single-buffered tile whose staging
// (with ~20 VALU of index arithmetic per element) never overlapped the MFMAs, and two barriers per tile.  Here:
//   * 16x16x4 tiles: a workgroup's 64 cout x 32 cin block is 4 x 2 tile columns, one per wave (8 waves, 2 per SIMD), and
//     every wave owns ALL KS*KS taps of its (16 cout, 16 cin) pair: KS*KS accumulators of 4 registers (100 VGPRs for 5x5),
//     identical work per wave whatever KS is;
//       A (16 x 4): lane l holds gy[co = l & 15][pixel k0 + (l >> 4)]
//       B (4 x 16): lane l holds x [ci = l & 15][pixel k0 + (l >> 4) + tap]
//       D (16 x 16): lane holds ci = l & 15, co = 4 (l >> 4) + register
//     one A fetch + KS*KS B fetches (ds_read_b32, `base + immediate`) per KS*KS MFMAs of 32 cycles: no VALU in the loop;
//   * channel-per-lane LDS tiles with a plane stride of 2 (mod 32) words: the 32 lanes of a ds_read_b32 pass
//     (16 channels x 2 pixels) hit 32 distinct banks;
//   * staging is a straight copy of 16-byte row chunks (buffer loads, tile origin in the descriptor base, offsets
//     hoisted, out-of-image chunks out of range), written with ds_write_b64; the next tile is requested before the
//     current tile's MFMAs and written after them; one barrier per tile.
// Needs W % 4 == 0 and 16-byte aligned slices (the launcher falls back to conv_wgrad_f32.hip otherwise); k in {1, 3, 5}.
// Partials -> workspace[split][tap][co][ci], summed in fixed order by wgrad_reduce_kernel: deterministic.

#include <type_traits>

#include "codon_common.h"

namespace codon {

typedef unsigned t16_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned t16_u32x2 __attribute__((ext_vector_type(2)));

typedef const volatile __attribute__((address_space(3))) float* lds_rf;

template <int N, class F, int I = 0>
__device__ __forceinline__ void t16_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    t16_static_for<N, F, I + 1>(static_cast<F&&>(f));
  }
}

struct WgradT16Params {
  const float* x;
  const float* gy;
  float* ws;  // [nsplit][taps][cout][cin]
  int H, W, cin, cout;
  long x_img, g_img, x_base, g_base;
  int tiles_x, nbands, nsplit;
};

// k = 1 (confuse*: HBM-bound -- 768 B per pixel for 16 K MACs): the workgroup covers ALL 128 cin (4 cin tiles per wave,
// waves = 4 cout tiles x 2 cin halves) so gy is read once, not once per 32-cin block, in 2-row tiles (double-buffered
// 101 KB).  Round 1's kernel moved 2 x the bytes single-buffered: 24 TF = 1.15 TB/s.
template <int KS>
__global__ __launch_bounds__(512, 2) void conv_wgrad_f32_t16_kernel(const WgradT16Params p) {
  constexpr int PAD = KS / 2, TAPS = KS * KS;
  constexpr int NCIT = KS == 1 ? 4 : 1;          // 16-cin tiles per wave
  constexpr int NCI = 32 * NCIT;                 // cin per workgroup
  constexpr int NACC = TAPS * NCIT;
  constexpr int TW = 32, TH = KS == 1 ? 2 : 4;
  constexpr int XL = KS == 1 ? 0 : 4;            // left / right margin: the tile's column origin tx0 - 4 is 16-byte aligned
  constexpr int XC = TW + 2 * XL, XR = TH + KS - 1;
  constexpr int XPL = ((XR * XC + 29) / 32) * 32 + 2;   // words per channel plane, == 2 (mod 32), >= XR * XC
  constexpr int GPL = TH * TW + 2;                      // == 2 (mod 32)
  static_assert(XPL >= XR * XC && XPL % 32 == 2 && GPL % 32 == 2, "plane strides");
  constexpr int XW = NCI * XPL, GW = 64 * GPL;   // words per buffer
  constexpr int NXC = NCI * XR * (XC / 4), NGC = 64 * TH * (TW / 4);   // 16-byte chunks per tile
  constexpr int NT = 512;
  constexpr int XE = (NXC + NT - 1) / NT, GE = NGC / NT;
  static_assert(NGC % NT == 0, "gy tile is a whole number of staging rounds");
  constexpr unsigned OOB = 0xFFFFFFF0u;

  __shared__ __attribute__((aligned(16))) float lds[2 * (XW + GW)];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  const int H = p.H, W = p.W;
  const long HW = (long)H * W;
  const unsigned HW4 = 4u * (unsigned)H * (unsigned)W;

  const int nci_b = p.cin / NCI;
  // workgroups are dealt round-robin over the 8 XCDs in dispatch order (x fastest): left alone, the channel blocks that
  // read the SAME image band (same blockIdx.y) land on 8 different XCDs and each L2 fetches the band again.  The bijective
  // remap gives every XCD a contiguous range of (band, channel block) pairs, so a band's channel blocks share one L2.
  const unsigned vb_ = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const int bx_ = (int)(vb_ % gridDim.x), by_ = (int)(vb_ / gridDim.x);   // bf16 5x5-128: 8.16 -> 7.90 ms (A/B, same box)
  const int cob = bx_ / nci_b, cib = bx_ % nci_b;       // 64-cout block, 32-cin block
  const int split = by_;
  const int b = split / p.nbands, band = split % p.nbands;
  const int tiles_y = (H + TH - 1) / TH;
  const int ty_begin = (int)((long)band * tiles_y / p.nbands);        // tile rows spread evenly over the bands
  const int ty_end = (int)((long)(band + 1) * tiles_y / p.nbands);
  const int ntile = (ty_end - ty_begin) * p.tiles_x;

  const float* const xg = p.x + b * p.x_img + p.x_base + (long)cib * NCI * HW;
  const float* const gg = p.gy + b * p.g_img + p.g_base + (long)cob * 64 * HW;

  // staging plan (tile independent): chunk e = tid + NT k -> (channel, row, 4-pixel column chunk)
  unsigned xrel[XE], grel[GE];
  int xrc[XE], grc[GE];
  int xlds[XE], glds[GE];
#pragma unroll
  for (int k = 0; k < XE; ++k) {
    const int e = tid + k * NT;
    const int ch = e % (XC / 4), r = (e / (XC / 4)) % XR, c = e / ((XC / 4) * XR);
    const bool in = (NXC % NT == 0) || e < NXC;
    xrel[k] = (unsigned)c * HW4 + 4u * (unsigned)(r * W + ch * 4);
    xrc[k] = in ? ((r << 8) | ch) : -1;
    xlds[k] = in ? c * XPL + r * XC + ch * 4 : 0;
  }
#pragma unroll
  for (int k = 0; k < GE; ++k) {
    const int e = tid + k * NT;
    const int ch = e % (TW / 4), r = (e / (TW / 4)) % TH, c = e / ((TW / 4) * TH);
    grel[k] = (unsigned)c * HW4 + 4u * (unsigned)(r * W + ch * 4);
    grc[k] = (r << 8) | ch;
    glds[k] = c * GPL + r * TW + ch * 4;
  }

  const int co16 = wave & 3, ci16 = wave >> 2;            // k = 1: ci16 = which 64-cin half
  const int a_lane = (co16 * 16 + l15) * GPL + kq;
  const int b_lane = (ci16 * 16 * NCIT + l15) * XPL + kq + (XL - PAD);

  f32x4 acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;

  t16_u32x4 xv[XE], gv[GE];
  auto load_tile = [&](int t) {
    const int ty = ty_begin + t / p.tiles_x, tx = t % p.tiles_x;
    const int tx0 = tx * TW, ty0 = ty * TH;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(xg + ((long)(ty0 - PAD) * W + (tx0 - XL))), 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(gg + ((long)ty0 * W + tx0)), 0, (int)OOB, 0x00020000);
#pragma unroll
    for (int k = 0; k < XE; ++k) {
      const int gy_ = ty0 - PAD + (xrc[k] >> 8), gx_ = tx0 - XL + 4 * (xrc[k] & 255);
      const bool ok = xrc[k] >= 0 && gy_ >= 0 && gy_ < H && gx_ >= 0 && gx_ < W;   // W % 4 == 0: a chunk is in or out
      const auto v = __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? xrel[k] : OOB, 0, 0);
      xv[k] = *reinterpret_cast<const t16_u32x4*>(&v);
    }
#pragma unroll
    for (int k = 0; k < GE; ++k) {
      const int gy_ = ty0 + (grc[k] >> 8), gx_ = tx0 + 4 * (grc[k] & 255);
      const bool ok = gy_ < H && gx_ < W;
      const auto v = __builtin_amdgcn_raw_buffer_load_b128(gr, ok ? grel[k] : OOB, 0, 0);
      gv[k] = *reinterpret_cast<const t16_u32x4*>(&v);
    }
  };
  auto store_tile = [&](int buf) {
    float* xs = lds + buf * (XW + GW);
    float* gs = xs + XW;
#pragma unroll
    for (int k = 0; k < XE; ++k)
      if ((NXC % NT == 0) || tid + k * NT < NXC) {      // plane strides are even: 8-byte aligned halves
        *reinterpret_cast<t16_u32x2*>(xs + xlds[k]) = t16_u32x2{xv[k][0], xv[k][1]};
        *reinterpret_cast<t16_u32x2*>(xs + xlds[k] + 2) = t16_u32x2{xv[k][2], xv[k][3]};
      }
#pragma unroll
    for (int k = 0; k < GE; ++k) {
      *reinterpret_cast<t16_u32x2*>(gs + glds[k]) = t16_u32x2{gv[k][0], gv[k][1]};
      *reinterpret_cast<t16_u32x2*>(gs + glds[k] + 2) = t16_u32x2{gv[k][2], gv[k][3]};
    }
  };

  if (ntile > 0) {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();

#pragma unroll 1
  for (int t = 0; t < ntile; ++t) {
    const bool has_next = t + 1 < ntile;
    if (has_next) load_tile(t + 1);

    const float* xs = lds + (t & 1) * (XW + GW);
    const float* gs = xs + XW;
    const lds_rf ap = (lds_rf)(gs + a_lane);
    const lds_rf bp = (lds_rf)(xs + b_lane);
    // K loop: 4 horizontally adjacent pixels per MFMA; every fragment address is `lane base + immediate`.  The
    // fragments of k-step s + 1 (1 + KS*KS volatile ds_read_b32, issued back to back) are requested before the KS*KS
    // MFMAs of k-step s; sched_barrier keeps that order (left alone, hipcc hoists hundreds of reads and spills).
    float a[2], bv[2][NACC];
#define T16_FETCH(set_, r_, q_)                                                           \
    {                                                                                     \
      a[set_] = ap[(r_) * TW + (q_)];                                                     \
      _Pragma("unroll") for (int ct = 0; ct < NCIT; ++ct)                                 \
        _Pragma("unroll") for (int dy = 0; dy < KS; ++dy)                                 \
          _Pragma("unroll") for (int dx = 0; dx < KS; ++dx)                               \
            bv[set_][ct * TAPS + dy * KS + dx] = bp[ct * 16 * XPL + ((r_) + dy) * XC + (q_) + dx]; \
    }
    T16_FETCH(0, 0, 0)
    t16_static_for<TH * (TW / 4)>([&](auto sc) {
      constexpr int s_ = decltype(sc)::value;
      constexpr int cur = s_ & 1;
      if constexpr (s_ + 1 < TH * (TW / 4)) T16_FETCH(cur ^ 1, (s_ + 1) / (TW / 4), ((s_ + 1) % (TW / 4)) * 4)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur], bv[cur][j], acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    });
#undef T16_FETCH
    if (has_next) store_tile((t + 1) & 1);
    __syncthreads();
  }

  // partial dW -> workspace[split][tap][co][ci]: lane = ci column, 4 consecutive co rows in registers
  float* __restrict__ wsp = p.ws + (long)split * TAPS * p.cout * p.cin;
#pragma unroll
  for (int ct = 0; ct < NCIT; ++ct) {
    const int ci = cib * NCI + (ci16 * NCIT + ct) * 16 + l15;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = cob * 64 + co16 * 16 + 4 * kq + r;
        wsp[((long)tap * p.cout + co) * p.cin + ci] = acc[ct * TAPS + tap][r];
      }
  }
}

bool conv_wgrad_f32_t16_shape(const codon_conv_desc* d) {
  if (d->ksize == 1 && d->cin % 128 != 0) return false;
  return (d->ksize == 1 || d->ksize == 3 || d->ksize == 5) && d->width % 4 == 0 && d->cout % 64 == 0 && d->cin % 32 == 0 &&
         (long)d->height * d->width * 4 * 128 < 0xFFFFFFF0L;         // 32-bit chunk offsets: up to 128 planes of a slice
}

bool conv_wgrad_f32_t16_supported(const codon_conv_desc* d, const void* x, const void* gy) {
  if (!conv_wgrad_f32_t16_shape(d)) return false;
  const long HW = (long)d->height * d->width;
  // every chunk address = slice start + 4 * (plane * HW + row * W + 4 k): 16-byte aligned iff the slice start is
  const uintptr_t xa = reinterpret_cast<uintptr_t>(x) + 4 * (uintptr_t)(d->x_coff * HW);
  const uintptr_t ga = reinterpret_cast<uintptr_t>(gy) + 4 * (uintptr_t)(d->y_coff * HW);
  return (xa % 16) == 0 && (ga % 16) == 0 && ((d->x_ctotal * HW * 4) % 16) == 0 && ((d->y_ctotal * HW * 4) % 16) == 0;
}

int launch_wgrad_f32_t16(const codon_conv_desc* d, const float* x, const float* gy, float* workspace, int nbands,
                         int nsplit, hipStream_t stream) {
  const long HW = (long)d->height * d->width;
  WgradT16Params p;
  p.x = x; p.gy = gy; p.ws = workspace;
  p.H = d->height; p.W = d->width; p.cin = d->cin; p.cout = d->cout;
  p.x_img = d->x_ctotal * HW; p.g_img = d->y_ctotal * HW;
  p.x_base = d->x_coff * HW; p.g_base = d->y_coff * HW;
  p.tiles_x = (d->width + 31) / 32; p.nbands = nbands; p.nsplit = nsplit;
  const dim3 grid((d->cout / 64) * (d->cin / (d->ksize == 1 ? 128 : 32)), nsplit);
  if (d->ksize == 5) hipLaunchKernelGGL(conv_wgrad_f32_t16_kernel<5>, grid, dim3(512), 0, stream, p);
  else if (d->ksize == 3) hipLaunchKernelGGL(conv_wgrad_f32_t16_kernel<3>, grid, dim3(512), 0, stream, p);
  else hipLaunchKernelGGL(conv_wgrad_f32_t16_kernel<1>, grid, dim3(512), 0, stream, p);
  return check_launch("conv_wgrad_f32_t16_kernel");
}

}  // namespace codon
