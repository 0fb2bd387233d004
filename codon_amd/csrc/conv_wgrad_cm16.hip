// 16-bit weight gradient with CHANNEL-MAJOR LDS tiles (v_mfma_f32_32x32x16_{bf16,f16}, fp32 accumulate, fp32 dW).
//
//   dW[co][ci][dy][dx] = sum_{b,h,w} gy[b,co,h,w] * x[b,ci,h+dy-p,w+dx-p]        (autograd of nn.Conv2d)
//
// GEMM: M = cout, N = cin (per tap), K = pixels; one MFMA consumes 16 pixels of one image row, 8 per lane:
//   A (32 x 16): lane l holds gy[co = l&31][pix 8h .. 8h+7]
//   B (16 x 32): lane l holds x [ci = l&31][pix 8h + dx - p .. + 7]
// conv_wgrad_bf16.hip fetches every tap's B fragment separately from a PIXEL-major x tile with transposing reads.
// Measured there (ablation, conv5x5 128->128, 9.99 ms): MFMAs + operand reads alone 5.71 ms, staging alone 6.37 ms,
// and the two barely overlap -- the transposing layout costs 15 four-byte loads, ~220 VALU address ops and 23
// bank-conflicted 2/4-byte LDS writes per thread per 128-pixel tile.  Here instead:
//   * both tiles stay channel-major, exactly as in HBM, so staging is a straight copy: 16-byte buffer loads (offsets
//     hoisted, tile origin in the descriptor base = SALU) and conflict-free ds_write_b128 -- 5 + 5 per thread per tile;
//   * a wave owns one filter ROW dy (and one 32-cout tile).  The KS shifted B fragments of a k-step overlap in all but
//     KS-1 pixels, so the lane reads ONE aligned 16-pixel window of its channel (ds_read_b64 + b128 + b64) and derives
//     the KS fragments in registers: even shift = a register offset, odd shift = four v_alignbit_b32 (VALU is nearly
//     free beside 16-bit MFMAs).  LDS bytes per MFMA: (1 KB A + 2 KB window) / KS = 614 B for 5x5 (was 1 170 B).
// Needs W % 8 == 0 and 16-byte aligned slices (the launcher falls back to conv_wgrad_bf16.hip otherwise).
// k = 1 (confuse*): no margins, no shifts; see KSPLIT below.
//
// Workgroup = 2*KS waves = (2 cout tiles) x (KS filter rows): 64 cout x 32 cin x all taps, streaming an image band in
// CM_TH x 32 pixel tiles, LDS double-buffered.  Partials -> workspace[split][tap][co][ci] (fp32), summed in fixed order by
// wgrad_reduce_kernel: deterministic.

#include "codon_common.h"

namespace codon {

typedef __bf16 cm_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 cm_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned cm_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned cm_u32x2 __attribute__((ext_vector_type(2)));
struct CmBf16 {
  typedef cm_bf16x8 vec8;
  __device__ static f32x16 mfma(vec8 a, vec8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
struct CmF16 {
  typedef cm_f16x8 vec8;
  __device__ static f32x16 mfma(vec8 a, vec8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

#ifndef CODON_CM_TH
#define CODON_CM_TH 4
#endif
constexpr int CM_TH = CODON_CM_TH;
constexpr int CM_CIB1 = 4;          // k = 1: 128 cin per workgroup   // tile rows; the launcher's bands are whole multiples of 8 rows

struct WgradCmParams {
  const unsigned short* x;
  const unsigned short* gy;
  float* ws;
  int H, W, cin, cout;
  long x_img, g_img, x_base, g_base;
  int tiles_x, band_tiles_y, nbands, nsplit;
};

// KS == 1 has a single "filter row": its 8 waves are (2 cout tiles) x (TH tile rows) instead -- each wave takes the
// k-steps of one tile row, and the TH partial accumulators are summed through LDS once, after the band.
template <class E, int KS>
__global__ __launch_bounds__(KS == 1 ? 2 * CM_TH * 64 : 2 * KS * 64, 3) void conv_wgrad_cm16_kernel(const WgradCmParams p) {
  typedef typename E::vec8 vec8;
  typedef const __attribute__((address_space(3))) cm_u32x4* lds_r128;
  typedef const __attribute__((address_space(3))) cm_u32x2* lds_r64;
  constexpr bool KSPLIT = (KS == 1);
  constexpr int CIB = KSPLIT ? CM_CIB1 : 1;      // 32-cin tiles per workgroup (k = 1 is HBM-bound: gy is re-read per cin block)
  constexpr int NT = KSPLIT ? 2 * CM_TH * 64 : 2 * KS * 64;
  constexpr int PAD = KS / 2;
  constexpr int TW = 32, TH = CM_TH;
  constexpr int XL = KSPLIT ? 0 : 8;           // left margin of the x tile: its column origin tx0 - 8 is 16-byte aligned
  constexpr int XC = XL + TW + (KSPLIT ? 0 : 8);   // 48 columns = 6 chunks of 8 pixels (k = 1: the 32 tile columns)
  constexpr int XR = TH + KS - 1;
  constexpr int XROW = XC * 2;                 // bytes per tile row of one channel
  constexpr int XPITCH = XR * XROW + 16;       // bytes per channel: +16 -> 4 banks per lane step (conflict-free b128 / b64)
  constexpr int GPITCH = TH * TW * 2 + 16;     // bytes per cout row of the gy tile
  constexpr int XBYTES = 32 * CIB * XPITCH, GBYTES = 64 * GPITCH;
  constexpr int NXC = 32 * CIB * XR * (XC / 8);      // 16-byte chunks of the x tile
  constexpr int NGC = 64 * TH * (TW / 8);
  constexpr int XE = (NXC + NT - 1) / NT, GE = (NGC + NT - 1) / NT;
  constexpr int NK = TH * (TW / 16);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  static_assert((XBYTES % 16) == 0 && (GBYTES % 16) == 0, "tile buffers are 16-byte multiples");

  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * (XBYTES + GBYTES)];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int H = p.H, W = p.W;
  const long HW = (long)H * W;
  const unsigned HW2 = 2u * (unsigned)H * (unsigned)W;

  const int nci_t = p.cin / (32 * CIB);
  // workgroups are dealt round-robin over the 8 XCDs in dispatch order (x fastest): left alone, the channel blocks that
  // read the SAME image band (same blockIdx.y) land on 8 different XCDs and each L2 fetches the band again.  The bijective
  // remap gives every XCD a contiguous range of (band, channel block) pairs, so a band's channel blocks share one L2.
  const unsigned vb_ = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const int bx_ = (int)(vb_ % gridDim.x), by_ = (int)(vb_ / gridDim.x);   // bf16 5x5-128: 8.16 -> 7.90 ms (A/B, same box)
  const int cob = bx_ / nci_t, cib = bx_ % nci_t;   // 64-cout block, 32-cin block
  const int split = by_;
  const int b = split / p.nbands, band = split % p.nbands;
  const int tiles_y = (H + TH - 1) / TH;
  const int ty_begin = (int)((long)band * tiles_y / p.nbands);          // tile rows spread evenly over the bands
  const int ty_end = (int)((long)(band + 1) * tiles_y / p.nbands);
  const int ntile = (ty_end - ty_begin) * p.tiles_x;

  const unsigned short* const xg = p.x + b * p.x_img + p.x_base + (long)cib * 32 * CIB * HW;
  const unsigned short* const gg = p.gy + b * p.g_img + p.g_base + (long)cob * 64 * HW;

  // staging plan (tile independent): chunk e = tid + NT k
  unsigned xrel[XE], grel[GE];      // byte offset of the chunk relative to the tile origin (channel, row, 8-pixel column)
  int xrc[XE], grc[GE];             // (row << 8) | chunk column, for the border masks
  int xlds[XE], glds[GE];           // LDS byte address inside a buffer
#pragma unroll
  for (int k = 0; k < XE; ++k) {
    const int e = tid + k * NT;
    const int ch = e % (XC / 8), r = (e / (XC / 8)) % XR, c = e / ((XC / 8) * XR);
    const bool in = (NXC % NT == 0) || e < NXC;
    xrel[k] = (unsigned)c * HW2 + 2u * (unsigned)(r * W + ch * 8);
    xrc[k] = in ? ((r << 8) | ch) : -1;                          // padding chunks of the last staging round: never loaded
    xlds[k] = in ? c * XPITCH + r * XROW + ch * 16 : 0;
  }
#pragma unroll
  for (int k = 0; k < GE; ++k) {
    const int e = tid + k * NT;
    const int ch = e % (TW / 8), r = (e / (TW / 8)) % TH, c = e / ((TW / 8) * TH);
    const bool in = (NGC % NT == 0) || e < NGC;
    grel[k] = (unsigned)c * HW2 + 2u * (unsigned)(r * W + ch * 8);
    grc[k] = in ? ((r << 8) | ch) : -1;
    glds[k] = in ? c * GPITCH + (r * TW + ch * 8) * 2 : 0;
  }

  const int co_t = wave & 1, dy = KSPLIT ? 0 : (wave >> 1);
  const int krow = wave >> 1;   // KSPLIT: the tile row whose k-steps this wave takes
  const int a_lane = (co_t * 32 + l31) * GPITCH + (8 * half) * 2;
  const int b_lane = l31 * XPITCH + (dy * XC + 8 * half + (KSPLIT ? 0 : 4)) * 2;   // window = pixels 4 .. 19 past (row, c0 + 8h)

  constexpr int NACC = KSPLIT ? CIB : KS;
  f32x16 acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  cm_u32x4 xv[XE], gv[GE];
  auto load_tile = [&](int t) {
    const int ty = ty_begin + t / p.tiles_x, tx = t % p.tiles_x;
    const int tx0 = tx * TW, ty0 = ty * TH;
    // the tile origin (possibly before the slice start: such chunks are masked) goes into the descriptor base
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(xg + ((long)(ty0 - PAD) * W + (tx0 - XL))), 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(gg + ((long)ty0 * W + tx0)), 0, (int)OOB, 0x00020000);
#pragma unroll
    for (int k = 0; k < XE; ++k) {
      const int gy_ = ty0 - PAD + (xrc[k] >> 8), gx_ = tx0 - XL + 8 * (xrc[k] & 255);
      const bool ok = xrc[k] >= 0 && gy_ >= 0 && gy_ < H && gx_ >= 0 && gx_ < W;   // W % 8 == 0: a chunk is in or out
      const auto v = __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? xrel[k] : OOB, 0, 0);
      xv[k] = *reinterpret_cast<const cm_u32x4*>(&v);
    }
#pragma unroll
    for (int k = 0; k < GE; ++k) {
      const int gy_ = ty0 + (grc[k] >> 8), gx_ = tx0 + 8 * (grc[k] & 255);
      const bool ok = grc[k] >= 0 && gy_ < H && gx_ < W;
      const auto v = __builtin_amdgcn_raw_buffer_load_b128(gr, ok ? grel[k] : OOB, 0, 0);
      gv[k] = *reinterpret_cast<const cm_u32x4*>(&v);
    }
  };
  auto store_tile = [&](int buf) {
    unsigned char* xs = lds + buf * (XBYTES + GBYTES);
    unsigned char* gs = xs + XBYTES;
#pragma unroll
    for (int k = 0; k < XE; ++k)
      if ((NXC % NT == 0) || tid + k * NT < NXC) *reinterpret_cast<cm_u32x4*>(xs + xlds[k]) = xv[k];
#pragma unroll
    for (int k = 0; k < GE; ++k)
      if ((NGC % NT == 0) || tid + k * NT < NGC) *reinterpret_cast<cm_u32x4*>(gs + glds[k]) = gv[k];
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

    const unsigned char* xs = lds + (t & 1) * (XBYTES + GBYTES);
    const unsigned char* gs = xs + XBYTES;
    cm_u32x4 a2[2], wm[2], wk[CIB];
    cm_u32x2 wl[2], wh[2];
#define CM_READ(ks_, s_)                                                                     \
    {                                                                                        \
      const int r_ = (ks_) / (TW / 16), c0_ = ((ks_) % (TW / 16)) * 16;                      \
      a2[s_] = *(lds_r128)(gs + a_lane + (r_ * TW + c0_) * 2);                               \
      const unsigned char* bp_ = xs + b_lane + (r_ * XC + c0_) * 2;                          \
      if constexpr (KSPLIT) {               /* k = 1: the fragments are the aligned 8 pixels of each cin tile */ \
        _Pragma("unroll") for (int j = 0; j < CIB; ++j) wk[j] = *(lds_r128)(bp_ + j * 32 * XPITCH); \
      } else {                                                                               \
        wl[s_] = *(lds_r64)(bp_);                                                            \
        wm[s_] = *(lds_r128)(bp_ + 8);                                                       \
        wh[s_] = *(lds_r64)(bp_ + 24);                                                       \
      }                                                                                      \
    }
    if constexpr (KSPLIT) {
#pragma unroll
      for (int c = 0; c < TW / 16; ++c) {
        CM_READ(krow * (TW / 16) + c, 0)
#pragma unroll
        for (int j = 0; j < CIB; ++j)
          acc[j] = E::mfma(*reinterpret_cast<const vec8*>(&a2[0]), *reinterpret_cast<const vec8*>(&wk[j]), acc[j]);
      }
    } else {
    CM_READ(0, 0)
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
      const int cur = ks & 1;
      if (ks + 1 < NK) CM_READ(ks + 1, cur ^ 1)
      // window words w[0..7] = pixels 4 .. 19; fragment of tap dx = window elements s0 .. s0+7, s0 = 4 - PAD + dx
      const unsigned w[8] = {wl[cur][0], wl[cur][1], wm[cur][0], wm[cur][1], wm[cur][2], wm[cur][3], wh[cur][0], wh[cur][1]};
      const vec8 av = *reinterpret_cast<const vec8*>(&a2[cur]);
#pragma unroll
      for (int dx = 0; dx < KS; ++dx) {
        constexpr int base = 4 - PAD;
        const int s0 = base + dx, m = s0 / 2;
        cm_u32x4 f;
        if (s0 % 2 == 0) {
          f = cm_u32x4{w[m], w[m + 1], w[m + 2], w[m + 3]};
        } else {
          f = cm_u32x4{__builtin_amdgcn_alignbit(w[m + 1], w[m], 16), __builtin_amdgcn_alignbit(w[m + 2], w[m + 1], 16),
                       __builtin_amdgcn_alignbit(w[m + 3], w[m + 2], 16), __builtin_amdgcn_alignbit(w[m + 4], w[m + 3], 16)};
        }
        acc[dx] = E::mfma(av, *reinterpret_cast<const vec8*>(&f), acc[dx]);
      }
    }
    }
#undef CM_READ
    if (has_next) store_tile((t + 1) & 1);
    __syncthreads();
  }

  float* __restrict__ wsp = p.ws + (long)split * (KS * KS) * p.cout * p.cin;
  if constexpr (KSPLIT) {
    // sum the TH row-group partials of each (cout tile, cin tile) in fixed order (deterministic) through LDS
    float* red = reinterpret_cast<float*>(lds);                  // [krow][co_t][16][64] floats = 8 KB per row group
#pragma unroll
    for (int j = 0; j < CIB; ++j) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((krow * 2 + co_t) * 16 + r) * 64 + lane] = acc[j][r];
      __syncthreads();
      if (krow == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = red[((0 * 2 + co_t) * 16 + r) * 64 + lane];
#pragma unroll
          for (int g = 1; g < TH; ++g) v += red[((g * 2 + co_t) * 16 + r) * 64 + lane];
          const int co = cob * 64 + co_t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          wsp[(long)co * p.cin + (cib * CIB + j) * 32 + l31] = v;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int dx = 0; dx < KS; ++dx) {
    const int tap = dy * KS + dx;
    const int ci = cib * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cob * 64 + co_t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      wsp[((long)tap * p.cout + co) * p.cin + ci] = acc[dx][r];
    }
  }
}

bool conv_wgrad_cm16_supported(const codon_conv_desc* d, const void* x, const void* gy) {
  if (!(d->ksize == 1 || d->ksize == 3 || d->ksize == 5)) return false;
  if (d->ksize == 1 && d->cin % (32 * CM_CIB1) != 0) return false;
  if (d->width % 8 != 0) return false;
  const long HW = (long)d->height * d->width;
  if (HW * 2 * 128 >= 0xFFFFFFF0L) return false;                      // 32-bit chunk offsets: up to 128 planes of a slice
  // every chunk address = slice start + 2 * (plane * HW + row * W + 8 * k): 16-byte aligned iff the slice start is
  // (W % 8 == 0 makes HW % 8 == 0)
  const uintptr_t xa = reinterpret_cast<uintptr_t>(x) + 2 * (uintptr_t)(d->x_coff * HW);
  const uintptr_t ga = reinterpret_cast<uintptr_t>(gy) + 2 * (uintptr_t)(d->y_coff * HW);
  return (xa % 16) == 0 && (ga % 16) == 0 && ((d->x_ctotal * HW * 2) % 16) == 0 && ((d->y_ctotal * HW * 2) % 16) == 0;
}

int launch_wgrad_cm16(const codon_conv_desc* d, const void* x, const void* gy, float* workspace, int tiles_x,
                      int band_rows, int nbands, int nsplit, int nchan_blocks, hipStream_t stream) {
  const long HW = (long)d->height * d->width;
  WgradCmParams p;
  p.x = (const unsigned short*)x; p.gy = (const unsigned short*)gy; p.ws = workspace;
  p.H = d->height; p.W = d->width; p.cin = d->cin; p.cout = d->cout;
  p.x_img = d->x_ctotal * HW; p.g_img = d->y_ctotal * HW;
  p.x_base = d->x_coff * HW; p.g_base = d->y_coff * HW;
  p.tiles_x = tiles_x; p.band_tiles_y = band_rows / CM_TH;   // band_rows: a multiple of 8
  p.nbands = nbands; p.nsplit = nsplit;
  dim3 grid(nchan_blocks, nsplit);
  if (d->ksize == 1) grid.x = (d->cout / 64) * (d->cin / (32 * CM_CIB1));
  const bool f16 = d->dtype == CODON_F16;
  if (d->ksize == 5) {
    if (f16) hipLaunchKernelGGL((conv_wgrad_cm16_kernel<CmF16, 5>), grid, dim3(640), 0, stream, p);
    else hipLaunchKernelGGL((conv_wgrad_cm16_kernel<CmBf16, 5>), grid, dim3(640), 0, stream, p);
  } else if (d->ksize == 3) {
    if (f16) hipLaunchKernelGGL((conv_wgrad_cm16_kernel<CmF16, 3>), grid, dim3(384), 0, stream, p);
    else hipLaunchKernelGGL((conv_wgrad_cm16_kernel<CmBf16, 3>), grid, dim3(384), 0, stream, p);
  } else {
    if (f16) hipLaunchKernelGGL((conv_wgrad_cm16_kernel<CmF16, 1>), grid, dim3(2 * CM_TH * 64), 0, stream, p);
    else hipLaunchKernelGGL((conv_wgrad_cm16_kernel<CmBf16, 1>), grid, dim3(2 * CM_TH * 64), 0, stream, p);
  }
  return check_launch("conv_wgrad_cm16_kernel");
}

}  // namespace codon
