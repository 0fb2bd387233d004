// bf16 weight gradient on the gfx950 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate, fp32 dW).
//
//   dW[co][ci][dy][dx] = sum_{b,h,w} gy[b,co,h,w] * x[b,ci,h+dy-p,w+dx-p]        (autograd of nn.Conv2d)
//
// GEMM: M = cout, N = cin (per tap), K = pixels; one MFMA consumes 16 pixels of one image row,
// 8 consecutive ones per lane:
//   A (32 x 16): lane l holds gy[co = l&31][pix 8h .. 8h+7]      <- gs[co][pixel]   : one aligned ds_read_b128
//   B (16 x 32): lane l holds x [pix 8h .. 8h+7 (+tap)][ci = l&31]
// The tap shift (dy,dx) moves B's pixels by dx ELEMENTS (2 bytes), which would misalign a 16-byte row read
// of a channel-major tile.  So the x halo tile is stored PIXEL-major, xs[pixel][32 ci] (72-byte rows: 64 data
// + 8 pad), where a tap is a whole-row offset, and the fragment is fetched with the gfx950 transposing read
// ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group, delivered channel-per-lane): two reads per
// fragment, always 8-byte aligned, the four rows of a half-wave cover 256 contiguous bytes.
//
// Workgroup = 512 threads = 8 waves: 64 cout x 32 cin x all KS*KS taps.  Waves 0-3 own co tile 0, waves 4-7
// co tile 1; within a group wave g owns taps g, g+4, g+8, ... (<= 7 accumulator tiles = 112 VGPRs).  Per
// 16-pixel k-step a wave issues 1 A read + 2 transposing reads per tap + 1 MFMA per tap.
// A workgroup streams one image band in 4 x 32 pixel tiles, LDS double-buffered with the next tile
// prefetched to registers during the MFMAs.  Partials -> workspace[split][tap][co][ci] (fp32), summed in
// fixed order by wgrad_reduce_kernel (shared with the fp32 path): deterministic.

#include "codon_common.h"

namespace codon {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
struct WBf16 {
  typedef bf16x8 vec8;
  __device__ static f32x16 mfma(vec8 a, vec8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
struct WF16 {
  typedef f16x8 vec8;
  __device__ static f32x16 mfma(vec8 a, vec8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

struct Wgrad16Params {
  const u16* x;
  const u16* gy;
  float* ws;
  int H, W, cin, cout;
  long x_img, g_img, x_base, g_base;
  int tiles_x, band_tiles_y, nbands, nsplit;
  int pair;  // W even and 4-byte aligned planes: 4-byte pixel-pair staging loads
};

template <class E, int KS, bool PAIR>
__global__ __launch_bounds__(512, 2) void conv_wgrad_bf16_kernel(const Wgrad16Params p) {
  typedef typename E::vec8 vec8;
  constexpr int PAD = KS / 2;
  constexpr int TW = 32, TH = 4;
  constexpr int PADX = (KS == 1) ? 0 : 2;   // even column origin of the halo tile, so pixel pairs stay 4-byte aligned
  constexpr int XR = TH + KS - 1, XQ = TW + 2 * PADX, NP = XR * XQ;
  constexpr int XROWW = 18;              // 32-bit words per pixel row of xs: 16 data + 2 pad (72 B)
  constexpr int XSW = NP * XROWW;        // words per x buffer
  constexpr int GPLW = 68;               // words per cout row of gs: 128 pixels + 8 pad (272 B)
  constexpr int GSW = 64 * GPLW;
  constexpr int TAPS = KS * KS;
  constexpr int TPW = (TAPS + 3) / 4;    // taps per wave
  constexpr int XWORDS = 16 * NP;        // channel-pair words in the x tile
  constexpr int XE = (XWORDS + 511) / 512;
  constexpr int GWORDS = 64 * (TH * TW / 2);
  constexpr int GE = GWORDS / 512;

  __shared__ __attribute__((aligned(16))) unsigned lds[2 * (XSW + GSW)];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int H = p.H, W = p.W;
  const long HW = (long)H * W;

  const int nci_t = p.cin / 32;
  const int cob = blockIdx.x / nci_t, cib = blockIdx.x % nci_t;   // 64-cout block, 32-cin block
  const int split = blockIdx.y;
  const int b = split / p.nbands, band = split % p.nbands;
  const int tiles_y = (H + TH - 1) / TH;
  const int ty_begin = (int)((long)band * tiles_y / p.nbands);          // tile rows spread evenly over the bands
  const int ty_end = (int)((long)(band + 1) * tiles_y / p.nbands);
  const int ntile = (ty_end - ty_begin) * p.tiles_x;

  const u16* __restrict__ xg = p.x + b * p.x_img + p.x_base + (long)cib * 32 * HW;
  const u16* __restrict__ gg = p.gy + b * p.g_img + p.g_base + (long)cob * 64 * HW;

  const int co_t = wave >> 2, wg = wave & 3;
  // per-lane byte offsets of the fragments inside a buffer
  const int li = lane & 15, q = li >> 2, pp = li & 3, cblk = (lane >> 4) & 1;
  const int b_lane = (8 * half + q) * (XROWW * 4) + (16 * cblk + 4 * pp) * 2;   // transposing read address
  const int a_lane = ((co_t * 32 + l31) * GPLW) * 4 + (8 * half) * 2;          // gy row, 8 pixels
  int tap_off[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    int tap = wg + 4 * j;
    if (tap >= TAPS) tap = TAPS - 1;
    tap_off[j] = ((tap / KS) * XQ + (tap % KS) + (PADX - PAD)) * (XROWW * 4);
  }

  f32x16 acc[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // Staging registers.  PAIR path (W even, 4-byte aligned planes): every global access is a 4-byte pixel pair.
  //   x : unit u = (channel pair cp, pixel pair) -> two loads d0 (ch 2cp), d1 (ch 2cp+1), each = pixels (p, p+1);
  //       written as four ds_write_b16 at STORE time (no VALU between load and store, so the prefetch is not
  //       consumed -- and waited for -- before the MFMAs)
  //   gy: word = pixels (p, p+1) of one channel = the LDS word itself
  // Fallback (odd W): 2-byte loads.
  constexpr int XUNITS = 16 * (NP / 2);
  constexpr int XU = (XUNITS + 511) / 512;
  unsigned xd0[PAIR ? XU : 1], xd1[PAIR ? XU : 1], gr[PAIR ? GE : 1];
  u16 xlo[PAIR ? 1 : XE], xhi[PAIR ? 1 : XE], glo[PAIR ? 1 : GE], ghi[PAIR ? 1 : GE];

  auto load_tile = [&](int t) {
    const int ty = ty_begin + t / p.tiles_x, tx = t % p.tiles_x;
    const int tx0 = tx * TW, ty0 = ty * TH;
    if constexpr (PAIR) {
#pragma unroll
      for (int k = 0; k < XU; ++k) {
        const int u = tid + k * 512;
        const int cp = u / (NP / 2);
        const int pw = u - cp * (NP / 2);
        const int r = pw / (XQ / 2), q2 = (pw - r * (XQ / 2)) * 2;
        const int gy_ = ty0 + r - PAD, gx_ = tx0 + q2 - PADX;
        const bool ok = u < XUNITS && gy_ >= 0 && gy_ < H && gx_ >= 0 && gx_ < W;   // W even: pair in or out
        const long o = ok ? (long)(2 * cp) * HW + (long)gy_ * W + gx_ : 0;
        xd0[k] = *reinterpret_cast<const unsigned*>(xg + o);
        xd1[k] = *reinterpret_cast<const unsigned*>(xg + o + HW);
      }
#pragma unroll
      for (int k = 0; k < GE; ++k) {
        const int e = tid + k * 512;
        const int c = e / (TH * TW / 2);
        const int pw = e - c * (TH * TW / 2);
        const int r = pw / (TW / 2), q2 = (pw - r * (TW / 2)) * 2;
        const int gy_ = ty0 + r, gx_ = tx0 + q2;
        const bool ok = gy_ < H && gx_ < W;
        gr[k] = *reinterpret_cast<const unsigned*>(gg + (ok ? (long)c * HW + (long)gy_ * W + gx_ : 0));
      }
    } else {
#pragma unroll
      for (int k = 0; k < XE; ++k) {
        const int e = tid + k * 512;
        const int cp = e / NP;
        const int pi = e - cp * NP;
        const int r = pi / XQ, qq = pi - r * XQ;
        const int gy_ = ty0 + r - PAD, gx_ = tx0 + qq - PADX;
        const bool ok = e < XWORDS && gy_ >= 0 && gy_ < H && gx_ >= 0 && gx_ < W;
        const long o = ok ? (long)(2 * cp) * HW + (long)gy_ * W + gx_ : 0;
        xlo[k] = xg[o];
        xhi[k] = xg[o + HW];
      }
#pragma unroll
      for (int k = 0; k < GE; ++k) {
        const int e = tid + k * 512;
        const int c = e / (TH * TW / 2);
        const int pw = e - c * (TH * TW / 2);
        const int r = pw / (TW / 2), q2 = (pw - r * (TW / 2)) * 2;
        const int gy_ = ty0 + r, gx_ = tx0 + q2;
        const bool ok0 = gy_ < H && gx_ < W, ok1 = gy_ < H && gx_ + 1 < W;
        const long o = (long)c * HW + (long)gy_ * W + gx_;
        glo[k] = gg[ok0 ? o : 0];
        ghi[k] = gg[ok1 ? o + 1 : 0];
      }
    }
  };
  auto store_tile = [&](int buf, int t) {   // t: the tile the registers hold (for the zero-padding masks)
    const int ty = ty_begin + t / p.tiles_x, tx = t % p.tiles_x;
    const int tx0 = tx * TW, ty0 = ty * TH;
    unsigned* xs = lds + buf * (XSW + GSW);
    unsigned* gs = xs + XSW;
    u16* xs16 = reinterpret_cast<u16*>(xs);
    u16* gs16 = reinterpret_cast<u16*>(gs);
    if constexpr (PAIR) {
#pragma unroll
      for (int k = 0; k < XU; ++k) {
        const int u = tid + k * 512;
        const int cp = u / (NP / 2);
        const int pw = u - cp * (NP / 2);
        const int r = pw / (XQ / 2), q2 = (pw - r * (XQ / 2)) * 2;
        const int gy_ = ty0 + r - PAD, gx_ = tx0 + q2 - PADX;
        const bool ok = gy_ >= 0 && gy_ < H && gx_ >= 0 && gx_ < W;
        if (u < XUNITS) {
          const int pi = r * XQ + q2;
          const unsigned d0 = ok ? xd0[k] : 0u, d1 = ok ? xd1[k] : 0u;
          xs16[(pi * XROWW + cp) * 2] = (u16)d0;
          xs16[(pi * XROWW + cp) * 2 + 1] = (u16)d1;
          xs16[((pi + 1) * XROWW + cp) * 2] = (u16)(d0 >> 16);
          xs16[((pi + 1) * XROWW + cp) * 2 + 1] = (u16)(d1 >> 16);
        }
      }
#pragma unroll
      for (int k = 0; k < GE; ++k) {
        const int e = tid + k * 512;
        const int c = e / (TH * TW / 2);
        const int pw = e - c * (TH * TW / 2);
        const int r = pw / (TW / 2), q2 = (pw - r * (TW / 2)) * 2;
        const bool ok = ty0 + r < H && tx0 + q2 < W;
        gs[c * GPLW + pw] = ok ? gr[k] : 0u;
      }
    } else {
#pragma unroll
      for (int k = 0; k < XE; ++k) {
        const int e = tid + k * 512;
        const int cp = e / NP;
        const int pi = e - cp * NP;
        const int r = pi / XQ, qq = pi - r * XQ;
        const int gy_ = ty0 + r - PAD, gx_ = tx0 + qq - PADX;
        const bool ok = gy_ >= 0 && gy_ < H && gx_ >= 0 && gx_ < W;
        if (e < XWORDS) {
          xs16[(pi * XROWW + cp) * 2] = ok ? xlo[k] : (u16)0;
          xs16[(pi * XROWW + cp) * 2 + 1] = ok ? xhi[k] : (u16)0;
        }
      }
#pragma unroll
      for (int k = 0; k < GE; ++k) {
        const int e = tid + k * 512;
        const int c = e / (TH * TW / 2);
        const int pw = e - c * (TH * TW / 2);
        const int r = pw / (TW / 2), q2 = (pw - r * (TW / 2)) * 2;
        const bool ok0 = ty0 + r < H && tx0 + q2 < W, ok1 = ty0 + r < H && tx0 + q2 + 1 < W;
        gs16[(c * GPLW + pw) * 2] = ok0 ? glo[k] : (u16)0;
        gs16[(c * GPLW + pw) * 2 + 1] = ok1 ? ghi[k] : (u16)0;
      }
    }
  };

  if (ntile > 0) {
    load_tile(0);
    store_tile(0, 0);
  }
  __syncthreads();

#pragma unroll 1
  for (int t = 0; t < ntile; ++t) {
    const bool has_next = t + 1 < ntile;
    if (has_next) load_tile(t + 1);

    const char* xs = reinterpret_cast<const char*>(lds + (t & 1) * (XSW + GSW));
    const char* gs = xs + XSW * 4;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    // explicit software pipeline over the NK 16-pixel k-steps: the fragments of step ks+1 are read (one A
    // ds_read_b128 + 2 transposing reads per tap) BEFORE the MFMAs of step ks issue, into a second register set
    constexpr int NK = TH * (TW / 16);
    uint4 a_cur, a_nxt;
    s16x4 bl_cur[TPW], bh_cur[TPW], bl_nxt[TPW], bh_nxt[TPW];
#define WG_READ(ks_, A_, BL_, BH_)                                                                   \
    {                                                                                                \
      const int r_ = (ks_) / (TW / 16), c0_ = ((ks_) % (TW / 16)) * 16;                              \
      A_ = *reinterpret_cast<const uint4*>(gs + a_lane + (r_ * TW + c0_) * 2);                       \
      _Pragma("unroll") for (int j = 0; j < TPW; ++j) {                                              \
        const char* bp_ = xs + b_lane + tap_off[j] + (r_ * XQ + c0_) * (XROWW * 4);                  \
        BL_[j] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(bp_));                         \
        BH_[j] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(bp_ + 4 * XROWW * 4));         \
      }                                                                                              \
    }
#define WG_MFMA(A_, BL_, BH_)                                                                        \
    {                                                                                                \
      const vec8 av_ = *reinterpret_cast<const vec8*>(&A_);                                          \
      _Pragma("unroll") for (int j = 0; j < TPW; ++j) {                                              \
        union { struct { s16x4 l, h; } s; vec8 v; } u_;                                              \
        u_.s.l = BL_[j]; u_.s.h = BH_[j];                                                            \
        acc[j] = E::mfma(av_, u_.v, acc[j]);                                                         \
      }                                                                                              \
    }
    // NOTE: at the 256-VGPR cap hipcc re-serialises this into {2 reads, lgkmcnt(0), 1 MFMA} per tap with one
    // fragment set; pinning the order with sched_group_barrier made it spill (20.9 vs 14.8 ms) -- the fix is fewer
    // live staging registers (next round), not scheduling directives.
    WG_READ(0, a_cur, bl_cur, bh_cur);
#pragma unroll
    for (int ks = 0; ks < NK; ks += 2) {
      if (ks + 1 < NK) WG_READ(ks + 1, a_nxt, bl_nxt, bh_nxt);
      WG_MFMA(a_cur, bl_cur, bh_cur);
      if (ks + 2 < NK) WG_READ(ks + 2, a_cur, bl_cur, bh_cur);
      if (ks + 1 < NK) WG_MFMA(a_nxt, bl_nxt, bh_nxt);
    }
#undef WG_READ
#undef WG_MFMA
    if (has_next) store_tile((t + 1) & 1, t + 1);
    __syncthreads();
  }

  float* __restrict__ wsp = p.ws + (long)split * TAPS * p.cout * p.cin;
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int tap = wg + 4 * j;
    if (tap < TAPS) {
      const int ci = cib * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = cob * 64 + co_t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        wsp[((long)tap * p.cout + co) * p.cin + ci] = acc[j][r];
      }
    }
  }
}

// defined in conv_wgrad_f32.hip: dw[co][ci][tap] (+)= sum_s ws[s][tap][co][ci], fixed order
int launch_wgrad_reduce(const float* ws, float* dw, int cout, int cin, int taps, int nsplit, int accumulate,
                        hipStream_t stream);

// conv_wgrad_cm16.hip: channel-major tiles, one filter row per wave (W % 8 == 0, 16-byte aligned slices, k in {3,5})
bool conv_wgrad_cm16_supported(const codon_conv_desc* d, const void* x, const void* gy);
int launch_wgrad_cm16(const codon_conv_desc* d, const void* x, const void* gy, float* workspace, int tiles_x,
                      int band_tiles_y, int nbands, int nsplit, int nchan_blocks, hipStream_t stream);

#ifndef CODON_WGRAD16_TARGET
#define CODON_WGRAD16_TARGET 256
#endif
constexpr int WGRAD16_TARGET_BLOCKS = CODON_WGRAD16_TARGET;   // workgroups per launch the band split aims for

struct Wgrad16Plan {
  int nbands, band_tiles_y, nsplit, nchan_blocks;
};

static bool wgrad16_plan(const codon_conv_desc* d, Wgrad16Plan* pl) {
  const int k = d->ksize, ci = d->cin, co = d->cout;
  if (!((k == 1 || k == 3 || k == 5) && ci % 32 == 0 && co % 64 == 0)) return false;
  pl->nchan_blocks = (co / 64) * (ci / 32);
  // every band gets floor or ceil of tiles_y / nbands tile rows (both kernels use 4-row tiles)
  const int tiles_y = (d->height + 3) / 4;
  // k = 1 through conv_wgrad_cm16.hip covers 128 cin per workgroup: size the split for that (smaller) grid
  const int nb = (k == 1 && ci % 128 == 0) ? (co / 64) * (ci / 128) : pl->nchan_blocks;
  int want = (WGRAD16_TARGET_BLOCKS + nb * d->batch - 1) / (nb * d->batch);
  if (want < 1) want = 1;
  if (want > tiles_y) want = tiles_y;
  pl->nbands = want;
  pl->band_tiles_y = (tiles_y + want - 1) / want;     // upper bound (informational)
  pl->nsplit = d->batch * pl->nbands;
  return true;
}

size_t conv_wgrad_bf16_workspace_bytes(const codon_conv_desc* d) {
  Wgrad16Plan pl;
  if (!wgrad16_plan(d, &pl)) return 0;
  return (size_t)pl.nsplit * d->cout * d->cin * d->ksize * d->ksize * sizeof(float);
}

int conv2d_wgrad_bf16(const codon_conv_desc* d, const void* x, const void* gy, float* dw, float* workspace,
                      size_t ws_bytes, int accumulate, hipStream_t stream) {
  Wgrad16Plan pl;
  if (!wgrad16_plan(d, &pl)) {
    set_error("conv2d_wgrad: no bf16 kernel for k=%d cin=%d cout=%d", d->ksize, d->cin, d->cout);
    return CODON_ERR_UNSUPPORTED;
  }
  CODON_REQUIRE(ws_bytes >= conv_wgrad_bf16_workspace_bytes(d), CODON_ERR_BAD_ARG,
                "conv2d_wgrad: workspace %zu B < required %zu B", ws_bytes, conv_wgrad_bf16_workspace_bytes(d));
  CODON_REQUIRE(pl.nsplit <= 65535, CODON_ERR_UNSUPPORTED, "conv2d_wgrad: %d splits > 65535", pl.nsplit);
  static const bool cm_env = getenv("CODON_WGRAD_CM") ? atoi(getenv("CODON_WGRAD_CM")) != 0 : true;
  if (cm_env && conv_wgrad_cm16_supported(d, x, gy)) {
    const int st = launch_wgrad_cm16(d, x, gy, workspace, (d->width + 31) / 32, pl.band_tiles_y * 4, pl.nbands, pl.nsplit,
                                     pl.nchan_blocks, stream);
    if (st != CODON_OK) return st;
    return launch_wgrad_reduce(workspace, dw, d->cout, d->cin, d->ksize * d->ksize, pl.nsplit, accumulate, stream);
  }
  const long HW = (long)d->height * d->width;
  Wgrad16Params p;
  p.x = (const u16*)x; p.gy = (const u16*)gy; p.ws = workspace;
  p.H = d->height; p.W = d->width; p.cin = d->cin; p.cout = d->cout;
  p.x_img = d->x_ctotal * HW; p.g_img = d->y_ctotal * HW;
  p.x_base = d->x_coff * HW; p.g_base = d->y_coff * HW;
  p.tiles_x = (d->width + 31) / 32;
  p.band_tiles_y = pl.band_tiles_y; p.nbands = pl.nbands; p.nsplit = pl.nsplit;
  p.pair = (d->width % 2 == 0) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gy)) % 4 == 0);
  const dim3 grid(pl.nchan_blocks, pl.nsplit);
  const bool f16 = d->dtype == CODON_F16;
#define WG_LAUNCH(KS_)                                                                                         \
  {                                                                                                            \
    if (f16 && p.pair) hipLaunchKernelGGL((conv_wgrad_bf16_kernel<WF16, KS_, true>), grid, dim3(512), 0, stream, p);        \
    else if (f16) hipLaunchKernelGGL((conv_wgrad_bf16_kernel<WF16, KS_, false>), grid, dim3(512), 0, stream, p);            \
    else if (p.pair) hipLaunchKernelGGL((conv_wgrad_bf16_kernel<WBf16, KS_, true>), grid, dim3(512), 0, stream, p);         \
    else hipLaunchKernelGGL((conv_wgrad_bf16_kernel<WBf16, KS_, false>), grid, dim3(512), 0, stream, p);                    \
  }
  if (d->ksize == 5) WG_LAUNCH(5) else if (d->ksize == 3) WG_LAUNCH(3) else WG_LAUNCH(1)
#undef WG_LAUNCH
  int st = check_launch("conv_wgrad_bf16_kernel");
  if (st != CODON_OK) return st;
  return launch_wgrad_reduce(workspace, dw, d->cout, d->cin, d->ksize * d->ksize, pl.nsplit, accumulate, stream);
}

}  // namespace codon
