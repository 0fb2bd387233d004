// fp32 weight gradient of a stride-1 "same" conv on the matrix cores (v_mfma_f32_32x32x2_f32).
//
//   dW[co][ci][dy][dx] = sum_{b,h,w} gy[b,co,h,w] * x[b,ci,h+dy-p,w+dx-p]
//
// (autograd of nn.Conv2d, /root/reference/CODON_X4/CODON_x4.py:24-47; the reference has no explicit
// backward -- SURVEY.md 3.4 -- this is what torch.autograd computes for it.)
//
// GEMM view: M = cout, N = cin (per filter tap), K = pixels.
//   A (32 x 2): lane l holds gy[co = l&31][pixel k0 + (l>>5)]
//   B (2 x 32): lane l holds x [pixel k0 + (l>>5) + tap][ci = l&31]
//   D (32 x 32) per (co tile, ci tile, tap): lane holds ci column, 16 co rows.
// Both operands index CHANNEL by lane, so the LDS tiles are [channel][pixels] with an ODD plane
// stride: 32 lanes -> 32 distinct banks.  The two k values are two horizontally adjacent pixels.
//
// Workgroup (256 threads, 4 waves) owns CO_T x CI_T channel tiles x all KS*KS taps (accumulators
// spread over the waves round-robin, <= 9 tiles = 144 VGPRs per wave) and streams a pixel range
// (one image band) through LDS in 4 x 32 pixel tiles; its partial dW goes to
// workspace[split][tap][co][ci] (coalesced) and wgrad_reduce_kernel sums the splits in fixed order
// (deterministic, no atomics) into OIHW, optionally accumulating (the 5x / 3x weight sharing).

#include "codon_common.h"

namespace codon {

struct WgradParams {
  const float* x;
  const float* gy;
  float* ws;  // [nsplit][taps][cout][cin]
  int H, W, cin, cout;
  long x_img, g_img, x_base, g_base;
  int tiles_x, band_tiles_y, nbands;  // tile rows per band, bands per image
  int nsplit;
};

template <int KS, int CO_T, int CI_T>
__global__ __launch_bounds__(256, 2) void conv_wgrad_f32_kernel(const WgradParams p) {
  constexpr int PAD = KS / 2;
  constexpr int TW = 32, TH = 4;
  constexpr int XR = TH + KS - 1, XQ = TW + KS - 1;
  constexpr int XPL = (XR * XQ) | 1;  // odd plane strides: conflict-free channel-per-lane reads
  constexpr int GPL = (TH * TW) | 1;
  constexpr int NCI = CI_T * 32, NCO = CO_T * 32;
  constexpr int XS = NCI * XPL, GS = NCO * GPL;
  constexpr int TAPS = KS * KS;
  constexpr int NT = CO_T * CI_T * TAPS;
  constexpr int TPW = (NT + 3) / 4;
  constexpr int XE = (NCI * XR * XQ + 255) / 256;
  constexpr int GE = (NCO * TH * TW) / 256;

  __shared__ float lds[XS + GS];
  float* const xs = lds;
  float* const gs = lds + XS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int H = p.H, W = p.W;
  const long HW = (long)H * W;

  // blockIdx.x = channel-tile pair, blockIdx.y = split (image, band)
  const int nci_t = p.cin / NCI;
  const int cot = blockIdx.x / nci_t, cit = blockIdx.x % nci_t;
  const int split = blockIdx.y;
  const int b = split / p.nbands, band = split % p.nbands;
  const int ty_begin = band * p.band_tiles_y;
  const int tiles_y = (H + TH - 1) / TH;
  const int ty_end = min(ty_begin + p.band_tiles_y, tiles_y);

  const float* __restrict__ xg = p.x + b * p.x_img + p.x_base + (long)cit * NCI * HW;
  const float* __restrict__ gg = p.gy + b * p.g_img + p.g_base + (long)cot * NCO * HW;

  // per-wave tile list: t = wave + 4*j -> (co_t, ci_t, tap); LDS base addresses per tile
  int a_off[TPW], b_off[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    int t = wave + 4 * j;
    if (t >= NT) t = NT - 1;  // padded slot: recomputes the last tile, result discarded
    const int tap = t % TAPS;
    const int cc = t / TAPS;
    const int ci_t = cc % CI_T, co_t = cc / CI_T;
    const int dy = tap / KS, dx = tap % KS;
    a_off[j] = (co_t * 32 + l31) * GPL + half;
    b_off[j] = (ci_t * 32 + l31) * XPL + dy * XQ + dx + half;
  }

  f32x16 acc[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

#pragma unroll 1
  for (int ty = ty_begin; ty < ty_end; ++ty) {
#pragma unroll 1
    for (int tx = 0; tx < p.tiles_x; ++tx) {
      const int tx0 = tx * TW, ty0 = ty * TH;
      __syncthreads();  // previous tile's reads done
      // stage x halo tile (zero padded) and gy tile (zero outside the image); modest unroll keeps the
      // loads-in-flight registers small next to the 112-144 accumulator VGPRs
#pragma unroll 6
      for (int k = 0; k < XE; ++k) {
        const int e = tid + k * 256;
        const int c = e / (XR * XQ);
        const int rem = e - c * (XR * XQ);
        const int r = rem / XQ, q = rem - r * XQ;
        const int gy_ = ty0 + r - PAD, gx_ = tx0 + q - PAD;
        if (e < NCI * XR * XQ) {
          const bool ok = gy_ >= 0 && gy_ < H && gx_ >= 0 && gx_ < W;
          const float v = xg[ok ? c * HW + (long)gy_ * W + gx_ : 0];   // unconditional load, then select
          xs[c * XPL + rem] = ok ? v : 0.f;
        }
      }
#pragma unroll 8
      for (int k = 0; k < GE; ++k) {
        const int e = tid + k * 256;
        const int c = e / (TH * TW);
        const int rem = e - c * (TH * TW);
        const int r = rem / TW, q = rem - r * TW;
        const int gy_ = ty0 + r, gx_ = tx0 + q;
        const bool ok = gy_ < H && gx_ < W;
        const float v = gg[ok ? c * HW + (long)gy_ * W + gx_ : 0];
        gs[c * GPL + rem] = ok ? v : 0.f;
      }
      __syncthreads();
      // K loop over the tile's 128 pixels, two horizontally adjacent pixels per MFMA
#pragma unroll 1
      for (int r = 0; r < TH; ++r) {
#pragma unroll 4
        for (int q = 0; q < TW; q += 2) {
          const int po = r * TW + q, xo = r * XQ + q;
#pragma unroll
          for (int j = 0; j < TPW; ++j) {
            const float a = gs[a_off[j] + po];
            const float bv = xs[b_off[j] + xo];
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[j], 0, 0, 0);
          }
        }
      }
    }
  }

  // partial dW -> workspace[split][tap][co][ci]
  float* __restrict__ wsp = p.ws + (long)split * TAPS * p.cout * p.cin;
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int t = wave + 4 * j;
    if (t < NT) {
      const int tap = t % TAPS;
      const int cc = t / TAPS;
      const int ci_t = cc % CI_T, co_t = cc / CI_T;
      const int ci = cit * NCI + ci_t * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = cot * NCO + co_t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        wsp[((long)tap * p.cout + co) * p.cin + ci] = acc[j][r];
      }
    }
  }
}

// dw[co][ci][tap] (+)= sum_s ws[s][tap][co][ci], fixed order over s.  A thread sums 4 consecutive ci (16-byte loads, 8
// splits in flight): the pass is a plain stream over the workspace (52 MB for a 5x5 128->128: 45 -> ~15 us; one element
// per thread with a 4-byte load per split was latency-bound at 1.2 TB/s and 2.5 ms of a bf16 training step).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                           int cout, int cin, int taps, int nsplit, int accumulate) {
  const long n = (long)cout * cin * taps;
  const long i = (blockIdx.x * 256L + threadIdx.x) * 4;  // index in [tap][co][ci] order (coalesced reads); cin % 4 == 0
  if (i >= n) return;
  const int ci = (int)(i % cin);
  const long t = i / cin;
  const int co = (int)(t % cout);
  const int tap = (int)(t / cout);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  int k = 0;
  for (; k + 8 <= nsplit; k += 8) {
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4*>(ws + (long)(k + j) * n + i);
#pragma unroll
    for (int j = 0; j < 8; ++j) { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; }
  }
  for (; k < nsplit; ++k) {
    const float4 v = *reinterpret_cast<const float4*>(ws + (long)k * n + i);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  float* o = dw + ((long)co * cin + ci) * taps + tap;
  const float r[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) o[(long)j * taps] = accumulate ? o[(long)j * taps] + r[j] : r[j];
}

int launch_wgrad_reduce(const float* ws, float* dw, int cout, int cin, int taps, int nsplit, int accumulate,
                        hipStream_t stream) {
  const long n = (long)cout * cin * taps;
  if (cin % 4 != 0 || ((uintptr_t)ws % 16) != 0) {
    set_error("wgrad_reduce: cin %d not a multiple of 4 or workspace not 16-byte aligned", cin);
    return CODON_ERR_BAD_ARG;
  }
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, ws, dw, cout, cin,
                     taps, nsplit, accumulate);
  return check_launch("wgrad_reduce_kernel");
}

// conv_wgrad_f32_t16.hip: 16x16x4 tiles, 8 balanced waves, double-buffered (W % 4 == 0, 16-byte aligned slices, k in {3,5})
bool conv_wgrad_f32_t16_shape(const codon_conv_desc* d);
bool conv_wgrad_f32_t16_supported(const codon_conv_desc* d, const void* x, const void* gy);
int launch_wgrad_f32_t16(const codon_conv_desc* d, const float* x, const float* gy, float* workspace, int nbands,
                         int nsplit, hipStream_t stream);

struct WgradPlan {
  int co_t, ci_t, nbands, band_tiles_y, nsplit, nchan_blocks;
  bool t16;
};

static bool wgrad_plan(const codon_conv_desc* d, WgradPlan* pl) {
  const int k = d->ksize, ci = d->cin, co = d->cout;
  if (k == 5 && ((ci == 128 && co == 128) || (ci == 64 && co == 64))) { pl->co_t = 1; pl->ci_t = 1; }
  else if (k == 3 && ((ci == 64 || ci == 128) && co == 64)) { pl->co_t = 2; pl->ci_t = 1; }
  else if (k == 1 && ci == 128 && co == 64) { pl->co_t = 2; pl->ci_t = 2; }
  else return false;
  pl->nchan_blocks = (co / (32 * pl->co_t)) * (ci / (32 * pl->ci_t));
  // the plan (and with it the workspace size) depends on the descriptor only: shapes the 16x16x4 kernel covers are
  // planned for its grid (64 cout x 32 cin per workgroup, one workgroup per CU); if the pointers then turn out
  // misaligned, the round-1 kernel runs on the same band split
  pl->t16 = conv_wgrad_f32_t16_shape(d);
  const int plan_blocks = pl->t16 ? (co / 64) * (ci / (k == 1 ? 128 : 32)) : pl->nchan_blocks;
  const int target = pl->t16 ? 256 : 1024;       // workgroups per launch: 1 per CU (A/B: 256 beats 512 / 768 by 2-4 % on the 64-channel convs) / 2 x 2 per CU
  const int tiles_y = (d->height + 3) / 4;
  // enough workgroups to fill the chip, but bounded workspace: bands per image
  int want = (target + plan_blocks * d->batch - 1) / (plan_blocks * d->batch);
  if (want < 1) want = 1;
  if (want > tiles_y) want = tiles_y;
  pl->band_tiles_y = (tiles_y + want - 1) / want;
  pl->nbands = (tiles_y + pl->band_tiles_y - 1) / pl->band_tiles_y;
  pl->nsplit = d->batch * pl->nbands;
  return true;
}

size_t conv_wgrad_workspace_bytes(const codon_conv_desc* d) {
  WgradPlan pl;
  if (!wgrad_plan(d, &pl)) return 0;
  return (size_t)pl.nsplit * d->cout * d->cin * d->ksize * d->ksize * sizeof(float);
}

template <int KS, int CO_T, int CI_T>
static void launch_wgrad(const WgradParams& p, int nchan_blocks, hipStream_t stream) {
  hipLaunchKernelGGL((conv_wgrad_f32_kernel<KS, CO_T, CI_T>), dim3(nchan_blocks, p.nsplit), dim3(256), 0, stream, p);
}

int conv2d_wgrad_f32(const codon_conv_desc* d, const float* x, const float* gy, float* dw, float* workspace,
                     size_t ws_bytes, int accumulate, hipStream_t stream) {
  WgradPlan pl;
  if (!wgrad_plan(d, &pl)) {
    set_error("conv2d_wgrad: no f32 kernel for k=%d cin=%d cout=%d", d->ksize, d->cin, d->cout);
    return CODON_ERR_UNSUPPORTED;
  }
  CODON_REQUIRE(ws_bytes >= conv_wgrad_workspace_bytes(d), CODON_ERR_BAD_ARG,
                "conv2d_wgrad: workspace %zu B < required %zu B", ws_bytes, conv_wgrad_workspace_bytes(d));
  CODON_REQUIRE(pl.nsplit <= 65535, CODON_ERR_UNSUPPORTED, "conv2d_wgrad: %d splits > 65535", pl.nsplit);
  static const bool t16_env = getenv("CODON_WGRAD_T16") ? atoi(getenv("CODON_WGRAD_T16")) != 0 : true;   // 0: round-1 kernel (A/B)
  if (pl.t16 && t16_env && conv_wgrad_f32_t16_supported(d, x, gy)) {
    const int st = launch_wgrad_f32_t16(d, x, gy, workspace, pl.nbands, pl.nsplit, stream);
    if (st != CODON_OK || accumulate == CODON_WGRAD_DEFER) return st;
    return launch_wgrad_reduce(workspace, dw, d->cout, d->cin, d->ksize * d->ksize, pl.nsplit, accumulate, stream);
  }
  const long HW = (long)d->height * d->width;
  WgradParams p;
  p.x = x; p.gy = gy; p.ws = workspace;
  p.H = d->height; p.W = d->width; p.cin = d->cin; p.cout = d->cout;
  p.x_img = d->x_ctotal * HW; p.g_img = d->y_ctotal * HW;
  p.x_base = d->x_coff * HW; p.g_base = d->y_coff * HW;
  p.tiles_x = (d->width + 31) / 32;
  p.band_tiles_y = pl.band_tiles_y; p.nbands = pl.nbands; p.nsplit = pl.nsplit;
  if (d->ksize == 5) launch_wgrad<5, 1, 1>(p, pl.nchan_blocks, stream);
  else if (d->ksize == 3) launch_wgrad<3, 2, 1>(p, pl.nchan_blocks, stream);
  else launch_wgrad<1, 2, 2>(p, pl.nchan_blocks, stream);
  int st = check_launch("conv_wgrad_f32_kernel");
  if (st != CODON_OK || accumulate == CODON_WGRAD_DEFER) return st;
  return launch_wgrad_reduce(workspace, dw, d->cout, d->cin, d->ksize * d->ksize, pl.nsplit, accumulate, stream);
}

}  // namespace codon
