// 16-bit weight gradient over CHANNEL-BLOCKED tensors (c8.h): v_mfma_f32_32x32x16_{bf16,f16}, fp32 accumulate, fp32 dW.
//
//   dW[co][ci][dy][dx] = sum_{b,h,w} gy[b,co,h,w] * x[b,ci,h+dy-p,w+dx-p]        (autograd of nn.Conv2d;
//                                                  the reference has no explicit backward)
//
// GEMM: M = cout, N = cin (per tap), K = pixels; one MFMA consumes 16 pixels of one image row, 8 per lane:
//   A (32 x 16): lane l holds gy[co = l&31][pix 8h .. 8h+7]
//   B (16 x 32): lane l holds x [ci = l&31][pix 8h + dx - p .. + 7]
// Both operands want PIXELS along the lane's 8 elements, the C8 layout has CHANNELS there -- the LDS tiles therefore
// keep the HBM form (plane-major [8-channel plane][pixel][8 ch], staging = straight 16-byte copies, conflict-free
// ds_write_b128) and the transpose is done by the gfx950 transposing read ds_read_b64_tr_b16: a 16-lane group reads
// 4 pixels x 16 channels (two planes) and every lane receives ONE channel's 4 pixels.  Plane pitch = 64 (mod 256)
// bytes, so the four planes a half-wave touches sit on four disjoint bank quarters: conflict-free.
// A wave owns one filter ROW dy and one 32-cout tile.  The KS shifted B fragments of a k-step overlap in all but KS-1
// pixels, so the lane reads ONE 12-pixel window of its channel (three transposing reads) and derives the KS fragments
// in registers: even shift = a register offset, odd shift = four v_alignbit_b32.  Per k-step: 2 + 3 reads of 512 B
// for KS MFMAs.  In pixel-major LDS every shift is a whole-vector offset, so there is no W % 8 condition.
//
// Workgroup = 2*KS waves = (2 cout tiles) x (KS filter rows): 64 cout x 32 cin x all taps, streaming an image band in
// 4 x 32 pixel tiles, LDS double-buffered.  Partials -> workspace[split][tap][co][ci] (fp32), summed in fixed order by
// wgrad_reduce_kernel: deterministic.  k = 1 (confuse*): 8 waves = 2 cout tiles x 4 tile rows over 128 cin, partials
// summed through LDS.

#include "c8.h"

namespace codon {

constexpr int WC8_TH = 4;
// k = 5 (round 5): a staged tile is 10 rows x 32 pixels.  Its x halo is 14 rows for 10 (1.4x) instead of 8 for 4 (2x): the HBM
// traffic of a 128 -> 128 launch fell from 7.60 GB (4 rows) to 6.34 (8 rows) to 6.07 GB (10 rows) = 1.51x -> 1.26x -> 1.21x
// its 5.03 GB algorithmic bytes (PMC), 5.92 -> 5.72 -> 5.66 ms stand-alone; at 8 rows the matrix pipe went 75 -> 79 % busy at
// 1.71 -> 1.67 GHz (the kernel runs at the socket's power cap: bytes not moved come back as issue slots).  TWO buffers
// (2 x 73 KB of LDS at 10 rows; 12 rows would need 171 KB) instead of the three of the 4-row form: the DMA of tile t+2 still
// has two 4-row tile times to land, and there are fewer publishing barriers.  (-DCODON_WC8_TH5=8 / =4: the earlier forms.)
#ifndef CODON_WC8_TH5
#define CODON_WC8_TH5 10
#endif
#ifndef CODON_WC8_TH3
#define CODON_WC8_TH3 6          // k = 3 (round 5): an 8-row halo for 6 rows (1.33x) instead of 6 for 4 (1.5x), 125 KB of LDS: 0.74 -> 0.72 ms; 4 = A/B
#endif
template <int KS> struct Wc8Th { static constexpr int value = KS == 5 ? CODON_WC8_TH5 : KS == 3 ? CODON_WC8_TH3 : WC8_TH; };
constexpr int WC8_CIB1 = 4;          // k = 1: 128 cin per workgroup
// k = 3: cin tiles per workgroup, each with its own 2*KS waves: 2 -> 64 cout x 64 cin, 12 waves = 3 per SIMD (balanced),
// twice the MFMAs per staged byte and per barrier
template <int KS> struct Wc8Cit { static constexpr int value = KS == 3 ? 2 : 1; };
// k = 5: 2 x 5 (cout tile, filter row) waves would sit 3/3/2/2 on the four SIMDs (15 tap-MFMAs per k-step on the busy ones
// against 12.5 on average).  The row waves therefore take taps dx = 0..3 only and two more waves (one per cout tile) take
// the COLUMN dx = 4 of all five rows: 12 waves, 12/12/13/13 per SIMD (waves i, i+4, i+8 share a SIMD).
template <int KS> struct Wc8Waves {
  static constexpr int rows = 2 * KS * Wc8Cit<KS>::value;
  static constexpr int value = KS == 1 ? 2 * WC8_TH : KS == 5 ? rows + 2 : rows;
};

template <int N, class F, int I = 0>
__device__ __forceinline__ void static_for_wc8(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_wc8<N, F, I + 1>(static_cast<F&&>(f));
  }
}

// Transposing LDS read as inline asm, for the LDS-DMA kernels: hipcc puts an `s_waitcnt vmcnt(0)` in front of every
// __builtin_amdgcn_ds_read_tr16_b64 that follows a `buffer_load ... lds` (it cannot tell that the DMA fills the OTHER
// buffer), which exposed the whole DMA latency once per tile (ISA of round 3's first DMA version; ablation: staging cost
// 1.3 of 6.8 ms).  The asm form is invisible to that rule; its completion is tracked by hand (wc8_wait_lgkm: LDS
// operations return in order, so "at most N younger reads outstanding" = the older ones have landed).
template <int OFF>
__device__ __forceinline__ s16x4 wc8_tr_read(unsigned addr) {
  s16x4 d;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
  return d;
}
// wait until at most N LDS reads are outstanding; the operands the caller is about to use are tied to the wait
template <int N>
__device__ __forceinline__ void wc8_wait_lgkm(s16x4 (&a)[2], s16x4 (&b)[3]) {
  asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]) : "n"(N));
}
template <int N>
__device__ __forceinline__ void wc8_wait_lgkm(s16x4 (&a)[2], s16x4 (&b)[10]) {
  asm volatile("s_waitcnt lgkmcnt(%12)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]),
                 "+v"(b[7]), "+v"(b[8]), "+v"(b[9])
               : "n"(N));
}

struct WgradC8Params {
  const uint4* x;
  const uint4* gy;
  float* ws;
  int H, W, cin, cout;
  long x_img, g_img, x_base, g_base;   // 16-byte vectors
  int tiles_x, nbands, nsplit;
  // k = 1, DG: the same pass also produces the conv's input gradient  gx = (W^T gy) * [x > 0]   (x = relu(...) is the
  // 1x1 conv's input: its own ReLU mask) from the tiles it has in LDS anyway
  const uint4* dg_w;                   // CODON_PACK_DGRAD image of the (64,128,1,1) weight: [ks][h][128 rows][8]
  uint4* dg_y;
  long dy_img, dy_base;
  // k = 1, DG, GB (round 4): `gy` is the block's dL/d(out) of one stream and the 1x1 conv's output gradient dL/d(pre) is
  // formed from it WHILE THE TILE IS STAGED -- the CAC gate backward (cac_bwd_apply, one stream of it):
  //   g_pre[c,p] = g_out[c,p] * (ch[c] * sp[p]) + g_avg[c] / HW + g_mean[p] / 128
  //                + [p == argpix[c]] g_max[c] + [c == argch[p]] g_cmax[p]
  // (direct term; global avg-pool broadcast; channel-mean broadcast; global max-pool routing; channel max-pool routing to
  // the first arg-max channel in Fcat order -- CAC_module.py:43,47,81), bit for bit the value cac_bwd_apply stores
  const float* gb_ch;                  // (B,64)
  const float* gb_sp;                  // (B,1,H,W)
  const float* gb_gpooled;             // (B,2,H,W): dL/d(channel max), dL/d(channel mean)
  const float* gb_gpools;              // (B,2,128): dL/d(global avg pool), dL/d(global max pool), Fcat channel order
  const int* gb_argpix;                // (B,128): arg-max pixel of every Fcat channel's global max pool
  const int* gb_argch;                 // (B,H,W): first arg-max Fcat channel of every pixel (cac_bwd_reduce, ACC)
  int gb_fbase;                        // this stream's first Fcat channel: 0 = colour, 64 = depth
  float gb_inv_hw;
};
static_assert(sizeof(WgradC8Params) <= CODON_KERNARG_LIMIT, "passed by value as a kernel argument");

// TAG_CI / TAG_CO: the channel shape of the launch, carried in the kernel NAME only (the code reads p.cin / p.cout): the
// 5x5 weight gradients of the 128 -> 128 and the 64 -> 64 convs run the same code on the same 256-workgroup grid, and
// per-shape rows in the rocprofv3 kernel statistics / PMC tables are what the roofline accounting needs (0 = untagged)
template <class E, int KS, bool DG = false, int TAG_CI = 0, int TAG_CO = 0, bool GB = false>
__global__ __launch_bounds__(Wc8Waves<KS>::value * 64, KS == 1 ? 2 : 3) void conv_wgrad_c8_kernel(const WgradC8Params p) {
  static_assert(!DG || KS == 1, "the fused input gradient exists for the 1x1 conv only");
  static_assert(!GB || DG, "the gate backward is formed in the staging of the fused 1x1 backward");
  typedef typename E::vec8 vec8;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  constexpr bool KSPLIT = (KS == 1);
  constexpr int CIT = KSPLIT ? 1 : Wc8Cit<KS>::value;   // 32-cin tiles taken by separate wave groups
  constexpr int CIB = KSPLIT ? WC8_CIB1 : CIT;          // 32-cin tiles per workgroup
  constexpr int NT = Wc8Waves<KS>::value * 64;
  constexpr bool BAL = (KS == 5);                      // row waves take KS-1 taps, two column waves the last one
  constexpr int NROWW = Wc8Waves<KS>::rows, NTAP = BAL ? KS - 1 : KS;
  constexpr int PAD = KS / 2;
  constexpr int TW = 32, TH = Wc8Th<KS>::value;
  constexpr int XC = KSPLIT ? TW : TW + 4;       // tile columns: origin tx0 - PAD; the 12-pixel windows reach column 35
  constexpr int XR = TH + KS - 1;
  constexpr int XPL = 4 * CIB, GPL = 8;          // 8-channel planes per tile
  constexpr int pitch64 = 64;
  constexpr int XPITCH = ((XR * XC * 16 - pitch64 + 255) / 256) * 256 + pitch64;   // == 64 (mod 256), >= plane bytes
  constexpr int GPITCH = ((TH * TW * 16 - pitch64 + 255) / 256) * 256 + pitch64;
  static_assert(XPITCH >= XR * XC * 16 && GPITCH >= TH * TW * 16, "plane pitch covers the plane");
  // k = 5 and k = 3: the tile is staged by LDS-DMA (buffer_load_dwordx4 ... lds: no staging registers, no ds_write).  A wave
  // instruction fills 64 consecutive 16-byte slots, so each tensor's region is a run of 1 KiB pieces over its PITCHED planes;
  // a lane carries the global offset of its slot, border and pitch-padding slots are out of range and land as zeros.
  // Measured (same box, with the asm reads and the tile pipeline below): 5x5 128->128 6.76 -> 6.12 ms, 5x5 64->64 1.79 -> 1.61,
  // 3x3 64->64 0.845 -> 0.77; the 2-k-step tiles of k = 1 (0.69 -> 0.71 with DMA) keep the register path.
  constexpr bool DMA = !KSPLIT;
  constexpr int XPS = XPITCH / 16, GPS = GPITCH / 16;       // plane pitch in slots
  constexpr int XPIECES = (XPL * XPS + 63) / 64, GPIECES = (GPL * GPS + 63) / 64;
  constexpr int XBYTES = XPIECES * 1024, GBYTES = GPIECES * 1024;
  constexpr int NPIECE = XPIECES + GPIECES, NWV = NT / 64;
  constexpr int PPW = (NPIECE + NWV - 1) / NWV;             // DMA: pieces per wave
  constexpr int NXE = XPL * XR * XC, NGE = GPL * TH * TW;   // register path: 16-byte elements per tile
  constexpr int XE = DMA ? 1 : (NXE + NT - 1) / NT, GE = DMA ? 1 : (NGE + NT - 1) / NT;
  constexpr int NK = TH * (TW / 16);

  // k = 5: THREE tile buffers -- the DMA of tile t+3 is issued behind tile t's MFMAs and has two tile times (not one) to
  // land; the wait in front of the publishing barrier leaves the youngest tile's pieces in flight (counted vmcnt: every
  // wave issues exactly PPW pieces per tile, loads return in order).  110 KB of LDS: the kernel runs one workgroup per
  // CU anyway (12 waves).
  constexpr int NBUF = (DMA && KS == 5 && TH == 4) ? 3 : 2;
  static_assert(NBUF == 2 || (NBUF == 3 && NPIECE % NWV == 0 && PPW <= 15), "counted vmcnt needs the same piece count in every wave");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NBUF * (XBYTES + GBYTES)];
  typedef __attribute__((address_space(3))) void lds_void;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int H = p.H, W = p.W;
  const long HW = (long)H * W;
  const unsigned HW16 = 16u * (unsigned)HW;

  const int nci_t = p.cin / (32 * CIB);
  // the channel blocks that read the SAME image band go to one XCD (they share the band in that L2)
  const unsigned vb_ = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const int bx_ = (int)(vb_ % gridDim.x), by_ = (int)(vb_ / gridDim.x);
  const int cob = bx_ / nci_t, cib = bx_ % nci_t;   // 64-cout block, 32*CIB-cin block
  const int split = by_;
  const int b = split / p.nbands, band = split % p.nbands;
  const int tiles_y = (H + TH - 1) / TH;
  const int ty_begin = (int)((long)band * tiles_y / p.nbands);          // tile rows spread evenly over the bands
  const int ty_end = (int)((long)(band + 1) * tiles_y / p.nbands);
  const int ntile = (ty_end - ty_begin) * p.tiles_x;

  const uint4* const xg = p.x + b * p.x_img + p.x_base + (long)cib * XPL * HW;
  const uint4* const gg = p.gy + b * p.g_img + p.g_base + (long)cob * GPL * HW;

  // staging plan (tile independent): this wave's pieces wave, wave + NWV, ...; slot = piece * 64 + lane
  unsigned rel[DMA ? PPW : 1];     // byte offset relative to the tile origin of the piece's tensor
  int rc[DMA ? PPW : 1];           // (row << 8) | col inside the tile, or -1: padding slot (never loaded: lands as zero)
  unsigned reli[DMA ? PPW : 1];    // the offset an INTERIOR tile uses (no border tests): rel, or out of range for padding
#pragma unroll
  for (int k = 0; k < (DMA ? PPW : 0); ++k) {
    const int pc = wave + NWV * k;
    const bool isx = pc < XPIECES;
    const int s_ = (isx ? pc : pc - XPIECES) * 64 + lane;
    const int ps = isx ? XPS : GPS, npl = isx ? XPL : GPL, pl_real = isx ? XR * XC : TH * TW, cols = isx ? XC : TW;
    const int plane = s_ / ps, idx = s_ - plane * ps;
    const bool in = pc < NPIECE && plane < npl && idx < pl_real;
    const int r = idx / cols, q = idx - r * cols;
    rel[k] = (unsigned)plane * HW16 + 16u * (unsigned)(r * W + q);
    rc[k] = in ? ((r << 8) | q) : -1;
    reli[k] = in ? rel[k] : C8_OOB;
  }

  // register path: staging plan (tile independent): element e = tid + NT k = (plane, row, col)
  unsigned xrel[XE], grel[GE];      // byte offset relative to the tile origin
  int xrc[XE], grc[GE];             // (row << 8) | col, or -1 for the padding elements of the last round
  int xlds[XE], glds[GE];           // LDS byte address inside a buffer
#pragma unroll
  for (int k = 0; k < (DMA ? 0 : XE); ++k) {
    const int e = tid + k * NT;
    const int q = e % XC, r = (e / XC) % XR, c = e / (XC * XR);
    const bool in = (NXE % NT == 0) || e < NXE;
    xrel[k] = (unsigned)c * HW16 + 16u * (unsigned)(r * W + q);
    xrc[k] = in ? ((r << 8) | q) : -1;
    xlds[k] = in ? c * XPITCH + (r * XC + q) * 16 : 0;
  }
#pragma unroll
  for (int k = 0; k < (DMA ? 0 : GE); ++k) {
    const int e = tid + k * NT;
    const int q = e % TW, r = (e / TW) % TH, c = e / (TW * TH);
    const bool in = (NGE % NT == 0) || e < NGE;
    grel[k] = (unsigned)c * HW16 + 16u * (unsigned)(r * W + q);
    grc[k] = in ? ((r << 8) | q) : -1;
    glds[k] = in ? c * GPITCH + (r * TW + q) * 16 : 0;
  }

  const bool colrole = BAL && wave >= NROWW;      // wave-uniform
  const int co_t = wave & 1, ci_t = KSPLIT || colrole ? 0 : (wave >> 1) % CIT, dy = KSPLIT || colrole ? 0 : (wave >> 1) / CIT;
  const int krow = wave >> 1;   // KSPLIT: the tile row whose k-steps this wave takes
  // transposing-read lane addresses: lane 4q+p of a 16-lane group supplies pixel q, channels 4p..4p+3 of the group's 16
  const int li = lane & 15, tq = li >> 2, tp = li & 3, cblk = (lane >> 4) & 1;
  const int a_lane = (co_t * 4 + 2 * cblk + (tp >> 1)) * GPITCH + (8 * half + tq) * 16 + (tp & 1) * 8;
  const int b_lane = (ci_t * 4 + 2 * cblk + (tp >> 1)) * XPITCH + (dy * XC + 8 * half + tq) * 16 + (tp & 1) * 8;

  constexpr int NACC = KSPLIT ? CIB : KS;
  f32x16 acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  auto stage_tile = [&](int t, int buf) {
    const int ty = ty_begin + t / p.tiles_x, tx = t % p.tiles_x;
    const int tx0 = tx * TW, ty0 = ty * TH;
    // the tile origin (possibly before the slice start: such slots are masked) goes into the descriptor base
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(xg + ((long)(ty0 - PAD) * W + (tx0 - PAD))), 0, (int)C8_OOB, C8_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(gg + ((long)ty0 * W + tx0)), 0, (int)C8_OOB, C8_RSRC_FLAGS);
    // 9 tiles in 10 touch no image border: their offsets are tile independent (3 instructions per piece instead of ~30)
    // (every STAGED column must be inside the image row -- the tile is XC columns wide from tx0 - PAD, which for KS = 3 is two
    // more than TW + 2 PAD: with `tx0 + TW + PAD <= W` a W % 32 == 1 image let them wrap into the next row, and past the end
    // of x on the last row of the last plane)
    const bool interior = ty0 >= PAD && ty0 + TH + PAD <= H && tx0 >= PAD && tx0 - PAD + XC <= W;   // wave-uniform
    if (interior) {
#pragma unroll
      for (int k = 0; k < PPW; ++k) {
        const int pc = wave + NWV * k;             // wave-uniform
        if (pc >= NPIECE) break;
        const unsigned vo_ = reli[k];
        lds_void* dst = (lds_void*)(lds + buf * (XBYTES + GBYTES) + pc * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(pc < XPIECES ? xr : gr, dst, 16, vo_, 0, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
      const int pc = wave + NWV * k;               // wave-uniform
      if (pc >= NPIECE) break;
      const bool isx = pc < XPIECES;
      const int pad = isx ? PAD : 0;
      const int gy_ = ty0 - pad + (rc[k] >> 8), gx_ = tx0 - pad + (rc[k] & 255);
      const bool ok = rc[k] >= 0 && gy_ >= 0 && gy_ < H && gx_ >= 0 && gx_ < W;
      const unsigned vo_ = ok ? rel[k] : C8_OOB;
      lds_void* dst = (lds_void*)(lds + buf * (XBYTES + GBYTES) + pc * 1024);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(isx ? xr : gr, dst, 16, vo_, 0, 0, 0);
    }
  };

  // GB: per-channel constants of this image in LDS as [plane][field][8]: ch, g_avg / HW, g_max, argpix (as int bits)
  __shared__ float gbt[GB ? 8 * 4 * 8 : 1];
  if constexpr (GB) {
    if (tid < 64) {
      const int c = tid, f = p.gb_fbase + c;
      float* row = gbt + (c >> 3) * 32 + (c & 7);
      row[0] = p.gb_ch[b * 64 + c];
      row[8] = p.gb_gpools[((long)b * 2 + 0) * 128 + f] * p.gb_inv_hw;
      row[16] = p.gb_gpools[((long)b * 2 + 1) * 128 + f];
      row[24] = __int_as_float(p.gb_argpix[(long)b * 128 + f]);
    }
    __syncthreads();
  }
  u32x4 xv[XE], gv[GE];
  float m_sp[GB ? GE : 1], m_mean[GB ? GE : 1], m_cmax[GB ? GE : 1];   // GB: the element's pixel maps (0 outside the image)
  int m_pix[GB ? GE : 1], m_arg[GB ? GE : 1];                           //     its pixel index (-1 outside) and arg-max channel
  auto load_tile = [&](int t) {
    const int ty = ty_begin + t / p.tiles_x, tx = t % p.tiles_x;
    const int tx0 = tx * TW, ty0 = ty * TH;
    // the tile origin (possibly before the slice start: such elements are masked) goes into the descriptor base
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(xg + ((long)(ty0 - PAD) * W + (tx0 - PAD))), 0, (int)C8_OOB, C8_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(gg + ((long)ty0 * W + tx0)), 0, (int)C8_OOB, C8_RSRC_FLAGS);
#pragma unroll
    for (int k = 0; k < XE; ++k) {
      const int gy_ = ty0 - PAD + (xrc[k] >> 8), gx_ = tx0 - PAD + (xrc[k] & 255);
      const bool ok = xrc[k] >= 0 && gy_ >= 0 && gy_ < H && gx_ >= 0 && gx_ < W;
      xv[k] = c8_ld(xr, ok ? xrel[k] : C8_OOB, 0);
    }
#pragma unroll
    for (int k = 0; k < GE; ++k) {
      const int gy_ = ty0 + (grc[k] >> 8), gx_ = tx0 + (grc[k] & 255);
      const bool ok = grc[k] >= 0 && gy_ < H && gx_ < W;
      gv[k] = c8_ld(gr, ok ? grel[k] : C8_OOB, 0);
      if constexpr (GB) {
        // a thread's elements are NT apart and a plane of the tile is TW * TH elements: when that divides NT they are all the
        // SAME pixel (of different 8-channel planes), whose maps are loaded once -- four loads per tile instead of four per plane
        constexpr bool SAME_PIXEL = NT % (TW * TH) == 0 && NGE % NT == 0;
        if (k == 0 || !SAME_PIXEL) {
          const long q = ok ? (long)gy_ * W + gx_ : 0;        // unconditional loads from a valid address, masked below
          const long bq = (long)b * HW + q;
          m_sp[k] = p.gb_sp[bq];
          m_cmax[k] = p.gb_gpooled[(long)b * 2 * HW + q];
          m_mean[k] = p.gb_gpooled[(long)b * 2 * HW + HW + q] * (1.f / 128.f);
          m_arg[k] = p.gb_argch[bq];
          m_pix[k] = ok ? (int)q : -1;
        } else {
          m_sp[k] = m_sp[0]; m_cmax[k] = m_cmax[0]; m_mean[k] = m_mean[0]; m_arg[k] = m_arg[0]; m_pix[k] = m_pix[0];
        }
      }
    }
  };
  auto store_tile = [&](int buf) {
    unsigned char* xs = lds + buf * (XBYTES + GBYTES);
    unsigned char* gs = xs + XBYTES;
#pragma unroll
    for (int k = 0; k < XE; ++k)
      if ((NXE % NT == 0) || tid + k * NT < NXE) *reinterpret_cast<u32x4*>(xs + xlds[k]) = xv[k];
#pragma unroll
    for (int k = 0; k < GE; ++k)
      if ((NGE % NT == 0) || tid + k * NT < NGE) {
        if constexpr (GB) {
          const int c = (tid + k * NT) / (TW * TH);           // 8-channel plane of the element (tile independent)
          const float* row = gbt + c * 32;
          const float4 c0 = *reinterpret_cast<const float4*>(row), c1 = *reinterpret_cast<const float4*>(row + 4),
                       a0 = *reinterpret_cast<const float4*>(row + 8), a1 = *reinterpret_cast<const float4*>(row + 12),
                       x0 = *reinterpret_cast<const float4*>(row + 16), x1 = *reinterpret_cast<const float4*>(row + 20),
                       i0 = *reinterpret_cast<const float4*>(row + 24), i1 = *reinterpret_cast<const float4*>(row + 28);
          const float chc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
          const float ga[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
          const float gm[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
          const float ai[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
          float go[8], o[8];
          c8_unpack<E>(gv[k], go);
          const bool in = m_pix[k] >= 0;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float v = go[j] * (chc[j] * m_sp[k]) + ga[j] + m_mean[k];      // the expression of cac_bwd_apply_c8_kernel
            if (m_pix[k] == __float_as_int(ai[j])) v += gm[j];
            if (m_arg[k] == p.gb_fbase + c * 8 + j) v += m_cmax[k];
            o[j] = in ? v : 0.f;
          }
          *reinterpret_cast<u32x4*>(gs + glds[k]) = c8_pack<E>(o);
        } else {
          *reinterpret_cast<u32x4*>(gs + glds[k]) = gv[k];
        }
      }
  };

  if constexpr (DMA) {
    // Software pipeline over tiles: the barrier that publishes tile t+1 sits at the START of tile t's last k-step (after
    // that step's operands are in registers = every read of buffer t&1 is done), so the first operands of tile t+1 are
    // requested BEHIND it and the DMA of tile t+2 is issued after the last step's MFMAs -- no drain at the tile edge.
    static_assert(NK % 2 == 0, "operand sets alternate: the first k-step of every tile uses set 0");
    const unsigned lds0 = (unsigned)(unsigned long)(lds_void*)lds;
    auto run = [&](auto rolec) {
      constexpr bool COL = decltype(rolec)::value;          // column wave (BAL): tap dx = KS-1 of every filter row
      constexpr int NB = COL ? 2 * KS : 3;                  // B reads per k-step
      s16x4 a2[2][2], wb[2][NB];
      auto read_step = [&](auto kc, auto sc, unsigned ab, unsigned bb) {
        constexpr int ks = decltype(kc)::value, set = decltype(sc)::value;
        constexpr int r_ = ks / (TW / 16), c0_ = (ks % (TW / 16)) * 16;
        constexpr int ao = XBYTES + (r_ * TW + c0_) * 16, bo = (r_ * XC + c0_) * 16;
        a2[set][0] = wc8_tr_read<ao>(ab);
        a2[set][1] = wc8_tr_read<ao + 64>(ab);
        static_for_wc8<NB>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          // row wave: pixels 0..11 of its row; column wave: pixels 4..11 (elements KS-1 .. KS+6) of row j/2
          constexpr int off = COL ? bo + (j / 2) * XC * 16 + 64 + (j % 2) * 64 : bo + j * 64;
          wb[set][j] = wc8_tr_read<off>(bb);
        });
      };
      if (ntile > 0) {
        stage_tile(0, 0);
        __builtin_amdgcn_s_waitcnt(0x0070);      // vmcnt(0): this wave's pieces have landed
        __syncthreads();
        read_step(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, lds0 + a_lane, lds0 + b_lane);
        if (ntile > 1) stage_tile(1, 1);
        if (NBUF == 3 && ntile > 2) stage_tile(2, 2);
      }
      int bi = 0;                                // buffer of tile t: t % NBUF
#pragma unroll 1
      for (int t = 0; t < ntile; ++t) {
        const int bn = bi + 1 == NBUF ? 0 : bi + 1;
        const unsigned cb = lds0 + bi * (XBYTES + GBYTES), nb = lds0 + bn * (XBYTES + GBYTES);
        static_for_wc8<NK>([&](auto kc) {
          constexpr int ks = decltype(kc)::value;
          constexpr int cur = ks & 1;
          if constexpr (ks + 1 < NK) {
            read_step(std::integral_constant<int, ks + 1>{}, std::integral_constant<int, cur ^ 1>{}, cb + a_lane, cb + b_lane);
            wc8_wait_lgkm<2 + NB>(a2[cur], wb[cur]);
          } else {
            wc8_wait_lgkm<0>(a2[cur], wb[cur]);
            // this wave's pieces of tile t+1 have landed (three buffers: tile t+2's PPW pieces may stay in flight)
            if (NBUF == 3 && t + 2 < ntile) __builtin_amdgcn_s_waitcnt(0x0070 | PPW);
            else __builtin_amdgcn_s_waitcnt(0x0070);
            __syncthreads();                     // ... everyone's have, and everyone is done reading tile t's buffer
            if (t + 1 < ntile) read_step(std::integral_constant<int, 0>{}, std::integral_constant<int, cur ^ 1>{}, nb + a_lane, nb + b_lane);
            __builtin_amdgcn_sched_barrier(0);
          }
          union { struct { s16x4 l, h; } s; vec8 v; } ua;
          ua.s.l = a2[cur][0]; ua.s.h = a2[cur][1];
          if constexpr (COL) {
#pragma unroll
            for (int r = 0; r < KS; ++r) {
              union { struct { s16x4 l, h; } s; vec8 v; } ub;
              ub.s.l = wb[cur][2 * r]; ub.s.h = wb[cur][2 * r + 1];
              acc[r] = E::mfma(ua.v, ub.v, acc[r]);
            }
          } else {
            // window words w[0..5] = pixels 0 .. 11 past (row, c0 + 8h) of the lane's channel; tap dx = elements dx .. dx+7
            union { s16x4 v[3]; unsigned w[6]; } uw;
            uw.v[0] = wb[cur][0]; uw.v[1] = wb[cur][1]; uw.v[2] = wb[cur][2];
            const unsigned* w = uw.w;
#pragma unroll
            for (int dx = 0; dx < NTAP; ++dx) {
              const int m = dx / 2;
              u32x4 f;
              if (dx % 2 == 0) {
                f = u32x4{w[m], w[m + 1], w[m + 2], w[m + 3]};
              } else {
                f = u32x4{__builtin_amdgcn_alignbit(w[m + 1], w[m], 16), __builtin_amdgcn_alignbit(w[m + 2], w[m + 1], 16),
                          __builtin_amdgcn_alignbit(w[m + 3], w[m + 2], 16), __builtin_amdgcn_alignbit(w[m + 4], w[m + 3], 16)};
              }
              acc[dx] = E::mfma(ua.v, *reinterpret_cast<const vec8*>(&f), acc[dx]);
            }
          }
          if constexpr (ks + 1 == NK) {            // behind the MFMAs: its address arithmetic runs beside them
            __builtin_amdgcn_sched_barrier(0);
            if (t + NBUF < ntile) stage_tile(t + NBUF, bi);
          }
        });
        bi = bn;
      }
    };
    if constexpr (BAL) {
      if (colrole) run(std::true_type{});
      else run(std::false_type{});
    } else {
      run(std::false_type{});
    }
  } else {
  // DG: this wave's A operands (rows = the 64 input channels 64 co_t .. +63 of the 1x1 conv, K = its 64 output channels)
  // stay in registers for the whole band; gx goes out through a descriptor on the image
  vec8 wa[DG ? 2 : 1][DG ? 4 : 1];
  const __amdgpu_buffer_rsrc_t dyr = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(DG ? p.dg_y + b * p.dy_img + p.dy_base : (uint4*)p.ws), 0, (int)(16u * HW16), C8_RSRC_FLAGS);
  if constexpr (DG) {
    const __amdgpu_buffer_rsrc_t wr_ = __builtin_amdgcn_make_buffer_rsrc((void*)p.dg_w, 0, 64 * 128 * 2, C8_RSRC_FLAGS);
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const u32x4 v = c8_ld(wr_, (unsigned)((half * 128 + (2 * co_t + t2) * 32 + l31) * 16), (unsigned)(ks * 2 * 128 * 16));
        wa[t2][ks] = *reinterpret_cast<const vec8*>(&v);
      }
  }
  if (ntile > 0) {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();

#define TR_READ(ptr_) __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ptr_))
#pragma unroll 1
  for (int t = 0; t < ntile; ++t) {
    const bool has_next = t + 1 < ntile;
    if (has_next) load_tile(t + 1);

    const unsigned char* xs = lds + (t & 1) * (XBYTES + GBYTES);
    const unsigned char* gs = xs + XBYTES;
    s16x4 a2[2][2], wd[2][3], wk[CIB][2];
#define WC8_READ(ks_, s_)                                                                    \
    {                                                                                        \
      const int r_ = (ks_) / (TW / 16), c0_ = ((ks_) % (TW / 16)) * 16;                      \
      const unsigned char* ap_ = gs + a_lane + (r_ * TW + c0_) * 16;                         \
      a2[s_][0] = TR_READ(ap_);                                                              \
      a2[s_][1] = TR_READ(ap_ + 64);                                                         \
      const unsigned char* bp_ = xs + b_lane + (r_ * XC + c0_) * 16;                         \
      if constexpr (KSPLIT) {                                                                \
        _Pragma("unroll") for (int j = 0; j < CIB; ++j) {                                    \
          wk[j][0] = TR_READ(bp_ + j * 4 * XPITCH);                                          \
          wk[j][1] = TR_READ(bp_ + j * 4 * XPITCH + 64);                                     \
        }                                                                                    \
      } else {                                                                               \
        wd[s_][0] = TR_READ(bp_);                                                            \
        wd[s_][1] = TR_READ(bp_ + 64);                                                       \
        wd[s_][2] = TR_READ(bp_ + 128);                                                      \
      }                                                                                      \
    }
    if constexpr (KSPLIT) {
#pragma unroll
      for (int c = 0; c < TW / 16; ++c) {
        WC8_READ(krow * (TW / 16) + c, 0)
        union { struct { s16x4 l, h; } s; vec8 v; } ua;
        ua.s.l = a2[0][0]; ua.s.h = a2[0][1];
#pragma unroll
        for (int j = 0; j < CIB; ++j) {
          union { struct { s16x4 l, h; } s; vec8 v; } ub;
          ub.s.l = wk[j][0]; ub.s.h = wk[j][1];
          acc[j] = E::mfma(ua.v, ub.v, acc[j]);
        }
      }
      if constexpr (DG) {
        // input gradient of tile row `krow`, channel tiles 2 co_t, 2 co_t + 1: B fragment of k-step ks = the 16-byte
        // vector of gy plane 2 ks + h at the lane's pixel (the LDS tile keeps the HBM form), 8 MFMAs; rows come out as
        // 8 consecutive channels per (tile, g) (swap23 packing), masked by x from the same LDS tile, one 16-byte store
        const int pix = krow * TW + l31;
        f32x16 d[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
          for (int r = 0; r < 16; ++r) d[t2][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const u32x4 bv = *reinterpret_cast<const u32x4*>(gs + (2 * ks + half) * GPITCH + pix * 16);
#pragma unroll
          for (int t2 = 0; t2 < 2; ++t2) d[t2] = E::mfma(wa[t2][ks], *reinterpret_cast<const vec8*>(&bv), d[t2]);
        }
        const int gy_ = (ty_begin + t / p.tiles_x) * TH + krow, gx_ = (t % p.tiles_x) * TW + l31;
        const unsigned vo = (gy_ < H && gx_ < W) ? (unsigned)half * HW16 + 16u * (unsigned)(gy_ * W + gx_) : C8_OOB;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            const int plane0 = (2 * co_t + t2) * 4 + 2 * g, plane = plane0 + half;     // plane0: wave-uniform
            const u32x4 rv = *reinterpret_cast<const u32x4*>(xs + plane * XPITCH + pix * 16);
            float r8[8], v8[8];
            c8_unpack<E>(rv, r8);
#pragma unroll
            for (int j = 0; j < 8; ++j) v8[j] = r8[j] > 0.f ? d[t2][8 * g + j] : 0.f;
            c8_st(c8_pack<E>(v8), dyr, vo, (unsigned)plane0 * HW16);
          }
      }
    } else {
      WC8_READ(0, 0)
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < NK) WC8_READ(ks + 1, cur ^ 1)
        union { struct { s16x4 l, h; } s; vec8 v; } ua;
        ua.s.l = a2[cur][0]; ua.s.h = a2[cur][1];
        // window words w[0..5] = pixels 0 .. 11 past (row, c0 + 8h) of the lane's channel; tap dx = elements dx .. dx+7
        union { s16x4 v[3]; unsigned w[6]; } uw;
        uw.v[0] = wd[cur][0]; uw.v[1] = wd[cur][1]; uw.v[2] = wd[cur][2];
        const unsigned* w = uw.w;
#pragma unroll
        for (int dx = 0; dx < KS; ++dx) {
          const int m = dx / 2;
          u32x4 f;
          if (dx % 2 == 0) {
            f = u32x4{w[m], w[m + 1], w[m + 2], w[m + 3]};
          } else {
            f = u32x4{__builtin_amdgcn_alignbit(w[m + 1], w[m], 16), __builtin_amdgcn_alignbit(w[m + 2], w[m + 1], 16),
                      __builtin_amdgcn_alignbit(w[m + 3], w[m + 2], 16), __builtin_amdgcn_alignbit(w[m + 4], w[m + 3], 16)};
          }
          acc[dx] = E::mfma(ua.v, *reinterpret_cast<const vec8*>(&f), acc[dx]);
        }
      }
    }
#undef WC8_READ
    if (has_next) store_tile((t + 1) & 1);
    __syncthreads();
  }
#undef TR_READ
  }   // !DMA

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
  for (int j = 0; j < KS; ++j) {
    if (BAL && !colrole && j >= NTAP) break;
    const int tap = colrole ? j * KS + (KS - 1) : dy * KS + j;      // column wave: accumulator j = filter row j, dx = KS-1
    const int ci = (cib * CIT + ci_t) * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cob * 64 + co_t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      wsp[((long)tap * p.cout + co) * p.cin + ci] = acc[j][r];
    }
  }
}

// defined in conv_wgrad_f32.hip: dw[co][ci][tap] (+)= sum_s ws[s][tap][co][ci], fixed order
int launch_wgrad_reduce(const float* ws, float* dw, int cout, int cin, int taps, int nsplit, int accumulate,
                        hipStream_t stream);

constexpr int WGRAD16_TARGET_BLOCKS = 256;   // workgroups per launch the band split aims for: one per CU

struct Wgrad16Plan {
  int nbands, nsplit, nchan_blocks;
};

static bool wgrad16_plan(const codon_conv_desc* d, Wgrad16Plan* pl) {
  const int k = d->ksize, ci = d->cin, co = d->cout;
  if (!((k == 1 || k == 3 || k == 5) && ci % 32 == 0 && co % 64 == 0)) return false;
  if (k == 1 && ci % (32 * WC8_CIB1) != 0) return false;
  if (k == 3 && ci % (32 * Wc8Cit<3>::value) != 0) return false;
  pl->nchan_blocks = (co / 64) * (ci / (k == 1 ? 32 * WC8_CIB1 : k == 3 ? 32 * Wc8Cit<3>::value : 32));
  const int th = k == 5 ? Wc8Th<5>::value : k == 3 ? Wc8Th<3>::value : WC8_TH;
  const int tiles_y = (d->height + th - 1) / th;
  int want = (WGRAD16_TARGET_BLOCKS + pl->nchan_blocks * d->batch - 1) / (pl->nchan_blocks * d->batch);
  if (want < 1) want = 1;
  if (want > tiles_y) want = tiles_y;
  pl->nbands = want;
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
    set_error("conv2d_wgrad: no 16-bit kernel for k=%d cin=%d cout=%d", d->ksize, d->cin, d->cout);
    return CODON_ERR_UNSUPPORTED;
  }
  CODON_REQUIRE(c8_slice_ok(d->x_ctotal, d->x_coff, d->cin) && c8_slice_ok(d->y_ctotal, d->y_coff, d->cout),
                CODON_ERR_BAD_ARG, "conv2d_wgrad: 16-bit tensors are channel-blocked: ctotal / coff multiples of 8");
  CODON_REQUIRE(ws_bytes >= conv_wgrad_bf16_workspace_bytes(d), CODON_ERR_BAD_ARG,
                "conv2d_wgrad: workspace %zu B < required %zu B", ws_bytes, conv_wgrad_bf16_workspace_bytes(d));
  CODON_REQUIRE(pl.nsplit <= 65535, CODON_ERR_UNSUPPORTED, "conv2d_wgrad: %d splits > 65535", pl.nsplit);
  const long HW = (long)d->height * d->width;
  CODON_REQUIRE(HW * 2 * 128 < (long)C8_OOB, CODON_ERR_UNSUPPORTED,
                "conv2d_wgrad: %dx%d image: 128 channels exceed the 4 GiB buffer-descriptor range", d->height, d->width);
  WgradC8Params p;
  p.x = (const uint4*)x; p.gy = (const uint4*)gy; p.ws = workspace;
  p.H = d->height; p.W = d->width; p.cin = d->cin; p.cout = d->cout;
  p.x_img = (d->x_ctotal / 8) * HW; p.g_img = (d->y_ctotal / 8) * HW;
  p.x_base = (d->x_coff / 8) * HW; p.g_base = (d->y_coff / 8) * HW;
  p.tiles_x = (d->width + 31) / 32;
  p.nbands = pl.nbands; p.nsplit = pl.nsplit;
  p.dg_w = nullptr; p.dg_y = nullptr; p.dy_img = p.dy_base = 0;
  p.gb_ch = p.gb_sp = p.gb_gpooled = p.gb_gpools = nullptr; p.gb_argpix = p.gb_argch = nullptr; p.gb_fbase = 0; p.gb_inv_hw = 0.f;
  const dim3 grid(pl.nchan_blocks, pl.nsplit);
  const bool f16 = d->dtype == CODON_F16;
  if (d->ksize == 5) {
    const dim3 blk(Wc8Waves<5>::value * 64);
    const int shape = d->cin * 1000 + d->cout;
    if (f16) hipLaunchKernelGGL((conv_wgrad_c8_kernel<C8F16, 5>), grid, blk, 0, stream, p);
    else if (shape == 128128) hipLaunchKernelGGL((conv_wgrad_c8_kernel<C8Bf16, 5, false, 128, 128>), grid, blk, 0, stream, p);
    else if (shape == 64064) hipLaunchKernelGGL((conv_wgrad_c8_kernel<C8Bf16, 5, false, 64, 64>), grid, blk, 0, stream, p);
    else hipLaunchKernelGGL((conv_wgrad_c8_kernel<C8Bf16, 5>), grid, blk, 0, stream, p);
  } else if (d->ksize == 3) {
    if (f16) hipLaunchKernelGGL((conv_wgrad_c8_kernel<C8F16, 3>), grid, dim3(Wc8Waves<3>::value * 64), 0, stream, p);
    else hipLaunchKernelGGL((conv_wgrad_c8_kernel<C8Bf16, 3>), grid, dim3(Wc8Waves<3>::value * 64), 0, stream, p);
  } else {
    if (f16) hipLaunchKernelGGL((conv_wgrad_c8_kernel<C8F16, 1>), grid, dim3(2 * WC8_TH * 64), 0, stream, p);
    else hipLaunchKernelGGL((conv_wgrad_c8_kernel<C8Bf16, 1>), grid, dim3(2 * WC8_TH * 64), 0, stream, p);
  }
  const int st = check_launch("conv_wgrad_c8_kernel");
  if (st != CODON_OK || accumulate == CODON_WGRAD_DEFER) return st;   // DEFER: the splits stay in the workspace (codon_reduce_multi)
  return launch_wgrad_reduce(workspace, dw, d->cout, d->cin, d->ksize * d->ksize, pl.nsplit, accumulate, stream);
}

// dL/dw AND dL/dx of a 1x1 conv 128 -> 64 whose input is a ReLU output (confuse / confuse_c / confuse_fuse,
// CODON_x4.py:84,83,127), one pass: both are HBM-bound on the same two tensors (x = r2 128 ch, gy 64 ch)
struct Conv1x1GateBwd {      // GB operands (see WgradC8Params); null = plain codon_conv1x1_bwd
  const float* ch; const float* sp; const float* g_pooled; const float* g_pools; const int* argpix; const int* argch;
  int fbase;
};
int conv1x1_bwd_16(const codon_conv_desc* d, const void* x, const void* gy, const void* w_dgrad, const codon_tensor* gx,
                   float* dw, float* workspace, size_t ws_bytes, int accumulate, hipStream_t stream,
                   const Conv1x1GateBwd* gb) {
  Wgrad16Plan pl;
  CODON_REQUIRE(d->ksize == 1 && d->cin == 128 && d->cout == 64 && wgrad16_plan(d, &pl), CODON_ERR_UNSUPPORTED,
                "conv1x1_bwd: k=%d cin=%d cout=%d (the fused pass exists for the 128 -> 64 1x1 convs)", d->ksize, d->cin, d->cout);
  CODON_REQUIRE(c8_slice_ok(d->x_ctotal, d->x_coff, d->cin) && c8_slice_ok(d->y_ctotal, d->y_coff, d->cout) &&
                    c8_slice_ok(gx->ctotal, gx->coff, d->cin),
                CODON_ERR_BAD_ARG, "conv1x1_bwd: 16-bit tensors are channel-blocked: ctotal / coff multiples of 8");
  CODON_REQUIRE(ws_bytes >= conv_wgrad_bf16_workspace_bytes(d), CODON_ERR_BAD_ARG,
                "conv1x1_bwd: workspace %zu B < required %zu B", ws_bytes, conv_wgrad_bf16_workspace_bytes(d));
  CODON_REQUIRE(pl.nsplit <= 65535, CODON_ERR_UNSUPPORTED, "conv1x1_bwd: %d splits > 65535", pl.nsplit);
  const long HW = (long)d->height * d->width;
  CODON_REQUIRE(HW * 2 * 128 < (long)C8_OOB, CODON_ERR_UNSUPPORTED,
                "conv1x1_bwd: %dx%d image: 128 channels exceed the 4 GiB buffer-descriptor range", d->height, d->width);
  WgradC8Params p;
  p.x = (const uint4*)x; p.gy = (const uint4*)gy; p.ws = workspace;
  p.H = d->height; p.W = d->width; p.cin = d->cin; p.cout = d->cout;
  p.x_img = (d->x_ctotal / 8) * HW; p.g_img = (d->y_ctotal / 8) * HW;
  p.x_base = (d->x_coff / 8) * HW; p.g_base = (d->y_coff / 8) * HW;
  p.tiles_x = (d->width + 31) / 32;
  p.nbands = pl.nbands; p.nsplit = pl.nsplit;
  p.dg_w = (const uint4*)w_dgrad; p.dg_y = (uint4*)gx->data;
  p.dy_img = (gx->ctotal / 8) * HW; p.dy_base = (gx->coff / 8) * HW;
  const dim3 grid(pl.nchan_blocks, pl.nsplit);
  if (gb) {
    p.gb_ch = gb->ch; p.gb_sp = gb->sp; p.gb_gpooled = gb->g_pooled; p.gb_gpools = gb->g_pools;
    p.gb_argpix = gb->argpix; p.gb_argch = gb->argch; p.gb_fbase = gb->fbase;
    p.gb_inv_hw = (float)(1.0 / (double)HW);                  // as cac_bwd_apply forms it
    if (d->dtype == CODON_F16)
      hipLaunchKernelGGL((conv_wgrad_c8_kernel<C8F16, 1, true, 0, 0, true>), grid, dim3(2 * WC8_TH * 64), 0, stream, p);
    else
      hipLaunchKernelGGL((conv_wgrad_c8_kernel<C8Bf16, 1, true, 0, 0, true>), grid, dim3(2 * WC8_TH * 64), 0, stream, p);
  } else {
    p.gb_ch = p.gb_sp = p.gb_gpooled = p.gb_gpools = nullptr; p.gb_argpix = p.gb_argch = nullptr; p.gb_fbase = 0; p.gb_inv_hw = 0.f;
    if (d->dtype == CODON_F16) hipLaunchKernelGGL((conv_wgrad_c8_kernel<C8F16, 1, true>), grid, dim3(2 * WC8_TH * 64), 0, stream, p);
    else hipLaunchKernelGGL((conv_wgrad_c8_kernel<C8Bf16, 1, true>), grid, dim3(2 * WC8_TH * 64), 0, stream, p);
  }
  const int st = check_launch("conv_wgrad_c8_kernel<1, dgrad>");
  if (st != CODON_OK || accumulate == CODON_WGRAD_DEFER) return st;
  return launch_wgrad_reduce(workspace, dw, d->cout, d->cin, 1, pl.nsplit, accumulate, stream);
}

}  // namespace codon
