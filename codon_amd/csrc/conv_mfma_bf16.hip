// bf16 implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate).
// Same conv as conv_mfma_f32.hip (nn.Conv2d stride 1 / pad k//2 / no bias of CODON_x4.py:24-47), for
// BASELINE.json configs[2] and [4] (bf16 activations and weights, fp32 master weights kept by the caller).
//
// Operand roles as in the fp32 kernel (A = weights -> cout rows, B = activations -> pixel columns), so
// the 32x32 result again has one pixel column per lane and 16 cout rows in registers: every accumulator
// register stores as 32 consecutive NCHW pixels of one cout plane.  What changes is K: one MFMA now
// consumes 16 input channels, 8 consecutive ones per lane (A: W[co][8h..8h+7], B: X[8h..8h+7][pix]), so
// both LDS images are channel-blocked:
//   xs[cb][row][col] : 16-byte elements = 8 channels of one pixel (NHWC-within-8);  B fragment of a
//                      half-wave = 32 consecutive elements of one tile row  -> conflict-free ds_read_b128
//   ws[dx][cb][cout] : 16-byte elements = 8 input channels of one (tap, cout);      A fragment likewise
// HBM stays NCHW (coalesced along W); the channel-blocking transpose happens while staging: a thread owns one
// 16-byte element (8 channels of one pixel), loads its 8 channel planes with 8 two-byte buffer loads (a wave = 64
// consecutive pixels of one plane per instruction) and writes the element with one conflict-free ds_write_b128.
// The packed weight image is produced once per weight version in exactly the ws stage order.
//
// K loop: stages = (chunk of 16 channels, filter row dy): KS taps x 1 MFMA k-step x (2 pixel rows x COUT/32)
// MFMAs per wave per stage; xs / ws double-buffered, next stage prefetched to registers before the MFMAs
// and written to LDS after them, one barrier per stage (same schedule as the fp32 kernel).

#include <hip/hip_bf16.h>

#include <stdlib.h>
#include <type_traits>

#include "codon_common.h"

namespace codon {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

// 16-bit element traits: storage is raw 16 bits either way; E picks the MFMA opcode and the conversions
struct EBf16 {
  typedef bf16x8 vec8;
  __device__ static f32x16 mfma(vec8 a, vec8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
  __device__ static float to_f32(u16 v) { return __uint_as_float((unsigned)v << 16); }
  __device__ static u16 from_f32(float f) { __bf16 b = (__bf16)f; return *reinterpret_cast<u16*>(&b); }
  __device__ static float lo(unsigned w) { return __uint_as_float(w << 16); }
  __device__ static float hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
};
struct EF16 {
  typedef f16x8 vec8;
  __device__ static f32x16 mfma(vec8 a, vec8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
  __device__ static float to_f32(u16 v) { return (float)*reinterpret_cast<const _Float16*>(&v); }
  __device__ static u16 from_f32(float f) { const _Float16 h = (_Float16)f; return *reinterpret_cast<const u16*>(&h); }
  __device__ static float lo(unsigned w) { return to_f32((u16)(w & 0xffffu)); }
  __device__ static float hi(unsigned w) { return to_f32((u16)(w >> 16)); }
};

struct Conv16Params {
  const u16* x;
  const uint4* w;  // packed: [chunk][dy][dx][cb in chunk (2)][cout] x 16 B
  u16* y;
  const u16* res;
  int H, W;
  long x_img, y_img, r_img;
  long x_base, y_base, r_base;
  int tiles_x, tiles_y, nblk;
  int flags;
#ifdef CODON_TIMING
  long long* dbg;
#endif
  // FUSE only: chained 1x1 (128 -> 64) from the accumulators, see conv_mfma_f32.hip
  const uint4* w2;  // [t2][t][g][lane] x 16 B: W1[t2*32 + (lane&31)][t*32 + 16g + (j&3) + 8(j>>2) + 4(lane>>5)], j = 0..7
  u16* y2;
  long y2_img, y2_base;
};

__device__ __forceinline__ float bf16_to_f32(u16 v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ u16 f32_to_bf16(float f) {
  // round-to-nearest-even via the compiler's native conversion (v_cvt_pk_bf16_f32 on gfx950; keeps NaN a NaN)
  __bf16 b = (__bf16)f;
  return *reinterpret_cast<u16*>(&b);
}

// VALU BUDGET (see conv_mfma_f32.hip): a VALU instruction costs 4 matrix-pipe cycles, i.e. 1/8 of a 32-cycle
// 16-bit MFMA, so this kernel keeps VALU out of the steady state altogether:
//   * every global access is a buffer instruction (SGPR channel / chunk / stage offset + one hoisted 32-bit VGPR
//     offset per element; out-of-image and padding elements are out of range: loads return 0, stores are dropped);
//   * the stage loop is unrolled over a chunk PAIR x KS filter rows, so both LDS double-buffer parities are compile
//     time and every ds_read_b128 / ds_write_b16 / ds_write_b128 is `base VGPR + immediate`;
//   * operand fetches are volatile LDS reads (never re-paired / re-based by the compiler) issued one filter tap
//     ahead of their MFMAs, held there by sched_barrier;
//   * the two channels of a staged word are written as two ds_write_b16 (no v_perm pack, no zero-fill select).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned BUF16_OOB = 0xFFFFFFF0u;
constexpr int BUF16_FLAGS = 0x00020000;

template <int N, class F, int I = 0>
__device__ __forceinline__ void static_for16(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for16<N, F, I + 1>(static_cast<F&&>(f));
  }
}

__device__ __forceinline__ float relu1_16(float v) {   // one v_max (fmaxf adds a canonicalising v_max(v, v))
  float o;
  asm("v_max_f32 %0, 0, %1" : "=v"(o) : "v"(v));
  return o;
}

enum { RES16_NONE = 0, RES16_ADD = 1, RES16_MASK = 2 };

// pixel rows per wave: the 5x5 64-cout kernel takes 4 (16x32 tile) so that, like the 128-cout ones, a filter tap is 8 MFMAs
// on 6 operand fetches (0.75 ds_read_b128 per MFMA instead of 1.0 -- LDS bandwidth is what these kernels run out of)
// Measured A/B on one box: conv5x5 64->64 2.21 -> 2.05 ms; the 3x3 convs (3 taps per stage) do not gain (1.09 -> 1.10).
#ifndef CODON_PSEG3
#define CODON_PSEG3 2
#endif
#ifndef CODON_RPS3
#define CODON_RPS3 1
#endif
template <int KS, int COUT> struct Conv16Pseg { static constexpr int value = (COUT == 64 && KS == 5) ? 4 : (COUT == 64 && KS == 3) ? CODON_PSEG3 : 2; };
// filter rows per stage: the 3x3 64-cout convs take all three (one barrier per 16-channel chunk, 36 MFMAs per wave
// between barriers instead of 12 -- with one row per stage these kernels spent their time at the barrier: 30 % of the
// matrix peak and 33 % of HBM, bound by neither)
template <int KS, int COUT> struct Conv16Rps { static constexpr int value = (KS == 3 && COUT == 64) ? CODON_RPS3 : 1; };

#ifndef CODON_OCC3
#define CODON_OCC3 2
#endif
// XW (W % 4 == 0, 8-byte aligned input slice): the halo tile is fetched with 8-byte loads -- 4 adjacent pixels of one
// channel plane per lane -- instead of 2-byte ones.  Measured (tools/probes/vmem_issue_probe.hip): a vector-memory
// instruction costs the CU ~3.4 ns whether a lane fetches 2 or 4 or 8 bytes, and the 64-cout kernels spent as long in
// 2-byte loads/stores (96 + 64 per wave and tile for 3x3 64->64) as in their MFMAs.  The tile's column origin moves to
// tx0 - 4 (8-byte aligned; 40 columns staged for the 32 + KS - 1 needed) and a thread owns 4 adjacent 16-byte elements.
template <class E, int KS, int CIN, int COUT, bool FUSE = false, bool XW = false>
__global__ __launch_bounds__(256, ((KS == 3 && COUT == 64) ? CODON_OCC3 : 2)) void conv_mfma_bf16_kernel(const Conv16Params p) {
  typedef typename E::vec8 vec8;
  typedef const volatile __attribute__((address_space(3))) u32x4* lds_rd;
  typedef volatile __attribute__((address_space(3))) u32x4* lds_w128;
  constexpr int PAD = KS / 2;
  constexpr int PSEG = Conv16Pseg<KS, COUT>::value;
  constexpr int TW = 32, TH = 4 * PSEG;
  constexpr int XLQ = XW ? 4 : PAD;               // columns staged left of the tile
  constexpr int XR = TH + KS - 1, XQ = XW ? TW + 8 : TW + KS - 1;
  constexpr int QR = XQ / 4;                      // XW: 4-pixel groups per tile row
  constexpr int CK = 16, NCB = CK / 8;
  constexpr int NCHUNK = CIN / CK;
  constexpr int RPS = Conv16Rps<KS, COUT>::value;   // filter rows per stage
  constexpr int SPC = KS / RPS;                     // stages per chunk
  static_assert(SPC * RPS == KS, "rows per stage divides the filter height");
  constexpr int XS = NCB * XR * XQ;       // 16-byte elements per input tile
  constexpr int WS = RPS * KS * NCB * COUT;     // 16-byte elements per weight stage
  constexpr int CT = COUT / 32;
  constexpr int NST = NCHUNK * SPC;
  constexpr int NQ = NCB * XR * QR;       // XW: 4-element groups per input tile
  constexpr int XE = XW ? (NQ + 255) / 256 : (XS + 255) / 256;   // staging rounds per chunk (XW: 4 elements per thread and round)
  constexpr int WE = (WS + 255) / 256;
  constexpr int XSP = ((XS + 255) / 256) * 256, WSP = WE * 256;   // padded to whole staging rounds (no store predicates)
  static_assert(NCHUNK % 2 == 0, "the stage loop is unrolled over chunk pairs");
  static_assert(2 * XSP * 16 < 65536 && 2 * WSP * 16 < 65536, "LDS immediates are 16 bits per region");

  __shared__ uint4 lds[2 * XSP + 2 * WSP];
  uint4* const xs0 = lds;
  uint4* const ws0 = lds + 2 * XSP;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;

  CODON_TSTAMP(p.dbg, 0)
  unsigned bid = xcd_remap(blockIdx.x, (unsigned)p.nblk);
  const int tx = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int ty = bid % p.tiles_y;
  const int b = bid / p.tiles_y;
  const int tx0 = tx * TW, ty0 = ty * TH;
  const int H = p.H, W = p.W;
  const unsigned HW2 = 2u * (unsigned)H * (unsigned)W;   // bytes per channel plane

  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.x + (long)b * p.x_img + p.x_base), 0, (int)((unsigned)CIN * HW2), BUF16_FLAGS);
  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)(NST * WS * 16), BUF16_FLAGS);

  // gather plan: 16-byte element e = tid + 256 k = (channel block cb, row r, col q) of the channel-blocked LDS image
  // xs[cb][r][q] -- the LDS index IS e, so a staging round is one conflict-free ds_write_b128 per thread at
  // `tid*16 + immediate` (the 4-byte-per-thread version wrote 16-byte-strided words: 8-way bank conflicts on the
  // resource this kernel is shortest of).  Its 8 channels come from 8 plane loads (wave = 64 consecutive pixels of
  // one plane per instruction) whose plane term is an SGPR.  xoff: byte offset of (plane cb*8, pixel) or out of range.
  unsigned xoff[XE];
  if constexpr (XW) {
    // group e = tid + 256 k = (channel block, row, 4-pixel group); W % 4 == 0: a group is wholly inside or outside
#pragma unroll
    for (int k = 0; k < XE; ++k) {
      const int e = tid + k * 256;
      const int cb = e / (XR * QR), rem = e - cb * (XR * QR);
      const int r = rem / QR, g = rem - r * QR;
      const int gy = ty0 + r - PAD, gx = tx0 + 4 * g - XLQ;
      const bool ok = e < NQ && gy >= 0 && gy < H && gx >= 0 && gx < W;
      xoff[k] = ok ? (unsigned)(8 * cb) * HW2 + 2u * (unsigned)(gy * W + gx) : BUF16_OOB;
    }
  } else {
    constexpr int DQ = 256 % XQ, DR = (256 / XQ) % XR, DC = (256 / XQ) / XR;
    int cb = tid / (XR * XQ);
    int rem = tid - cb * (XR * XQ);
    int r = rem / XQ, q = rem - r * XQ;
#pragma unroll
    for (int k = 0; k < XE; ++k) {
      const int gy = ty0 + r - PAD, gx = tx0 + q - PAD;
      const bool ok = cb < NCB && gy >= 0 && gy < H && gx >= 0 && gx < W;
      xoff[k] = ok ? (unsigned)(8 * cb) * HW2 + 2u * (unsigned)(gy * W + gx) : BUF16_OOB;
      q += DQ; r += DR; cb += DC;
      if (q >= XQ) { q -= XQ; r += 1; }
      if (r >= XR) { r -= XR; cb += 1; }
    }
  }
  const lds_w128 xwr = (lds_w128)(xs0 + (XW ? 4 * tid : tid));
  const unsigned wvo = (unsigned)tid * 16u;
  const unsigned wvo_last = (WS % 256 == 0 || tid + (WE - 1) * 256 < WS) ? wvo : BUF16_OOB;
  const lds_w128 ww = (lds_w128)(ws0 + tid);
  const lds_rd xrd = (lds_rd)(xs0 + (half * XR + wave * PSEG) * XQ + l31 + (XLQ - PAD));
  const lds_rd wrd = (lds_rd)(ws0 + half * COUT + l31);

  // the next chunk's halo tile is fetched in two halves, during the last two filter rows of the current chunk
  constexpr int XE1 = (KS >= 3 && SPC >= 2) ? XE / 2 : 0, XEH = XE - XE1;
  u16 xv[XW ? 1 : XEH][8];
  u32x2 xq[XW ? XEH : 1][8];           // XW: [round][channel] = pixels (0,1) | (2,3) of the lane's 4-pixel group
  u32x4 wr[WE];

#define LOAD_X(chunk_, k0_, k1_)                                                        \
  {                                                                                     \
    const unsigned so_ = (unsigned)(chunk_) * (unsigned)CK * HW2;                       \
    _Pragma("unroll") for (int k = (k0_); k < (k1_); ++k)                               \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                   \
        if constexpr (XW) {                                                             \
          const auto q_ = __builtin_amdgcn_raw_buffer_load_b64(xrsrc, xoff[k], so_ + (unsigned)j * HW2, 0); \
          xq[k - (k0_)][j] = *reinterpret_cast<const u32x2*>(&q_);                      \
        } else {                                                                        \
          xv[k - (k0_)][j] = __builtin_amdgcn_raw_buffer_load_b16(xrsrc, xoff[k], so_ + (unsigned)j * HW2, 0); \
        }                                                                               \
      }                                                                                 \
  }
#define STORE_X(buf_, k0_, k1_)   /* buf_ compile time: immediate offsets */            \
  {                                                                                     \
    _Pragma("unroll") for (int k = (k0_); k < (k1_); ++k) {                             \
      if constexpr (XW) {                                                               \
        if ((NQ % 256 == 0) || k + 1 < XE || tid + k * 256 < NQ) {                      \
          _Pragma("unroll") for (int i = 0; i < 4; ++i) {   /* pixel i of the group: channels 2w, 2w+1 -> word w */ \
            u32x4 v_;                                                                   \
            _Pragma("unroll") for (int w = 0; w < 4; ++w) {                             \
              const unsigned a_ = xq[k - (k0_)][2 * w][i >> 1], b_ = xq[k - (k0_)][2 * w + 1][i >> 1]; \
              v_[w] = (i & 1) ? ((a_ >> 16) | (b_ & 0xffff0000u)) : ((a_ & 0xffffu) | (b_ << 16)); \
            }                                                                           \
            xwr[(buf_) * XSP + k * 1024 + i] = v_;                                      \
          }                                                                             \
        }                                                                               \
      } else {                                                                          \
        u32x4 v_;                                                                       \
        _Pragma("unroll") for (int w = 0; w < 4; ++w)                                   \
          v_[w] = (unsigned)xv[k - (k0_)][2 * w] | ((unsigned)xv[k - (k0_)][2 * w + 1] << 16); \
        xwr[(buf_) * XSP + k * 256] = v_;                                               \
      }                                                                                 \
    }                                                                                   \
  }
#define LOAD_W(stage_)                                                                  \
  {                                                                                     \
    const unsigned so_ = (unsigned)(stage_) * (unsigned)(WS * 16);                      \
    _Pragma("unroll") for (int k = 0; k < WE; ++k) {                                    \
      const auto v_ = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, k == WE - 1 ? wvo_last : wvo, so_ + k * 4096u, 0); \
      wr[k] = *reinterpret_cast<const u32x4*>(&v_);                                     \
    }                                                                                   \
  }
#define STORE_W(buf_)                                                                   \
  {                                                                                     \
    _Pragma("unroll") for (int k = 0; k < WE; ++k) ww[(buf_) * WSP + k * 256] = wr[k];  \
  }

  f32x16 acc[PSEG][CT];
#pragma unroll
  for (int i = 0; i < PSEG; ++i)
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;

  if constexpr (XE1 > 0) {
    LOAD_X(0, 0, XE1);
    STORE_X(0, 0, XE1);
  }
  LOAD_X(0, XE1, XE);
  LOAD_W(0);
  STORE_X(0, XE1, XE);
  STORE_W(0);
  CODON_TSTAMP(p.dbg, 1)
  __syncthreads();
  CODON_TSTAMP(p.dbg, 2)

  // stage (chunk, dy): weights of filter row dy for 16 channels in ws[(chunk*KS + dy) & 1], the chunk's halo tile
  // in xs[chunk & 1].  Unrolled over (chunk parity, dy): KS odd, so the stage parity is (par + dy) & 1.
#pragma unroll 1
  for (int c2 = 0; c2 < NCHUNK; c2 += 2) {
    static_for16<2 * SPC>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      constexpr int par = u / SPC, dyg = u % SPC;                 // chunk parity, stage within the chunk
      constexpr int sbuf = (par * SPC + dyg) & 1;                 // c2 is even: stage parity is compile time
      const int chunk = c2 + par;
      const int s = chunk * SPC + dyg;
      constexpr bool tail = (par == 1 && dyg == SPC - 1);         // last stage of the pair
      const bool has_next = !tail || (c2 + 2 < NCHUNK);
      if (has_next) {
        LOAD_W(s + 1);
        if constexpr (dyg == SPC - 1) LOAD_X(chunk + 1, XE1, XE);
      }
      if constexpr (XE1 > 0 && dyg == SPC - 2) {
        if (!(par == 1 && c2 + 2 >= NCHUNK)) LOAD_X(chunk + 1, 0, XE1);
      }

      // Operand fetch one filter tap ahead of its MFMAs, in two register sets.  PIN: sched_barrier holds that order
      // (the scheduler otherwise sinks every fetch down to its use) -- measured on one box (A/B, same call): pinned
      // wins on the 8-MFMA taps of conv5x5-128 (6.59 vs 6.75 ms), unpinned on the 3x3 convs (1.13 vs 1.21 ms).
      constexpr bool PIN = (KS == 5 && PSEG * CT == 8);
      vec8 a[2][CT], bv[2][PSEG];
      // tap index q = rr * KS + dx walks the stage's RPS filter rows; dy = dyg * RPS + rr
#define FETCH_A(q_, t_)                                                                                   \
  {                                                                                                       \
    const u32x4 v_ = wrd[sbuf * WSP + (q_) * NCB * COUT + (t_) * 32];                                     \
    a[(q_) & 1][t_] = *reinterpret_cast<const vec8*>(&v_);                                                \
  }
#define FETCH_B(q_)                                                                                       \
  {                                                                                                       \
    _Pragma("unroll") for (int i = 0; i < PSEG; ++i) {                                                    \
      const u32x4 v_ = xrd[par * XSP + (dyg * RPS + (q_) / KS + i) * XQ + ((q_) % KS)];                   \
      bv[(q_) & 1][i] = *reinterpret_cast<const vec8*>(&v_);                                              \
    }                                                                                                     \
  }
      static_for16<CT>([&](auto tc) { FETCH_A(0, decltype(tc)::value) });
      FETCH_B(0)
      static_for16<RPS * KS>([&](auto dc) {
        constexpr int q = decltype(dc)::value;
        if constexpr (q + 1 < RPS * KS) {
          FETCH_B(q + 1)
          static_for16<CT>([&](auto tc) { FETCH_A(q + 1, decltype(tc)::value) });
        }
        if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
          for (int i = 0; i < PSEG; ++i) acc[i][t] = E::mfma(a[q & 1][t], bv[q & 1][i], acc[i][t]);
        if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
      });
#undef FETCH_A
#undef FETCH_B

      if constexpr (XE1 > 0 && dyg == SPC - 2) {
        if (!(par == 1 && c2 + 2 >= NCHUNK)) STORE_X(par ^ 1, 0, XE1);
      }
      if (has_next) {
        STORE_W(sbuf ^ 1);
        if constexpr (dyg == SPC - 1) STORE_X(par ^ 1, XE1, XE);
      }
      __syncthreads();
    });
  }
#undef LOAD_X
#undef STORE_X
#undef LOAD_W
#undef STORE_W
  CODON_TSTAMP(p.dbg, 3)

  // epilogue.  Lane term of every output address: pixel (row of this wave's segment i, column l31) of cout plane
  // 4*half; the cout term (t, r) is wave-uniform and goes into the SGPR offset.  Off-image pixels are out of range.
  const int gx = tx0 + l31;
  unsigned vo[PSEG];
#pragma unroll
  for (int i = 0; i < PSEG; ++i) {
    const int gy = ty0 + wave * PSEG + i;
    vo[i] = (gx < W && gy < H) ? (unsigned)(4 * half) * HW2 + 2u * (unsigned)(gy * W + gx) : BUF16_OOB;
  }
  const bool relu = p.flags & CODON_CONV_RELU;
  const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.res ? p.res + (long)b * p.r_img + p.r_base : p.x), 0, (int)((unsigned)(FUSE ? 64 : COUT) * HW2), BUF16_FLAGS);
  auto cplane = [&](int t, int r) { return (unsigned)(t * 32 + (r & 3) + 8 * (r >> 2)) * HW2; };

  if constexpr (FUSE) {
    // Chained 1x1 from the accumulators: registers 8g..8g+7 of a 32x32 D tile, rounded to 16 bits (exactly the
    // values the unfused path would have stored and reloaded), are the 8 k-values a lane feeds to the next
    // 32x32x16 MFMA as its B operand -- channels t*32 + 16g + {0,1,2,3,8,9,10,11} + 4*half; the packer permutes
    // W1 to that k order.  16 MFMAs per pixel row instead of a second kernel and an HBM round trip.
    static_assert(!FUSE || COUT == 128, "chained 1x1 is 128 -> 64");
    vec8 pk[PSEG][CT][2];
#pragma unroll
    for (int i = 0; i < PSEG; ++i)
#pragma unroll
      for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          u16 h8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float v = acc[i][t][8 * g + j];
            h8[j] = E::from_f32(relu ? relu1_16(v) : v);
          }
          pk[i][t][g] = *reinterpret_cast<const vec8*>(h8);
        }
    if (p.y) {
      const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(p.y + (long)b * p.y_img + p.y_base), 0, (int)((unsigned)COUT * HW2), BUF16_FLAGS);
#pragma unroll
      for (int i = 0; i < PSEG; ++i)
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            const u16* h8 = reinterpret_cast<const u16*>(&pk[i][t][g]);
#pragma unroll
            for (int j = 0; j < 8; ++j) __builtin_amdgcn_raw_buffer_store_b16(h8[j], yrsrc, vo[i], cplane(t, 8 * g + j), 0);
          }
    }
    const __amdgpu_buffer_rsrc_t w2rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, 64 * 128 * 2, BUF16_FLAGS);
    const __amdgpu_buffer_rsrc_t y2rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.y2 + (long)b * p.y2_img + p.y2_base), 0, (int)(64u * HW2), BUF16_FLAGS);
    const unsigned w2vo = (unsigned)lane * 16u;
    f32x16 d[2][PSEG];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
      for (int i = 0; i < PSEG; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) d[t2][i][r] = 0.f;
#pragma unroll
      for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const auto av = __builtin_amdgcn_raw_buffer_load_b128(w2rsrc, w2vo, (unsigned)(((t2 * CT + t) * 2 + g) * 1024), 0);
          const vec8 a = *reinterpret_cast<const vec8*>(&av);
#pragma unroll
          for (int i = 0; i < PSEG; ++i) d[t2][i] = E::mfma(a, pk[i][t][g], d[t2][i]);
        }
    }
#pragma unroll
    for (int i = 0; i < PSEG; ++i)
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        if (p.res) {
          float rv[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) rv[r] = E::to_f32(__builtin_amdgcn_raw_buffer_load_b16(rrsrc, vo[i], cplane(t2, r), 0));
#pragma unroll
          for (int r = 0; r < 16; ++r)
            __builtin_amdgcn_raw_buffer_store_b16(E::from_f32(d[t2][i][r] + rv[r]), y2rsrc, vo[i], cplane(t2, r), 0);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            __builtin_amdgcn_raw_buffer_store_b16(E::from_f32(d[t2][i][r]), y2rsrc, vo[i], cplane(t2, r), 0);
        }
      }
    CODON_TSTAMP(p.dbg, 4)
    return;
  }

  // ReLU / residual / accumulate as compile-time variants selected by wave-uniform branches
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.y + (long)b * p.y_img + p.y_base), 0, (int)((unsigned)COUT * HW2), BUF16_FLAGS);
  auto epi = [&](auto relu_c, auto res_c, auto acc_c) {
    constexpr bool RELU = decltype(relu_c)::value;
    constexpr int RES = decltype(res_c)::value;
    constexpr bool ACC = decltype(acc_c)::value;
#pragma unroll
    for (int i = 0; i < PSEG; ++i) {
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        u16 rv[16], av[16];
        if constexpr (RES != RES16_NONE) {
#pragma unroll
          for (int r = 0; r < 16; ++r) rv[r] = __builtin_amdgcn_raw_buffer_load_b16(rrsrc, vo[i], cplane(t, r), 0);
        }
        if constexpr (ACC) {
#pragma unroll
          for (int r = 0; r < 16; ++r) av[r] = __builtin_amdgcn_raw_buffer_load_b16(yrsrc, vo[i], cplane(t, r), 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[i][t][r];
          if constexpr (RELU) v = relu1_16(v);
          if constexpr (RES == RES16_ADD) v += E::to_f32(rv[r]);
          if constexpr (RES == RES16_MASK) v = E::to_f32(rv[r]) > 0.f ? v : 0.f;
          if constexpr (ACC) v += E::to_f32(av[r]);
          __builtin_amdgcn_raw_buffer_store_b16(E::from_f32(v), yrsrc, vo[i], cplane(t, r), 0);
        }
      }
    }
  };
  using T = std::true_type;
  using F = std::false_type;
  using R0 = std::integral_constant<int, RES16_NONE>;
  using R1 = std::integral_constant<int, RES16_ADD>;
  using R2 = std::integral_constant<int, RES16_MASK>;
  const int res_mode = !p.res ? RES16_NONE : (p.flags & CODON_CONV_MASK_RELU) ? RES16_MASK
                                           : (p.flags & CODON_CONV_ADD_RESIDUAL) ? RES16_ADD : RES16_NONE;
  const bool accum = p.flags & CODON_CONV_ACCUM_OUT;
  auto by_acc = [&](auto relu_c, auto res_c) {
    if (accum) epi(relu_c, res_c, T{});
    else epi(relu_c, res_c, F{});
  };
  auto by_res = [&](auto relu_c) {
    if (res_mode == RES16_NONE) by_acc(relu_c, R0{});
    else if (res_mode == RES16_ADD) by_acc(relu_c, R1{});
    else by_acc(relu_c, R2{});
  };
  if (relu) by_res(T{});
  else by_res(F{});
  CODON_TSTAMP(p.dbg, 4)
}

// ---- 1x1 convolution (confuse / confuse_c / confuse_fuse and their dgrad): HBM-bound -----------------
// Y[co][pix] = sum_ci W[co][ci] X[ci][pix] is a plain GEMM over the flattened pixels of one image: no halo,
// so no LDS at all.  A wave owns 64 consecutive pixels (two 32-pixel MFMA column tiles) and all COUT rows.
// B fragment of lane (pixel l&31, half h) for k-step ks = channels 16ks+8h .. +7 of that pixel: eight 2-byte
// loads straight from the NCHW planes (each plane contributes a 64-byte run per half-wave; the two column
// tiles of the wave use the two halves of every 128-byte line).  A fragment = 16 bytes of the packed weight
// image (<= 16 KB, L1/L2 resident), loaded straight from global.  Epilogue identical to the k x k kernel.
// PAIRED = true (HW even): the wave's two 32-column MFMA tiles are the EVEN and the ODD pixels of its 64-pixel
// run, so one 4-byte load per lane (pixel pair 2j, 2j+1 of one channel plane) feeds both tiles and a half-wave
// touches one full 128-byte line per instruction; outputs / residual / mask are 4-byte accesses likewise.
// PAIRED = false: tiles are pixels [0,32) and [32,64), 2-byte accesses (any HW).
constexpr int C1_ITER = 1;  // >1 measured slower (4: 5.4 vs 3.2 ms): the wave is latency-bound, more workgroups hide it better
template <class E, int CIN, int COUT, bool PAIRED>
__global__ __launch_bounds__(256) void conv1x1_bf16_kernel(const Conv16Params p) {
  typedef typename E::vec8 vec8;
  constexpr int NKS = CIN / 16, CT = COUT / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const long HW = (long)p.H * p.W;
  const int b = blockIdx.y;
  const u16* __restrict__ xg = p.x + (long)b * p.x_img + p.x_base;
  const uint4* __restrict__ wg = p.w;                      // [chunk=ks][dy=0][dx=0][cb (2)][cout] x 16 B
  u16* __restrict__ yg = p.y + (long)b * p.y_img + p.y_base;
  const u16* __restrict__ rg = p.res ? p.res + (long)b * p.r_img + p.r_base : nullptr;
  const bool relu = p.flags & CODON_CONV_RELU;
  const bool addr = (p.flags & CODON_CONV_ADD_RESIDUAL) && rg;
  const bool accum = p.flags & CODON_CONV_ACCUM_OUT;
  const bool mask = (p.flags & CODON_CONV_MASK_RELU) && rg;

  // a workgroup owns C1_ITER * 256 consecutive pixels: each plane sees a 2 KB contiguous run per workgroup
  // (DRAM-page friendly for the 64..256 channel planes touched); waves interleave 64-pixel groups
#pragma unroll 1
  for (int it = 0; it < C1_ITER; ++it) {
  const long pix0 = (((long)blockIdx.x * C1_ITER + it) * 4 + wave) * 64;   // first pixel of this wave's group
  if (pix0 >= HW) break;                                   // wave-uniform

  // PAIRED: this lane's pixel pair starts at pp (tile 0 = pp, tile 1 = pp + 1); else tile i pixel = px[i]
  const long pp = pix0 + 2 * l31;
  const bool pok = pp < HW;                                // HW even: both pixels in or out together
  long px[2];
  bool ok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    px[i] = PAIRED ? pp + i : pix0 + i * 32 + l31;
    ok[i] = px[i] < HW;
  }

  // ALL activation loads of the wave are issued before the first MFMA (NKS*8 registers): one memory round trip
  // per wave instead of one per k-step (hipcc otherwise emits {8 loads, wait, MFMAs} per k-step)
  unsigned dd[NKS][PAIRED ? 8 : 16];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    if constexpr (PAIRED) {
      const u16* src = xg + (long)(ks * 16 + half * 8) * HW + (pok ? pp : 0);
#pragma unroll
      for (int j = 0; j < 8; ++j) dd[ks][j] = *reinterpret_cast<const unsigned*>(src + j * HW);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const u16* src = xg + (long)(ks * 16 + half * 8) * HW + (ok[i] ? px[i] : 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) dd[ks][i * 8 + j] = src[j * HW];
      }
    }
  }
  // B fragments of k-step ks from the raw words (rebuilt per cout half: a few VALU ops, fewer live registers)
  auto make_bv = [&](const int ks, vec8* bv) {
    if constexpr (PAIRED) {
      unsigned e[4], o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned d0 = pok ? dd[ks][2 * j] : 0u, d1 = pok ? dd[ks][2 * j + 1] : 0u;
        e[j] = (d0 & 0xffffu) | (d1 << 16);          // even pixel: channels 2j, 2j+1
        o[j] = (d0 >> 16) | (d1 & 0xffff0000u);      // odd pixel
      }
      const uint4 ve = make_uint4(e[0], e[1], e[2], e[3]), vo = make_uint4(o[0], o[1], o[2], o[3]);
      bv[0] = *reinterpret_cast<const vec8*>(&ve);
      bv[1] = *reinterpret_cast<const vec8*>(&vo);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        unsigned w4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned lo = ok[i] ? dd[ks][i * 8 + 2 * j] : 0u, hi = ok[i] ? dd[ks][i * 8 + 2 * j + 1] : 0u;
          w4[j] = lo | (hi << 16);
        }
        const uint4 v = make_uint4(w4[0], w4[1], w4[2], w4[3]);
        bv[i] = *reinterpret_cast<const vec8*>(&v);
      }
    }
  };
  // The COUT rows are processed CTB tiles (64 couts) at a time: the residual / mask / accumulate words of a half are
  // requested BEFORE its MFMAs, so a wave makes one memory round trip per half (the all-CT version with its
  // register-limited one-tile batches made CT + 1 serialized trips: the 64 -> 128 dgrad ran at 1.9 TB/s).
  constexpr int CTB = CT > 2 ? 2 : CT;
  auto half_pass = [&](auto has_r, auto has_acc, const int t0) {
    unsigned rm[CTB][16], am[CTB][16];      // PAIRED: one word = the lane's pixel pair; else [i * 8 + ..] halves
    u16 rs[PAIRED ? 1 : 2][CTB][16], as[PAIRED ? 1 : 2][CTB][16];
    auto fetch_words = [&]() {
    if constexpr (PAIRED) {
      if constexpr (decltype(has_r)::value) {
#pragma unroll
        for (int g = 0; g < CTB; ++g)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            rm[g][r] = *reinterpret_cast<const unsigned*>(rg + ((t0 + g) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * HW + (pok ? pp : 0));
      }
      if constexpr (decltype(has_acc)::value) {
#pragma unroll
        for (int g = 0; g < CTB; ++g)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            am[g][r] = *reinterpret_cast<const unsigned*>(yg + ((t0 + g) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * HW + (pok ? pp : 0));
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < CTB; ++g)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const long o = ((t0 + g) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * HW + (ok[i] ? px[i] : 0);
            if constexpr (decltype(has_r)::value) rs[i][g][r] = rg[o];
            if constexpr (decltype(has_acc)::value) as[i][g][r] = yg[o];
          }
    }
    };
    if constexpr (CT > 2) fetch_words();   // two passes: request this half's words ahead of its MFMAs
    f32x16 acc[2][CTB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < CTB; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][g][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      vec8 a[CTB], bv[2];
#pragma unroll
      for (int g = 0; g < CTB; ++g) {
        const uint4 v = wg[(ks * 2 + half) * COUT + (t0 + g) * 32 + l31];
        a[g] = *reinterpret_cast<const vec8*>(&v);
      }
      make_bv(ks, bv);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < CTB; ++g) acc[i][g] = E::mfma(a[g], bv[i], acc[i][g]);
    }
    if constexpr (CT <= 2) fetch_words();  // single pass: after the MFMAs (fewer live registers; measured faster)
    if constexpr (PAIRED) {
      if (pok) {
#pragma unroll
        for (int g = 0; g < CTB; ++g) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = (t0 + g) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            float v0 = acc[0][g][r], v1 = acc[1][g][r];
            if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
            if constexpr (decltype(has_r)::value) {
              const float m0 = E::lo(rm[g][r]), m1 = E::hi(rm[g][r]);
              if (addr) { v0 += m0; v1 += m1; }
              if (mask) { v0 = m0 > 0.f ? v0 : 0.f; v1 = m1 > 0.f ? v1 : 0.f; }
            }
            if constexpr (decltype(has_acc)::value) { v0 += E::lo(am[g][r]); v1 += E::hi(am[g][r]); }
            *reinterpret_cast<unsigned*>(yg + co * HW + pp) = (unsigned)E::from_f32(v0) | ((unsigned)E::from_f32(v1) << 16);
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (!ok[i]) continue;
#pragma unroll
        for (int g = 0; g < CTB; ++g)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = (t0 + g) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            float v = acc[i][g][r];
            if (relu) v = fmaxf(v, 0.f);
            if constexpr (decltype(has_r)::value) {
              const float m = E::to_f32(rs[i][g][r]);
              if (addr) v += m;
              if (mask) v = m > 0.f ? v : 0.f;
            }
            if constexpr (decltype(has_acc)::value) v += E::to_f32(as[i][g][r]);
            yg[co * HW + px[i]] = E::from_f32(v);
          }
      }
    }
  };
  const bool has_r = addr || mask;
#pragma unroll
  for (int t0 = 0; t0 < CT; t0 += CTB) {
    if (has_r && accum) half_pass(std::true_type{}, std::true_type{}, t0);
    else if (has_r) half_pass(std::true_type{}, std::false_type{}, t0);
    else if (accum) half_pass(std::false_type{}, std::true_type{}, t0);
    else half_pass(std::false_type{}, std::false_type{}, t0);
  }
  }  // it
}

template <class E, int CIN, int COUT>
static int launch_conv1x1_16(const codon_conv_desc* d, const void* x, const void* w, void* y, const void* res,
                             hipStream_t stream) {
  Conv16Params p;
  p.x = (const u16*)x; p.w = (const uint4*)w; p.y = (u16*)y; p.res = (const u16*)res;
  p.H = d->height; p.W = d->width;
  const long HW = (long)d->height * d->width;
  p.x_img = d->x_ctotal * HW; p.y_img = d->y_ctotal * HW; p.r_img = d->r_ctotal * HW;
  p.x_base = d->x_coff * HW; p.y_base = d->y_coff * HW; p.r_base = d->r_coff * HW;
  p.tiles_x = p.tiles_y = p.nblk = 0;
  p.flags = d->flags;
  CODON_REQUIRE(d->batch <= 65535, CODON_ERR_UNSUPPORTED, "conv2d_fwd: batch %d > 65535", d->batch);
  const unsigned gx = (unsigned)((HW + 256 * C1_ITER - 1) / (256 * C1_ITER));
  const bool al4 = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) |
                     reinterpret_cast<uintptr_t>(res)) % 4) == 0;
  if (HW % 2 == 0 && al4)
    hipLaunchKernelGGL((conv1x1_bf16_kernel<E, CIN, COUT, true>), dim3(gx, d->batch), dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL((conv1x1_bf16_kernel<E, CIN, COUT, false>), dim3(gx, d->batch), dim3(256), 0, stream, p);
  return check_launch("conv1x1_bf16_kernel");
}

// OIHW fp32 -> bf16 packed [chunk][dy][dx][cb (2)][cout][8 ch]; DGRAD: flipped taps, in/out swapped.
template <class E>
__global__ void pack_weight_bf16_kernel(const float* __restrict__ w, u16* __restrict__ out, int cout, int cin, int ks,
                                        int dgrad) {
  const int kin = dgrad ? cout : cin, kout = dgrad ? cin : cout;
  const long n = (long)kin * kout * ks * ks;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long t = i;
    const int j = t % 8; t /= 8;
    const int o = t % kout; t /= kout;
    const int cb = t % 2; t /= 2;
    const int dx = t % ks; t /= ks;
    const int dy = t % ks; t /= ks;
    const int chunk = (int)t;
    const int ci = chunk * 16 + cb * 8 + j;
    float v;
    if (!dgrad) v = w[(((long)o * cin + ci) * ks + dy) * ks + dx];
    else v = w[(((long)ci * cin + o) * ks + (ks - 1 - dy)) * ks + (ks - 1 - dx)];
    out[i] = E::from_f32(v);
  }
}

// OIHW (64,128,1,1) fp32 -> the chained-1x1 A-operand image [t2][t][g][lane][8]
template <class E>
__global__ void pack_chain1x1_16_kernel(const float* __restrict__ w, u16* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // 64 * 128 values
  if (i >= 64 * 128) return;
  const int j = i & 7, lane = (i >> 3) & 63, g = (i >> 9) & 1, t = (i >> 10) & 3, t2 = i >> 12;
  const int co2 = t2 * 32 + (lane & 31);
  const int c = t * 32 + 16 * g + (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5);
  out[i] = E::from_f32(w[co2 * 128 + c]);
}

int pack_chain1x1_16(const float* w, void* out, int dtype, hipStream_t stream) {
  if (dtype == CODON_F16)
    hipLaunchKernelGGL(pack_chain1x1_16_kernel<EF16>, dim3(32), dim3(256), 0, stream, w, (u16*)out);
  else
    hipLaunchKernelGGL(pack_chain1x1_16_kernel<EBf16>, dim3(32), dim3(256), 0, stream, w, (u16*)out);
  return check_launch("pack_chain1x1_16_kernel");
}

// 8-byte input loads need W % 4 == 0 (every plane row then starts 8-byte aligned relative to the slice) and an 8-byte
// aligned slice start; CODON_CONV16_XW=0 forces the 2-byte path (A/B)
static bool conv16_xwide(const codon_conv_desc* d, const void* x) {
  static const bool env = getenv("CODON_CONV16_XW") ? atoi(getenv("CODON_CONV16_XW")) != 0 : true;
  const long HW = (long)d->height * d->width;
  return env && d->width % 4 == 0 && (reinterpret_cast<uintptr_t>(x) + 2 * (uintptr_t)(d->x_coff * HW)) % 8 == 0 &&
         (2 * d->x_ctotal * HW) % 8 == 0;
}

template <class E>
static int launch_chain16(const codon_conv_desc* d, const void* x, const void* w, void* y, const void* w_chain,
                          const codon_tensor* out, const codon_tensor* res, hipStream_t stream) {
  Conv16Params p;
  p.x = (const u16*)x; p.w = (const uint4*)w; p.y = (u16*)y; p.res = res ? (const u16*)res->data : nullptr;
  p.H = d->height; p.W = d->width;
  const long HW = (long)d->height * d->width;
  p.x_img = d->x_ctotal * HW; p.y_img = d->y_ctotal * HW; p.r_img = res ? res->ctotal * HW : 0;
  p.x_base = d->x_coff * HW; p.y_base = d->y_coff * HW; p.r_base = res ? res->coff * HW : 0;
  p.w2 = (const uint4*)w_chain; p.y2 = (u16*)out->data; p.y2_img = out->ctotal * HW; p.y2_base = out->coff * HW;
  p.tiles_x = (d->width + 31) / 32;
  p.tiles_y = (d->height + 7) / 8;
  const long nblk = (long)p.tiles_x * p.tiles_y * d->batch;
  CODON_REQUIRE(nblk < (1L << 31), CODON_ERR_UNSUPPORTED, "conv_chain1x1_fwd: grid too large (%ld blocks)", nblk);
  p.nblk = (int)nblk;
  p.flags = d->flags;
#ifdef CODON_TIMING
  p.dbg = codon_dbg_ptr();
#endif
  if (conv16_xwide(d, x))
    hipLaunchKernelGGL((conv_mfma_bf16_kernel<E, 5, 128, 128, true, true>), dim3((unsigned)nblk), dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL((conv_mfma_bf16_kernel<E, 5, 128, 128, true>), dim3((unsigned)nblk), dim3(256), 0, stream, p);
  return check_launch("conv_mfma_bf16_kernel<fused 1x1>");
}

int conv_chain1x1_fwd_16(const codon_conv_desc* d, const void* x, const void* w, void* y, const void* w_chain,
                         const codon_tensor* out, const codon_tensor* res, hipStream_t stream) {
  CODON_REQUIRE(d->ksize == 5 && d->cin == 128 && d->cout == 128, CODON_ERR_UNSUPPORTED,
                "conv_chain1x1_fwd: 16-bit kernel is conv5x5 128->128 + 1x1 128->64 (got k=%d %d->%d)", d->ksize, d->cin, d->cout);
  return d->dtype == CODON_F16 ? launch_chain16<EF16>(d, x, w, y, w_chain, out, res, stream)
                               : launch_chain16<EBf16>(d, x, w, y, w_chain, out, res, stream);
}

template <class E, int KS, int CIN, int COUT>
static int launch_conv16(const codon_conv_desc* d, const void* x, const void* w, void* y, const void* res,
                         hipStream_t stream) {
  Conv16Params p;
  p.x = (const u16*)x; p.w = (const uint4*)w; p.y = (u16*)y; p.res = (const u16*)res;
  p.H = d->height; p.W = d->width;
  const long HW = (long)d->height * d->width;
  p.x_img = d->x_ctotal * HW; p.y_img = d->y_ctotal * HW; p.r_img = d->r_ctotal * HW;
  p.x_base = d->x_coff * HW; p.y_base = d->y_coff * HW; p.r_base = d->r_coff * HW;
  constexpr int TH = 4 * Conv16Pseg<KS, COUT>::value;
  p.tiles_x = (d->width + 31) / 32;
  p.tiles_y = (d->height + TH - 1) / TH;
  const long nblk = (long)p.tiles_x * p.tiles_y * d->batch;
  CODON_REQUIRE(nblk < (1L << 31), CODON_ERR_UNSUPPORTED, "conv2d_fwd: grid too large (%ld blocks)", nblk);
  CODON_REQUIRE(HW * 2 * 128 < (long)BUF16_OOB, CODON_ERR_UNSUPPORTED,
                "conv2d_fwd: %dx%d image: 128 channel planes exceed the 4 GiB buffer-descriptor range", d->height, d->width);
  p.nblk = (int)nblk;
  p.flags = d->flags;
#ifdef CODON_TIMING
  p.dbg = codon_dbg_ptr();
#endif
  p.w2 = nullptr; p.y2 = nullptr; p.y2_img = p.y2_base = 0;
  // 5x5 only: for the 3x3 kernels the 40-column tile costs a resident workgroup (3 instead of 4 per CU) and that costs
  // more than the narrower loads (A/B same box: 0.965 vs 0.98 ms)
  if (KS == 5 && conv16_xwide(d, x))
    hipLaunchKernelGGL((conv_mfma_bf16_kernel<E, KS, CIN, COUT, false, (KS == 5)>), dim3((unsigned)nblk), dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL((conv_mfma_bf16_kernel<E, KS, CIN, COUT>), dim3((unsigned)nblk), dim3(256), 0, stream, p);
  return check_launch("conv_mfma_bf16_kernel");
}

template <class E>
static int conv2d_fwd_16(const codon_conv_desc* d, const void* x, const void* w, void* y, const void* res,
                         hipStream_t stream) {
  const int key = d->ksize * 1000000 + d->cin * 1000 + d->cout;
  switch (key) {
    case 5128128: return launch_conv16<E, 5, 128, 128>(d, x, w, y, res, stream);
    case 5064064: return launch_conv16<E, 5, 64, 64>(d, x, w, y, res, stream);
    case 3064064: return launch_conv16<E, 3, 64, 64>(d, x, w, y, res, stream);
    case 3128064: return launch_conv16<E, 3, 128, 64>(d, x, w, y, res, stream);
    case 3064128: return launch_conv16<E, 3, 64, 128>(d, x, w, y, res, stream);
    case 1128064: return launch_conv1x1_16<E, 128, 64>(d, x, w, y, res, stream);
    case 1064128: return launch_conv1x1_16<E, 64, 128>(d, x, w, y, res, stream);
    default:
      set_error("conv2d_fwd: no 16-bit kernel for k=%d cin=%d cout=%d", d->ksize, d->cin, d->cout);
      return CODON_ERR_UNSUPPORTED;
  }
}

int conv2d_fwd_bf16(const codon_conv_desc* d, const void* x, const void* w, void* y, const void* res,
                    hipStream_t stream) {
  return d->dtype == CODON_F16 ? conv2d_fwd_16<EF16>(d, x, w, y, res, stream)
                               : conv2d_fwd_16<EBf16>(d, x, w, y, res, stream);
}

int pack_weight_bf16(const float* w, void* out, int cout, int cin, int ks, int mode, int dtype, hipStream_t stream) {
  const long n = (long)cout * cin * ks * ks;
  const int blocks = (int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
  if (dtype == CODON_F16)
    hipLaunchKernelGGL(pack_weight_bf16_kernel<EF16>, dim3(blocks), dim3(256), 0, stream, w, (u16*)out, cout, cin, ks,
                       mode == CODON_PACK_DGRAD ? 1 : 0);
  else
    hipLaunchKernelGGL(pack_weight_bf16_kernel<EBf16>, dim3(blocks), dim3(256), 0, stream, w, (u16*)out, cout, cin,
                       ks, mode == CODON_PACK_DGRAD ? 1 : 0);
  return check_launch("pack_weight_bf16_kernel");
}

}  // namespace codon
