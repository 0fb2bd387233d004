// OPT-IN split-precision convolution for the fp32 path ("f16x3"): fp32 activations and weights in HBM, fp32
// accumulate, but every operand is split x = hi + lo (two fp16 values, 11 + 11 significand bits) on the way into
// LDS and each product is evaluated as  Wh*Xh + Wh*Xl + Wl*Xh  with three v_mfma_f32_32x32x16_f16.
//
// Why: gfx950 has no xf32/TF32 matrix path; v_mfma_f32_32x32x2_f32 runs at 1/16 of the 16-bit MFMA rate
// (157 TF chip peak), which bounds BASELINE configs[1] at 928 ms.  Three f16 MFMAs cost 3/16 of one fp32 MFMA
// step's time for the same K, i.e. a 5.3x higher ceiling (~830 TF-equivalent), with a per-product relative error
// of ~2^-22 (the dropped Wl*Xl term and the residual of the two-term split) -- 4x the fp32 rounding unit, far
// inside the 1e-4 RMSE parity bar (measured in tests/test_gpu_f16x3.py).  NOT the default: bench.py's headline
// line is the exact-fp32 kernel; this mode is selected with model.set_conv_precision("f16x3").
//
// Range: |activation| must stay below 65504 (fp16 max).  Weights are pre-scaled by 2^10 when packed (exact; moves
// Wl out of the fp16 subnormal range) and the accumulator is scaled back by 2^-10 in the epilogue (exact).
//
// Structure = conv_mfma_bf16.hip (channel-blocked LDS images, (chunk of 16 channels, filter row) stages, next stage
// prefetched to registers before the MFMAs and converted/written after them), with two images per operand.
// Workgroup tile: 8 rows x 32 pixels x 64 couts, 4 waves, xs (hi+lo, single-buffered) + ws (hi+lo, double-buffered)
// = 67.6 KB -> two workgroups per CU; the 5x5 128->128 convs use 16 x 32 x 128, 8 waves, 126 KB (one workgroup per
// CU, 120 MFMAs per wave per barrier).
// Measured (rocprofv3 PMC, profiles/): the matrix pipes are 75 % busy in CYCLES on the wide kernel, but the chip
// holds only ~1.59 GHz under this load (GRBM_GUI_ACTIVE / 8 / time), so 1 326-1 438 TF of issued f16 MFMA is what
// 75 % buys: the kernel is power/clock-limited, not stall-limited (MI355X_MICROARCH.md "DVFS give-back").

#include <stdlib.h>

#include <type_traits>

#include "codon_common.h"

namespace codon {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int F3_WSCALE_LOG2 = 10;

struct ConvS3Params {
  const float* x;
  const uint4* w;  // packed: [cout64 block][chunk][dy][part (hi, lo)][dx][cb (2)][64 cout] x 16 B
  float* y;
  const float* res;
  int H, W;
  long x_img, y_img, r_img;
  long x_base, y_base, r_base;
  int tiles_x, tiles_y, nblk, ncob;
  int flags;
  // FUSE only: chained 1x1 (128 -> 64) from the accumulators (conv_mfma_f32.hip), itself evaluated as 3 f16 MFMAs
  const uint4* w2;  // [t2][t][g][part (hi, lo)][lane] x 16 B, W1 * 2^10 split; k order as in conv_mfma_bf16.hip
  float* y2;
  long y2_img, y2_base;
};

__device__ __forceinline__ u16 f2h_bits(float f) {
  const _Float16 h = (_Float16)f;
  return *reinterpret_cast<const u16*>(&h);
}
__device__ __forceinline__ float h2f_bits(u16 v) { return (float)*reinterpret_cast<const _Float16*>(&v); }

// COUTB couts and NW waves (NW*2 pixel rows) per workgroup:  <64, 4> = 8x32x64 tile, 68.6 KB LDS, 2 workgroups/CU;
// <128, 8> = 16x32x128 tile, 126 KB LDS, one 8-wave workgroup per CU with 120 MFMAs per wave per barrier.
template <int KS, int CIN, int COUTB, int NW, bool FUSE = false>
__global__ __launch_bounds__(NW * 64, 2) void conv_mfma_f32x3_kernel(const ConvS3Params p) {
  constexpr int NT = NW * 64;
  constexpr int PAD = KS / 2;
  constexpr int PSEG = 2, CT = COUTB / 32;
  constexpr int TW = 32, TH = NW * PSEG;
  constexpr int XR = TH + KS - 1, XQ = TW + KS - 1;
  constexpr int CK = 16, NCB = 2;
  constexpr int NCHUNK = CIN / CK;
  constexpr int XS = NCB * XR * XQ;          // 16-byte elements per x image (hi or lo)
  constexpr int WS1 = KS * NCB * COUTB;      // 16-byte elements per weight part of a stage
  constexpr int WS = 2 * WS1;                // hi + lo
  constexpr int NST = NCHUNK * KS;
  constexpr int XE = (XS + NT - 1) / NT;     // 16-byte elements per thread per chunk
  constexpr int WE = (WS + NT - 1) / NT;

  __shared__ uint4 lds[2 * XS + 2 * WS];
  uint4* const xh = lds;            // x hi image
  uint4* const xl = lds + XS;       // x lo image
  uint4* const ws0 = lds + 2 * XS;  // two weight stage buffers, each [part][dx][cb][cout]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;

  unsigned bid = xcd_remap(blockIdx.x, (unsigned)p.nblk);
  const int cob = bid % p.ncob;
  bid /= p.ncob;
  const int tx = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int ty = bid % p.tiles_y;
  const int b = bid / p.tiles_y;
  const int tx0 = tx * TW, ty0 = ty * TH;
  const int H = p.H, W = p.W;
  const long HW = (long)H * W;

  const float* __restrict__ xg = p.x + (long)b * p.x_img + p.x_base;
  const uint4* __restrict__ wg = p.w + (long)cob * NST * WS;   // packed per COUTB-cout block
  const __amdgpu_buffer_rsrc_t xrsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)xg, 0, (int)((unsigned)CIN * 4u * (unsigned)HW), 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wg, 0, (int)(NST * WS * 16), 0x00020000);

  // staging unit = one 16-byte element of the channel-blocked images = 8 channels of one pixel: element e = tid + NT k
  // = (cb, r, q) has LDS index e in both images, so a round is two conflict-free ds_write_b128 per thread (the
  // per-channel-pair version wrote 16-byte-strided words: 8-way bank conflicts, twice).  Out-of-image / padding
  // elements use an out-of-range buffer offset: the loads return 0.
  unsigned xoff[XE];
#pragma unroll
  for (int k = 0; k < XE; ++k) {
    const int e = tid + k * NT;
    const int cb = e / (XR * XQ);
    const int rem = e - cb * (XR * XQ);
    const int r = rem / XQ, q = rem - r * XQ;
    const int gy = ty0 + r - PAD, gx = tx0 + q - PAD;
    const bool ok = (e < XS) && gy >= 0 && gy < H && gx >= 0 && gx < W;
    xoff[k] = ok ? 4u * (unsigned)((8 * cb) * HW + (long)gy * W + gx) : 0xFFFFFFF0u;   // byte offset in the image slice
  }

  float xv[XE][8];  // the 8 channels of an element, raw fp32, until the split at store time
  uint4 wr[WE];

#define LOAD_X(chunk_)                                                                   \
  {                                                                                      \
    const unsigned so_ = (unsigned)(chunk_) * (unsigned)(CK * 4) * (unsigned)HW;         \
    _Pragma("unroll") for (int k = 0; k < XE; ++k)                                       \
      _Pragma("unroll") for (int j = 0; j < 8; ++j)                                      \
        xv[k][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, xoff[k], so_ + (unsigned)j * 4u * (unsigned)HW, 0)); \
  }
#define STORE_X()                                                                        \
  {                                                                                      \
    _Pragma("unroll") for (int k = 0; k < XE; ++k)                                       \
        if (XS % NT == 0 || tid + k * NT < XS) {                                         \
          unsigned hw_[4], lw_[4];                                                       \
          _Pragma("unroll") for (int w = 0; w < 4; ++w) {                                \
            const float a_ = xv[k][2 * w], b_ = xv[k][2 * w + 1];                        \
            const u16 ah_ = f2h_bits(a_), bh_ = f2h_bits(b_);                            \
            const u16 al_ = f2h_bits(a_ - h2f_bits(ah_)), bl_ = f2h_bits(b_ - h2f_bits(bh_)); \
            hw_[w] = (unsigned)ah_ | ((unsigned)bh_ << 16);                              \
            lw_[w] = (unsigned)al_ | ((unsigned)bl_ << 16);                              \
          }                                                                              \
          xh[tid + k * NT] = make_uint4(hw_[0], hw_[1], hw_[2], hw_[3]);                 \
          xl[tid + k * NT] = make_uint4(lw_[0], lw_[1], lw_[2], lw_[3]);                 \
        }                                                                                \
  }
#define LOAD_W(stage_)                                                                   \
  {                                                                                      \
    const unsigned wso_ = (unsigned)(stage_) * (unsigned)(WS * 16);                      \
    _Pragma("unroll") for (int k = 0; k < WE; ++k) {                                     \
      const unsigned vo_ = (WS % NT == 0 || tid + k * NT < WS) ? (unsigned)(tid + k * NT) * 16u : 0u; \
      const auto v_ = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, vo_, wso_, 0);         \
      wr[k] = *reinterpret_cast<const uint4*>(&v_);                                      \
    }                                                                                    \
  }
#define STORE_W(buf_)                                                                    \
  {                                                                                      \
    uint4* dst_ = ws0 + (buf_) * WS;                                                     \
    _Pragma("unroll") for (int k = 0; k < WE; ++k)                                       \
        if (WS % NT == 0 || tid + k * NT < WS) dst_[tid + k * NT] = wr[k];               \
  }

  f32x16 acc[PSEG][CT];
#pragma unroll
  for (int i = 0; i < PSEG; ++i)
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;

  LOAD_X(0);
  LOAD_W(0);
  STORE_X();
  STORE_W(0);
  __syncthreads();

#pragma unroll 1
  for (int s = 0; s < NST; ++s) {
    const int chunk = s / KS;
    const int dy = s - chunk * KS;
    const bool has_next = (s + 1 < NST);
    const bool next_chunk = has_next && (dy == KS - 1);
    if (has_next) LOAD_W(s + 1);
    if (next_chunk) LOAD_X(chunk + 1);

    const uint4* xbh = xh + (half * XR + wave * PSEG + dy) * XQ + l31;
    const uint4* xbl = xl + (half * XR + wave * PSEG + dy) * XQ + l31;
    const uint4* wbh = ws0 + (s & 1) * WS + half * COUTB + l31;
    const uint4* wbl = wbh + WS1;
#pragma unroll
    for (int dx = 0; dx < KS; ++dx) {
      f16x8 ah[CT], al[CT], bh[PSEG], bl[PSEG];
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        const uint4 v = wbh[dx * NCB * COUTB + t * 32], u = wbl[dx * NCB * COUTB + t * 32];
        ah[t] = *reinterpret_cast<const f16x8*>(&v);
        al[t] = *reinterpret_cast<const f16x8*>(&u);
      }
#pragma unroll
      for (int i = 0; i < PSEG; ++i) {
        const uint4 v = xbh[i * XQ + dx], u = xbl[i * XQ + dx];
        bh[i] = *reinterpret_cast<const f16x8*>(&v);
        bl[i] = *reinterpret_cast<const f16x8*>(&u);
      }
#pragma unroll
      for (int i = 0; i < PSEG; ++i)
#pragma unroll
        for (int t = 0; t < CT; ++t) {
          acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], bh[i], acc[i][t], 0, 0, 0);   // small terms first
          acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bl[i], acc[i][t], 0, 0, 0);
          acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bh[i], acc[i][t], 0, 0, 0);
        }
    }

    if (has_next) STORE_W((s + 1) & 1);
    if (next_chunk) {
      __syncthreads();   // x images are single-buffered: every wave is done reading this chunk
      STORE_X();
    }
    __syncthreads();
  }
#undef LOAD_X
#undef STORE_X
#undef LOAD_W
#undef STORE_W

  // epilogue: buffer stores -- wave-uniform cout-plane term in the SGPR offset, one hoisted VGPR offset per pixel
  // row, off-image pixels out of range (dropped); see conv_mfma_f32.hip.
  const int gx = tx0 + l31;
  const unsigned HW4 = 4u * (unsigned)H * (unsigned)W;
  constexpr unsigned OOB = 0xFFFFFFF0u;
  unsigned vo[PSEG];
#pragma unroll
  for (int i = 0; i < PSEG; ++i) {
    const int gy = ty0 + wave * PSEG + i;
    vo[i] = (gx < W && gy < H) ? (unsigned)(4 * half) * HW4 + 4u * (unsigned)(gy * W + gx) : OOB;
  }
  auto cplane = [&](int t, int r) { return (unsigned)(t * 32 + (r & 3) + 8 * (r >> 2)) * HW4; };
  auto relu1 = [](float v) { float o; asm("v_max_f32 %0, 0, %1" : "=v"(o) : "v"(v)); return o; };
  auto ld = [](__amdgpu_buffer_rsrc_t r, unsigned v, unsigned so) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, v, so, 0));
  };
  auto st = [](float x, __amdgpu_buffer_rsrc_t r, unsigned v, unsigned so) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), r, v, so, 0);
  };
  const bool relu = p.flags & CODON_CONV_RELU;
  constexpr float unscale = 1.f / (float)(1 << F3_WSCALE_LOG2);
  const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.res ? p.res + (long)b * p.r_img + p.r_base + (FUSE ? 0 : (long)cob * COUTB * HW) : p.x), 0,
      (int)((unsigned)(FUSE ? 64 : COUTB) * HW4), 0x00020000);

  if constexpr (FUSE) {
    // Chained 1x1: the fp32 tile (unscaled, ReLU'd) is split into fp16 hi + lo in registers -- registers 8g..8g+7
    // of a D tile are the 8 k-values of the next MFMA's B operand -- and multiplied by the split, 2^10-scaled W1
    // with the same three products as the main loop.
    static_assert(!FUSE || COUTB == 128, "chained 1x1 is 128 -> 64");
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.y ? p.y + (long)b * p.y_img + p.y_base : p.y2), 0, (int)(128u * HW4), 0x00020000);
    f16x8 bh[PSEG][CT][2], bl[PSEG][CT][2];
#pragma unroll
    for (int i = 0; i < PSEG; ++i) {
      const unsigned voy = p.y ? vo[i] : OOB;
#pragma unroll
      for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          u16 h8[8], l8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float v = acc[i][t][8 * g + j] * unscale;
            if (relu) v = relu1(v);
            st(v, yrsrc, voy, cplane(t, 8 * g + j));
            h8[j] = f2h_bits(v);
            l8[j] = f2h_bits(v - h2f_bits(h8[j]));
          }
          bh[i][t][g] = *reinterpret_cast<const f16x8*>(h8);
          bl[i][t][g] = *reinterpret_cast<const f16x8*>(l8);
        }
    }
    const __amdgpu_buffer_rsrc_t w2rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, 64 * 128 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t y2rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.y2 + (long)b * p.y2_img + p.y2_base), 0, (int)(64u * HW4), 0x00020000);
    const unsigned w2vo = (unsigned)lane * 16u;
#pragma unroll 1
    for (int t2 = 0; t2 < 2; ++t2) {
      f32x16 d[PSEG];
#pragma unroll
      for (int i = 0; i < PSEG; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) d[i][r] = 0.f;
#pragma unroll
      for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const unsigned so = (unsigned)((((t2 * CT + t) * 2 + g) * 2) * 1024);
          const auto hv = __builtin_amdgcn_raw_buffer_load_b128(w2rsrc, w2vo, so, 0);
          const auto lv = __builtin_amdgcn_raw_buffer_load_b128(w2rsrc, w2vo, so + 1024u, 0);
          const f16x8 ah = *reinterpret_cast<const f16x8*>(&hv), al = *reinterpret_cast<const f16x8*>(&lv);
#pragma unroll
          for (int i = 0; i < PSEG; ++i) {
            d[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[i][t][g], d[i], 0, 0, 0);
            d[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[i][t][g], d[i], 0, 0, 0);
            d[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[i][t][g], d[i], 0, 0, 0);
          }
        }
      if (p.res) {
#pragma unroll
        for (int i = 0; i < PSEG; ++i) {
          float rv[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) rv[r] = ld(rrsrc, vo[i], cplane(t2, r));
#pragma unroll
          for (int r = 0; r < 16; ++r) st(d[i][r] * unscale + rv[r], y2rsrc, vo[i], cplane(t2, r));
        }
      } else {
#pragma unroll
        for (int i = 0; i < PSEG; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) st(d[i][r] * unscale, y2rsrc, vo[i], cplane(t2, r));
      }
    }
    return;
  }

  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.y + (long)b * p.y_img + p.y_base + (long)cob * COUTB * HW), 0, (int)((unsigned)COUTB * HW4), 0x00020000);
  auto epi = [&](auto relu_c, auto res_c, auto acc_c) {
    constexpr bool RELU = decltype(relu_c)::value;
    constexpr int RES = decltype(res_c)::value;   // 0 none, 1 add, 2 mask
    constexpr bool ACC = decltype(acc_c)::value;
#pragma unroll
    for (int i = 0; i < PSEG; ++i) {
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        float rv[16], av[16];
        if constexpr (RES != 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) rv[r] = ld(rrsrc, vo[i], cplane(t, r));
        }
        if constexpr (ACC) {
#pragma unroll
          for (int r = 0; r < 16; ++r) av[r] = ld(yrsrc, vo[i], cplane(t, r));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[i][t][r] * unscale;
          if constexpr (RELU) v = relu1(v);
          if constexpr (RES == 1) v += rv[r];
          if constexpr (RES == 2) v = rv[r] > 0.f ? v : 0.f;
          if constexpr (ACC) v += av[r];
          st(v, yrsrc, vo[i], cplane(t, r));
        }
      }
    }
  };
  using T = std::true_type;
  using F = std::false_type;
  using R0 = std::integral_constant<int, 0>;
  using R1 = std::integral_constant<int, 1>;
  using R2 = std::integral_constant<int, 2>;
  const int res_mode = !p.res ? 0 : (p.flags & CODON_CONV_MASK_RELU) ? 2 : (p.flags & CODON_CONV_ADD_RESIDUAL) ? 1 : 0;
  const bool accum = p.flags & CODON_CONV_ACCUM_OUT;
  auto by_acc = [&](auto relu_c, auto res_c) {
    if (accum) epi(relu_c, res_c, T{});
    else epi(relu_c, res_c, F{});
  };
  auto by_res = [&](auto relu_c) {
    if (res_mode == 0) by_acc(relu_c, R0{});
    else if (res_mode == 1) by_acc(relu_c, R1{});
    else by_acc(relu_c, R2{});
  };
  if (relu) by_res(T{});
  else by_res(F{});
}

// OIHW fp32 -> [cout64 block][chunk][dy][part][dx][cb (2)][64 cout][8 ch] fp16, scaled by 2^10, split hi/lo
__global__ void pack_weight_f32x3_kernel(const float* __restrict__ w, u16* __restrict__ out, int cout, int cin, int ks,
                                         int coutb) {
  const long n = (long)cout * cin * ks * ks;  // elements per part
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long t = i;
    const int j = t % 8; t /= 8;
    const int o = t % coutb; t /= coutb;
    const int cb = t % 2; t /= 2;
    const int dx = t % ks; t /= ks;
    const int dy = t % ks; t /= ks;
    const int nchunk = cin / 16;
    const int chunk = t % nchunk; t /= nchunk;
    const int cob = (int)t;
    const int ci = chunk * 16 + cb * 8 + j, co = cob * coutb + o;
    const float v = w[(((long)co * cin + ci) * ks + dy) * ks + dx] * (float)(1 << F3_WSCALE_LOG2);
    const u16 hi = f2h_bits(v);
    const u16 lo = f2h_bits(v - h2f_bits(hi));
    // destination: stage (cob, chunk, dy) holds [part][dx][cb][o][j]
    const long stage = ((long)cob * nchunk + chunk) * ks + dy;
    const long within = (((long)dx * 2 + cb) * coutb + o) * 8 + j;
    const long part = (long)ks * 2 * coutb * 8;
    out[stage * 2 * part + within] = hi;
    out[stage * 2 * part + part + within] = lo;
  }
}

// OIHW (64,128,1,1) fp32 -> chained-1x1 A-operand image [t2][t][g][part][lane][8] fp16 (x 2^10, hi/lo split)
__global__ void pack_chain1x1_f32x3_kernel(const float* __restrict__ w, u16* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // 64 * 128 values
  if (i >= 64 * 128) return;
  const int j = i & 7, lane = (i >> 3) & 63, g = (i >> 9) & 1, t = (i >> 10) & 3, t2 = i >> 12;
  const int co2 = t2 * 32 + (lane & 31);
  const int c = t * 32 + 16 * g + (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5);
  const float v = w[co2 * 128 + c] * (float)(1 << F3_WSCALE_LOG2);
  const u16 hi = f2h_bits(v);
  const u16 lo = f2h_bits(v - h2f_bits(hi));
  const int base = (((t2 * 4 + t) * 2 + g) * 2) * 512 + lane * 8 + j;
  out[base] = hi;
  out[base + 512] = lo;
}

int pack_chain1x1_f32x3(const float* w, void* out, hipStream_t stream) {
  hipLaunchKernelGGL(pack_chain1x1_f32x3_kernel, dim3(32), dim3(256), 0, stream, w, (u16*)out);
  return check_launch("pack_chain1x1_f32x3_kernel");
}

int conv_chain1x1_fwd_f32x3(const codon_conv_desc* d, const float* x, const void* w, float* y, const void* w_chain,
                            const codon_tensor* out, const codon_tensor* res, hipStream_t stream) {
  CODON_REQUIRE(d->ksize == 5 && d->cin == 128 && d->cout == 128, CODON_ERR_UNSUPPORTED,
                "conv_chain1x1_fwd: f16x3 kernel is conv5x5 128->128 + 1x1 128->64 (got k=%d %d->%d)", d->ksize, d->cin, d->cout);
  constexpr int NW = 8;
  ConvS3Params p;
  p.x = x; p.w = (const uint4*)w; p.y = y; p.res = res ? (const float*)res->data : nullptr;
  p.H = d->height; p.W = d->width;
  const long HW = (long)d->height * d->width;
  p.x_img = d->x_ctotal * HW; p.y_img = d->y_ctotal * HW; p.r_img = res ? res->ctotal * HW : 0;
  p.x_base = d->x_coff * HW; p.y_base = d->y_coff * HW; p.r_base = res ? res->coff * HW : 0;
  p.w2 = (const uint4*)w_chain; p.y2 = (float*)out->data; p.y2_img = out->ctotal * HW; p.y2_base = out->coff * HW;
  p.tiles_x = (d->width + 31) / 32;
  p.tiles_y = (d->height + 2 * NW - 1) / (2 * NW);
  p.ncob = 1;
  const long nblk = (long)p.tiles_x * p.tiles_y * d->batch;
  CODON_REQUIRE(nblk < (1L << 31), CODON_ERR_UNSUPPORTED, "conv_chain1x1_fwd: grid too large (%ld blocks)", nblk);
  p.nblk = (int)nblk;
  p.flags = d->flags;
  hipLaunchKernelGGL((conv_mfma_f32x3_kernel<5, 128, 128, NW, true>), dim3((unsigned)nblk), dim3(NW * 64), 0, stream, p);
  return check_launch("conv_mfma_f32x3_kernel<fused 1x1>");
}

template <int KS, int CIN, int COUTB, int NW>
static int launch_s3(const codon_conv_desc* d, const float* x, const void* w, float* y, const float* res,
                     hipStream_t stream) {
  ConvS3Params p;
  p.x = x; p.w = (const uint4*)w; p.y = y; p.res = res;
  p.H = d->height; p.W = d->width;
  const long HW = (long)d->height * d->width;
  p.x_img = d->x_ctotal * HW; p.y_img = d->y_ctotal * HW; p.r_img = d->r_ctotal * HW;
  p.x_base = d->x_coff * HW; p.y_base = d->y_coff * HW; p.r_base = d->r_coff * HW;
  p.tiles_x = (d->width + 31) / 32;
  p.tiles_y = (d->height + 2 * NW - 1) / (2 * NW);
  p.ncob = d->cout / COUTB;
  const long nblk = (long)p.tiles_x * p.tiles_y * d->batch * p.ncob;
  CODON_REQUIRE(nblk < (1L << 31), CODON_ERR_UNSUPPORTED, "conv2d_fwd: grid too large (%ld blocks)", nblk);
  CODON_REQUIRE(HW * 4 * 128 < 0xFFFFFFF0L, CODON_ERR_UNSUPPORTED,
                "conv2d_fwd: %dx%d image: 128 channel planes exceed the 4 GiB buffer-descriptor range", d->height, d->width);
  p.nblk = (int)nblk;
  p.flags = d->flags;
  p.w2 = nullptr; p.y2 = nullptr; p.y2_img = p.y2_base = 0;
  hipLaunchKernelGGL((conv_mfma_f32x3_kernel<KS, CIN, COUTB, NW>), dim3((unsigned)nblk), dim3(NW * 64), 0, stream, p);
  return check_launch("conv_mfma_f32x3_kernel");
}

// couts per workgroup (also the block size of the packed weight image): 128 for the 5x5 128->128 convs
static int f32x3_coutb_for(int ks, int cin, int cout) {
  return (ks == 5 && cin == 128 && cout == 128) ? 128 : 64;   // wide tile measured 7 % faster there
}
static int f32x3_coutb(const codon_conv_desc* d) { return f32x3_coutb_for(d->ksize, d->cin, d->cout); }

bool conv_f32x3_supported(const codon_conv_desc* d) {
  return (d->ksize == 3 || d->ksize == 5) && (d->cin == 64 || d->cin == 128) && (d->cout == 64 || d->cout == 128);
}

int conv2d_fwd_f32x3(const codon_conv_desc* d, const float* x, const void* w, float* y, const float* res,
                     hipStream_t stream) {
  const int key = d->ksize * 1000 + d->cin;
  if (f32x3_coutb(d) == 128) {   // 128-cout convs (conv3/6/10): one 8-wave 16x32x128 workgroup per CU
    if (key == 5128) return launch_s3<5, 128, 128, 8>(d, x, w, y, res, stream);
  }
  switch (key) {
    case 5128: return launch_s3<5, 128, 64, 4>(d, x, w, y, res, stream);
    case 5064: return launch_s3<5, 64, 64, 4>(d, x, w, y, res, stream);
    case 3064: return launch_s3<3, 64, 64, 4>(d, x, w, y, res, stream);
    case 3128: return launch_s3<3, 128, 64, 4>(d, x, w, y, res, stream);
    default:
      set_error("conv2d_fwd: no f16x3 kernel for k=%d cin=%d cout=%d", d->ksize, d->cin, d->cout);
      return CODON_ERR_UNSUPPORTED;
  }
}

int pack_weight_f32x3(const float* w, void* out, int cout, int cin, int ks, hipStream_t stream) {
  const long n = (long)cout * cin * ks * ks;
  const int blocks = (int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
  hipLaunchKernelGGL(pack_weight_f32x3_kernel, dim3(blocks), dim3(256), 0, stream, w, (u16*)out, cout, cin, ks,
                     f32x3_coutb_for(ks, cin, cout));
  return check_launch("pack_weight_f32x3_kernel");
}

}  // namespace codon
