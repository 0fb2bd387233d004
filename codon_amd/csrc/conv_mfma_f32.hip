// fp32 implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces nn.Conv2d(stride 1, pad k//2, bias=False) [+ ReLU | + residual] of
// /root/reference/CODON_X4/CODON_x4.py:24-47 as called at :69,72,75-78,81-84,120,123-129.
//
// GEMM view per image:  Y[co][pix] = sum_{ci,dy,dx} Wt[co][(ci,dy,dx)] * X[(ci)][pix + (dy,dx)]
//   MFMA A operand (32 x 2) = weights   : lane l holds W[co = l&31][k = l>>5]
//   MFMA B operand (2 x 32) = activations: lane l holds X[k = l>>5][pixel = l&31]
//   D (32 x 32): lane holds ONE pixel column (l&31) and 16 cout rows
//                row(reg) = (reg&3) + 8*(reg>>2) + 4*(l>>5)      (cdna_hip_programming.md section 3)
// so that every accumulator register stores as two 128-byte NCHW row segments (32 consecutive
// pixels of one cout plane per half-wave): fully coalesced, no transpose.
// The two k values of one MFMA are two input CHANNELS at the same filter tap, so the B operand
// is 32 consecutive floats of one LDS tile row per half-wave: conflict-free ds_read_b32.
//
// Workgroup = 256 threads = 4 waves, output tile TH x 32 pixels x all COUT channels.
//   wave w owns PSEG pixel rows (w*PSEG .. w*PSEG+PSEG-1) x COUT/32 cout tiles.
// K loop is cut into stages (channel chunk of CK, filter row dy); per stage the workgroup needs
//   xs[chunk&1] : CK x (TH+KS-1) x (32+KS-1) input halo tile (zero padded)      -- per chunk
//   ws[stage&1] : CK x KS x COUT weights of filter row dy (contiguous in the packed image)
// both double-buffered in LDS; the next stage is prefetched global->registers BEFORE the
// current stage's MFMAs and written to LDS after them (T14 issue-early / write-late), one
// barrier per stage.  fp32 MFMA is 64 cycles per instruction per SIMD, so LDS and the
// staging traffic (<= 1 ds_read_b32 per MFMA) sit far below their limits; the kernel is bound
// by the fp32 matrix rate (157 TFLOP/s chip peak).

#include <type_traits>

#include "codon_common.h"
#include "pair.h"

namespace codon {

struct ConvParams {
  const float* x;
  const float* w;  // packed: [chunk][dy][c in CK][dx][COUT]
  float* y;
  const float* res;
  int H, W;
  long x_img, y_img, r_img;  // elements per image of the x / y / residual buffers
  long x_base, y_base, r_base;  // channel offset * H * W
  int tiles_x, tiles_y, nblk;
  int flags;
#ifdef CODON_TIMING
  long long* dbg;
#endif
  // GATE only: the conv input is formed while staging, x = pre * (ch * sp) + in  (the CAC gate-apply of the
  // producing block, CODON_x4.py:89-91,117-118): p.x = pre, same slice of `in2`, ch (B,64), sp (B,1,H,W)
  const float* in2;
  const float* ch;
  const float* sp;
  long in_img, in_base;
  // GATE, optional: the gate-applied input is ALSO written here (each tile its own pixels), so that a sibling conv on the
  // same input runs plain (codon_conv2d_gated_emit_fwd)
  float* gout;
  long go_img, go_base;
  // FUSE only: the chained 1x1 (128 -> 64) applied to the tile while it is still in the accumulators
  const float* w2;  // [t2][t][lane][16]: W1[t2*32 + (lane&31)][t*32 + (r&3) + 8*(r>>2) + 4*(lane>>5)]
  float* y2;
  long y2_img, y2_base;
  // FUSE + ST only: CAC statistics of the 64 channels this launch produces, from the epilogue (no pass over the tensor)
  float* st_pool;   // (B,2,H,W): per pixel { max, SUM } over this stream's 64 channels (ChannelPool, CAC_module.py:81)
  float* st_part;   // (B, H * tiles_x, 128, 2): per ROW STRIP of 32 pixels, per channel { sum, max } (first stage of the pools, :43,47)
  int st_choff;     // 0 = colour stream (Fcat channels 0..63), 64 = depth stream
};
static_assert(sizeof(ConvParams) <= CODON_KERNARG_LIMIT, "passed by value as a kernel argument");

template <int KS, int CIN>
struct ConvCfg {
  static constexpr int CK = (KS == 1) ? 16 : 8;
  static constexpr int NCHUNK = CIN / CK;
};

// VALU BUDGET.  Measured on gfx950 (tools/exp_timing.py): a VALU instruction of ANY wave of the SIMD
// takes 4 cycles out of the matrix pipe (2 dummy v_add per MFMA: 55.8 -> 62.5 ms on conv5x5-128), while SALU / LDS /
// VMEM issue is free.  Once the MFMA stream is dense, every address computation, select and convert is therefore paid
// for in MFMA time -- and a wave in its (VALU-only) prologue or epilogue gets one issue slot per 64-cycle MFMA of its
// co-resident waves.  So every global access below is a BUFFER instruction: a wave-uniform SGPR offset carries the
// channel / chunk / stage term, one hoisted 32-bit VGPR offset per element carries the lane term, out-of-image and
// padding elements use an out-of-range offset (the load returns 0, the store is dropped), and no 64-bit address
// arithmetic, exec-mask branch or zero-fill select is left in the per-stage or per-element code.
constexpr unsigned BUF_OOB = 0xFFFFFFF0u;     // >= num_records of every descriptor below (checked by the launcher)
constexpr int BUF_FLAGS = 0x00020000;         // raw buffer, 32-bit data format

__device__ __forceinline__ float buf_ld(__amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, vo, so, 0));
}
__device__ __forceinline__ void buf_st(float v, __amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, vo, so, 0);
}

// ReLU in ONE VALU op: fmaxf() compiles to a canonicalising v_max(v, v) plus the v_max(0, v) on MFMA results
__device__ __forceinline__ float relu1(float v) {
  float o;
  asm("v_max_f32 %0, 0, %1" : "=v"(o) : "v"(v));
  return o;
}

template <int N, class F, int I = 0>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, F, I + 1>(static_cast<F&&>(f));
  }
}

// residual modes of the epilogue (compile-time variants: one VALU op per element each)
enum { RES_NONE = 0, RES_ADD = 1, RES_MASK = 2 };

// NW = waves per workgroup: 4 (8 x 32 tile, two workgroups per CU) or 8 (16 x 32 tile, one 8-wave workgroup per CU:
// the halo is 20 x 36 / (16 x 32) = 1.41x the tile instead of 1.69x, and the weight stage is staged once per CU).
// The body takes the parameter block by reference and its (XCD-remapped) tile index as an argument: the same code serves the
// one-conv launch and the pair launch (conv_mfma_f32_pair_kernel).
// CSPLIT (round 5, small grids only): the four waves of a workgroup are 2 pixel rows x 2 HALVES OF THE COUTS instead of 4 rows
// -- a 2 x 32 tile, twice the workgroups, half the serial MFMA chain per wave.  One 128 x 128 image is 128 tiles of 4 x 32:
// 512 waves on the chip's 1024 SIMDs, each running ONE chain of 6400 fp32 MFMAs (171 us at 2.4 GHz); split, it is 1024 waves
// of 3200.  Every output is still the same fma chain over k in the same order (a wave owns whole cout tiles), so results are
// bit-identical to the unsplit kernel; the chained 1x1 needs all 128 intermediate channels of a pixel row, which now sit in
// two waves: they meet through LDS (free after the last stage) in the accumulator layout, which IS the B operand layout.
// ST (round 6, FUSE only): the epilogue also leaves the CAC statistics of its 64 output channels -- per pixel { max, sum } and
// per channel { sum, max } over ROW STRIPS of 32 pixels.  A strip's 32 values of a channel sit in the 32 lanes of a half-wave
// whatever the tile shape (8 x 32, 4 x 32, 2 x 32 cout-split), and are summed by one fixed DPP tree; the strips are folded in
// index order afterwards (cac_tail_kernel): the statistics -- hence the gates, hence the image -- do not depend on the tiling,
// i.e. on the batch an image arrives in.  One 128 x 128 image: five 22-us statistics passes per forward gone.
// LDS of one workgroup, in floats (the body's own arithmetic, restated for kernels that hold the arena themselves)
template <int KS, int CIN, int COUT, int PSEG, int NW, bool CSPLIT>
constexpr int conv_f32_lds_floats() {
  constexpr int TW = 32, TH = (CSPLIT ? NW / 2 : NW) * PSEG, XR = TH + KS - 1, XQ = TW + KS - 1;
  constexpr int CK = ConvCfg<KS, CIN>::CK, XS = CK * XR * XQ, WS = CK * KS * COUT, NT = NW * 64;
  constexpr int W4 = WS / 4, WE = (W4 + NT - 1) / NT, XSP = (XS + 3) & ~3, WSP = WE * NT * 4;
  return 2 * XSP + 2 * WSP;
}

// EXTLDS (round 6): the staging arena is the CALLER's (a kernel that runs two different bodies in one grid, mix53 below,
// declares one arena of the larger size); every other instantiation declares its own, as before.
template <int KS, int CIN, int COUT, int PSEG, bool FUSE = false, bool GATE = false, int NW = 4, bool CSPLIT = false,
          bool ST = false, bool EXTLDS = false>
__device__ __forceinline__ void conv_mfma_f32_body(const ConvParams& p, unsigned bid, float* ext_lds = nullptr) {
  static_assert(!CSPLIT || (PSEG == 1 && NW == 4 && COUT % 64 == 0), "the cout-split form is a small-grid conv");
  static_assert(!ST || FUSE, "statistics come out of the chained 1x1's epilogue");
  constexpr int NT = NW * 64;
  constexpr int PAD = KS / 2;
  constexpr int TW = 32, TH = (CSPLIT ? NW / 2 : NW) * PSEG;
  constexpr int XR = TH + KS - 1, XQ = TW + KS - 1;
  constexpr int CK = ConvCfg<KS, CIN>::CK;
  constexpr int NCHUNK = CIN / CK;
  constexpr int XS = CK * XR * XQ;    // floats per input tile
  constexpr int WS = CK * KS * COUT;  // floats per weight stage
  constexpr int CTALL = COUT / 32, CT = CSPLIT ? CTALL / 2 : CTALL;   // cout tiles of the conv / of one wave
  constexpr int NST = NCHUNK * KS;
  constexpr int W4 = WS / 4;            // float4 per weight stage
  constexpr int WE = (W4 + NT - 1) / NT;  // float4 per thread per stage
  constexpr int XSP = (XS + 3) & ~3, WSP = WE * NT * 4;   // the weight buffer is padded to whole rounds: no store predicates
  static_assert(WS % 4 == 0, "weight stage must be whole float4s");

  static_assert(2 * XSP + 2 * WSP == conv_f32_lds_floats<KS, CIN, COUT, PSEG, NW, CSPLIT>(), "conv_f32_lds_floats restates this");
  __shared__ __attribute__((aligned(16))) float lds_own[EXTLDS ? 4 : 2 * XSP + 2 * WSP];
  float* const lds = EXTLDS ? ext_lds : lds_own;
  float* const xs0 = lds;
  float* const ws0 = lds + 2 * XSP;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31;
  const int half = lane >> 5;
  const int wrow = CSPLIT ? (wave & 1) : wave;          // this wave's pixel-row segment inside the tile
  const int cbase = CSPLIT ? (wave >> 1) * CT : 0;      // ... and its first cout tile

  CODON_TSTAMP(p.dbg, 0)
  const int tx = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int ty = bid % p.tiles_y;
  const int b = bid / p.tiles_y;
  const int tx0 = tx * TW, ty0 = ty * TH;
  const int H = p.H, W = p.W;
  const unsigned HW4 = 4u * (unsigned)H * (unsigned)W;   // bytes per channel plane (launcher: 32 planes < 4 GiB)
  const long HWl = (long)H * W;
  // descriptor over planes [plane, plane + n) of a slice: the 64-bit part of every address is SALU work, done once
  // per chunk / cout tile; what is left for the VGPR + SGPR offsets stays below 32 planes
  auto planes = [&](const float* base, int plane, int n) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (long)plane * HWl), 0, (int)((unsigned)n * HW4), BUF_FLAGS);
  };

  // wave-uniform buffer descriptors: this image's input slice, the packed weights
  const float* const xbase = p.x + (long)b * p.x_img + p.x_base;
  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)(NST * WS * 4), BUF_FLAGS);

  // Position-major staging: a thread owns PJ pixel positions of the halo tile (same for every chunk) and walks the
  // chunk's CK channels over them -- the plane term is an SGPR offset, so the whole gather plan is PJ hoisted 32-bit
  // offsets (out-of-image / padding positions -> BUF_OOB) and costs ~15 VALU per position once per workgroup.
  // GATE: the per-pixel gate sp sits in PJ registers for the whole tile and the per-channel gate ch is a scalar per
  // chunk:  x = fma(pre, ch * sp, in)  exactly as cac_apply_kernel computes it.
  constexpr int NPOS = XR * XQ, PJ = (NPOS + NT - 1) / NT;
  unsigned poff[PJ];
  unsigned pown = 0;                       // GATE: bit j = position j is one of the tile's own pixels (not halo), in the image
  float spv[GATE ? PJ : 1];
  const float* const inbase = GATE ? p.in2 + (long)b * p.in_img + p.in_base : nullptr;
  const float* const chp = GATE ? p.ch + (long)b * 64 : nullptr;
  const bool emit = GATE && p.gout != nullptr;           // wave-uniform
  float* const gobase = emit ? p.gout + (long)b * p.go_img + p.go_base : nullptr;
  {
    __amdgpu_buffer_rsrc_t sprsrc;
    if constexpr (GATE)
      sprsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.sp + (long)b * HWl), 0, (int)HW4, BUF_FLAGS);
#pragma unroll
    for (int j = 0; j < PJ; ++j) {
      const int pos = tid + j * NT;
      const int r = pos / XQ, q = pos - r * XQ;
      const int gy = ty0 + r - PAD, gx = tx0 + q - PAD;
      const bool ok = pos < NPOS && gy >= 0 && gy < H && gx >= 0 && gx < W;
      poff[j] = ok ? 4u * (unsigned)(gy * W + gx) : BUF_OOB;
      if constexpr (GATE) {
        spv[j] = buf_ld(sprsrc, poff[j], 0u);
        if (ok && r >= PAD && r < PAD + TH && q >= PAD && q < PAD + TW) pown |= 1u << j;
      }
    }
  }
  // Round 3: x and weight stages go global -> LDS by LDS-DMA (buffer_load_dword / dwordx4 ... lds): no staging registers,
  // no ds_write instructions, no lgkmcnt drain in front of the barrier.  Same box, same call: conv5x5-128 + chained 1x1
  // 55.35 -> 53.67 ms (94.3 % -> 97.3 % of the fp32 MFMA peak), config-2 forward 1 005 -> 980 ms; bit-identical results.
  // Lanes past the tile's last position are exec-masked (they would land in the next channel's plane); out-of-image
  // positions carry an out-of-range offset and land as zeros.  k = 1 (0.385 -> 0.392 ms) and the gated staging (needs
  // the VALU) keep the register path.
  constexpr bool DMA = !GATE && KS != 1;
  typedef __attribute__((address_space(3))) void lds_void;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  float xg_[DMA ? 1 : CK][DMA ? 1 : PJ], xi_[GATE ? CK : 1][GATE ? PJ : 1], chs[GATE ? CK : 1];

  // weight stage: float4 element tid + 256 k of the stage; the padding round is out of range
  const unsigned wvo = (unsigned)tid * 16u;
  const unsigned wvo_last = (W4 % NT == 0 || tid + (WE - 1) * NT < W4) ? wvo : BUF_OOB;

  float4 wr[DMA ? 1 : WE];

// staging steps as macros (not lambdas): keeps xr/wr in registers (no alloca left for scratch)
#define LOAD_X(chunk_, buf_)                                                       \
  {                                                                                \
    const __amdgpu_buffer_rsrc_t xr_ = planes(xbase, (chunk_) * CK, CK);           \
    __amdgpu_buffer_rsrc_t ir_;                                                    \
    if constexpr (GATE) ir_ = planes(inbase, (chunk_) * CK, CK);                   \
    _Pragma("unroll") for (int c = 0; c < CK; ++c) {                               \
      if constexpr (GATE) chs[c] = chp[(((chunk_) * CK) & 63) + c];   /* wave-uniform: scalar load */ \
      _Pragma("unroll") for (int j = 0; j < PJ; ++j) {                             \
        if constexpr (DMA) {                                                       \
          const unsigned vo_ = poff[j];                                            \
          if (NPOS % NT == 0 || tid + j * NT < NPOS)   /* exec-masked lanes write nothing */ \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr_, (lds_void*)(lds + (buf_) * XSP + c * NPOS + j * NT + wave_u * 64), \
                                                     4, vo_, (unsigned)c * HW4, 0, 0); \
        } else {                                                                   \
          xg_[c][j] = buf_ld(xr_, poff[j], (unsigned)c * HW4);                     \
          if constexpr (GATE) xi_[c][j] = buf_ld(ir_, poff[j], (unsigned)c * HW4); \
        }                                                                          \
      }                                                                            \
    }                                                                              \
  }
#define STORE_X(chunk_, buf_)                                                      \
  if constexpr (!DMA) {                                                            \
    float* dst_ = xs0 + (buf_) * XSP + tid;                                        \
    __amdgpu_buffer_rsrc_t gor_;                                                   \
    if constexpr (GATE) { if (emit) gor_ = planes(gobase, (chunk_) * CK, CK); }    \
    _Pragma("unroll") for (int j = 0; j < PJ; ++j)                                 \
      if (NPOS % NT == 0 || tid + j * NT < NPOS) {                                 \
        _Pragma("unroll") for (int c = 0; c < CK; ++c) {                           \
          if constexpr (GATE) {                                                    \
            const float gv_ = fmaf(xg_[c][j], chs[c] * spv[j], xi_[c][j]);         \
            dst_[c * NPOS + j * NT] = gv_;                                         \
            if (emit) buf_st(gv_, gor_, ((pown >> j) & 1u) ? poff[j] : BUF_OOB, (unsigned)c * HW4); \
          } else dst_[c * NPOS + j * NT] = xg_[c][j];                              \
        }                                                                          \
      }                                                                            \
  }
#define LOAD_W(stage_, buf_)                                                       \
  {                                                                                \
    const unsigned so_ = (unsigned)(stage_) * (unsigned)(WS * 4);                  \
    _Pragma("unroll") for (int k = 0; k < WE; ++k) {                               \
      if constexpr (DMA) {                                                         \
        /* a wave whose 64 float4 slots all lie past the stage (last round) issues nothing: never read */ \
        if (W4 % NT == 0 || k < WE - 1 || k * NT + wave_u * 64 < W4)               \
          __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_void*)(lds + 2 * XSP + (buf_) * WSP + (k * NT + wave_u * 64) * 4), \
                                                   16, k == WE - 1 ? wvo_last : wvo, so_ + k * (NT * 16u), 0, 0); \
      } else {                                                                     \
        const auto v_ = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, k == WE - 1 ? wvo_last : wvo, so_ + k * (NT * 16u), 0); \
        wr[k] = *reinterpret_cast<const float4*>(&v_);                             \
      }                                                                            \
    }                                                                              \
  }
#define STORE_W(buf_)                                                              \
  if constexpr (!DMA) {                                                            \
    float4* dst_ = reinterpret_cast<float4*>(ws0 + (buf_) * WSP) + tid;            \
    _Pragma("unroll") for (int k = 0; k < WE; ++k) dst_[k * NT] = wr[k];           \
  }

  f32x16 acc[PSEG][CT];
#pragma unroll
  for (int i = 0; i < PSEG; ++i)
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;

  // prologue
  LOAD_X(0, 0);
  LOAD_W(0, 0);
  STORE_X(0, 0);
  STORE_W(0);
  CODON_TSTAMP(p.dbg, 1)
  if constexpr (DMA) __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0): this wave's DMA pieces have landed
  __syncthreads();
  CODON_TSTAMP(p.dbg, 2)

#pragma unroll 1
  for (int s = 0; s < NST; ++s) {
    const int chunk = s / KS;
    const int dy = s - chunk * KS;
    const bool has_next = (s + 1 < NST);
    const bool next_chunk = has_next && (dy == KS - 1);
    if (has_next) LOAD_W(s + 1, (s + 1) & 1);
    if (next_chunk) LOAD_X(chunk + 1, (chunk + 1) & 1);

    // volatile: keeps every operand fetch a ds_read_b32 with a 16-bit immediate offset off ONE base register; left
    // alone, hipcc pairs them into ds_read2_b32 (8-bit offsets) and pays a v_add_u32 re-base per pair -- VALU ops
    // that cost matrix-pipe time, where the extra LDS instructions are free.
    typedef const volatile __attribute__((address_space(3))) float* lds_cvp;
    const lds_cvp xb = (lds_cvp)(xs0 + (chunk & 1) * XSP + (half * XR + wrow * PSEG + dy) * XQ + l31);
    const lds_cvp wb = (lds_cvp)(ws0 + (s & 1) * WSP + half * (KS * COUT) + l31 + cbase * 32);
    // operand fetch for group g = (dx, cp) is issued one group ahead of its MFMAs (two register sets)
    constexpr int NG = KS * (CK / 2);
    float a[2][CT], bv[2][PSEG];
#define FETCH(g_)                                                                                        \
  {                                                                                                      \
    constexpr int dx_ = (g_) / (CK / 2), cp_ = (g_) % (CK / 2);                                          \
    _Pragma("unroll") for (int t = 0; t < CT; ++t) a[(g_) & 1][t] = wb[((2 * cp_) * KS + dx_) * COUT + t * 32]; \
    _Pragma("unroll") for (int i = 0; i < PSEG; ++i) bv[(g_) & 1][i] = xb[(2 * cp_) * (XR * XQ) + i * XQ + dx_]; \
  }
    FETCH(0)
    static_for<NG>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      if constexpr (g + 1 < NG) FETCH(g + 1)
      __builtin_amdgcn_sched_barrier(0);   // keep the fetch of g+1 ahead of the MFMAs of g (else it is sunk to its use)
#pragma unroll
      for (int i = 0; i < PSEG; ++i)
#pragma unroll
        for (int t = 0; t < CT; ++t)
          acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][t], bv[g & 1][i], acc[i][t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    });
#undef FETCH

    if (has_next) STORE_W((s + 1) & 1);
    if (next_chunk) STORE_X(chunk + 1, (chunk + 1) & 1);
    if constexpr (DMA) __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) before the barrier
    __syncthreads();
  }

#undef LOAD_X
#undef STORE_X
#undef LOAD_W
#undef STORE_W
  CODON_TSTAMP(p.dbg, 3)

  // epilogue.  Lane term of every output address: pixel (row of this wave's segment i, column l31) of cout plane
  // 4*half; the cout term (t, r) is wave-uniform and goes into the SGPR offset.  Off-image pixels -> BUF_OOB.
  const int gx = tx0 + l31;
  unsigned vo[PSEG];
#pragma unroll
  for (int i = 0; i < PSEG; ++i) {
    const int gy = ty0 + wrow * PSEG + i;
    vo[i] = (gx < W && gy < H) ? (unsigned)(4 * half) * HW4 + 4u * (unsigned)(gy * W + gx) : BUF_OOB;
  }
  const bool relu = p.flags & CODON_CONV_RELU;
  const float* const rbase = p.res ? p.res + (long)b * p.r_img + p.r_base : p.x;
  auto inplane = [&](int r) { return (unsigned)((r & 3) + 8 * (r >> 2)) * HW4; };   // cout plane inside a 32-cout tile

  // ST: statistics of one pixel row of one 32-channel output tile (dv: lane = pixel column l31, register r = channel
  // t2 * 32 + (r&3) + 8(r>>2) + 4 half).  Returns this lane's { sum, max } over its 16 channels (serial in r); stores, per
  // channel, { sum, max } over the row strip's 32 pixels (off-image columns: 0 / -inf) -- the reduction over the 32 lanes of
  // a half is five DPP steps whose result stands in lanes 31 and 63.
  auto half_red = [](float v, auto maxc) {
    constexpr bool MX = decltype(maxc)::value;
    auto op = [](float a_, float b_) { return MX ? fmaxf(a_, b_) : a_ + b_; };
    v = op(v, dpp_take<0xB1>(v));            // quad_perm [1,0,3,2]
    v = op(v, dpp_take<0x4E>(v));            // quad_perm [2,3,0,1]
    v = op(v, dpp_take<0x124>(v));           // row_ror:4
    v = op(v, dpp_take<0x128>(v));           // row_ror:8   -> every lane holds its 16-lane row's result
    v = op(v, dpp_take<0x142, 0xa>(v));      // row_bcast:15 into rows 1, 3 -> lanes 16..31 / 48..63 hold their half's
    return v;
  };
  auto row_stats = [&](const f32x16& dv, int t2, int gy, bool colok, float& ls, float& lm) {
    ls = 0.f; lm = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) { ls += dv[r]; lm = fmaxf(lm, dv[r]); }
    if (gy >= H) return;                     // wave-uniform: no such strip
    float2* const out = reinterpret_cast<float2*>(p.st_part) +
                        (((long)b * H + gy) * p.tiles_x + tx) * 128 + p.st_choff + t2 * 32 + 4 * half;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float cs = half_red(colok ? dv[r] : 0.f, std::false_type{});
      const float cm = half_red(colok ? dv[r] : -INFINITY, std::true_type{});
      if (l31 == 31) out[(r & 3) + 8 * (r >> 2)] = make_float2(cs, cm);
    }
  };
  (void)half_red; (void)row_stats;

  if constexpr (FUSE) {
    // Chained 1x1: the D layout of the 32x32 MFMA (lane = pixel l&31, register r = channel (r&3)+8(r>>2)+4(l>>5))
    // IS a B operand of the next MFMA for the channel pair {c, c+4}: lanes 0-31 carry k = 0, lanes 32-63 k = 1 of
    // the same 32 pixels.  So Y2[co2][pix] = sum_c W1[co2][c] relu(acc)[c][pix] runs straight from the accumulator
    // registers -- 64 MFMAs per 32 output channels and pixel row, no LDS round trip, and the 128-channel
    // intermediate never has to reach HBM (p.y == nullptr).  W1 is pre-permuted to that k order by the packer.
    static_assert(!FUSE || COUT == 128, "chained 1x1 is 128 -> 64");
    if (relu) {
#pragma unroll
      for (int i = 0; i < PSEG; ++i)
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][t][r] = relu1(acc[i][t][r]);
    }
    if (p.y) {
      const float* const ybase = p.y + (long)b * p.y_img + p.y_base;
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        const __amdgpu_buffer_rsrc_t yr_ = planes(ybase, (cbase + t) * 32, 32);
#pragma unroll
        for (int i = 0; i < PSEG; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) buf_st(acc[i][t][r], yr_, vo[i], inplane(r));
      }
    }
    if constexpr (CSPLIT) {
      // the row's 128 intermediate channels: this wave's two tiles and its partner's two, through LDS as [tile][r][lane]
      // (the staging buffers are free: every wave is past the last stage's barrier)
      static_assert(2 * CTALL * 16 * 64 <= 2 * XSP + 2 * WSP, "the exchange fits the stage buffers");
      float* const xc = lds + wrow * (CTALL * 16 * 64);
#pragma unroll
      for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) xc[((cbase + t) * 16 + r) * 64 + lane] = acc[0][t][r];
      __syncthreads();
      const int t2 = wave >> 1;                            // this wave's tile of the 64 output channels
      const __amdgpu_buffer_rsrc_t w2r_ = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, 64 * 128 * 4, BUF_FLAGS);
      const unsigned w2o_ = (unsigned)lane * 64u;
      f32x16 d;
#pragma unroll
      for (int r = 0; r < 16; ++r) d[r] = 0.f;
#pragma unroll
      for (int t = 0; t < CTALL; ++t) {                    // k in the order of the unsplit kernel: tiles 0..3, registers 0..15
        float4 a4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const auto v_ = __builtin_amdgcn_raw_buffer_load_b128(w2r_, w2o_, (unsigned)((t2 * CTALL + t) * 4096 + q * 16), 0);
          a4[q] = *reinterpret_cast<const float4*>(&v_);
        }
        const float* a = reinterpret_cast<const float*>(a4);
#pragma unroll
        for (int r = 0; r < 16; ++r)
          d = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], xc[(t * 16 + r) * 64 + lane], d, 0, 0, 0);
      }
      const float* const y2b_ = p.y2 + (long)b * p.y2_img + p.y2_base;
      const __amdgpu_buffer_rsrc_t y2r_ = planes(y2b_, t2 * 32, 32), rr_ = planes(rbase, t2 * 32, 32);
      if (p.res) {
        float rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = buf_ld(rr_, vo[0], inplane(r));
#pragma unroll
        for (int r = 0; r < 16; ++r) buf_st(d[r] + rv[r], y2r_, vo[0], inplane(r));
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) buf_st(d[r], y2r_, vo[0], inplane(r));
      }
      if constexpr (ST) {
        // this wave: row wrow, output tile t2.  The pixel's 64 channels sit in the two waves of its row: the per-tile
        // { sum, max } meet in LDS behind the exchange region, the t2 = 0 wave adds them P[0] + P[1] -- the unsplit order
        const int gy = ty0 + wrow;
        float ls, lm;
        row_stats(d, t2, gy, gx < W, ls, lm);
        ls += __shfl_xor(ls, 32, 64);
        lm = fmaxf(lm, __shfl_xor(lm, 32, 64));
        float* const pp = lds + 2 * CTALL * 16 * 64;           // [row][t2][32][2]
        static_assert(2 * CTALL * 16 * 64 + 256 <= 2 * XSP + 2 * WSP, "the pixel statistics fit behind the exchange");
        if (half == 0) { pp[((wrow * 2 + t2) * 32 + l31) * 2] = ls; pp[((wrow * 2 + t2) * 32 + l31) * 2 + 1] = lm; }
        __syncthreads();
        if (t2 == 0 && half == 0 && vo[0] != BUF_OOB) {
          const float s1 = pp[((wrow * 2 + 1) * 32 + l31) * 2], m1 = pp[((wrow * 2 + 1) * 32 + l31) * 2 + 1];
          const long q = (long)gy * W + gx;
          p.st_pool[(long)b * 2 * HWl + q] = fmaxf(lm, m1);
          p.st_pool[(long)b * 2 * HWl + HWl + q] = ls + s1;
        }
      }
      CODON_TSTAMP(p.dbg, 4)
      return;
    }
    const __amdgpu_buffer_rsrc_t w2rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, 64 * 128 * 4, BUF_FLAGS);
    const float* const y2base = p.y2 + (long)b * p.y2_img + p.y2_base;
    const unsigned w2vo = (unsigned)lane * 64u;
    float psum[ST ? PSEG : 1], pmax[ST ? PSEG : 1];
#pragma unroll 1
    for (int t2 = 0; t2 < 2; ++t2) {
      f32x16 d[PSEG];
#pragma unroll
      for (int i = 0; i < PSEG; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) d[i][r] = 0.f;
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        float4 a4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const auto v_ = __builtin_amdgcn_raw_buffer_load_b128(w2rsrc, w2vo, (unsigned)((t2 * CT + t) * 4096 + q * 16), 0);
          a4[q] = *reinterpret_cast<const float4*>(&v_);
        }
        const float* a = reinterpret_cast<const float*>(a4);
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int i = 0; i < PSEG; ++i)
            d[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], acc[i][t][r], d[i], 0, 0, 0);
      }
      const __amdgpu_buffer_rsrc_t y2rsrc = planes(y2base, t2 * 32, 32), rrsrc = planes(rbase, t2 * 32, 32);
      if (p.res) {
#pragma unroll
        for (int i = 0; i < PSEG; ++i) {
          float rv[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) rv[r] = buf_ld(rrsrc, vo[i], inplane(r));
#pragma unroll
          for (int r = 0; r < 16; ++r) buf_st(d[i][r] + rv[r], y2rsrc, vo[i], inplane(r));
        }
      } else {
#pragma unroll
        for (int i = 0; i < PSEG; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) buf_st(d[i][r], y2rsrc, vo[i], inplane(r));
      }
      if constexpr (ST) {
#pragma unroll
        for (int i = 0; i < PSEG; ++i) {
          float ls, lm;
          row_stats(d[i], t2, ty0 + wrow * PSEG + i, gx < W, ls, lm);
          ls += __shfl_xor(ls, 32, 64);                     // P[t2]: the tile's 32 channels of this pixel
          lm = fmaxf(lm, __shfl_xor(lm, 32, 64));
          psum[i] = t2 == 0 ? ls : psum[i] + ls;            // P[0] + P[1]
          pmax[i] = t2 == 0 ? lm : fmaxf(pmax[i], lm);
        }
      }
    }
    if constexpr (ST) {
#pragma unroll
      for (int i = 0; i < PSEG; ++i)
        if (half == 0 && vo[i] != BUF_OOB) {
          const long q = (long)(ty0 + wrow * PSEG + i) * W + gx;
          p.st_pool[(long)b * 2 * HWl + q] = pmax[i];
          p.st_pool[(long)b * 2 * HWl + HWl + q] = psum[i];
        }
    }
    CODON_TSTAMP(p.dbg, 4)
    return;
  }

  // ReLU / residual / accumulate as compile-time variants selected by wave-uniform branches: inside a variant every
  // element costs its store plus at most two VALU ops, and the 16 loads of a tile are issued back to back.
  const float* const ybase = p.y + (long)b * p.y_img + p.y_base;
  auto epi = [&](auto relu_c, auto res_c, auto acc_c, auto ms_c) {
    constexpr bool RELU = decltype(relu_c)::value;
    constexpr int RES = decltype(res_c)::value;
    constexpr bool ACC = decltype(acc_c)::value;
    constexpr bool MS = decltype(ms_c)::value;      // MASK_SUM: the mask applies to conv + previous value
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      const __amdgpu_buffer_rsrc_t yrsrc = planes(ybase, (cbase + t) * 32, 32), rrsrc = planes(rbase, (cbase + t) * 32, 32);
#pragma unroll
      for (int i = 0; i < PSEG; ++i) {
        float rv[16], av[16];
        if constexpr (RES != RES_NONE) {
#pragma unroll
          for (int r = 0; r < 16; ++r) rv[r] = buf_ld(rrsrc, vo[i], inplane(r));
        }
        if constexpr (ACC) {
#pragma unroll
          for (int r = 0; r < 16; ++r) av[r] = buf_ld(yrsrc, vo[i], inplane(r));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[i][t][r];
          if constexpr (RELU) v = relu1(v);
          if constexpr (RES == RES_ADD) v += rv[r];
          if constexpr (RES == RES_MASK && !MS) v = rv[r] > 0.f ? v : 0.f;
          if constexpr (ACC) v += av[r];
          if constexpr (RES == RES_MASK && MS) v = rv[r] > 0.f ? v : 0.f;
          buf_st(v, yrsrc, vo[i], inplane(r));
        }
      }
    }
  };
  using T = std::true_type;
  using F = std::false_type;
  using R0 = std::integral_constant<int, RES_NONE>;
  using R1 = std::integral_constant<int, RES_ADD>;
  using R2 = std::integral_constant<int, RES_MASK>;
  const int res_mode = !p.res ? RES_NONE : (p.flags & CODON_CONV_MASK_RELU) ? RES_MASK
                                         : (p.flags & CODON_CONV_ADD_RESIDUAL) ? RES_ADD : RES_NONE;
  const bool accum = p.flags & CODON_CONV_ACCUM_OUT;
  const bool msum = p.flags & CODON_CONV_MASK_SUM;
  auto by_acc = [&](auto relu_c, auto res_c) {
    if constexpr (!decltype(relu_c)::value && decltype(res_c)::value == RES_MASK) {
      if (accum && msum) { epi(relu_c, res_c, T{}, T{}); return; }
    }
    if (accum) epi(relu_c, res_c, T{}, F{});
    else epi(relu_c, res_c, F{}, F{});
  };
  auto by_res = [&](auto relu_c) {
    if (res_mode == RES_NONE) by_acc(relu_c, R0{});
    else if (res_mode == RES_ADD) by_acc(relu_c, R1{});
    else by_acc(relu_c, R2{});
  };
  if (relu) by_res(T{});
  else by_res(F{});
  CODON_TSTAMP(p.dbg, 4)
}

template <int KS, int CIN, int COUT, int PSEG, bool FUSE = false, bool GATE = false, int NW = 4, bool CSPLIT = false,
          bool ST = false>
__global__ __launch_bounds__(NW * 64, 2) void conv_mfma_f32_kernel(const ConvParams p) {
  conv_mfma_f32_body<KS, CIN, COUT, PSEG, FUSE, GATE, NW, CSPLIT, ST>(p, xcd_remap(blockIdx.x, (unsigned)p.nblk));
}

// Two convs of one shape and kernel variant as ONE grid of 2 * nblk workgroups (codon_conv_pair_begin / _end): the depth and
// the colour stream of a block (/root/reference/CODON_X4/CODON_x4.py:75-84) at one image per call -- BASELINE configs[0]: two
// launches of 128 workgroups on two HIP streams cost 12 + 18 us of fork / join events per block
// (profiles/r05_b1_fp32_128x128_timeline.txt); one grid of 256 workgroups has no seam.  Same code per tile: same bits.
struct ConvPair {
  ConvParams a, b;
};
static_assert(sizeof(ConvPair) <= CODON_KERNARG_LIMIT, "two parameter blocks passed by value as one kernel argument");
template <int KS, int CIN, int COUT, int PSEG, bool FUSE = false, bool GATE = false, int NW = 4, bool CSPLIT = false,
          bool ST = false>
__global__ __launch_bounds__(NW * 64, 2) void conv_mfma_f32_pair_kernel(const ConvPair pp) {
  const unsigned nblk = (unsigned)pp.a.nblk;                       // == pp.b.nblk (checked on the host)
  const unsigned v = xcd_remap(blockIdx.x, 2u * nblk);
  const bool second = v >= nblk;                                   // workgroup-uniform
  conv_mfma_f32_body<KS, CIN, COUT, PSEG, FUSE, GATE, NW, CSPLIT, ST>(second ? pp.b : pp.a, second ? v - nblk : v);
}

// mix53 (round 6): conv8 (5x5 64->64) and conv9 (3x3 64->64) of the fusion trunk read the same tensor and are independent
// (/root/reference/CODON_X4/CODON_x4.py:123-124); at one image per call each is a launch of a few rounds with its own ramp and
// drain (35 + 17 us at 1 x 128 x 128).  Held in one pair bracket they leave as ONE grid: workgroups [0, nA) run the 5x5 body on
// its tiles, [nA, nA + nB) the 3x3 body on its own -- the small-grid cout-split forms of both, one LDS arena of the larger size.
// No XCD remap: consecutive workgroups go to consecutive XCDs, so every XCD takes every 8th workgroup of BOTH kinds (a
// contiguous split would give five XCDs the expensive tiles and three the cheap ones), the 5x5 tiles are dispatched first
// and the 3x3 ones fill the tail.  Same body code per tile: same bits as the separate launches.
__global__ __launch_bounds__(256, 2) void conv_mfma_f32_mix53_kernel(const ConvPair pp) {
  constexpr int LA = conv_f32_lds_floats<5, 64, 64, 1, 4, true>(), LB = conv_f32_lds_floats<3, 64, 64, 1, 4, true>();
  __shared__ __attribute__((aligned(16))) float arena[LA > LB ? LA : LB];
  const unsigned nA = (unsigned)pp.a.nblk;
  if (blockIdx.x < nA)                                             // workgroup-uniform
    conv_mfma_f32_body<5, 64, 64, 1, false, false, 4, true, false, true>(pp.a, blockIdx.x, arena);
  else
    conv_mfma_f32_body<3, 64, 64, 1, false, false, 4, true, false, true>(pp.b, blockIdx.x - nA, arena);
}

// OIHW fp32 -> packed [chunk][dy][c][dx][cout]; DGRAD mode packs w'[ci][co][KS-1-dy][KS-1-dx].
__global__ void pack_weight_f32_kernel(const float* __restrict__ w, float* __restrict__ out, int cout,
                                       int cin, int ks, int ck, int dgrad) {
  // packed conv has KIN input channels and KOUT output channels
  const int kin = dgrad ? cout : cin, kout = dgrad ? cin : cout;
  const long n = (long)kin * kout * ks * ks;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long t = i;
    const int o = t % kout; t /= kout;
    const int dx = t % ks; t /= ks;
    const int c = t % ck; t /= ck;
    const int dy = t % ks; t /= ks;
    const int chunk = (int)t;
    const int ci = chunk * ck + c;
    float v;
    if (!dgrad) v = w[(((long)o * cin + ci) * ks + dy) * ks + dx];
    else v = w[(((long)ci * cin + o) * ks + (ks - 1 - dy)) * ks + (ks - 1 - dx)];
    out[i] = v;
  }
}

// OIHW (64,128,1,1) fp32 -> the chained-1x1 A-operand image [t2][t][lane][r]
__global__ void pack_chain1x1_f32_kernel(const float* __restrict__ w, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // 64 * 128 values
  if (i >= 64 * 128) return;
  const int r = i & 15, lane = (i >> 4) & 63, t = (i >> 10) & 3, t2 = i >> 12;
  const int co2 = t2 * 32 + (lane & 31);
  const int c = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
  out[i] = w[co2 * 128 + c];
}

int pack_chain1x1_f32(const float* w, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(pack_chain1x1_f32_kernel, dim3(32), dim3(256), 0, stream, w, out);
  return check_launch("pack_chain1x1_f32_kernel");
}

// SMALL-GRID MODE.  A launch whose 8 x 32 tiling gives fewer workgroups than ~1.5 per CU (single images: the reference
// script's own use, test.py:125, and BASELINE configs[0]) is latency-bound on one workgroup's K loop; it takes 4 x 32
// tiles instead (PSEG = 1: twice the workgroups, half the MFMAs each).  Per-pixel arithmetic and its order are the
// same, so results stay bit-identical to the large-grid kernels (batch-independence tests compare the two).
constexpr long SMALL_GRID = 384;
constexpr long CSPLIT_MAX_BLOCKS = 192;     // 4 x 32 tiles: up to here a lone launch leaves a quarter of the SIMDs without a wave
constexpr long CSPLIT_PAIR_MAX_BLOCKS = 128;   // ... and a PAIR of launches split this way is at most 512 workgroups: one round
static bool small_grid(const codon_conv_desc* d) {
  return (long)((d->width + 31) / 32) * ((d->height + 7) / 8) * d->batch < SMALL_GRID;
}

// GRIDS OF A FEW ROUNDS (one image of a few hundred rows).  Two 4-wave workgroups share a CU, and a CU gives a lone workgroup
// almost the throughput it gives two: 8 x 32 tiles of the chained conv take 0.72 ms as a pair and 0.36 ms alone.  So a launch
// runs in rounds of 512 workgroups, and its last round costs a full pair time wherever the dispatcher happens to put two
// of the leftover workgroups on one CU: 705 tiles (1 x 370 x 463) ran in 1.10 ms or in 1.43 ms depending on nothing the
// host controls (profiles/r05_grid_rounds.txt: same kernel, same operands, two builds).  Priced that way -- the last round
// at its full price -- the three ways to tile a launch are
//   8 x 32 tiles, two per CU:   ceil(n8 / 512) rounds of 3.945   (units: half a 4 x 32 pair round, 0.1835 ms on the chain)
//   4 x 32 tiles, two per CU:   ceil(n4 / 512) rounds of 2
//   4 x 32 tiles, one per CU:   ceil(n4 / 256) rounds of SOLO    (the LDS padding of solo_lds_pad; no pairing to be lucky with)
// and the cheapest is taken.  8 x 32 wins by 1.4 % once the rounds are many (the headline batch: 75 rounds); the ratios
// were measured on 15 image heights between 200 and 820 rows (tools/probes/grid_mode_sweep.sh).  Per-pixel arithmetic and
// its order do not depend on the tiling: same bits.  3 x 3 and 1 x 1 convs (memory-bound, no such step pattern) keep 8 x 32.
enum GridMode { GRID_8X32, GRID_4X32, GRID_4X32_SOLO };
constexpr float GRID_SOLO_CHAIN = 1.093f, GRID_SOLO_CONV64 = 1.184f;
static GridMode grid_mode(const codon_conv_desc* d, float solo) {
  if (small_grid(d)) return GRID_4X32_SOLO;
  const long tx = (d->width + 31) / 32;
  const long n8 = tx * ((d->height + 7) / 8) * d->batch, n4 = tx * ((d->height + 3) / 4) * d->batch;
  const float c8 = 3.945f * (float)((n8 + 511) / 512), c4 = 2.f * (float)((n4 + 511) / 512);
  const float cs = solo * (float)((n4 + 255) / 256);
  if (c8 <= c4 && c8 <= cs) return GRID_8X32;
  return c4 <= cs ? GRID_4X32 : GRID_4X32_SOLO;
}

// Small grids leave CUs empty, and the host runs the two streams of a block on two HIP streams (model.py) -- but the
// dispatcher puts the workgroups of two concurrent 128-workgroup launches on the SAME CUs (rocprofv3 trace: both overlap in
// time and each takes 345 us instead of 202).  A dynamic-LDS request that brings a workgroup above half of the CU's 160 KB
// makes every workgroup the only one on its CU, so a concurrent launch must take the free CUs.  Bytes of padding for `kernel`.
template <auto kernel>            // a template VALUE: one table per kernel instantiation (a type parameter would share one between all
static unsigned solo_lds_pad() {  // kernels of the same signature, i.e. give every kernel the padding of the first one that ran)
  // hipFuncSetAttribute acts on the CURRENT device's function object: one slot per (kernel instantiation, device), so
  // every GPU of a single-process multi-GPU caller (nn.DataParallel, per-device threads) raises its own limit before its
  // first padded launch.  A racing first call on a device computes and stores the same value.
  constexpr int MAXDEV = 64;
  static int pads[MAXDEV];
  static bool init = [] { for (int i = 0; i < MAXDEV; ++i) pads[i] = -1; return true; }();
  (void)init;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) {
    (void)hipGetLastError();
    return 0;                               // unknown device: launch unpadded (speed only)
  }
  int pad = __atomic_load_n(&pads[dev], __ATOMIC_ACQUIRE);
  if (pad < 0) {
    int want = 0;
    hipFuncAttributes a;
    if (hipFuncGetAttributes(&a, (const void*)kernel) == hipSuccess) {
      want = 82 * 1024 - (int)a.sharedSizeBytes;
      if (want < 0) want = 0;
      if (want > 0 && hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, want) != hipSuccess)
        want = 0;
    }
    (void)hipGetLastError();
    pad = want;
    __atomic_store_n(&pads[dev], pad, __ATOMIC_RELEASE);
  }
  return (unsigned)pad;
}

// one launch of a filled parameter block / two blocks of the same variant as one grid (pair.h).  `small` launches ask for the
// solo-LDS padding so that every workgroup has a CU to itself.
template <int KS, int CIN, int COUT, int PSEG, bool FUSE, bool GATE, int NW, bool SOLO, bool CSPLIT = false, bool ST = false>
static int launch_single_f32(const void* pv, hipStream_t stream) {
  const ConvParams& p = *static_cast<const ConvParams*>(pv);
  constexpr auto kern = conv_mfma_f32_kernel<KS, CIN, COUT, PSEG, FUSE, GATE, NW, CSPLIT, ST>;
  const unsigned dyn = SOLO ? solo_lds_pad<kern>() : 0u;
  hipLaunchKernelGGL(kern, dim3((unsigned)p.nblk), dim3(NW * 64), dyn, stream, p);
  return check_launch(CSPLIT ? "conv_mfma_f32_kernel<cout split>" : "conv_mfma_f32_kernel");
}
// CSPLIT pairs (2 x 32 tiles, couts split over the waves): up to 256 workgroups in all take a CU each; 257 .. 512 run two to
// a CU with no padding -- each SIMD then holds two waves of half the serial MFMA chain, which hide each other's stage waits
// (BASELINE configs[0]: the pair of chained convs 225 -> 218 us, of 5x5 64->64 convs 62 -> 57, of 3x3 26 -> 24).
template <int KS, int CIN, int COUT, int PSEG, bool FUSE, bool GATE, int NW, bool SOLO, bool CSPLIT = false, bool ST = false>
static int launch_pair_f32(const void* av, const void* bv, hipStream_t stream) {
  ConvPair pp;
  pp.a = *static_cast<const ConvParams*>(av);
  pp.b = *static_cast<const ConvParams*>(bv);
  constexpr auto kern = conv_mfma_f32_pair_kernel<KS, CIN, COUT, PSEG, FUSE, GATE, NW, CSPLIT, ST>;
  const bool solo = SOLO && (!CSPLIT || 2 * pp.a.nblk <= 256);
  const unsigned dyn = solo ? solo_lds_pad<kern>() : 0u;
  hipLaunchKernelGGL(kern, dim3(2u * (unsigned)pp.a.nblk), dim3(NW * 64), dyn, stream, pp);
  return check_launch("conv_mfma_f32_pair_kernel");
}
// a cout-split launch: held back by an open pair bracket, else alone with a CU per workgroup
static int launch_mix53_f32(const void* five, const void* three, hipStream_t stream) {
  ConvPair pp;
  pp.a = *static_cast<const ConvParams*>(five);
  pp.b = *static_cast<const ConvParams*>(three);
  hipLaunchKernelGGL(conv_mfma_f32_mix53_kernel, dim3((unsigned)pp.a.nblk + (unsigned)pp.b.nblk), dim3(256), 0, stream, pp);
  return check_launch("conv_mfma_f32_mix53_kernel");
}
template <int KS, int CIN, int COUT, bool FUSE, int NW, bool ST = false, bool GATE = false>
static int launch_or_hold_csplit_f32(const ConvParams& p, hipStream_t stream) {
  // the plain 64 -> 64 convs can leave as a mix53 grid with their sibling of the other filter size
  constexpr bool MIXABLE = !FUSE && !GATE && !ST && CIN == 64 && COUT == 64 && NW == 4 && (KS == 5 || KS == 3);
  constexpr int kind = !MIXABLE ? MIX_NONE : KS == 5 ? MIX_F32_CSPLIT_5 : MIX_F32_CSPLIT_3;
  if (const int held = pair_hold(p, &launch_single_f32<KS, CIN, COUT, 1, FUSE, GATE, NW, true, true, ST>,
                                 &launch_pair_f32<KS, CIN, COUT, 1, FUSE, GATE, NW, true, true, ST>, stream, kind,
                                 (MIXABLE && KS == 5) ? &launch_mix53_f32 : nullptr))
    return held < 0 ? held : CODON_OK;
  return launch_single_f32<KS, CIN, COUT, 1, FUSE, GATE, NW, true, true, ST>(&p, stream);
}
// small-grid launches (PSEG = 1) can be held back by an open pair bracket; everything else launches at once
template <int KS, int CIN, int COUT, int PSEG, bool FUSE, bool GATE, int NW, bool ST = false>
static int launch_or_hold_f32(const ConvParams& p, bool small, hipStream_t stream) {
  if constexpr (PSEG == 1) {
    if (small) {
      if (const int held = pair_hold(p, &launch_single_f32<KS, CIN, COUT, PSEG, FUSE, GATE, NW, true, false, ST>,
                                     &launch_pair_f32<KS, CIN, COUT, PSEG, FUSE, GATE, NW, true, false, ST>, stream))
        return held < 0 ? held : CODON_OK;
      return launch_single_f32<KS, CIN, COUT, PSEG, FUSE, GATE, NW, true, false, ST>(&p, stream);
    }
  }
  return launch_single_f32<KS, CIN, COUT, PSEG, FUSE, GATE, NW, false, false, ST>(&p, stream);
}

template <int KS, int CIN, int COUT, int PSEG, bool CSPLIT = false>
static int launch_conv_p(const codon_conv_desc* d, const float* x, const float* w, float* y,
                         const float* res, bool solo, hipStream_t stream) {
  constexpr int TH = (CSPLIT ? 2 : 4) * PSEG;
  ConvParams p;
  p.x = x; p.w = w; p.y = y; p.res = res;
  p.H = d->height; p.W = d->width;
  const long HW = (long)d->height * d->width;
  p.x_img = d->x_ctotal * HW; p.y_img = d->y_ctotal * HW; p.r_img = d->r_ctotal * HW;
  p.x_base = d->x_coff * HW; p.y_base = d->y_coff * HW; p.r_base = d->r_coff * HW;
  p.tiles_x = (d->width + 31) / 32;
  p.tiles_y = (d->height + TH - 1) / TH;
  const long nblk = (long)p.tiles_x * p.tiles_y * d->batch;
  CODON_REQUIRE(nblk < (1L << 31), CODON_ERR_UNSUPPORTED, "conv2d_fwd: grid too large (%ld blocks)", nblk);
  CODON_REQUIRE(HW * 4 * 32 < (long)BUF_OOB, CODON_ERR_UNSUPPORTED,
                "conv2d_fwd: %dx%d image: 32 channel planes exceed the 4 GiB buffer-descriptor range", d->height, d->width);
  p.nblk = (int)nblk;
  p.flags = d->flags;
#ifdef CODON_TIMING
  p.dbg = codon_dbg_ptr();
#endif
  p.w2 = nullptr; p.y2 = nullptr; p.y2_img = p.y2_base = 0;
  p.in2 = nullptr; p.ch = nullptr; p.sp = nullptr; p.in_img = p.in_base = 0;
  p.gout = nullptr; p.go_img = p.go_base = 0;
  p.st_pool = nullptr; p.st_part = nullptr; p.st_choff = 0;
  if constexpr (CSPLIT)            // a small launch (see conv_chain1x1_fwd_f32): 2 x 32 tiles, couts split over the waves
    return launch_or_hold_csplit_f32<KS, CIN, COUT, false, 4>(p, stream);
  return launch_or_hold_f32<KS, CIN, COUT, PSEG, false, false, 4>(p, PSEG == 1 && solo, stream);
}

template <int KS, int CIN, int COUT, int PSEG>
static int launch_conv(const codon_conv_desc* d, const float* x, const float* w, float* y,
                       const float* res, hipStream_t stream) {
  const GridMode mode = KS == 5 ? grid_mode(d, GRID_SOLO_CONV64) : small_grid(d) ? GRID_4X32_SOLO : GRID_8X32;
  if (mode == GRID_4X32_SOLO) {
    if constexpr (COUT % 64 == 0 && KS != 1) {
      const long nblk4 = (long)((d->width + 31) / 32) * ((d->height + 3) / 4) * d->batch;
      if (nblk4 <= (pair_recorder() ? CSPLIT_PAIR_MAX_BLOCKS : CSPLIT_MAX_BLOCKS))
        return launch_conv_p<KS, CIN, COUT, 1, true>(d, x, w, y, res, true, stream);
    }
    return launch_conv_p<KS, CIN, COUT, 1>(d, x, w, y, res, true, stream);
  }
  if (mode == GRID_4X32) return launch_conv_p<KS, CIN, COUT, 1>(d, x, w, y, res, false, stream);
  return launch_conv_p<KS, CIN, COUT, PSEG>(d, x, w, y, res, false, stream);
}

int conv_ck(int ks) { return ks == 1 ? 16 : 8; }

// codon_conv_tiling_f32 (include/codon_hip.h): the tiling the launchers above give a plain (chained = 0) / chained fp32 conv
// of this shape outside / inside a pair bracket -- the same rule, restated without a launch (host code only)
int conv_tiling_f32(const codon_conv_desc* d, int chained, int in_pair) {
  const bool k5 = d->ksize == 5;
  const GridMode mode = chained ? grid_mode(d, GRID_SOLO_CHAIN)
                        : k5    ? grid_mode(d, GRID_SOLO_CONV64) : small_grid(d) ? GRID_4X32_SOLO : GRID_8X32;
  if (mode == GRID_8X32) return CODON_TILING_8X32;
  if (mode == GRID_4X32) return CODON_TILING_4X32;
  const bool splittable = chained || (d->cout % 64 == 0 && d->ksize != 1);
  const long nblk4 = (long)((d->width + 31) / 32) * ((d->height + 3) / 4) * d->batch;
  if (splittable && nblk4 <= (in_pair ? CSPLIT_PAIR_MAX_BLOCKS : CSPLIT_MAX_BLOCKS)) return CODON_TILING_2X32_COUT_SPLIT;
  return CODON_TILING_4X32_SOLO;
}

template <int KS, int CIN, int COUT, int PSEG, bool CSPLIT = false>
static int launch_gated_p(const codon_conv_desc* d, const float* pre, const codon_tensor* in2, const float* ch,
                          const float* sp, const float* w, float* y, const codon_tensor* gated_out, bool solo,
                          hipStream_t stream) {
  constexpr int TH = (CSPLIT ? 2 : 4) * PSEG;
  ConvParams p;
  p.x = pre; p.w = w; p.y = y; p.res = nullptr;
  p.H = d->height; p.W = d->width;
  const long HW = (long)d->height * d->width;
  p.x_img = d->x_ctotal * HW; p.y_img = d->y_ctotal * HW; p.r_img = 0;
  p.x_base = d->x_coff * HW; p.y_base = d->y_coff * HW; p.r_base = 0;
  p.in2 = (const float*)in2->data; p.in_img = in2->ctotal * HW; p.in_base = in2->coff * HW;
  p.ch = ch; p.sp = sp;
  p.gout = gated_out ? (float*)gated_out->data : nullptr;
  p.go_img = gated_out ? gated_out->ctotal * HW : 0; p.go_base = gated_out ? gated_out->coff * HW : 0;
  p.w2 = nullptr; p.y2 = nullptr; p.y2_img = p.y2_base = 0;
  p.st_pool = nullptr; p.st_part = nullptr; p.st_choff = 0;
  p.tiles_x = (d->width + 31) / 32;
  p.tiles_y = (d->height + TH - 1) / TH;
  const long nblk = (long)p.tiles_x * p.tiles_y * d->batch;
  CODON_REQUIRE(nblk < (1L << 31), CODON_ERR_UNSUPPORTED, "conv2d_gated_fwd: grid too large (%ld blocks)", nblk);
  CODON_REQUIRE(HW * 4 * 32 < (long)BUF_OOB, CODON_ERR_UNSUPPORTED,
                "conv2d_gated_fwd: %dx%d image: 32 channel planes exceed the 4 GiB buffer-descriptor range", d->height, d->width);
  p.nblk = (int)nblk;
  p.flags = d->flags;
#ifdef CODON_TIMING
  p.dbg = codon_dbg_ptr();
#endif
  if constexpr (CSPLIT) return launch_or_hold_csplit_f32<KS, CIN, COUT, false, 4, false, true>(p, stream);
  return launch_or_hold_f32<KS, CIN, COUT, PSEG, false, true, 4>(p, PSEG == 1 && solo, stream);
}

template <int KS, int CIN, int COUT>
static int launch_gated(const codon_conv_desc* d, const float* pre, const codon_tensor* in2, const float* ch,
                        const float* sp, const float* w, float* y, const codon_tensor* gated_out, hipStream_t stream) {
  const GridMode mode = KS == 5 ? grid_mode(d, GRID_SOLO_CONV64) : small_grid(d) ? GRID_4X32_SOLO : GRID_8X32;
  if (mode == GRID_4X32_SOLO) {
    // round 6: the gated convs of a small launch take the cout split too (one 128 x 128 image: conv7 alone is 128 tiles whose
    // waves each run 1 152 MFMAs back to back; the gated 5x5 pairs 1 600) -- the staging arithmetic is the workgroup's, the split
    // only divides the MFMA chain; same fma chain per output: same bits
    const long nblk4 = (long)((d->width + 31) / 32) * ((d->height + 3) / 4) * d->batch;
    if (nblk4 <= (pair_recorder() ? CSPLIT_PAIR_MAX_BLOCKS : CSPLIT_MAX_BLOCKS))
      return launch_gated_p<KS, CIN, COUT, 1, true>(d, pre, in2, ch, sp, w, y, gated_out, true, stream);
  }
  if (mode != GRID_8X32) return launch_gated_p<KS, CIN, COUT, 1>(d, pre, in2, ch, sp, w, y, gated_out, mode == GRID_4X32_SOLO, stream);
  return launch_gated_p<KS, CIN, COUT, 2>(d, pre, in2, ch, sp, w, y, gated_out, false, stream);
}

int conv2d_gated_fwd_f32(const codon_conv_desc* d, const float* pre, const codon_tensor* in2, const float* ch,
                         const float* sp, const float* w, float* y, const codon_tensor* gated_out, hipStream_t stream) {
  const int key = d->ksize * 1000000 + d->cin * 1000 + d->cout;
  switch (key) {
    case 5064064: return launch_gated<5, 64, 64>(d, pre, in2, ch, sp, w, y, gated_out, stream);
    case 3064064: return launch_gated<3, 64, 64>(d, pre, in2, ch, sp, w, y, gated_out, stream);
    case 3128064: return launch_gated<3, 128, 64>(d, pre, in2, ch, sp, w, y, gated_out, stream);
    default:
      set_error("conv2d_gated_fwd: no f32 kernel for k=%d cin=%d cout=%d", d->ksize, d->cin, d->cout);
      return CODON_ERR_UNSUPPORTED;
  }
}

// d: the 5x5 128 -> 128 conv (y nullable); out / res: 64-channel slices of the chained 1x1; st_pool / st_part non-null: the
// epilogue also leaves the CAC statistics of the 64 output channels (ST kernels; any tiling gives the same bits)
int conv_chain1x1_fwd_f32(const codon_conv_desc* d, const float* x, const float* w, float* y, const float* w_chain,
                          const codon_tensor* out, const codon_tensor* res, float* st_pool, float* st_part, int st_choff,
                          hipStream_t stream) {
  CODON_REQUIRE(d->ksize == 5 && d->cin == 128 && d->cout == 128, CODON_ERR_UNSUPPORTED,
                "conv_chain1x1_fwd: f32 kernel is conv5x5 128->128 + 1x1 128->64 (got k=%d %d->%d)", d->ksize, d->cin, d->cout);
  constexpr int NWC = 4;     // waves per workgroup (one 8-wave workgroup per CU on a 16x32 tile measured slower: DESIGN.md 3.1)
  const GridMode mode = grid_mode(d, GRID_SOLO_CHAIN);
  const bool small = mode == GRID_4X32_SOLO;
  const int TH = (mode == GRID_8X32 ? 2 : 1) * NWC;
  ConvParams p;
  p.x = x; p.w = w; p.y = y; p.res = res ? (const float*)res->data : nullptr;
  p.H = d->height; p.W = d->width;
  const long HW = (long)d->height * d->width;
  p.x_img = d->x_ctotal * HW; p.y_img = d->y_ctotal * HW; p.r_img = res ? res->ctotal * HW : 0;
  p.x_base = d->x_coff * HW; p.y_base = d->y_coff * HW; p.r_base = res ? res->coff * HW : 0;
  p.in2 = nullptr; p.ch = nullptr; p.sp = nullptr; p.in_img = p.in_base = 0;
  p.gout = nullptr; p.go_img = p.go_base = 0;
  p.w2 = w_chain; p.y2 = (float*)out->data; p.y2_img = out->ctotal * HW; p.y2_base = out->coff * HW;
  p.st_pool = st_pool; p.st_part = st_part; p.st_choff = st_choff;
  const bool st = st_part != nullptr;
  CODON_REQUIRE(!st || (st_pool && !res), CODON_ERR_BAD_ARG, "conv_chain1x1_stats_fwd: statistics need both outputs and no residual");
  p.tiles_x = (d->width + 31) / 32;
  p.tiles_y = (d->height + TH - 1) / TH;
  const long nblk = (long)p.tiles_x * p.tiles_y * d->batch;
  CODON_REQUIRE(nblk < (1L << 31), CODON_ERR_UNSUPPORTED, "conv_chain1x1_fwd: grid too large (%ld blocks)", nblk);
  CODON_REQUIRE(HW * 4 * 32 < (long)BUF_OOB, CODON_ERR_UNSUPPORTED,
                "conv_chain1x1_fwd: %dx%d image: 32 channel planes exceed the 4 GiB buffer-descriptor range", d->height, d->width);
  p.nblk = (int)nblk;
  p.flags = d->flags;
#ifdef CODON_TIMING
  p.dbg = codon_dbg_ptr();
#endif
  if (small && nblk <= (pair_recorder() ? CSPLIT_PAIR_MAX_BLOCKS : CSPLIT_MAX_BLOCKS)) {
    // at most 192 tiles of 4 x 32 alone on the chip (the trunk's conv10 + confuse_fuse at one image per call), or at most
    // 128 in each launch of a pair (the two streams of a block): 2 x 32 tiles, couts split over the waves
    p.tiles_y = (d->height + 1) / 2;
    p.nblk = (int)((long)p.tiles_x * p.tiles_y * d->batch);
    if (st) return launch_or_hold_csplit_f32<5, 128, 128, true, NWC, true>(p, stream);
    return launch_or_hold_csplit_f32<5, 128, 128, true, NWC>(p, stream);
  }
  if (st) {
    if (mode != GRID_8X32) return launch_or_hold_f32<5, 128, 128, 1, true, false, NWC, true>(p, small, stream);
    return launch_or_hold_f32<5, 128, 128, 2, true, false, NWC, true>(p, false, stream);
  }
  if (mode != GRID_8X32) return launch_or_hold_f32<5, 128, 128, 1, true, false, NWC>(p, small, stream);
  return launch_or_hold_f32<5, 128, 128, 2, true, false, NWC>(p, false, stream);
}

// pixel rows per wave: 2 = 8 x 32 tile (4 operand fetches per 4 MFMAs) everywhere; the 16 x 32 tile (6 per 8) was measured on the
// 64-cout convs in rounds 2 and 4: no gain (profiles/HISTORY.md)
int conv2d_fwd_f32(const codon_conv_desc* d, const float* x, const float* w, float* y, const float* res,
                   hipStream_t stream) {
  const int key = d->ksize * 1000000 + d->cin * 1000 + d->cout;
  switch (key) {
    case 5128128: return launch_conv<5, 128, 128, 2>(d, x, w, y, res, stream);
    case 5064064: return launch_conv<5, 64, 64, 2>(d, x, w, y, res, stream);
    case 3064064: return launch_conv<3, 64, 64, 2>(d, x, w, y, res, stream);
    case 3128064: return launch_conv<3, 128, 64, 2>(d, x, w, y, res, stream);
    case 3064128: return launch_conv<3, 64, 128, 2>(d, x, w, y, res, stream);  // dgrad of conv7
    case 1128064: return launch_conv<1, 128, 64, 2>(d, x, w, y, res, stream);
    case 1064128: return launch_conv<1, 64, 128, 2>(d, x, w, y, res, stream);  // dgrad of confuse*
    default:
      set_error("conv2d_fwd: no f32 kernel for k=%d cin=%d cout=%d", d->ksize, d->cin, d->cout);
      return CODON_ERR_UNSUPPORTED;
  }
}

int pack_weight_f32(const float* w, float* out, int cout, int cin, int ks, int mode, hipStream_t stream) {
  const long n = (long)cout * cin * ks * ks;
  const int blocks = (int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
  hipLaunchKernelGGL(pack_weight_f32_kernel, dim3(blocks), dim3(256), 0, stream, w, out, cout, cin, ks,
                     conv_ck(ks), mode == CODON_PACK_DGRAD ? 1 : 0);
  return check_launch("pack_weight_f32_kernel");
}

}  // namespace codon
