// 16-bit (bf16 / fp16) implicit-GEMM convolution on the gfx950 matrix cores over CHANNEL-BLOCKED activations (c8.h):
// v_mfma_f32_32x32x16_{bf16,f16}, fp32 accumulate.  The conv is nn.Conv2d stride 1 / pad k//2 / no bias of
// /root/reference/CODON_X4/CODON_x4.py:24-47; BASELINE.json configs[2] and [4], and the reference script's own .half().
//
// Operand roles as in the fp32 kernel (conv_mfma_f32.hip): A = weights -> cout rows, B = activations -> pixel columns,
// so a 32x32 result tile has one pixel column per lane and 16 cout rows in registers.  K per MFMA = 16 input channels,
// 8 consecutive ones per lane -- which is exactly one 16-byte vector of the C8 layout:
//   xs[cb][row][col] : halo tile of one 16-channel chunk, 16-byte elements; staged by straight 16-byte copies
//                      (one buffer_load_b128 + one conflict-free ds_write_b128 per element: the LDS index IS the
//                      thread's element number); a half-wave's B fragment = 32 consecutive elements of a tile row
//   ws[dx][cb][cout] : 16-byte elements = 8 input channels of one (tap, cout); the packer puts cout
//                      tile*32 + swap23(i) in row i of every 32-row tile, so accumulator registers 8g..8g+7 of lane
//                      half h are the 8 channels of output plane tile*4 + 2g + h: the epilogue converts them pairwise
//                      and writes ONE 16-byte vector (a half-wave = 512 contiguous bytes of one plane).
// K loop: stages = (chunk of 16 channels, filter row dy): KS taps x (PSEG pixel rows x COUT/32) MFMAs per wave and
// stage; xs / ws double-buffered, the next stage requested before the MFMAs and written to LDS after them, one
// barrier per stage; the stage loop is unrolled over a chunk pair x KS so every LDS access is base + immediate.
// The chained 1x1 (FUSE): the rounded 8-channel vectors a lane would store are already the B operand (k = 8h + j in
// NATURAL channel order, thanks to swap23) of the 128 -> 64 conv that consumes them.

#include <math.h>
#include <stdlib.h>
#include <type_traits>

#include "c8.h"
#include "pair.h"

namespace codon {

struct ConvC8Params {
  const uint4* x;
  const uint4* w;       // packed: [chunk][dy][dx][cb in chunk (2)][cout position] x 16 B
  uint4* y;
  const uint4* res;
  int H, W;
  long x_img, y_img, r_img;      // 16-byte vectors per image of each buffer: (ctotal / 8) * H * W
  long x_base, y_base, r_base;   // first vector of the slice inside an image: (coff / 8) * H * W
  int tiles_x, tiles_y, nblk;
  int flags;
  // FUSE only: chained 1x1 (128 -> 64)
  const uint4* w2;      // [t2][t][g][lane] x 16 B: W1[t2*32 + swap23(lane&31)][t*32 + 16g + 8(lane>>5) + j], j = 0..7
  uint4* y2;
  long y2_img, y2_base;
  // GATE only: the staged input is pre * (ch * sp) + inputs (CAC gate-apply of the producing block), x = pre
  const uint4* gin;     // `inputs` buffer
  long g_img, g_base;
  const float* gch;     // (B,64) channel gate: channel c of the slice uses gch[c & 63]
  const float* gsp;     // (B,1,H,W) spatial gate
  uint4* gout;          // optional: the gate-applied input itself is ALSO written here (each tile its own pixels), so a
  long go_img, go_base; // sibling conv on the same input can run plain (codon_conv2d_gated_emit_fwd)
  // FUSE only, optional (st_pool != nullptr): CAC statistics of the 64 channels this launch produces, from the epilogue
  float* st_pool;       // (B,2,H,W): per pixel { max, SUM } over this stream's 64 channels (ChannelPool, CAC_module.py:81)
  float* st_part;       // (B, tiles, 128, 2): per tile, per channel { sum, max } (first stage of the pools, :43,47)
  int st_choff;         // 0 = colour stream (Fcat channels 0..63), 64 = depth stream
};
static_assert(sizeof(ConvC8Params) <= CODON_KERNARG_LIMIT, "passed by value as a kernel argument");

template <int N, class F, int I = 0>
__device__ __forceinline__ void static_for_c8(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_c8<N, F, I + 1>(static_cast<F&&>(f));
  }
}

__device__ __forceinline__ float relu1_c8(float v) {   // one v_max (fmaxf adds a canonicalising v_max(v, v))
  float o;
  asm("v_max_f32 %0, 0, %1" : "=v"(o) : "v"(v));
  return o;
}

enum { RESC8_NONE = 0, RESC8_ADD = 1, RESC8_MASK = 2, RESC8_SUMINTO = 3 };
// internal flag bit (never part of the ABI's CODON_CONV_* set): p.res is a READ-WRITE running sum that takes the value this
// launch stores, sum += y, in the same epilogue (codon_conv2d_sum_into_fwd; conv5x5 64->64 only)
constexpr int CONV_C8_SUM_INTO = 1 << 16;

// pixel rows per wave: the 5x5 64-cout kernel takes 4 (16x32 tile) so that, like the 128-cout ones, a filter tap is
// 8 MFMAs on 6 operand fetches.  (Measured and settled, profiles/HISTORY.md: 16x32 tiles for the 3x3 convs, 8-wave
// workgroups for any of the shapes, staging through registers instead of LDS-DMA -- all slower; the switches are gone.)
constexpr bool C8_DMA = true;     // x and weights go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds); GATE kernels stage x through registers
template <int KS, int COUT> struct ConvC8Pseg { static constexpr int value = (COUT == 64 && KS == 5) ? 4 : 2; };
// waves per workgroup of the staged kernels (the resident-filter conv3x3 runs 16)
template <int KS, int COUT> struct ConvC8Nw { static constexpr int value = 4; };

// PERSIST (round 4): a workgroup walks a contiguous RANGE of tiles instead of one.  The stage pipeline simply continues
// across the tile edge: the last stage of tile t requests stage 0 of tile t+1 (its first halo chunk through tile t+1's
// gather plan, the first weight row), the epilogue of tile t runs while those pieces are in flight -- the per-tile
// prologue (plan + a full memory round trip with nothing to overlap it inside the workgroup) is paid once per range.
// Same MFMAs on the same operands in the same order per tile: results are bit-identical to the one-tile-per-workgroup form.
// RESW (with PERSIST): the WHOLE packed filter stays resident in LDS for the workgroup's lifetime -- loaded once, never
// re-staged (a one-tile conv3x3 64->64 workgroup stages 72 KB of weights for 43 KB of input and 32 KB of output); the only
// per-chunk traffic left is the halo tile, and the only barrier the one that publishes it (per chunk, not per stage).
// The body takes its parameter block by reference and the (XCD-remapped) tile it starts on as an argument, so that the
// same code serves the one-conv launch and the PAIR launch (conv_c8_pair_kernel below: two convs of one shape, e.g. the
// depth and the colour stream of a block, as one grid).
// LDS of one workgroup of the staged (non-resident) kernels, in 16-byte elements (the body's own arithmetic, restated for
// kernels that hold the arena themselves)
template <int KS, int CIN, int COUT, int NW> constexpr int conv_c8_lds_vecs();

// EXTLDS (round 6): the staging arena is the CALLER's (mix53 below runs two different bodies in one grid on one arena of the
// larger size); every other instantiation declares its own, as before.
template <class E, int KS, int CIN, int COUT, bool FUSE = false, int NW = 4, bool GATE = false, bool PERSIST = false,
          bool RESW = false, bool EXTLDS = false>
__device__ __forceinline__ void conv_c8_body(const ConvC8Params& p, const int tile0, uint4* ext_lds = nullptr) {
  static_assert(!PERSIST || (!FUSE && !GATE && C8_DMA), "the tile loop exists for the plain LDS-DMA convs");
  static_assert(!RESW || PERSIST, "a resident filter pays only over many tiles");
  typedef typename E::vec8 vec8;
  typedef const volatile __attribute__((address_space(3))) u32x4* lds_rd;
  typedef volatile __attribute__((address_space(3))) u32x4* lds_w128;
  constexpr int PAD = KS / 2;
  constexpr int PSEG = ConvC8Pseg<KS, COUT>::value;
  constexpr int NT = 64 * NW;
  constexpr int TW = 32, TH = NW * PSEG;
  constexpr int XR = TH + KS - 1, XQ = TW + KS - 1;
  constexpr int CK = 16, NCB = CK / 8;
  constexpr int NCHUNK = CIN / CK;
  constexpr int XS = NCB * XR * XQ;        // 16-byte elements per input tile
  constexpr int WS = KS * NCB * COUT;      // 16-byte elements per weight stage (one filter row of one chunk)
  constexpr int CT = COUT / 32;
  constexpr int NST = NCHUNK * KS;
  constexpr int XE = (XS + NT - 1) / NT, WE = (WS + NT - 1) / NT;
  // padded to whole staging rounds (no store predicates); RESW: to whole waves (a wave past the tile issues nothing)
  constexpr int XSP = RESW ? ((XS + 63) / 64) * 64 : XE * NT, WSP = WE * NT;
  constexpr int WTOT = RESW ? NST * WS : 2 * WSP;  // weight region: the whole filter, or two stage buffers
  static_assert(NCHUNK % 2 == 0, "the stage loop is unrolled over chunk pairs");
  static_assert((XSP + (KS + PSEG) * XQ) * 16 < 65536 && 2 * WSP * 16 < 65536 && (!RESW || 2 * KS * WS * 16 < 65536),
                "LDS immediates are 16 bits per region");

  static_assert(RESW || 2 * XSP + WTOT == conv_c8_lds_vecs<KS, CIN, COUT, NW>(), "conv_c8_lds_vecs restates this");
  __shared__ uint4 lds_own[EXTLDS ? 1 : 2 * XSP + WTOT];
  uint4* const lds = EXTLDS ? ext_lds : lds_own;
  uint4* const xs0 = lds;
  uint4* const ws0 = lds + 2 * XSP;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;

  // this workgroup's tiles: one (the XCD-remapped block index), or -- PERSIST -- every gridDim.x-th tile from there: at any
  // moment the resident workgroups work on ~gridDim.x CONSECUTIVE tiles, as a one-tile-per-workgroup launch does, so
  // neighbouring tiles still meet in an XCD's L2 (a contiguous range per workgroup was measured first: 0.96 vs 0.82 ms,
  // every halo re-read came from HBM)
  int t_cur = tile0, t_step;
  if constexpr (PERSIST) t_step = (int)gridDim.x;
  else t_step = p.nblk;
  int tx, ty, b;
  auto decode = [&](int tile, int& tx_, int& ty_, int& b_) {
    unsigned bid = (unsigned)tile;
    tx_ = bid % p.tiles_x;
    bid /= p.tiles_x;
    ty_ = bid % p.tiles_y;
    b_ = bid / p.tiles_y;
  };
  decode(t_cur, tx, ty, b);
  int tx0 = tx * TW, ty0 = ty * TH;
  const int H = p.H, W = p.W;
  const unsigned HW16 = 16u * (unsigned)H * (unsigned)W;   // bytes per 8-channel plane

  __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.x + (long)b * p.x_img + p.x_base), 0, (int)((unsigned)(CIN / 8) * HW16), C8_RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)(NST * WS * 16), C8_RSRC_FLAGS);

  // gather plan: element e = tid + NT k = (plane cb of the chunk, row r, col q) of xs[cb][r][q]; xoff = byte offset of
  // that pixel's vector in plane cb, or out of range (zero padding; padding elements of the last round likewise)
  unsigned xoff[XE];
  int xcb[GATE ? XE : 1];        // GATE: plane (0 / 1) of the element inside its chunk, and its pixel's spatial gate
  float xsp[GATE ? XE : 1];
  unsigned xown = 0;             // GATE: bit k = element k is one of the tile's OWN pixels (not halo), inside the image
  // PERSIST: the gather plan of another tile, straight into xoff[] (element e = tid + NT k -> (cb, r, q) by constant divisions)
  auto plan_for = [&](int ty0_, int tx0_) {
#pragma unroll
    for (int k = 0; k < (PERSIST ? XE : 0); ++k) {
      const int e_ = tid + k * NT;
      const int cb_ = e_ / (XR * XQ), rem_ = e_ - cb_ * (XR * XQ);
      const int r_ = rem_ / XQ, q_ = rem_ - r_ * XQ;
      const int gy = ty0_ + r_ - PAD, gx = tx0_ + q_ - PAD;
      const bool ok = cb_ < NCB && gy >= 0 && gy < H && gx >= 0 && gx < W;
      xoff[k] = ok ? (unsigned)cb_ * HW16 + 16u * (unsigned)(gy * W + gx) : C8_OOB;
    }
  };
  {
    constexpr int DQ = NT % XQ, DR = (NT / XQ) % XR, DC = (NT / XQ) / XR;
    int cb = tid / (XR * XQ);
    int rem = tid - cb * (XR * XQ);
    int r = rem / XQ, q = rem - r * XQ;
#pragma unroll
    for (int k = 0; k < XE; ++k) {
      const int gy = ty0 + r - PAD, gx = tx0 + q - PAD;
      const bool ok = cb < NCB && gy >= 0 && gy < H && gx >= 0 && gx < W;
      xoff[k] = ok ? (unsigned)cb * HW16 + 16u * (unsigned)(gy * W + gx) : C8_OOB;
      if constexpr (GATE) {
        xcb[k] = cb & 1;
        const float g_ = p.gsp[(long)b * H * W + (ok ? gy * W + gx : 0)];
        xsp[k] = ok ? g_ : 0.f;
        if (ok && r >= PAD && r < PAD + TH && q >= PAD && q < PAD + TW) xown |= 1u << k;
      }
      q += DQ; r += DR; cb += DC;
      if (q >= XQ) { q -= XQ; r += 1; }
      if (r >= XR) { r -= XR; cb += 1; }
    }
  }
  __shared__ float chs[GATE ? 64 : 1];
  if constexpr (GATE) {
    if (tid < 64) chs[tid] = p.gch[b * 64 + tid];
  }
  const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(GATE ? p.gin + (long)b * p.g_img + p.g_base : p.x), 0, (int)((unsigned)(CIN / 8) * HW16), C8_RSRC_FLAGS);
  const bool emit = GATE && p.gout != nullptr;           // wave-uniform
  const __amdgpu_buffer_rsrc_t gorsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(emit ? p.gout + (long)b * p.go_img + p.go_base : (uint4*)p.y), 0, (int)((unsigned)(CIN / 8) * HW16), C8_RSRC_FLAGS);
  const lds_w128 xwr = (lds_w128)(xs0 + tid);
  const unsigned wvo = (unsigned)tid * 16u;
  const unsigned wvo_last = (WS % NT == 0 || tid + (WE - 1) * NT < WS) ? wvo : C8_OOB;
  const lds_w128 ww = (lds_w128)(ws0 + tid);
  const lds_rd xrd = (lds_rd)(xs0 + (half * XR + wave * PSEG) * XQ + l31);
  const lds_rd wrd = (lds_rd)(ws0 + half * COUT + l31);

  // the next chunk's halo tile is requested in two halves, during the last two filter rows of the current chunk
  constexpr int XE1 = XE / 2, XEH = XE - XE1;
  constexpr bool DMA = C8_DMA && !GATE;      // the halo tile: by LDS-DMA, or -- GATE: the gate arithmetic needs the VALU -- through registers
  // (the weight rows of a GATE kernel by LDS-DMA while x goes through registers: measured in round 5, 1.80 vs 1.74 ms gated,
  // 1.98 vs 1.91 gated + emitting -- the vmcnt(0) in front of the barrier then also waits for the emit stores;
  // tools/probes/gate_weights_dma_experiment.patch)
  constexpr bool DMAW = DMA;
  typedef __attribute__((address_space(3))) void lds_void;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  u32x4 xv[DMA ? 1 : XEH];
  u32x4 xg[GATE ? XEH : 1];
  u32x4 wr[DMAW ? 1 : WE];

#define LOAD_XP(rs_, off_, chunk_, buf_, k0_, k1_)                                      \
  {                                                                                     \
    const unsigned so_ = (unsigned)(chunk_) * (unsigned)NCB * HW16;                     \
    _Pragma("unroll") for (int k = (k0_); k < (k1_); ++k) {                             \
      if constexpr (DMA) {                                                              \
        const unsigned vo_ = off_[k];   /* passed as off_[k] the host pass drops the kernel's stub (hipcc 7.2) */ \
        /* a wave whose 64 slots all lie past the tile (last round) issues nothing: the slots are never read */ \
        if (XS % NT == 0 || k < XE - 1 || k * NT + wave_u * 64 < XS)                    \
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_void*)(lds + (buf_) * XSP + k * NT + wave_u * 64), 16, \
                                                   vo_, so_, 0, 0);                     \
      } else {                                                                          \
        xv[k - (k0_)] = c8_ld(rs_, off_[k], so_);                                       \
        if constexpr (GATE) xg[k - (k0_)] = c8_ld(grsrc, off_[k], so_);                 \
      }                                                                                 \
    }                                                                                   \
  }
#define LOAD_X(chunk_, buf_, k0_, k1_) LOAD_XP(xrsrc, xoff, chunk_, buf_, k0_, k1_)
  // GATE: the staged vector is pre * (ch * sp) + inputs, formed in fp32 and rounded once -- the arithmetic of
  // cac_apply_c8_kernel followed by a plain load, bit for bit (out-of-image elements: 0 * g + 0 = 0)
#define STORE_X(chunk_, buf_, k0_, k1_)   /* buf_ compile time: immediate offsets */    \
  if constexpr (!DMA) {                                                                 \
    _Pragma("unroll") for (int k = (k0_); k < (k1_); ++k) {                             \
      if constexpr (GATE) {                                                             \
        const float* cg_ = chs + ((((chunk_) * NCB + xcb[k]) * 8) & 63);                \
        const float4 c0_ = *reinterpret_cast<const float4*>(cg_), c1_ = *reinterpret_cast<const float4*>(cg_ + 4); \
        const float cg8_[8] = {c0_.x, c0_.y, c0_.z, c0_.w, c1_.x, c1_.y, c1_.z, c1_.w};   \
        float v_[8], q_[8];                                                             \
        c8_unpack<E>(xv[k - (k0_)], v_);                                                \
        c8_unpack<E>(xg[k - (k0_)], q_);                                                \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) v_[j] = fmaf(v_[j], cg8_[j] * xsp[k], q_[j]); \
        const u32x4 pv_ = c8_pack<E>(v_);                                               \
        xwr[(buf_) * XSP + k * NT] = pv_;                                               \
        if (emit) c8_st(pv_, gorsrc, ((xown >> k) & 1u) ? xoff[k] : C8_OOB, (unsigned)(chunk_) * (unsigned)NCB * HW16); \
      } else {                                                                          \
        xwr[(buf_) * XSP + k * NT] = xv[k - (k0_)];                                     \
      }                                                                                 \
    }                                                                                   \
  }
#define LOAD_W(stage_, buf_)                                                            \
  if constexpr (!RESW) {                                                                \
    const unsigned so_ = (unsigned)(stage_) * (unsigned)(WS * 16);                      \
    _Pragma("unroll") for (int k = 0; k < WE; ++k) {                                    \
      if constexpr (DMAW) {                                                              \
        if (WS % NT == 0 || k < WE - 1 || k * NT + wave_u * 64 < WS)                    \
          __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_void*)(lds + 2 * XSP + (buf_) * WSP + k * NT + wave_u * 64), 16, \
                                                   k == WE - 1 ? wvo_last : wvo, so_ + k * (unsigned)(NT * 16), 0, 0); \
      } else {                                                                          \
        wr[k] = c8_ld(wrsrc, k == WE - 1 ? wvo_last : wvo, so_ + k * (unsigned)(NT * 16)); \
      }                                                                                 \
    }                                                                                   \
  }
#define STORE_W(buf_)                                                                   \
  if constexpr (!DMAW) {                                                                 \
    _Pragma("unroll") for (int k = 0; k < WE; ++k) ww[(buf_) * WSP + k * NT] = wr[k];   \
  }

  f32x16 acc[PSEG][CT];

  if constexpr (GATE) __syncthreads();        // chs
  if constexpr (XE1 > 0) {
    LOAD_X(0, 0, 0, XE1);
    STORE_X(0, 0, 0, XE1);
  }
  LOAD_X(0, 0, XE1, XE);
  LOAD_W(0, 0);
  if constexpr (RESW) {                              // the whole filter, once: NST * WS vectors in rounds of NT
    constexpr int WALL = NST * WS, WR = (WALL + NT - 1) / NT;
#pragma unroll
    for (int k = 0; k < WR; ++k) {
      const unsigned wo_ = (WALL % NT == 0 || k < WR - 1 || tid + k * NT < WALL) ? wvo : C8_OOB;
      if (WALL % NT == 0 || k < WR - 1 || k * NT + wave_u * 64 < WALL)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_void*)(lds + 2 * XSP + k * NT + wave_u * 64), 16, wo_,
                                                 (unsigned)k * (unsigned)(NT * 16), 0, 0);
    }
  }
  STORE_X(0, 0, XE1, XE);
  STORE_W(0);
  if constexpr (DMAW) __builtin_amdgcn_s_waitcnt(0x0070);    // vmcnt(0): this wave's pieces have landed
  __syncthreads();

  static_assert(!PERSIST || (NST % 2 == 0), "the tile loop re-enters with stage parity 0");
#pragma unroll 1
  for (;;) {       // tiles of this workgroup (one trip unless PERSIST)
  const bool next_tile = PERSIST && (t_cur + t_step < p.nblk);       // wave-uniform
  // the CURRENT tile's coordinates for the epilogue (tx, ty, b, tx0, ty0 move on to the next tile inside the last chunk)
  const int tx_e = tx, ty_e = ty, b_e = b, tx0_e = tx0, ty0_e = ty0;
  (void)tx_e; (void)ty_e;
#pragma unroll
  for (int i = 0; i < PSEG; ++i)
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;

  // stage (chunk, dy): weights of filter row dy for 16 channels in ws[(chunk*KS + dy) & 1], the chunk's halo tile in
  // xs[chunk & 1].  Unrolled over (chunk parity, dy): KS odd, so the stage parity is (par + dy) & 1.
#pragma unroll 1
  for (int c2 = 0; c2 < NCHUNK; c2 += 2) {
    const lds_rd wrd_c2 = wrd + (RESW ? c2 * KS * WS : 0);       // resident filter: this chunk pair's rows
    (void)wrd_c2;
    static_for_c8<2 * KS>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      constexpr int par = u / KS, dy = u % KS;                    // chunk parity, filter row
      constexpr int sbuf = (par * KS + dy) & 1;                   // c2 is even: stage parity is compile time
      const int chunk = c2 + par;
      const int s = chunk * KS + dy;
      constexpr bool tail = (par == 1 && dy == KS - 1);           // last stage of the pair
      const bool has_next = !tail || (c2 + 2 < NCHUNK);
      // PERSIST: behind the last chunk comes chunk 0 / stage 0 of the NEXT tile (buffer parities continue: NST is even).
      // The last chunk's halo tile was requested during the chunk before it, so from its first stage on xoff[] / xrsrc
      // are free to become the next tile's plan: no second set of plan registers is ever live.
      const bool wrap = PERSIST && par == 1 && (c2 + 2 >= NCHUNK) && next_tile;   // wave-uniform
      if constexpr (PERSIST && par == 1 && dy == 0) {
        if (wrap) {
          decode(t_cur + t_step, tx, ty, b);
          tx0 = tx * TW; ty0 = ty * TH;
          plan_for(ty0, tx0);
          xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (long)b * p.x_img + p.x_base), 0,
                                                    (int)((unsigned)(CIN / 8) * HW16), C8_RSRC_FLAGS);
        }
      }
      // RESW: nothing is waited for inside a chunk, so the NEXT chunk's halo tile is requested at the chunk's FIRST stage --
      // a whole chunk of lead instead of one or two stages.  (Counters on the 3x3 kernels, tools/probes/pmc_conv3.sh: waves
      // parked at s_waitcnt / barrier 52-55 % of their cycles against 9-13 % in the 5x5 kernels -- with 12 MFMAs per stage a
      // tile half requested at the start of the chunk's last stage has 0.2 us to cross the memory system.)
      constexpr bool XEARLY = RESW;
      if constexpr (XEARLY && dy == 0) {
        if (!(par == 1 && c2 + 2 >= NCHUNK)) { LOAD_X(chunk + 1, par ^ 1, 0, XE); }
        else if (wrap) { LOAD_X(0, par ^ 1, 0, XE); }
      }
      if (has_next) {
        LOAD_W(s + 1, sbuf ^ 1);
        if constexpr (dy == KS - 1 && !XEARLY) LOAD_X(chunk + 1, par ^ 1, XE1, XE);
      } else if constexpr (PERSIST && tail) {
        if (wrap) {
          LOAD_W(0, sbuf ^ 1);
          if constexpr (!XEARLY) LOAD_X(0, par ^ 1, XE1, XE);
        }
      }
      if constexpr (XE1 > 0 && dy == KS - 2 && !XEARLY) {
        if (!(par == 1 && c2 + 2 >= NCHUNK)) { LOAD_X(chunk + 1, par ^ 1, 0, XE1); }
        else if constexpr (PERSIST) { if (wrap) LOAD_X(0, par ^ 1, 0, XE1); }
      }

      // operand fetch one filter tap ahead of its MFMAs, in two register sets.  PIN: sched_barrier holds that order
      // (measured r2: pinned wins on the 8-MFMA taps of the 5x5 convs, unpinned on the 3x3 ones)
      constexpr bool PIN = (KS == 5 && PSEG * CT == 8);
      vec8 a[2][CT], bv[2][PSEG];
#define FETCH_A(q_, t_)                                                                                   \
  {                                                                                                       \
    const u32x4 v_ = RESW ? wrd_c2[(par * KS + dy) * WS + (q_) * NCB * COUT + (t_) * 32]                  \
                          : wrd[sbuf * WSP + (q_) * NCB * COUT + (t_) * 32];                              \
    a[(q_) & 1][t_] = *reinterpret_cast<const vec8*>(&v_);                                                \
  }
#define FETCH_B(q_)                                                                                       \
  {                                                                                                       \
    _Pragma("unroll") for (int i = 0; i < PSEG; ++i) {                                                    \
      const u32x4 v_ = xrd[par * XSP + (dy + i) * XQ + (q_)];                                             \
      bv[(q_) & 1][i] = *reinterpret_cast<const vec8*>(&v_);                                              \
    }                                                                                                     \
  }
      static_for_c8<CT>([&](auto tc) { FETCH_A(0, decltype(tc)::value) });
      FETCH_B(0)
      static_for_c8<KS>([&](auto dc) {
        constexpr int q = decltype(dc)::value;
        if constexpr (q + 1 < KS) {
          FETCH_B(q + 1)
          static_for_c8<CT>([&](auto tc) { FETCH_A(q + 1, decltype(tc)::value) });
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

      if constexpr (XE1 > 0 && dy == KS - 2) {
        if (!(par == 1 && c2 + 2 >= NCHUNK)) STORE_X(chunk + 1, par ^ 1, 0, XE1);
      }
      if (has_next) {
        STORE_W(sbuf ^ 1);
        if constexpr (dy == KS - 1) STORE_X(chunk + 1, par ^ 1, XE1, XE);
      }
      // RESW: nothing changes hands inside a chunk -- one barrier per chunk, the one that publishes the next halo tile
      if constexpr (!RESW || dy == KS - 1) {
        if constexpr (DMAW) __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) before the barrier: DMA pieces of the next stage are in LDS
        __syncthreads();
      }
    });
  }
#undef LOAD_X
#undef LOAD_XP
#undef STORE_X
#undef LOAD_W
#undef STORE_W

  // epilogue.  Lane term of every output address: this lane's pixel in plane `half`; the plane pair (t, g) is wave
  // uniform and goes into the scalar offset.  Off-image pixels are out of range.
  const int gx = tx0_e + l31;
  unsigned vo[PSEG];
#pragma unroll
  for (int i = 0; i < PSEG; ++i) {
    const int gy = ty0_e + wave * PSEG + i;
    vo[i] = (gx < W && gy < H) ? (unsigned)half * HW16 + 16u * (unsigned)(gy * W + gx) : C8_OOB;
  }
  const bool relu = p.flags & CODON_CONV_RELU;
  const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.res ? p.res + (long)b_e * p.r_img + p.r_base : p.x), 0, (int)((unsigned)((FUSE ? 64 : COUT) / 8) * HW16),
      C8_RSRC_FLAGS);
  auto cplane = [&](int t, int g) { return (unsigned)(t * 4 + 2 * g) * HW16; };

  if constexpr (FUSE) {
    static_assert(!FUSE || COUT == 128, "chained 1x1 is 128 -> 64");
    u32x4 pk[PSEG][CT][2];
#pragma unroll
    for (int i = 0; i < PSEG; ++i)
#pragma unroll
      for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          float v8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float v = acc[i][t][8 * g + j];
            v8[j] = relu ? relu1_c8(v) : v;
          }
          pk[i][t][g] = c8_pack<E>(v8);
        }
    if (p.y) {
      const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(p.y + (long)b_e * p.y_img + p.y_base), 0, (int)((unsigned)(COUT / 8) * HW16), C8_RSRC_FLAGS);
#pragma unroll
      for (int i = 0; i < PSEG; ++i)
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
          for (int g = 0; g < 2; ++g) c8_st(pk[i][t][g], yrsrc, vo[i], cplane(t, g));
    }
    const __amdgpu_buffer_rsrc_t w2rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, 64 * 128 * 2, C8_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t y2rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.y2 + (long)b_e * p.y2_img + p.y2_base), 0, (int)(8u * HW16), C8_RSRC_FLAGS);
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
          const u32x4 av = c8_ld(w2rsrc, w2vo, (unsigned)(((t2 * CT + t) * 2 + g) * 1024));
          const vec8 a = *reinterpret_cast<const vec8*>(&av);
#pragma unroll
          for (int i = 0; i < PSEG; ++i) d[t2][i] = E::mfma(a, *reinterpret_cast<const vec8*>(&pk[i][t][g]), d[t2][i]);
        }
    }
    float pmx[PSEG], psm[PSEG], csum[32], cmax[32];
    bool valid[PSEG];
#pragma unroll
    for (int i = 0; i < PSEG; ++i) { pmx[i] = -INFINITY; psm[i] = 0.f; valid[i] = vo[i] != C8_OOB; }
#pragma unroll
    for (int k = 0; k < 32; ++k) { csum[k] = 0.f; cmax[k] = -INFINITY; }
#pragma unroll
    for (int i = 0; i < PSEG; ++i) {
      u32x4 rv[2][2];
      if (p.res) {
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
          for (int g = 0; g < 2; ++g) rv[t2][g] = c8_ld(rrsrc, vo[i], cplane(t2, g));
      }
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          float v8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v8[j] = d[t2][i][8 * g + j];
          if (p.res) {
            float r8[8];
            c8_unpack<E>(rv[t2][g], r8);
#pragma unroll
            for (int j = 0; j < 8; ++j) v8[j] += r8[j];
          }
          const u32x4 q = c8_pack<E>(v8);
          c8_st(q, y2rsrc, vo[i], cplane(t2, g));
          if (p.st_pool) {       // statistics of the values AS STORED (rounded to 16 bits), like a pass over the tensor
            float r8[8];
            c8_unpack<E>(q, r8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              pmx[i] = fmaxf(pmx[i], r8[j]);
              psm[i] += r8[j];
              csum[t2 * 16 + g * 8 + j] += valid[i] ? r8[j] : 0.f;
              cmax[t2 * 16 + g * 8 + j] = valid[i] ? fmaxf(cmax[t2 * 16 + g * 8 + j], r8[j]) : cmax[t2 * 16 + g * 8 + j];
            }
          }
        }
    }
    if (p.st_pool) {
      // per pixel: this lane holds 32 of the pixel's 64 channels, lane + 32 the other 32
      const long HWl = (long)H * W;
#pragma unroll
      for (int i = 0; i < PSEG; ++i) {
        const float m2 = fmaxf(pmx[i], __shfl_xor(pmx[i], 32, 64)), s2 = psm[i] + __shfl_xor(psm[i], 32, 64);
        if (half == 0 && valid[i]) {
          const long q = (long)(ty0_e + wave * PSEG + i) * W + gx;
          p.st_pool[(long)b_e * 2 * HWl + q] = m2;
          p.st_pool[(long)b_e * 2 * HWl + HWl + q] = s2;
        }
      }
      // per channel: transpose through LDS (free after the last stage's barrier): lane L writes its 32 values as a
      // row of 36 floats (144-byte pitch: conflict-free ds_write_b128), then lane (c, half) sums column c over the 32
      // rows of its half; sums first, maxima second, through the same region
      float* const tr = reinterpret_cast<float*>(lds) + wave * (64 * 36);
      float* const red = reinterpret_cast<float*>(lds) + NW * (64 * 36);      // [2][NW][64]
      static_assert((NW * 64 * 36 + 2 * NW * 64) * 4 <= (2 * XSP + 2 * WSP) * 16, "statistics scratch fits the stage buffers");
      float rs = 0.f, rm = -INFINITY;
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const float* src = pass ? cmax : csum;
#pragma unroll
        for (int k = 0; k < 8; ++k)
          *reinterpret_cast<float4*>(tr + lane * 36 + 4 * k) = make_float4(src[4 * k], src[4 * k + 1], src[4 * k + 2], src[4 * k + 3]);
        __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0): this wave's own writes have landed
        __builtin_amdgcn_wave_barrier();
        float a = pass ? -INFINITY : 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) {
          const float v = tr[(half * 32 + r) * 36 + l31];
          a = pass ? fmaxf(a, v) : a + v;
        }
        if (pass) rm = a; else rs = a;
        __builtin_amdgcn_wave_barrier();
      }
      red[(0 * NW + wave) * 64 + lane] = rs;
      red[(1 * NW + wave) * 64 + lane] = rm;
      __syncthreads();
      if (wave == 0) {
        float s = red[lane], m = red[NW * 64 + lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) {            // fixed order: deterministic
          s += red[w * 64 + lane];
          m = fmaxf(m, red[(NW + w) * 64 + lane]);
        }
        const int ch = (l31 >> 4) * 32 + ((l31 >> 3) & 1) * 16 + 8 * half + (l31 & 7);   // value index -> channel (swap23 layout)
        const long tile = (long)ty_e * p.tiles_x + tx_e;
        float2* out = reinterpret_cast<float2*>(p.st_part) +
                      (((long)b_e * p.tiles_x * p.tiles_y + tile) * 128 + p.st_choff + ch);
        *out = make_float2(s, m);
      }
    }
    return;
  }

  // ReLU / residual / mask / accumulate as compile-time variants selected by wave-uniform branches
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.y + (long)b_e * p.y_img + p.y_base), 0, (int)((unsigned)(COUT / 8) * HW16), C8_RSRC_FLAGS);
  auto epi = [&](auto relu_c, auto res_c, auto acc_c, auto ms_c) {
    constexpr bool RELU = decltype(relu_c)::value;
    constexpr int RES = decltype(res_c)::value;
    constexpr bool ACC = decltype(acc_c)::value;
    constexpr bool MS = decltype(ms_c)::value;      // MASK_SUM: the mask applies to conv + previous value
#pragma unroll
    for (int i = 0; i < PSEG; ++i) {
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        u32x4 rv[2], av[2];
        if constexpr (RES != RESC8_NONE) {
#pragma unroll
          for (int g = 0; g < 2; ++g) rv[g] = c8_ld(rrsrc, vo[i], cplane(t, g));
        }
        if constexpr (ACC) {
#pragma unroll
          for (int g = 0; g < 2; ++g) av[g] = c8_ld(yrsrc, vo[i], cplane(t, g));
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          float v8[8], r8[8], a8[8];
          if constexpr (RES != RESC8_NONE) c8_unpack<E>(rv[g], r8);
          if constexpr (ACC) c8_unpack<E>(av[g], a8);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float v = acc[i][t][8 * g + j];
            if constexpr (RELU) v = relu1_c8(v);
            if constexpr (RES == RESC8_ADD) v += r8[j];
            if constexpr (RES == RESC8_MASK && !MS) v = r8[j] > 0.f ? v : 0.f;
            if constexpr (ACC) v += a8[j];
            if constexpr (RES == RESC8_MASK && MS) v = r8[j] > 0.f ? v : 0.f;
            v8[j] = v;
          }
          const u32x4 q = c8_pack<E>(v8);
          c8_st(q, yrsrc, vo[i], cplane(t, g));
          if constexpr (RES == RESC8_SUMINTO) {
            // sum += the value AS STORED (one 16-bit rounding each): the arithmetic of the pass this replaces
            // (cac_bwd_reduce_acc: g_inputs += g_out read back from HBM)
            float y8[8];
            c8_unpack<E>(q, y8);
#pragma unroll
            for (int j = 0; j < 8; ++j) r8[j] += y8[j];
            c8_st(c8_pack<E>(r8), rrsrc, vo[i], cplane(t, g));
          }
        }
      }
    }
  };
  using T = std::true_type;
  using F = std::false_type;
  using R0 = std::integral_constant<int, RESC8_NONE>;
  using R1 = std::integral_constant<int, RESC8_ADD>;
  using R2 = std::integral_constant<int, RESC8_MASK>;
  constexpr bool HAS_SUMINTO = KS == 5 && CIN == 64 && COUT == 64 && !FUSE && !GATE && !PERSIST;
  if constexpr (HAS_SUMINTO) {
    if (p.res && (p.flags & CONV_C8_SUM_INTO)) {          // wave-uniform
      using R3 = std::integral_constant<int, RESC8_SUMINTO>;
      if (p.flags & CODON_CONV_ACCUM_OUT) epi(F{}, R3{}, T{}, F{});
      else epi(F{}, R3{}, F{}, F{});
      break;
    }
  }
  const int res_mode = !p.res ? RESC8_NONE : (p.flags & CODON_CONV_MASK_RELU) ? RESC8_MASK
                                           : (p.flags & CODON_CONV_ADD_RESIDUAL) ? RESC8_ADD : RESC8_NONE;
  const bool accum = p.flags & CODON_CONV_ACCUM_OUT;
  const bool msum = p.flags & CODON_CONV_MASK_SUM;
  auto by_acc = [&](auto relu_c, auto res_c) {
    if constexpr (!decltype(relu_c)::value && decltype(res_c)::value == RESC8_MASK) {
      if (accum && msum) { epi(relu_c, res_c, T{}, T{}); return; }
    }
    if (accum) epi(relu_c, res_c, T{}, F{});
    else epi(relu_c, res_c, F{}, F{});
  };
  auto by_res = [&](auto relu_c) {
    if (res_mode == RESC8_NONE) by_acc(relu_c, R0{});
    else if (res_mode == RESC8_ADD) by_acc(relu_c, R1{});
    else by_acc(relu_c, R2{});
  };
  if (relu) by_res(T{});
  else by_res(F{});

  if constexpr (!PERSIST) {
    break;
  } else {
    if (!next_tile) break;
    t_cur += t_step;       // its plan, descriptor and coordinates were installed in the last chunk; stage 0 is in LDS
  }
  }   // tiles
}

template <class E, int KS, int CIN, int COUT, bool FUSE = false, int NW = 4, bool GATE = false, bool PERSIST = false,
          bool RESW = false>
__global__ __launch_bounds__(64 * NW, (NW > 8 ? 1 : 2)) void conv_c8_kernel(const ConvC8Params p) {
  conv_c8_body<E, KS, CIN, COUT, FUSE, NW, GATE, PERSIST, RESW>(p, (int)xcd_remap(blockIdx.x, PERSIST ? gridDim.x : (unsigned)p.nblk));
}

// Two convs of ONE shape and kernel variant as one grid of 2 * nblk workgroups (round 5).  The depth and the colour stream of
// a block are independent up to the CAC gate (/root/reference/CODON_X4/CODON_x4.py:75-84: conv2 | conv4, conv1 | conv5,
// conv3 + confuse | conv6 + confuse_c); at one image per call each of their launches fills the chip 1.4 times (370 x 463:
// 705 tiles on 512 slots), i.e. runs two rounds, the second a third full.  Two HIP streams overlapped them only as far as the
// fork / join events let them (12 + 18 us per block, profiles/r05_b1_*_timeline.txt); one grid has no seam.  Each workgroup
// picks its parameter block by its (XCD-remapped) index: same code, same order of operations per tile, same bits.
struct ConvC8Pair {
  ConvC8Params a, b;
};
static_assert(sizeof(ConvC8Pair) <= CODON_KERNARG_LIMIT, "two parameter blocks passed by value as one kernel argument");
template <class E, int KS, int CIN, int COUT, bool FUSE = false, int NW = 4, bool GATE = false>
__global__ __launch_bounds__(64 * NW, 2) void conv_c8_pair_kernel(const ConvC8Pair pp) {
  const int nblk = pp.a.nblk;                                       // == pp.b.nblk (checked on the host)
  const int v = (int)xcd_remap(blockIdx.x, 2u * (unsigned)nblk);
  const bool second = v >= nblk;                                    // workgroup-uniform
  conv_c8_body<E, KS, CIN, COUT, FUSE, NW, GATE, false, false>(second ? pp.b : pp.a, second ? v - nblk : v);
}

template <int KS, int CIN, int COUT, int NW> constexpr int conv_c8_lds_vecs() {
  constexpr int PSEG = ConvC8Pseg<KS, COUT>::value, NT = 64 * NW, TH = NW * PSEG, XR = TH + KS - 1, XQ = 32 + KS - 1;
  constexpr int XS = 2 * XR * XQ, WS = KS * 2 * COUT, XE = (XS + NT - 1) / NT, WE = (WS + NT - 1) / NT;
  return 2 * XE * NT + 2 * WE * NT;
}

// mix53 (round 6): conv8 (5x5 64->64, 16 x 32 tiles) and conv9 (3x3 64->64, 8 x 32 tiles) of the fusion trunk read the same
// tensor and are independent (/root/reference/CODON_X4/CODON_x4.py:123-124); at one image per call they are two launches of
// 2 and 3 rounds of workgroups (38 + 16 us at 370 x 463).  Held in one pair bracket they leave as ONE grid: workgroups
// [0, nA) run the 5x5 body on its tiles, [nA, nA + nB) the 3x3 body on its own, on one LDS arena of the larger size.  No XCD
// remap: every XCD takes every 8th workgroup of BOTH kinds; the expensive tiles are dispatched first, the cheap ones fill
// the tail.  Same body code per tile: same bits as the separate launches.
template <class E>
__global__ __launch_bounds__(256, 2) void conv_c8_mix53_kernel(const ConvC8Pair pp) {
  constexpr int LA = conv_c8_lds_vecs<5, 64, 64, 4>(), LB = conv_c8_lds_vecs<3, 64, 64, 4>();
  __shared__ uint4 arena[LA > LB ? LA : LB];
  const int nA = pp.a.nblk;
  if ((int)blockIdx.x < nA)                                          // workgroup-uniform
    conv_c8_body<E, 5, 64, 64, false, 4, false, false, false, true>(pp.a, (int)blockIdx.x, arena);
  else
    conv_c8_body<E, 3, 64, 64, false, 4, false, false, false, true>(pp.b, (int)blockIdx.x - nA, arena);
}

// ---- 1x1 convolution (stand-alone confuse* and their dgrad): HBM-bound ---------------------------------------------
// Y[co][pix] = sum_ci W[co][ci] X[ci][pix]: a plain GEMM over the flattened pixels of one image, no halo, no LDS.  A
// wave owns 64 consecutive pixels (two 32-pixel MFMA column tiles) and all COUT rows.  B fragment of lane (pixel,
// half h) for k-step ks = the 16-byte vector of plane 2 ks + h at that pixel: one buffer_load_b128 (a half-wave reads
// 512 contiguous bytes).  All of a wave's activation loads are issued before its first MFMA.  A fragment = 16 bytes
// of the packed weight image (<= 16 KB, L1/L2 resident), straight from global.  Epilogue as the k x k kernel.
template <class E, int CIN, int COUT>
__global__ __launch_bounds__(256) void conv1x1_c8_kernel(const ConvC8Params p) {
  typedef typename E::vec8 vec8;
  constexpr int NKS = CIN / 16, CT = COUT / 32;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const long HW = (long)p.H * p.W;
  const unsigned HW16 = 16u * (unsigned)HW;
  const int b = blockIdx.y;
  const long pix0 = ((long)blockIdx.x * 4 + wave) * 64;     // first pixel of this wave's group
  if (pix0 >= HW) return;                                    // wave-uniform
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.x + (long)b * p.x_img + p.x_base), 0, (int)((unsigned)(CIN / 8) * HW16), C8_RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.y + (long)b * p.y_img + p.y_base), 0, (int)((unsigned)(COUT / 8) * HW16), C8_RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.res ? p.res + (long)b * p.r_img + p.r_base : p.x), 0, (int)((unsigned)(COUT / 8) * HW16), C8_RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)(CIN * COUT * 2), C8_RSRC_FLAGS);
  const bool relu = p.flags & CODON_CONV_RELU;
  const bool rg = p.res != nullptr;
  const bool addr = (p.flags & CODON_CONV_ADD_RESIDUAL) && rg;
  const bool accum = p.flags & CODON_CONV_ACCUM_OUT;
  const bool mask = (p.flags & CODON_CONV_MASK_RELU) && rg;
  const bool msum = (p.flags & CODON_CONV_MASK_SUM) != 0;

  unsigned vo[2];                                            // tile i: pixel pix0 + 32 i + l31, plane `half`
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const long px = pix0 + i * 32 + l31;
    vo[i] = px < HW ? (unsigned)half * HW16 + 16u * (unsigned)px : C8_OOB;
  }
  u32x4 dd[NKS][2];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
    for (int i = 0; i < 2; ++i) dd[ks][i] = c8_ld(xrsrc, vo[i], (unsigned)(2 * ks) * HW16);

  // the COUT rows are processed 64 at a time; the residual / mask / accumulate vectors of a half are requested before
  // its MFMAs, so a wave makes one memory round trip per half
  constexpr int CTB = CT > 2 ? 2 : CT;
  auto cplane = [&](int t, int g) { return (unsigned)(t * 4 + 2 * g) * HW16; };
  auto half_pass = [&](auto has_r, auto has_acc, const int t0) {
    constexpr bool HR = decltype(has_r)::value, HA = decltype(has_acc)::value;
    u32x4 rm[2][CTB][2], am[2][CTB][2];
    if constexpr (HR) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < CTB; ++t)
#pragma unroll
          for (int g = 0; g < 2; ++g) rm[i][t][g] = c8_ld(rrsrc, vo[i], cplane(t0 + t, g));
    }
    if constexpr (HA) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < CTB; ++t)
#pragma unroll
          for (int g = 0; g < 2; ++g) am[i][t][g] = c8_ld(yrsrc, vo[i], cplane(t0 + t, g));
    }
    f32x16 acc[2][CTB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int t = 0; t < CTB; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      vec8 a[CTB];
#pragma unroll
      for (int t = 0; t < CTB; ++t) {
        const u32x4 v = c8_ld(wrsrc, (unsigned)((half * COUT + (t0 + t) * 32 + l31) * 16), (unsigned)(ks * 2 * COUT * 16));
        a[t] = *reinterpret_cast<const vec8*>(&v);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < CTB; ++t)
          acc[i][t] = E::mfma(a[t], *reinterpret_cast<const vec8*>(&dd[ks][i]), acc[i][t]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int t = 0; t < CTB; ++t)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          float v8[8], r8[8], a8[8];
          if constexpr (HR) c8_unpack<E>(rm[i][t][g], r8);
          if constexpr (HA) c8_unpack<E>(am[i][t][g], a8);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float v = acc[i][t][8 * g + j];
            if (relu) v = fmaxf(v, 0.f);
            if constexpr (HR) {
              if (addr) v += r8[j];
              if (mask && !msum) v = r8[j] > 0.f ? v : 0.f;
            }
            if constexpr (HA) v += a8[j];
            if constexpr (HR) {
              if (mask && msum) v = r8[j] > 0.f ? v : 0.f;
            }
            v8[j] = v;
          }
          c8_st(c8_pack<E>(v8), yrsrc, vo[i], cplane(t0 + t, g));
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
}

// ---- weight packers -------------------------------------------------------------------------------------------------
// OIHW fp32 -> 16-bit packed [chunk][dy][dx][cb (2)][cout position][8 ch]; position p of a 32-row tile holds cout
// tile*32 + swap23(p & 31).  DGRAD: flipped taps, in/out swapped (the same kernel then computes dL/dx).
template <class E>
__global__ void pack_weight_c8_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int cout, int cin,
                                      int ks, int dgrad) {
  const int kin = dgrad ? cout : cin, kout = dgrad ? cin : cout;
  const long n = (long)kin * kout * ks * ks;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long t = i;
    const int j = t % 8; t /= 8;
    const int op = t % kout; t /= kout;
    const int cb = t % 2; t /= 2;
    const int dx = t % ks; t /= ks;
    const int dy = t % ks; t /= ks;
    const int chunk = (int)t;
    const int o = (op & ~31) | swap23(op & 31);
    const int ci = chunk * 16 + cb * 8 + j;
    float v;
    if (!dgrad) v = w[(((long)o * cin + ci) * ks + dy) * ks + dx];
    else v = w[(((long)ci * cin + o) * ks + (ks - 1 - dy)) * ks + (ks - 1 - dx)];
    out[i] = (unsigned short)(E::pack2(v, 0.f) & 0xffffu);
  }
}

// OIHW (64,128,1,1) fp32 -> the chained-1x1 A-operand image [t2][t][g][lane][8]
template <class E>
__global__ void pack_chain1x1_c8_kernel(const float* __restrict__ w, unsigned short* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // 64 * 128 values
  if (i >= 64 * 128) return;
  const int j = i & 7, lane = (i >> 3) & 63, g = (i >> 9) & 1, t = (i >> 10) & 3, t2 = i >> 12;
  const int co2 = t2 * 32 + swap23(lane & 31);
  const int c = t * 32 + 16 * g + 8 * (lane >> 5) + j;
  out[i] = (unsigned short)(E::pack2(w[co2 * 128 + c], 0.f) & 0xffffu);
}

int pack_chain1x1_16(const float* w, void* out, int dtype, hipStream_t stream) {
  if (dtype == CODON_F16)
    hipLaunchKernelGGL(pack_chain1x1_c8_kernel<C8F16>, dim3(32), dim3(256), 0, stream, w, (unsigned short*)out);
  else
    hipLaunchKernelGGL(pack_chain1x1_c8_kernel<C8Bf16>, dim3(32), dim3(256), 0, stream, w, (unsigned short*)out);
  return check_launch("pack_chain1x1_c8_kernel");
}

int pack_weight_bf16(const float* w, void* out, int cout, int cin, int ks, int mode, int dtype, hipStream_t stream) {
  const long n = (long)cout * cin * ks * ks;
  const int blocks = (int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
  const int kout = mode == CODON_PACK_DGRAD ? cin : cout;
  CODON_REQUIRE(kout % 32 == 0, CODON_ERR_UNSUPPORTED, "conv_pack_weight: 16-bit packing needs 32-row output tiles (got %d)", kout);
  if (dtype == CODON_F16)
    hipLaunchKernelGGL(pack_weight_c8_kernel<C8F16>, dim3(blocks), dim3(256), 0, stream, w, (unsigned short*)out, cout,
                       cin, ks, mode == CODON_PACK_DGRAD ? 1 : 0);
  else
    hipLaunchKernelGGL(pack_weight_c8_kernel<C8Bf16>, dim3(blocks), dim3(256), 0, stream, w, (unsigned short*)out, cout,
                       cin, ks, mode == CODON_PACK_DGRAD ? 1 : 0);
  return check_launch("pack_weight_c8_kernel");
}

// ---- launchers ------------------------------------------------------------------------------------------------------
static bool c8_desc_ok(const codon_conv_desc* d, bool with_res) {
  if (!c8_slice_ok(d->x_ctotal, d->x_coff, d->cin) || !c8_slice_ok(d->y_ctotal, d->y_coff, d->cout)) return false;
  if (with_res && !c8_slice_ok(d->r_ctotal, d->r_coff, d->cout)) return false;
  return true;
}

static void c8_fill(ConvC8Params& p, const codon_conv_desc* d, const void* x, const void* w, void* y, const void* res) {
  const long HW = (long)d->height * d->width;
  p.x = (const uint4*)x; p.w = (const uint4*)w; p.y = (uint4*)y; p.res = (const uint4*)res;
  p.H = d->height; p.W = d->width;
  p.x_img = (d->x_ctotal / 8) * HW; p.y_img = (d->y_ctotal / 8) * HW; p.r_img = (d->r_ctotal / 8) * HW;
  p.x_base = (d->x_coff / 8) * HW; p.y_base = (d->y_coff / 8) * HW; p.r_base = (d->r_coff / 8) * HW;
  p.tiles_x = p.tiles_y = p.nblk = 0;
  p.flags = d->flags;
  p.w2 = nullptr; p.y2 = nullptr; p.y2_img = p.y2_base = 0;
  p.st_pool = nullptr; p.st_part = nullptr; p.st_choff = 0;
  p.gin = nullptr; p.g_img = p.g_base = 0; p.gch = nullptr; p.gsp = nullptr;
  p.gout = nullptr; p.go_img = p.go_base = 0;
}

template <class E, int CIN, int COUT>
static int launch_conv1x1_c8(const codon_conv_desc* d, const void* x, const void* w, void* y, const void* res,
                             hipStream_t stream) {
  ConvC8Params p;
  c8_fill(p, d, x, w, y, res);
  const long HW = (long)d->height * d->width;
  CODON_REQUIRE(d->batch <= 65535, CODON_ERR_UNSUPPORTED, "conv2d_fwd: batch %d > 65535", d->batch);
  const unsigned gx = (unsigned)((HW + 255) / 256);
  hipLaunchKernelGGL((conv1x1_c8_kernel<E, CIN, COUT>), dim3(gx, d->batch), dim3(256), 0, stream, p);
  return check_launch("conv1x1_c8_kernel");
}

// Workgroups of `kernel` that are resident on the whole chip at once (CUs x occupancy), per device; 0 if unknown.
template <class K>
static int c8_resident_blocks(K kernel, int threads) {
  constexpr int MAXDEV = 64;
  static int cached[MAXDEV];
  static bool init = [] { for (int i = 0; i < MAXDEV; ++i) cached[i] = -1; return true; }();
  (void)init;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) { (void)hipGetLastError(); return 0; }
  int v = __atomic_load_n(&cached[dev], __ATOMIC_ACQUIRE);
  if (v < 0) {
    int ncu = 0, occ = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kernel, threads, 0) != hipSuccess)
      ncu = occ = 0;
    (void)hipGetLastError();
    v = ncu * occ;
    __atomic_store_n(&cached[dev], v, __ATOMIC_RELEASE);
  }
  return v;
}

// The resident-filter conv3x3 64->64 is a PERSIST launch: one resident generation of workgroups, each walking at least
// C8_RESIDENT_MIN_TILES tiles; smaller problems keep one tile per workgroup (nothing to amortise, and the small-grid latency
// path wants many short workgroups).  (The tile loop with STAGED weights, for the plain 3x3 and 5x5 64->64 convs, was
// measured as a loss and is gone: tools/probes/conv_c8_persist_staged_experiment.patch, profiles/HISTORY.md.)
constexpr int C8_RESIDENT_MIN_TILES = 8;
template <class E, int KS, int CIN, int COUT, bool FUSE, int NW, bool GATE>
static int launch_single_c8(const void* pv, hipStream_t stream) {
  const ConvC8Params& p = *static_cast<const ConvC8Params*>(pv);
  hipLaunchKernelGGL((conv_c8_kernel<E, KS, CIN, COUT, FUSE, NW, GATE>), dim3((unsigned)p.nblk), dim3(64 * NW), 0, stream, p);
  return check_launch("conv_c8_kernel");
}
template <class E, int KS, int CIN, int COUT, bool FUSE, int NW, bool GATE>
static int launch_pair_c8(const void* av, const void* bv, hipStream_t stream) {
  ConvC8Pair pp;
  pp.a = *static_cast<const ConvC8Params*>(av);
  pp.b = *static_cast<const ConvC8Params*>(bv);
  hipLaunchKernelGGL((conv_c8_pair_kernel<E, KS, CIN, COUT, FUSE, NW, GATE>), dim3(2u * (unsigned)pp.a.nblk), dim3(64 * NW), 0,
                     stream, pp);
  return check_launch("conv_c8_pair_kernel");
}

template <class E>
static int launch_mix53_c8(const void* five, const void* three, hipStream_t stream) {
  ConvC8Pair pp;
  pp.a = *static_cast<const ConvC8Params*>(five);
  pp.b = *static_cast<const ConvC8Params*>(three);
  hipLaunchKernelGGL((conv_c8_mix53_kernel<E>), dim3((unsigned)pp.a.nblk + (unsigned)pp.b.nblk), dim3(256), 0, stream, pp);
  return check_launch("conv_c8_mix53_kernel");
}

template <class E, int KS, int CIN, int COUT, bool FUSE, bool GATE = false>
static int launch_conv_c8(ConvC8Params& p, const codon_conv_desc* d, hipStream_t stream) {
  constexpr int NW = ConvC8Nw<KS, COUT>::value;
  constexpr int TH = NW * ConvC8Pseg<KS, COUT>::value;
  p.tiles_x = (d->width + 31) / 32;
  p.tiles_y = (d->height + TH - 1) / TH;
  const long nblk = (long)p.tiles_x * p.tiles_y * d->batch;
  CODON_REQUIRE(nblk < (1L << 31), CODON_ERR_UNSUPPORTED, "conv2d_fwd: grid too large (%ld blocks)", nblk);
  p.nblk = (int)nblk;
  if constexpr (KS == 3 && CIN == 64 && COUT == 64 && !FUSE && !GATE && C8_DMA) {
    // resident-filter persistent form: 16 waves, 32 x 32 tiles, one workgroup per CU
    constexpr int NWR = 16, THR = NWR * ConvC8Pseg<KS, COUT>::value;
    constexpr bool RW = true;
    const int res = c8_resident_blocks(conv_c8_kernel<E, KS, CIN, COUT, FUSE, NWR, GATE, true, RW>, 64 * NWR);
    const int tyr = (d->height + THR - 1) / THR;
    const long nblkr = (long)p.tiles_x * tyr * d->batch;
    if (res > 0 && nblkr >= (long)res * C8_RESIDENT_MIN_TILES) {
      ConvC8Params pr = p;
      pr.tiles_y = tyr;
      pr.nblk = (int)nblkr;
      hipLaunchKernelGGL((conv_c8_kernel<E, KS, CIN, COUT, FUSE, NWR, GATE, true, RW>), dim3((unsigned)res), dim3(64 * NWR), 0,
                         stream, pr);
      return check_launch("conv_c8_kernel<resident>");
    }
  }
  // inside codon_conv_pair_begin / _end the launch is held back: pair_end issues two held launches of the same kernel variant
  // on the same grid as ONE, anything else one by one in the order they came
  // (the plain 64 -> 64 convs can also leave as a mix53 grid with their sibling of the other filter size)
  constexpr bool MIXABLE = !FUSE && !GATE && CIN == 64 && COUT == 64 && NW == 4 && (KS == 5 || KS == 3);
  constexpr int k5 = std::is_same<E, C8F16>::value ? MIX_C8F16_5 : MIX_C8BF16_5;
  constexpr int kind = !MIXABLE ? MIX_NONE : KS == 5 ? k5 : k5 + 1;
  if (const int held = pair_hold(p, &launch_single_c8<E, KS, CIN, COUT, FUSE, NW, GATE>,
                                 &launch_pair_c8<E, KS, CIN, COUT, FUSE, NW, GATE>, stream, kind,
                                 (MIXABLE && KS == 5) ? &launch_mix53_c8<E> : nullptr))
    return held < 0 ? held : CODON_OK;
  return launch_single_c8<E, KS, CIN, COUT, FUSE, NW, GATE>(&p, stream);
}

template <class E>
static int conv2d_fwd_c8(const codon_conv_desc* d, const void* x, const void* w, void* y, const void* res,
                         hipStream_t stream) {
  ConvC8Params p;
  c8_fill(p, d, x, w, y, res);
  const int key = d->ksize * 1000000 + d->cin * 1000 + d->cout;
  switch (key) {
    case 5128128: return launch_conv_c8<E, 5, 128, 128, false>(p, d, stream);
    case 5064064: return launch_conv_c8<E, 5, 64, 64, false>(p, d, stream);
    case 3064064: return launch_conv_c8<E, 3, 64, 64, false>(p, d, stream);
    case 3128064: return launch_conv_c8<E, 3, 128, 64, false>(p, d, stream);
    case 3064128: return launch_conv_c8<E, 3, 64, 128, false>(p, d, stream);
    case 1128064: return launch_conv1x1_c8<E, 128, 64>(d, x, w, y, res, stream);
    case 1064128: return launch_conv1x1_c8<E, 64, 128>(d, x, w, y, res, stream);
    default:
      set_error("conv2d_fwd: no 16-bit kernel for k=%d cin=%d cout=%d", d->ksize, d->cin, d->cout);
      return CODON_ERR_UNSUPPORTED;
  }
}

int conv2d_fwd_bf16(const codon_conv_desc* d, const void* x, const void* w, void* y, const void* res,
                    hipStream_t stream) {
  const bool with_res = res && (d->flags & (CODON_CONV_ADD_RESIDUAL | CODON_CONV_MASK_RELU));
  CODON_REQUIRE(c8_desc_ok(d, with_res), CODON_ERR_BAD_ARG,
                "conv2d_fwd: 16-bit tensors are channel-blocked: ctotal / coff / channels must be multiples of 8");
  const long HW = (long)d->height * d->width;
  CODON_REQUIRE(HW * 2 * 128 < (long)C8_OOB, CODON_ERR_UNSUPPORTED,
                "conv2d_fwd: %dx%d image: 128 channels exceed the 4 GiB buffer-descriptor range", d->height, d->width);
  return d->dtype == CODON_F16 ? conv2d_fwd_c8<C8F16>(d, x, w, y, with_res ? res : nullptr, stream)
                               : conv2d_fwd_c8<C8Bf16>(d, x, w, y, with_res ? res : nullptr, stream);
}

// y (+)= conv5x5(x) (64 -> 64) and, in the same epilogue, sum += y
int conv2d_sum_into_16(const codon_conv_desc* d, const void* x, const void* w, void* y, void* sum, hipStream_t stream) {
  CODON_REQUIRE(d->ksize == 5 && d->cin == 64 && d->cout == 64, CODON_ERR_UNSUPPORTED,
                "conv2d_sum_into_fwd: the 16-bit conv5x5 64->64 only (got k=%d %d->%d)", d->ksize, d->cin, d->cout);
  CODON_REQUIRE((d->flags & ~CODON_CONV_ACCUM_OUT) == 0, CODON_ERR_BAD_ARG, "conv2d_sum_into_fwd: only ACCUM_OUT applies");
  CODON_REQUIRE(c8_desc_ok(d, true), CODON_ERR_BAD_ARG,
                "conv2d_sum_into_fwd: 16-bit tensors are channel-blocked: ctotal / coff / channels must be multiples of 8");
  const long HW = (long)d->height * d->width;
  CODON_REQUIRE(HW * 2 * 128 < (long)C8_OOB, CODON_ERR_UNSUPPORTED,
                "conv2d_sum_into_fwd: %dx%d image: 128 channels exceed the 4 GiB buffer-descriptor range", d->height, d->width);
  ConvC8Params p;
  c8_fill(p, d, x, w, y, sum);
  p.flags |= CONV_C8_SUM_INTO;
  // the plain one-tile-per-workgroup kernel (the variant is not compiled into the tile-loop forms)
  constexpr int NW = ConvC8Nw<5, 64>::value, TH = NW * ConvC8Pseg<5, 64>::value;
  p.tiles_x = (d->width + 31) / 32;
  p.tiles_y = (d->height + TH - 1) / TH;
  const long nblk = (long)p.tiles_x * p.tiles_y * d->batch;
  CODON_REQUIRE(nblk < (1L << 31), CODON_ERR_UNSUPPORTED, "conv2d_sum_into_fwd: grid too large (%ld blocks)", nblk);
  p.nblk = (int)nblk;
  if (d->dtype == CODON_F16)
    hipLaunchKernelGGL((conv_c8_kernel<C8F16, 5, 64, 64, false, NW, false>), dim3((unsigned)nblk), dim3(64 * NW), 0, stream, p);
  else
    hipLaunchKernelGGL((conv_c8_kernel<C8Bf16, 5, 64, 64, false, NW, false>), dim3((unsigned)nblk), dim3(64 * NW), 0, stream, p);
  return check_launch("conv_c8_kernel<sum into>");
}

int conv_chain1x1_fwd_16(const codon_conv_desc* d, const void* x, const void* w, void* y, const void* w_chain,
                         const codon_tensor* out, const codon_tensor* res, float* st_pool, float* st_part, int st_choff,
                         hipStream_t stream) {
  CODON_REQUIRE(d->ksize == 5 && d->cin == 128 && d->cout == 128, CODON_ERR_UNSUPPORTED,
                "conv_chain1x1_fwd: 16-bit kernel is conv5x5 128->128 + 1x1 128->64 (got k=%d %d->%d)", d->ksize, d->cin, d->cout);
  CODON_REQUIRE(c8_slice_ok(d->x_ctotal, d->x_coff, 128) && (!y || c8_slice_ok(d->y_ctotal, d->y_coff, 128)) &&
                    c8_slice_ok(out->ctotal, out->coff, 64) && (!res || c8_slice_ok(res->ctotal, res->coff, 64)),
                CODON_ERR_BAD_ARG, "conv_chain1x1_fwd: 16-bit tensors are channel-blocked: ctotal / coff multiples of 8");
  const long HW = (long)d->height * d->width;
  CODON_REQUIRE(HW * 2 * 128 < (long)C8_OOB, CODON_ERR_UNSUPPORTED,
                "conv_chain1x1_fwd: %dx%d image: 128 channels exceed the 4 GiB buffer-descriptor range", d->height, d->width);
  ConvC8Params p;
  c8_fill(p, d, x, w, y, res ? res->data : nullptr);
  if (!y) { p.y_img = p.y_base = 0; }
  p.r_img = res ? (res->ctotal / 8) * HW : 0;
  p.r_base = res ? (res->coff / 8) * HW : 0;
  p.w2 = (const uint4*)w_chain; p.y2 = (uint4*)out->data;
  p.y2_img = (out->ctotal / 8) * HW; p.y2_base = (out->coff / 8) * HW;
  p.st_pool = st_pool; p.st_part = st_part; p.st_choff = st_choff;
  return d->dtype == CODON_F16 ? launch_conv_c8<C8F16, 5, 128, 128, true>(p, d, stream)
                               : launch_conv_c8<C8Bf16, 5, 128, 128, true>(p, d, stream);
}

// y = conv(pre * (ch * sp) + inputs) [ReLU]: the gate-apply formed while the halo tile is staged (inference)
template <class E>
static int conv2d_gated_c8(ConvC8Params& p, const codon_conv_desc* d, hipStream_t stream) {
  const int key = d->ksize * 1000000 + d->cin * 1000 + d->cout;
  switch (key) {
    case 5064064: return launch_conv_c8<E, 5, 64, 64, false, true>(p, d, stream);
    case 3064064: return launch_conv_c8<E, 3, 64, 64, false, true>(p, d, stream);
    case 3128064: return launch_conv_c8<E, 3, 128, 64, false, true>(p, d, stream);
    default:
      set_error("conv2d_gated_fwd: no 16-bit kernel for k=%d cin=%d cout=%d", d->ksize, d->cin, d->cout);
      return CODON_ERR_UNSUPPORTED;
  }
}

int conv2d_gated_fwd_16(const codon_conv_desc* d, const void* pre, const codon_tensor* inputs, const float* ch,
                        const float* sp, const void* w, void* y, const codon_tensor* gated_out, hipStream_t stream) {
  CODON_REQUIRE(c8_desc_ok(d, false) && c8_slice_ok(inputs->ctotal, inputs->coff, d->cin), CODON_ERR_BAD_ARG,
                "conv2d_gated_fwd: 16-bit tensors are channel-blocked: ctotal / coff / channels multiples of 8");
  const long HW = (long)d->height * d->width;
  CODON_REQUIRE(HW * 2 * 128 < (long)C8_OOB, CODON_ERR_UNSUPPORTED,
                "conv2d_gated_fwd: %dx%d image: 128 channels exceed the 4 GiB buffer-descriptor range", d->height, d->width);
  ConvC8Params p;
  c8_fill(p, d, pre, w, y, nullptr);
  p.gin = (const uint4*)inputs->data; p.g_img = (inputs->ctotal / 8) * HW; p.g_base = (inputs->coff / 8) * HW;
  p.gch = ch; p.gsp = sp;
  if (gated_out != nullptr) {
    CODON_REQUIRE(gated_out->data != nullptr && c8_slice_ok(gated_out->ctotal, gated_out->coff, d->cin), CODON_ERR_BAD_ARG,
                  "conv2d_gated_emit_fwd: gated_out slice [%d,%d) of %d channels", gated_out->coff,
                  gated_out->coff + d->cin, gated_out->ctotal);
    p.gout = (uint4*)gated_out->data; p.go_img = (gated_out->ctotal / 8) * HW; p.go_base = (gated_out->coff / 8) * HW;
  }
  return d->dtype == CODON_F16 ? conv2d_gated_c8<C8F16>(p, d, stream) : conv2d_gated_c8<C8Bf16>(p, d, stream);
}

// tiles of the fused-statistics partials: the conv5x5 128->128 kernel's 8 x 32 (NW * PSEG rows) pixel tiles.
// (Round 6 measured the alternative the round-5 review asked for -- 4 x 32 tiles for launches of a few rounds of workgroups,
// with the statistics per 4-row strip so that both tilings give the same bits: a 4 x 32 tile costs 0.62 of an 8 x 32 one,
// not 0.5 -- its weight staging is L2-bound, 77 B/clk per CU against the 64 a CU draws -- so it was slower at every image
// height (one 370 x 463 image as a pair 219.7 -> 248.3 us), and the strips doubled the one-launch gate's fold (20 -> 31 us
// per block).  Every 16-bit launch keeps the 8 x 32 tile at any batch, so per-tile partials are batch-invariant as they are.
// tools/probes/c8_4x32_tiles_strip_statistics_experiment.patch, tools/probes/c8_tile_ab.py, profiles/HISTORY.md.)
int cac_fused_tiles(int H, int W) {
  constexpr int TH = ConvC8Nw<5, 128>::value * ConvC8Pseg<5, 128>::value;
  return ((W + 31) / 32) * ((H + TH - 1) / TH);
}

}  // namespace codon
