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
  // FUSE only: the chained 1x1 (128 -> 64) applied to the tile while it is still in the accumulators
  const float* w2;  // [t2][t][lane][16]: W1[t2*32 + (lane&31)][t*32 + (r&3) + 8*(r>>2) + 4*(lane>>5)]
  float* y2;
  long y2_img, y2_base;
};

template <int KS, int CIN>
struct ConvCfg {
  static constexpr int CK = (KS == 1) ? 16 : 8;
  static constexpr int NCHUNK = CIN / CK;
};

template <int KS, int CIN, int COUT, int PSEG, bool FUSE = false>
__global__ __launch_bounds__(256, 2) void conv_mfma_f32_kernel(const ConvParams p) {
  constexpr int PAD = KS / 2;
  constexpr int TW = 32, TH = 4 * PSEG;
  constexpr int XR = TH + KS - 1, XQ = TW + KS - 1;
  constexpr int CK = ConvCfg<KS, CIN>::CK;
  constexpr int NCHUNK = CIN / CK;
  constexpr int XS = CK * XR * XQ;    // floats per input buffer
  constexpr int WS = CK * KS * COUT;  // floats per weight stage
  constexpr int CT = COUT / 32;
  constexpr int NST = NCHUNK * KS;
  constexpr int XE = (XS + 255) / 256;  // x elements per thread per chunk
  constexpr int W4 = WS / 4;            // float4 per weight stage
  constexpr int WE = (W4 + 255) / 256;  // float4 per thread per stage
  static_assert(WS % 4 == 0, "weight stage must be whole float4s");

  __shared__ __attribute__((aligned(16))) float lds[2 * XS + 2 * WS];
  float* const xs0 = lds;
  float* const ws0 = lds + 2 * XS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31;
  const int half = lane >> 5;

  unsigned bid = xcd_remap(blockIdx.x, (unsigned)p.nblk);
  const int tx = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int ty = bid % p.tiles_y;
  const int b = bid / p.tiles_y;
  const int tx0 = tx * TW, ty0 = ty * TH;
  const int H = p.H, W = p.W;
  const long HW = (long)H * W;

  const float* __restrict__ xg = p.x + (long)b * p.x_img + p.x_base;
  const float4* __restrict__ wg = reinterpret_cast<const float4*>(p.w);

  // per-thread gather plan for the input halo tile (same for every chunk)
  int xoff[XE];
  unsigned xmask = 0;
#pragma unroll
  for (int k = 0; k < XE; ++k) {
    const int e = tid + k * 256;
    const int c = e / (XR * XQ);
    const int rem = e - c * (XR * XQ);
    const int r = rem / XQ;
    const int q = rem - r * XQ;
    const int gy = ty0 + r - PAD, gx = tx0 + q - PAD;
    const bool ok = (e < XS) && gy >= 0 && gy < H && gx >= 0 && gx < W;
    xoff[k] = ok ? (int)(c * HW + (long)gy * W + gx) : 0;
    xmask |= ok ? (1u << k) : 0u;
  }

  float xr[XE];
  float4 wr[WE];

// staging steps as macros (not lambdas): keeps xr/wr in registers (no alloca left for scratch)
#define LOAD_X(chunk_)                                                             \
  {                                                                                \
    const float* src_ = xg + (long)(chunk_) * CK * HW;                             \
    _Pragma("unroll") for (int k = 0; k < XE; ++k)                                 \
        xr[k] = src_[xoff[k]];  /* unconditional (xoff = 0, in bounds, when masked); mask applied at STORE_X */ \
  }
#define STORE_X(buf_)                                                              \
  {                                                                                \
    float* dst_ = xs0 + (buf_) * XS;                                               \
    _Pragma("unroll") for (int k = 0; k < XE; ++k) {                               \
      const int e_ = tid + k * 256;                                                \
      if (XS % 256 == 0 || e_ < XS) dst_[e_] = ((xmask >> k) & 1u) ? xr[k] : 0.f;  \
    }                                                                              \
  }
#define LOAD_W(stage_)                                                             \
  {                                                                                \
    const float4* src_ = wg + (long)(stage_) * W4;                                 \
    _Pragma("unroll") for (int k = 0; k < WE; ++k)                                 \
        wr[k] = (W4 % 256 == 0 || tid + k * 256 < W4) ? src_[tid + k * 256]        \
                                                      : make_float4(0, 0, 0, 0);   \
  }
#define STORE_W(buf_)                                                              \
  {                                                                                \
    float4* dst_ = reinterpret_cast<float4*>(ws0 + (buf_) * WS);                   \
    _Pragma("unroll") for (int k = 0; k < WE; ++k)                                 \
        if (W4 % 256 == 0 || tid + k * 256 < W4) dst_[tid + k * 256] = wr[k];      \
  }

  f32x16 acc[PSEG][CT];
#pragma unroll
  for (int i = 0; i < PSEG; ++i)
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;

  // prologue
  LOAD_X(0);
  LOAD_W(0);
  STORE_X(0);
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

    const float* xb = xs0 + (chunk & 1) * XS + (half * XR + wave * PSEG + dy) * XQ + l31;
    const float* wb = ws0 + (s & 1) * WS + half * (KS * COUT) + l31;
#pragma unroll
    for (int dx = 0; dx < KS; ++dx) {
#pragma unroll
      for (int cp = 0; cp < CK / 2; ++cp) {
        float a[CT], bv[PSEG];
#pragma unroll
        for (int t = 0; t < CT; ++t) a[t] = wb[((2 * cp) * KS + dx) * COUT + t * 32];
#pragma unroll
        for (int i = 0; i < PSEG; ++i) bv[i] = xb[(2 * cp) * (XR * XQ) + i * XQ + dx];
#pragma unroll
        for (int i = 0; i < PSEG; ++i)
#pragma unroll
          for (int t = 0; t < CT; ++t)
            acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], bv[i], acc[i][t], 0, 0, 0);
      }
    }

    if (has_next) STORE_W((s + 1) & 1);
    if (next_chunk) STORE_X((chunk + 1) & 1);
    __syncthreads();
  }

#undef LOAD_X
#undef STORE_X
#undef LOAD_W
#undef STORE_W

  // epilogue: ReLU / residual / accumulate, coalesced NCHW stores.  Flag tests hoisted into four wave-uniform
  // variants: inside a variant the residual / accumulate loads of a tile are unconditional and issued back to
  // back (a per-element `if (flag) v += rg[..]` compiles to a load + vmcnt(0) per element).
  const int gx = tx0 + l31;
  if constexpr (FUSE) {
    // Chained 1x1: the D layout of the 32x32 MFMA (lane = pixel l&31, register r = channel (r&3)+8(r>>2)+4(l>>5))
    // IS a B operand of the next MFMA for the channel pair {c, c+4}: lanes 0-31 carry k = 0, lanes 32-63 k = 1 of
    // the same 32 pixels.  So Y2[co2][pix] = sum_c W1[co2][c] relu(acc)[c][pix] runs straight from the accumulator
    // registers -- 64 MFMAs per 32 output channels and pixel row, no LDS round trip, and the 128-channel
    // intermediate never has to reach HBM (p.y == nullptr).  W1 is pre-permuted to that k order by the packer.
    static_assert(!FUSE || COUT == 128, "chained 1x1 is 128 -> 64");
    const bool relu = p.flags & CODON_CONV_RELU;
#pragma unroll
    for (int i = 0; i < PSEG; ++i)
#pragma unroll
      for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][t][r] = relu ? fmaxf(acc[i][t][r], 0.f) : acc[i][t][r];
    if (p.y && gx < W) {
      float* __restrict__ yg = p.y + (long)b * p.y_img + p.y_base;
#pragma unroll
      for (int i = 0; i < PSEG; ++i) {
        const int gy = ty0 + wave * PSEG + i;
        if (gy < H) {
#pragma unroll
          for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              yg[(t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * HW + (long)gy * W + gx] = acc[i][t][r];
        }
      }
    }
    const float4* __restrict__ w2 = reinterpret_cast<const float4*>(p.w2) + lane * 4;
    float* __restrict__ y2 = p.y2 + (long)b * p.y2_img + p.y2_base;
    const float* __restrict__ rg = p.res ? p.res + (long)b * p.r_img + p.r_base : nullptr;
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
        for (int q = 0; q < 4; ++q) a4[q] = w2[((t2 * CT + t) * 64) * 4 + q];
        const float* a = reinterpret_cast<const float*>(a4);
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int i = 0; i < PSEG; ++i)
            d[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], acc[i][t][r], d[i], 0, 0, 0);
      }
      if (gx < W) {
#pragma unroll
        for (int i = 0; i < PSEG; ++i) {
          const int gy = ty0 + wave * PSEG + i;
          if (gy < H) {
            const long pix = (long)gy * W + gx;
            if (rg) {
              float rv[16];
#pragma unroll
              for (int r = 0; r < 16; ++r) rv[r] = rg[(t2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * HW + pix];
#pragma unroll
              for (int r = 0; r < 16; ++r) y2[(t2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * HW + pix] = d[i][r] + rv[r];
            } else {
#pragma unroll
              for (int r = 0; r < 16; ++r) y2[(t2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * HW + pix] = d[i][r];
            }
          }
        }
      }
    }
    return;
  }
  if (gx < W) {
    float* __restrict__ yg = p.y + (long)b * p.y_img + p.y_base;
    const float* __restrict__ rg = p.res ? p.res + (long)b * p.r_img + p.r_base : nullptr;
    const bool relu = p.flags & CODON_CONV_RELU;
    const bool addr = (p.flags & CODON_CONV_ADD_RESIDUAL) && rg;
    const bool accum = p.flags & CODON_CONV_ACCUM_OUT;
    const bool mask = (p.flags & CODON_CONV_MASK_RELU) && rg;
    auto epi = [&](auto has_r, auto has_acc) {
#pragma unroll
      for (int i = 0; i < PSEG; ++i) {
        const int gy = ty0 + wave * PSEG + i;
        if (gy < H) {
          const long pix = (long)gy * W + gx;
#pragma unroll
          for (int t = 0; t < CT; ++t) {
            float rv[16], av[16];
            if constexpr (decltype(has_r)::value) {
#pragma unroll
              for (int r = 0; r < 16; ++r) rv[r] = rg[(t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * HW + pix];
            }
            if constexpr (decltype(has_acc)::value) {
#pragma unroll
              for (int r = 0; r < 16; ++r) av[r] = yg[(t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * HW + pix];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int co = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
              float v = acc[i][t][r];
              if (relu) v = fmaxf(v, 0.f);
              if constexpr (decltype(has_r)::value) {
                if (addr) v += rv[r];
                if (mask) v = rv[r] > 0.f ? v : 0.f;
              }
              if constexpr (decltype(has_acc)::value) v += av[r];
              yg[co * HW + pix] = v;
            }
          }
        }
      }
    };
    const bool has_r = addr || mask;
    if (has_r && accum) epi(std::true_type{}, std::true_type{});
    else if (has_r) epi(std::true_type{}, std::false_type{});
    else if (accum) epi(std::false_type{}, std::true_type{});
    else epi(std::false_type{}, std::false_type{});
  }
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

template <int KS, int CIN, int COUT, int PSEG>
static int launch_conv(const codon_conv_desc* d, const float* x, const float* w, float* y,
                       const float* res, hipStream_t stream) {
  constexpr int TH = 4 * PSEG;
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
  p.nblk = (int)nblk;
  p.flags = d->flags;
  p.w2 = nullptr; p.y2 = nullptr; p.y2_img = p.y2_base = 0;
  hipLaunchKernelGGL((conv_mfma_f32_kernel<KS, CIN, COUT, PSEG>), dim3((unsigned)nblk), dim3(256), 0, stream, p);
  return check_launch("conv_mfma_f32_kernel");
}

int conv_ck(int ks) { return ks == 1 ? 16 : 8; }

// d: the 5x5 128 -> 128 conv (y nullable); out / res: 64-channel slices of the chained 1x1
int conv_chain1x1_fwd_f32(const codon_conv_desc* d, const float* x, const float* w, float* y, const float* w_chain,
                          const codon_tensor* out, const codon_tensor* res, hipStream_t stream) {
  CODON_REQUIRE(d->ksize == 5 && d->cin == 128 && d->cout == 128, CODON_ERR_UNSUPPORTED,
                "conv_chain1x1_fwd: f32 kernel is conv5x5 128->128 + 1x1 128->64 (got k=%d %d->%d)", d->ksize, d->cin, d->cout);
  constexpr int TH = 8;
  ConvParams p;
  p.x = x; p.w = w; p.y = y; p.res = res ? (const float*)res->data : nullptr;
  p.H = d->height; p.W = d->width;
  const long HW = (long)d->height * d->width;
  p.x_img = d->x_ctotal * HW; p.y_img = d->y_ctotal * HW; p.r_img = res ? res->ctotal * HW : 0;
  p.x_base = d->x_coff * HW; p.y_base = d->y_coff * HW; p.r_base = res ? res->coff * HW : 0;
  p.w2 = w_chain; p.y2 = (float*)out->data; p.y2_img = out->ctotal * HW; p.y2_base = out->coff * HW;
  p.tiles_x = (d->width + 31) / 32;
  p.tiles_y = (d->height + TH - 1) / TH;
  const long nblk = (long)p.tiles_x * p.tiles_y * d->batch;
  CODON_REQUIRE(nblk < (1L << 31), CODON_ERR_UNSUPPORTED, "conv_chain1x1_fwd: grid too large (%ld blocks)", nblk);
  p.nblk = (int)nblk;
  p.flags = d->flags;
  hipLaunchKernelGGL((conv_mfma_f32_kernel<5, 128, 128, 2, true>), dim3((unsigned)nblk), dim3(256), 0, stream, p);
  return check_launch("conv_mfma_f32_kernel<fused 1x1>");
}

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
