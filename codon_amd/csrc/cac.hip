// CAC cross-domain gate kernels (HBM-bound).  Reference: /root/reference/CODON_X4/CAC_module.py
//   CAC_channel.forward :38-63, ChannelPool :78-81, CAC_spatial.forward :90-94,
//   gate application + block residual CODON_x4.py:85-118.
// Fcat = cat(out_c, out) (colour channels 0..63, depth channels 64..127) is never built: the two
// 64-channel producers are read in place.  ONE pass over Fcat yields everything both gates need:
//   per pixel : max and mean over the 128 channels           (ChannelPool)
//   per (b,c) : sum and max over H*W, as per-tile partials    (avg_pool2d / max_pool2d, stage 1)
// The reference reads Fcat four times for the same quantities and memsets a 2.5 GB host tensor
// per call (CAC_module.py:39); none of that is reproduced.

#include <math.h>

#include "codon_common.h"
#include "px8.h"

namespace codon {

constexpr int STATS_TILE = PX_TILE;  // pixels per workgroup (8 per thread)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}

// grid = (ntiles, B).  P = pixel-ownership policy (px8.h): 8 pixels per thread, widest coalesced access.
template <class P>
__global__ __launch_bounds__(256) void cac_stats_kernel(const typename P::T* __restrict__ pre_c, long pc_img,
                                                        const typename P::T* __restrict__ pre, long p_img,
                                                        float* __restrict__ pooled, float* __restrict__ partials,
                                                        long HW, int ntiles, const float* __restrict__ chs) {
  // chs (optional, (B,64)): every value of channel c is multiplied by chs[b][c & 63] first -- the statistics of the
  // CHANNEL-GATED features, which the sequential-gate ablation feeds to its spatial gate
  // (CODON_X4/base_net_withoutBN.py:2246-2250)
  __shared__ float red[128][4][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = blockIdx.x, b = blockIdx.y;
  const long tile0 = (long)tile * PX_TILE;
  bool ok[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) ok[i] = P::pix(tile0, tid, i) < HW;
  float pmax[8], psum[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { pmax[i] = -INFINITY; psum[i] = 0.f; }

  // 8 channels per trip, all their loads issued before the first is used (one image of 128 x 128 is 8 workgroups walking
  // 128 planes: with one load per trip the pass was a chain of 128 memory latencies, 83 us of a 4.4 ms forward)
  constexpr int CB = 8;
#pragma unroll 1
  for (int c0 = 0; c0 < 128; c0 += CB) {
    float vv[CB][8];
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      const int c = c0 + j;
      const typename P::T* plane = (c < 64 ? pre_c + b * pc_img + c * HW : pre + b * p_img + (c - 64) * HW);
      P::load(plane, tile0, tid, HW, vv[j]);
    }
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      const int c = c0 + j;
      float (&v)[8] = vv[j];
      if (chs) {
        const float g = chs[b * 64 + (c & 63)];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= g;
      }
      float s = 0.f, m = -INFINITY;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        pmax[i] = fmaxf(pmax[i], v[i]);
        psum[i] += v[i];
        s += v[i];                          // out-of-range pixels load as 0
        m = ok[i] ? fmaxf(m, v[i]) : m;
      }
      s = wave_sum(s);
      m = wave_max(m);
      if (lane == 0) { red[c][wave][0] = s; red[c][wave][1] = m; }
    }
  }
  // per-pixel outputs: plane 0 = channel max, plane 1 = channel mean (max FIRST, CAC_module.py:81)
  float* pm = pooled + (long)b * 2 * HW;
#pragma unroll
  for (int i = 0; i < 8; ++i) psum[i] *= (1.f / 128.f);
  P::storef(pm, tile0, tid, HW, pmax);
  P::storef(pm + HW, tile0, tid, HW, psum);
  __syncthreads();
  if (tid < 128) {
    const float s = (red[tid][0][0] + red[tid][1][0]) + (red[tid][2][0] + red[tid][3][0]);
    const float m = fmaxf(fmaxf(red[tid][0][1], red[tid][1][1]), fmaxf(red[tid][2][1], red[tid][3][1]));
    float2* out = reinterpret_cast<float2*>(partials + (((long)b * ntiles + tile) * 128 + tid) * 2);
    *out = make_float2(s, m);
  }
}

// grid = B, 128 threads.  Second (fixed-order) stage of the global pools + shared MLP + sigmoid.
__global__ __launch_bounds__(128) void cac_gate_kernel(const float* __restrict__ partials, const float* __restrict__ w1,
                                                       const float* __restrict__ b1, const float* __restrict__ w2,
                                                       const float* __restrict__ b2, float* __restrict__ ch,
                                                       float* __restrict__ pools_out, int ntiles, float inv_hw) {
  __shared__ float pool[2][128];
  __shared__ float hid[2][8];
  const int c = threadIdx.x, b = blockIdx.x;
  float s = 0.f, m = -INFINITY;
  const float2* p = reinterpret_cast<const float2*>(partials) + (long)b * ntiles * 128 + c;
  for (int t = 0; t < ntiles; ++t) {
    const float2 v = p[(long)t * 128];
    s += v.x;
    m = fmaxf(m, v.y);
  }
  pool[0][c] = s * inv_hw;
  pool[1][c] = m;
  if (pools_out) {
    pools_out[((long)b * 2 + 0) * 128 + c] = s * inv_hw;
    pools_out[((long)b * 2 + 1) * 128 + c] = m;
  }
  __syncthreads();
  if (c < 16) {  // hidden layer: Linear(128, 8) + ReLU, for avg (c<8) and max (c>=8)
    const int which = c >> 3, j = c & 7;
    float a = b1[j];
    for (int k = 0; k < 128; ++k) a = fmaf(w1[j * 128 + k], pool[which][k], a);
    hid[which][j] = fmaxf(a, 0.f);
  }
  __syncthreads();
  if (c < 64) {  // Linear(8, 64) for both pools, summed, sigmoid
    float a0 = b2[c], a1 = b2[c];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a0 = fmaf(w2[c * 8 + j], hid[0][j], a0);
      a1 = fmaf(w2[c * 8 + j], hid[1][j], a1);
    }
    const float z = a0 + a1;
    ch[(long)b * 64 + c] = 1.f / (1.f + expf(-z));
  }
}

// sp = sigmoid(conv5x5_{2->1, pad 2, no bias}(pooled)).  A 32 x 32 pixel tile per workgroup: the two pooled planes are
// staged once with their 2-pixel halo (zeros outside the image = the conv's zero padding), a thread owns 4 consecutive
// rows of one column and reads each plane's 8 x 5 window once; the 50 weights come by scalar loads.  Taps in the order
// (plane, dy, dx) of the previous one-load-per-tap kernel: same sums bit for bit (a padded tap adds w * 0).
// (Round 2: three aligned float4 loads per (plane, row) and 4 pixels, 0.13 ms for a 40 MB map; now 0.04.)
constexpr int SPF_T = 32, SPF_HALO = SPF_T + 4, SPF_PITCH = SPF_HALO + 1;
__global__ __launch_bounds__(256) void cac_spatial_kernel(const float* __restrict__ pooled, const float* __restrict__ w,
                                                          float* __restrict__ sp, int H, int W, int tiles_x, int tiles_y) {
  __shared__ float tl[2][SPF_HALO][SPF_PITCH];
  const int tid = threadIdx.x;
  const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y, b = blockIdx.x / (tiles_x * tiles_y);
  const int x0 = tx * SPF_T, y0 = ty * SPF_T;
  const long HW = (long)H * W;
  const float* base = pooled + (long)b * 2 * HW;
  {  // two phases, fully unrolled: all loads of the halo tile in flight before the first use (see cac_tail_kernel)
    constexpr int NE = (2 * SPF_HALO * SPF_HALO + 255) / 256;
    float vv[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      const int e = min(tid + k * 256, 2 * SPF_HALO * SPF_HALO - 1);
      const int c = e / (SPF_HALO * SPF_HALO), rem = e - c * (SPF_HALO * SPF_HALO);
      const int r = rem / SPF_HALO, q = rem - r * SPF_HALO;
      const int yy = min(max(y0 + r - 2, 0), H - 1), xx = min(max(x0 + q - 2, 0), W - 1);
      vv[k] = base[c * HW + (long)yy * W + xx];
    }
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      const int e = tid + k * 256;
      if (e >= 2 * SPF_HALO * SPF_HALO) break;
      const int c = e / (SPF_HALO * SPF_HALO), rem = e - c * (SPF_HALO * SPF_HALO);
      const int r = rem / SPF_HALO, q = rem - r * SPF_HALO;
      const int yy = y0 + r - 2, xx = x0 + q - 2;
      tl[c][r][q] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? vv[k] : 0.f;
    }
  }
  __syncthreads();
  const int cx = tid & 31, r0 = (tid >> 5) * 4;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    float win[8][5];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr)
#pragma unroll
      for (int dx = 0; dx < 5; ++dx) win[rr][dx] = tl[c][r0 + rr][cx + dx];
#pragma unroll
    for (int dy = 0; dy < 5; ++dy)
#pragma unroll
      for (int dx = 0; dx < 5; ++dx) {
        const float k = w[(c * 5 + dy) * 5 + dx];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = fmaf(k, win[i + dy][dx], a[i]);
      }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gy = y0 + r0 + i, gx = x0 + cx;
    if (gy < H && gx < W) sp[(long)b * HW + (long)gy * W + gx] = 1.f / (1.f + expf(-a[i]));
  }
}

// out = pre * (ch*sp) + inputs  for both streams (blockIdx.z selects the stream).  grid = (ntiles, B*64, 2)
template <class T>
struct ApplyStream {
  const T* pre; const T* in; T* out;
  long pre_img, in_img, out_img;  // elements per image (ctotal*HW); base pointers include coff*HW
};
template <class P>
__global__ __launch_bounds__(256) void cac_apply_kernel(const ApplyStream<typename P::T> sd,
                                                        const ApplyStream<typename P::T> sc,
                                                        const float* __restrict__ ch, const float* __restrict__ sp,
                                                        long HW) {
  const int bc = blockIdx.y;  // b*64 + c
  const int b = bc >> 6, c = bc & 63;
  const int tid = threadIdx.x;
  const long tile0 = (long)blockIdx.x * PX_TILE;
  const ApplyStream<typename P::T>& s = blockIdx.z ? sc : sd;
  float v[8], q[8], g[8];
  P::load(s.pre + b * s.pre_img + c * HW, tile0, tid, HW, v);
  P::load(s.in + b * s.in_img + c * HW, tile0, tid, HW, q);
  P::loadf(sp + (long)b * HW, tile0, tid, HW, g);
  const float gc = ch[bc];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], gc * g[i], q[i]);
  P::store(s.out + b * s.out_img + c * HW, tile0, tid, HW, v);
}

// ew_c8.hip: the same passes over channel-blocked 16-bit tensors
int cac_stats_fwd_c8(int, int, int, const codon_tensor*, const codon_tensor*, float*, float*, int, hipStream_t, const float*);
int cac_apply_fwd_c8(int, int, int, const codon_tensor*, const codon_tensor*, const float*, const float*, const codon_tensor*,
                     const codon_tensor*, const codon_tensor*, const codon_tensor*, int, hipStream_t);
int ew_sq_scale_c8(int, int, int, const codon_tensor*, const float*, const codon_tensor*, int, hipStream_t);

static bool aligned16(const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr) {
  return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
           reinterpret_cast<uintptr_t>(d)) % 16) == 0;
}

// SMALL IMAGES (one 128 x 128 pair per call, BASELINE configs[0]): 2048-pixel tiles make 8 workgroups that walk 128 planes in
// 16 dependent trips -- 57 us of a 2.9 ms forward on a chip with 256 CUs (profiles/r05_b1_fp32_128x128_timeline.txt).  Up to
// STATS_SMALL_HW pixels a tile is 256 pixels (one per thread, 64 workgroups for 128 x 128) and a trip holds 32 planes.  The
// choice depends on H * W only, never on the batch: an image's statistics do not depend on what else is in the batch.
constexpr long STATS_SMALL_HW = 32768;
constexpr int STATS_SMALL_TILE = 256;
static bool stats_small(long HW) { return HW <= STATS_SMALL_HW; }
int cac_stats_tiles(int H, int W) {
  const long HW = (long)H * W;
  return stats_small(HW) ? (int)((HW + STATS_SMALL_TILE - 1) / STATS_SMALL_TILE) : (int)((HW + STATS_TILE - 1) / STATS_TILE);
}

// grid = (ntiles, B), 256 threads, thread = pixel.  Same outputs as cac_stats_kernel (per-pixel channel max / mean, per-tile
// per-channel {sum, max}); sums in a different (fixed) order: DPP tree per channel and wave, then the four waves.
__global__ __launch_bounds__(256) void cac_stats_small_kernel(const float* __restrict__ pre_c, long pc_img,
                                                              const float* __restrict__ pre, long p_img,
                                                              float* __restrict__ pooled, float* __restrict__ partials,
                                                              long HW, int ntiles, const float* __restrict__ chs) {
  __shared__ float red[128][4][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = blockIdx.x, b = blockIdx.y;
  const long q = (long)tile * STATS_SMALL_TILE + tid;
  const bool ok = q < HW;
  const long qq = ok ? q : 0;                       // unconditional loads from a valid address, masked below
  float pmax = -INFINITY, psum = 0.f;
  constexpr int CB = 32;
#pragma unroll 1
  for (int c0 = 0; c0 < 128; c0 += CB) {
    float v[CB];
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      const int c = c0 + j;
      v[j] = (c < 64 ? pre_c + b * pc_img + c * HW : pre + b * p_img + (c - 64) * HW)[qq];
    }
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      const int c = c0 + j;
      float x = v[j];
      if (chs) x *= chs[b * 64 + (c & 63)];
      pmax = fmaxf(pmax, x);
      psum += x;
      const float s = wave_red63<false>(ok ? x : 0.f), m = wave_red63<true>(ok ? x : -INFINITY);
      if (lane == 63) { red[c][wave][0] = s; red[c][wave][1] = m; }
    }
  }
  if (ok) {
    float* pm = pooled + (long)b * 2 * HW;
    pm[q] = pmax;
    pm[HW + q] = psum * (1.f / 128.f);
  }
  __syncthreads();
  if (tid < 128) {
    const float s = (red[tid][0][0] + red[tid][1][0]) + (red[tid][2][0] + red[tid][3][0]);
    const float m = fmaxf(fmaxf(red[tid][0][1], red[tid][1][1]), fmaxf(red[tid][2][1], red[tid][3][1]));
    float2* out = reinterpret_cast<float2*>(partials + (((long)b * ntiles + tile) * 128 + tid) * 2);
    *out = make_float2(s, m);
  }
}

int cac_stats_fwd(int B, int H, int W, const codon_tensor* pc, const codon_tensor* pd, float* pooled,
                  float* partials, int dtype, hipStream_t stream, const float* chs) {
  if (dtype != CODON_F32) return cac_stats_fwd_c8(B, H, W, pc, pd, pooled, partials, dtype, stream, chs);
  const long HW = (long)H * W;
  const int nt = cac_stats_tiles(H, W);
  CODON_REQUIRE(B <= 65535, CODON_ERR_UNSUPPORTED, "cac_stats_fwd: batch %d > 65535", B);
  const size_t es = 4;
  const char* pre_c = (const char*)pc->data + pc->coff * HW * es;
  const char* pre = (const char*)pd->data + pd->coff * HW * es;
  if (stats_small(HW)) {
    hipLaunchKernelGGL(cac_stats_small_kernel, dim3(nt, B), dim3(256), 0, stream, (const float*)pre_c, pc->ctotal * HW,
                       (const float*)pre, pd->ctotal * HW, pooled, partials, HW, nt, chs);
    return check_launch("cac_stats_small_kernel");
  }
  px_dispatch(dtype, HW, aligned16(pre_c, pre, pooled), [&](auto pol) {
    using P = decltype(pol);
    hipLaunchKernelGGL(cac_stats_kernel<P>, dim3(nt, B), dim3(256), 0, stream, (const typename P::T*)pre_c,
                       pc->ctotal * HW, (const typename P::T*)pre, pd->ctotal * HW, pooled, partials, HW, nt, chs);
  });
  return check_launch("cac_stats_kernel");
}

int cac_gate_fwd(int B, int H, int W, const float* partials, const float* w1, const float* b1, const float* w2,
                 const float* b2, float* ch, float* pools_out, hipStream_t stream) {
  const int nt = cac_stats_tiles(H, W);
  hipLaunchKernelGGL(cac_gate_kernel, dim3(B), dim3(128), 0, stream, partials, w1, b1, w2, b2, ch, pools_out, nt,
                     (float)(1.0 / ((double)H * W)));
  return check_launch("cac_gate_kernel");
}

int cac_gate_fwd_n(int B, int ntiles, float inv_hw, const float* partials, const float* w1, const float* b1, const float* w2,
                   const float* b2, float* ch, float* pools_out, hipStream_t stream) {
  hipLaunchKernelGGL(cac_gate_kernel, dim3(B), dim3(128), 0, stream, partials, w1, b1, w2, b2, ch, pools_out, ntiles, inv_hw);
  return check_launch("cac_gate_kernel");
}

// ---- fused-statistics finish (16-bit path: the conv5x5 + 1x1 epilogue produced the statistics, conv_c8.hip) -----------
// fold   : (B, ntiles, 128, 2) per-tile {sum, max} -> (B, CODON_CAC_FOLDS, 128, 2): fold f adds tiles [f*per, (f+1)*per) in order
//          (fixed order: deterministic, batch invariant); cac_gate_kernel then finishes over the CODON_CAC_FOLDS rows
// combine: pooled (B,2,H,W) = { max(max_c, max_d), (sum_c + sum_d) / 128 } from the two per-stream maps
// s += the { sum } and m = max of the { max } of rows [t0, t1) of one channel, IN ROW ORDER -- eight rows' loads in flight at a
// time (a plain loop issues one load, waits, adds: one memory round trip per row; one 128 x 128 fp32 image folds 32 strips per
// workgroup, one 370 x 463 fp16 image 44 tiles)
__device__ __forceinline__ void fold_rows(const float2* __restrict__ p, int t0, int t1, float& s, float& m) {
  int t = t0;
  for (; t + 8 <= t1; t += 8) {
    float2 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = p[(long)(t + k) * 128];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s += v[k].x; m = fmaxf(m, v[k].y); }
  }
  for (; t < t1; ++t) {
    const float2 v = p[(long)t * 128];
    s += v.x;
    m = fmaxf(m, v.y);
  }
}

__global__ __launch_bounds__(128) void cac_fold_kernel(const float* __restrict__ partials, float* __restrict__ folded,
                                                       int ntiles, int per) {
  const int c = threadIdx.x, f = blockIdx.x, b = blockIdx.y;
  const int t0 = f * per, t1 = min(t0 + per, ntiles);
  float s = 0.f, m = -INFINITY;
  const float2* p = reinterpret_cast<const float2*>(partials) + (long)b * ntiles * 128 + c;
  fold_rows(p, t0, t1, s, m);
  reinterpret_cast<float2*>(folded)[((long)b * gridDim.x + f) * 128 + c] = make_float2(s, m);
}

__global__ __launch_bounds__(256) void cac_pool_combine_kernel(const float* __restrict__ pc, const float* __restrict__ pd,
                                                               float* __restrict__ pooled, long HW, long total) {
  const long i = blockIdx.x * 256L + threadIdx.x;      // over B * HW pixels
  if (i >= total) return;
  const long b = i / HW, q = i - b * HW;
  const long o = b * 2 * HW + q;
  pooled[o] = fmaxf(pc[o], pd[o]);
  pooled[o + HW] = (pc[o + HW] + pd[o + HW]) * (1.f / 128.f);
}

int cac_fused_finish(int B, int H, int W, int ntiles, const float* partials, const float* pool_c, const float* pool_d,
                     float* folded, float* pooled, hipStream_t stream) {
  const long HW = (long)H * W;
  const int per = (ntiles + CODON_CAC_FOLDS - 1) / CODON_CAC_FOLDS;
  hipLaunchKernelGGL(cac_fold_kernel, dim3(CODON_CAC_FOLDS, B), dim3(128), 0, stream, partials, folded, ntiles, per);
  int st = check_launch("cac_fold_kernel");
  if (st != CODON_OK) return st;
  const long total = (long)B * HW;
  CODON_REQUIRE((total + 255) / 256 < (1L << 31), CODON_ERR_UNSUPPORTED, "cac_fused_finish: grid too large");
  hipLaunchKernelGGL(cac_pool_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, pool_c, pool_d,
                     pooled, HW, total);
  return check_launch("cac_pool_combine_kernel");
}

// ---- the whole CAC gate of a block in ONE launch (round 5) ---------------------------------------------------------------
// At one image per call (the reference script's own calling pattern, /root/reference/CODON_X4/test.py:116-125) the gate of a
// block was four dependent launches of a few microseconds each -- fold, combine, pool finish + MLP, spatial conv -- i.e. 20
// launches per forward that cost more in dependency gaps than in work.  Here one grid does all of it:
//   workgroups [0, nsp)          : a 32 x 32 tile of sp = sigmoid(conv5x5_{2->1}(pooled)) (CAC_module.py:88,92-93), the pooled
//                                  planes formed while the halo tile is staged (pool_c != null: { max(max_c, max_d),
//                                  (sum_c + sum_d) / 128 } from the two per-stream maps, CAC_module.py:81) and, when
//                                  `pooled_out` is given (training keeps it), written for the tile's own pixels;
//   workgroups [nsp, nsp + F B)  : fold f of image b over the per-tile {sum, max} rows (cac_fold_kernel's ranges); the LAST
//                                  fold of an image to arrive (one atomic per workgroup, as wsum.hip) finishes the pools over
//                                  the F folded rows and runs the MLP + sigmoid (cac_gate_kernel's arithmetic and order).
// Every sum keeps the order of the four-launch form: same bits (tests/test_gpu_c8.py).  counters: B int32, zero on entry,
// left at zero.
struct CacTailArgs {
  const float* partials; const float* pool_c; const float* pool_d; const float* pooled_in; float* pooled_out;
  float* folded; int* counters;
  const float* w1; const float* b1; const float* w2; const float* b2; const float* ws;
  float* ch; float* pools_out; float* sp;
  int H, W, tiles_x, tiles_y, nsp, ntiles, per;
  float inv_hw;
};
static_assert(sizeof(CacTailArgs) <= CODON_KERNARG_LIMIT, "passed by value as a kernel argument");

__global__ __launch_bounds__(256) void cac_tail_kernel(const CacTailArgs a) {
  __shared__ float tl[2][SPF_HALO][SPF_PITCH];
  __shared__ bool last;
  const int tid = threadIdx.x;
  const int H = a.H, W = a.W;
  const long HW = (long)H * W;
  if ((int)blockIdx.x < a.nsp) {
    const int tx = blockIdx.x % a.tiles_x, ty = (blockIdx.x / a.tiles_x) % a.tiles_y, b = blockIdx.x / (a.tiles_x * a.tiles_y);
    const int x0 = tx * SPF_T, y0 = ty * SPF_T;
    const long ib = (long)b * 2 * HW;
    // two phases, fully unrolled: every load of the halo tile is in flight before the first value is used or stored (a
    // load / combine / store loop ran the tile's 11 rounds as 11 dependent memory round trips: 287 us per launch at
    // 32 x 480 x 640 against 156 us for the four kernels this one replaces)
    constexpr int NE = (2 * SPF_HALO * SPF_HALO + 255) / 256;
    float va[NE], vb[NE];
    long oo[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      const int e = tid + k * 256;
      const int c = e / (SPF_HALO * SPF_HALO), rem = e - c * (SPF_HALO * SPF_HALO);
      const int r = rem / SPF_HALO, q = rem - r * SPF_HALO;
      const int yy = y0 + r - 2, xx = x0 + q - 2;
      const bool in = e < 2 * SPF_HALO * SPF_HALO && yy >= 0 && yy < H && xx >= 0 && xx < W;
      oo[k] = in ? ib + c * HW + (long)yy * W + xx : -1;
      const long o = in ? oo[k] : ib;                         // unconditional loads from a valid address, masked below
      if (a.pool_c) { va[k] = a.pool_c[o]; vb[k] = a.pool_d[o]; }
      else { va[k] = a.pooled_in[o]; vb[k] = 0.f; }
    }
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      const int e = tid + k * 256;
      if (e >= 2 * SPF_HALO * SPF_HALO) break;
      const int c = e / (SPF_HALO * SPF_HALO), rem = e - c * (SPF_HALO * SPF_HALO);
      const int r = rem / SPF_HALO, q = rem - r * SPF_HALO;
      float v = 0.f;
      if (oo[k] >= 0) {
        if (a.pool_c) {
          v = c == 0 ? fmaxf(va[k], vb[k]) : (va[k] + vb[k]) * (1.f / 128.f);
          if (a.pooled_out && r >= 2 && r < 2 + SPF_T && q >= 2 && q < 2 + SPF_T) a.pooled_out[oo[k]] = v;
        } else {
          v = va[k];
        }
      }
      tl[c][r][q] = v;
    }
    __syncthreads();
    const int cx = tid & 31, r0 = (tid >> 5) * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      float win[8][5];
#pragma unroll
      for (int rr = 0; rr < 8; ++rr)
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) win[rr][dx] = tl[c][r0 + rr][cx + dx];
#pragma unroll
      for (int dy = 0; dy < 5; ++dy)
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) {
          const float k = a.ws[(c * 5 + dy) * 5 + dx];
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] = fmaf(k, win[i + dy][dx], acc[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gy = y0 + r0 + i, gx = x0 + cx;
      if (gy < H && gx < W) a.sp[(long)b * HW + (long)gy * W + gx] = 1.f / (1.f + expf(-acc[i]));
    }
    return;
  }
  // fold f of image b, then -- last arrival only -- the pool finish + MLP
  const int fb = (int)blockIdx.x - a.nsp;
  const int f = fb % CODON_CAC_FOLDS, b = fb / CODON_CAC_FOLDS;
  float2* const folded = reinterpret_cast<float2*>(a.folded) + (long)b * CODON_CAC_FOLDS * 128;
  if (tid < 128) {
    const int t0 = f * a.per, t1 = min(t0 + a.per, a.ntiles);
    float s = 0.f, m = -INFINITY;
    const float2* p = reinterpret_cast<const float2*>(a.partials) + (long)b * a.ntiles * 128 + tid;
    fold_rows(p, t0, t1, s, m);
    folded[f * 128 + tid] = make_float2(s, m);
  }
  __threadfence();                 // this workgroup's row is visible device-wide before its arrival is
  __syncthreads();
  if (tid == 0) last = atomicAdd(&a.counters[b], 1) == CODON_CAC_FOLDS - 1;
  __syncthreads();
  if (!last) return;
  __threadfence();                 // acquire: the other workgroups' rows
  float* const pool = &tl[0][0][0];          // [2][128]
  float* const hid = pool + 256;             // [2][8]
  if (tid == 0) a.counters[b] = 0;           // ready for the next block's launch on this buffer
  if (tid < 128) {
    float s = 0.f, m = -INFINITY;
    for (int t = 0; t < CODON_CAC_FOLDS; ++t) {
      // (atomic loads: never served from a stale line of this XCD's cache)
      const unsigned* q = reinterpret_cast<const unsigned*>(a.folded) + (((long)b * CODON_CAC_FOLDS + t) * 128 + tid) * 2;
      s += __uint_as_float(__atomic_load_n(q, __ATOMIC_RELAXED));
      m = fmaxf(m, __uint_as_float(__atomic_load_n(q + 1, __ATOMIC_RELAXED)));
    }
    pool[tid] = s * a.inv_hw;
    pool[128 + tid] = m;
    if (a.pools_out) {
      a.pools_out[((long)b * 2 + 0) * 128 + tid] = s * a.inv_hw;
      a.pools_out[((long)b * 2 + 1) * 128 + tid] = m;
    }
  }
  __syncthreads();
  if (tid < 16) {  // hidden layer: Linear(128, 8) + ReLU, for avg (tid < 8) and max
    const int which = tid >> 3, j = tid & 7;
    float v = a.b1[j];
    for (int k = 0; k < 128; ++k) v = fmaf(a.w1[j * 128 + k], pool[which * 128 + k], v);
    hid[which * 8 + j] = fmaxf(v, 0.f);
  }
  __syncthreads();
  if (tid < 64) {  // Linear(8, 64) for both pools, summed, sigmoid
    float a0 = a.b2[tid], a1 = a.b2[tid];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a0 = fmaf(a.w2[tid * 8 + j], hid[j], a0);
      a1 = fmaf(a.w2[tid * 8 + j], hid[8 + j], a1);
    }
    const float z = a0 + a1;
    a.ch[(long)b * 64 + tid] = 1.f / (1.f + expf(-z));
  }
}

int cac_tail_fwd(int B, int H, int W, int ntiles, const float* partials, const float* pool_c, const float* pool_d,
                 const float* pooled_in, float* pooled_out, float* folded, int* counters, const float* w1, const float* b1,
                 const float* w2, const float* b2, const float* ws, float* ch, float* pools_out, float* sp, hipStream_t stream) {
  CacTailArgs a;
  a.partials = partials; a.pool_c = pool_c; a.pool_d = pool_d; a.pooled_in = pooled_in; a.pooled_out = pooled_out;
  a.folded = folded; a.counters = counters; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.ws = ws;
  a.ch = ch; a.pools_out = pools_out; a.sp = sp;
  a.H = H; a.W = W;
  a.tiles_x = (W + SPF_T - 1) / SPF_T; a.tiles_y = (H + SPF_T - 1) / SPF_T;
  const long nsp = (long)B * a.tiles_x * a.tiles_y, nblk = nsp + (long)B * CODON_CAC_FOLDS;
  CODON_REQUIRE(nblk < (1L << 31), CODON_ERR_UNSUPPORTED, "cac_tail_fwd: grid too large");
  a.nsp = (int)nsp; a.ntiles = ntiles;
  a.per = (ntiles + CODON_CAC_FOLDS - 1) / CODON_CAC_FOLDS;
  a.inv_hw = (float)(1.0 / ((double)H * W));
  hipLaunchKernelGGL(cac_tail_kernel, dim3((unsigned)nblk), dim3(256), 0, stream, a);
  return check_launch("cac_tail_kernel");
}

int cac_spatial_fwd(int B, int H, int W, const float* pooled, const float* w, float* sp, hipStream_t stream) {
  const int tiles_x = (W + SPF_T - 1) / SPF_T, tiles_y = (H + SPF_T - 1) / SPF_T;
  const long blocks = (long)B * tiles_x * tiles_y;
  CODON_REQUIRE(blocks < (1L << 31), CODON_ERR_UNSUPPORTED, "cac_spatial_fwd: grid too large");
  hipLaunchKernelGGL(cac_spatial_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, pooled, w, sp, H, W, tiles_x, tiles_y);
  return check_launch("cac_spatial_kernel");
}

int cac_apply_fwd(int B, int H, int W, const codon_tensor* pre, const codon_tensor* pre_c, const float* ch,
                  const float* sp, const codon_tensor* in, const codon_tensor* in_c, const codon_tensor* out,
                  const codon_tensor* out_c, int dtype, hipStream_t stream) {
  if (dtype != CODON_F32) return cac_apply_fwd_c8(B, H, W, pre, pre_c, ch, sp, in, in_c, out, out_c, dtype, stream);
  const long HW = (long)H * W;
  CODON_REQUIRE((long)B * 64 <= 65535, CODON_ERR_UNSUPPORTED, "cac_apply_fwd: batch %d too large", B);
  const size_t es = 4;
  auto base = [&](const codon_tensor* t) { return (char*)t->data + t->coff * HW * es; };
  const bool al = aligned16(base(pre), base(in), base(out), sp) && aligned16(base(pre_c), base(in_c), base(out_c));
  const unsigned nt = (unsigned)((HW + PX_TILE - 1) / PX_TILE);
  px_dispatch(dtype, HW, al, [&](auto pol) {
    using P = decltype(pol);
    using T = typename P::T;
    auto mk = [&](const codon_tensor* p, const codon_tensor* i, const codon_tensor* o) {
      ApplyStream<T> s;
      s.pre = (const T*)base(p); s.pre_img = p->ctotal * HW;
      s.in = (const T*)base(i); s.in_img = i->ctotal * HW;
      s.out = (T*)base(o); s.out_img = o->ctotal * HW;
      return s;
    };
    hipLaunchKernelGGL(cac_apply_kernel<P>, dim3(nt, B * 64, 2), dim3(256), 0, stream, mk(pre, in, out),
                       mk(pre_c, in_c, out_c), ch, sp, HW);
  });
  return check_launch("cac_apply_kernel");
}

// y = x * x * ch[b][c]   (64 channels): fuse * ChannelGate(fuse) of the sequential-gate ablation's trunk, where
// ChannelGate.forward returns x * scale (attention/ResCBAM.py:60-61) and the caller multiplies by x again
// (CODON_X4/base_net_withoutBN.py:2297-2298).  grid = (tiles, B * 64).
template <class P>
__global__ __launch_bounds__(256) void ew_sq_scale_kernel(const typename P::T* __restrict__ x, long x_img,
                                                          const float* __restrict__ ch, typename P::T* __restrict__ y,
                                                          long y_img, long HW) {
  const int tid = threadIdx.x;
  const int b = blockIdx.y >> 6, c = blockIdx.y & 63;
  const long tile0 = (long)blockIdx.x * PX_TILE;
  const float g = ch[b * 64 + c];
  float v[8];
  P::load(x + b * x_img + c * HW, tile0, tid, HW, v);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = v[i] * v[i] * g;
  P::store(y + b * y_img + c * HW, tile0, tid, HW, v);
}

int ew_sq_scale(int B, int H, int W, const codon_tensor* x, const float* ch, const codon_tensor* y, int dtype,
                hipStream_t stream) {
  if (dtype != CODON_F32) return ew_sq_scale_c8(B, H, W, x, ch, y, dtype, stream);
  const long HW = (long)H * W;
  CODON_REQUIRE((long)B * 64 <= 65535, CODON_ERR_UNSUPPORTED, "ew_sq_scale: batch %d too large", B);
  const size_t es = 4;
  const char* xp = (const char*)x->data + x->coff * HW * es;
  char* yp = (char*)y->data + y->coff * HW * es;
  const int nt = (int)((HW + PX_TILE - 1) / PX_TILE);
  px_dispatch(dtype, HW, aligned16(xp, yp, yp), [&](auto pol) {
    using P = decltype(pol);
    using T = typename P::T;
    hipLaunchKernelGGL(ew_sq_scale_kernel<P>, dim3(nt, B * 64), dim3(256), 0, stream, (const T*)xp, x->ctotal * HW, ch,
                       (T*)yp, y->ctotal * HW, HW);
  });
  return check_launch("ew_sq_scale_kernel");
}

}  // namespace codon
