// Backward of the CAC gate block (what torch.autograd computes for
// /root/reference/CODON_X4/CODON_x4.py:85-118 + CAC_module.py:38-63,78-94; the reference has no
// explicit backward).  Forward:
//   Fcat = [pre_c | pre]                      (colour channels 0..63, depth 64..127)
//   ch  = sigmoid(mlp(avgpool(Fcat)) + mlp(maxpool(Fcat)))          (B,64)
//   sp  = sigmoid(conv5x5(cat(chmax(Fcat), chmean(Fcat))))          (B,1,H,W)
//   out = pre*ch*sp + inputs ; out_c = pre_c*ch*sp + inputs_c
// Given g_out, g_out_c this file produces g_pre, g_pre_c, the parameter gradients, and adds
// g_out / g_out_c into the running gradient of inputs / inputs_c.  Four kernels:
//   A cac_bwd_reduce : one pass over {g_out, g_out_c, pre, pre_c}: per-(b,c) partial sums of dL/dch,
//                      dL/dz (pre-sigmoid spatial logits) per pixel, first-arg-max pixel candidates
//                      for both global max-pools
//   B cac_bwd_gate   : per image: finish the reductions, sigmoid', MLP backward (per-image parameter
//                      gradient partials), dL/davg, dL/dmax, arg-max pixels
//   C cac_bwd_spatial: dL/dpooled (transposed 5x5) + per-block partials of the 5x5 weight gradient
//   D cac_bwd_apply  : one pass producing g_pre / g_pre_c (direct term + avg-pool broadcast + max-pool
//                      routing + channel-mean broadcast + channel-max routing to the FIRST arg-max
//                      channel in Fcat order, as torch.max does) and g_inputs (+)= g_out
// All reductions are two-stage with a fixed order: deterministic, batch-invariant.

#include <limits.h>
#include <math.h>

#include "codon_common.h"
#include "px8.h"

namespace codon {

constexpr int BWD_TILE = PX_TILE;  // pixels per workgroup, 8 per thread (same tiling as cac_stats)

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ int wmin(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = min(v, __shfl_xor(v, m, 64));
  return v;
}

template <class T>
struct Sl {  // 64-channel slice: base pointer already offset by coff*HW; img = ctotal*HW
  const T* p;
  long img;
};
template <class T>
struct SlW {
  T* p;
  long img;
};

// ---- A -------------------------------------------------------------------------------------------
template <class P>
__global__ __launch_bounds__(256) void cac_bwd_reduce_kernel(Sl<typename P::T> g_out, Sl<typename P::T> g_outc,
                                                             Sl<typename P::T> pre, Sl<typename P::T> pre_c,
                                                             const float* __restrict__ ch,
                                                             const float* __restrict__ sp,
                                                             const float* __restrict__ pools,  // (B,2,128)
                                                             float* __restrict__ g_z,          // (B,1,H,W)
                                                             float* __restrict__ part_gch,     // (B,nt,64)
                                                             int* __restrict__ part_arg,       // (B,nt,128)
                                                             long HW, int ntiles) {
  __shared__ float red_s[64][4];
  __shared__ int red_a[128][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = blockIdx.x, b = blockIdx.y;
  const long tile0 = (long)tile * BWD_TILE;
  bool ok[8];
  int pidx[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const long q = P::pix(tile0, tid, i);
    ok[i] = q < HW;
    pidx[i] = (int)q;
  }
  float spv[8], gsp[8];
  P::loadf(sp + (long)b * HW, tile0, tid, HW, spv);
#pragma unroll
  for (int i = 0; i < 8; ++i) gsp[i] = 0.f;
  const float* mx = pools + ((long)b * 2 + 1) * 128;

#pragma unroll 1
  for (int c = 0; c < 64; ++c) {
    float go[8], gc[8], p[8], pc[8];
    P::load(g_out.p + b * g_out.img + c * HW, tile0, tid, HW, go);
    P::load(g_outc.p + b * g_outc.img + c * HW, tile0, tid, HW, gc);
    P::load(pre.p + b * pre.img + c * HW, tile0, tid, HW, p);
    P::load(pre_c.p + b * pre_c.img + c * HW, tile0, tid, HW, pc);
    const float chc = ch[b * 64 + c];
    const float mxc = mx[c], mxd = mx[64 + c];  // Fcat order: colour c, depth 64+c
    float s = 0.f;
    int ad = INT_MAX, ac = INT_MAX;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float gg = go[i] * p[i] + gc[i] * pc[i];
      s = fmaf(gg, spv[i], s);
      gsp[i] = fmaf(gg, chc, gsp[i]);
      if (ok[i] && p[i] == mxd) ad = min(ad, pidx[i]);
      if (ok[i] && pc[i] == mxc) ac = min(ac, pidx[i]);
    }
    s = wsum(s);
    ad = wmin(ad);
    ac = wmin(ac);
    if (lane == 0) { red_s[c][wave] = s; red_a[c][wave] = ac; red_a[64 + c][wave] = ad; }
  }
  // dL/dz = dL/dsp * sp * (1 - sp)
#pragma unroll
  for (int i = 0; i < 8; ++i) gsp[i] = gsp[i] * spv[i] * (1.f - spv[i]);
  P::storef(g_z + (long)b * HW, tile0, tid, HW, gsp);
  __syncthreads();
  if (tid < 64)
    part_gch[((long)b * ntiles + tile) * 64 + tid] =
        (red_s[tid][0] + red_s[tid][1]) + (red_s[tid][2] + red_s[tid][3]);
  if (tid < 128)
    part_arg[((long)b * ntiles + tile) * 128 + tid] =
        min(min(red_a[tid][0], red_a[tid][1]), min(red_a[tid][2], red_a[tid][3]));
}

// ---- B -------------------------------------------------------------------------------------------
// grid = B.  param partial layout per image: w1 (8x128) | b1 (8) | w2 (64x8) | b2 (64) = 1608
constexpr int GATE_NPARAM = 8 * 128 + 8 + 64 * 8 + 64;

// 1024 threads = 128 channels x 8 slices of the tile range (round 5): the two walks over the per-tile partials -- the arg-max
// pixel of every channel, the 64 gate-gradient sums -- were ONE thread per channel stepping through 1200 tiles (480 x 640):
// 97 us per launch, latency.  Each slice walks its contiguous share, the eight results meet in LDS in slice order (fixed
// order: deterministic; min is exact, the sums are re-associated once).
constexpr int GATE_SLICES = 8;
__global__ __launch_bounds__(128 * GATE_SLICES) void cac_bwd_gate_kernel(const float* __restrict__ part_gch,
                                                           const int* __restrict__ part_arg,
                                                           const float* __restrict__ ch,
                                                           const float* __restrict__ pools,
                                                           const float* __restrict__ w1, const float* __restrict__ b1,
                                                           const float* __restrict__ w2,
                                                           float* __restrict__ g_pools,    // (B,2,128)
                                                           int* __restrict__ argpix,       // (B,128)
                                                           float* __restrict__ part_param, // (B,1608)
                                                           int ntiles) {
  __shared__ float pool[2][128], gs[64], hid[2][8], ghid[2][8];
  __shared__ int s_am[GATE_SLICES][128];
  __shared__ float s_gs[GATE_SLICES][64];
  const int t = threadIdx.x & 127, sl = threadIdx.x >> 7, b = blockIdx.x;
  {
    const int per = (ntiles + GATE_SLICES - 1) / GATE_SLICES;
    const int k0 = sl * per, k1 = min(k0 + per, ntiles);
    int am = INT_MAX;
    for (int k = k0; k < k1; ++k) am = min(am, part_arg[((long)b * ntiles + k) * 128 + t]);
    s_am[sl][t] = am;
    if (t < 64) {
      float s = 0.f;
      for (int k = k0; k < k1; ++k) s += part_gch[((long)b * ntiles + k) * 64 + t];
      s_gs[sl][t] = s;
    }
  }
  __syncthreads();
  if (sl != 0) return;                      // (no barrier below is reached by a subset: the other slices are done)
  pool[0][t] = pools[((long)b * 2 + 0) * 128 + t];
  pool[1][t] = pools[((long)b * 2 + 1) * 128 + t];
  int am = s_am[0][t];
#pragma unroll
  for (int q = 1; q < GATE_SLICES; ++q) am = min(am, s_am[q][t]);
  argpix[(long)b * 128 + t] = am;
  if (t < 64) {
    float s = s_gs[0][t];
#pragma unroll
    for (int q = 1; q < GATE_SLICES; ++q) s += s_gs[q][t];
    const float c = ch[(long)b * 64 + t];
    gs[t] = s * c * (1.f - c);
  }
  __syncthreads();
  if (t < 16) {
    const int which = t >> 3, j = t & 7;
    float a = b1[j];
    for (int k = 0; k < 128; ++k) a = fmaf(w1[j * 128 + k], pool[which][k], a);
    hid[which][j] = fmaxf(a, 0.f);
    float g = 0.f;
    for (int o = 0; o < 64; ++o) g = fmaf(w2[o * 8 + j], gs[o], g);
    ghid[which][j] = a > 0.f ? g : 0.f;
  }
  __syncthreads();
  float* pp = part_param + (long)b * GATE_NPARAM;
  {  // dL/dpools
    float ga = 0.f, gm = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      ga = fmaf(w1[j * 128 + t], ghid[0][j], ga);
      gm = fmaf(w1[j * 128 + t], ghid[1][j], gm);
    }
    g_pools[((long)b * 2 + 0) * 128 + t] = ga;
    g_pools[((long)b * 2 + 1) * 128 + t] = gm;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) pp[j * 128 + t] = ghid[0][j] * pool[0][t] + ghid[1][j] * pool[1][t];  // dW1
  if (t < 8) pp[1024 + t] = ghid[0][t] + ghid[1][t];                                               // db1
  if (t < 64) {
#pragma unroll
    for (int j = 0; j < 8; ++j) pp[1032 + t * 8 + j] = gs[t] * (hid[0][j] + hid[1][j]);            // dW2
    pp[1032 + 512 + t] = 2.f * gs[t];                                                              // db2
  }
}

// out[i] (+)= sum_k part[k*stride + i], fixed order
__global__ __launch_bounds__(256) void partial_sum_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                          int n, int nparts, long stride, int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < nparts; ++k) s += part[(long)k * stride + i];
  out[i] = accumulate ? out[i] + s : s;
}

// chunk c sums rows [c*per, min((c+1)*per, nrows)) of part (nrows, n) into row c*per (in place), fixed order
__global__ __launch_bounds__(64) void partial_chunk_sum_kernel(float* __restrict__ part, int n, int nrows, int per) {
  const int i = threadIdx.x;
  if (i >= n) return;
  const int r0 = blockIdx.x * per, r1 = min(r0 + per, nrows);
  if (r0 >= nrows) { return; }
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += part[(long)r * n + i];
  part[(long)r0 * n + i] = s;
}

// ---- C -------------------------------------------------------------------------------------------
// g_pooled[c][q] = sum_{dy,dx} w[c][dy][dx] * g_z[q - (dy-2, dx-2)]
// dW[c][dy][dx]  = sum_q g_z[q] * pooled[c][q + (dy-2, dx-2)]      (per-block partials)
// A 32 x 32 pixel tile per workgroup: g_z and the two pooled planes are staged ONCE with their 2-pixel halo (zeros outside
// the image: every border condition of the two sums becomes a zero operand), the 75 operand reads per pixel come from LDS
// (one global load per pixel and tap cost 0.67 - 1.07 ms for a 40 MB map in rounds 2 - 3; the three planes are 0.12 GB).
// A thread owns 4 consecutive pixels of one column and keeps the 50 weight-gradient partials in registers;
// they are reduced over the workgroup once per tile (fixed order: deterministic).
constexpr int SPB_T = 32, SPB_HALO = SPB_T + 4, SPB_PITCH = SPB_HALO + 1;
__global__ __launch_bounds__(256) void cac_bwd_spatial_kernel(const float* __restrict__ g_z,
                                                              const float* __restrict__ pooled,
                                                              const float* __restrict__ w,
                                                              float* __restrict__ g_pooled,
                                                              float* __restrict__ part_w,  // (nblk, 50)
                                                              int H, int W, int tiles_x, int tiles_y) {
  __shared__ float red[50][4];
  __shared__ float tl[3][SPB_HALO][SPB_PITCH];      // g_z, pooled[0], pooled[1]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y, b = blockIdx.x / (tiles_x * tiles_y);
  const int x0 = tx * SPB_T, y0 = ty * SPB_T;
  const long HW = (long)H * W;
  const float* gz = g_z + (long)b * HW;
  const float* pl = pooled + (long)b * 2 * HW;
  // two phases, fully unrolled: every load of the three halo planes is in flight before the first value is used (`in ? load
  // : 0` in a loop compiled to a branch and a full wait per round: 16 dependent memory round trips per workgroup)
  constexpr int NE = (3 * SPB_HALO * SPB_HALO + 255) / 256;
  float vv[NE];
#pragma unroll
  for (int k = 0; k < NE; ++k) {
    const int e = min(tid + k * 256, 3 * SPB_HALO * SPB_HALO - 1);
    const int c = e / (SPB_HALO * SPB_HALO), rem = e - c * (SPB_HALO * SPB_HALO);
    const int r = rem / SPB_HALO, q = rem - r * SPB_HALO;
    const int yy = min(max(y0 + r - 2, 0), H - 1), xx = min(max(x0 + q - 2, 0), W - 1);   // clamped: a valid address, masked below
    const float* src = c == 0 ? gz : pl + (long)(c - 1) * HW;
    vv[k] = src[(long)yy * W + xx];
  }
#pragma unroll
  for (int k = 0; k < NE; ++k) {
    const int e = tid + k * 256;
    if (e >= 3 * SPB_HALO * SPB_HALO) break;
    const int c = e / (SPB_HALO * SPB_HALO), rem = e - c * (SPB_HALO * SPB_HALO);
    const int r = rem / SPB_HALO, q = rem - r * SPB_HALO;
    const int yy = y0 + r - 2, xx = x0 + q - 2;
    tl[c][r][q] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? vv[k] : 0.f;
  }
  __syncthreads();
  // a thread owns 4 consecutive rows of one column: the 8 x 5 window of a plane is read once for its 4 pixels
  const int cx = tid & 31, r0 = (tid >> 5) * 4;
  float a0[25], a1[25], win[8][5], gq[4], o0[4], o1[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) gq[k] = tl[0][r0 + k + 2][cx + 2];   // 0 outside the image: its weight-gradient terms vanish
#define SPB_WINDOW(c_)                                                    \
  _Pragma("unroll") for (int rr = 0; rr < 8; ++rr)                        \
    _Pragma("unroll") for (int dx = 0; dx < 5; ++dx) win[rr][dx] = tl[c_][r0 + rr][cx + dx];
  SPB_WINDOW(1)
#pragma unroll
  for (int t = 0; t < 25; ++t) {                   // pooled at q + (dy-2, dx-2)
    float s_ = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) s_ = fmaf(gq[k], win[k + t / 5][t % 5], s_);
    a0[t] = s_;
  }
  SPB_WINDOW(2)
#pragma unroll
  for (int t = 0; t < 25; ++t) {
    float s_ = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) s_ = fmaf(gq[k], win[k + t / 5][t % 5], s_);
    a1[t] = s_;
  }
  SPB_WINDOW(0)
#undef SPB_WINDOW
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    o0[k] = 0.f; o1[k] = 0.f;
#pragma unroll
    for (int t = 0; t < 25; ++t) {                 // transposed conv: neighbour at q - (dy-2, dx-2); w: scalar loads
      const float g = win[k + 4 - t / 5][4 - t % 5];
      o0[k] = fmaf(w[t], g, o0[k]);
      o1[k] = fmaf(w[25 + t], g, o1[k]);
    }
    const int gy = y0 + r0 + k, gx = x0 + cx;
    if (gy < H && gx < W) {
      g_pooled[(long)b * 2 * HW + (long)gy * W + gx] = o0[k];
      g_pooled[(long)b * 2 * HW + HW + (long)gy * W + gx] = o1[k];
    }
  }
#pragma unroll
  for (int t = 0; t < 25; ++t) {
    const float p0 = wave_red63<false>(a0[t]), p1 = wave_red63<false>(a1[t]);      // DPP: 6 VALU each, result in lane 63
    if (lane == 63) { red[t][wave] = p0; red[25 + t][wave] = p1; }
  }
  __syncthreads();
  if (tid < 50) part_w[(long)blockIdx.x * 50 + tid] = (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]);
}

// ---- D -------------------------------------------------------------------------------------------
template <class P>
__global__ __launch_bounds__(256) void cac_bwd_apply_kernel(Sl<typename P::T> g_out, Sl<typename P::T> g_outc,
                                                            Sl<typename P::T> pre, Sl<typename P::T> pre_c,
                                                            const float* __restrict__ ch,
                                                            const float* __restrict__ sp,
                                                            const float* __restrict__ pooled,    // (B,2,H,W): max, mean
                                                            const float* __restrict__ g_pooled,  // (B,2,H,W)
                                                            const float* __restrict__ g_pools,   // (B,2,128): avg, max
                                                            const int* __restrict__ argpix,      // (B,128)
                                                            SlW<typename P::T> g_pre, SlW<typename P::T> g_pre_c,
                                                            SlW<typename P::T> g_in, SlW<typename P::T> g_in_c,
                                                            int accumulate_in, long HW, float inv_hw) {
  const int tid = threadIdx.x;
  const int tile = blockIdx.x, b = blockIdx.y;
  const long tile0 = (long)tile * BWD_TILE;
  int pidx[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) pidx[i] = (int)P::pix(tile0, tid, i);
  float spv[8], pmax[8], gpmax[8], gpmean[8];
  P::loadf(sp + (long)b * HW, tile0, tid, HW, spv);
  P::loadf(pooled + (long)b * 2 * HW, tile0, tid, HW, pmax);
  P::loadf(g_pooled + (long)b * 2 * HW, tile0, tid, HW, gpmax);
  P::loadf(g_pooled + (long)b * 2 * HW + HW, tile0, tid, HW, gpmean);
  bool done[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { done[i] = false; gpmean[i] *= (1.f / 128.f); }
  const float* gavg = g_pools + ((long)b * 2 + 0) * 128;
  const float* gmax = g_pools + ((long)b * 2 + 1) * 128;
  const int* ap = argpix + (long)b * 128;

#pragma unroll 1
  for (int fc = 0; fc < 128; ++fc) {  // Fcat order: colour first, so ties route like torch.max(dim=1)
    const bool colour = fc < 64;
    const int c = fc & 63;
    const Sl<typename P::T>& go_s = colour ? g_outc : g_out;
    const Sl<typename P::T>& pr_s = colour ? pre_c : pre;
    const SlW<typename P::T>& gp_s = colour ? g_pre_c : g_pre;
    const SlW<typename P::T>& gi_s = colour ? g_in_c : g_in;
    float go[8], p[8], gi[8];
    P::load(go_s.p + b * go_s.img + c * HW, tile0, tid, HW, go);
    P::load(pr_s.p + b * pr_s.img + c * HW, tile0, tid, HW, p);
    if (accumulate_in) P::load(gi_s.p + b * gi_s.img + c * HW, tile0, tid, HW, gi);
    const float chc = ch[b * 64 + c];
    const float ga = gavg[fc] * inv_hw, gm = gmax[fc];
    const int apx = ap[fc];
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float v = go[i] * (chc * spv[i]) + ga + gpmean[i];
      if (pidx[i] == apx) v += gm;
      const bool hit = !done[i] && p[i] == pmax[i];
      if (hit) v += gpmax[i];
      done[i] = done[i] || hit;
      o[i] = v;
      gi[i] = accumulate_in ? gi[i] + go[i] : go[i];
    }
    P::store(gp_s.p + b * gp_s.img + c * HW, tile0, tid, HW, o);
    P::store(gi_s.p + b * gi_s.img + c * HW, tile0, tid, HW, gi);
  }
}

// ---- elementwise: dst = [dst +] src, then dst = mask > 0 ? dst : 0 ---------------------------------
template <class P>
__global__ __launch_bounds__(256) void ew_add_mask_kernel(typename P::T* __restrict__ dst, long d_img,
                                                          const typename P::T* __restrict__ src, long s_img,
                                                          const typename P::T* __restrict__ mask, long m_img, int C,
                                                          long HW, int accumulate) {
  const int bc = blockIdx.y;
  const int b = bc / C, c = bc % C;
  const int tid = threadIdx.x;
  const long tile0 = (long)blockIdx.x * PX_TILE;
  float v[8], t[8];
  typename P::T* d = dst + b * d_img + c * HW;
  if (accumulate || !src) P::load(d, tile0, tid, HW, v);
  else {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.f;
  }
  if (src) {
    P::load(src + b * s_img + c * HW, tile0, tid, HW, t);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += t[i];
  }
  if (mask) {
    P::load(mask + b * m_img + c * HW, tile0, tid, HW, t);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = t[i] > 0.f ? v[i] : 0.f;
  }
  P::store(d, tile0, tid, HW, v);
}

// ---- host ------------------------------------------------------------------------------------------
// ew_c8.hip: passes A and D and the elementwise helper over channel-blocked 16-bit tensors
int cac_bwd_reduce_c8(int, int, int, const codon_tensor*, const codon_tensor*, const codon_tensor*, const codon_tensor*,
                      const float*, const float*, const float*, float*, float*, int*, int, hipStream_t, const float*, int*,
                      const codon_tensor*, const codon_tensor*, int);
int cac_bwd_apply_c8(int, int, int, const codon_tensor*, const codon_tensor*, const codon_tensor*, const codon_tensor*,
                     const float*, const float*, const float*, const float*, const float*, const int*, const codon_tensor*,
                     const codon_tensor*, const codon_tensor*, const codon_tensor*, int, int, hipStream_t);
int ew_add_mask_c8(int, int, int, int, const codon_tensor*, const codon_tensor*, const codon_tensor*, int, int, hipStream_t);
int ew_sum_mask_c8(int, int, int, int, const codon_tensor*, int, const codon_tensor* const*, const codon_tensor*, int, hipStream_t);
static bool al16(const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr) {
  return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
           reinterpret_cast<uintptr_t>(d)) % 16) == 0;
}
template <class T>
static Sl<T> mk(const codon_tensor* t, long HW) { return Sl<T>{(const T*)t->data + t->coff * HW, t->ctotal * HW}; }
template <class T>
static SlW<T> mkw(const codon_tensor* t, long HW) { return SlW<T>{(T*)t->data + t->coff * HW, t->ctotal * HW}; }
static const char* basep(const codon_tensor* t, long HW, int dtype) {
  return (const char*)t->data + t->coff * HW * 4;
}

int cac_bwd_tiles(int H, int W) { return (int)(((long)H * W + BWD_TILE - 1) / BWD_TILE); }
int cac_bwd_spatial_blocks(int B, int H, int W) { return B * ((H + SPB_T - 1) / SPB_T) * ((W + SPB_T - 1) / SPB_T); }

int cac_bwd_reduce(int B, int H, int W, const codon_tensor* g_out, const codon_tensor* g_outc,
                   const codon_tensor* pre, const codon_tensor* pre_c, const float* ch, const float* sp,
                   const float* pools, float* g_z, float* part_gch, int* part_arg, int dtype, hipStream_t stream) {
  if (dtype != CODON_F32)
    return cac_bwd_reduce_c8(B, H, W, g_out, g_outc, pre, pre_c, ch, sp, pools, g_z, part_gch, part_arg, dtype, stream,
                             nullptr, nullptr, nullptr, nullptr, 0);
  const long HW = (long)H * W;
  const int nt = cac_bwd_tiles(H, W);
  const bool al = al16(basep(g_out, HW, dtype), basep(g_outc, HW, dtype), basep(pre, HW, dtype),
                       basep(pre_c, HW, dtype)) && al16(sp, g_z);
  px_dispatch(dtype, HW, al, [&](auto pol) {
    using P = decltype(pol);
    using T = typename P::T;
    hipLaunchKernelGGL(cac_bwd_reduce_kernel<P>, dim3(nt, B), dim3(256), 0, stream, mk<T>(g_out, HW),
                       mk<T>(g_outc, HW), mk<T>(pre, HW), mk<T>(pre_c, HW), ch, sp, pools, g_z, part_gch, part_arg, HW,
                       nt);
  });
  return check_launch("cac_bwd_reduce_kernel");
}

// 16-bit tensors only: pass A that also records the per-pixel arg-max channel and folds dL/d(out) into dL/d(inputs)
int cac_bwd_reduce_acc(int B, int H, int W, const codon_tensor* g_out, const codon_tensor* g_outc, const codon_tensor* pre,
                       const codon_tensor* pre_c, const float* ch, const float* sp, const float* pools, const float* pooled,
                       float* g_z, float* part_gch, int* part_arg, int* argch, const codon_tensor* g_in,
                       const codon_tensor* g_in_c, int accumulate_in, int dtype, hipStream_t stream) {
  return cac_bwd_reduce_c8(B, H, W, g_out, g_outc, pre, pre_c, ch, sp, pools, g_z, part_gch, part_arg, dtype, stream, pooled,
                           argch, g_in, g_in_c, accumulate_in);
}

int cac_bwd_gate(int B, int H, int W, const float* part_gch, const int* part_arg, const float* ch,
                 const float* pools, const float* w1, const float* b1, const float* w2, float* g_pools, int* argpix,
                 float* part_param, float* dw1, float* db1, float* dw2, float* db2, hipStream_t stream) {
  const int nt = cac_bwd_tiles(H, W);
  hipLaunchKernelGGL(cac_bwd_gate_kernel, dim3(B), dim3(128 * GATE_SLICES), 0, stream, part_gch, part_arg, ch, pools, w1, b1, w2,
                     g_pools, argpix, part_param, nt);
  int st = check_launch("cac_bwd_gate_kernel");
  if (st != CODON_OK || !dw1) return st;       // dw1 == null: the rows stay in part_param for codon_reduce_multi
  // the four gradients are contiguous slices of the per-image partial rows; sum over images, fixed order
  struct { float* out; int off, n; } parts[4] = {{dw1, 0, 1024}, {db1, 1024, 8}, {dw2, 1032, 512}, {db2, 1544, 64}};
  for (auto& q : parts) {
    hipLaunchKernelGGL(partial_sum_kernel, dim3((q.n + 255) / 256), dim3(256), 0, stream, part_param + q.off, q.out,
                       q.n, B, (long)GATE_NPARAM, 0);
    st = check_launch("partial_sum_kernel");
    if (st != CODON_OK) return st;
  }
  return CODON_OK;
}

int cac_bwd_spatial(int B, int H, int W, const float* g_z, const float* pooled, const float* w, float* g_pooled,
                    float* part_w, float* dw, hipStream_t stream) {
  const int nblk = cac_bwd_spatial_blocks(B, H, W);
  hipLaunchKernelGGL(cac_bwd_spatial_kernel, dim3(nblk), dim3(256), 0, stream, g_z, pooled, w, g_pooled, part_w, H, W,
                     (W + SPB_T - 1) / SPB_T, (H + SPB_T - 1) / SPB_T);
  int st = check_launch("cac_bwd_spatial_kernel");
  if (st != CODON_OK || !dw) return st;        // dw == null: the rows stay in part_w for codon_reduce_multi
  // two-level fixed-order sum of the (nblk, 50) partials: 64 chunks in place, then the chunk sums
  const int nchunk = nblk < 64 ? 1 : 64;
  const int per = (nblk + nchunk - 1) / nchunk;
  if (nchunk > 1) {
    hipLaunchKernelGGL(partial_chunk_sum_kernel, dim3(nchunk), dim3(64), 0, stream, part_w, 50, nblk, per);
    st = check_launch("partial_chunk_sum_kernel");
    if (st != CODON_OK) return st;
    // only the chunks that HOLD rows: with nblk not a multiple of 64 the last chunks start past the last row (round 5: the
    // second stage used to walk all 64 and read past part_w -- e.g. 300 blocks for one 480 x 640 image: per 5, 60 chunks)
    const int used = (nblk + per - 1) / per;
    hipLaunchKernelGGL(partial_sum_kernel, dim3(1), dim3(256), 0, stream, part_w, dw, 50, used, 50L * per, 0);
  } else {
    hipLaunchKernelGGL(partial_sum_kernel, dim3(1), dim3(256), 0, stream, part_w, dw, 50, nblk, 50L, 0);
  }
  return check_launch("partial_sum_kernel");
}

int cac_bwd_apply(int B, int H, int W, const codon_tensor* g_out, const codon_tensor* g_outc,
                  const codon_tensor* pre, const codon_tensor* pre_c, const float* ch, const float* sp,
                  const float* pooled, const float* g_pooled, const float* g_pools, const int* argpix,
                  const codon_tensor* g_pre, const codon_tensor* g_pre_c, const codon_tensor* g_in,
                  const codon_tensor* g_in_c, int accumulate_in, int dtype, hipStream_t stream) {
  if (dtype != CODON_F32)
    return cac_bwd_apply_c8(B, H, W, g_out, g_outc, pre, pre_c, ch, sp, pooled, g_pooled, g_pools, argpix, g_pre, g_pre_c,
                            g_in, g_in_c, accumulate_in, dtype, stream);
  const long HW = (long)H * W;
  const int nt = cac_bwd_tiles(H, W);
  const float inv = (float)(1.0 / (double)HW);
  const bool al = al16(basep(g_out, HW, dtype), basep(g_outc, HW, dtype), basep(pre, HW, dtype),
                       basep(pre_c, HW, dtype)) &&
                  al16(basep(g_pre, HW, dtype), basep(g_pre_c, HW, dtype), basep(g_in, HW, dtype),
                       basep(g_in_c, HW, dtype)) && al16(sp, pooled, g_pooled);
  px_dispatch(dtype, HW, al, [&](auto pol) {
    using P = decltype(pol);
    using T = typename P::T;
    hipLaunchKernelGGL(cac_bwd_apply_kernel<P>, dim3(nt, B), dim3(256), 0, stream, mk<T>(g_out, HW),
                       mk<T>(g_outc, HW), mk<T>(pre, HW), mk<T>(pre_c, HW), ch, sp, pooled, g_pooled, g_pools, argpix,
                       mkw<T>(g_pre, HW), mkw<T>(g_pre_c, HW), mkw<T>(g_in, HW), mkw<T>(g_in_c, HW), accumulate_in,
                       HW, inv);
  });
  return check_launch("cac_bwd_apply_kernel");
}

int ew_add_mask(int B, int H, int W, int C, const codon_tensor* dst, const codon_tensor* src,
                const codon_tensor* mask, int accumulate, int dtype, hipStream_t stream) {
  if (dtype != CODON_F32) return ew_add_mask_c8(B, H, W, C, dst, src, mask, accumulate, dtype, stream);
  const long HW = (long)H * W;
  CODON_REQUIRE((long)B * C <= 65535, CODON_ERR_UNSUPPORTED, "ew_add_mask: batch*channels too large");
  const char* d = basep(dst, HW, dtype);
  const char* s = src ? basep(src, HW, dtype) : nullptr;
  const char* m = mask ? basep(mask, HW, dtype) : nullptr;
  const long s_img = src ? src->ctotal * HW : 0, m_img = mask ? mask->ctotal * HW : 0;
  const unsigned nt = (unsigned)((HW + PX_TILE - 1) / PX_TILE);
  px_dispatch(dtype, HW, al16(d, s, m), [&](auto pol) {
    using P = decltype(pol);
    using T = typename P::T;
    hipLaunchKernelGGL(ew_add_mask_kernel<P>, dim3(nt, B * C), dim3(256), 0, stream, (T*)d, dst->ctotal * HW,
                       (const T*)s, s_img, (const T*)m, m_img, C, HW, accumulate);
  });
  return check_launch("ew_add_mask_kernel");
}

// dst = mask > 0 ? srcs[0] + ... + srcs[nsrc-1] : 0 (1 <= nsrc <= 4; dst aliases no source).  16-bit tensors: one pass, fp32
// sum rounded once.  fp32 tensors: the same left-to-right sum as a copy and nsrc - 1 accumulating passes of ew_add_mask.
int ew_sum_mask(int B, int H, int W, int C, const codon_tensor* dst, int nsrc, const codon_tensor* const* srcs,
                const codon_tensor* mask, int dtype, hipStream_t stream) {
  if (dtype != CODON_F32) return ew_sum_mask_c8(B, H, W, C, dst, nsrc, srcs, mask, dtype, stream);
  CODON_REQUIRE(nsrc >= 1 && nsrc <= 4, CODON_ERR_UNSUPPORTED, "ew_sum_mask: %d sources (1..4)", nsrc);
  for (int i = 0; i < nsrc; ++i) {
    const int st = ew_add_mask(B, H, W, C, dst, srcs[i], i + 1 == nsrc ? mask : nullptr, i > 0 ? 1 : 0, dtype, stream);
    if (st != CODON_OK) return st;
  }
  return CODON_OK;
}

}  // namespace codon
