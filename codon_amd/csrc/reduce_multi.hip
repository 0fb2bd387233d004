// One launch for every deferred fixed-order reduction of a backward pass (codon_reduce_multi).
//
// The weight gradients of a training step (autograd of /root/reference/CODON_X4/CODON_x4.py:66-132; the reference has no
// explicit backward) leave the kernels that produce them as per-split partials; each one used to be
// followed by its own small reduce launch (56 wgrad_reduce + 30 partial_sum / chunk_sum + 3 conv1ch reduces per step,
// 10 - 50 us each and latency-bound).  Here all of them are items of ONE launch, and the result can be ADDED straight into
// the caller's gradient buffer (the flat all-reduce buffer of codon_amd.dist.GradSync): no per-tensor add afterwards.
//
// Item kinds (block-uniform):
//   wgrad : out[co][ci][tap] (+)= r_0 + r_1 + ... over the item's `nuse` workspaces (the 5 / 3 uses of a shared weight, in
//           the order they were produced), r_u = sum_s part_u[s][tap][co][ci], s serial from zero -- bit for bit what
//           nuse calls of wgrad_reduce_kernel (accumulate on all but the first) leave.  16-byte loads, 8 splits in flight.
//   rows  : out[f(i)] (+)= sum_k part[k * stride + i], two-level when nchunk > 1: chunk c = rows [c per, (c+1) per) summed
//           serially, then the chunk sums in order (the order of partial_chunk_sum + partial_sum, and of
//           conv1ch_wgrad_reduce with nchunk = 16); f = identity, or the 3x3 tap flip of the head's weight gradient.

#include "codon_common.h"

namespace codon {

struct ReduceMultiArgs {
  codon_reduce_item it[CODON_REDUCE_MAX_ITEMS];
  int first_block[CODON_REDUCE_MAX_ITEMS + 1];
  int n;
};
static_assert(sizeof(ReduceMultiArgs) <= CODON_KERNARG_LIMIT, "passed by value: one more field must not push the launch past the kernel-argument limit");

__global__ __launch_bounds__(256) void reduce_multi_kernel(const ReduceMultiArgs a) {
  __shared__ float red[256];
  int k = 0;
  while (k + 1 < a.n && (int)blockIdx.x >= a.first_block[k + 1]) ++k;     // block-uniform
  const codon_reduce_item& it = a.it[k];
  const int blk = (int)blockIdx.x - a.first_block[k];
  const bool acc = it.flags & CODON_REDUCE_ACCUMULATE;
  if (it.flags & CODON_REDUCE_WGRAD) {
    const int cin = it.cin, cout = it.cout, taps = it.taps, nsplit = it.nparts;
    const long n = (long)cout * cin * taps;
    const long i = (blk * 256L + threadIdx.x) * 4;   // index in [tap][co][ci] order (coalesced reads); cin % 4 == 0
    if (i >= n) return;
    const int ci = (int)(i % cin);
    const long t = i / cin;
    const int co = (int)(t % cout);
    const int tap = (int)(t / cout);
    float* o = it.out + ((long)co * cin + ci) * taps + tap;
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = acc ? o[(long)j * taps] : 0.f;
    for (int u = 0; u < it.nuse; ++u) {
      const float* __restrict__ ws = it.part[u];
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      int q = 0;
      for (; q + 8 <= nsplit; q += 8) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4*>(ws + (long)(q + j) * n + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; }
      }
      for (; q < nsplit; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(ws + (long)q * n + i);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      if (u == 0 && !acc) { r[0] = s.x; r[1] = s.y; r[2] = s.z; r[3] = s.w; }
      else { r[0] += s.x; r[1] += s.y; r[2] += s.z; r[3] += s.w; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) o[(long)j * taps] = r[j];
    return;
  }
  // rows: 256 threads = nchunk chunks x (256 / nchunk) columns
  const int nchunk = it.nchunk, cols = 256 / nchunk, n = it.cin;
  const int c = threadIdx.x / cols, li = threadIdx.x % cols;
  const int i = blk * cols + li;
  const int per = (it.nparts + nchunk - 1) / nchunk;
  const int p0 = c * per, p1 = min(p0 + per, it.nparts);
  const float* __restrict__ part = it.part[0];
  float s = 0.f;
  if (i < n) {
    int p = p0;
    for (; p + 8 <= p1; p += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = part[(long)(p + j) * it.stride + i];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; p < p1; ++p) s += part[(long)p * it.stride + i];
  }
  if (nchunk > 1) {
    red[threadIdx.x] = s;
    __syncthreads();
    if (c != 0) return;
    // chunks past the last row hold 0.f: adding them leaves the sum unchanged, so the walk may cover all nchunk
    // (partial_chunk_sum's order when nparts is not a multiple: its empty chunks were never summed either)
    const int used = (it.nparts + per - 1) / per;
    s = red[li];
    for (int g = 1; g < used; ++g) s += red[g * cols + li];
  }
  if (i >= n) return;
  const int oi = (it.flags & CODON_REDUCE_FLIP9) ? (i / 9) * 9 + 8 - (i % 9) : i;
  it.out[oi] = acc ? it.out[oi] + s : s;
}

int reduce_multi(const codon_reduce_item* items, int n_items, hipStream_t stream) {
  for (int base = 0; base < n_items; base += CODON_REDUCE_MAX_ITEMS) {
    ReduceMultiArgs a;
    a.n = n_items - base < CODON_REDUCE_MAX_ITEMS ? n_items - base : CODON_REDUCE_MAX_ITEMS;
    int nb = 0;
    for (int k = 0; k < a.n; ++k) {
      const codon_reduce_item& it = items[base + k];
      CODON_REQUIRE(it.out && it.part[0] && it.nparts >= 1 && it.cin >= 1, CODON_ERR_BAD_ARG, "reduce_multi: item %d: null pointer or empty", base + k);
      a.it[k] = it;
      a.first_block[k] = nb;
      if (it.flags & CODON_REDUCE_WGRAD) {
        CODON_REQUIRE(it.nuse >= 1 && it.nuse <= CODON_REDUCE_MAX_USES && it.cout >= 1 && it.taps >= 1 && it.cin % 4 == 0,
                      CODON_ERR_BAD_ARG, "reduce_multi: item %d: nuse %d, cout %d, cin %d, taps %d", base + k, it.nuse, it.cout,
                      it.cin, it.taps);
        for (int u = 0; u < it.nuse; ++u)
          CODON_REQUIRE(it.part[u] && ((uintptr_t)it.part[u] % 16) == 0, CODON_ERR_BAD_ARG,
                        "reduce_multi: item %d: workspace %d null or not 16-byte aligned", base + k, u);
        const long n = (long)it.cout * it.cin * it.taps;
        nb += (int)((n / 4 + 255) / 256);
      } else {
        CODON_REQUIRE(it.nchunk == 1 || it.nchunk == 16 || it.nchunk == 64, CODON_ERR_BAD_ARG,
                      "reduce_multi: item %d: nchunk %d (1, 16 or 64)", base + k, it.nchunk);
        const int cols = 256 / it.nchunk;
        nb += (it.cin + cols - 1) / cols;
      }
    }
    a.first_block[a.n] = nb;
    hipLaunchKernelGGL(reduce_multi_kernel, dim3(nb), dim3(256), 0, stream, a);
    const int st = check_launch("reduce_multi_kernel");
    if (st != CODON_OK) return st;
  }
  return CODON_OK;
}

}  // namespace codon
