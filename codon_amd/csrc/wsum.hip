// Stale-packed-weight guard: ONE launch per forward folds a position-dependent 64-bit checksum of the raw bytes of every
// MFMA conv weight (17 tensors, 7.4 MB in fp32) and compares it with the checksum taken when the packed images were
// built.  A write that the host-side cache key (data_ptr, Tensor._version) cannot see -- `w.data.normal_()`, the
// reference's own init idiom, /root/reference/CODON_X4/CODON_x4.py:50-53 -- sets a sticky flag in host-visible memory;
// codon_amd.model raises on the next call or synchronisation point instead of serving stale weights silently.
//
// The checksum is a sum of 64-bit terms (integer addition: any order gives the same value, so the grid may reduce in
// any shape): term(i, w) = (w + 0x9E3779B9) * (2 i + 1) * 0x9E3779B97F4A7C15 mod 2^64 for the 32-bit word w at global word
// index i.  Changing one word by d != 0 changes the sum by d * odd * odd != 0 mod 2^64: every single-word change is
// detected, and the odd position factor makes swaps visible.
#include "codon_common.h"

namespace codon {

constexpr int WS_BLOCKS = 1024, WS_THREADS = 256;   // 262 144 threads: <= 2 vectors each for the 7.4 MB of fp32 weights

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct WsumArgs {
  int n;
  unsigned nvec[CODON_WSUM_MAX + 1];      // prefix sums of 16-byte vectors per tensor
  const u32x4* data[CODON_WSUM_MAX];
};

__device__ __forceinline__ unsigned long long wsum_term(unsigned w, unsigned long long i) {
  return ((unsigned long long)w + 0x9E3779B9ull) * (2ull * i + 1ull) * 0x9E3779B97F4A7C15ull;
}

__global__ __launch_bounds__(WS_THREADS) void wsum_kernel(const WsumArgs a, unsigned long long* __restrict__ ws,
                                                           unsigned long long* __restrict__ ref, int mode,
                                                           int* __restrict__ flag) {
  __shared__ unsigned long long red[WS_THREADS / 64];
  __shared__ bool last;
  const unsigned total = a.nvec[a.n];
  unsigned long long acc = 0;
  for (unsigned q = blockIdx.x * WS_THREADS + threadIdx.x; q < total; q += WS_BLOCKS * WS_THREADS) {
    int t = 0;
    while (q >= a.nvec[t + 1]) ++t;       // <= 17 scalar-table compares
    const u32x4 v = __builtin_nontemporal_load(a.data[t] + (q - a.nvec[t]));
    const unsigned long long i = 4ull * q;
    acc += wsum_term(v.x, i) + wsum_term(v.y, i + 1) + wsum_term(v.z, i + 2) + wsum_term(v.w, i + 3);
  }
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned lo = __shfl_xor((unsigned)acc, o), hi = __shfl_xor((unsigned)(acc >> 32), o);
    acc += ((unsigned long long)hi << 32) | lo;
  }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long s = 0;
    for (int k = 0; k < WS_THREADS / 64; ++k) s += red[k];
    ws[1 + blockIdx.x] = s;
    __threadfence();
    last = atomicAdd(&ws[0], 1ull) == WS_BLOCKS - 1;     // ws[0]: arrival counter, zero between launches
  }
  __syncthreads();
  if (!last) return;
  // the last block to arrive folds the per-block sums (every one of them is visible: fence above, acquire below)
  __threadfence();
  unsigned long long s = 0;
  for (int k = threadIdx.x; k < WS_BLOCKS; k += WS_THREADS) s += __atomic_load_n(&ws[1 + k], __ATOMIC_RELAXED);
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned lo = __shfl_xor((unsigned)s, o), hi = __shfl_xor((unsigned)(s >> 32), o);
    s += ((unsigned long long)hi << 32) | lo;
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    s = 0;
    for (int k = 0; k < WS_THREADS / 64; ++k) s += red[k];
    ws[0] = 0;                                            // ready for the next launch on this workspace
    ws[1 + WS_BLOCKS] = s;                                // last value seen (diagnostics)
    if (mode == 0) {
      *ref = s;
    } else if (s != *ref) {
      __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // host-visible, sticky
    }
  }
}

size_t weight_checksum_workspace_bytes() { return (size_t)(WS_BLOCKS + 2) * sizeof(unsigned long long); }

int weight_checksum(const codon_wsum_desc* d, void* ws, unsigned long long* ref, int mode, int* flag, hipStream_t s) {
  WsumArgs a;
  a.n = d->n;
  a.nvec[0] = 0;
  for (int t = 0; t < d->n; ++t) {
    a.data[t] = (const u32x4*)d->data[t];
    a.nvec[t + 1] = a.nvec[t] + (unsigned)(d->bytes[t] / 16);
  }
  for (int t = d->n; t < CODON_WSUM_MAX; ++t) {
    a.data[t] = nullptr;
    a.nvec[t + 1] = a.nvec[d->n];
  }
  hipLaunchKernelGGL(wsum_kernel, dim3(WS_BLOCKS), dim3(WS_THREADS), 0, s, a, (unsigned long long*)ws, ref, mode, flag);
  return check_launch("weight_checksum");
}

}  // namespace codon
