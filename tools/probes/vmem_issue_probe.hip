// Probe: what does a vector-memory instruction cost the CU by access width?  Every wave streams a small
// (L2/L1-resident) buffer with 16 loads per iteration of width 2 / 4 / 16 bytes per lane; reports wave-instructions
// per microsecond per CU and bytes/ns/CU.  Question behind it: are the 16-bit conv kernels (2-byte buffer loads
// and stores, 8 per 16-byte LDS element) bound by vector-memory ISSUE rather than by bytes?
#include <hip/hip_runtime.h>
#include <cstdio>

template <int WIDTH>
__global__ __launch_bounds__(256) void probe(const unsigned char* __restrict__ src, unsigned* __restrict__ out, int iters) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1 << 20, 0x00020000);
  const unsigned lane_off = threadIdx.x * WIDTH;            // a wave reads 64 * WIDTH contiguous bytes per instruction
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned so = (unsigned)((it * 7 + blockIdx.x) & 63) * 4096u;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if constexpr (WIDTH == 2) acc += __builtin_amdgcn_raw_buffer_load_b16(rs, lane_off, so + k * 512u, 0);
      else if constexpr (WIDTH == 4) acc += __builtin_amdgcn_raw_buffer_load_b32(rs, lane_off, so + k * 1024u, 0);
      else { const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off, so + k * 4096u, 0); acc += v[0] ^ v[1] ^ v[2] ^ v[3]; }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int WIDTH>
__global__ __launch_bounds__(256) void probe_st(unsigned char* __restrict__ dst, int iters) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(dst + (size_t)blockIdx.x * (1 << 20)), 0, 1 << 20, 0x00020000);
  const unsigned lane_off = threadIdx.x * WIDTH;
  for (int it = 0; it < iters; ++it) {
    const unsigned so = (unsigned)((it * 7) & 15) * 16384u;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if constexpr (WIDTH == 2) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)it, rs, lane_off, so + k * 512u, 0);
      else if constexpr (WIDTH == 4) __builtin_amdgcn_raw_buffer_store_b32((unsigned)it, rs, lane_off, so + k * 1024u, 0);
      else { typedef unsigned u4 __attribute__((ext_vector_type(4))); u4 v = {(unsigned)it, 1u, 2u, 3u}; __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane_off, so + k * 1024u, 0); }
    }
  }
}

int main() {
  unsigned char* src; unsigned* out; unsigned char* dst;
  const int nblk = 256 * 8, iters = 2000;
  hipMalloc(&src, 2 << 20); hipMemset(src, 1, 2 << 20); hipMalloc(&out, nblk * 256 * 4);
  hipMalloc(&dst, (size_t)nblk << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch, int width, int its) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ninstr = (double)nblk * 4 * its * 16;                      // wave-instructions
    printf("%s width %2d B/lane: %.2f ms  %.1f wave-instr/us/CU = %.2f ns each per CU;  %.1f B/ns/CU\n", name, width, ms,
           ninstr / (ms * 1e3) / 256, ms * 1e6 / (ninstr / 256), ninstr * 64 * width / (ms * 1e6) / 256);
  };
  run("load ", [&] { hipLaunchKernelGGL(probe<2>, dim3(nblk), dim3(256), 0, 0, src, out, iters); }, 2, iters);
  run("load ", [&] { hipLaunchKernelGGL(probe<4>, dim3(nblk), dim3(256), 0, 0, src, out, iters); }, 4, iters);
  run("load ", [&] { hipLaunchKernelGGL(probe<16>, dim3(nblk), dim3(256), 0, 0, src, out, iters); }, 16, iters);
  run("store", [&] { hipLaunchKernelGGL(probe_st<2>, dim3(nblk), dim3(256), 0, 0, dst, iters / 4); }, 2, iters / 4);
  run("store", [&] { hipLaunchKernelGGL(probe_st<4>, dim3(nblk), dim3(256), 0, 0, dst, iters / 4); }, 4, iters / 4);
  run("store", [&] { hipLaunchKernelGGL(probe_st<16>, dim3(nblk), dim3(256), 0, 0, dst, iters / 4); }, 16, iters / 4);
  return 0;
}
