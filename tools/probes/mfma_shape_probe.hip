// Probe: does the bf16 MFMA shape change delivered FLOP/s on THIS MI355X under a conv-like operand pattern?
// (MI355X_MICROARCH.md 'DVFS give-back' item 7: 16x16x32 delivered 1.12-1.15x the FLOP/s of 32x32x16 at equal cycles.)
// Each wave owns a 128 (M) x 64 (N) fp32 tile: 32x32x16 -> 4x2 tiles of 16 regs; 16x16x32 -> 8x4 tiles of 4 regs.
// Per K = 32 step both variants read the same 12 x 16 B fragments per lane from LDS (random bf16 data) -- the
// conv5x5-128 kernel's 0.75 ds_read_b128 per 32x32x16 MFMA.  256 threads, 2 workgroups per CU, like the conv kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_shape_probe.hip -o /tmp/mfma_shape_probe && /tmp/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KSTEPS = 64;      // K = 32 steps per "stage" (all served from one 48 KB LDS image, re-read)
constexpr int STAGES = 200;

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void probe(const uint4* __restrict__ src, float* __restrict__ out) {
  __shared__ uint4 lds[3072];   // 48 KB
  for (int i = threadIdx.x; i < 3072; i += 256) lds[i] = src[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint4* base = lds + wave * 64 + lane;
  if constexpr (SHAPE == 32) {
    f32x16 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll 1
    for (int s = 0; s < STAGES; ++s) {
#pragma unroll 4
      for (int k = 0; k < KSTEPS; ++k) {
        bf16x8 a[2][4], b[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int i = 0; i < 4; ++i) { const uint4 v = base[((k * 12 + h * 6 + i) * 256) % 2816]; a[h][i] = *reinterpret_cast<const bf16x8*>(&v); }
#pragma unroll
          for (int j = 0; j < 2; ++j) { const uint4 v = base[((k * 12 + h * 6 + 4 + j) * 256) % 2816]; b[h][j] = *reinterpret_cast<const bf16x8*>(&v); }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[h][i], b[h][j], acc[i][j], 0, 0, 0);
      }
    }
    float t = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) t += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = t;
  } else if constexpr (SHAPE == 17) {
    // 16x16x32 as the conv5x5-128 kernel could use it without growing its LDS: the 128 couts are visited in two
    // halves per K = 32 step, each re-reading the 4 pixel fragments (8 reads per 16 MFMAs instead of 12 per 32)
    f32x4 acc[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
#pragma unroll 1
    for (int s = 0; s < STAGES; ++s) {
#pragma unroll 4
      for (int k = 0; k < KSTEPS; ++k) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          bf16x8 a[4], b[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) { const uint4 v = base[((k * 12 + hh * 4 + i) * 256) % 2816]; a[i] = *reinterpret_cast<const bf16x8*>(&v); }
#pragma unroll
          for (int j = 0; j < 4; ++j) { const uint4 v = base[((k * 12 + 8 + j) * 256) % 2816]; b[j] = *reinterpret_cast<const bf16x8*>(&v); }
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[hh * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[hh * 4 + i][j], 0, 0, 0);
        }
      }
    }
    float t = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) t += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = t;
  } else {
    f32x4 acc[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
#pragma unroll 1
    for (int s = 0; s < STAGES; ++s) {
#pragma unroll 4
      for (int k = 0; k < KSTEPS; ++k) {
        bf16x8 a[8], b[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) { const uint4 v = base[((k * 12 + i) * 256) % 2816]; a[i] = *reinterpret_cast<const bf16x8*>(&v); }
#pragma unroll
        for (int j = 0; j < 4; ++j) { const uint4 v = base[((k * 12 + 8 + j) * 256) % 2816]; b[j] = *reinterpret_cast<const bf16x8*>(&v); }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    float t = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) t += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = t;
  }
}

int main() {
  const int nblk = 512 * 8;
  std::vector<unsigned short> h(3072 * 8);
  srand(1);
  for (auto& v : h) { float f = (rand() / (float)RAND_MAX) * 2.f - 1.f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  uint4* src; float* out;
  hipMalloc(&src, 3072 * 16); hipMalloc(&out, nblk * 256 * 4);
  hipMemcpy(src, h.data(), 3072 * 16, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double flop = (double)nblk * 4 /*waves*/ * STAGES * KSTEPS * 2.0 * 128 * 64 * 32;
  for (int rep = 0; rep < 3; ++rep) {
    for (int shape : {32, 16, 17}) {
      for (int w = 0; w < 2; ++w) {   // warm, then timed (>= 0.3 s of back-to-back work so the clock settles)
        hipEventRecord(e0);
        for (int it = 0; it < 4; ++it) {
          if (shape == 32) hipLaunchKernelGGL(probe<32>, dim3(nblk), dim3(256), 0, 0, src, out);
          else if (shape == 16) hipLaunchKernelGGL(probe<16>, dim3(nblk), dim3(256), 0, 0, src, out);
          else hipLaunchKernelGGL(probe<17>, dim3(nblk), dim3(256), 0, 0, src, out);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (w) printf("shape %s: %.2f ms  %.0f TFLOP/s\n", shape == 32 ? "32x32x16" : shape == 16 ? "16x16x32" : "16x16x32, cout halves (8 reads / 16 MFMAs)", ms / 4, flop / (ms / 4 * 1e-3) / 1e12);
      }
    }
  }
  return 0;
}
