// "8 pixels per thread" access policies shared by the HBM-bound fp32 (NCHW) kernels (cac*.hip).
// A workgroup of 256 threads owns a tile of 2048 consecutive pixels of one image plane; a thread owns
// 8 of them, chosen so that every load/store is the widest coalesced access the layout allows:
//   PxF32V : HW % 4 == 0, 16-B aligned planes : two float4 at  tile0 + j*1024 + tid*4      (j = 0,1)
//   PxF32S : any HW : eight scalars at tile0 + j*256 + tid
// 16-bit activations are channel-blocked (c8.h) and have their own kernels (ew_c8.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace codon {

constexpr int PX_TILE = 2048;

struct PxF32V {
  typedef float T;
  __device__ static void loadf(const float* pl, long tile0, int tid, long HW, float (&v)[8]) { load(pl, tile0, tid, HW, v); }
  __device__ static void storef(float* pl, long tile0, int tid, long HW, const float (&v)[8]) { store(pl, tile0, tid, HW, v); }
  __device__ static long pix(long tile0, int tid, int i) { return tile0 + (long)(i >> 2) * 1024 + tid * 4 + (i & 3); }
  __device__ static void load(const float* pl, long tile0, int tid, long HW, float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const long p = tile0 + (long)j * 1024 + tid * 4;
      const float4 q = p < HW ? *reinterpret_cast<const float4*>(pl + p) : make_float4(0, 0, 0, 0);
      v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
    }
  }
  __device__ static void store(float* pl, long tile0, int tid, long HW, const float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const long p = tile0 + (long)j * 1024 + tid * 4;
      if (p < HW) *reinterpret_cast<float4*>(pl + p) = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
    }
  }
};

struct PxF32S {
  typedef float T;
  __device__ static void loadf(const float* pl, long tile0, int tid, long HW, float (&v)[8]) { load(pl, tile0, tid, HW, v); }
  __device__ static void storef(float* pl, long tile0, int tid, long HW, const float (&v)[8]) { store(pl, tile0, tid, HW, v); }
  __device__ static long pix(long tile0, int tid, int i) { return tile0 + (long)i * 256 + tid; }
  __device__ static void load(const float* pl, long tile0, int tid, long HW, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long p = tile0 + (long)i * 256 + tid;
      v[i] = p < HW ? pl[p] : 0.f;
    }
  }
  __device__ static void store(float* pl, long tile0, int tid, long HW, const float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long p = tile0 + (long)i * 256 + tid;
      if (p < HW) pl[p] = v[i];
    }
  }
};

// Host-side dispatch: calls fn(Policy{}) with the widest policy the shapes/pointers allow.
template <typename F>
static inline void px_dispatch(int dtype, long HW, bool aligned16, F&& fn) {
  (void)dtype;   // fp32 only: 16-bit tensors are channel-blocked and take the kernels of ew_c8.hip
  if (HW % 4 == 0 && aligned16) fn(PxF32V{});
  else fn(PxF32S{});
}

}  // namespace codon
