// "8 pixels per thread" access policies shared by the HBM-bound kernels (cac*.hip).
// A workgroup of 256 threads owns a tile of 2048 consecutive pixels of one image plane; a thread owns
// 8 of them.  Which 8 depends on the element type so that every load/store is the widest coalesced
// access the layout allows:
//   PxF32V : fp32, HW % 4 == 0, 16-B aligned planes : two float4 at  tile0 + j*1024 + tid*4      (j = 0,1)
//   PxB16V : bf16, HW % 8 == 0, 16-B aligned planes : one 16-byte vector at tile0 + tid*8
//   PxF32S / PxB16S : any HW : eight scalars at tile0 + j*256 + tid
// All arithmetic is fp32; bf16 is a storage format (round-to-nearest-even on store).
#pragma once
#include <hip/hip_runtime.h>

namespace codon {

constexpr int PX_TILE = 2048;
typedef unsigned short u16_t;   // bf16 storage
struct h16_t { unsigned short bits; };   // fp16 storage (distinct C++ type so overloads can tell them apart)

__device__ __forceinline__ float b2f(u16_t v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ u16_t f2b(float f) {
  __bf16 b = (__bf16)f;
  return *reinterpret_cast<u16_t*>(&b);
}

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

struct PxB16V {
  typedef u16_t T;
  // fp32 side planes (pooled maps, gates) with the SAME pixel ownership: 8 consecutive floats = 2 x float4
  __device__ static void loadf(const float* pl, long tile0, int tid, long HW, float (&v)[8]) {
    const long p = tile0 + tid * 8;
    const bool ok = p < HW;
    const float4 a = ok ? *reinterpret_cast<const float4*>(pl + p) : make_float4(0, 0, 0, 0);
    const float4 b = ok ? *reinterpret_cast<const float4*>(pl + p + 4) : make_float4(0, 0, 0, 0);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  __device__ static void storef(float* pl, long tile0, int tid, long HW, const float (&v)[8]) {
    const long p = tile0 + tid * 8;
    if (p < HW) {
      *reinterpret_cast<float4*>(pl + p) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(pl + p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
  }
  __device__ static long pix(long tile0, int tid, int i) { return tile0 + tid * 8 + i; }
  __device__ static void load(const u16_t* pl, long tile0, int tid, long HW, float (&v)[8]) {
    const long p = tile0 + tid * 8;
    const uint4 q = p < HW ? *reinterpret_cast<const uint4*>(pl + p) : make_uint4(0, 0, 0, 0);
    v[0] = __uint_as_float(q.x << 16); v[1] = __uint_as_float(q.x & 0xffff0000u);
    v[2] = __uint_as_float(q.y << 16); v[3] = __uint_as_float(q.y & 0xffff0000u);
    v[4] = __uint_as_float(q.z << 16); v[5] = __uint_as_float(q.z & 0xffff0000u);
    v[6] = __uint_as_float(q.w << 16); v[7] = __uint_as_float(q.w & 0xffff0000u);
  }
  __device__ static void store(u16_t* pl, long tile0, int tid, long HW, const float (&v)[8]) {
    const long p = tile0 + tid * 8;
    if (p < HW) {
      uint4 q;
      q.x = (unsigned)f2b(v[0]) | ((unsigned)f2b(v[1]) << 16);
      q.y = (unsigned)f2b(v[2]) | ((unsigned)f2b(v[3]) << 16);
      q.z = (unsigned)f2b(v[4]) | ((unsigned)f2b(v[5]) << 16);
      q.w = (unsigned)f2b(v[6]) | ((unsigned)f2b(v[7]) << 16);
      *reinterpret_cast<uint4*>(pl + p) = q;
    }
  }
};

struct PxB16S {
  typedef u16_t T;
  __device__ static void loadf(const float* pl, long tile0, int tid, long HW, float (&v)[8]) {
    PxF32S::load(pl, tile0, tid, HW, v);
  }
  __device__ static void storef(float* pl, long tile0, int tid, long HW, const float (&v)[8]) {
    PxF32S::store(pl, tile0, tid, HW, v);
  }
  __device__ static long pix(long tile0, int tid, int i) { return tile0 + (long)i * 256 + tid; }
  __device__ static void load(const u16_t* pl, long tile0, int tid, long HW, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long p = tile0 + (long)i * 256 + tid;
      v[i] = p < HW ? b2f(pl[p]) : 0.f;
    }
  }
  __device__ static void store(u16_t* pl, long tile0, int tid, long HW, const float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long p = tile0 + (long)i * 256 + tid;
      if (p < HW) pl[p] = f2b(v[i]);
    }
  }
};

// 16-bit conversions by storage type
__device__ __forceinline__ float h2f(unsigned short v) { return (float)*reinterpret_cast<const _Float16*>(&v); }
__device__ __forceinline__ unsigned short f2h(float f) {
  const _Float16 h = (_Float16)f;
  return *reinterpret_cast<const unsigned short*>(&h);
}
struct CvtB16 {
  typedef u16_t T;
  __device__ static float lo(unsigned w) { return __uint_as_float(w << 16); }
  __device__ static float hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
  __device__ static float one(u16_t v) { return b2f(v); }
  __device__ static unsigned short pack(float f) { return f2b(f); }
};
struct CvtH16 {
  typedef h16_t T;
  __device__ static float lo(unsigned w) { return h2f((unsigned short)(w & 0xffffu)); }
  __device__ static float hi(unsigned w) { return h2f((unsigned short)(w >> 16)); }
  __device__ static float one(h16_t v) { return h2f(v.bits); }
  __device__ static unsigned short pack(float f) { return f2h(f); }
};

template <class C>
struct Px16V {
  typedef typename C::T T;
  __device__ static long pix(long tile0, int tid, int i) { return tile0 + tid * 8 + i; }
  __device__ static void loadf(const float* pl, long tile0, int tid, long HW, float (&v)[8]) {
    PxB16V::loadf(pl, tile0, tid, HW, v);
  }
  __device__ static void storef(float* pl, long tile0, int tid, long HW, const float (&v)[8]) {
    PxB16V::storef(pl, tile0, tid, HW, v);
  }
  __device__ static void load(const T* pl, long tile0, int tid, long HW, float (&v)[8]) {
    const long p = tile0 + tid * 8;
    const uint4 q = p < HW ? *reinterpret_cast<const uint4*>(pl + p) : make_uint4(0, 0, 0, 0);
    v[0] = C::lo(q.x); v[1] = C::hi(q.x); v[2] = C::lo(q.y); v[3] = C::hi(q.y);
    v[4] = C::lo(q.z); v[5] = C::hi(q.z); v[6] = C::lo(q.w); v[7] = C::hi(q.w);
  }
  __device__ static void store(T* pl, long tile0, int tid, long HW, const float (&v)[8]) {
    const long p = tile0 + tid * 8;
    if (p < HW) {
      uint4 q;
      q.x = (unsigned)C::pack(v[0]) | ((unsigned)C::pack(v[1]) << 16);
      q.y = (unsigned)C::pack(v[2]) | ((unsigned)C::pack(v[3]) << 16);
      q.z = (unsigned)C::pack(v[4]) | ((unsigned)C::pack(v[5]) << 16);
      q.w = (unsigned)C::pack(v[6]) | ((unsigned)C::pack(v[7]) << 16);
      *reinterpret_cast<uint4*>(pl + p) = q;
    }
  }
};
template <class C>
struct Px16S {
  typedef typename C::T T;
  __device__ static long pix(long tile0, int tid, int i) { return tile0 + (long)i * 256 + tid; }
  __device__ static void loadf(const float* pl, long tile0, int tid, long HW, float (&v)[8]) {
    PxF32S::load(pl, tile0, tid, HW, v);
  }
  __device__ static void storef(float* pl, long tile0, int tid, long HW, const float (&v)[8]) {
    PxF32S::store(pl, tile0, tid, HW, v);
  }
  __device__ static void load(const T* pl, long tile0, int tid, long HW, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long p = tile0 + (long)i * 256 + tid;
      v[i] = p < HW ? C::one(pl[p]) : 0.f;
    }
  }
  __device__ static void store(T* pl, long tile0, int tid, long HW, const float (&v)[8]) {
    const unsigned short* dummy = nullptr; (void)dummy;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long p = tile0 + (long)i * 256 + tid;
      if (p < HW) *reinterpret_cast<unsigned short*>(pl + p) = C::pack(v[i]);
    }
  }
};

// Host-side dispatch: calls fn(Policy{}) with the widest policy the shapes/pointers allow.
template <typename F>
static inline void px_dispatch(int dtype, long HW, bool aligned16, F&& fn) {
  if (dtype == 1 /* CODON_BF16 */) {
    if (HW % 8 == 0 && aligned16) fn(PxB16V{});
    else fn(PxB16S{});
  } else if (dtype == 2 /* CODON_F16 */) {
    if (HW % 8 == 0 && aligned16) fn(Px16V<CvtH16>{});
    else fn(Px16S<CvtH16>{});
  } else {
    if (HW % 4 == 0 && aligned16) fn(PxF32V{});
    else fn(PxF32S{});
  }
}

}  // namespace codon
