// Channel-blocked ("C8") layout of the 16-bit activations and activation gradients inside the network.
//
//   element (b, c, h, w) of a (B, C, H, W) tensor lives at   data[(((b*C/8 + c/8)*H + h)*W + w)*8 + c%8]
//
// i.e. [B][C/8][H][W][8]: one 16-byte vector = 8 consecutive channels of one pixel.  Only the 64/128-channel tensors
// between the stem and the head take this form; the 1-channel maps that cross the module boundary (x, y, the output:
// CODON_x4.py:66-68,130-132), the pooled maps and the gates stay plain fp32 NCHW.  What it buys on gfx950:
//   * the MFMA B operand of a 16-bit conv is 8 consecutive input channels of one pixel (v_mfma_f32_32x32x16_*), so the
//     halo tile is staged with straight 16-byte global -> LDS copies: no 2-byte loads, no pack VALU, no W % 4 / W % 8
//     fast-path conditions (the reference's 463 / 447 / 343-wide images take the same path as 640);
//   * with the weight packer permuting the cout rows of every 32-row MFMA tile (swap23 below) a lane's accumulator
//     registers 8g .. 8g+7 are 8 CONSECUTIVE output channels of its pixel: the epilogue is two 16-byte stores per tile
//     instead of sixteen 2-byte ones, and residual / mask / accumulate operands are 16-byte loads;
//   * every elementwise / CAC kernel reads and writes whole 16-byte vectors at any H, W.
// A channel slice (ctotal, coff) of a wider buffer is a range of 8-channel planes: coff % 8 == 0.
#pragma once
#include <hip/hip_bf16.h>

#include "codon_common.h"

namespace codon {

typedef __bf16 c8_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 c8_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr unsigned C8_OOB = 0xFFFFFFF0u;     // buffer offset no descriptor of ours reaches: loads return 0, stores are dropped
constexpr int C8_RSRC_FLAGS = 0x00020000;    // raw buffer, 32-bit offsets, bounds-checked

// D layout of a 32x32 MFMA tile: lane (col = l & 31, half = l >> 5), register r holds row (r&3) + 8(r>>2) + 4*half.
// If A row i carries cout tile_base + swap23(i), then register r of half h holds cout tile_base + 16(r>>3) + 8h + (r&7):
// registers 8g .. 8g+7 = the 8 channels of plane (tile_base/8 + 2g + h).
__host__ __device__ __forceinline__ constexpr int swap23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

// 16-bit element traits: storage is raw 16 bits; E picks the MFMA opcode and the conversions
struct C8Bf16 {
  typedef c8_bf16x8 vec8;
  __device__ static f32x16 mfma(vec8 a, vec8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
  __device__ static float lo(unsigned w) { return __uint_as_float(w << 16); }
  __device__ static float hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
  __device__ static unsigned pack2(float a, float b) {   // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    bf2 v;
    v[0] = (__bf16)a;
    v[1] = (__bf16)b;
    return *reinterpret_cast<const unsigned*>(&v);
  }
};
struct C8F16 {
  typedef c8_f16x8 vec8;
  __device__ static f32x16 mfma(vec8 a, vec8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
  __device__ static float lo(unsigned w) {
    const unsigned short s = (unsigned short)(w & 0xffffu);
    return (float)*reinterpret_cast<const _Float16*>(&s);
  }
  __device__ static float hi(unsigned w) {
    const unsigned short s = (unsigned short)(w >> 16);
    return (float)*reinterpret_cast<const _Float16*>(&s);
  }
  __device__ static unsigned pack2(float a, float b) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 v;
    v[0] = (_Float16)a;
    v[1] = (_Float16)b;
    return *reinterpret_cast<const unsigned*>(&v);
  }
};

template <class E>
__device__ __forceinline__ void c8_unpack(const u32x4 q, float (&v)[8]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[2 * j] = E::lo(q[j]);
    v[2 * j + 1] = E::hi(q[j]);
  }
}
template <class E>
__device__ __forceinline__ u32x4 c8_pack(const float (&v)[8]) {
  u32x4 q;
#pragma unroll
  for (int j = 0; j < 4; ++j) q[j] = E::pack2(v[2 * j], v[2 * j + 1]);
  return q;
}

__device__ __forceinline__ u32x4 c8_ld(const __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  return *reinterpret_cast<const u32x4*>(&v);
}
// HARDWARE HAZARD (measured on MI355X, round 3): a buffer_store_dwordx4 whose data VGPRs are overwritten by the VALU
// instruction right behind it stores the NEW value in lanes 12-15 of every 16 (the store reads its 128 data bits over
// several cycles).  hipcc pads this "store of more than 64 bits, then VALU write of the data registers" case with an
// s_nop only when the store has no SGPR soffset (it takes the SGPR form to be exempt); every plane offset here IS an
// SGPR soffset, and `buffer_store_dwordx4 v[74:77], .., s14 offen ; v_pk_add_f32 v[74:75], ..` corrupted channels
// 2, 3 of a plane in exactly those lanes (conv 3x3 64->64, mask + accumulate epilogue; tests/test_gpu_c8.py).  The two
// wait states are pinned behind the store; they cost nothing beside 16 bytes per lane of memory traffic.
__device__ __forceinline__ void c8_st(const u32x4 q, const __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  typedef decltype(__builtin_amdgcn_raw_buffer_load_b128(r, 0u, 0u, 0)) raw_t;
  __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const raw_t*>(&q), r, voff, soff, 0);   // default cache policy (nt / sc0 / sc1 stores measured in round 4: no gain)
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 1");
  __builtin_amdgcn_sched_barrier(0);
}

// host side: a C8 slice argument.  planes = 8-channel planes per image in the buffer, plane0 = first plane of the slice
static inline bool c8_slice_ok(int ctotal, int coff, int c) {
  return ctotal % 8 == 0 && coff % 8 == 0 && c % 8 == 0 && coff >= 0 && coff + c <= ctotal;
}

}  // namespace codon
