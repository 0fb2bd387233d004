// Shared host-side helpers for libcodon_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "codon_hip.h"

namespace codon {

void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

// Kernel arguments travel in a 4 KiB kernarg segment on gfx950: structs passed BY VALUE to a kernel assert against this where
// they are defined, so that a field added later fails the build, not the launch.
#define CODON_KERNARG_LIMIT 4096

// Call after a kernel launch; converts a sticky launch error into CODON_ERR_LAUNCH.
int check_launch(const char* what);

#define CODON_REQUIRE(cond, code, ...)       \
  do {                                       \
    if (!(cond)) {                           \
      ::codon::set_error(__VA_ARGS__);       \
      return (code);                         \
    }                                        \
  } while (0)

// Bijective XCD-aware block remap (cdna_hip_programming.md T1): blocks b and b+8 share an
// XCD under the observed round-robin dispatch, so give each XCD a contiguous range of tiles
// and neighbouring tiles (which share halos / weights) hit the same 4 MiB L2.  Speed only.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
  const unsigned q = nblk >> 3, r = nblk & 7u, xcd = bid & 7u, k = bid >> 3;
  const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + k;
}

// Debug builds only (make EXTRA=-DCODON_TIMING, tools/exp_timing.py): per-workgroup phase timestamps (100 MHz
// wall clock) written to the buffer whose address is in the environment variable CODON_DBG_PTR at launch time.
#ifdef CODON_TIMING
#define CODON_TSTAMP(dbg_, k_)                                                                          \
  if (threadIdx.x == 0 && (dbg_)) {                                                                     \
    (dbg_)[(long)blockIdx.x * 8 + (k_)] = (long long)wall_clock64();                                    \
    if ((k_) == 0) { /* where it ran: HW_ID (wave/simd/cu/sh/se) and XCC_ID, for per-CU timelines */    \
      (dbg_)[(long)blockIdx.x * 8 + 6] = (long long)__builtin_amdgcn_s_getreg(0xF804);                  \
      (dbg_)[(long)blockIdx.x * 8 + 7] = (long long)__builtin_amdgcn_s_getreg(0xF814);                  \
    }                                                                                                   \
  }
inline long long* codon_dbg_ptr() {
  const char* e = getenv("CODON_DBG_PTR");
  return e ? (long long*)strtoull(e, nullptr, 16) : nullptr;
}
#else
#define CODON_TSTAMP(dbg_, k_)
#endif

// Wave-wide sum / max by DPP (row rotations inside the 16-lane rows, then the two row broadcasts): six VALU instructions, the
// result in LANE 63.  __shfl_xor goes through ds_bpermute -- 1536 dependent LDS-crossbar round trips per thread for the 128
// channels of a pixel took 25 of this kernel's 37 us.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_take(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <bool MAX>
__device__ __forceinline__ float wave_red63(float v) {
  auto op = [](float a, float b) { return MAX ? fmaxf(a, b) : a + b; };
  v = op(v, dpp_take<0xB1>(v));            // quad_perm [1,0,3,2]
  v = op(v, dpp_take<0x4E>(v));            // quad_perm [2,3,0,1]
  v = op(v, dpp_take<0x124>(v));           // row_ror:4
  v = op(v, dpp_take<0x128>(v));           // row_ror:8   -> every lane holds its row's result
  v = op(v, dpp_take<0x142, 0xa>(v));      // row_bcast:15 into rows 1, 3
  v = op(v, dpp_take<0x143, 0xc>(v));      // row_bcast:31 into rows 2, 3 -> lane 63 holds the wave's
  return v;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

}  // namespace codon
