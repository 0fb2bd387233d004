// extern "C" surface of libcodon_hip.so (see include/codon_hip.h): argument validation, dtype
// dispatch, error strings.  Never throws, never allocates device memory, never synchronises.

#include <stdarg.h>
#include <string.h>

#include "codon_common.h"
#include "pair.h"

namespace codon {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return CODON_ERR_LAUNCH;
  }
  return CODON_OK;
}

// kernels (defined in the other translation units)
int conv_ck(int ks);
int conv2d_fwd_f32(const codon_conv_desc*, const float*, const float*, float*, const float*, hipStream_t);
size_t conv_wgrad_workspace_bytes(const codon_conv_desc*);
int conv2d_wgrad_f32(const codon_conv_desc*, const float*, const float*, float*, float*, size_t, int, hipStream_t);
size_t conv_wgrad_bf16_workspace_bytes(const codon_conv_desc*);
int conv2d_wgrad_bf16(const codon_conv_desc*, const void*, const void*, float*, float*, size_t, int, hipStream_t);
int pack_weight_f32(const float*, float*, int, int, int, int, hipStream_t);
int pack_chain1x1_f32(const float*, float*, hipStream_t);
int conv2d_gated_fwd_f32(const codon_conv_desc*, const float*, const codon_tensor*, const float*, const float*, const float*,
                         float*, const codon_tensor*, hipStream_t);
int pack_chain1x1_16(const float*, void*, int, hipStream_t);
int pack_chain1x1_f32x3(const float*, void*, hipStream_t);
int conv_chain1x1_fwd_f32x3(const codon_conv_desc*, const float*, const void*, float*, const void*, const codon_tensor*,
                            const codon_tensor*, hipStream_t);
int conv_chain1x1_fwd_16(const codon_conv_desc*, const void*, const void*, void*, const void*, const codon_tensor*,
                         const codon_tensor*, float*, float*, int, hipStream_t);
int conv2d_gated_fwd_16(const codon_conv_desc*, const void*, const codon_tensor*, const float*, const float*, const void*,
                        void*, const codon_tensor*, hipStream_t);
struct Conv1x1GateBwd {
  const float* ch; const float* sp; const float* g_pooled; const float* g_pools; const int* argpix; const int* argch;
  int fbase;
};
int conv1x1_bwd_16(const codon_conv_desc*, const void*, const void*, const void*, const codon_tensor*, float*, float*, size_t,
                   int, hipStream_t, const Conv1x1GateBwd*);
int cac_bwd_reduce_acc(int, int, int, const codon_tensor*, const codon_tensor*, const codon_tensor*, const codon_tensor*,
                       const float*, const float*, const float*, const float*, float*, float*, int*, int*, const codon_tensor*,
                       const codon_tensor*, int, int, hipStream_t);
int cac_fused_tiles(int, int);
int cac_tail_fwd(int, int, int, int, const float*, const float*, const float*, const float*, float*, float*, int*, const float*,
                 const float*, const float*, const float*, const float*, float*, float*, float*, hipStream_t);
int cac_fused_finish(int, int, int, int, const float*, const float*, const float*, float*, float*, hipStream_t);
int cac_gate_fwd_n(int, int, float, const float*, const float*, const float*, const float*, const float*, float*, float*,
                   hipStream_t);
int conv_tiling_f32(const codon_conv_desc*, int, int);
int conv_chain1x1_fwd_f32(const codon_conv_desc*, const float*, const float*, float*, const float*, const codon_tensor*,
                          const codon_tensor*, float*, float*, int, hipStream_t);
bool conv_f32x3_supported(const codon_conv_desc*);
int conv2d_fwd_f32x3(const codon_conv_desc*, const float*, const void*, float*, const float*, hipStream_t);
int pack_weight_f32x3(const float*, void*, int, int, int, hipStream_t);
int stem_fwd(int, int, int, const float*, const float*, void*, int, int, int, const void*, int, int, int, hipStream_t);
int stem_pair_fwd(int, int, int, const float*, const float*, void*, int, int, const float*, const float*, void*, int, int, int,
                  hipStream_t);
size_t conv1ch_wgrad_workspace_bytes(int, int, int);
int conv1ch_wgrad(int, int, int, const void*, int, int, const float*, float*, int, float*, size_t, int, hipStream_t);
int cac_bwd_tiles(int, int);
int cac_bwd_spatial_blocks(int, int, int);
int cac_bwd_reduce(int, int, int, const codon_tensor*, const codon_tensor*, const codon_tensor*, const codon_tensor*,
                   const float*, const float*, const float*, float*, float*, int*, int, hipStream_t);
int cac_bwd_gate(int, int, int, const float*, const int*, const float*, const float*, const float*, const float*,
                 const float*, float*, int*, float*, float*, float*, float*, float*, hipStream_t);
int cac_bwd_spatial(int, int, int, const float*, const float*, const float*, float*, float*, float*, hipStream_t);
int cac_bwd_apply(int, int, int, const codon_tensor*, const codon_tensor*, const codon_tensor*, const codon_tensor*,
                  const float*, const float*, const float*, const float*, const float*, const int*,
                  const codon_tensor*, const codon_tensor*, const codon_tensor*, const codon_tensor*, int, int,
                  hipStream_t);
int ew_add_mask(int, int, int, int, const codon_tensor*, const codon_tensor*, const codon_tensor*, int, int,
                hipStream_t);
int ew_sum_mask(int, int, int, int, const codon_tensor*, int, const codon_tensor* const*, const codon_tensor*, int, hipStream_t);
int head_fwd(int, int, int, const void*, int, int, const float*, const float*, void*, bool, int, hipStream_t);
int cac_stats_tiles(int, int);
int cac_stats_fwd(int, int, int, const codon_tensor*, const codon_tensor*, float*, float*, int, hipStream_t, const float*);
int ew_sq_scale(int, int, int, const codon_tensor*, const float*, const codon_tensor*, int, hipStream_t);
int conv2d_fwd_bf16(const codon_conv_desc*, const void*, const void*, void*, const void*, hipStream_t);
int conv2d_sum_into_16(const codon_conv_desc*, const void*, const void*, void*, void*, hipStream_t);
int pack_weight_bf16(const float*, void*, int, int, int, int, int, hipStream_t);
int cac_gate_fwd(int, int, int, const float*, const float*, const float*, const float*, const float*, float*, float*,
                 hipStream_t);
int cac_spatial_fwd(int, int, int, const float*, const float*, float*, hipStream_t);
int cac_apply_fwd(int, int, int, const codon_tensor*, const codon_tensor*, const float*, const float*,
                  const codon_tensor*, const codon_tensor*, const codon_tensor*, const codon_tensor*, int, hipStream_t);

int bicubic_upsample(int, int, int, int, const float*, const float*, float*, hipStream_t);

int postprocess_u8(const float*, unsigned char*, long, hipStream_t);
int postprocess_u8_f16(const void*, unsigned char*, long, hipStream_t);
int masked_sqerr(const unsigned char*, const unsigned char*, long, unsigned long long*, hipStream_t);
int ssim_tiles(int, int, int);
int ssim_fwd(int, int, int, const float*, const float*, float*, float*, double*, hipStream_t);
int l1_fwd(long, const float*, const float*, float*, int, double*, hipStream_t);
int ssim_l1_bwd(int, int, int, const float*, const float*, const float*, float*, float*, float, float, hipStream_t);

int reduce_multi(const codon_reduce_item*, int, hipStream_t);
int cast_multi(const codon_cast_desc*, float*, hipStream_t);
int adam_step(const codon_adam_desc*, const float*, float*, float*, float, float, float, float, float, int, hipStream_t);
size_t weight_checksum_workspace_bytes();
int weight_checksum(const codon_wsum_desc*, void*, unsigned long long*, int, int*, hipStream_t);

static thread_local PairRecorder g_pair;
PairRecorder* pair_recorder() { return g_pair.active ? &g_pair : nullptr; }

static bool slice_ok(const codon_tensor* t) { return t && t->data && t->coff >= 0 && t->coff + 64 <= t->ctotal; }

static bool shape_ok(int b, int h, int w) { return b > 0 && h > 0 && w > 0 && (long)h * w < (1L << 31); }

}  // namespace codon

using namespace codon;

extern "C" {

int codon_abi_version(void) { return CODON_ABI_VERSION; }

const char* codon_last_error_string(void) { return g_err; }

#ifndef CODON_SOURCE_HASH
#define CODON_SOURCE_HASH "unknown"
#endif
const char* codon_build_source_hash(void) { return CODON_SOURCE_HASH; }

size_t codon_conv_packed_weight_bytes(int32_t cout, int32_t cin, int32_t ksize, int32_t dtype) {
  if (cout <= 0 || cin <= 0 || (ksize != 1 && ksize != 3 && ksize != 5)) return 0;
  return (size_t)cout * cin * ksize * ksize * (dtype == CODON_F32 ? 4 : 2);
}

int codon_conv_pack_weight(const float* w_oihw, void* w_packed, int32_t cout, int32_t cin, int32_t ksize,
                           int32_t mode, int32_t dtype, codon_stream_t stream) {
  CODON_REQUIRE(w_oihw && w_packed, CODON_ERR_BAD_ARG, "conv_pack_weight: null pointer");
  CODON_REQUIRE(ksize == 1 || ksize == 3 || ksize == 5, CODON_ERR_UNSUPPORTED, "conv_pack_weight: ksize %d", ksize);
  if (mode == CODON_PACK_FWD_F16X3) {
    CODON_REQUIRE(dtype == CODON_F32 && (ksize == 3 || ksize == 5) && cin % 16 == 0 && cout % 64 == 0,
                  CODON_ERR_UNSUPPORTED, "conv_pack_weight: f16x3 packing needs fp32, k in {3,5}, cin%%16==0, cout%%64==0");
    return pack_weight_f32x3(w_oihw, w_packed, cout, cin, ksize, (hipStream_t)stream);
  }
  if (mode == CODON_PACK_CHAIN1X1 || mode == CODON_PACK_CHAIN1X1_F16X3) {
    CODON_REQUIRE(ksize == 1 && cin == 128 && cout == 64, CODON_ERR_UNSUPPORTED,
                  "conv_pack_weight: CHAIN1X1 packs the (64,128,1,1) weights only");
    if (mode == CODON_PACK_CHAIN1X1_F16X3) {
      CODON_REQUIRE(dtype == CODON_F32, CODON_ERR_UNSUPPORTED, "conv_pack_weight: CHAIN1X1_F16X3 is for fp32 tensors");
      return pack_chain1x1_f32x3(w_oihw, w_packed, (hipStream_t)stream);
    }
    if (dtype == CODON_BF16 || dtype == CODON_F16) return pack_chain1x1_16(w_oihw, w_packed, dtype, (hipStream_t)stream);
    CODON_REQUIRE(dtype == CODON_F32, CODON_ERR_UNSUPPORTED, "conv_pack_weight: CHAIN1X1 dtype %d", dtype);
    return pack_chain1x1_f32(w_oihw, (float*)w_packed, (hipStream_t)stream);
  }
  CODON_REQUIRE(mode == CODON_PACK_FWD || mode == CODON_PACK_DGRAD, CODON_ERR_BAD_ARG, "conv_pack_weight: mode %d", mode);
  const int kin = mode == CODON_PACK_DGRAD ? cout : cin;
  CODON_REQUIRE(cout > 0 && cin > 0 && kin % conv_ck(ksize) == 0, CODON_ERR_UNSUPPORTED,
                "conv_pack_weight: cin=%d cout=%d not a multiple of the channel chunk", cin, cout);
  if (dtype == CODON_BF16 || dtype == CODON_F16) {
    CODON_REQUIRE(kin % 16 == 0, CODON_ERR_UNSUPPORTED, "conv_pack_weight: 16-bit packing needs cin %% 16 == 0");
    return pack_weight_bf16(w_oihw, w_packed, cout, cin, ksize, mode, dtype, (hipStream_t)stream);
  }
  CODON_REQUIRE(dtype == CODON_F32, CODON_ERR_UNSUPPORTED, "conv_pack_weight: dtype %d", dtype);
  return pack_weight_f32(w_oihw, (float*)w_packed, cout, cin, ksize, mode, (hipStream_t)stream);
}

int codon_conv2d_fwd(const codon_conv_desc* d, const void* x, const void* w_packed, void* y, const void* residual,
                     codon_stream_t stream) {
  CODON_REQUIRE(d && x && w_packed && y, CODON_ERR_BAD_ARG, "conv2d_fwd: null pointer");
  CODON_REQUIRE(shape_ok(d->batch, d->height, d->width), CODON_ERR_BAD_ARG, "conv2d_fwd: bad shape %dx%dx%d",
                d->batch, d->height, d->width);
  CODON_REQUIRE(d->x_coff >= 0 && d->x_coff + d->cin <= d->x_ctotal, CODON_ERR_BAD_ARG,
                "conv2d_fwd: input slice [%d,%d) outside %d channels", d->x_coff, d->x_coff + d->cin, d->x_ctotal);
  CODON_REQUIRE(d->y_coff >= 0 && d->y_coff + d->cout <= d->y_ctotal, CODON_ERR_BAD_ARG,
                "conv2d_fwd: output slice [%d,%d) outside %d channels", d->y_coff, d->y_coff + d->cout, d->y_ctotal);
  // only the public CODON_CONV_* bits: the kernels keep internal selector bits above them (conv_c8.hip: a read-write running
  // sum in the residual slot), which no caller of THIS entry point may reach -- `residual` is const here
  CODON_REQUIRE((d->flags & ~(CODON_CONV_RELU | CODON_CONV_ADD_RESIDUAL | CODON_CONV_ACCUM_OUT | CODON_CONV_MASK_RELU |
                              CODON_CONV_F16X3 | CODON_CONV_MASK_SUM)) == 0,
                CODON_ERR_BAD_ARG, "conv2d_fwd: unknown flag bits 0x%x", (unsigned)d->flags);
  CODON_REQUIRE(!((d->flags & CODON_CONV_ADD_RESIDUAL) && (d->flags & CODON_CONV_MASK_RELU)), CODON_ERR_BAD_ARG,
                "conv2d_fwd: ADD_RESIDUAL and MASK_RELU share the residual slot");
  if (d->flags & (CODON_CONV_ADD_RESIDUAL | CODON_CONV_MASK_RELU)) {
    CODON_REQUIRE(residual, CODON_ERR_BAD_ARG, "conv2d_fwd: ADD_RESIDUAL without a residual pointer");
    CODON_REQUIRE(d->r_coff >= 0 && d->r_coff + d->cout <= d->r_ctotal, CODON_ERR_BAD_ARG,
                  "conv2d_fwd: residual slice outside its buffer");
  }
  CODON_REQUIRE(!(d->flags & CODON_CONV_MASK_SUM) ||
                    ((d->flags & CODON_CONV_MASK_RELU) && (d->flags & CODON_CONV_ACCUM_OUT) && !(d->flags & (CODON_CONV_RELU | CODON_CONV_F16X3))),
                CODON_ERR_BAD_ARG, "conv2d_fwd: MASK_SUM needs MASK_RELU | ACCUM_OUT (and neither RELU nor F16X3)");
  CODON_REQUIRE(((uintptr_t)w_packed % 16) == 0, CODON_ERR_BAD_ARG, "conv2d_fwd: packed weights not 16-byte aligned");
  if (d->dtype == CODON_BF16 || d->dtype == CODON_F16)
    return conv2d_fwd_bf16(d, x, w_packed, y, residual, (hipStream_t)stream);
  CODON_REQUIRE(d->dtype == CODON_F32, CODON_ERR_UNSUPPORTED, "conv2d_fwd: dtype %d", d->dtype);
  if (d->flags & CODON_CONV_F16X3) {
    CODON_REQUIRE(conv_f32x3_supported(d), CODON_ERR_UNSUPPORTED, "conv2d_fwd: no f16x3 kernel for k=%d cin=%d cout=%d",
                  d->ksize, d->cin, d->cout);
    return conv2d_fwd_f32x3(d, (const float*)x, w_packed, (float*)y, (const float*)residual, (hipStream_t)stream);
  }
  return conv2d_fwd_f32(d, (const float*)x, (const float*)w_packed, (float*)y, (const float*)residual,
                        (hipStream_t)stream);
}

int codon_conv2d_sum_into_fwd(const codon_conv_desc* d, const void* x, const void* w_packed, void* y, void* sum,
                              codon_stream_t stream) {
  CODON_REQUIRE(d && x && w_packed && y && sum, CODON_ERR_BAD_ARG, "conv2d_sum_into_fwd: null pointer");
  CODON_REQUIRE(shape_ok(d->batch, d->height, d->width), CODON_ERR_BAD_ARG, "conv2d_sum_into_fwd: bad shape %dx%dx%d",
                d->batch, d->height, d->width);
  CODON_REQUIRE(d->x_coff >= 0 && d->x_coff + d->cin <= d->x_ctotal && d->y_coff >= 0 && d->y_coff + d->cout <= d->y_ctotal &&
                    d->r_coff >= 0 && d->r_coff + d->cout <= d->r_ctotal,
                CODON_ERR_BAD_ARG, "conv2d_sum_into_fwd: channel slice outside its buffer");
  CODON_REQUIRE(((uintptr_t)w_packed % 16) == 0, CODON_ERR_BAD_ARG, "conv2d_sum_into_fwd: packed weights not 16-byte aligned");
  CODON_REQUIRE(d->dtype == CODON_BF16 || d->dtype == CODON_F16, CODON_ERR_UNSUPPORTED,
                "conv2d_sum_into_fwd: 16-bit dtypes only (dtype %d)", d->dtype);
  // `sum` may be another channel slice of the buffer that holds y (or x), never an overlapping one
  auto disjoint = [](int a0, int an, int b0, int bn) { return a0 + an <= b0 || b0 + bn <= a0; };
  CODON_REQUIRE((sum != y || (d->r_ctotal == d->y_ctotal && disjoint(d->r_coff, d->cout, d->y_coff, d->cout))) &&
                    (sum != x || (d->r_ctotal == d->x_ctotal && disjoint(d->r_coff, d->cout, d->x_coff, d->cin))),
                CODON_ERR_BAD_ARG, "conv2d_sum_into_fwd: sum overlaps x or y");
  return conv2d_sum_into_16(d, x, w_packed, y, sum, (hipStream_t)stream);
}

int codon_conv_chain1x1_fwd(const codon_conv_desc* d, const void* x, const void* w_packed, void* y, const void* w_chain,
                            const codon_tensor* out, const codon_tensor* residual, codon_stream_t stream) {
  CODON_REQUIRE(d && x && w_packed && w_chain && out && out->data, CODON_ERR_BAD_ARG, "conv_chain1x1_fwd: null pointer");
  CODON_REQUIRE(shape_ok(d->batch, d->height, d->width), CODON_ERR_BAD_ARG, "conv_chain1x1_fwd: bad shape %dx%dx%d",
                d->batch, d->height, d->width);
  CODON_REQUIRE(d->x_coff >= 0 && d->x_coff + d->cin <= d->x_ctotal, CODON_ERR_BAD_ARG,
                "conv_chain1x1_fwd: input slice outside its buffer");
  CODON_REQUIRE(!y || (d->y_coff >= 0 && d->y_coff + d->cout <= d->y_ctotal), CODON_ERR_BAD_ARG,
                "conv_chain1x1_fwd: intermediate slice outside its buffer");
  CODON_REQUIRE(out->coff >= 0 && out->coff + 64 <= out->ctotal, CODON_ERR_BAD_ARG,
                "conv_chain1x1_fwd: output slice outside its buffer");
  CODON_REQUIRE(!residual || (residual->data && residual->coff >= 0 && residual->coff + 64 <= residual->ctotal),
                CODON_ERR_BAD_ARG, "conv_chain1x1_fwd: residual slice outside its buffer");
  CODON_REQUIRE((d->flags & ~(CODON_CONV_RELU | CODON_CONV_F16X3)) == 0, CODON_ERR_BAD_ARG,
                "conv_chain1x1_fwd: only RELU / F16X3 flags apply to the 5x5 stage");
  CODON_REQUIRE(((uintptr_t)w_packed % 16) == 0 && ((uintptr_t)w_chain % 16) == 0, CODON_ERR_BAD_ARG,
                "conv_chain1x1_fwd: packed weights not 16-byte aligned");
  if (d->dtype == CODON_BF16 || d->dtype == CODON_F16) {
    CODON_REQUIRE(!(d->flags & CODON_CONV_F16X3), CODON_ERR_BAD_ARG, "conv_chain1x1_fwd: F16X3 applies to fp32 tensors");
    return conv_chain1x1_fwd_16(d, x, w_packed, y, w_chain, out, residual, nullptr, nullptr, 0, (hipStream_t)stream);
  }
  CODON_REQUIRE(d->dtype == CODON_F32, CODON_ERR_UNSUPPORTED, "conv_chain1x1_fwd: dtype %d", d->dtype);
  if (d->flags & CODON_CONV_F16X3)
    return conv_chain1x1_fwd_f32x3(d, (const float*)x, w_packed, (float*)y, w_chain, out, residual, (hipStream_t)stream);
  return conv_chain1x1_fwd_f32(d, (const float*)x, (const float*)w_packed, (float*)y, (const float*)w_chain, out,
                               residual, nullptr, nullptr, 0, (hipStream_t)stream);
}

int codon_conv_chain1x1_stats_fwd(const codon_conv_desc* d, const void* x, const void* w_packed, void* y,
                                  const void* w_chain, const codon_tensor* out, const codon_tensor* residual,
                                  float* stats_pool, float* stats_partials, int32_t stats_choff, codon_stream_t stream) {
  CODON_REQUIRE(d && x && w_packed && w_chain && out && out->data && stats_pool && stats_partials, CODON_ERR_BAD_ARG,
                "conv_chain1x1_stats_fwd: null pointer");
  CODON_REQUIRE(shape_ok(d->batch, d->height, d->width), CODON_ERR_BAD_ARG, "conv_chain1x1_stats_fwd: bad shape %dx%dx%d",
                d->batch, d->height, d->width);
  CODON_REQUIRE(stats_choff == 0 || stats_choff == 64, CODON_ERR_BAD_ARG,
                "conv_chain1x1_stats_fwd: stats_choff %d (0 = colour stream, 64 = depth stream)", stats_choff);
  CODON_REQUIRE((d->flags & ~CODON_CONV_RELU) == 0, CODON_ERR_BAD_ARG, "conv_chain1x1_stats_fwd: only the RELU flag applies");
  CODON_REQUIRE(((uintptr_t)w_packed % 16) == 0 && ((uintptr_t)w_chain % 16) == 0 && ((uintptr_t)stats_partials % 8) == 0,
                CODON_ERR_BAD_ARG, "conv_chain1x1_stats_fwd: misaligned buffer");
  CODON_REQUIRE(d->x_coff >= 0 && d->x_coff + d->cin <= d->x_ctotal && (!y || (d->y_coff >= 0 && d->y_coff + d->cout <= d->y_ctotal)) &&
                    out->coff >= 0 && out->coff + 64 <= out->ctotal, CODON_ERR_BAD_ARG,
                "conv_chain1x1_stats_fwd: slice outside its buffer");
  if (d->dtype == CODON_F32) {
    // fp32 (round 6): per-row-strip partials, codon_cac_fused_parts(height, width, CODON_F32) rows per image
    CODON_REQUIRE(!residual, CODON_ERR_BAD_ARG, "conv_chain1x1_stats_fwd: fp32 statistics come without a residual");
    return conv_chain1x1_fwd_f32(d, (const float*)x, (const float*)w_packed, (float*)y, (const float*)w_chain, out, nullptr,
                                 stats_pool, stats_partials, stats_choff, (hipStream_t)stream);
  }
  CODON_REQUIRE(d->dtype == CODON_BF16 || d->dtype == CODON_F16, CODON_ERR_UNSUPPORTED, "conv_chain1x1_stats_fwd: dtype %d", d->dtype);
  return conv_chain1x1_fwd_16(d, x, w_packed, y, w_chain, out, residual, stats_pool, stats_partials, stats_choff,
                              (hipStream_t)stream);
}

int32_t codon_cac_fused_tiles(int32_t height, int32_t width) {
  return (height > 0 && width > 0) ? cac_fused_tiles(height, width) : 0;
}

int32_t codon_cac_fused_parts(int32_t height, int32_t width, int32_t dtype) {
  if (height <= 0 || width <= 0) return 0;
  if (dtype == CODON_F32) return height * ((width + 31) / 32);         // one row strip of 32 pixels each
  if (dtype == CODON_BF16 || dtype == CODON_F16) return cac_fused_tiles(height, width);
  return 0;
}

int codon_cac_fused_finish(int32_t batch, int32_t height, int32_t width, const float* partials, const float* pool_c,
                           const float* pool_d, float* folded, float* pooled, codon_stream_t stream) {
  CODON_REQUIRE(partials && pool_c && pool_d && folded && pooled, CODON_ERR_BAD_ARG, "cac_fused_finish: null pointer");
  CODON_REQUIRE(shape_ok(batch, height, width) && batch <= 65535, CODON_ERR_BAD_ARG, "cac_fused_finish: bad shape");
  return cac_fused_finish(batch, height, width, cac_fused_tiles(height, width), partials, pool_c, pool_d, folded, pooled,
                          (hipStream_t)stream);
}

int codon_cac_tail_fwd(int32_t batch, int32_t height, int32_t width, int32_t ntiles, const float* partials,
                       const float* pool_c, const float* pool_d, float* pooled, float* folded, int32_t* counters,
                       const float* w1, const float* b1, const float* w2, const float* b2, const float* w_spatial, float* ch,
                       float* pools_out, float* sp, codon_stream_t stream) {
  CODON_REQUIRE(partials && folded && counters && w1 && b1 && w2 && b2 && w_spatial && ch && sp, CODON_ERR_BAD_ARG,
                "cac_tail_fwd: null pointer");
  CODON_REQUIRE((pool_c != nullptr) == (pool_d != nullptr) && (pool_c || pooled), CODON_ERR_BAD_ARG,
                "cac_tail_fwd: pool_c and pool_d together, or pooled to read");
  CODON_REQUIRE(shape_ok(batch, height, width) && batch <= 65535, CODON_ERR_BAD_ARG, "cac_tail_fwd: bad shape");
  CODON_REQUIRE(pool_c ? (ntiles == cac_fused_tiles(height, width) || ntiles == codon_cac_fused_parts(height, width, CODON_F32))
                       : ntiles == cac_stats_tiles(height, width), CODON_ERR_BAD_ARG,
                "cac_tail_fwd: ntiles %d does not match the producer of the partials", ntiles);
  return cac_tail_fwd(batch, height, width, ntiles, partials, pool_c, pool_d, pool_c ? nullptr : pooled, pool_c ? pooled : nullptr,
                      folded, counters, w1, b1, w2, b2, w_spatial, ch, pools_out, sp, (hipStream_t)stream);
}

int codon_cac_gate_folded_fwd(int32_t batch, int32_t height, int32_t width, const float* folded, const float* w1,
                              const float* b1, const float* w2, const float* b2, float* ch, float* pools_out,
                              codon_stream_t stream) {
  CODON_REQUIRE(folded && w1 && b1 && w2 && b2 && ch, CODON_ERR_BAD_ARG, "cac_gate_folded_fwd: null pointer");
  CODON_REQUIRE(shape_ok(batch, height, width), CODON_ERR_BAD_ARG, "cac_gate_folded_fwd: bad shape");
  return cac_gate_fwd_n(batch, CODON_CAC_FOLDS, (float)(1.0 / ((double)height * width)), folded, w1, b1, w2, b2, ch,
                        pools_out, (hipStream_t)stream);
}

static int gated_fwd(const codon_conv_desc* d, const void* pre, const codon_tensor* inputs, const float* ch,
                     const float* sp, const void* w_packed, void* y, const codon_tensor* gated_out, codon_stream_t stream) {
  CODON_REQUIRE(d && pre && inputs && inputs->data && ch && sp && w_packed && y, CODON_ERR_BAD_ARG,
                "conv2d_gated_fwd: null pointer");
  CODON_REQUIRE(shape_ok(d->batch, d->height, d->width), CODON_ERR_BAD_ARG, "conv2d_gated_fwd: bad shape %dx%dx%d",
                d->batch, d->height, d->width);
  CODON_REQUIRE(d->x_coff >= 0 && d->x_coff + d->cin <= d->x_ctotal && d->x_coff % 64 == 0, CODON_ERR_BAD_ARG,
                "conv2d_gated_fwd: input slice [%d,%d) of %d channels (must start at a multiple of 64)", d->x_coff,
                d->x_coff + d->cin, d->x_ctotal);
  CODON_REQUIRE(inputs->coff >= 0 && inputs->coff + d->cin <= inputs->ctotal, CODON_ERR_BAD_ARG,
                "conv2d_gated_fwd: `inputs` slice outside its buffer");
  CODON_REQUIRE(d->y_coff >= 0 && d->y_coff + d->cout <= d->y_ctotal, CODON_ERR_BAD_ARG,
                "conv2d_gated_fwd: output slice outside its buffer");
  CODON_REQUIRE((d->flags & ~CODON_CONV_RELU) == 0, CODON_ERR_BAD_ARG, "conv2d_gated_fwd: only the RELU flag applies");
  CODON_REQUIRE(((uintptr_t)w_packed % 16) == 0, CODON_ERR_BAD_ARG, "conv2d_gated_fwd: packed weights not 16-byte aligned");
  if (d->dtype == CODON_BF16 || d->dtype == CODON_F16)
    return conv2d_gated_fwd_16(d, pre, inputs, ch, sp, w_packed, y, gated_out, (hipStream_t)stream);
  CODON_REQUIRE(d->dtype == CODON_F32, CODON_ERR_UNSUPPORTED, "conv2d_gated_fwd: dtype %d", d->dtype);
  CODON_REQUIRE(!gated_out || (gated_out->coff >= 0 && gated_out->coff + d->cin <= gated_out->ctotal), CODON_ERR_BAD_ARG,
                "conv2d_gated_emit_fwd: gated_out slice outside its buffer");
  return conv2d_gated_fwd_f32(d, (const float*)pre, inputs, ch, sp, (const float*)w_packed, (float*)y, gated_out, (hipStream_t)stream);
}

int codon_conv2d_gated_fwd(const codon_conv_desc* d, const void* pre, const codon_tensor* inputs, const float* ch,
                           const float* sp, const void* w_packed, void* y, codon_stream_t stream) {
  return gated_fwd(d, pre, inputs, ch, sp, w_packed, y, nullptr, stream);
}

int codon_conv2d_gated_emit_fwd(const codon_conv_desc* d, const void* pre, const codon_tensor* inputs, const float* ch,
                                const float* sp, const void* w_packed, void* y, const codon_tensor* gated_out,
                                codon_stream_t stream) {
  CODON_REQUIRE(gated_out && gated_out->data, CODON_ERR_BAD_ARG, "conv2d_gated_emit_fwd: null gated_out");
  return gated_fwd(d, pre, inputs, ch, sp, w_packed, y, gated_out, stream);
}

size_t codon_conv_wgrad_workspace_bytes(const codon_conv_desc* d) {
  if (!d || !shape_ok(d->batch, d->height, d->width)) return 0;
  if (d->dtype == CODON_BF16 || d->dtype == CODON_F16) return conv_wgrad_bf16_workspace_bytes(d);
  return conv_wgrad_workspace_bytes(d);
}

int codon_conv2d_wgrad(const codon_conv_desc* d, const void* x, const void* gy, float* dw, void* workspace,
                       size_t workspace_bytes, int32_t accumulate, codon_stream_t stream) {
  CODON_REQUIRE(d && x && gy && (dw || accumulate == CODON_WGRAD_DEFER) && workspace, CODON_ERR_BAD_ARG, "conv2d_wgrad: null pointer");
  CODON_REQUIRE(accumulate >= 0 && accumulate <= CODON_WGRAD_DEFER, CODON_ERR_BAD_ARG, "conv2d_wgrad: accumulate %d", accumulate);
  CODON_REQUIRE(shape_ok(d->batch, d->height, d->width), CODON_ERR_BAD_ARG, "conv2d_wgrad: bad shape");
  CODON_REQUIRE(d->x_coff >= 0 && d->x_coff + d->cin <= d->x_ctotal && d->y_coff >= 0 &&
                    d->y_coff + d->cout <= d->y_ctotal,
                CODON_ERR_BAD_ARG, "conv2d_wgrad: channel slice outside its buffer");
  if (d->dtype == CODON_BF16 || d->dtype == CODON_F16)
    return conv2d_wgrad_bf16(d, x, gy, dw, (float*)workspace, workspace_bytes, accumulate, (hipStream_t)stream);
  CODON_REQUIRE(d->dtype == CODON_F32, CODON_ERR_UNSUPPORTED, "conv2d_wgrad: dtype %d", d->dtype);
  return conv2d_wgrad_f32(d, (const float*)x, (const float*)gy, dw, (float*)workspace, workspace_bytes, accumulate,
                          (hipStream_t)stream);
}

int codon_conv1x1_bwd(const codon_conv_desc* d, const void* x, const void* gy, const void* w_packed_dgrad,
                      const codon_tensor* gx, float* dw, void* workspace, size_t workspace_bytes, int32_t accumulate,
                      codon_stream_t stream) {
  CODON_REQUIRE(d && x && gy && w_packed_dgrad && gx && gx->data && (dw || accumulate == CODON_WGRAD_DEFER) && workspace,
                CODON_ERR_BAD_ARG, "conv1x1_bwd: null pointer");
  CODON_REQUIRE(accumulate >= 0 && accumulate <= CODON_WGRAD_DEFER, CODON_ERR_BAD_ARG, "conv1x1_bwd: accumulate %d", accumulate);
  CODON_REQUIRE(shape_ok(d->batch, d->height, d->width), CODON_ERR_BAD_ARG, "conv1x1_bwd: bad shape");
  CODON_REQUIRE(d->x_coff >= 0 && d->x_coff + d->cin <= d->x_ctotal && d->y_coff >= 0 &&
                    d->y_coff + d->cout <= d->y_ctotal && gx->coff >= 0 && gx->coff + d->cin <= gx->ctotal,
                CODON_ERR_BAD_ARG, "conv1x1_bwd: channel slice outside its buffer");
  CODON_REQUIRE(((uintptr_t)w_packed_dgrad % 16) == 0, CODON_ERR_BAD_ARG, "conv1x1_bwd: packed weights not 16-byte aligned");
  CODON_REQUIRE(d->dtype == CODON_BF16 || d->dtype == CODON_F16, CODON_ERR_UNSUPPORTED,
                "conv1x1_bwd: 16-bit dtypes only (dtype %d): call codon_conv2d_wgrad + codon_conv2d_fwd", d->dtype);
  return conv1x1_bwd_16(d, x, gy, w_packed_dgrad, gx, dw, (float*)workspace, workspace_bytes, accumulate, (hipStream_t)stream,
                        nullptr);
}

int codon_conv1x1_bwd_gated(const codon_conv_desc* d, const void* x, const void* g_out, const void* w_packed_dgrad,
                            const codon_tensor* gx, float* dw, void* workspace, size_t workspace_bytes, int32_t accumulate,
                            const float* ch, const float* sp, const float* g_pooled, const float* g_pools,
                            const int32_t* argpix, const int32_t* argch, int32_t fcat_base, codon_stream_t stream) {
  CODON_REQUIRE(d && x && g_out && w_packed_dgrad && gx && gx->data && (dw || accumulate == CODON_WGRAD_DEFER) && workspace &&
                    ch && sp && g_pooled && g_pools && argpix && argch,
                CODON_ERR_BAD_ARG, "conv1x1_bwd_gated: null pointer");
  CODON_REQUIRE(accumulate >= 0 && accumulate <= CODON_WGRAD_DEFER, CODON_ERR_BAD_ARG, "conv1x1_bwd_gated: accumulate %d", accumulate);
  CODON_REQUIRE(shape_ok(d->batch, d->height, d->width), CODON_ERR_BAD_ARG, "conv1x1_bwd_gated: bad shape");
  CODON_REQUIRE(d->x_coff >= 0 && d->x_coff + d->cin <= d->x_ctotal && d->y_coff >= 0 &&
                    d->y_coff + d->cout <= d->y_ctotal && gx->coff >= 0 && gx->coff + d->cin <= gx->ctotal,
                CODON_ERR_BAD_ARG, "conv1x1_bwd_gated: channel slice outside its buffer");
  CODON_REQUIRE(fcat_base == 0 || fcat_base == 64, CODON_ERR_BAD_ARG, "conv1x1_bwd_gated: fcat_base %d (0 colour, 64 depth)", fcat_base);
  CODON_REQUIRE(((uintptr_t)w_packed_dgrad % 16) == 0, CODON_ERR_BAD_ARG, "conv1x1_bwd_gated: packed weights not 16-byte aligned");
  CODON_REQUIRE(d->dtype == CODON_BF16 || d->dtype == CODON_F16, CODON_ERR_UNSUPPORTED,
                "conv1x1_bwd_gated: 16-bit dtypes only (dtype %d)", d->dtype);
  const Conv1x1GateBwd gb{ch, sp, g_pooled, g_pools, argpix, argch, fcat_base};
  return conv1x1_bwd_16(d, x, g_out, w_packed_dgrad, gx, dw, (float*)workspace, workspace_bytes, accumulate, (hipStream_t)stream,
                        &gb);
}

int codon_stem_fwd(int32_t batch, int32_t height, int32_t width, const float* x, const float* w_oihw, void* y,
                   int32_t y_ctotal, int32_t y_coff, int32_t dtype, codon_stream_t stream) {
  CODON_REQUIRE(x && w_oihw && y, CODON_ERR_BAD_ARG, "stem_fwd: null pointer");
  CODON_REQUIRE(shape_ok(batch, height, width), CODON_ERR_BAD_ARG, "stem_fwd: bad shape");
  CODON_REQUIRE(y_coff >= 0 && y_coff + 64 <= y_ctotal, CODON_ERR_BAD_ARG, "stem_fwd: output slice outside buffer");
  CODON_REQUIRE((dtype == CODON_F32 || dtype == CODON_BF16 || dtype == CODON_F16), CODON_ERR_UNSUPPORTED, "stem_fwd: dtype %d", dtype);
  return stem_fwd(batch, height, width, x, w_oihw, y, y_ctotal, y_coff, 1, nullptr, 0, 0, dtype,
                  (hipStream_t)stream);
}

int codon_stem_pair_fwd(int32_t batch, int32_t height, int32_t width, const float* xa, const float* wa_oihw, void* ya,
                        int32_t ya_ctotal, int32_t ya_coff, const float* xb, const float* wb_oihw, void* yb,
                        int32_t yb_ctotal, int32_t yb_coff, int32_t dtype, codon_stream_t stream) {
  CODON_REQUIRE(xa && wa_oihw && ya && xb && wb_oihw && yb, CODON_ERR_BAD_ARG, "stem_pair_fwd: null pointer");
  CODON_REQUIRE(shape_ok(batch, height, width), CODON_ERR_BAD_ARG, "stem_pair_fwd: bad shape");
  CODON_REQUIRE(ya_coff >= 0 && ya_coff + 64 <= ya_ctotal && yb_coff >= 0 && yb_coff + 64 <= yb_ctotal, CODON_ERR_BAD_ARG,
                "stem_pair_fwd: output slice outside buffer");
  CODON_REQUIRE(ya != yb || ya_coff + 64 <= yb_coff || yb_coff + 64 <= ya_coff, CODON_ERR_BAD_ARG,
                "stem_pair_fwd: the two output slices overlap");
  CODON_REQUIRE((dtype == CODON_F32 || dtype == CODON_BF16 || dtype == CODON_F16), CODON_ERR_UNSUPPORTED, "stem_pair_fwd: dtype %d", dtype);
  return stem_pair_fwd(batch, height, width, xa, wa_oihw, ya, ya_ctotal, ya_coff, xb, wb_oihw, yb, yb_ctotal, yb_coff, dtype,
                       (hipStream_t)stream);
}

int codon_head_fwd(int32_t batch, int32_t height, int32_t width, const void* x, int32_t x_ctotal, int32_t x_coff,
                   const float* w_oihw, const float* residual, float* y, int32_t dtype, codon_stream_t stream) {
  CODON_REQUIRE(x && w_oihw && residual && y, CODON_ERR_BAD_ARG, "head_fwd: null pointer");
  CODON_REQUIRE(shape_ok(batch, height, width), CODON_ERR_BAD_ARG, "head_fwd: bad shape");
  CODON_REQUIRE(x_coff >= 0 && x_coff + 64 <= x_ctotal, CODON_ERR_BAD_ARG, "head_fwd: input slice outside buffer");
  CODON_REQUIRE((dtype == CODON_F32 || dtype == CODON_BF16 || dtype == CODON_F16), CODON_ERR_UNSUPPORTED, "head_fwd: dtype %d", dtype);
  return head_fwd(batch, height, width, x, x_ctotal, x_coff, w_oihw, residual, y, false, dtype, (hipStream_t)stream);
}

int codon_head_fwd_y16(int32_t batch, int32_t height, int32_t width, const void* x, int32_t x_ctotal, int32_t x_coff,
                       const float* w_oihw, const float* residual, void* y16, int32_t dtype, codon_stream_t stream) {
  CODON_REQUIRE(x && w_oihw && residual && y16, CODON_ERR_BAD_ARG, "head_fwd_y16: null pointer");
  CODON_REQUIRE(shape_ok(batch, height, width), CODON_ERR_BAD_ARG, "head_fwd_y16: bad shape");
  CODON_REQUIRE(x_coff >= 0 && x_coff + 64 <= x_ctotal, CODON_ERR_BAD_ARG, "head_fwd_y16: input slice outside buffer");
  CODON_REQUIRE((dtype == CODON_BF16 || dtype == CODON_F16), CODON_ERR_UNSUPPORTED,
                "head_fwd_y16: dtype %d (the 16-bit output exists for 16-bit activations)", dtype);
  CODON_REQUIRE(((uintptr_t)y16 % 2) == 0, CODON_ERR_BAD_ARG, "head_fwd_y16: output not 2-byte aligned");
  return head_fwd(batch, height, width, x, x_ctotal, x_coff, w_oihw, residual, y16, true, dtype, (hipStream_t)stream);
}

int32_t codon_cac_stats_tiles(int32_t height, int32_t width) {
  if (height <= 0 || width <= 0) return 0;
  return cac_stats_tiles(height, width);
}

int codon_cac_stats_fwd(int32_t batch, int32_t height, int32_t width, const codon_tensor* pre_c,
                        const codon_tensor* pre, float* pooled, float* partials, int32_t dtype,
                        codon_stream_t stream) {
  CODON_REQUIRE(slice_ok(pre_c) && slice_ok(pre) && pooled && partials, CODON_ERR_BAD_ARG,
                "cac_stats_fwd: null pointer or bad channel slice");
  CODON_REQUIRE(shape_ok(batch, height, width), CODON_ERR_BAD_ARG, "cac_stats_fwd: bad shape");
  CODON_REQUIRE(((uintptr_t)partials % 8) == 0, CODON_ERR_BAD_ARG, "cac_stats_fwd: partials not 8-byte aligned");
  CODON_REQUIRE((dtype == CODON_F32 || dtype == CODON_BF16 || dtype == CODON_F16), CODON_ERR_UNSUPPORTED, "cac_stats_fwd: dtype %d", dtype);
  return cac_stats_fwd(batch, height, width, pre_c, pre, pooled, partials, dtype, (hipStream_t)stream, nullptr);
}

int codon_cac_stats_scaled_fwd(int32_t batch, int32_t height, int32_t width, const codon_tensor* pre_c,
                               const codon_tensor* pre, const float* ch, float* pooled, float* partials, int32_t dtype,
                               codon_stream_t stream) {
  CODON_REQUIRE(slice_ok(pre_c) && slice_ok(pre) && ch && pooled && partials, CODON_ERR_BAD_ARG,
                "cac_stats_scaled_fwd: null pointer or bad channel slice");
  CODON_REQUIRE(shape_ok(batch, height, width), CODON_ERR_BAD_ARG, "cac_stats_scaled_fwd: bad shape");
  CODON_REQUIRE(((uintptr_t)partials % 8) == 0, CODON_ERR_BAD_ARG, "cac_stats_scaled_fwd: partials not 8-byte aligned");
  CODON_REQUIRE((dtype == CODON_F32 || dtype == CODON_BF16 || dtype == CODON_F16), CODON_ERR_UNSUPPORTED, "cac_stats_scaled_fwd: dtype %d", dtype);
  return cac_stats_fwd(batch, height, width, pre_c, pre, pooled, partials, dtype, (hipStream_t)stream, ch);
}

int codon_ew_sq_scale(int32_t batch, int32_t height, int32_t width, const codon_tensor* x, const float* ch,
                      const codon_tensor* y, int32_t dtype, codon_stream_t stream) {
  CODON_REQUIRE(slice_ok(x) && slice_ok(y) && ch, CODON_ERR_BAD_ARG, "ew_sq_scale: null pointer or bad channel slice");
  CODON_REQUIRE(shape_ok(batch, height, width), CODON_ERR_BAD_ARG, "ew_sq_scale: bad shape");
  CODON_REQUIRE((dtype == CODON_F32 || dtype == CODON_BF16 || dtype == CODON_F16), CODON_ERR_UNSUPPORTED, "ew_sq_scale: dtype %d", dtype);
  return ew_sq_scale(batch, height, width, x, ch, y, dtype, (hipStream_t)stream);
}

int codon_cac_gate_fwd(int32_t batch, int32_t height, int32_t width, const float* partials, const float* w1,
                       const float* b1, const float* w2, const float* b2, float* ch, float* pools_out,
                       codon_stream_t stream) {
  CODON_REQUIRE(partials && w1 && b1 && w2 && b2 && ch, CODON_ERR_BAD_ARG, "cac_gate_fwd: null pointer");
  CODON_REQUIRE(shape_ok(batch, height, width), CODON_ERR_BAD_ARG, "cac_gate_fwd: bad shape");
  return cac_gate_fwd(batch, height, width, partials, w1, b1, w2, b2, ch, pools_out, (hipStream_t)stream);
}

int codon_cac_spatial_fwd(int32_t batch, int32_t height, int32_t width, const float* pooled, const float* w_spatial,
                          float* sp, codon_stream_t stream) {
  CODON_REQUIRE(pooled && w_spatial && sp, CODON_ERR_BAD_ARG, "cac_spatial_fwd: null pointer");
  CODON_REQUIRE(shape_ok(batch, height, width), CODON_ERR_BAD_ARG, "cac_spatial_fwd: bad shape");
  return cac_spatial_fwd(batch, height, width, pooled, w_spatial, sp, (hipStream_t)stream);
}

int codon_cac_apply_fwd(int32_t batch, int32_t height, int32_t width, const codon_tensor* pre,
                        const codon_tensor* pre_c, const float* ch, const float* sp, const codon_tensor* inputs,
                        const codon_tensor* inputs_c, const codon_tensor* out, const codon_tensor* out_c,
                        int32_t dtype, codon_stream_t stream) {
  CODON_REQUIRE(slice_ok(pre) && slice_ok(pre_c) && ch && sp && slice_ok(inputs) && slice_ok(inputs_c) &&
                    slice_ok(out) && slice_ok(out_c),
                CODON_ERR_BAD_ARG, "cac_apply_fwd: null pointer or bad channel slice");
  CODON_REQUIRE(shape_ok(batch, height, width), CODON_ERR_BAD_ARG, "cac_apply_fwd: bad shape");
  CODON_REQUIRE((dtype == CODON_F32 || dtype == CODON_BF16 || dtype == CODON_F16), CODON_ERR_UNSUPPORTED, "cac_apply_fwd: dtype %d", dtype);
  return cac_apply_fwd(batch, height, width, pre, pre_c, ch, sp, inputs, inputs_c, out, out_c, dtype,
                       (hipStream_t)stream);
}

int codon_stencil_1to64(int32_t batch, int32_t height, int32_t width, const float* x, const float* w_64x9,
                        const codon_tensor* y, int32_t flags, const codon_tensor* mask, int32_t dtype,
                        codon_stream_t stream) {
  CODON_REQUIRE(x && w_64x9 && slice_ok(y) && (!mask || slice_ok(mask)), CODON_ERR_BAD_ARG,
                "stencil_1to64: null pointer or bad channel slice");
  CODON_REQUIRE(shape_ok(batch, height, width), CODON_ERR_BAD_ARG, "stencil_1to64: bad shape");
  CODON_REQUIRE((dtype == CODON_F32 || dtype == CODON_BF16 || dtype == CODON_F16), CODON_ERR_UNSUPPORTED, "stencil_1to64: dtype %d", dtype);
  return stem_fwd(batch, height, width, x, w_64x9, y->data, y->ctotal, y->coff, flags, mask ? mask->data : nullptr,
                  mask ? mask->ctotal : 0, mask ? mask->coff : 0, dtype, (hipStream_t)stream);
}

size_t codon_conv1ch_wgrad_workspace_bytes(int32_t batch, int32_t height, int32_t width) {
  if (!shape_ok(batch, height, width)) return 0;
  return conv1ch_wgrad_workspace_bytes(batch, height, width);
}

int codon_conv1ch_wgrad(int32_t batch, int32_t height, int32_t width, const codon_tensor* a, const float* s,
                        float* dw, int32_t flip, void* workspace, size_t workspace_bytes, int32_t dtype,
                        codon_stream_t stream) {
  CODON_REQUIRE((dtype == CODON_F32 || dtype == CODON_BF16 || dtype == CODON_F16), CODON_ERR_UNSUPPORTED, "conv1ch_wgrad: dtype %d", dtype);
  CODON_REQUIRE(slice_ok(a) && s && (dw || (flip & CODON_W1_DEFER)) && workspace, CODON_ERR_BAD_ARG,
                "conv1ch_wgrad: null pointer or bad slice");
  CODON_REQUIRE((flip & ~(CODON_W1_FLIP | CODON_W1_ACCUMULATE | CODON_W1_DEFER)) == 0, CODON_ERR_BAD_ARG, "conv1ch_wgrad: flip 0x%x", (unsigned)flip);
  CODON_REQUIRE(shape_ok(batch, height, width), CODON_ERR_BAD_ARG, "conv1ch_wgrad: bad shape");
  return conv1ch_wgrad(batch, height, width, a->data, a->ctotal, a->coff, s, dw, flip, (float*)workspace,
                       workspace_bytes, dtype, (hipStream_t)stream);
}

int codon_conv_pair_begin(void) {
  CODON_REQUIRE(!g_pair.active, CODON_ERR_BAD_ARG, "conv_pair_begin: already inside a pair on this thread");
  g_pair.active = true;
  g_pair.n = 0;
  return CODON_OK;
}

int codon_conv_pair_end(codon_stream_t stream) {
  CODON_REQUIRE(g_pair.active, CODON_ERR_BAD_ARG, "conv_pair_end without conv_pair_begin on this thread");
  g_pair.active = false;
  const int n = g_pair.n;
  g_pair.n = 0;
  const PairCall& a = g_pair.call[0];
  const PairCall& b = g_pair.call[1];
  // one grid only for two calls that were given pair_end's own stream: a caller that issued them on different streams
  // ordered its other work against THOSE streams
  if (n == 2 && a.pair == b.pair && a.nblk == b.nblk && a.tiles_x == b.tiles_x && a.tiles_y == b.tiles_y &&
      a.stream == (hipStream_t)stream && b.stream == (hipStream_t)stream && (long)a.nblk * 2 < (1L << 31)) {
    const int st = a.pair(a.blob, b.blob, (hipStream_t)stream);
    return st == CODON_OK ? 1 : st;
  }
  // a conv5x5 64->64 and the conv3x3 64->64 of the same family (mix53): one grid of both kinds of workgroup
  if (n == 2 && a.stream == (hipStream_t)stream && b.stream == (hipStream_t)stream && a.mix_kind && b.mix_kind &&
      (long)a.nblk + b.nblk < (1L << 31)) {
    const PairCall* five = (a.mix_kind & 1) ? &a : &b;
    const PairCall* three = (a.mix_kind & 1) ? &b : &a;
    if ((five->mix_kind & 1) && three->mix_kind == five->mix_kind + 1 && five->mix) {
      const int st = five->mix(five->blob, three->blob, (hipStream_t)stream);
      return st == CODON_OK ? 1 : st;
    }
  }
  for (int k = 0; k < n; ++k) {          // one by one, each on the stream its call named
    const int st = g_pair.call[k].single(g_pair.call[k].blob, g_pair.call[k].stream);
    if (st != CODON_OK) return st;
  }
  return n;
}

int codon_conv_tiling_f32(const codon_conv_desc* d, int chained, int in_pair) {
  CODON_REQUIRE(d && shape_ok(d->batch, d->height, d->width), CODON_ERR_BAD_ARG, "conv_tiling_f32: null descriptor or bad shape");
  return conv_tiling_f32(d, chained, in_pair);
}

int codon_cast_multi(const codon_cast_desc* desc, float* dst, codon_stream_t stream) {
  CODON_REQUIRE(desc && dst, CODON_ERR_BAD_ARG, "cast_multi: null pointer");
  return cast_multi(desc, dst, (hipStream_t)stream);
}

int codon_adam_step(const codon_adam_desc* desc, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int32_t step, codon_stream_t stream) {
  CODON_REQUIRE(desc && grad && exp_avg && exp_avg_sq, CODON_ERR_BAD_ARG, "adam_step: null pointer");
  return adam_step(desc, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, (hipStream_t)stream);
}

int codon_reduce_multi(const codon_reduce_item* items, int32_t n_items, codon_stream_t stream) {
  CODON_REQUIRE(items && n_items >= 1, CODON_ERR_BAD_ARG, "reduce_multi: no items");
  return reduce_multi(items, n_items, (hipStream_t)stream);
}

int codon_ew_add_mask(int32_t batch, int32_t height, int32_t width, int32_t channels, const codon_tensor* dst,
                      const codon_tensor* src, const codon_tensor* mask, int32_t accumulate, int32_t dtype,
                      codon_stream_t stream) {
  CODON_REQUIRE((dtype == CODON_F32 || dtype == CODON_BF16 || dtype == CODON_F16), CODON_ERR_UNSUPPORTED, "ew_add_mask: dtype %d", dtype);
  auto ok = [&](const codon_tensor* t) { return t && t->data && t->coff >= 0 && t->coff + channels <= t->ctotal; };
  CODON_REQUIRE(channels > 0 && ok(dst) && (!src || ok(src)) && (!mask || ok(mask)), CODON_ERR_BAD_ARG,
                "ew_add_mask: null pointer or bad channel slice");
  CODON_REQUIRE(shape_ok(batch, height, width), CODON_ERR_BAD_ARG, "ew_add_mask: bad shape");
  return ew_add_mask(batch, height, width, channels, dst, src, mask, accumulate, dtype, (hipStream_t)stream);
}

int codon_ew_sum_mask(int32_t batch, int32_t height, int32_t width, int32_t channels, const codon_tensor* dst, int32_t nsrc,
                      const codon_tensor* src0, const codon_tensor* src1, const codon_tensor* src2, const codon_tensor* src3,
                      const codon_tensor* mask, int32_t dtype, codon_stream_t stream) {
  CODON_REQUIRE((dtype == CODON_F32 || dtype == CODON_BF16 || dtype == CODON_F16), CODON_ERR_UNSUPPORTED, "ew_sum_mask: dtype %d", dtype);
  auto ok = [&](const codon_tensor* t) { return t && t->data && t->coff >= 0 && t->coff + channels <= t->ctotal; };
  const codon_tensor* srcs[4] = {src0, src1, src2, src3};
  CODON_REQUIRE(nsrc >= 1 && nsrc <= 4, CODON_ERR_BAD_ARG, "ew_sum_mask: %d sources (1..4)", nsrc);
  for (int i = 0; i < nsrc; ++i)
    CODON_REQUIRE(ok(srcs[i]) && srcs[i]->data != dst->data, CODON_ERR_BAD_ARG, "ew_sum_mask: source %d null, out of range or aliasing dst", i);
  CODON_REQUIRE(channels > 0 && ok(dst) && (!mask || ok(mask)), CODON_ERR_BAD_ARG, "ew_sum_mask: null pointer or bad channel slice");
  CODON_REQUIRE(shape_ok(batch, height, width), CODON_ERR_BAD_ARG, "ew_sum_mask: bad shape");
  return ew_sum_mask(batch, height, width, channels, dst, nsrc, srcs, mask, dtype, (hipStream_t)stream);
}

int32_t codon_cac_bwd_tiles(int32_t height, int32_t width) {
  return (height > 0 && width > 0) ? cac_bwd_tiles(height, width) : 0;
}
int32_t codon_cac_bwd_spatial_blocks(int32_t batch, int32_t height, int32_t width) {
  return shape_ok(batch, height, width) ? cac_bwd_spatial_blocks(batch, height, width) : 0;
}

int codon_cac_bwd_reduce(int32_t batch, int32_t height, int32_t width, const codon_tensor* g_out,
                         const codon_tensor* g_out_c, const codon_tensor* pre, const codon_tensor* pre_c,
                         const float* ch, const float* sp, const float* pools, float* g_z, float* part_gch,
                         int32_t* part_arg, int32_t dtype, codon_stream_t stream) {
  CODON_REQUIRE((dtype == CODON_F32 || dtype == CODON_BF16 || dtype == CODON_F16), CODON_ERR_UNSUPPORTED, "cac_bwd_reduce: dtype %d", dtype);
  CODON_REQUIRE(slice_ok(g_out) && slice_ok(g_out_c) && slice_ok(pre) && slice_ok(pre_c) && ch && sp && pools &&
                    g_z && part_gch && part_arg,
                CODON_ERR_BAD_ARG, "cac_bwd_reduce: null pointer or bad channel slice");
  CODON_REQUIRE(shape_ok(batch, height, width) && batch <= 65535, CODON_ERR_BAD_ARG, "cac_bwd_reduce: bad shape");
  return cac_bwd_reduce(batch, height, width, g_out, g_out_c, pre, pre_c, ch, sp, pools, g_z, part_gch, part_arg,
                        dtype, (hipStream_t)stream);
}

int codon_cac_bwd_reduce_acc(int32_t batch, int32_t height, int32_t width, const codon_tensor* g_out,
                             const codon_tensor* g_out_c, const codon_tensor* pre, const codon_tensor* pre_c,
                             const float* ch, const float* sp, const float* pools, const float* pooled, float* g_z,
                             float* part_gch, int32_t* part_arg, int32_t* argch, const codon_tensor* g_in,
                             const codon_tensor* g_in_c, int32_t accumulate_in, int32_t dtype, codon_stream_t stream) {
  CODON_REQUIRE(dtype == CODON_BF16 || dtype == CODON_F16, CODON_ERR_UNSUPPORTED,
                "cac_bwd_reduce_acc: 16-bit dtypes only (dtype %d): fp32 takes codon_cac_bwd_reduce + codon_cac_bwd_apply", dtype);
  CODON_REQUIRE(slice_ok(g_out) && slice_ok(g_out_c) && slice_ok(pre) && slice_ok(pre_c) && slice_ok(g_in) && slice_ok(g_in_c) &&
                    ch && sp && pools && pooled && g_z && part_gch && part_arg && argch,
                CODON_ERR_BAD_ARG, "cac_bwd_reduce_acc: null pointer or bad channel slice");
  CODON_REQUIRE(shape_ok(batch, height, width) && batch <= 65535, CODON_ERR_BAD_ARG, "cac_bwd_reduce_acc: bad shape");
  return cac_bwd_reduce_acc(batch, height, width, g_out, g_out_c, pre, pre_c, ch, sp, pools, pooled, g_z, part_gch, part_arg,
                            argch, g_in, g_in_c, accumulate_in, dtype, (hipStream_t)stream);
}

int codon_cac_bwd_gate(int32_t batch, int32_t height, int32_t width, const float* part_gch, const int32_t* part_arg,
                       const float* ch, const float* pools, const float* w1, const float* b1, const float* w2,
                       float* g_pools, int32_t* argpix, float* part_param, float* dw1, float* db1, float* dw2,
                       float* db2, codon_stream_t stream) {
  CODON_REQUIRE(part_gch && part_arg && ch && pools && w1 && b1 && w2 && g_pools && argpix && part_param &&
                    ((dw1 && db1 && dw2 && db2) || (!dw1 && !db1 && !dw2 && !db2)),
                CODON_ERR_BAD_ARG, "cac_bwd_gate: null pointer (dw1 / db1 / dw2 / db2: all four or none)");
  CODON_REQUIRE(shape_ok(batch, height, width), CODON_ERR_BAD_ARG, "cac_bwd_gate: bad shape");
  return cac_bwd_gate(batch, height, width, part_gch, part_arg, ch, pools, w1, b1, w2, g_pools, argpix, part_param,
                      dw1, db1, dw2, db2, (hipStream_t)stream);
}

int codon_cac_bwd_spatial(int32_t batch, int32_t height, int32_t width, const float* g_z, const float* pooled,
                          const float* w_spatial, float* g_pooled, float* part_w, float* dw_spatial,
                          codon_stream_t stream) {
  CODON_REQUIRE(g_z && pooled && w_spatial && g_pooled && part_w, CODON_ERR_BAD_ARG, "cac_bwd_spatial: null pointer");
  CODON_REQUIRE(shape_ok(batch, height, width), CODON_ERR_BAD_ARG, "cac_bwd_spatial: bad shape");
  return cac_bwd_spatial(batch, height, width, g_z, pooled, w_spatial, g_pooled, part_w, dw_spatial,
                         (hipStream_t)stream);
}

int codon_cac_bwd_apply(int32_t batch, int32_t height, int32_t width, const codon_tensor* g_out,
                        const codon_tensor* g_out_c, const codon_tensor* pre, const codon_tensor* pre_c,
                        const float* ch, const float* sp, const float* pooled, const float* g_pooled,
                        const float* g_pools, const int32_t* argpix, const codon_tensor* g_pre,
                        const codon_tensor* g_pre_c, const codon_tensor* g_in, const codon_tensor* g_in_c,
                        int32_t accumulate_in, int32_t dtype, codon_stream_t stream) {
  CODON_REQUIRE((dtype == CODON_F32 || dtype == CODON_BF16 || dtype == CODON_F16), CODON_ERR_UNSUPPORTED, "cac_bwd_apply: dtype %d", dtype);
  CODON_REQUIRE(slice_ok(g_out) && slice_ok(g_out_c) && slice_ok(pre) && slice_ok(pre_c) && ch && sp && pooled &&
                    g_pooled && g_pools && argpix && slice_ok(g_pre) && slice_ok(g_pre_c) && slice_ok(g_in) &&
                    slice_ok(g_in_c),
                CODON_ERR_BAD_ARG, "cac_bwd_apply: null pointer or bad channel slice");
  CODON_REQUIRE(shape_ok(batch, height, width) && batch <= 65535, CODON_ERR_BAD_ARG, "cac_bwd_apply: bad shape");
  return cac_bwd_apply(batch, height, width, g_out, g_out_c, pre, pre_c, ch, sp, pooled, g_pooled, g_pools, argpix,
                       g_pre, g_pre_c, g_in, g_in_c, accumulate_in, dtype, (hipStream_t)stream);
}

int codon_postprocess_u8(int64_t n, const float* x, uint8_t* out, codon_stream_t stream) {
  CODON_REQUIRE(x && out && n > 0, CODON_ERR_BAD_ARG, "postprocess_u8: null pointer or n <= 0");
  return postprocess_u8(x, out, (long)n, (hipStream_t)stream);
}

int codon_postprocess_u8_dt(int64_t n, const void* x, int32_t dtype, uint8_t* out, codon_stream_t stream) {
  CODON_REQUIRE(x && out && n > 0, CODON_ERR_BAD_ARG, "postprocess_u8_dt: null pointer or n <= 0");
  CODON_REQUIRE(dtype == CODON_F32 || dtype == CODON_F16, CODON_ERR_UNSUPPORTED,
                "postprocess_u8_dt: dtype %d (fp32 or fp16; numpy has no bf16 -- upcast bf16 to fp32 first)", dtype);
  if (dtype == CODON_F16) return postprocess_u8_f16(x, out, (long)n, (hipStream_t)stream);
  return postprocess_u8((const float*)x, out, (long)n, (hipStream_t)stream);
}

int codon_masked_sqerr(int64_t n, const uint8_t* label, const uint8_t* out, uint64_t* acc, codon_stream_t stream) {
  CODON_REQUIRE(label && out && acc && n > 0, CODON_ERR_BAD_ARG, "masked_sqerr: null pointer or n <= 0");
  return masked_sqerr(label, out, (long)n, (unsigned long long*)acc, (hipStream_t)stream);
}

int32_t codon_ssim_tiles(int32_t batch, int32_t height, int32_t width) {
  return shape_ok(batch, height, width) ? ssim_tiles(batch, height, width) : 0;
}

int codon_ssim_fwd(int32_t batch, int32_t height, int32_t width, const float* a, const float* b, float* partial,
                   float* dmaps, double* value, codon_stream_t stream) {
  CODON_REQUIRE(a && b && partial && value, CODON_ERR_BAD_ARG, "ssim_fwd: null pointer");
  CODON_REQUIRE(shape_ok(batch, height, width), CODON_ERR_BAD_ARG, "ssim_fwd: bad shape");
  return ssim_fwd(batch, height, width, a, b, partial, dmaps, value, (hipStream_t)stream);
}

int codon_l1_fwd(int64_t n, const float* a, const float* b, float* partial, int32_t nparts, double* value,
                 codon_stream_t stream) {
  CODON_REQUIRE(a && b && partial && value && n > 0 && nparts > 0, CODON_ERR_BAD_ARG, "l1_fwd: bad argument");
  return l1_fwd((long)n, a, b, partial, nparts, value, (hipStream_t)stream);
}

int codon_ssim_l1_bwd(int32_t batch, int32_t height, int32_t width, const float* a, const float* b,
                      const float* dmaps, float* tmp, float* ga, float ssim_scale, float l1_scale,
                      codon_stream_t stream) {
  CODON_REQUIRE(a && b && dmaps && tmp && ga, CODON_ERR_BAD_ARG, "ssim_l1_bwd: null pointer");
  CODON_REQUIRE(shape_ok(batch, height, width) && height >= 7 && width >= 7, CODON_ERR_UNSUPPORTED,
                "ssim_l1_bwd: needs H, W >= 7 (got %dx%d)", height, width);
  return ssim_l1_bwd(batch, height, width, a, b, dmaps, tmp, ga, ssim_scale, l1_scale, (hipStream_t)stream);
}

int codon_bicubic_upsample(int32_t batch, int32_t lr_height, int32_t lr_width, int32_t scale, const float* lr,
                           const float* phase_weights, float* out, codon_stream_t stream) {
  CODON_REQUIRE(lr && phase_weights && out, CODON_ERR_BAD_ARG, "bicubic_upsample: null pointer");
  CODON_REQUIRE(scale == 4 || scale == 8 || scale == 16, CODON_ERR_UNSUPPORTED, "bicubic_upsample: scale %d", scale);
  CODON_REQUIRE(shape_ok(batch, lr_height * scale, lr_width * scale), CODON_ERR_BAD_ARG, "bicubic_upsample: bad shape");
  return bicubic_upsample(batch, lr_height, lr_width, scale, lr, phase_weights, out, (hipStream_t)stream);
}

size_t codon_weight_checksum_workspace_bytes(void) { return weight_checksum_workspace_bytes(); }

int codon_weight_checksum(const codon_wsum_desc* desc, void* ws, uint64_t* ref, int32_t mode, int32_t* flag,
                          codon_stream_t stream) {
  CODON_REQUIRE(desc && ws && ref && flag, CODON_ERR_BAD_ARG, "weight_checksum: null pointer");
  CODON_REQUIRE(desc->n > 0 && desc->n <= CODON_WSUM_MAX && (mode == 0 || mode == 1), CODON_ERR_BAD_ARG,
                "weight_checksum: n %d (1..%d), mode %d (0, 1)", desc->n, CODON_WSUM_MAX, mode);
  uint64_t total = 0;
  for (int t = 0; t < desc->n; ++t) {
    CODON_REQUIRE(desc->data[t] && desc->bytes[t] % 16 == 0 && ((uintptr_t)desc->data[t] & 15) == 0, CODON_ERR_BAD_ARG,
                  "weight_checksum: tensor %d must be 16-byte aligned with a multiple of 16 bytes", t);
    total += desc->bytes[t];
  }
  CODON_REQUIRE(total / 16 < (1ull << 32), CODON_ERR_UNSUPPORTED, "weight_checksum: more than 64 GiB of weights");
  return weight_checksum(desc, ws, (unsigned long long*)ref, mode, flag, (hipStream_t)stream);
}

}  // extern "C"
