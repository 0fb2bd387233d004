// Stem (1->64, 3x3, ReLU) and head (64->1, 3x3, + global residual) stencils: HBM-bound VALU kernels.
//   stem: self.relu(self.input(x)) / self.relu(self.input_c(y))   CODON_x4.py:68,71
//   head: torch.add(self.output(out), residual)                    CODON_x4.py:130-131
// Both walk NCHW rows with VEC consecutive pixels per lane, so a wave touches 64*VEC*4
// contiguous bytes per load/store instruction (1 KiB at VEC=4).  VEC=4 needs W % 4 == 0 (every
// plane row is then 16-byte aligned); other widths take the VEC=1 instantiation.

#include "codon_common.h"

namespace codon {

template <int VEC>
struct Vec;
template <>
struct Vec<4> { using T = float4; };
template <>
struct Vec<1> { using T = float; };

template <int VEC>
__device__ __forceinline__ void load_row(const float* __restrict__ plane, int gy, int gx0, int H, int W,
                                         float (&v)[VEC + 2]) {
  // v[0] = pixel gx0-1 ... v[VEC+1] = pixel gx0+VEC, zero outside the image
  if (gy < 0 || gy >= H) {
#pragma unroll
    for (int i = 0; i < VEC + 2; ++i) v[i] = 0.f;
    return;
  }
  const float* row = plane + (long)gy * W;
  if constexpr (VEC == 4) {
    const float4 c = *reinterpret_cast<const float4*>(row + gx0);
    v[1] = c.x; v[2] = c.y; v[3] = c.z; v[4] = c.w;
  } else {
    v[1] = row[gx0];
  }
  v[0] = gx0 > 0 ? row[gx0 - 1] : 0.f;
  v[VEC + 1] = gx0 + VEC < W ? row[gx0 + VEC] : 0.f;
}

template <int VEC>
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                   float* __restrict__ y, int H, int W, long y_img,
                                                   long y_base, long total) {
  __shared__ float wsh[64 * 9];
  for (int i = threadIdx.x; i < 576; i += 256) wsh[i] = w[i];
  __syncthreads();
  const long idx = blockIdx.x * 256L + threadIdx.x;
  if (idx >= total) return;
  const int WV = W / VEC;
  const int gxv = (int)(idx % WV);
  const long t = idx / WV;
  const int gy = (int)(t % H);
  const int b = (int)(t / H);
  const int gx0 = gxv * VEC;
  const float* plane = x + (long)b * H * W;
  float r0[VEC + 2], r1[VEC + 2], r2[VEC + 2];
  load_row<VEC>(plane, gy - 1, gx0, H, W, r0);
  load_row<VEC>(plane, gy, gx0, H, W, r1);
  load_row<VEC>(plane, gy + 1, gx0, H, W, r2);
  const long HW = (long)H * W;
  float* yo = y + (long)b * y_img + y_base + (long)gy * W + gx0;
#pragma unroll 4
  for (int co = 0; co < 64; ++co) {
    const float* k = wsh + co * 9;
    float o[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float a = k[0] * r0[i];
      a = fmaf(k[1], r0[i + 1], a);
      a = fmaf(k[2], r0[i + 2], a);
      a = fmaf(k[3], r1[i], a);
      a = fmaf(k[4], r1[i + 1], a);
      a = fmaf(k[5], r1[i + 2], a);
      a = fmaf(k[6], r2[i], a);
      a = fmaf(k[7], r2[i + 1], a);
      a = fmaf(k[8], r2[i + 2], a);
      o[i] = fmaxf(a, 0.f);
    }
    if constexpr (VEC == 4) *reinterpret_cast<float4*>(yo + co * HW) = make_float4(o[0], o[1], o[2], o[3]);
    else yo[co * HW] = o[0];
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void head_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ res, float* __restrict__ y, int H,
                                                   int W, long x_img, long x_base, long total) {
  __shared__ float wsh[64 * 9];
  for (int i = threadIdx.x; i < 576; i += 256) wsh[i] = w[i];
  __syncthreads();
  const long idx = blockIdx.x * 256L + threadIdx.x;
  if (idx >= total) return;
  const int WV = W / VEC;
  const int gxv = (int)(idx % WV);
  const long t = idx / WV;
  const int gy = (int)(t % H);
  const int b = (int)(t / H);
  const int gx0 = gxv * VEC;
  const long HW = (long)H * W;
  const float* xb = x + (long)b * x_img + x_base;
  float o[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) o[i] = 0.f;
#pragma unroll 2
  for (int c = 0; c < 64; ++c) {
    const float* plane = xb + c * HW;
    const float* k = wsh + c * 9;
    float r0[VEC + 2], r1[VEC + 2], r2[VEC + 2];
    load_row<VEC>(plane, gy - 1, gx0, H, W, r0);
    load_row<VEC>(plane, gy, gx0, H, W, r1);
    load_row<VEC>(plane, gy + 1, gx0, H, W, r2);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float a = o[i];
      a = fmaf(k[0], r0[i], a);
      a = fmaf(k[1], r0[i + 1], a);
      a = fmaf(k[2], r0[i + 2], a);
      a = fmaf(k[3], r1[i], a);
      a = fmaf(k[4], r1[i + 1], a);
      a = fmaf(k[5], r1[i + 2], a);
      a = fmaf(k[6], r2[i], a);
      a = fmaf(k[7], r2[i + 1], a);
      a = fmaf(k[8], r2[i + 2], a);
      o[i] = a;
    }
  }
  const long off = (long)b * HW + (long)gy * W + gx0;
  if constexpr (VEC == 4) {
    const float4 r = *reinterpret_cast<const float4*>(res + off);
    *reinterpret_cast<float4*>(y + off) = make_float4(o[0] + r.x, o[1] + r.y, o[2] + r.z, o[3] + r.w);
  } else {
    y[off] = o[0] + res[off];
  }
}

int stem_fwd_f32(int B, int H, int W, const float* x, const float* w, float* y, int y_ctotal, int y_coff,
                 hipStream_t stream) {
  const long HW = (long)H * W;
  const bool v4 = (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) % 16 == 0);
  const long total = (long)B * H * (v4 ? W / 4 : W);
  const long blocks = (total + 255) / 256;
  CODON_REQUIRE(blocks < (1L << 31), CODON_ERR_UNSUPPORTED, "stem_fwd: grid too large");
  if (v4)
    hipLaunchKernelGGL(stem_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, stream, x, w, y, H, W, y_ctotal * HW,
                       y_coff * HW, total);
  else
    hipLaunchKernelGGL(stem_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, stream, x, w, y, H, W, y_ctotal * HW,
                       y_coff * HW, total);
  return check_launch("stem_kernel");
}

int head_fwd_f32(int B, int H, int W, const float* x, int x_ctotal, int x_coff, const float* w, const float* res,
                 float* y, hipStream_t stream) {
  const long HW = (long)H * W;
  const bool v4 = (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) |
                                    reinterpret_cast<uintptr_t>(res)) % 16 == 0);
  const long total = (long)B * H * (v4 ? W / 4 : W);
  const long blocks = (total + 255) / 256;
  CODON_REQUIRE(blocks < (1L << 31), CODON_ERR_UNSUPPORTED, "head_fwd: grid too large");
  if (v4)
    hipLaunchKernelGGL(head_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, stream, x, w, res, y, H, W,
                       x_ctotal * HW, x_coff * HW, total);
  else
    hipLaunchKernelGGL(head_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, stream, x, w, res, y, H, W,
                       x_ctotal * HW, x_coff * HW, total);
  return check_launch("head_kernel");
}

}  // namespace codon
