// x4 / x8 / x16 bicubic upsample of a 1-channel map (synthetic-input generator: the reference's
// network has NO upsample -- its depth inputs are bicubic-upsampled offline,
// /root/reference/CODON_X4/test.py:70-77 -- so this has no reference counterpart; it is defined
// once here and in oracle/upsample_oracle.py and the two must agree BIT FOR BIT).
//
// Index arithmetic (all integer): half-pixel centres, src = (dst + 0.5)/s - 0.5.
//   dst = s*q + r, r in [0,s):  i0 = q - (2r+1 < s),  phase r selects the 4 weights,
//   taps i0-1 .. i0+2 clamped to [0, n-1].
// Weights: Keys cubic a = -0.75 evaluated in fp64 on the host, rounded once to fp32, one row of 4
// per phase (table of s*4 floats shared by both implementations).
// Arithmetic: h_k = (w0*p0 + w1*p1) + (w2*p2 + w3*p3) per tap row, then the same form vertically,
// every operation individually rounded (no FMA contraction) so that numpy reproduces it exactly.

#include "codon_common.h"

// HIP's __fmul_rn/__fadd_rn are plain * and + and would be contracted into v_fma under the default
// -ffp-contract=fast; this file must round every operation separately.
#pragma clang fp contract(off)

namespace codon {

__device__ __forceinline__ float dot4_rn(float w0, float w1, float w2, float w3, float p0, float p1, float p2,
                                         float p3) {
  return __fadd_rn(__fadd_rn(__fmul_rn(w0, p0), __fmul_rn(w1, p1)), __fadd_rn(__fmul_rn(w2, p2), __fmul_rn(w3, p3)));
}

__global__ __launch_bounds__(256) void bicubic_kernel(const float* __restrict__ lr, const float* __restrict__ wtab,
                                                      float* __restrict__ out, int h, int w, int s, long total) {
  const long idx = blockIdx.x * 256L + threadIdx.x;
  if (idx >= total) return;
  const int W = w * s, H = h * s;
  const int gx = (int)(idx % W);
  const long t = idx / W;
  const int gy = (int)(t % H);
  const int b = (int)(t / H);
  const int qx = gx / s, rx = gx - qx * s, qy = gy / s, ry = gy - qy * s;
  const int ix0 = qx - ((2 * rx + 1 < s) ? 1 : 0), iy0 = qy - ((2 * ry + 1 < s) ? 1 : 0);
  const float* wx = wtab + rx * 4;
  const float* wy = wtab + ry * 4;
  int xs[4], ys[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    xs[k] = min(max(ix0 - 1 + k, 0), w - 1);
    ys[k] = min(max(iy0 - 1 + k, 0), h - 1);
  }
  const float* p = lr + (long)b * h * w;
  float hrow[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float* r = p + (long)ys[k] * w;
    hrow[k] = dot4_rn(wx[0], wx[1], wx[2], wx[3], r[xs[0]], r[xs[1]], r[xs[2]], r[xs[3]]);
  }
  out[idx] = dot4_rn(wy[0], wy[1], wy[2], wy[3], hrow[0], hrow[1], hrow[2], hrow[3]);
}

int bicubic_upsample(int B, int h, int w, int s, const float* lr, const float* wtab, float* out, hipStream_t stream) {
  const long total = (long)B * h * s * w * s;
  const long blocks = (total + 255) / 256;
  CODON_REQUIRE(blocks < (1L << 31), CODON_ERR_UNSUPPORTED, "bicubic_upsample: grid too large");
  hipLaunchKernelGGL(bicubic_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, lr, wtab, out, h, w, s, total);
  return check_launch("bicubic_kernel");
}

}  // namespace codon
