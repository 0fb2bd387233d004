// One optimizer step of Adam over the 44 used parameter tensors in ONE launch (round 6).  The reference ships no training code
// (SURVEY.md D8); the bench's training step (BASELINE.json configs[2]) used torch.optim.Adam -- five ATen multi_tensor_apply
// launches inside the timed region.  The gradients already sit in codon_amd.dist.GradSync's flat all-reduce buffer, in the
// parameters' order; the two moment buffers are flat alike, so the step is elementwise over (tensor, offset):
//   g  = grad (+ weight_decay * p)
//   m  = m + (1 - beta1) * (g - m)                      (torch: exp_avg.lerp_(grad, 1 - beta1))
//   v  = beta2 * v + (1 - beta2) * g * g                (torch: exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2))
//   p -= (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
// -- torch.optim.Adam's arithmetic (amsgrad = False, maximize = False), to fp32 rounding (tests/test_gpu_reduce.py).
#include <math.h>

#include "codon_common.h"

namespace codon {

struct AdamArgs {
  float* p[CODON_ADAM_MAX];
  unsigned start[CODON_ADAM_MAX + 1];        // tensor t -> flat elements [start[t], start[t + 1])
  int first_block[CODON_ADAM_MAX + 1];       // ... -> workgroups [first_block[t], first_block[t + 1]), 1024 elements each
  int n;
  const float* g;
  float* m;
  float* v;
  float step_size, w1, beta2, one_minus_beta2, inv_sqrt_bc2, eps, wd;
};
static_assert(sizeof(AdamArgs) <= CODON_KERNARG_LIMIT, "passed by value as a kernel argument");

__global__ __launch_bounds__(256) void adam_step_kernel(const AdamArgs a) {
  int t = 0;
  while (t + 1 < a.n && (int)blockIdx.x >= a.first_block[t + 1]) ++t;        // block-uniform
  const unsigned cnt = a.start[t + 1] - a.start[t];
  const unsigned base = ((unsigned)blockIdx.x - (unsigned)a.first_block[t]) * 1024u;
  float* const p = a.p[t];
  const float* const g = a.g + a.start[t];
  float* const m = a.m + a.start[t];
  float* const v = a.v + a.start[t];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned i = base + k * 256u + threadIdx.x;
    if (i >= cnt) break;
    const float pi = p[i];
    const float gi = a.wd != 0.f ? fmaf(a.wd, pi, g[i]) : g[i];
    const float mi = m[i] + a.w1 * (gi - m[i]);
    const float vi = a.beta2 * v[i] + a.one_minus_beta2 * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = pi - a.step_size * (mi / (sqrtf(vi) * a.inv_sqrt_bc2 + a.eps));
  }
}

int adam_step(const codon_adam_desc* d, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2,
              float eps, float weight_decay, int step, hipStream_t stream) {
  CODON_REQUIRE(d->n >= 1 && d->n <= CODON_ADAM_MAX, CODON_ERR_BAD_ARG, "adam_step: %d tensors (1..%d)", d->n, CODON_ADAM_MAX);
  CODON_REQUIRE(step >= 1 && lr >= 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f,
                CODON_ERR_BAD_ARG, "adam_step: step %d lr %g betas (%g, %g) eps %g", step, lr, beta1, beta2, eps);
  AdamArgs a;
  a.n = d->n;
  a.start[0] = 0;
  a.first_block[0] = 0;
  for (int t = 0; t < d->n; ++t) {
    CODON_REQUIRE(d->param[t] && d->count[t] > 0 && ((uintptr_t)d->param[t] % 4) == 0, CODON_ERR_BAD_ARG,
                  "adam_step: tensor %d: null, empty or misaligned", t);
    const long nb = (d->count[t] + 1023) / 1024;
    CODON_REQUIRE((long)a.start[t] + d->count[t] < (1L << 32) && (long)a.first_block[t] + nb < (1L << 31), CODON_ERR_UNSUPPORTED,
                  "adam_step: too many elements");
    a.p[t] = (float*)d->param[t];
    a.start[t + 1] = a.start[t] + (unsigned)d->count[t];
    a.first_block[t + 1] = a.first_block[t] + (int)nb;
  }
  for (int t = d->n; t < CODON_ADAM_MAX; ++t) { a.p[t] = nullptr; a.start[t + 1] = a.start[d->n]; a.first_block[t + 1] = a.first_block[d->n]; }
  a.g = grad; a.m = exp_avg; a.v = exp_avg_sq;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  a.step_size = (float)((double)lr / bc1);
  a.w1 = 1.f - beta1;
  a.beta2 = beta2;
  a.one_minus_beta2 = 1.f - beta2;
  a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  a.eps = eps;
  a.wd = weight_decay;
  hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)a.first_block[d->n]), dim3(256), 0, stream, a);
  return check_launch("adam_step_kernel");
}

}  // namespace codon
