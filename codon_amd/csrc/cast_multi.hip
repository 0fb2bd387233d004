// The small fp32-arithmetic parameters of the network -- the two stems, the head, the 25 gate tensors -- as ONE flat fp32
// buffer in one launch.  A model cast to 16 bits as a whole (`model.cuda().half()`, the reference script's own use,
// /root/reference/CODON_X4/test.py:52) holds them in 16 bits; the kernels that consume them take fp32, and converting them
// one by one cost ~30 launches of 5 us per forward at one image per call (profiles/r05_b1_fp16_370x463_timeline.txt).
// Read from the LIVE parameters on every forward: nothing is cached, so nothing can go stale.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include "codon_common.h"

namespace codon {

struct CastMultiArgs {
  int n;
  unsigned start[CODON_CAST_MAX + 1];     // prefix sums of element counts: tensor t -> dst[start[t] .. start[t+1])
  const void* src[CODON_CAST_MAX];
  int dtype[CODON_CAST_MAX];
};
static_assert(sizeof(CastMultiArgs) <= CODON_KERNARG_LIMIT, "passed by value as a kernel argument");

__global__ __launch_bounds__(256) void cast_multi_kernel(const CastMultiArgs a, float* __restrict__ dst) {
  const unsigned total = a.start[a.n];
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    int t = 0;
    while (t + 1 < a.n && i >= a.start[t + 1]) ++t;
    const unsigned k = i - a.start[t];
    float v;
    if (a.dtype[t] == CODON_F16) v = __half2float(reinterpret_cast<const __half*>(a.src[t])[k]);
    else if (a.dtype[t] == CODON_BF16) v = __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(a.src[t])[k] << 16);
    else v = reinterpret_cast<const float*>(a.src[t])[k];
    dst[i] = v;
  }
}

int cast_multi(const codon_cast_desc* d, float* dst, hipStream_t stream) {
  CODON_REQUIRE(d->n >= 1 && d->n <= CODON_CAST_MAX, CODON_ERR_BAD_ARG, "cast_multi: %d tensors (1..%d)", d->n, CODON_CAST_MAX);
  CastMultiArgs a;
  a.n = d->n;
  a.start[0] = 0;
  for (int t = 0; t < d->n; ++t) {
    CODON_REQUIRE(d->src[t] && d->count[t] > 0 && (d->dtype[t] == CODON_F32 || d->dtype[t] == CODON_BF16 || d->dtype[t] == CODON_F16),
                  CODON_ERR_BAD_ARG, "cast_multi: tensor %d: null, empty or dtype %d", t, d->dtype[t]);
    a.src[t] = d->src[t];
    a.dtype[t] = d->dtype[t];
    a.start[t + 1] = a.start[t] + (unsigned)d->count[t];
  }
  for (int t = d->n; t < CODON_CAST_MAX; ++t) { a.src[t] = nullptr; a.dtype[t] = 0; a.start[t + 1] = a.start[d->n]; }
  const unsigned total = a.start[d->n];
  // a few thousand parameter values, or -- a 16-bit model's input images ride along (codon_amd/model.py) -- two images
  const unsigned blocks = (total + 255) / 256;
  hipLaunchKernelGGL(cast_multi_kernel, dim3(blocks < 2048 ? blocks : 2048), dim3(256), 0, stream, a, dst);
  return check_launch("cast_multi_kernel");
}

}  // namespace codon
