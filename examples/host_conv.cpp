// C++ host (no Python, no torch) calling libcodon_hip.so through include/codon_hip.h -- the snippet of
// INTEGRATION.md section 2 as a complete program.  One conv5x5 128->128 + ReLU (the op of
// /root/reference/CODON_X4/CODON_x4.py:81, `self.relu(self.conv3(...))`) on deterministic inputs; the fp32 result
// is written to argv[1] so tests/test_boundary.py can compare it with the ctypes path bit for bit.
//
//   hipcc --offload-arch=gfx950 -I include examples/host_conv.cpp -L codon_amd -lcodon_hip -Wl,-rpath,$PWD/codon_amd -o host_conv
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#include "codon_hip.h"

#define HIP_OK(e)                                                                 \
  do {                                                                            \
    hipError_t err_ = (e);                                                        \
    if (err_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(err_)); \
      return 2;                                                                   \
    }                                                                             \
  } while (0)

static float lcg(uint32_t i, uint32_t salt) {  // same generator as tests/test_boundary.py::_lcg
  const uint32_t u = i * 2654435761u + salt;
  return (float)((u >> 8) & 0xffff) / 65536.0f - 0.5f;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s out.f32\n", argv[0]); return 2; }
  const int B = 2, H = 19, W = 45, C = 128, K = 5;
  if (codon_abi_version() != CODON_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 2; }
  std::vector<float> x((size_t)B * C * H * W), w((size_t)C * C * K * K), y(x.size());
  for (size_t i = 0; i < x.size(); ++i) x[i] = lcg((uint32_t)i, 17u);
  for (size_t i = 0; i < w.size(); ++i) w[i] = lcg((uint32_t)i, 99u) * 0.05f;

  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  float *x_dev, *w_oihw_dev, *w_packed_dev, *y_dev;
  HIP_OK(hipMalloc(&x_dev, x.size() * 4));
  HIP_OK(hipMalloc(&w_oihw_dev, w.size() * 4));
  HIP_OK(hipMalloc(&w_packed_dev, codon_conv_packed_weight_bytes(C, C, K, CODON_F32)));
  HIP_OK(hipMalloc(&y_dev, y.size() * 4));
  HIP_OK(hipMemcpyAsync(x_dev, x.data(), x.size() * 4, hipMemcpyHostToDevice, stream));
  HIP_OK(hipMemcpyAsync(w_oihw_dev, w.data(), w.size() * 4, hipMemcpyHostToDevice, stream));

  codon_conv_desc d = {B, H, W, C, C, K, C, 0, C, 0, 0, 0, CODON_CONV_RELU, CODON_F32};
  if (codon_conv_pack_weight(w_oihw_dev, w_packed_dev, C, C, K, CODON_PACK_FWD, CODON_F32, stream) != CODON_OK ||
      codon_conv2d_fwd(&d, x_dev, w_packed_dev, y_dev, nullptr, stream) != CODON_OK) {
    fprintf(stderr, "%s\n", codon_last_error_string());
    return 1;
  }
  // error convention: a bad descriptor is refused with a status and a message, nothing is launched
  codon_conv_desc bad = d;
  bad.ksize = 4;
  if (codon_conv2d_fwd(&bad, x_dev, w_packed_dev, y_dev, nullptr, stream) == CODON_OK ||
      codon_last_error_string()[0] == 0) {
    fprintf(stderr, "ksize 4 was not refused\n");
    return 1;
  }
  HIP_OK(hipMemcpyAsync(y.data(), y_dev, y.size() * 4, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));
  FILE* f = fopen(argv[1], "wb");
  if (!f || fwrite(y.data(), 4, y.size(), f) != y.size()) { fprintf(stderr, "cannot write %s\n", argv[1]); return 2; }
  fclose(f);
  double s = 0;
  for (float v : y) s += v;
  printf("host_conv: %zu outputs, sum %.6f\n", y.size(), s);
  return 0;
}
