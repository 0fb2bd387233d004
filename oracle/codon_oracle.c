/* Plain-C restatement of CODONNet.forward -- TEST INFRASTRUCTURE ONLY (tests/, smoke, bench cpu leg).
 *
 * Independent of PyTorch: direct convolution loops with double accumulation, written from
 *   /root/reference/CODON_X4/CODON_x4.py:66-132      (CODONNet.forward; CODON_X8 identical; x16: same math)
 *   /root/reference/CODON_X4/CAC_module.py:38-63      (CAC_channel.forward)
 *   /root/reference/CODON_X4/CAC_module.py:78-94      (ChannelPool, CAC_spatial.forward)
 * Pinned by the same fixtures as the torch-CPU restatement (tests/golden/, recorded from the imported reference):
 * tests/test_oracle_c.py.  Activations are stored as float (what the reference's fp32 run stores); every dot
 * product is accumulated in double, so this sits between the reference's fp32 and fp64 runs.
 *
 * Parameters: one flat float buffer in the reference's state_dict order for the 19 convs
 *   input(64,1,3,3) conv_input(64,64,3,3) conv1(64,64,3,3) conv2(64,64,5,5) conv3(128,128,5,5) confuse(64,128,1,1)
 *   input_c conv_input_c conv4(64,64,5,5) conv5(64,64,3,3) conv6(128,128,5,5) confuse_c(64,128,1,1)
 *   conv7(64,128,3,3) conv8(64,64,5,5) conv9(64,64,3,3) conv10(128,128,5,5) confuse_fuse(64,128,1,1)
 *   conv11(64,64,3,3) output(1,64,3,3)
 * then for i = 0..4: attention_c{i}.mlp.1.weight(8,128) .bias(8) .mlp.3.weight(64,8) .bias(64),
 * then for i = 0..4: attention_s{i}.spatial.conv.weight(1,2,5,5).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static void conv2d(const float* x, const float* w, float* y, int B, int Ci, int Co, int H, int W, int k, int relu) {
  const int p = k / 2;
#pragma omp parallel for collapse(2)
  for (int b = 0; b < B; ++b)
    for (int o = 0; o < Co; ++o)
      for (int i = 0; i < H; ++i)
        for (int j = 0; j < W; ++j) {
          double s = 0.0;
          for (int c = 0; c < Ci; ++c)
            for (int di = 0; di < k; ++di) {
              const int ii = i + di - p;
              if (ii < 0 || ii >= H) continue;
              for (int dj = 0; dj < k; ++dj) {
                const int jj = j + dj - p;
                if (jj < 0 || jj >= W) continue;
                s += (double)w[((o * Ci + c) * k + di) * k + dj] * (double)x[((size_t)(b * Ci + c) * H + ii) * W + jj];
              }
            }
          float v = (float)s;
          if (relu && v < 0.f) v = 0.f;
          y[((size_t)(b * Co + o) * H + i) * W + j] = v;
        }
}

static void cat2(const float* a, const float* b, float* y, int B, int Ca, int Cb, size_t HW) {
  for (int n = 0; n < B; ++n) {
    memcpy(y + (size_t)n * (Ca + Cb) * HW, a + (size_t)n * Ca * HW, sizeof(float) * Ca * HW);
    memcpy(y + ((size_t)n * (Ca + Cb) + Ca) * HW, b + (size_t)n * Cb * HW, sizeof(float) * Cb * HW);
  }
}

static double sigmoid(double z) { return 1.0 / (1.0 + exp(-z)); }

/* out = out * (ch * sp) + inputs, out_c likewise; Fcat = cat(out_c, out)  (CODON_x4.py:85-118) */
static void cac_block(float* out, float* out_c, const float* inputs, const float* inputs_c, const float* w1,
                      const float* b1, const float* w2, const float* b2, const float* ws, int B, int H, int W) {
  const size_t HW = (size_t)H * W;
  float* F = (float*)malloc(sizeof(float) * B * 128 * HW);
  cat2(out_c, out, F, B, 64, 64, HW);
  float* ch = (float*)malloc(sizeof(float) * B * 64);
  float* comp = (float*)malloc(sizeof(float) * B * 2 * HW);
  float* sp = (float*)malloc(sizeof(float) * B * HW);
  for (int b = 0; b < B; ++b) {
    double avg[128], mx[128];
    for (int c = 0; c < 128; ++c) {                       /* CAC_module.py:43,47 */
      double s = 0.0, m = -INFINITY;
      const float* pl = F + ((size_t)b * 128 + c) * HW;
      for (size_t q = 0; q < HW; ++q) { s += pl[q]; if (pl[q] > m) m = pl[q]; }
      avg[c] = s / (double)HW; mx[c] = m;
    }
    double att[64];
    for (int o = 0; o < 64; ++o) att[o] = 0.0;
    for (int which = 0; which < 2; ++which) {             /* shared MLP on both pools, :30-35,44,48,58-61 */
      const double* v = which ? mx : avg;
      double h[8];
      for (int j = 0; j < 8; ++j) {
        double a = b1[j];
        for (int c = 0; c < 128; ++c) a += (double)w1[j * 128 + c] * v[c];
        h[j] = a > 0 ? a : 0;
      }
      for (int o = 0; o < 64; ++o) {
        double a = b2[o];
        for (int j = 0; j < 8; ++j) a += (double)w2[o * 8 + j] * h[j];
        att[o] += a;
      }
    }
    for (int o = 0; o < 64; ++o) ch[b * 64 + o] = (float)sigmoid(att[o]);   /* :62 */
    for (size_t q = 0; q < HW; ++q) {                     /* ChannelPool :81 (max first, then mean) */
      double s = 0.0, m = -INFINITY;
      for (int c = 0; c < 128; ++c) { const float v = F[((size_t)b * 128 + c) * HW + q]; s += v; if (v > m) m = v; }
      comp[((size_t)b * 2 + 0) * HW + q] = (float)m;
      comp[((size_t)b * 2 + 1) * HW + q] = (float)(s / 128.0);
    }
  }
  conv2d(comp, ws, sp, B, 2, 1, H, W, 5, 0);              /* :88,92 */
  for (size_t q = 0; q < (size_t)B * HW; ++q) sp[q] = (float)sigmoid(sp[q]);   /* :93 */
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < 64; ++c)
      for (size_t q = 0; q < HW; ++q) {
        const float g = ch[b * 64 + c] * sp[(size_t)b * HW + q];             /* :89 */
        const size_t i = ((size_t)b * 64 + c) * HW + q;
        out[i] = out[i] * g + inputs[i];                                     /* :90,118 */
        out_c[i] = out_c[i] * g + inputs_c[i];                               /* :91,117 */
      }
  free(F); free(ch); free(comp); free(sp);
}

/* returns 0 on success.  params: see the header comment.  x, y, out: (B,1,H,W). */
int codon_oracle_forward(const float* params, const float* x, const float* y, float* out, int B, int H, int W) {
  const size_t HW = (size_t)H * W, n64 = (size_t)B * 64 * HW, n128 = 2 * n64;
  const float* p = params;
#define TAKE(name, n) const float* name = p; p += (n);
  TAKE(w_input, 64 * 9) TAKE(w_conv_input, 64 * 64 * 9) TAKE(w1, 64 * 64 * 9) TAKE(w2, 64 * 64 * 25)
  TAKE(w3, 128 * 128 * 25) TAKE(w_confuse, 64 * 128)
  TAKE(w_input_c, 64 * 9) TAKE(w_conv_input_c, 64 * 64 * 9) TAKE(w4, 64 * 64 * 25) TAKE(w5, 64 * 64 * 9)
  TAKE(w6, 128 * 128 * 25) TAKE(w_confuse_c, 64 * 128)
  TAKE(w7, 64 * 128 * 9) TAKE(w8, 64 * 64 * 25) TAKE(w9, 64 * 64 * 9) TAKE(w10, 128 * 128 * 25)
  TAKE(w_confuse_fuse, 64 * 128) TAKE(w11, 64 * 64 * 9) TAKE(w_output, 64 * 9)
  const float* att_c = p; p += 5 * (8 * 128 + 8 + 64 * 8 + 64);
  const float* att_s = p;
#undef TAKE
  float *t = malloc(sizeof(float) * n64), *inputs = malloc(sizeof(float) * n64), *inputs_c = malloc(sizeof(float) * n64);
  float *o = malloc(sizeof(float) * n64), *oc = malloc(sizeof(float) * n64), *a = malloc(sizeof(float) * n64),
        *b_ = malloc(sizeof(float) * n64), *st = malloc(sizeof(float) * n128), *r2 = malloc(sizeof(float) * n128);
  if (!t || !inputs || !inputs_c || !o || !oc || !a || !b_ || !st || !r2) return -1;
  conv2d(x, w_input, t, B, 1, 64, H, W, 3, 1);  conv2d(t, w_conv_input, inputs, B, 64, 64, H, W, 3, 1);      /* :68-69 */
  conv2d(y, w_input_c, t, B, 1, 64, H, W, 3, 1); conv2d(t, w_conv_input_c, inputs_c, B, 64, 64, H, W, 3, 1); /* :71-72 */
  memcpy(o, inputs, sizeof(float) * n64); memcpy(oc, inputs_c, sizeof(float) * n64);
  for (int i = 0; i < 5; ++i) {
    conv2d(o, w1, a, B, 64, 64, H, W, 3, 1); conv2d(o, w2, b_, B, 64, 64, H, W, 5, 1);       /* :75,77 */
    cat2(a, b_, st, B, 64, 64, HW);                                                           /* :79 */
    conv2d(st, w3, r2, B, 128, 128, H, W, 5, 1); conv2d(r2, w_confuse, o, B, 128, 64, H, W, 1, 0);  /* :81,84 */
    conv2d(oc, w4, a, B, 64, 64, H, W, 5, 1); conv2d(oc, w5, b_, B, 64, 64, H, W, 3, 1);     /* :78,76 */
    cat2(a, b_, st, B, 64, 64, HW);                                                           /* :80 */
    conv2d(st, w6, r2, B, 128, 128, H, W, 5, 1); conv2d(r2, w_confuse_c, oc, B, 128, 64, H, W, 1, 0); /* :82,83 */
    const float* ac = att_c + (size_t)i * (8 * 128 + 8 + 64 * 8 + 64);
    cac_block(o, oc, inputs, inputs_c, ac, ac + 1024, ac + 1032, ac + 1544, att_s + i * 50, B, H, W);
  }
  cat2(o, oc, st, B, 64, 64, HW);                                                             /* :119 */
  float* fuse = inputs;  /* reuse */
  conv2d(st, w7, fuse, B, 128, 64, H, W, 3, 1);                                               /* :120 */
  memcpy(o, fuse, sizeof(float) * n64);
  for (int i = 0; i < 3; ++i) {
    conv2d(o, w8, a, B, 64, 64, H, W, 5, 1); conv2d(o, w9, b_, B, 64, 64, H, W, 3, 1);       /* :123-124 */
    cat2(a, b_, st, B, 64, 64, HW);
    conv2d(st, w10, r2, B, 128, 128, H, W, 5, 1); conv2d(r2, w_confuse_fuse, o, B, 128, 64, H, W, 1, 0);
    for (size_t q = 0; q < n64; ++q) o[q] += fuse[q];                                         /* :128 */
  }
  conv2d(o, w11, t, B, 64, 64, H, W, 3, 1);                                                   /* :129 */
  conv2d(t, w_output, out, B, 64, 1, H, W, 3, 0);                                             /* :130 */
  for (size_t q = 0; q < (size_t)B * HW; ++q) out[q] += x[q];                                 /* :131 */
  free(t); free(inputs); free(inputs_c); free(o); free(oc); free(a); free(b_); free(st); free(r2);
  return 0;
}
