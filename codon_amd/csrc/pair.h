// codon_conv_pair_begin / codon_conv_pair_end (include/codon_hip.h): between the two calls, on one host thread, the conv
// launchers do not launch -- they leave their filled parameter block here, with a function that launches it alone and one
// that launches two blocks of the same kernel variant as a single grid.  Shared by the 16-bit (conv_c8.hip) and the fp32
// (conv_mfma_f32.hip) launchers; the state itself lives in codon_abi.hip.
#pragma once
#include <string.h>

#include "codon_common.h"

namespace codon {

struct PairCall {
  alignas(16) unsigned char blob[384];       // ConvC8Params / ConvParams
  int nblk, tiles_x, tiles_y;
  hipStream_t stream;                        // the stream the held call was given: pair_end pairs only calls of ITS stream
  int (*single)(const void*, hipStream_t);
  int (*pair)(const void*, const void*, hipStream_t);   // same pointer = same kernel variant
  // round 6: a conv5x5 64->64 and a conv3x3 64->64 on the SAME input (conv8 | conv9 of the fusion trunk,
  // /root/reference/CODON_X4/CODON_x4.py:123-124) are two DIFFERENT kernel bodies; held together they leave as one grid of
  // both kinds of workgroup (mix53).  mix_kind: 0 = none, odd = the 5x5 of a family, the next even number = its 3x3.
  int mix_kind;
  int (*mix)(const void* five, const void* three, hipStream_t);     // set by the 5x5 call
};
enum { MIX_NONE = 0, MIX_F32_CSPLIT_5 = 1, MIX_F32_CSPLIT_3 = 2, MIX_C8F16_5 = 3, MIX_C8F16_3 = 4, MIX_C8BF16_5 = 5, MIX_C8BF16_3 = 6 };
struct PairRecorder {
  bool active = false;
  int n = 0;
  PairCall call[2];
};
PairRecorder* pair_recorder();               // this thread's recorder while a bracket is open, else nullptr

// Hold `p` back if a bracket is open: 1 = held (the caller returns CODON_OK), 0 = no bracket (the caller launches it itself),
// CODON_ERR_BAD_ARG = a THIRD call the pair form covers inside one bracket -- launching it now would put it ahead of the two
// held ones, so it is refused.
template <class P>
inline int pair_hold(const P& p, int (*single)(const void*, hipStream_t), int (*pair)(const void*, const void*, hipStream_t),
                     hipStream_t stream, int mix_kind = MIX_NONE, int (*mix)(const void*, const void*, hipStream_t) = nullptr) {
  static_assert(sizeof(P) <= sizeof(PairCall::blob), "parameter block fits the recorder");
  PairRecorder* r = pair_recorder();
  if (!r) return 0;
  CODON_REQUIRE(r->n < 2, CODON_ERR_BAD_ARG, "conv pair: a third conv call inside one codon_conv_pair_begin / _end bracket");
  PairCall& c = r->call[r->n++];
  memcpy(c.blob, &p, sizeof(P));
  c.nblk = p.nblk; c.tiles_x = p.tiles_x; c.tiles_y = p.tiles_y;
  c.stream = stream;
  c.single = single; c.pair = pair;
  c.mix_kind = mix_kind; c.mix = mix;
  return 1;
}

}  // namespace codon
