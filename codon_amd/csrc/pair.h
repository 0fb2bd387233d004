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
};
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
                     hipStream_t stream) {
  static_assert(sizeof(P) <= sizeof(PairCall::blob), "parameter block fits the recorder");
  PairRecorder* r = pair_recorder();
  if (!r) return 0;
  CODON_REQUIRE(r->n < 2, CODON_ERR_BAD_ARG, "conv pair: a third conv call inside one codon_conv_pair_begin / _end bracket");
  PairCall& c = r->call[r->n++];
  memcpy(c.blob, &p, sizeof(P));
  c.nblk = p.nblk; c.tiles_x = p.tiles_x; c.tiles_y = p.tiles_y;
  c.stream = stream;
  c.single = single; c.pair = pair;
  return 1;
}

}  // namespace codon
