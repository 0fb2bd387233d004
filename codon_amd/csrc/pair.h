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
  int (*single)(const void*, hipStream_t);
  int (*pair)(const void*, const void*, hipStream_t);   // same pointer = same kernel variant
};
struct PairRecorder {
  bool active = false;
  int n = 0;
  PairCall call[2];
};
PairRecorder* pair_recorder();               // this thread's recorder while a bracket is open, else nullptr

// hold `p` back if a bracket is open (true), else the caller launches it itself
template <class P>
inline bool pair_hold(const P& p, int (*single)(const void*, hipStream_t), int (*pair)(const void*, const void*, hipStream_t)) {
  static_assert(sizeof(P) <= sizeof(PairCall::blob), "parameter block fits the recorder");
  PairRecorder* r = pair_recorder();
  if (!r || r->n >= 2) return false;
  PairCall& c = r->call[r->n++];
  memcpy(c.blob, &p, sizeof(P));
  c.nblk = p.nblk; c.tiles_x = p.tiles_x; c.tiles_y = p.tiles_y;
  c.single = single; c.pair = pair;
  return true;
}

}  // namespace codon
