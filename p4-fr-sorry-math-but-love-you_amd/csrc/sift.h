// DecodingManager rules on the device (postprocessing/postprocessing.py:293-388 MemoryNode).
// rules: int32 [V + 8] = per-token word (flag bits | run-length limit << 8) then ids {sos, eos, "", "_", "{", "}", -, -}
// state: {current token, run length, #"{", #"}"} per sample.
#pragma once
#include "common.h"

enum { SIFT_NEXT_UNDERBAR = 1, SIFT_NEXT_LBRACKET = 2, SIFT_NOT_UNDERBAR = 4, SIFT_NOT_LBRACKET = 8, SIFT_NOT_INITIAL = 16 };

struct SiftState { int cur, series, nl, nr; };

// MemoryNode._look_back (:337-388): is token c forbidden at the next step?
DEVI bool sift_forbidden(int c, const SiftState& st, const int32_t* rules, int V) {
  const int sos = rules[V], eos = rules[V + 1], empty = rules[V + 2], under = rules[V + 3], lbr = rules[V + 4], rbr = rules[V + 5];
  if (c == sos || c == empty) return true;
  if (c == rbr && st.nl == st.nr) return true;
  if (st.cur == eos) return false;
  if (st.cur == sos) return (rules[c] & SIFT_NOT_INITIAL) != 0;
  const int w = rules[st.cur];
  if (w & SIFT_NEXT_UNDERBAR) return c != under;
  if (w & SIFT_NEXT_LBRACKET) return c != lbr;
  if ((w & SIFT_NOT_UNDERBAR) && c == under) return true;
  if ((w & SIFT_NOT_LBRACKET) && c == lbr) return true;
  const int lim = w >> 8;
  return lim > 0 && st.series >= lim && c == st.cur;
}

// MemoryNode.record (:317-335)
DEVI void sift_record(SiftState& st, int tok, const int32_t* rules, int V) {
  st.series = (st.cur == tok) ? st.series + 1 : 1;
  if (tok == rules[V + 4]) st.nl += 1;
  else if (tok == rules[V + 5]) st.nr += 1;
  st.cur = tok;
}

// One wavefront: softmax of x[0..V) (fp32, like F.softmax), forbidden entries zeroed, written to out[0..V); returns the
// argmax of the masked probabilities (lowest index wins ties, like torch.argmax).  x and out may alias.  All 64 lanes call.
DEVI int sift_wave(const float* x, float* out, int V, const SiftState& st, const int32_t* rules, int lane) {
  float mx = -INFINITY;
  for (int c = lane; c < V; c += 64) mx = fmaxf(mx, x[c]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int c = lane; c < V; c += 64) sum += expf(x[c] - mx);
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  float best = -1.f;
  int bi = 0x7fffffff;
  for (int c = lane; c < V; c += 64) {
    float pr = expf(x[c] - mx) * inv;
    if (sift_forbidden(c, st, rules, V)) pr = 0.f;
    out[c] = pr;
    if (pr > best) { best = pr; bi = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  return bi;
}
