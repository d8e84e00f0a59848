// host_driver.hpp -- native host-side pieces of the Monte-Carlo inner loop, batched over replicas
// (plain C++, no device code).  The reference runs these in pure Python once per MC step per replica:
//   SimScore            utils/sim_score.py:28-147        (0.8 ms at L=200, measured in the survey)
//   get_mutation_position / expand_cases / mutate_sequence   utils/sequence_utils.py:926-1136 (0.4 ms)
//   mc_delta            utils/replica_exchange_monte_carlo.py:26-57
// With the folds on the GPU (~1 ms per batch of 64) that Python would be 3-4x the fold time, so the same
// rules are provided here for all R replicas per call.  Semantics kept: confusion matrix per
// position and the rounding of mcc / recall / precision; targeted positions = false negatives + false
// positives of the current MFE structure widened by +-3 (position 0 never enters),
// chosen with the per-shelf probability, else a uniform mutable position; unpaired positions change to a
// different allowed letter, paired positions change together to a compatible (WC or GU) pair.  The random
// stream is one splitmix64 state per replica (the reference uses one Python Mersenne stream per replica,
// re-seeded every exchange step); draws are not bit-compatible with CPython's, which the reference's own
// set-ordering dependence makes moot (see desirna_amd/design.py).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace drna_host {

static inline uint64_t splitmix64(uint64_t& s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static inline double rnd01(uint64_t& s) { return (double)(splitmix64(s) >> 11) * (1.0 / 9007199254740992.0); }
static inline int rnd_below(uint64_t& s, int n) { return (int)(rnd01(s) * n) % (n > 0 ? n : 1); }

static inline int bracket_family(char ch, bool& open) {
  static const char OP[] = "([<{ABCDE", CL[] = ")]>}abcde";
  for (int k = 0; k < 9; k++) {
    if (ch == OP[k]) { open = true; return k; }
    if (ch == CL[k]) { open = false; return k; }
  }
  return -1;
}

// partner[i] = j or -1; returns false on unbalanced input
static inline bool pair_table(const char* s, int n, int* partner) {
  int stk[9][2048];
  int sp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (n > 2048) return false;
  for (int i = 0; i < n; i++) {
    partner[i] = -1;
    bool open;
    const int f = bracket_family(s[i], open);
    if (f < 0) continue;
    if (open) stk[f][sp[f]++] = i;
    else {
      if (!sp[f]) return false;
      const int o = stk[f][--sp[f]];
      partner[o] = i; partner[i] = o;
    }
  }
  for (int f = 0; f < 9; f++)
    if (sp[f]) return false;
  return true;
}

static inline double py_round3(double x) {   // CPython round(x, 3): correctly rounded decimal, like glibc printf
  char buf[64];
  snprintf(buf, sizeof buf, "%.3f", x);
  return strtod(buf, nullptr);
}

}  // namespace drna_host
