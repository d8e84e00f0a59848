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
// stream is the reference's: one MT19937 state per replica with CPython's seeding (random.seed(int) = init_by_array of
// the integer's 32-bit digits) and CPython's draw mapping -- random() = (a >> 5, b >> 6) 53-bit, choice() =
// _randbelow_with_getrandbits (rejection on getrandbits(bit_length(n))), choices([a, b], weights) = one random()
// bisected on the cumulative weights -- so that a replica re-seeded with its index every exchange step
// (utils/replica_exchange_monte_carlo.py:227-228,250) draws the same positions and letters as the reference's worker
// does (tests/test_host_golden.py: the 900 recorded proposals; the reference's own dependence on str-set iteration
// order for the second letter of a pair move and the snake state order is the only exemption).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <functional>
#include <thread>
#include <vector>
#include <sched.h>

namespace drna_host {

// ---- MT19937 with CPython's interface (Lib/random.py, Modules/_randommodule.c; the public Mersenne Twister algorithm)
constexpr int MT_N = 624, MT_M = 397;
constexpr int RNG_WORDS = MT_N + 1;           // state words per replica in the C ABI: mt[624] + index

struct Mt {
  uint32_t* mt;                               // 624 state words followed by the index
  uint32_t& idx() { return mt[MT_N]; }
};

static inline void mt_init_genrand(Mt g, uint32_t s) {
  g.mt[0] = s;
  for (int i = 1; i < MT_N; i++) g.mt[i] = 1812433253u * (g.mt[i - 1] ^ (g.mt[i - 1] >> 30)) + (uint32_t)i;
  g.idx() = MT_N;
}
static inline void mt_init_by_array(Mt g, const uint32_t* key, int len) {
  mt_init_genrand(g, 19650218u);
  int i = 1, j = 0;
  for (int k = MT_N > len ? MT_N : len; k; k--) {
    g.mt[i] = (g.mt[i] ^ ((g.mt[i - 1] ^ (g.mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
    i++; j++;
    if (i >= MT_N) { g.mt[0] = g.mt[MT_N - 1]; i = 1; }
    if (j >= len) j = 0;
  }
  for (int k = MT_N - 1; k; k--) {
    g.mt[i] = (g.mt[i] ^ ((g.mt[i - 1] ^ (g.mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
    i++;
    if (i >= MT_N) { g.mt[0] = g.mt[MT_N - 1]; i = 1; }
  }
  g.mt[0] = 0x80000000u;
}
// random.seed(a) for a non-negative integer a < 2^64: the key is a's 32-bit digits, least significant first (one digit for 0)
static inline void mt_seed_int(Mt g, uint64_t a) {
  uint32_t key[2] = {(uint32_t)(a & 0xffffffffull), (uint32_t)(a >> 32)};
  mt_init_by_array(g, key, key[1] ? 2 : 1);
}
static inline uint32_t mt_genrand(Mt g) {
  static const uint32_t mag01[2] = {0u, 0x9908b0dfu};
  if (g.idx() >= (uint32_t)MT_N) {
    int kk;
    uint32_t y;
    for (kk = 0; kk < MT_N - MT_M; kk++) {
      y = (g.mt[kk] & 0x80000000u) | (g.mt[kk + 1] & 0x7fffffffu);
      g.mt[kk] = g.mt[kk + MT_M] ^ (y >> 1) ^ mag01[y & 1u];
    }
    for (; kk < MT_N - 1; kk++) {
      y = (g.mt[kk] & 0x80000000u) | (g.mt[kk + 1] & 0x7fffffffu);
      g.mt[kk] = g.mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ mag01[y & 1u];
    }
    y = (g.mt[MT_N - 1] & 0x80000000u) | (g.mt[0] & 0x7fffffffu);
    g.mt[MT_N - 1] = g.mt[MT_M - 1] ^ (y >> 1) ^ mag01[y & 1u];
    g.idx() = 0;
  }
  uint32_t y = g.mt[g.idx()++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}
// random.random()
static inline double rnd01(Mt g) {
  const uint32_t a = mt_genrand(g) >> 5, b = mt_genrand(g) >> 6;
  return ((double)a * 67108864.0 + (double)b) * (1.0 / 9007199254740992.0);
}
// random.choice(seq) index = Random._randbelow_with_getrandbits(n), 0 < n < 2^32
static inline int rnd_below(Mt g, int n) {
  if (n <= 0) return 0;
  const int k = 32 - __builtin_clz((unsigned)n);          // n.bit_length()
  uint32_t r = mt_genrand(g) >> (32 - k);
  while (r >= (uint32_t)n) r = mt_genrand(g) >> (32 - k);
  return (int)r;
}
// random.choices([first, second], weights=[w, 1 - w])[0]: 0 = first
static inline int rnd_choices2(Mt g, double w) {
  const double c0 = w, c1 = w + (1.0 - w);                // itertools.accumulate
  const double x = rnd01(g) * (c1 + 0.0);
  return x < c0 ? 0 : 1;                                  // bisect_right(cum_weights, x, 0, 1)
}

// bracket families of the reference's SimScore (utils/sim_score.py:28-59): code = family << 1 | opening, 0xff = no bracket
struct BracketLut {
  unsigned char code[256];
  BracketLut() {
    static const char OP[] = "([<{ABCDE", CL[] = ")]>}abcde";
    for (int k = 0; k < 256; k++) code[k] = 0xff;
    for (int k = 0; k < 9; k++) { code[(unsigned char)OP[k]] = (unsigned char)(k << 1 | 1); code[(unsigned char)CL[k]] = (unsigned char)(k << 1); }
  }
};
static inline const BracketLut& bracket_lut() { static const BracketLut lut; return lut; }

// partner[i] = j or -1; returns false on unbalanced input.  One stack per bracket family, kept as linked lists through
// `link` (n ints) so that a call touches O(n) memory whatever the number of families
static inline bool pair_table(const char* s, int n, int* partner) {
  if (n > 2048) return false;
  const unsigned char* code = bracket_lut().code;
  int top[9] = {-1, -1, -1, -1, -1, -1, -1, -1, -1};
  int link[2048];
  for (int i = 0; i < n; i++) {
    partner[i] = -1;
    const unsigned c = code[(unsigned char)s[i]];
    if (c == 0xff) continue;
    const int f = (int)(c >> 1);
    if (c & 1u) { link[i] = top[f]; top[f] = i; }
    else {
      const int o = top[f];
      if (o < 0) return false;
      top[f] = link[o];
      partner[o] = i; partner[i] = o;
    }
  }
  for (int f = 0; f < 9; f++)
    if (top[f] >= 0) return false;
  return true;
}

// confusion matrix per position of a structure against the reference structure and the three rounded metrics
// (utils/sim_score.py:62-147): a correct pair counts twice, a base paired to another partner is a false negative
struct SimMetrics { double mcc, recall, precision; };
static inline double py_round3_slow(double x) {
  char buf[64];
  snprintf(buf, sizeof buf, "%.3f", x);
  return strtod(buf, nullptr);
}
// CPython round(x, 3): the correctly rounded decimal, like glibc printf("%.3f") read back.  Away from a tie the rounded
// thousandths r are nearbyint(1000 x) and the nearest double to the decimal r / 1000 is the IEEE quotient r / 1000.0 (both
// operands exact); within 1e-7 of a tie (where the product's own rounding could decide) the decimal conversion decides
static inline double py_round3(double x) {
  if (!(std::fabs(x) < 1e5)) return py_round3_slow(x);
  const double y = x * 1000.0, f = y - std::floor(y);
  if (std::fabs(f - 0.5) < 1e-7) return py_round3_slow(x);
  return std::nearbyint(y) / 1000.0;
}

static inline SimMetrics sim_metrics(const int* pr, const int* pq, int n) {
  long tp = 0, fp = 0, fn = 0, tn = 0;
  for (int i = 0; i < n; i++) {
    if (pr[i] == pq[i]) { if (pr[i] != -1) tp++; else tn++; }
    else if (pr[i] == -1) fp++;
    else fn++;
  }
  double num, den;
  if (tp == 0 && fp == 0 && fn == 0 && tn != 0) { num = 1; den = 1; }
  else {
    num = (double)(tp * tn) - (double)(fp * fn);
    den = std::sqrt((double)((tp + fp) * (tp + fn) * (tn + fn) * (tn + fp)));
  }
  return SimMetrics{py_round3(num / (den + 0.00001)), py_round3((double)tp / ((double)(tp + fn) + 0.001)),
                    py_round3((double)tp / ((double)(tp + fp) + 0.001))};
}

// ---- worker threads of the Monte-Carlo inner loop (drna_mc_run).  The per-replica host work of an iteration (SimScore,
// Metropolis, state update, next proposal: ~2 us per replica) sits between two kernel launches, i.e. on the critical path of
// every iteration, and replicas are independent (own random stream, own state), so it is dealt to T threads in contiguous
// replica ranges.  The workers live for one drna_mc_run call and SPIN between jobs (a futex wake-up costs more than the job);
// the calling thread is worker 0.
struct McPool {
  int T = 1;
  std::vector<std::thread> th;
  std::atomic<int> gen{0}, left{0};
  std::atomic<bool> quit{false};
  std::function<void(int)> job;                  // job(worker index)
  explicit McPool(int threads) : T(threads < 1 ? 1 : threads) {
    for (int w = 1; w < T; w++)
      th.emplace_back([this, w] {
        int seen = 0;
        for (;;) {
          int spins = 0;
          while (gen.load(std::memory_order_acquire) == seen) {
            if (quit.load(std::memory_order_relaxed)) return;
            if (++spins > 4096) { sched_yield(); spins = 0; } else __builtin_ia32_pause();
          }
          seen++;
          job(w);
          left.fetch_sub(1, std::memory_order_acq_rel);
        }
      });
  }
  template <class F> void run(F&& f) {
    if (T == 1) { f(0); return; }
    job = std::forward<F>(f);
    left.store(T - 1, std::memory_order_relaxed);
    gen.fetch_add(1, std::memory_order_release);
    job(0);
    while (left.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();
  }
  ~McPool() {
    quit.store(true);
    for (auto& t : th) t.join();
  }
};
// CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (a GPU lease shows the host's 256 hardware
// threads and allows a share of them)
static inline int usable_cpus() {
  cpu_set_t set;
  int n = 1;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[32]; double per = 0;
    if (fscanf(f, "%31s %lf", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) {
      const int c = (int)(atof(q) / per + 0.5);
      if (c >= 1 && c < n) n = c;
    }
    fclose(f);
  }
  return n < 1 ? 1 : n;
}

}  // namespace drna_host
