// fold_common.hpp -- device helpers shared by the gfx950 fold kernels (wave64).
#pragma once
#include <stdint.h>

#include "tables.hpp"
// the hardware primitives (agent-scope loads / stores, counted waits, the spin clock, ds_bpermute, the fp64 matrix instruction):
// gfx950_prims.hpp -- or the CPU stand-ins the kernels' test emulation supplies (tests/emu; the ONLY place the product knows of it)
#ifdef DRNA_PRIMS_HEADER
#include DRNA_PRIMS_HEADER
#else
#include "gfx950_prims.hpp"
#endif

namespace drna {

constexpr int WAVE = 64;
constexpr int MAXN = 2048;        // static LDS sizing of the per-sequence arrays
constexpr int PART_ITEMS = 32;    // (cell-block, chunk) work items per diagonal held in LDS

// tower entries (generic interior loops carried per nested-cell tower): residues of the inner diagonal
constexpr int GRES = 28;          // 27 live entries + 1 spare
enum : int { TW_LIVE = 0, TW_BIRTH = 1, TW_KILL = 2, TW_DEAD = 3 };   // flag in the low bits of a tower-table word

// kernel status codes (per sequence)
enum : int { ST_OK = 0, ST_BAD_CHAR = 1, ST_TRACEBACK = 2, ST_PF_RANGE = 3, ST_SYNC = 4 };

// Ragged batches (drna_score_ragged): sequences of different lengths in one launch.  idx maps a workgroup to its sequence
// (the host sorts by length, longest first, and splits the list between the LDS-resident and the general kernels), len is
// the sequence's length, off its byte offset in the concatenated sequence / structure buffers.  All null = uniform batch.
struct Ragged {
  const int* idx = nullptr;
  const int* len = nullptr;
  const int* off = nullptr;
};

// ---- two-workgroup kernel (fold_mfe_dual.hpp): one sequence is folded by a MAIN workgroup (finalize,
// towers, near shapes) and a HELPER workgroup on another CU (multiloop splits, far shapes) that runs a few diagonals
// behind on rows the main one publishes.  Cross-CU visibility follows the CDNA4 guide: payload by agent-scope (sc1,
// write-through / L1-bypassing) stores and loads, every storing wave drains vmcnt, workgroup barrier, then ONE lane stores
// the flag; the consumer polls the flag from one lane with sc1 loads and loads the payload only after it has matched.
// Flag values grow monotonically over diagonals, pseudoknot rounds and calls: ((epoch * 8 + round) << 10) + diagonal,
// compared wrap-safe, so nothing is ever reset; DONE closes a call.  Every wait is bounded (ST_SYNC on expiry).
// The helper works on diagonal D with rows <= D - DLAG only: multiloop split points further than KEDGE from either end of the
// range, loop shapes whose inner pair is at least DLAG diagonals back.  The slack this buys (DLAG - 3 steps of the main
// workgroup) has to cover the round trip through L2 / fabric (~3 us) plus the helper's own step.
constexpr int DLAG = 10, KEDGE = DLAG - 5;      // (8 / 9 / 10 / 11 / 12 measured: 0.439 / 0.442 / 0.440 / 0.440 / 0.458 ms)
constexpr int XP = 224;           // row pitch of the exchange tables for n <= 200: rows are whole 128-byte lines
struct DualLink {
  int* flagA = nullptr;           // written by the main workgroup: base + last published diagonal (base + 3 = round prologue done)
  int* flagB = nullptr;           // written by the helper: base + last diagonal whose results are published
  int32_t* xs = nullptr;          // main -> helper: pairing codes Sp[0 .. n+1] of the round (4 = masked)
  void* xa = nullptr;             // main -> helper rows (ring word, fML)
  void* xb = nullptr;             // helper -> main rows (split minima, far-shape minima)
  int base = 0;                   // ((epoch * 8 + round) << 10)
  int epoch = 0;
};
__device__ __forceinline__ int dual_base(int epoch, int round) { return (int)((((unsigned)epoch * 8u + (unsigned)round) << 10)); }   // wraps: compares are wrap-safe
__device__ __forceinline__ int dual_done(int epoch) { return (int)((((unsigned)epoch * 8u + 7u) << 10) + 1023u); }
__device__ __forceinline__ bool flag_ge(int v, int target) { return (int)((unsigned)v - (unsigned)target) >= 0; }

// all lanes of the calling wave poll the same word (one request); returns false when the wait expired
__device__ __forceinline__ bool wait_flag_wave(const int* flag, int target) {
  SpinClock clk;
  for (;;) {
    if (flag_ge(__builtin_amdgcn_readfirstlane(ld_agent(flag)), target)) return true;
    if (clk.expired()) return false;
    spin_pause();
  }
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    int w = __shfl_xor(v, o);
    v = w < v ? w : v;
  }
  return v;
}

__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// fixed-order butterfly sum: every lane ends with the same bits
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Workgroup barrier that orders LDS traffic only: global stores stay in flight across it (a plain
// __syncthreads() also drains vmcnt, which puts HBM/L2 store latency on the per-diagonal critical path).
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// value of lane `l` (wave-uniform index) as a scalar: small tables live one entry per lane
__device__ __forceinline__ int lane_table(int v, int l) { return __builtin_amdgcn_readlane(v, l); }

// Keep a wave-uniform index in a VGPR (identity quad_perm DPP move).  hipcc otherwise turns a uniform-address
// LDS read into ds_read + s_waitcnt lgkmcnt(0) + v_readfirstlane: a full LDS round trip per value, serialised.
__device__ __forceinline__ int as_vector(int x) { return __builtin_amdgcn_update_dpp(x, x, 0xE4, 0xF, 0xF, false); }

// Pin a loop-invariant value loaded from global memory in registers: without this hipcc re-executes the load
// inside the diagonal loop (cheaper in registers, but a full L2 round trip per diagonal on the critical path).
__device__ __forceinline__ int keep_i32(int x) { return as_vector(x); }

// kind of plan entry e from the first entries of the kinds (scalar compares; entries past the last special kind are generic)
__device__ __forceinline__ int plan_kind(const int (&seg)[PK_NKINDS], int e) {
  int k = 0;
#pragma unroll
  for (int x = 1; x < PK_NKINDS; x++) k += e >= seg[x];
  return k;
}
__device__ __forceinline__ double keep_f64(double x) {
  return __hiloint2double(as_vector(__double2hiint(x)), as_vector(__double2loint(x)));
}

// Pop one item index from a workgroup work queue in LDS (lane 0 does the atomic, the wave gets the value)
__device__ __forceinline__ int queue_pop(int* head, int lane) {
  int it = 0;
  if (lane == 0) it = atomicAdd(head, 1);
  return __builtin_amdgcn_readfirstlane(it);
}

// ---- buffer loads: 32-bit per-lane byte offset + wave-uniform (SGPR) byte offset, so an unrolled strided sweep costs
// one VALU add per running offset instead of a 64-bit address computation per load
template <typename RS>
__device__ __forceinline__ double buf_load_f64(RS rsrc, int voff, int soff) {
  const auto v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, 0);
  return __hiloint2double((int)v[1], (int)v[0]);
}

template <typename RS>
__device__ __forceinline__ int buf_load_i32(RS rsrc, int voff, int soff) {
  return (int)__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0);
}

struct f64x2 { double x, y; };
template <typename RS>
__device__ __forceinline__ f64x2 buf_load_f64x2(RS rsrc, int voff, int soff) {
  const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
  return f64x2{__hiloint2double((int)v[1], (int)v[0]), __hiloint2double((int)v[3], (int)v[2])};
}

struct i32x4 { int x, y, z, w; };
template <typename RS>
__device__ __forceinline__ i32x4 buf_load_i32x4(RS rsrc, int voff, int soff) {
  const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
  return i32x4{(int)v[0], (int)v[1], (int)v[2], (int)v[3]};
}

// ---- DPP wave reductions (no LDS traffic, fixed order => bit-reproducible)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, true);
  return v + __hiloint2double(hi, lo);
}
// inclusive scan inside each row of 16 lanes, then row totals chained: lane 63 ends with the wave total
__device__ __forceinline__ double wave_total_f64_lane63(double v) {
  v = dpp_add_f64<0x111, 0xF>(v);   // row_shr:1
  v = dpp_add_f64<0x112, 0xF>(v);   // row_shr:2
  v = dpp_add_f64<0x114, 0xF>(v);   // row_shr:4
  v = dpp_add_f64<0x118, 0xF>(v);   // row_shr:8
  v = dpp_add_f64<0x142, 0xA>(v);   // row_bcast:15 into rows 1 and 3
  v = dpp_add_f64<0x143, 0xC>(v);   // row_bcast:31 into rows 2 and 3
  return v;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_min_i32(int v) {
  // lanes without a source (row edge / masked row) see their own value: min is idempotent
  const int o = __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xF, false);
  return o < v ? o : v;
}
// lane 63 ends with the wave minimum
__device__ __forceinline__ int wave_min_i32_lane63(int v) {
  v = dpp_min_i32<0x111, 0xF>(v);
  v = dpp_min_i32<0x112, 0xF>(v);
  v = dpp_min_i32<0x114, 0xF>(v);
  v = dpp_min_i32<0x118, 0xF>(v);
  v = dpp_min_i32<0x142, 0xA>(v);
  v = dpp_min_i32<0x143, 0xC>(v);
  return v;
}

// lowest set lane of a ballot, or -1
__device__ __forceinline__ int first_lane(unsigned long long m) { return m ? (__ffsll((long long)m) - 1) : -1; }

__device__ __forceinline__ int enc_nt(char c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'U': case 'u': case 'T': case 't': return 3;
    default: return -1;
  }
}

// pair type from two pairing codes (0..3 = A C G U, 4 = must stay unpaired)
__device__ __forceinline__ int pair_type(int a, int b) {
  // CG=1 GC=2 GU=3 UG=4 AU=5 UA=6
  if (a > 3 || b > 3) return 0;
  const int s = a * 4 + b;
  // packed 4-bit LUT: index s -> type
  // (A,U)=3 ->5 ; (C,G)=6 ->1 ; (G,C)=9 ->2 ; (G,U)=11 ->3 ; (U,A)=12 ->6 ; (U,G)=14 ->4
  const unsigned long long lut = (5ull << (3 * 4)) | (1ull << (6 * 4)) | (2ull << (9 * 4)) | (3ull << (11 * 4)) |
                                 (6ull << (12 * 4)) | (4ull << (14 * 4));
  return (int)((lut >> (s * 4)) & 15ull);
}

__device__ __forceinline__ int rtype_of(int t) {
  // {0,2,1,4,3,6,5,7}
  const unsigned lut = 0x75634120u;
  return (int)((lut >> (t * 4)) & 15u);
}

// ---- strip kernels (fold_pf_strip.hpp, fold_mfe_strip.hpp): one sequence folded by several workgroups, each owning a strip
// of columns i; dependencies between strips run one way (towards smaller i), see fold_pf_strip.hpp
// Empty blocks per sequence behind its strips.  Workgroups of an XCD are dealt to its shader engines in turn and start in order:
// with four strips per sequence and no padding, strip s of EVERY sequence lands on engine s, the long-lived last strips queue
// behind each other on a quarter of the CUs and the rest idles (measured: tools/strip_clocks.py).  One empty block per
// sequence rotates the assignment.
constexpr int STRIP_PAD = 1;
// ... so that S + pad is ODD: with an even number of blocks per sequence the strips still fall on the same engines in turn
// (three strips + one empty block: strip s of every sequence on engine s again, n = 300 at R = 128: PF 2.58 vs 2.07 ms)
__host__ __device__ inline int strip_pad(int S) { return STRIP_PAD ? ((S & 1) ? 0 : 1) : 0; }
constexpr int STRIP_DONE = 4095, STRIP_FAIL = 4094;   // flag values above every diagonal
constexpr int STRIP_REC = 88;                         // doubles per exchange record
constexpr int STRIP_MAXS = 18;                         // strips per sequence at most
constexpr int STRIP_WMAX = 120;                       // widest strip of the production kernel (1024 threads)
constexpr int STRIP_NMAX = STRIP_MAXS * STRIP_WMAX;   // longest sequence

// Two workgroups per sequence in one launch (mfe_dual_kernel, pf_lds_kernel with helpers): blocks in groups of 16 -- eight main
// roles, then their eight helpers -- so that a sequence's two workgroups are 8 blocks apart, i.e. on ONE XCD when blocks are dealt
// round-robin over the 8 XCDs (speed only, never relied on): what they hand each other meets in that XCD's L2 (fold_fused.hpp has
// the measurement).  Blocks whose sequence is beyond the batch leave at once.
__host__ __device__ inline int pair_grid(int R) { return 16 * ((R + 7) / 8); }
__host__ __device__ inline void pair_block(int b, int& r, int& helper) {
  const int x = b & 15;
  r = (b >> 4) * 8 + (x & 7);
  helper = x >> 3;
}

struct StripLink {
  int* flags = nullptr;      // one 128-byte line per (sequence slot, strip)
  int base = 0;              // epoch << 12
  int nseq = 0;              // sequences of this launch
  int S = 0;                 // strips per sequence
  int pad = 0;               // empty blocks per sequence behind its strips (see strip_pad)
  int fault = 0;             // tests: the top strip of every sequence gives up at once (exercises the engine's fallback)
  int fark = 0;              // MFE strips: multiloop splits in blocked form (tile products + near split points; fold_mfe_strip.hpp)
  const int* idx = nullptr;  // sequence slot -> sequence (ragged batches), or null: slot q is sequence q + r0
  int r0 = 0;
  int* dbg = nullptr;        // diagnostics: 8 words per sequence slot, written by a strip whose wait failed
  long long* clk = nullptr;  // diagnostics: start / end wall clock (100 MHz) of every strip workgroup, [slot][STRIP_MAXS][2]
};
// Layout of the strips of an n-nt sequence.  The LAST strip (columns from 1: it lives through all n diagonals, and its
// multiloop sums are the longest) gets the mean width, at least 32 columns (a halo reaches 31); the strips above it share the
// rest evenly, at most `wmax` columns each.  (Narrowing the last strip to 75 / 55 / 40 % of the mean: 5.07 / 5.13 / 5.30 vs 4.81 ms
// at 400 nt x 256 -- the strips are bound by the latency of a step, not by the last strip's share.)
__host__ __device__ inline int strip_last_width(int n, int S) {
  int w = (n + S / 2) / S;
  return w < 32 ? 32 : w;
}
__host__ __device__ inline int strip_upper_width(int n, int S) { return S > 1 ? (n - strip_last_width(n, S) + S - 2) / (S - 1) : 0; }
__host__ __device__ inline int strip_count(int n, int wmax) {
  for (int S = 2; S <= STRIP_MAXS; S++)
    if (strip_last_width(n, S) <= wmax && strip_upper_width(n, S) <= wmax && strip_last_width(n, S) < n) return S;
  return STRIP_MAXS + 1;
}
// columns [c0, c1] of strip s (0 = highest columns); c0 > n: the strip is empty
__host__ __device__ inline void strip_bounds(int n, int S, int s, int& c0, int& c1) {
  const int t = S - 1 - s, wl = strip_last_width(n, S), wu = strip_upper_width(n, S);
  if (t == 0) { c0 = 1; c1 = wl < n ? wl : n; return; }
  c0 = wl + (t - 1) * wu + 1;
  c1 = c0 + wu - 1 < n ? c0 + wu - 1 : n;
}

struct StripRec {              // MFE strips: exchange records and list counts live in a buffer of their own
  int32_t* rec = nullptr;
  long long stride = 0;        // int32 per sequence
};
// wait until the strip's flag shows diagonal `target` (or DONE / FAIL); one wave, every lane returns the same value
__device__ __forceinline__ bool strip_wait(const int* flag, int base, int d, int& seen) {
  const int target = base + d;
  SpinClock clk;
  for (;;) {
    const int v = __builtin_amdgcn_readfirstlane(ld_agent(flag));
    seen = v;
    if (flag_ge(v, target)) return v != base + STRIP_FAIL;
    if (clk.expired()) return false;
    spin_pause();
  }
}

#define STRIP_BARRIER() __syncthreads()      // (drains the wave's stores too: the hand-over relies on it)
template <typename T> __device__ __forceinline__ void strip_store(T* p, T v) { st_agent(p, v); }
// 8-byte buffer load that bypasses this CU's L1 (sc1): for cells another workgroup stored
template <typename RS>
__device__ __forceinline__ double buf_load_f64_aux(RS rsrc, int voff, int soff) {
  const auto v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, 16);
  return __hiloint2double((int)v[1], (int)v[0]);
}
template <typename RS>
__device__ __forceinline__ f64x2 buf_load_f64x2_sc1(RS rsrc, int voff, int soff) {
  const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 16);     // aux 16 = sc1: bypasses this CU's L1
  return f64x2{__hiloint2double((int)v[1], (int)v[0]), __hiloint2double((int)v[3], (int)v[2])};
}

}  // namespace drna
