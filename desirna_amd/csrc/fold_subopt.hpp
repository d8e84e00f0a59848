// fold_subopt.hpp -- energy of the second-best secondary structure for one sequence per workgroup on gfx950.
// Replaces get_first_suboptimal_structure_and_energy(seq, fc, 1)[1] of the reference's negative-design option
// (-nd on; utils/energy_scores.py:105-107, :453-488): ViennaRNA's subopt enumeration (uniq_ML = 1) with a growing energy
// band until it holds two structures, sorted by energy, second entry taken -- SURVEY 8(f)-4.  Only that entry's ENERGY is
// used by the caller: the lowest energy over all structures other than one ground-state structure (0 if none lies
// within 49 kcal/mol of the MFE).
//
// Two-best dynamic programme over an unambiguous decomposition (every structure has one derivation, so the two smallest
// values of a table entry belong to two different structures):
//   F[j]    = { F[j-1] ; F[i-1] + C[i,j] + ext(i,j) }
//   C[i,j]  = { hairpin ; C[p,q] + interior ; M2[i+1,j-1] + closing }
//   M[i,j]  (>= 1 stem) = { M[i,j-1] + b ; (k-i) b + C[k,j] + stem ; M[i,k-1] + C[k,j] + stem }
//   M2[i,j] (>= 2 stems) = { M2[i,j-1] + b ; M[i,k-1] + C[k,j] + stem }
// One wave per cell: the lanes share the interior-loop shapes and the positions k, the (best, second) pairs are folded
// with a butterfly over disjoint lane groups.  Tables (pairs of int32, diagonal-major) live in HBM/L2.
#pragma once
#include "fold_mfe.hpp"

namespace drna {

struct SubArgs {
  const MfeTables* T = nullptr;
  const Plan* plan = nullptr;
  const int* hp_len = nullptr;
  const char* seqs = nullptr;     // R x L ASCII
  int L = 0, ld = 0;
  int32_t* ws = nullptr;          // per sequence: C, M, M2 as (best, second) int32 pairs: 6 ld*ld int32
  long long ws_stride = 0;
  int32_t* E2 = nullptr;          // R: the reference's subopt energy (dcal/mol; 0 = none within 4900)
  int32_t* E12 = nullptr;         // optional R x 2: the two lowest energies (second = INF_REF if there is one structure only)
  int32_t* status = nullptr;      // R
};

struct Top2 { int a, b; };
__device__ __forceinline__ void t2_add(Top2& t, int v) {
  if (v >= INF_DEV / 2) return;
  if (v < t.a) { t.b = t.a; t.a = v; }
  else if (v < t.b) t.b = v;
}
__device__ __forceinline__ void t2_add_sum(Top2& t, Top2 x, int e) {
  if (x.a < INF_DEV / 2) t2_add(t, x.a + e);
  if (x.b < INF_DEV / 2) t2_add(t, x.b + e);
}
__device__ __forceinline__ void t2_add_sum2(Top2& t, Top2 x, Top2 y, int e) {
  if (x.a >= INF_DEV / 2 || y.a >= INF_DEV / 2) return;
  t2_add(t, x.a + y.a + e);
  if (y.b < INF_DEV / 2) t2_add(t, x.a + y.b + e);
  if (x.b < INF_DEV / 2) t2_add(t, x.b + y.a + e);
}
// every lane ends with the two smallest values of the wave (the lane groups merged at each step are disjoint)
__device__ __forceinline__ Top2 wave_top2(Top2 t) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int oa = __shfl_xor(t.a, o), ob = __shfl_xor(t.b, o);
    const int lo = min(t.a, oa), hi = max(t.a, oa);
    t.b = min(hi, min(t.b, ob));
    t.a = lo;
  }
  return t;
}

struct SubSmem : MfeSmemCore<MAXN> {
  Top2 F[MAXN + 2];
};

template <int NT>
__global__ __launch_bounds__(NT) void subopt_kernel(SubArgs A) {
  __shared__ SubSmem sm;
  const MfeTables& T = *A.T;
  const Plan& P = *A.plan;
  const int r = blockIdx.x;
  const int n = A.L, ld = A.ld;
  const int tid = threadIdx.x, lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(wave_id());
  const int INF = INF_DEV, HALF = INF_DEV / 2;
  int32_t* base = A.ws + (long long)r * A.ws_stride;
  const long long tab = (long long)ld * ld;
  Top2* C = reinterpret_cast<Top2*>(base);
  Top2* M = reinterpret_cast<Top2*>(base + 2 * tab);
  Top2* M2 = reinterpret_cast<Top2*>(base + 4 * tab);

  for (int k = tid; k < 64; k += NT) sm.stack[k] = T.stack[k];
  for (int k = tid; k < 128; k += NT) {
    sm.mmH[k] = T.mmH[k]; sm.mmI[k] = T.mmI[k]; sm.mm1n[k] = T.mm1n[k];
    sm.mm23[k] = T.mm23[k]; sm.mmM[k] = T.mmM[k]; sm.mmExt[k] = T.mmExt[k];
  }
  for (int k = tid; k < 1024; k += NT) sm.int11[k] = T.int11[k];
  for (int k = tid; k < 32; k += NT) { sm.d5[k] = T.d5[k]; sm.d3[k] = T.d3[k]; }
  if (tid == 0) sm.flag = 0;
  __syncthreads();
  const char* seq = A.seqs + (long long)r * n;
  for (int k = tid; k < n; k += NT) {
    const int c = enc_nt(seq[k]);
    if (c < 0) sm.flag = 1;
    sm.S[k + 1] = (unsigned char)(c < 0 ? 0 : c);
  }
  // diagonals 0 .. TURN: no pair, no multiloop content
  for (int d = 0; d <= TURN && d < n; d++)
    for (int k = tid; k < ld; k += NT) { C[d * ld + k] = Top2{INF, INF}; M[d * ld + k] = Top2{INF, INF}; M2[d * ld + k] = Top2{INF, INF}; }
  __syncthreads();
  if (tid == 0) { sm.S[0] = sm.S[n]; sm.S[n + 1] = sm.S[1]; }
  __syncthreads();
  if (sm.flag) {
    if (tid == 0) { A.status[r] = ST_BAD_CHAR; A.E2[r] = 0; if (A.E12) { A.E12[2 * r] = 0; A.E12[2 * r + 1] = INF_REF; } }
    return;
  }

  for (int d = TURN + 1; d < n; d++) {
    const int ncell = n - d;
    for (int i = wave + 1; i <= ncell; i += NT / WAVE) {
      const int j = i + d;
      const int t = pair_type(sm.S[i], sm.S[j]);
      const int tau = t > 2 ? T.TermAU : 0;
      Top2 c{INF, INF};
      if (t) {
        const int si1 = sm.S[i + 1], sj1 = sm.S[j - 1];
        for (int e = lane; e < NPLAN; e += WAVE) {
          const int u1 = P.u1[e], u2 = P.u2[e];
          const int dp = d - 2 - u1 - u2;
          if (dp <= TURN) continue;
          const int p = i + 1 + u1, q = j - 1 - u2;
          const int t2 = pair_type(sm.S[p], sm.S[q]);
          if (!t2) continue;
          const Top2 cp = C[dp * ld + p];
          if (cp.a >= HALF) continue;
          const int info = (rtype_of(t2) << 4) | (sm.S[q + 1] << 2) | sm.S[p - 1];
          t2_add_sum(c, cp, mfe_intloop(sm, T, u1, u2, t, si1, sj1, info));
        }
        if (lane == 0) {
          t2_add(c, mfe_hairpin_e(sm, T, A.hp_len[d - 1], i, j, t));
          t2_add_sum(c, M2[(d - 2) * ld + i + 1], T.MLclosing + T.MLintern + tau + sm.mmM[rtype_of(t) * 16 + sj1 * 4 + si1]);
        }
        c = wave_top2(c);
      }
      Top2 m{INF, INF}, m2{INF, INF};
      if (lane == 0) {
        t2_add_sum(m, M[(d - 1) * ld + i], T.MLbase);
        t2_add_sum(m2, M2[(d - 1) * ld + i], T.MLbase);
      }
      for (int k = i + lane; k <= j - TURN - 1; k += WAVE) {
        const int tk = pair_type(sm.S[k], sm.S[j]);
        if (!tk) continue;
        const Top2 ck = k == i ? c : C[(j - k) * ld + k];
        if (ck.a >= HALF) continue;
        const int st = T.MLintern + (tk > 2 ? T.TermAU : 0) + sm.mmM[tk * 16 + sm.S[k - 1] * 4 + sm.S[j + 1]];
        t2_add_sum(m, ck, (k - i) * T.MLbase + st);
        if (k > i) {
          const Top2 mk = M[(k - 1 - i) * ld + i];
          t2_add_sum2(m, mk, ck, st);
          t2_add_sum2(m2, mk, ck, st);
        }
      }
      m = wave_top2(m);
      m2 = wave_top2(m2);
      if (lane == 0) { C[d * ld + i] = c; M[d * ld + i] = m; M2[d * ld + i] = m2; }
    }
    __syncthreads();
  }

  if (wave != 0) return;
  sm.F[0] = Top2{0, INF};
  for (int j = 1; j <= n; j++) {
    Top2 f{INF, INF};
    if (lane == 0) t2_add_sum(f, sm.F[j - 1], 0);
    for (int i = lane + 1; i <= j - TURN - 1; i += WAVE) {
      const int t = pair_type(sm.S[i], sm.S[j]);
      if (!t) continue;
      const Top2 cij = C[(j - i) * ld + i];
      if (cij.a >= HALF) continue;
      t2_add_sum2(f, sm.F[i - 1], cij, (t > 2 ? T.TermAU : 0) + mfe_extstem(sm, t, i, j, n));
    }
    f = wave_top2(f);
    sm.F[j] = f;                                   // every lane stores the same value
  }
  if (lane == 0) {
    const Top2 f = sm.F[n];
    A.status[r] = ST_OK;
    A.E2[r] = (f.b >= HALF || f.b - f.a > 4900) ? 0 : f.b;
    if (A.E12) { A.E12[2 * r] = f.a; A.E12[2 * r + 1] = f.b >= HALF ? INF_REF : f.b; }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// K lowest-energy structures (energies AND dot-bracket strings) of one sequence per workgroup.
// Replaces get_first_suboptimal_structure_and_energy(seq, fc, k)[0] for k = 1 .. #alt structures, the call behind
// get_alt_mcc() in the reference's final ranking of alternative-structure designs (utils/sequence_utils.py:766-793,
// utils/energy_scores.py:453-488): entry k of ViennaRNA's energy-sorted subopt list (uniq_ML = 1).  Same unambiguous
// decomposition as above with K-best lists per table entry, then one traceback per rank: a table entry's r-th value is
// expanded by re-enumerating the entry's candidates in a fixed order and taking, among those that reproduce the value,
// the one whose index equals the number of equal values ranked before r.  Different ranks of one entry thus expand to
// different derivations, i.e. different structures.  The order among structures of EQUAL energy is this enumeration
// order, not ViennaRNA's (which the reference pins nowhere).

template <int K>
struct TopK { int v[K]; };

template <int K>
__device__ __forceinline__ void tk_init(TopK<K>& t) {
#pragma unroll
  for (int r = 0; r < K; r++) t.v[r] = INF_DEV;
}
template <int K>
__device__ __forceinline__ void tk_add(TopK<K>& t, int v) {
  if (v >= INF_DEV / 2) return;
#pragma unroll
  for (int r = 0; r < K; r++) { const int lo = min(v, t.v[r]); v = max(v, t.v[r]); t.v[r] = lo; }
}
template <int K>
__device__ __forceinline__ void tk_add_sum(TopK<K>& t, const TopK<K>& x, int e) {
#pragma unroll
  for (int r = 0; r < K; r++) if (x.v[r] < INF_DEV / 2) tk_add(t, x.v[r] + e);
}
template <int K>
__device__ __forceinline__ void tk_add_sum2(TopK<K>& t, const TopK<K>& x, const TopK<K>& y, int e) {
#pragma unroll
  for (int a = 0; a < K; a++)
#pragma unroll
    for (int b = 0; a + b < K; b++)       // the r-th best sum never needs ranks with a + b > r
      if (x.v[a] < INF_DEV / 2 && y.v[b] < INF_DEV / 2) tk_add(t, x.v[a] + y.v[b] + e);
}
template <int K>
__device__ __forceinline__ TopK<K> wave_topk(TopK<K> t) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    TopK<K> x;
#pragma unroll
    for (int r = 0; r < K; r++) x.v[r] = __shfl_xor(t.v[r], o);
#pragma unroll
    for (int r = 0; r < K; r++) tk_add(t, x.v[r]);
  }
  return t;
}

struct KbArgs {
  const MfeTables* T = nullptr;
  const Plan* plan = nullptr;
  const int* hp_len = nullptr;
  const char* seqs = nullptr;     // R x L ASCII
  int L = 0, ld = 0;
  int32_t* ws = nullptr;          // per sequence: C, M, M2 as K-lists: 3 K ld*ld int32
  long long ws_stride = 0;
  int32_t* E = nullptr;           // R x K energies, ascending (INF_REF where the sequence has fewer structures)
  char* ss = nullptr;             // R x K x L dot-bracket strings (all dots where E = INF_REF)
  int32_t* status = nullptr;      // R
};

template <int K>
struct KbSmem : MfeSmemCore<MAXN> {
  char db[K][MAXN + 2];
};

enum { KB_F = 1, KB_C = 2, KB_M = 3, KB_M2 = 4 };
__device__ __forceinline__ int kb_pack(int kind, int i, int j, int r) { return i | (j << 12) | (kind << 24) | (r << 27); }

template <int K>
struct KbCtx {
  const KbSmem<K>* sm;
  const MfeTables* T;
  const Plan* P;
  const int* hp_len;
  const TopK<K>*C, *M, *M2, *F;
  int n, ld;
};

// candidates of element e of a table entry whose value is v; returns how many of them reproduce v and, when sel >= 0,
// leaves the children of the sel-th such candidate in (ca, cb) (0 = no child)
template <int K>
__device__ int kb_enum(const KbCtx<K>& X, int kind, int i, int j, int v, int e, int sel, int& ca, int& cb) {
  const KbSmem<K>& sm = *X.sm;
  const MfeTables& T = *X.T;
  const int ld = X.ld, n = X.n, HALF = INF_DEV / 2;
  int cnt = 0;
#define KB_HIT(A_, B_) do { if (cnt == sel) { ca = (A_); cb = (B_); } cnt++; } while (0)
  if (kind == KB_F) {
    if (e == 0) {
      const TopK<K> f = X.F[j - 1];
      for (int a = 0; a < K; a++) if (f.v[a] < HALF && f.v[a] == v) KB_HIT(kb_pack(KB_F, 0, j - 1, a), 0);
    } else {
      const int p = e;
      const int t = pair_type(sm.S[p], sm.S[j]);
      if (t) {
        const TopK<K> c = X.C[(j - p) * ld + p], f = X.F[p - 1];
        const int x = (t > 2 ? T.TermAU : 0) + mfe_extstem(sm, t, p, j, n);
        for (int a = 0; a < K; a++)
          for (int b = 0; b < K; b++)
            if (f.v[a] < HALF && c.v[b] < HALF && f.v[a] + c.v[b] + x == v) KB_HIT(kb_pack(KB_F, 0, p - 1, a), kb_pack(KB_C, p, j, b));
      }
    }
  } else if (kind == KB_C) {
    const int d = j - i;
    const int t = pair_type(sm.S[i], sm.S[j]);
    const int si1 = sm.S[i + 1], sj1 = sm.S[j - 1];
    if (e == 0) {
      if (mfe_hairpin_e(sm, T, X.hp_len[d - 1], i, j, t) == v) KB_HIT(0, 0);
      const TopK<K> m2 = X.M2[(d - 2) * ld + i + 1];
      const int x = T.MLclosing + T.MLintern + (t > 2 ? T.TermAU : 0) + sm.mmM[rtype_of(t) * 16 + sj1 * 4 + si1];
      for (int a = 0; a < K; a++) if (m2.v[a] < HALF && m2.v[a] + x == v) KB_HIT(kb_pack(KB_M2, i + 1, j - 1, a), 0);
    } else {
      const int u1 = X.P->u1[e - 1], u2 = X.P->u2[e - 1];
      const int dp = d - 2 - u1 - u2;
      if (dp > TURN) {
        const int p = i + 1 + u1, q = j - 1 - u2;
        const int t2 = pair_type(sm.S[p], sm.S[q]);
        if (t2) {
          const TopK<K> c = X.C[dp * ld + p];
          const int info = (rtype_of(t2) << 4) | (sm.S[q + 1] << 2) | sm.S[p - 1];
          const int x = mfe_intloop(sm, T, u1, u2, t, si1, sj1, info);
          for (int a = 0; a < K; a++) if (c.v[a] < HALF && c.v[a] + x == v) KB_HIT(kb_pack(KB_C, p, q, a), 0);
        }
      }
    }
  } else {
    const int d = j - i;
    const TopK<K>* own = kind == KB_M ? X.M : X.M2;
    if (e == 0) {
      const TopK<K> m = own[(d - 1) * ld + i];
      for (int a = 0; a < K; a++) if (m.v[a] < HALF && m.v[a] + T.MLbase == v) KB_HIT(kb_pack(kind, i, j - 1, a), 0);
    } else {
      const int k = i + e - 1;
      const int tk = pair_type(sm.S[k], sm.S[j]);
      if (tk) {
        const TopK<K> c = X.C[(j - k) * ld + k];
        const int st = T.MLintern + (tk > 2 ? T.TermAU : 0) + sm.mmM[tk * 16 + sm.S[k - 1] * 4 + sm.S[j + 1]];
        if (kind == KB_M)
          for (int a = 0; a < K; a++) if (c.v[a] < HALF && c.v[a] + (k - i) * T.MLbase + st == v) KB_HIT(kb_pack(KB_C, k, j, a), 0);
        if (k > i) {
          const TopK<K> m = X.M[(k - 1 - i) * ld + i];
          for (int a = 0; a < K; a++)
            for (int b = 0; b < K; b++)
              if (m.v[a] < HALF && c.v[b] < HALF && m.v[a] + c.v[b] + st == v) KB_HIT(kb_pack(KB_M, i, k - 1, a), kb_pack(KB_C, k, j, b));
        }
      }
    }
  }
#undef KB_HIT
  return cnt;
}

template <int NT, int K>
__global__ __launch_bounds__(NT) void kbest_kernel(KbArgs A) {
  __shared__ KbSmem<K> sm;
  const MfeTables& T = *A.T;
  const Plan& P = *A.plan;
  const int r = blockIdx.x;
  const int n = A.L, ld = A.ld;
  const int tid = threadIdx.x, lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(wave_id());
  const int INF = INF_DEV, HALF = INF_DEV / 2;
  int32_t* base = A.ws + (long long)r * A.ws_stride;
  const long long tab = (long long)ld * ld * K;
  TopK<K>* C = reinterpret_cast<TopK<K>*>(base);
  TopK<K>* M = reinterpret_cast<TopK<K>*>(base + tab);
  TopK<K>* M2 = reinterpret_cast<TopK<K>*>(base + 2 * tab);
  TopK<K>* F = C;                                       // rows 0 .. TURN of C are never read: row 0 holds F[0 .. n]
  int32_t* stacks = reinterpret_cast<int32_t*>(M2);     // rows 0, 1 of M2 are never read: one traceback stack of ld ints per rank

  for (int k = tid; k < 64; k += NT) sm.stack[k] = T.stack[k];
  for (int k = tid; k < 128; k += NT) {
    sm.mmH[k] = T.mmH[k]; sm.mmI[k] = T.mmI[k]; sm.mm1n[k] = T.mm1n[k];
    sm.mm23[k] = T.mm23[k]; sm.mmM[k] = T.mmM[k]; sm.mmExt[k] = T.mmExt[k];
  }
  for (int k = tid; k < 1024; k += NT) sm.int11[k] = T.int11[k];
  for (int k = tid; k < 32; k += NT) { sm.d5[k] = T.d5[k]; sm.d3[k] = T.d3[k]; }
  if (tid == 0) sm.flag = 0;
  __syncthreads();
  const char* seq = A.seqs + (long long)r * n;
  for (int k = tid; k < n; k += NT) {
    const int c = enc_nt(seq[k]);
    if (c < 0) sm.flag = 1;
    sm.S[k + 1] = (unsigned char)(c < 0 ? 0 : c);
  }
  TopK<K> none;
  tk_init(none);
  for (int d = 0; d <= TURN && d < n; d++)
    for (int k = tid; k < ld; k += NT) { if (d) C[d * ld + k] = none; M[d * ld + k] = none; if (d > 1) M2[d * ld + k] = none; }
  for (int x = tid; x < K * n; x += NT) A.ss[(long long)r * K * n + x] = '.';
  __syncthreads();
  if (tid == 0) { sm.S[0] = sm.S[n]; sm.S[n + 1] = sm.S[1]; }
  __syncthreads();
  if (sm.flag) {
    if (tid == 0) { A.status[r] = ST_BAD_CHAR; for (int k = 0; k < K; k++) A.E[r * K + k] = INF_REF; }
    return;
  }

  for (int d = TURN + 1; d < n; d++) {
    const int ncell = n - d;
    for (int i = wave + 1; i <= ncell; i += NT / WAVE) {
      const int j = i + d;
      const int t = pair_type(sm.S[i], sm.S[j]);
      const int tau = t > 2 ? T.TermAU : 0;
      TopK<K> c = none;
      if (t) {
        const int si1 = sm.S[i + 1], sj1 = sm.S[j - 1];
        for (int e = lane; e < NPLAN; e += WAVE) {
          const int u1 = P.u1[e], u2 = P.u2[e];
          const int dp = d - 2 - u1 - u2;
          if (dp <= TURN) continue;
          const int p = i + 1 + u1, q = j - 1 - u2;
          const int t2 = pair_type(sm.S[p], sm.S[q]);
          if (!t2) continue;
          const TopK<K> cp = C[dp * ld + p];
          if (cp.v[0] >= HALF) continue;
          const int info = (rtype_of(t2) << 4) | (sm.S[q + 1] << 2) | sm.S[p - 1];
          tk_add_sum(c, cp, mfe_intloop(sm, T, u1, u2, t, si1, sj1, info));
        }
        if (lane == 0) {
          tk_add(c, mfe_hairpin_e(sm, T, A.hp_len[d - 1], i, j, t));
          tk_add_sum(c, M2[(d - 2) * ld + i + 1], T.MLclosing + T.MLintern + tau + sm.mmM[rtype_of(t) * 16 + sj1 * 4 + si1]);
        }
        c = wave_topk(c);
      }
      TopK<K> m = none, m2 = none;
      if (lane == 0) {
        tk_add_sum(m, M[(d - 1) * ld + i], T.MLbase);
        tk_add_sum(m2, M2[(d - 1) * ld + i], T.MLbase);
      }
      for (int k = i + lane; k <= j - TURN - 1; k += WAVE) {
        const int tk = pair_type(sm.S[k], sm.S[j]);
        if (!tk) continue;
        const TopK<K> ck = k == i ? c : C[(j - k) * ld + k];
        if (ck.v[0] >= HALF) continue;
        const int st = T.MLintern + (tk > 2 ? T.TermAU : 0) + sm.mmM[tk * 16 + sm.S[k - 1] * 4 + sm.S[j + 1]];
        tk_add_sum(m, ck, (k - i) * T.MLbase + st);
        if (k > i) {
          const TopK<K> mk = M[(k - 1 - i) * ld + i];
          tk_add_sum2(m, mk, ck, st);
          tk_add_sum2(m2, mk, ck, st);
        }
      }
      m = wave_topk(m);
      m2 = wave_topk(m2);
      if (lane == 0) { C[d * ld + i] = c; M[d * ld + i] = m; M2[d * ld + i] = m2; }
    }
    __syncthreads();
  }

  if (wave == 0) {
    TopK<K> f0 = none;
    f0.v[0] = 0;
    F[0] = f0;                                       // every lane stores the same value (here and below)
    for (int j = 1; j <= n; j++) {
      TopK<K> f = none;
      if (lane == 0) tk_add_sum(f, F[j - 1], 0);
      for (int i = lane + 1; i <= j - TURN - 1; i += WAVE) {
        const int t = pair_type(sm.S[i], sm.S[j]);
        if (!t) continue;
        const TopK<K> cij = C[(j - i) * ld + i];
        if (cij.v[0] >= HALF) continue;
        tk_add_sum2(f, F[i - 1], cij, (t > 2 ? T.TermAU : 0) + mfe_extstem(sm, t, i, j, n));
      }
      f = wave_topk(f);
      F[j] = f;
    }
    if (lane == 0) {
      A.status[r] = ST_OK;
      for (int k = 0; k < K; k++) A.E[r * K + k] = F[n].v[k] >= HALF ? INF_REF : F[n].v[k];
    }
  }
  __syncthreads();

  // ---- one traceback per rank; a wave works on one rank at a time, every lane holds the same state
  KbCtx<K> X;
  X.sm = &sm; X.T = &T; X.P = &P; X.hp_len = A.hp_len; X.C = C; X.M = M; X.M2 = M2; X.F = F; X.n = n; X.ld = ld;
  for (int rank = wave; rank < K; rank += NT / WAVE) {
    if (F[n].v[rank] >= HALF) continue;
    char* db = sm.db[rank];
    for (int x = lane; x <= n; x += WAVE) db[x] = '.';
    (void)__ballot(true);                            // the dots are in place before any lane writes a bracket
    int32_t* stk = stacks + (long long)rank * ld;
    int sp = 0;
    bool bad = false;
    stk[sp++] = kb_pack(KB_F, 0, n, rank);
    while (sp > 0 && !bad) {
      const int it = stk[--sp];
      const int i = it & 4095, j = (it >> 12) & 4095, kind = (it >> 24) & 7, rk = it >> 27;
      if (kind == KB_F && j == 0) continue;
      const TopK<K>* tabp = kind == KB_F ? F + j : kind == KB_C ? C + (j - i) * ld + i : kind == KB_M ? M + (j - i) * ld + i : M2 + (j - i) * ld + i;
      const int v = tabp->v[rk];
      int m = 0;
      for (int a = 0; a < rk; a++) m += tabp->v[a] == v;
      if (kind == KB_C) { db[i] = '('; db[j] = ')'; }
      const int nel = kind == KB_F ? 1 + max(j - TURN - 1, 0) : kind == KB_C ? 1 + NPLAN : 1 + max(j - TURN - i, 0);
      bool found = false;
      int ca = 0, cb = 0;
      for (int b0 = 0; b0 < nel && !found; b0 += WAVE) {
        const int e = b0 + lane;
        int da = 0, dbb = 0;
        const int cnt = e < nel ? kb_enum<K>(X, kind, i, j, v, e, -1, da, dbb) : 0;
        int pre = cnt;                               // inclusive prefix over the lanes (= over the elements, in order)
        for (int o = 1; o < WAVE; o <<= 1) {
          const int x = __shfl(pre, lane >= o ? lane - o : lane);
          if (lane >= o) pre += x;
        }
        const int total = __shfl(pre, WAVE - 1);
        if (m < total) {
          const unsigned long long mask = __ballot(pre > m);
          const int win = __ffsll((long long)mask) - 1;
          const int sel = m - (__shfl(pre, win) - __shfl(cnt, win));
          if (lane == win) kb_enum<K>(X, kind, i, j, v, e, sel, da, dbb);
          ca = __shfl(da, win); cb = __shfl(dbb, win);
          found = true;
        } else m -= total;
      }
      if (!found) { bad = true; break; }
      if (cb) stk[sp++] = cb;
      if (ca) stk[sp++] = ca;
    }
    (void)__ballot(true);
    if (bad) { if (lane == 0) A.status[r] = ST_TRACEBACK; continue; }
    for (int x = lane; x < n; x += WAVE) A.ss[((long long)r * K + rank) * n + x] = db[x + 1];
  }
}

}  // namespace drna
