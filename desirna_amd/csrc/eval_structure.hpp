// eval_structure.hpp -- loop-energy sum of given structures for a batch of sequences on gfx950.
// Replaces fc.eval_structure(target) and the alt-structure evaluations:
// reference utils/energy_scores.py:75,99.  Model: SURVEY.md App. A.3 / A.6; only '(' ')'
// count as pairs (pk brackets are ignored exactly as ViennaRNA ignores them), non-canonical pairs
// get type 7, loops longer than MAXLOOP use the logarithmic extrapolation.
//
// One wave per (sequence, structure).  The pair table of each structure is parsed once on the host
// (it is the same for every replica); lanes take the loops (one closing pair each), walk their
// backbone in LDS and the wave sums the loop energies.
#pragma once
#include "fold_common.hpp"

namespace drna {

struct EvalArgs {
  const MfeTables* T;
  const int* hp_len;            // by loop size
  const int* bulge_len;         // by loop size (log extrapolation beyond 30)
  const int* int_len;           // by loop size
  const char* seqs;             // R x L ASCII
  const short* pt;              // n_targets x (L+2): pt[i] = partner (1-based) or 0
  int L;
  int n_targets;
  int32_t* Ed;                  // R x n_targets (dcal/mol)
  // ragged batch (one structure per sequence): lengths / offsets of the sequences, structure of each sequence and the
  // offset of its pair table in pt (pair tables concatenated, L_t + 2 shorts each); grid = R, Ed = R values
  Ragged rg;
  const int* target_of = nullptr;
  const int* pt_off = nullptr;
  // two strands (concatenated, no '&'): cut = length of the first one (0 = one strand); a loop whose backbone contains
  // the nick is scored like an exterior loop, DuplexInit once if any pair joins the strands
  int cut = 0;
  int DuplexInit = 0;
};

struct EvalSmem {
  unsigned char S[MAXN + 4];
  short pt[MAXN + 4];
};

__device__ __forceinline__ int eval_ptype(const EvalSmem& sm, int i, int j) {
  const int t = pair_type(sm.S[i], sm.S[j]);
  return t ? t : 7;
}

__device__ __forceinline__ int eval_mm(const int* tab, int t, int a, int b) { return tab[t * 16 + a * 4 + b]; }

// E_ExtLoop with explicit neighbour flags
__device__ __forceinline__ int eval_ext(const MfeTables& T, int t, bool h5, int s5, bool h3, int s3) {
  int x;
  if (h5 && h3) x = eval_mm(T.mmExt, t, s5, s3);
  else if (h5) x = T.d5[t * 4 + s5];
  else if (h3) x = T.d3[t * 4 + s3];
  else x = 0;
  return x + (t > 2 ? T.TermAU : 0);
}

__device__ inline int eval_loop(const EvalSmem& sm, const EvalArgs& A, int i, int j) {
  const MfeTables& T = *A.T;
  const int t = eval_ptype(sm, i, j);
  const int cut = A.cut;
  if (cut > 0 && i <= cut && cut < j) {
    // does the nick lie on THIS loop's backbone (not inside one of its stems)?
    bool nick = true;
    for (int p = i + 1; p < j;) {
      const int q = sm.pt[p];
      if (q > p) { if (p <= cut && cut < q) { nick = false; break; } p = q + 1; } else p++;
    }
    if (nick) {
      int e = eval_ext(T, eval_ptype(sm, j, i), j - 1 > cut || j <= cut, sm.S[j - 1], !(i <= cut && i + 1 > cut), sm.S[i + 1]);
      for (int p = i + 1; p < j;) {
        const int q = sm.pt[p];
        if (q > p) {
          e += eval_ext(T, eval_ptype(sm, p, q), !(p - 1 <= cut && p > cut), sm.S[p - 1], !(q <= cut && q + 1 > cut), sm.S[q + 1]);
          p = q + 1;
        } else p++;
      }
      return e;
    }
  }
  // walk the backbone: count stems and unpaired bases, remember the first stem
  int nst = 0, unp = 0, p1 = 0, q1 = 0;
  int e_stems = 0;
  for (int p = i + 1; p < j;) {
    const int q = sm.pt[p];
    if (q > p) {
      if (!nst) { p1 = p; q1 = q; }
      nst++;
      const int t2 = eval_ptype(sm, p, q);
      e_stems += T.MLintern + (t2 > 2 ? T.TermAU : 0) + eval_mm(T.mmM, t2, sm.S[p - 1], sm.S[q + 1]);
      p = q + 1;
    } else { unp++; p++; }
  }
  const int si1 = sm.S[i + 1], sj1 = sm.S[j - 1];
  if (nst == 0) {
    const int u = j - i - 1;
    const int e = A.hp_len[u];
    if (u < 3) return e;
    if (u == 3) {
      int code = 0;
      for (int k = 0; k < 5; k++) code |= sm.S[i + k] << (2 * k);
      for (int k = 0; k < T.n_tri; k++)
        if (T.tri_code[k] == code) return T.tri_e[k];
      return e + (t > 2 ? T.TermAU : 0);
    }
    if (u == 4) {
      int code = 0;
      for (int k = 0; k < 6; k++) code |= sm.S[i + k] << (2 * k);
      for (int k = 0; k < T.n_tetra; k++)
        if (T.tetra_code[k] == code) return T.tetra_e[k];
    } else if (u == 6) {
      int code = 0;
      for (int k = 0; k < 8; k++) code |= sm.S[i + k] << (2 * k);
      for (int k = 0; k < T.n_hexa; k++)
        if (T.hexa_code[k] == code) return T.hexa_e[k];
    }
    return e + eval_mm(T.mmH, t, si1, sj1);
  }
  if (nst == 1) {
    const int u1 = p1 - i - 1, u2 = j - q1 - 1;
    const int t2 = eval_ptype(sm, q1, p1);   // rtype of the inner pair
    const int sp1 = sm.S[p1 - 1], sq1 = sm.S[q1 + 1];
    const int nl = u1 > u2 ? u1 : u2, ns = u1 > u2 ? u2 : u1;
    if (nl == 0) return T.stack[t * 8 + t2];
    if (ns == 0) {
      int e = A.bulge_len[nl];
      if (nl == 1) return e + T.stack[t * 8 + t2];
      return e + (t > 2 ? T.TermAU : 0) + (t2 > 2 ? T.TermAU : 0);
    }
    if (ns == 1) {
      if (nl == 1) return T.int11[(t * 8 + t2) * 16 + si1 * 4 + sj1];
      if (nl == 2)
        return (u1 == 1) ? T.int21[(t * 8 + t2) * 64 + si1 * 16 + sq1 * 4 + sj1]
                         : T.int21[(t2 * 8 + t) * 64 + sq1 * 16 + si1 * 4 + sp1];
      return A.int_len[nl + 1] + min(T.max_ninio, (nl - ns) * T.ninio) + eval_mm(T.mm1n, t, si1, sj1) +
             eval_mm(T.mm1n, t2, sq1, sp1);
    }
    if (ns == 2) {
      if (nl == 2) return T.int22[(t * 8 + t2) * 256 + si1 * 64 + sp1 * 16 + sq1 * 4 + sj1];
      if (nl == 3) return T.interior[5] + T.ninio + eval_mm(T.mm23, t, si1, sj1) + eval_mm(T.mm23, t2, sq1, sp1);
    }
    return A.int_len[nl + ns] + min(T.max_ninio, (nl - ns) * T.ninio) + eval_mm(T.mmI, t, si1, sj1) +
           eval_mm(T.mmI, t2, sq1, sp1);
  }
  // multiloop: closing pair seen from inside is (j,i)
  const int tc = eval_ptype(sm, j, i);
  return T.MLclosing + T.MLintern + (tc > 2 ? T.TermAU : 0) + eval_mm(T.mmM, tc, sj1, si1) + unp * T.MLbase + e_stems;
}

// one (sequence, structure) evaluation by one wave; `sm` is that wave's own (the wave synchronises its LDS traffic itself, so
// this also runs inside a larger workgroup: the helper workgroups of pf_lds_kernel evaluate their sequence while they wait for
// the first rows of the fill)
__device__ __forceinline__ void eval_one(EvalSmem& sm, EvalArgs A, int r, int k, int lane) {
  const MfeTables& T = *A.T;
  const bool ragged = A.rg.len != nullptr;
  if (ragged) A.L = A.rg.len[r];
  const int n = A.L;
  const char* seq = A.seqs + (ragged ? (long long)A.rg.off[r] : (long long)r * n);
  const short* pt = ragged ? A.pt + A.pt_off[A.target_of[r]] : A.pt + (long long)k * (n + 2);
  bool bad = false;
  for (int x = lane; x < n; x += WAVE) {
    const int c = enc_nt(seq[x]);
    bad |= c < 0;
    sm.S[x + 1] = (unsigned char)(c < 0 ? 0 : c);
  }
  for (int x = lane; x < n + 2; x += WAVE) sm.pt[x] = pt[x];
  wave_lds_sync();
  if (lane == 0) { sm.S[0] = sm.S[n]; sm.S[n + 1] = sm.S[1]; }
  wave_lds_sync();
  int e = 0;
  for (int i = lane + 1; i <= n; i += WAVE)
    if (sm.pt[i] > i) e += eval_loop(sm, A, i, sm.pt[i]);
  if (lane == 0) {
    // exterior loop stems (energy_of_extLoop_pt, dangles = 2)
    for (int i = 1; i <= n;) {
      const int j = sm.pt[i];
      if (j > i) {
        const int t = eval_ptype(sm, i, j);
        const int cut = A.cut;
        e += eval_ext(T, t, i > 1 && !(cut > 0 && i - 1 <= cut && i > cut), sm.S[i - 1],
                      j < n && !(cut > 0 && j <= cut && j + 1 > cut), sm.S[j + 1]);
        if (cut > 0 && i <= cut && j > cut) e += A.DuplexInit;       // an exterior stem that joins the strands: at most one
        i = j + 1;
      } else i++;
    }
  }
  e = wave_sum_i32(e);
  const unsigned long long anybad = __ballot(bad);
  if (lane == 0) A.Ed[ragged ? (long long)r : (long long)r * A.n_targets + k] = anybad ? INF_REF : e;
  wave_lds_sync();                         // the wave may reuse sm for its next structure
}

// grid = R * n_targets workgroups of one wave
__global__ __launch_bounds__(WAVE) void eval_kernel(EvalArgs A) {
  __shared__ EvalSmem sm;
  const bool ragged = A.rg.len != nullptr;
  eval_one(sm, A, ragged ? blockIdx.x : blockIdx.x / A.n_targets, ragged ? 0 : blockIdx.x % A.n_targets, threadIdx.x);
}

}  // namespace drna
