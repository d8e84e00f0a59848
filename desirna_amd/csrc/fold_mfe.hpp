// fold_mfe.hpp -- Zuker MFE fill + traceback (+ greedy pseudoknot re-folds) for one sequence per
// workgroup on gfx950.  Replaces, for a whole batch of replicas at once:
//   fc.mfe()                     reference utils/energy_scores.py:151   
//   RNA.fold(seq)[1]             reference utils/energy_scores.py:354   (f5[n] of the same fill)
//   get_pk_struct(seq, ss, fc)   reference utils/sequence_utils.py:1166-1228
// Recursions and traceback order: SURVEY.md App. A.3/A.4 (ViennaRNA 2.6.4 model, dangles=2).
//
// Design (MI355X): one workgroup per sequence sweeps the anti-diagonals d = j-i.  Tables live in
// HBM/L2 in DIAGONAL-MAJOR layout (row d holds cells (i, i+d), i = 1..n-d) so that, with one LANE per
// cell, every operand stream of both inner loops is unit-stride across the wave:
//   interior loop  (p,q) = (i+1+u1, j-1-u2)  ->  row d-2-u1-u2, column i+1+u1   (u1,u2 wave-uniform)
//   multiloop split fML[i,u] + fML[u+1,j]     ->  row tt, column i   and   row d-tt-1, column i+tt+1
// Loop-size terms are wave-uniform scalars taken from a host-built plan (tables.hpp), the work of a
// diagonal is cut into (64-cell block) x (chunk of the candidate/split lists) items spread over the
// waves, partial minima meet in LDS, and one lane per cell finalises c / fML.
// Per cell two words are stored: Wc = (c << 8) | info  and  CI = c + mismatchI[info], so the dominant
// "generic" interior candidates cost one load + add + min.
#pragma once
#include "fold_common.hpp"

namespace drna {

struct MfeArgs {
  const MfeTables* T;
  const Plan* plan;
  const int* hp_len;           // hairpin energy by loop size, >= L+2 entries
  const char* seqs;            // R x L ASCII
  int L;
  int ld;                      // row pitch of the per-sequence tables (>= L+2)
  int pk_rounds;               // 0 = plain MFE, 3 = reference pseudoknot heuristic
  int32_t* ws;                 // workspace: per sequence 5 tables of ld*ld int32
  long long ws_stride;         // int32 per sequence
  int32_t* Emfe;               // R  (dcal/mol, ViennaRNA INF convention not needed: always finite)
  char* ss;                    // R x L structure (pk-annotated if pk_rounds)
  int32_t* status;             // R
  Ragged rg;                   // ragged batch: per-sequence length / offsets (L is then overwritten per workgroup)
};

template <int NLEN>
struct MfeSmemCore {
  int stack[64];
  int mmH[128], mmI[128], mm1n[128], mm23[128], mmM[128], mmExt[128];
  int int11[1024];
  int d5[32], d3[32];
  int f5[NLEN + 2];
  short sec_i[NLEN + 2], sec_j[NLEN + 2];
  unsigned char sec_ml[NLEN + 2];
  unsigned char S[NLEN + 4];    // nucleotide codes, S[0] = S[n], S[n+1] = S[1]
  unsigned char Sp[NLEN + 4];   // pairing codes (4 = hard-constrained unpaired)
  char ssw[NLEN + 4];           // structure of the current round
  char sspk[NLEN + 4];          // accumulated (pk-annotated) structure
  int flag;
};

// general path: every DP table in HBM/L2, any n up to MAXN
struct MfeSmem : MfeSmemCore<MAXN> {
  int partI[PART_ITEMS * WAVE];
  int partK[PART_ITEMS * WAVE];
  // pairable cells of a diagonal, compacted (ascending i), and the inverse map; double-buffered by diagonal parity
  unsigned short plist[2][MAXN], cpos[2][MAXN + 2];
  int pcnt[2];
  // interior-loop plan staged from HBM: u1 | u2 << 8 | kind << 16, and the size term
  int plan_u[NPLAN], plan_L[NPLAN];
  // generic entries in slots of four consecutive ones (same loop size, consecutive u1): u1 | u2 << 8 of the first entry, and
  // the four size terms (INF where the slot has fewer entries)
  int plan_q[NPAIR_MAX];
  i32x4 plan_qL[NPAIR_MAX];
};

// one wave: list of the cells (i, i+d) that can pair (hard constraints of the pseudoknot rounds included)
__device__ __forceinline__ void mfe_build_plist(MfeSmem& sm, int d, int n, int lane) {
  const int par = d & 1;
  int cnt = 0;
  for (int i0 = 1; i0 <= n - d; i0 += WAVE) {
    const int i = i0 + lane;
    const bool on = i <= n - d && pair_type(sm.Sp[i], sm.Sp[i + d]) != 0;
    const unsigned long long m = __ballot(on);
    if (on) {
      const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
      sm.plist[par][pos] = (unsigned short)i;
      sm.cpos[par][i] = (unsigned short)pos;
    }
    cnt += __popcll(m);
  }
  if (lane == 0) sm.pcnt[par] = cnt;
}

// ---- loop energies on the device (per-lane arguments; used by finalize and traceback)

// ViennaRNA E_Hairpin for the pair (i,j) of type t; e = size term hairpin[u] (log-extrapolated beyond 30)
template <class SM>
__device__ __forceinline__ int mfe_hairpin_e(const SM& sm, const MfeTables& T, int e, int i, int j, int t) {
  const int u = j - i - 1;
  const int tau = t > 2 ? T.TermAU : 0;
  if (u == 3) {
    if (T.n_tri) {
      int code = 0;
      for (int k = 0; k < 5; k++) code |= sm.S[i + k] << (2 * k);
      for (int k = 0; k < T.n_tri; k++)
        if (T.tri_code[k] == code) return T.tri_e[k];
    }
    return e + tau;
  }
  if (u == 4 && T.n_tetra) {
    int code = 0;
    for (int k = 0; k < 6; k++) code |= sm.S[i + k] << (2 * k);
    for (int k = 0; k < T.n_tetra; k++)
      if (T.tetra_code[k] == code) return T.tetra_e[k];
  } else if (u == 6 && T.n_hexa) {
    int code = 0;
    for (int k = 0; k < 8; k++) code |= sm.S[i + k] << (2 * k);
    for (int k = 0; k < T.n_hexa; k++)
      if (T.hexa_code[k] == code) return T.hexa_e[k];
  }
  return e + sm.mmH[t * 16 + sm.S[i + 1] * 4 + sm.S[j - 1]];
}
template <class SM>
__device__ __forceinline__ int mfe_hairpin(const SM& sm, const MfeArgs& A, int i, int j, int t) {
  return mfe_hairpin_e(sm, *A.T, A.hp_len[j - i - 1], i, j, t);
}

// ViennaRNA E_IntLoop with the inner pair given by its packed info byte; arbitrary per-lane (u1,u2)
template <class SM>
__device__ __forceinline__ int mfe_intloop(const SM& sm, const MfeTables& T, int u1, int u2, int t,
                                           int si1, int sj1, int info) {
  const int t2 = info >> 4, sq1 = (info >> 2) & 3, sp1 = info & 3;
  const int nl = u1 > u2 ? u1 : u2, ns = u1 > u2 ? u2 : u1;
  if (nl == 0) return sm.stack[t * 8 + t2];
  if (ns == 0) {
    int e = T.bulge[nl];
    if (nl == 1) return e + sm.stack[t * 8 + t2];
    return e + (t > 2 ? T.TermAU : 0) + (t2 > 2 ? T.TermAU : 0);
  }
  if (ns == 1) {
    if (nl == 1) return sm.int11[(t * 8 + t2) * 16 + si1 * 4 + sj1];
    if (nl == 2)
      return (u1 == 1) ? T.int21[(t * 8 + t2) * 64 + si1 * 16 + sq1 * 4 + sj1]
                       : T.int21[(t2 * 8 + t) * 64 + sq1 * 16 + si1 * 4 + sp1];
    int e = T.interior[nl + 1] + min(T.max_ninio, (nl - ns) * T.ninio);
    return e + sm.mm1n[t * 16 + si1 * 4 + sj1] + sm.mm1n[info];
  }
  if (ns == 2) {
    if (nl == 2) return T.int22[(t * 8 + t2) * 256 + si1 * 64 + sp1 * 16 + sq1 * 4 + sj1];
    if (nl == 3) return T.interior[5] + T.ninio + sm.mm23[t * 16 + si1 * 4 + sj1] + sm.mm23[info];
  }
  int e = T.interior[nl + ns] + min(T.max_ninio, (nl - ns) * T.ninio);
  return e + sm.mmI[t * 16 + si1 * 4 + sj1] + sm.mmI[info];
}

template <class SM>
__device__ __forceinline__ int mfe_extstem(const SM& sm, int t, int i, int j, int n) {
  // E_ExtLoop(type, i>1 ? S[i-1] : -1, j<n ? S[j+1] : -1), dangles = 2
  int e;
  if (i > 1 && j < n) e = sm.mmExt[t * 16 + sm.S[i - 1] * 4 + sm.S[j + 1]];
  else if (i > 1) e = sm.d5[t * 4 + sm.S[i - 1]];
  else if (j < n) e = sm.d3[t * 4 + sm.S[j + 1]];
  else e = 0;
  return e;
}

// ---- fill of one sequence (all threads of the workgroup)

template <int NT>
__device__ void mfe_fill(MfeSmem& sm, const MfeArgs& A, int32_t* __restrict__ Wc, int32_t* __restrict__ CI,
                         int32_t* __restrict__ FML, int32_t* __restrict__ DML, int32_t* __restrict__ EXT) {
  constexpr int NW = NT / WAVE;
  const MfeTables& T = *A.T;
  const Plan& P = *A.plan;
  const int n = A.L, ld = A.ld;
  const int tid = threadIdx.x, lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(wave_id());   // SGPR: items, plan entries and branches stay scalar
  const int INF = INF_DEV, HALF = INF_DEV / 2;
  const int segG = P.seg[PK_GENERIC];
  int seg[PK_NKINDS];                   // first entry of every kind (wave-uniform: the kind of an entry stays scalar)
  for (int k = 0; k < PK_NKINDS; k++) seg[k] = P.seg[k];

  // rows read before they are written: fML diag 3, decomp diags 2 and 3
  for (int k = tid; k < ld; k += NT) {
    FML[3 * ld + k] = INF;
    DML[2 * ld + k] = INF;
    DML[3 * ld + k] = INF;
  }
  for (int e = tid; e < NPLAN; e += NT) { sm.plan_u[e] = P.u1[e] | (P.u2[e] << 8) | (P.kind[e] << 16); sm.plan_L[e] = P.L[e]; }
  const int nquad = P.n_quad;
  for (int p = tid; p < nquad; p += NT) {
    const int e = P.quad_e[p], c = P.quad_n[p];
    sm.plan_q[p] = P.u1[e] | (P.u2[e] << 8);
    sm.plan_qL[p] = i32x4{P.L[e], c > 1 ? P.L[e + 1] : INF, c > 2 ? P.L[e + 2] : INF, c > 3 ? P.L[e + 3] : INF};
  }
  if (wave == 0 && TURN + 1 < n) mfe_build_plist(sm, TURN + 1, n, lane);
  __syncthreads();
  const auto rsF = __builtin_amdgcn_make_buffer_rsrc((void*)FML, (short)0, (int)((long long)ld * ld * 4), 0x00020000);
  const auto rsC = __builtin_amdgcn_make_buffer_rsrc((void*)CI, (short)0, (int)((long long)ld * ld * 4), 0x00020000);

  for (int d = TURN + 1; d < n; d++) {
    const int ncell = n - d, par = d & 1;
    const int nblk = (ncell + WAVE - 1) / WAVE;
    int H = NW / nblk;
    if (H < 1) H = 1;
    // interior loops run over the PAIRABLE cells only (compact list): nblkP blocks x HI chunks of the plan
    const int pc = sm.pcnt[par];
    const int nblkP = (pc + WAVE - 1) / WAVE;
    int HI = nblkP ? NW / nblkP : 1;
    if (HI < 1) HI = 1;
    const int nI = nblkP * HI, nK = nblk * H;

    // ---------------- phase A: candidate minima, lane = cell
    for (int item = wave; item < nI + nK; item += NW) {
      if (item < nI) {
        const int cb = item / HI, h = item - cb * HI;
        const int q = cb * WAVE + lane;
        const int i = sm.plist[par][q < pc ? q : pc - 1];
        const int j = i + d;
        const int t = pair_type(sm.Sp[i], sm.Sp[j]);
        const int si1 = sm.S[i + 1], sj1 = sm.S[j - 1];
        const int ij = t * 16 + si1 * 4 + sj1;
        int accI = INF;
        const int tau = t > 2 ? T.TermAU : 0;
        // Four plan entries per pass, written as stages (entries from LDS, then the four table loads, then the
        // arithmetic): one L2 round trip per four entries instead of two per entry.
        // special kinds (stack, bulges, 1x1, 2x1, 1xn, 2x2, 2x3): strided over the chunk
        for (int e = h; e < segG; e += 4 * HI) {
          int pu[4], pl[4], w[4];
          bool ok[4];
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int ee = as_vector(min(e + k * HI, segG - 1));
            pu[k] = sm.plan_u[ee]; pl[k] = sm.plan_L[ee];
          }
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int u1 = pu[k] & 255, u2 = (pu[k] >> 8) & 255;
            const int dp = d - 2 - u1 - u2;
            ok[k] = e + k * HI < segG && dp > TURN;
            w[k] = Wc[ok[k] ? dp * ld + i + 1 + u1 : 0];
          }
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int cpq = w[k] >> 8, info = w[k] & 127, t2 = info >> 4;
            int en;
            switch (plan_kind(seg, e + k * HI)) {
              case PK_STACK: en = cpq + sm.stack[t * 8 + t2]; break;
              case PK_BULGE1: en = cpq + pl[k] + sm.stack[t * 8 + t2]; break;
              case PK_BULGEN: en = cpq + pl[k] + tau + (t2 > 2 ? T.TermAU : 0); break;
              case PK_INT11: en = cpq + sm.int11[(t * 8 + t2) * 16 + si1 * 4 + sj1]; break;
              case PK_INT21: en = cpq + T.int21[(t * 8 + t2) * 64 + si1 * 16 + ((info >> 2) & 3) * 4 + sj1]; break;
              case PK_INT12: en = cpq + T.int21[(t2 * 8 + t) * 64 + ((info >> 2) & 3) * 16 + si1 * 4 + (info & 3)]; break;
              case PK_1XN: en = cpq + pl[k] + sm.mm1n[ij] + sm.mm1n[info]; break;
              case PK_INT22:
                en = cpq + T.int22[(t * 8 + t2) * 256 + si1 * 64 + (info & 3) * 16 + ((info >> 2) & 3) * 4 + sj1];
                break;
              default: /* PK_INT23 */ en = cpq + pl[k] + sm.mm23[ij] + sm.mm23[info]; break;
            }
            accI = min(accI, ok[k] ? en : INF);
          }
        }
        // generic interior loops: c + mismatchI(inner) precombined in CI, size term is a scalar
        int accG = INF;
        // (slots of four entries that sit in consecutive cells of one row of CI: one 16-byte load each)
        for (int p = h; p < nquad; p += 4 * HI) {
          int pu[4];
          i32x4 pl[4], v[4];
          bool ok[4];
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int pp = as_vector(min(p + k * HI, nquad - 1));
            pu[k] = sm.plan_q[pp]; pl[k] = sm.plan_qL[pp];
          }
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int u1 = pu[k] & 255, u2 = (pu[k] >> 8) & 255;
            const int dp = d - 2 - u1 - u2;
            ok[k] = p + k * HI < nquad && dp > TURN;
            v[k] = buf_load_i32x4(rsC, ok[k] ? (dp * ld + i + 1 + u1) * 4 : 0, 0);
          }
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int m = min(min(v[k].x + pl[k].x, v[k].y + pl[k].y), min(v[k].z + pl[k].z, v[k].w + pl[k].w));
            accG = min(accG, ok[k] ? m : INF);
          }
        }
        accI = min(accI, accG + sm.mmI[ij]);
        sm.partI[item * WAVE + lane] = accI;
      } else {
        // multiloop split: fML[i,u] + fML[u+1,j], u = i + tt
        const int it = item - nI;
        const int b = it / H, h = it - b * H;
        const int i0 = b * WAVE + lane + 1;
        const int i = i0 <= ncell ? i0 : ncell;
        // fML[i,i+tt] at (tt ld + i) 4, fML[i+tt+1,j] at ((d-tt-1) ld + i+tt+1) 4 bytes from FML; a step of H in tt moves them
        // by +4 H ld and -4 H (ld - 1): buffer loads with one running 32-bit offset each and the strides in SGPRs
        int acc0 = INF, acc1 = INF;
        int tt = TURN + 1 + h;
        const int stepA = 4 * H * ld, stepC = 4 * H * (ld - 1);
        int vA = (tt * ld + i) * 4;
        int vC = ((d - tt - 1) * ld + i + tt + 1) * 4;                                     // operand of tt (never negative)
        for (; tt + 7 * H <= d - TURN - 2; tt += 8 * H) {                                   // eight split points in flight
          const int vCl = vC - 7 * stepC;                                                   // operand of tt + 7 H: in range here
          int a[8], c[8];
#pragma unroll
          for (int k = 0; k < 8; k++) { a[k] = buf_load_i32(rsF, vA, k * stepA); c[k] = buf_load_i32(rsF, vCl, (7 - k) * stepC); }
          vA += 8 * stepA; vC -= 8 * stepC;
#pragma unroll
          for (int k = 0; k < 8; k += 2) { acc0 = min(acc0, a[k] + c[k]); acc1 = min(acc1, a[k + 1] + c[k + 1]); }
        }
        for (; tt + 3 * H <= d - TURN - 2; tt += 4 * H) {
          const int vCl = vC - 3 * stepC;                                                   // operand of tt + 3 H: in range here
          const int a0 = buf_load_i32(rsF, vA, 0), c0 = buf_load_i32(rsF, vCl, 3 * stepC);
          const int a1 = buf_load_i32(rsF, vA, stepA), c1 = buf_load_i32(rsF, vCl, 2 * stepC);
          const int a2 = buf_load_i32(rsF, vA, 2 * stepA), c2 = buf_load_i32(rsF, vCl, stepC);
          const int a3 = buf_load_i32(rsF, vA, 3 * stepA), c3 = buf_load_i32(rsF, vCl, 0);
          vA += 4 * stepA; vC -= 4 * stepC;
          acc0 = min(acc0, min(a0 + c0, a2 + c2)); acc1 = min(acc1, min(a1 + c1, a3 + c3));
        }
        for (; tt <= d - TURN - 2; tt += H) {
          acc0 = min(acc0, buf_load_i32(rsF, vA, 0) + buf_load_i32(rsF, vC, 0));
          vA += stepA; vC -= stepC;
        }
        sm.partK[it * WAVE + lane] = min(acc0, acc1);
      }
    }
    __syncthreads();

    // ---------------- phase B: one lane per cell finalises c and fML
    for (int i = tid + 1; i <= ncell; i += NT) {
      const int b = (i - 1) / WAVE, ln = (i - 1) % WAVE;
      int aI = INF, aK = INF;
      for (int h = 0; h < H; h++) aK = min(aK, sm.partK[(b * H + h) * WAVE + ln]);
      const int j = i + d;
      const int t = pair_type(sm.Sp[i], sm.Sp[j]);
      if (t) {
        const int pos = sm.cpos[par][i];
        for (int h = 0; h < HI; h++) aI = min(aI, sm.partI[((pos >> 6) * HI + h) * WAVE + (pos & 63)]);
      }
      const int tau = t > 2 ? T.TermAU : 0;
      int c = INF;
      int info = 0;
      if (t) {
        c = mfe_hairpin(sm, A, i, j, t);
        c = min(c, aI);
        const int dml = DML[(d - 2) * ld + i + 1];
        if (dml < HALF)
          c = min(c, dml + T.MLclosing + T.MLintern + tau + sm.mmM[rtype_of(t) * 16 + sm.S[j - 1] * 4 + sm.S[i + 1]]);
        if (c >= HALF) c = INF;
        info = (rtype_of(t) << 4) | (sm.S[j + 1] << 2) | sm.S[i - 1];
      }
      Wc[d * ld + i] = c * 256 + info;
      CI[d * ld + i] = c < INF ? c + sm.mmI[info] : INF;
      EXT[j * ld + i] = c < INF ? c + tau + mfe_extstem(sm, t, i, j, n) : INF;
      int f = INF;
      const int fa = FML[(d - 1) * ld + i + 1], fb = FML[(d - 1) * ld + i];
      if (fa < HALF) f = fa + T.MLbase;
      if (fb < HALF) f = min(f, fb + T.MLbase);
      if (c < INF) f = min(f, c + T.MLintern + tau + sm.mmM[t * 16 + sm.S[i - 1] * 4 + sm.S[j + 1]]);
      const int dec = aK >= HALF ? INF : aK;
      DML[d * ld + i] = dec;
      FML[d * ld + i] = min(f, dec);
    }
    if (wave == NW - 1 && d + 1 < n) mfe_build_plist(sm, d + 1, n, lane);     // list of the next diagonal
    __syncthreads();
  }

  // ---------------- exterior loop f5 (wave 0; column j of EXT is unit-stride)
  if (wave == 0) {
    for (int j = lane; j <= TURN + 1 && j <= n; j += WAVE) sm.f5[j] = 0;
    for (int j = TURN + 2; j <= n; j++) {
      int m = INF;
      for (int i = lane + 1; i <= j - TURN - 1; i += WAVE) {
        const int x = EXT[j * ld + i];
        if (x < HALF) m = min(m, sm.f5[i - 1] + x);
      }
      m = wave_min_i32(m);
      const int prev = sm.f5[j - 1];
      sm.f5[j] = prev < m ? prev : m;   // every lane stores the same value
    }
  }
  __syncthreads();
}

// ---- traceback by wave 0 (wave-uniform control flow, lanes scan candidates in ViennaRNA's order)

// fML accessor of the general path: diagonal-major table in global memory
struct FmlGlobal {
  const int32_t* p;
  int ld;
  __device__ __forceinline__ int operator()(int d, int i) const { return p[d * ld + i]; }
};

// Sector containers of the traceback.  TbStack: the private stack of the one-wave traceback (general and strip kernels).  TbShared
// (round 3, LDS-resident kernels): a queue in LDS that several waves work from -- the sectors of a structure (exterior stems, the
// branches of a multiloop) are independent, the traceback is ~0.9 us of mostly scalar control flow per event, and one wave took
// 0.074 of the MFE fold's 0.50 ms at 200 nt with fifteen waves idle.  Entry c is published by its ml + 1 in sec_ml[c] (0 = not
// written yet); tbq = {next to claim, next free, sectors pushed and not finished, 1 done / 2 failed}.  The caller zeroes sec_ml
// and tbq and writes entry 0 before the barrier in front of the traceback.
template <class SM> struct TbStack {
  SM& sm;
  int sp = 0;
  __device__ __forceinline__ void init(int n) { sm.sec_i[0] = 1; sm.sec_j[0] = (short)n; sm.sec_ml[0] = 0; sp = 1; }
  __device__ __forceinline__ bool pop(int& i, int& j, int& ml) {
    if (sp == 0) return false;
    sp--;
    i = sm.sec_i[sp]; j = sm.sec_j[sp]; ml = sm.sec_ml[sp];
    return true;
  }
  __device__ __forceinline__ void push(int i, int j, int ml) { sm.sec_i[sp] = (short)i; sm.sec_j[sp] = (short)j; sm.sec_ml[sp] = (unsigned char)ml; sp++; }
  __device__ __forceinline__ void done_one() {}
  __device__ __forceinline__ void fail() {}
  __device__ __forceinline__ bool failed() const { return false; }
};
template <class SM> struct TbShared {
  SM& sm;
  __device__ __forceinline__ void init(int) {}
  __device__ __forceinline__ bool pop(int& i, int& j, int& ml) {
    int c = 0;
    if (lane_id() == 0) c = atomicAdd(&sm.tbq[0], 1);
    c = __builtin_amdgcn_readfirstlane(c);
    if (c > (int)(sizeof(sm.sec_ml) / sizeof(sm.sec_ml[0])) - 1) return false;
    for (;;) {
      const int m = __builtin_amdgcn_readfirstlane((int)*reinterpret_cast<volatile unsigned char*>(&sm.sec_ml[c]));
      if (m) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");      // pairs with push()'s release: the payload loads stay behind the flag read
        i = sm.sec_i[c]; j = sm.sec_j[c]; ml = m - 1;
        return true;
      }
      if (__builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int*>(&sm.tbq[3]))) return false;
      spin_pause();
    }
  }
  __device__ __forceinline__ void push(int i, int j, int ml) {
    if (lane_id() == 0) {
      atomicAdd(&sm.tbq[2], 1);
      const int c = atomicAdd(&sm.tbq[1], 1);
      // The queue is append-only, so it must hold every sector of a structure at once: each push is a distinct interval that
      // holds a pair of its own (an exterior stem or a multiloop branch), hence at most n / 2 + 1 <= NLEN + 2 entries.  The guard
      // is for a corrupted table only; it ends the traceback as a failure (ST_TRACEBACK), like every table value it cannot reproduce.
      if (c > (int)(sizeof(sm.sec_ml) / sizeof(sm.sec_ml[0])) - 1) { *reinterpret_cast<volatile int*>(&sm.tbq[3]) = 2; return; }
      sm.sec_i[c] = (short)i; sm.sec_j[c] = (short)j;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      *reinterpret_cast<volatile unsigned char*>(&sm.sec_ml[c]) = (unsigned char)(ml + 1);
    }
  }
  __device__ __forceinline__ void done_one() {
    if (lane_id() == 0 && atomicAdd(&sm.tbq[2], -1) == 1) *reinterpret_cast<volatile int*>(&sm.tbq[3]) = 1;
  }
  __device__ __forceinline__ void fail() { if (lane_id() == 0) *reinterpret_cast<volatile int*>(&sm.tbq[3]) = 2; }
  __device__ __forceinline__ bool failed() const { return *reinterpret_cast<volatile int*>(&sm.tbq[3]) == 2; }
};

#ifdef DRNA_TL_TB
static __device__ long long* drna_tl_tb_ptr = nullptr;      // diagnostics (tools/timeline.py mfe, TL_TB=1): where block 0's traceback waves report
#endif
template <class SM, class FMLACC, class QUEUE>
__device__ __forceinline__ bool mfe_traceback_q(SM& sm, const MfeArgs& A, const int32_t* __restrict__ Wc,
                                       const FMLACC FML, const int32_t* __restrict__ EXT, QUEUE Q) {
  const MfeTables& T = *A.T;
  const Plan& P = *A.plan;
  const int n = A.L, ld = A.ld, lane = lane_id();
  const int HALF = INF_DEV / 2;
  bool ok = true;
  // The interior scan gives lane l of round r the candidate k = 64 r + l of the canonical order.  Its shape, its kind and its
  // size term (what E_IntLoop adds for the loop size and asymmetry) are fixed for the whole traceback: fetched and computed
  // once, kept in registers, so that a scan round is LDS lookups plus at most one global load (2x1 / 1x2 / 2x2 lanes).
  constexpr int NRND = (NPLAN + WAVE - 1) / WAVE;
  int tb_shape[NRND], tb_L[NRND];
#pragma unroll
  for (int rnd = 0; rnd < NRND; rnd++) {        // (packed on the host, tables.hpp: sixteen independent loads, first used by the first scan)
    const int k = rnd * WAVE + lane;
    tb_shape[rnd] = k < NPLAN ? P.tb_shape[k] : (PK_STACK << 16);
    tb_L[rnd] = k < NPLAN ? P.tb_L[k] : 0;
  }
  // one sector: an exterior interval (ml 0), a multiloop segment (ml 1); returns false when a table value cannot be reproduced
#ifdef DRNA_TL_TB
  int tb_events = 0;
#endif
  Q.init(n);
  auto sector = [&](int i, int j, const int ml) -> bool {
    bool have_pair = false;
    if (ml == 0) {
      // ---- exterior: vrna_BT_ext_loop_f5
      if (j < TURN + 2) return true;
      // largest jj <= j with f5[jj] != f5[jj-1]
      int jj = -1;
      for (int base = j; base >= 1 && jj < 0; base -= WAVE) {
        const int x = base - lane;
        const bool hit = x >= 1 && sm.f5[x] != sm.f5[x - 1];
        const int fl = first_lane(__ballot(hit));
        if (fl >= 0) jj = base - fl;
      }
      if (jj < TURN + 2) return true;
      const int fij = sm.f5[jj];
      int u = -1;
      for (int base = jj - TURN - 1; base >= 1 && u < 0; base -= WAVE) {
        const int x = base - lane;
        bool hit = false;
        if (x >= 1) {
          const int e = EXT[jj * ld + x];
          hit = e < HALF && fij == e + sm.f5[x - 1];
        }
        const int fl = first_lane(__ballot(hit));
        if (fl >= 0) u = base - fl;
      }
      if (u < 0) return false;
      Q.push(1, (u - 1), 0);
      i = u; j = jj; have_pair = true;
    } else if (ml == 1) {
      // ---- multiloop segment: vrna_BT_mb_loop_split
      // strip unpaired 3' bases: largest run k with fML[i,j-k] == fML[i,j-k-1] + MLbase
      for (;;) {
        const int jx = j - lane;            // lane tests position jx
        bool stop = true;
        if (jx > i) {
          const int a = FML(jx - i, i), bq = (jx - 1 - i) >= 0 ? FML(jx - 1 - i, i) : INF_DEV;
          stop = !(a == bq + T.MLbase);
        }
        const int fl = first_lane(__ballot(stop));
        if (fl >= 0) { j -= fl; break; }
        j -= WAVE;
      }
      for (;;) {
        const int ix = i + lane;
        bool stop = true;
        if (ix < j) {
          const int a = FML(j - ix, ix), bq = FML(j - ix - 1, ix + 1);
          stop = !(a == bq + T.MLbase);
        }
        const int fl = first_lane(__ballot(stop));
        if (fl >= 0) { i += fl; break; }
        i += WAVE;
      }
      if (j < i + TURN + 1) return false;
      const int d = j - i;
      const int fij = FML(d, i);
      const int w = Wc[d * ld + i];
      const int cij = w >> 8;
      const int t = pair_type(sm.Sp[i], sm.Sp[j]);
      if (t && cij < HALF &&
          fij == cij + T.MLintern + (t > 2 ? T.TermAU : 0) + sm.mmM[t * 16 + sm.S[i - 1] * 4 + sm.S[j + 1]]) {
        have_pair = true;
      } else {
        int u = -1;
        for (int base = i + TURN + 1; base <= j - TURN - 2 && u < 0; base += WAVE) {
          const int x = base + lane;
          bool hit = false;
          if (x <= j - TURN - 2) hit = fij == FML(x - i, i) + FML(j - x - 1, x + 1);
          const int fl = first_lane(__ballot(hit));
          if (fl >= 0) u = base + fl;
        }
        if (u < 0) return false;
        Q.push(i, u, 1);
        Q.push((u + 1), j, 1);
      }
    } else {
      have_pair = true;
    }
    // ---- pair (i,j): hairpin, interior (p ascending, q descending), multiloop
    while (have_pair) {
#ifdef DRNA_TL_TB
      tb_events++;
#endif
      if (lane == 0) { sm.ssw[i - 1] = '('; sm.ssw[j - 1] = ')'; }
      const int d = j - i;
      const int t = pair_type(sm.Sp[i], sm.Sp[j]);
      // (i,j) and the pair stacked on it are fetched together: along a helix the stack is the first interior candidate in
      // ViennaRNA's order, so one L2 round trip settles the step instead of three (cell, plan words, inner cells)
      const int w_in = d - 2 > TURN ? Wc[(d - 2) * ld + i + 1] : INF_DEV * 256;
      const int cij = Wc[d * ld + i] >> 8;
      if (cij == mfe_hairpin(sm, A, i, j, t)) break;
      const int si1 = sm.S[i + 1], sj1 = sm.S[j - 1];
      if ((w_in >> 8) < HALF && cij == (w_in >> 8) + mfe_intloop(sm, T, 0, 0, t, si1, sj1, w_in & 127)) {
        i++; j--;
        continue;
      }
      int found = -1;
      const int tij = t * 16 + si1 * 4 + sj1, tauI = t > 2 ? T.TermAU : 0;
#pragma unroll
      for (int rnd = 0; rnd < NRND; rnd++) {
        if (found >= 0) break;
        const int k = rnd * WAVE + lane;
        const int u1 = tb_shape[rnd] & 255, u2 = (tb_shape[rnd] >> 8) & 255, kind = tb_shape[rnd] >> 16;
        const int dp = d - 2 - u1 - u2;
        const bool live = k < NPLAN && dp > TURN;
        const int w = live ? Wc[dp * ld + i + 1 + u1] : INF_DEV * 256;
        const int cpq = w >> 8, info = w & 127, t2 = info >> 4, sq1 = (info >> 2) & 3, sp1 = info & 3;
        // the mismatch-pair kinds share one code path through the table of their kind (mm1n, mm23, mmI: LDS)
        const int* mm = kind == PK_1XN ? sm.mm1n : kind == PK_INT23 ? sm.mm23 : sm.mmI;
        int en = mm[tij] + mm[info];
        const int e_stack = sm.stack[t * 8 + t2], e_11 = sm.int11[(t * 8 + t2) * 16 + si1 * 4 + sj1];
        en = kind == PK_STACK || kind == PK_BULGE1 ? e_stack : en;
        en = kind == PK_BULGEN ? tauI + (t2 > 2 ? T.TermAU : 0) : en;
        en = kind == PK_INT11 ? e_11 : en;
        if (kind == PK_INT21) en = T.int21[(t * 8 + t2) * 64 + si1 * 16 + sq1 * 4 + sj1];
        else if (kind == PK_INT12) en = T.int21[(t2 * 8 + t) * 64 + sq1 * 16 + si1 * 4 + sp1];
        else if (kind == PK_INT22) en = T.int22[(t * 8 + t2) * 256 + si1 * 64 + sp1 * 16 + sq1 * 4 + sj1];
        const bool hit = live && cpq < HALF && cij == cpq + en + tb_L[rnd];
        const int fl = first_lane(__ballot(hit));
        if (fl >= 0) found = rnd * WAVE + fl;
      }
      if (found >= 0) {
        i = i + 1 + P.tb_u1[found];
        j = j - 1 - P.tb_u2[found];
        continue;
      }
      // multiloop closed by (i,j): vrna_BT_mb_loop
      const int e = cij - T.MLclosing - T.MLintern - (t > 2 ? T.TermAU : 0) -
                    sm.mmM[rtype_of(t) * 16 + sj1 * 4 + si1];
      int u = -1;
      for (int base = i + 2 + TURN; base < j - 2 - TURN && u < 0; base += WAVE) {
        const int x = base + lane;
        bool hit = false;
        if (x < j - 2 - TURN) hit = e == FML(x - i - 1, i + 1) + FML(j - 1 - x - 1, x + 1);
        const int fl = first_lane(__ballot(hit));
        if (fl >= 0) u = base + fl;
      }
      if (u < 0) return false;
      Q.push((i + 1), u, 1);
      Q.push((u + 1), (j - 1), 1);
      break;
    }
    return true;
  };
  int qi = 0, qj = 0, qml = 0;
#ifdef DRNA_TL_TB
  long long tb_t0 = (long long)wall_clock64(), tb_busy = 0;
  int tb_sectors = 0;
#endif
  while (Q.pop(qi, qj, qml)) {
#ifdef DRNA_TL_TB
    const long long s0 = (long long)wall_clock64();
#endif
    if (!sector(qi, qj, qml)) { Q.fail(); ok = false; break; }
    Q.done_one();
#ifdef DRNA_TL_TB
    tb_busy += (long long)wall_clock64() - s0; tb_sectors++;
#endif
  }
#ifdef DRNA_TL_TB
  if (lane == 0 && blockIdx.x == 0 && drna_tl_tb_ptr) {
    long long* o = drna_tl_tb_ptr + (72 << 8) + 8 * wave_id();       // (row 72 of the mark table: beyond the main role's and the helper's rows)
    o[0] = tb_t0; o[1] = (long long)wall_clock64(); o[2] = tb_busy; o[3] = tb_sectors; o[4] = tb_events;
  }
#endif
  return ok && !Q.failed();
}

// the one-wave traceback with a private stack
template <class SM, class FMLACC>
__device__ inline bool mfe_traceback(SM& sm, const MfeArgs& A, const int32_t* __restrict__ Wc,
                                     const FMLACC FML, const int32_t* __restrict__ EXT) {
  return mfe_traceback_q(sm, A, Wc, FML, EXT, TbStack<SM>{sm});
}


// ---- the kernel: one workgroup per sequence

template <int NT>
__global__ __launch_bounds__(NT) void mfe_kernel(MfeArgs A) {
  __shared__ MfeSmem sm;
  const int r = A.rg.idx ? A.rg.idx[blockIdx.x] : blockIdx.x;
  if (A.rg.len) A.L = A.rg.len[r];
  const long long so = A.rg.off ? (long long)A.rg.off[r] : (long long)r * A.L;      // offset in seqs / ss
  const int n = A.L, ld = A.ld, tid = threadIdx.x;
  const MfeTables& T = *A.T;
  int32_t* base = A.ws + (long long)r * A.ws_stride;
  const long long tab = (long long)ld * ld;
  int32_t* Wc = base;
  int32_t* CI = base + tab;
  int32_t* FML = base + 2 * tab;
  int32_t* DML = base + 3 * tab;
  int32_t* EXT = base + 4 * tab;

  for (int k = tid; k < 64; k += NT) sm.stack[k] = T.stack[k];
  for (int k = tid; k < 128; k += NT) {
    sm.mmH[k] = T.mmH[k]; sm.mmI[k] = T.mmI[k]; sm.mm1n[k] = T.mm1n[k];
    sm.mm23[k] = T.mm23[k]; sm.mmM[k] = T.mmM[k]; sm.mmExt[k] = T.mmExt[k];
  }
  for (int k = tid; k < 1024; k += NT) sm.int11[k] = T.int11[k];
  for (int k = tid; k < 32; k += NT) { sm.d5[k] = T.d5[k]; sm.d3[k] = T.d3[k]; }
  if (tid == 0) sm.flag = 0;
  __syncthreads();
  const char* seq = A.seqs + so;
  for (int k = tid; k < n; k += NT) {
    const int c = enc_nt(seq[k]);
    if (c < 0) sm.flag = 1;
    sm.S[k + 1] = (unsigned char)(c < 0 ? 0 : c);
    sm.Sp[k + 1] = (unsigned char)(c < 0 ? 4 : c);
    sm.sspk[k] = '.';
  }
  __syncthreads();
  if (tid == 0) {
    sm.S[0] = sm.S[n]; sm.S[n + 1] = sm.S[1];
    sm.Sp[0] = 4; sm.Sp[n + 1] = 4;
  }
  __syncthreads();
  if (sm.flag) {
    if (tid == 0) { A.status[r] = ST_BAD_CHAR; A.Emfe[r] = 0; }
    for (int k = tid; k < n; k += NT) A.ss[so + k] = '.';
    return;
  }

  int status = ST_OK;
  for (int round = 0; round <= A.pk_rounds; round++) {
    for (int k = tid; k < n; k += NT) sm.ssw[k] = '.';
    mfe_fill<NT>(sm, A, Wc, CI, FML, DML, EXT);     // ends with a barrier
    if (wave_id() == 0) {
      const bool ok = mfe_traceback(sm, A, Wc, FmlGlobal{FML, ld}, EXT);
      if (lane_id() == 0) {
        if (round == 0) A.Emfe[r] = sm.f5[n];
        sm.flag = ok ? 0 : 1;
      }
    }
    __syncthreads();
    if (sm.flag) { status = ST_TRACEBACK; break; }
    // merge this round into the annotated structure; bracket family of round k: () [] <> {}
    const char op = round == 0 ? '(' : round == 1 ? '[' : round == 2 ? '<' : '{';
    const char cl = round == 0 ? ')' : round == 1 ? ']' : round == 2 ? '>' : '}';
    __syncthreads();
    int any = 0;
    for (int k = tid; k < n; k += NT) {
      const char ch = sm.ssw[k];
      if (ch == '(') { sm.sspk[k] = op; any = 1; }
      else if (ch == ')') sm.sspk[k] = cl;
      if (sm.sspk[k] != '.') sm.Sp[k + 1] = 4;      // hc 'x': already paired positions stay unpaired
    }
    if (any) sm.flag = 2;
    __syncthreads();
    // reference sequence_utils.py:1194,1210: the next re-fold happens only if this one found a pair
    const bool more = (round == 0) || (sm.flag == 2);
    __syncthreads();
    if (tid == 0) sm.flag = 0;
    __syncthreads();
    if (!more) break;
  }
  for (int k = tid; k < n; k += NT) A.ss[so + k] = sm.sspk[k];
  if (tid == 0) A.status[r] = status;
}

}  // namespace drna
