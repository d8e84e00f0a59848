// fold_pf.hpp -- McCaskill partition function (inside only) for one sequence per workgroup on gfx950.
// Replaces fc.pf()[1] with md.compute_bpp = 0: reference utils/energy_scores.py:27-28,150.
// Recursions, pf_smooth'ed dangle factors and per-nucleotide scaling: SURVEY.md App. A.5.
//
// Same anti-diagonal / lane-per-cell / diagonal-major design as fold_mfe.hpp with (+,x) in fp64
// instead of (min,+) in int32.  Differences worth knowing:
//   * only Z is needed, so the exterior recursion is the 1-D q5[j] (not ViennaRNA's full q[i,j]);
//     B_alg for the PF half is therefore the "q5 only" figure of SURVEY 8(d);
//   * qm[i,j] = qm1[i,j] + D[i,j] + U[i,j] with D[i,j] = sum_k qm[i,k-1] qm1[k,j] (the O(n^3) part;
//     D[i+1,j-1] is also the multiloop term of qb[i,j]) and U[i,j] = sum_k expMLbase^(k-i) qm1[k,j]
//     carried by the O(1) recurrence U[i,j] = b (qm1[i+1,j] + U[i+1,j]);
//   * partial sums of a cell are combined in a fixed order, so results are bit-reproducible.
#pragma once
#include "fold_common.hpp"

namespace drna {

struct PfArgs {
  const PfTables* T;
  const Plan* plan;
  const double* hp_w;          // hairpin Boltzmann factor by size (scale[u+2] folded in)
  const double* scale;         // pf_scale^-k
  const double* eMLb;          // (expMLbase / pf_scale)^k
  const char* seqs;            // R x L ASCII
  int L;
  int ld;
  double* ws;                  // per sequence 7 tables of ld*ld doubles + 1 of ld*ld bytes (as doubles/8)
  long long ws_stride;         // doubles per sequence
  double* Epf;                 // R  (kcal/mol)
  int32_t* status;             // R
  double* q5out = nullptr;     // optional: q5[0..L] per sequence for the outside recursion (fold_outside.hpp)
  long long q5_stride = 0;     // doubles per sequence
  Ragged rg;                   // ragged batch: per-sequence length / offsets (L is then overwritten per workgroup)
  // pf_lds_kernel with a helper workgroup per sequence (fold_pf_lds.hpp, pf_kfar_helper): grid pair_grid(R), blocks by pair_block (fold_common.hpp)
  int* hflags = nullptr;       // per sequence two 128-byte lines: [r * 64] written by the main workgroup, [r * 64 + 32] by the helper
  int hbase = 0;               // epoch << 12; flag = hbase + last published diagonal (compares are wrap-safe)
  int helper = 0;
};

struct PfSmem {
  double stack[64];
  double mmH[128], mmI[128], mm1n[128], mm23[128], mmM[128], mmExt[128];
  double int11[1024];
  double d5[32], d3[32];
  double partI[PART_ITEMS * WAVE];
  double partK[PART_ITEMS * WAVE];
  double q5[MAXN + 2];
  // pairable cells of a diagonal, compacted (ascending i), and the inverse map; double-buffered by diagonal parity
  unsigned short plist[2][MAXN], cpos[2][MAXN + 2];
  int pcnt[2];
  unsigned char S[MAXN + 4];
  int flag;
  // interior-loop plan staged from HBM: u1 | u2 << 8 | kind << 16, and the Boltzmann factor of the size term
  int plan_u[NPLAN];
  double plan_W[NPLAN];
  // generic entries in slots of two consecutive ones (same loop size, consecutive u1): u1 | u2 << 8 | count << 16 of the
  // first entry, and the two Boltzmann factors
  int plan_p[NPAIR_MAX];
  f64x2 plan_pW[NPAIR_MAX];
};

// one wave: list of the cells (i, i+d) that can pair
__device__ __forceinline__ void pf_build_plist(PfSmem& sm, int d, int n, int lane) {
  const int par = d & 1;
  int cnt = 0;
  for (int i0 = 1; i0 <= n - d; i0 += WAVE) {
    const int i = i0 + lane;
    const bool on = i <= n - d && pair_type(sm.S[i], sm.S[i + d]) != 0;
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

__device__ __forceinline__ double pf_hairpin(const PfSmem& sm, const PfArgs& A, int i, int j, int t) {
  const PfTables& T = *A.T;
  const int u = j - i - 1;
  const double q = A.hp_w[u];
  if (u == 3) {
    if (T.n_tri) {
      int code = 0;
      for (int k = 0; k < 5; k++) code |= sm.S[i + k] << (2 * k);
      for (int k = 0; k < T.n_tri; k++)
        if (T.tri_code[k] == code) return T.tri_w[k] * A.scale[u + 2];
    }
    return t > 2 ? q * T.TermAU : q;
  }
  if (u == 4 && T.n_tetra) {
    int code = 0;
    for (int k = 0; k < 6; k++) code |= sm.S[i + k] << (2 * k);
    for (int k = 0; k < T.n_tetra; k++)
      if (T.tetra_code[k] == code) return T.tetra_w[k] * A.scale[u + 2];
  } else if (u == 6 && T.n_hexa) {
    int code = 0;
    for (int k = 0; k < 8; k++) code |= sm.S[i + k] << (2 * k);
    for (int k = 0; k < T.n_hexa; k++)
      if (T.hexa_code[k] == code) return T.hexa_w[k] * A.scale[u + 2];
  }
  return q * sm.mmH[t * 16 + sm.S[i + 1] * 4 + sm.S[j - 1]];
}

// exp_E_ExtLoop / exp_E_MLstem neighbour rule: a neighbour exists only inside the sequence
__device__ __forceinline__ double pf_endstem(const double* mm, const PfSmem& sm, int t, int i, int j, int n) {
  if (i > 1 && j < n) return mm[t * 16 + sm.S[i - 1] * 4 + sm.S[j + 1]];
  if (i > 1) return sm.d5[t * 4 + sm.S[i - 1]];
  if (j < n) return sm.d3[t * 4 + sm.S[j + 1]];
  return 1.0;
}

template <int NT>
__global__ __launch_bounds__(NT) void pf_kernel(PfArgs A) {
  __shared__ PfSmem sm;
  constexpr int NW = NT / WAVE;
  const PfTables& T = *A.T;
  const Plan& P = *A.plan;
  const int r = A.rg.idx ? A.rg.idx[blockIdx.x] : blockIdx.x;
  if (A.rg.len) A.L = A.rg.len[r];
  const int n = A.L, ld = A.ld;
  const int tid = threadIdx.x, lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(wave_id());
  const int segG = P.seg[PK_GENERIC];
  int seg[PK_NKINDS];                   // first entry of every kind (wave-uniform: the kind of an entry stays scalar)
  for (int k = 0; k < PK_NKINDS; k++) seg[k] = P.seg[k];

  double* base = A.ws + (long long)r * A.ws_stride;
  const long long tab = (long long)ld * ld;
  double* QB = base;
  double* QBI = base + tab;
  double* QM = base + 2 * tab;
  double* QM1 = base + 3 * tab;
  double* DQ = base + 4 * tab;
  double* UQ = base + 5 * tab;
  double* QEXT = base + 6 * tab;
  unsigned char* INFO = reinterpret_cast<unsigned char*>(base + 7 * tab);

  for (int k = tid; k < 64; k += NT) sm.stack[k] = T.stack[k];
  for (int k = tid; k < 128; k += NT) {
    sm.mmH[k] = T.mmH[k]; sm.mmI[k] = T.mmI[k]; sm.mm1n[k] = T.mm1n[k];
    sm.mm23[k] = T.mm23[k]; sm.mmM[k] = T.mmM[k]; sm.mmExt[k] = T.mmExt[k];
  }
  for (int k = tid; k < 1024; k += NT) sm.int11[k] = T.int11[k];
  for (int k = tid; k < 32; k += NT) { sm.d5[k] = T.d5[k]; sm.d3[k] = T.d3[k]; }
  for (int e = tid; e < NPLAN; e += NT) { sm.plan_u[e] = P.u1[e] | (P.u2[e] << 8) | (P.kind[e] << 16); sm.plan_W[e] = P.W[e]; }
  const int npair = P.n_pair;
  for (int p = tid; p < npair; p += NT) {
    const int e = P.pair_e[p], c = P.pair_n[p];
    sm.plan_p[p] = P.u1[e] | (P.u2[e] << 8) | (c << 16);
    sm.plan_pW[p] = f64x2{P.W[e], c > 1 ? P.W[e + 1] : 0.0};
  }
  if (tid == 0) sm.flag = 0;
  __syncthreads();
  const char* seq = A.seqs + (A.rg.off ? (long long)A.rg.off[r] : (long long)r * n);
  for (int k = tid; k < n; k += NT) {
    const int c = enc_nt(seq[k]);
    if (c < 0) sm.flag = 1;
    sm.S[k + 1] = (unsigned char)(c < 0 ? 0 : c);
  }
  // rows read before they are written (all zero): qm1 / U diag 3, D diags 2 and 3
  for (int k = tid; k < ld; k += NT) {
    QM1[3 * ld + k] = 0.0; UQ[3 * ld + k] = 0.0;
    DQ[2 * ld + k] = 0.0; DQ[3 * ld + k] = 0.0;
  }
  __syncthreads();
  if (tid == 0) { sm.S[0] = sm.S[n]; sm.S[n + 1] = sm.S[1]; }
  __syncthreads();
  if (sm.flag) {
    if (tid == 0) { A.status[r] = ST_BAD_CHAR; A.Epf[r] = 0.0; }
    return;
  }
  const double b1 = A.eMLb[1];
  const double sc2 = A.scale[2];
  if (wave == 0 && TURN + 1 < n) pf_build_plist(sm, TURN + 1, n, lane);
  __syncthreads();
  // QM and QM1 (adjacent tables) through one buffer descriptor
  const auto rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)QM, (short)0, (int)(2 * tab * 8), 0x00020000);
  const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)QBI, (short)0, (int)(tab * 8), 0x00020000);

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

    for (int item = wave; item < nI + nK; item += NW) {
      if (item < nI) {
        const int cb = item / HI, h = item - cb * HI;
        const int q = cb * WAVE + lane;
        const bool act = q < pc;
        const int i = sm.plist[par][act ? q : pc - 1];
        const int j = i + d;
        const int t = pair_type(sm.S[i], sm.S[j]);
        const int si1 = sm.S[i + 1], sj1 = sm.S[j - 1];
        const int ij = t * 16 + si1 * 4 + sj1;
        double accI = 0.0;
        const double tau = t > 2 ? T.TermAU : 1.0;
        // Four plan entries per pass, written as stages (entries from LDS, then the table loads, then the arithmetic):
        // one L2 round trip per four entries instead of two per entry.  The order of the sum is fixed by (h, HI).
        for (int e = h; e < segG; e += 4 * HI) {
          int pu[4], info4[4];
          double pw[4], qpq[4];
          bool ok[4];
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int ee = as_vector(min(e + k * HI, segG - 1));
            pu[k] = sm.plan_u[ee]; pw[k] = sm.plan_W[ee];
          }
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int u1 = pu[k] & 255, u2 = (pu[k] >> 8) & 255;
            const int dp = d - 2 - u1 - u2;
            ok[k] = e + k * HI < segG && dp > TURN;
            const int at = ok[k] ? dp * ld + i + 1 + u1 : 0;
            qpq[k] = QB[at];
            info4[k] = INFO[at] & 127;
          }
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int info = info4[k], t2 = info >> 4;
            double f;
            switch (plan_kind(seg, e + k * HI)) {
              case PK_STACK: f = sm.stack[t * 8 + t2]; break;
              case PK_BULGE1: f = sm.stack[t * 8 + t2]; break;
              case PK_BULGEN: f = tau * (t2 > 2 ? T.TermAU : 1.0); break;
              case PK_INT11: f = sm.int11[(t * 8 + t2) * 16 + si1 * 4 + sj1]; break;
              case PK_INT21: f = T.int21[(t * 8 + t2) * 64 + si1 * 16 + ((info >> 2) & 3) * 4 + sj1]; break;
              case PK_INT12: f = T.int21[(t2 * 8 + t) * 64 + ((info >> 2) & 3) * 16 + si1 * 4 + (info & 3)]; break;
              case PK_1XN: f = sm.mm1n[ij] * sm.mm1n[info]; break;
              case PK_INT22:
                f = T.int22[(t * 8 + t2) * 256 + si1 * 64 + (info & 3) * 16 + ((info >> 2) & 3) * 4 + sj1];
                break;
              default: /* PK_INT23 */ f = sm.mm23[ij] * sm.mm23[info]; break;
            }
            if (ok[k]) accI += qpq[k] * f * pw[k];
          }
        }
        double accG = 0.0;
        // (slots of two entries that sit in consecutive cells of one row of QBI: one 16-byte load each)
        for (int p = h; p < npair; p += 4 * HI) {
          int pu[4];
          f64x2 pw[4], v[4];
          bool ok[4];
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int pp = as_vector(min(p + k * HI, npair - 1));
            pu[k] = sm.plan_p[pp]; pw[k] = sm.plan_pW[pp];
          }
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int u1 = pu[k] & 255, u2 = (pu[k] >> 8) & 255;
            const int dp = d - 2 - u1 - u2;
            ok[k] = p + k * HI < npair && dp > TURN;
            v[k] = buf_load_f64x2(rsB, ok[k] ? (dp * ld + i + 1 + u1) * 8 : 0, 0);
          }
#pragma unroll
          for (int k = 0; k < 4; k++) {
            if (ok[k]) accG += v[k].x * pw[k].x;
            if (ok[k] && (pu[k] >> 16) > 1) accG += v[k].y * pw[k].y;
          }
        }
        accI += accG * sm.mmI[ij];
        sm.partI[item * WAVE + lane] = accI;
      } else {
        const int it = item - nI;
        const int b = it / H, h = it - b * H;
        const int i0 = b * WAVE + lane + 1;
        const int i = i0 <= ncell ? i0 : ncell;
        // qm[i,i+tt] at (tt ld + i) 8, qm1[i+tt+1,j] at (tab + (d-tt-1) ld + i+tt+1) 8 bytes from QM; a step of H in tt moves
        // them by +8 H ld and -8 H (ld - 1): buffer loads with one running 32-bit offset each and the strides in SGPRs
        double acc0 = 0.0, acc1 = 0.0;
        int tt = TURN + 1 + h;
        const int stepA = 8 * H * ld, stepC = 8 * H * (ld - 1);
        int vA = (tt * ld + i) * 8;
        int vC = (int)tab * 8 + ((d - tt - 1) * ld + i + tt + 1) * 8;                      // operand of tt (never negative)
        for (; tt + 7 * H <= d - TURN - 2; tt += 8 * H) {                                   // eight split points in flight
          const int vCl = vC - 7 * stepC;                                                   // operand of tt + 7 H: in range here
          double a[8], c[8];
#pragma unroll
          for (int k = 0; k < 8; k++) { a[k] = buf_load_f64(rsQ, vA, k * stepA); c[k] = buf_load_f64(rsQ, vCl, (7 - k) * stepC); }
          vA += 8 * stepA; vC -= 8 * stepC;
#pragma unroll
          for (int k = 0; k < 8; k += 2) { acc0 += a[k] * c[k]; acc1 += a[k + 1] * c[k + 1]; }   // same order as the 4-wide loop
        }
        for (; tt + 3 * H <= d - TURN - 2; tt += 4 * H) {
          const int vCl = vC - 3 * stepC;                                                   // operand of tt + 3 H: in range here
          const double a0 = buf_load_f64(rsQ, vA, 0), c0 = buf_load_f64(rsQ, vCl, 3 * stepC);
          const double a1 = buf_load_f64(rsQ, vA, stepA), c1 = buf_load_f64(rsQ, vCl, 2 * stepC);
          const double a2 = buf_load_f64(rsQ, vA, 2 * stepA), c2 = buf_load_f64(rsQ, vCl, stepC);
          const double a3 = buf_load_f64(rsQ, vA, 3 * stepA), c3 = buf_load_f64(rsQ, vCl, 0);
          vA += 4 * stepA; vC -= 4 * stepC;
          acc0 += a0 * c0; acc1 += a1 * c1; acc0 += a2 * c2; acc1 += a3 * c3;
        }
        for (; tt <= d - TURN - 2; tt += H) {
          acc0 += buf_load_f64(rsQ, vA, 0) * buf_load_f64(rsQ, vC, 0);
          vA += stepA; vC -= stepC;
        }
        sm.partK[it * WAVE + lane] = acc0 + acc1;
      }
    }
    __syncthreads();

    for (int i = tid + 1; i <= ncell; i += NT) {
      const int b = (i - 1) / WAVE, ln = (i - 1) % WAVE;
      double aI = 0.0, aK = 0.0;
      for (int h = 0; h < H; h++) aK += sm.partK[(b * H + h) * WAVE + ln];
      const int j = i + d;
      const int t = pair_type(sm.S[i], sm.S[j]);
      if (t) {
        const int pos = sm.cpos[par][i];
        for (int h = 0; h < HI; h++) aI += sm.partI[((pos >> 6) * HI + h) * WAVE + (pos & 63)];
      }
      const double tau = t > 2 ? T.TermAU : 1.0;
      double qb = 0.0;
      int info = 0;
      if (t) {
        qb = pf_hairpin(sm, A, i, j, t) + aI;
        qb += DQ[(d - 2) * ld + i + 1] * T.MLclosing * T.MLintern * tau *
              sm.mmM[rtype_of(t) * 16 + sm.S[j - 1] * 4 + sm.S[i + 1]] * sc2;
        info = (rtype_of(t) << 4) | (sm.S[j + 1] << 2) | sm.S[i - 1];
      }
      const int at = d * ld + i;
      QB[at] = qb;
      QBI[at] = qb * sm.mmI[info];
      INFO[at] = (unsigned char)info;
      QEXT[j * ld + i] = t ? qb * tau * pf_endstem(sm.mmExt, sm, t, i, j, n) : 0.0;
      double m1 = QM1[(d - 1) * ld + i] * b1;
      if (t) m1 += qb * T.MLintern * tau * pf_endstem(sm.mmM, sm, t, i, j, n);
      const double U = b1 * (QM1[(d - 1) * ld + i + 1] + UQ[(d - 1) * ld + i + 1]);
      QM1[at] = m1;
      UQ[at] = U;
      DQ[at] = aK;
      QM[at] = m1 + aK + U;
    }
    if (wave == NW - 1 && d + 1 < n) pf_build_plist(sm, d + 1, n, lane);     // list of the next diagonal
    __syncthreads();
  }

  // exterior: q5[j] = q5[j-1] scale[1] + sum_i q5[i-1] qb[i,j] expExt(i,j)
  if (wave == 0) {
    const double sc1 = A.scale[1];
    sm.q5[0] = 1.0;                         // every lane stores the same value
    for (int j = 1; j <= n; j++) {
      double s = 0.0;
      for (int i = lane + 1; i <= j - TURN - 1; i += WAVE) s += sm.q5[i - 1] * QEXT[j * ld + i];
      s = wave_sum_f64(s);
      sm.q5[j] = sm.q5[j - 1] * sc1 + s;    // every lane stores the same value
    }
    if (A.q5out)
      for (int j = lane; j <= n; j += WAVE) A.q5out[(long long)r * A.q5_stride + j] = sm.q5[j];
    if (lane == 0) {
      const double Z = sm.q5[n];
      if (!(Z > 0.0) || !(Z < 1.0e300)) {
        A.status[r] = ST_PF_RANGE;
        A.Epf[r] = 0.0;
      } else {
        A.status[r] = ST_OK;
        A.Epf[r] = (-log(Z) - (double)n * log(T.pf_scale)) * T.kT / 1000.0;
      }
    }
  }
}

}  // namespace drna
