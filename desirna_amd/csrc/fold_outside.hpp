// fold_outside.hpp -- McCaskill outside recursion, base-pair probabilities and ensemble defect for one
// sequence per workgroup on gfx950.  Replaces ScoreSeq.get_ensemble_defect(): reference
// utils/energy_scores.py:362-374 (fc.mfe(); fc.exp_params_rescale(mfe); fc.pf() with bpp on;
// fc.ensemble_defect(target)) -- SURVEY row a10, App. A.5.  exp_params_rescale only moves pf_scale,
// which cancels in every probability, so the engine keeps its default per-nucleotide scale.
//
// Runs after pf_kernel (fold_pf.hpp) on the tables that kernel left in the workspace (QB, QM, QM1, QEXT,
// INFO in diagonal-major layout) and its q5[] column.  The outside weights are GATHERED, not scattered:
// a cell pulls from the already finished cells of larger span, so every fp64 sum has one writer and a
// fixed order (bit-reproducible), and the sweep is the mirror image of the inside fill -- diagonals
// d = n-1 ... TURN+1, lane = cell, (64-cell block x chunk) work items, two barriers per diagonal.
//
// Outside weights (derivatives of Z with respect to the inside quantities):
//   A[i,j]   = weight of the split product at (i,j) = Om[i,j] + CL[i-1,j+1]
//              (qm[i,j] contains sum_k qm[i,k-1] qm1[k,j]; so does the multiloop closed by (i-1,j+1))
//   Om[i,j]  = sum_{j''>j} A[i,j''] qm1[j+1,j'']
//   Om1[i,j] = Om[i,j] + sum_{i'<i} A[i',j] qm[i',i-1] + V[i,j] + Om1[i,j+1] b,
//              V[i,j] = sum_{i'<i} Om[i',j] b^(i-i') = b (Om[i-1,j] + V[i-1,j])
//   Ob[i,j]  = Om1[i,j] MLstem(i,j) + q5[i-1] q3[j+1] Ext(i,j) + sum_{(i',j') encloses} Ob[i',j'] IntLoop
//   CL[i,j]  = Ob[i,j] MLclosing MLstem(j,i) scale^2
//   P[i,j]   = Ob[i,j] qb[i,j] / Z
#pragma once
#include "fold_pf.hpp"

namespace drna {

struct OutArgs {
  const PfTables* T = nullptr;
  const Plan* plan = nullptr;
  const double* scale = nullptr;   // pf_scale^-k
  const double* eMLb = nullptr;    // (expMLbase / pf_scale)^k
  const char* seqs = nullptr;      // R x L ASCII
  int L = 0;
  int ld = 0;
  double* ws = nullptr;            // the PF workspace as pf_kernel left it (DQ / UQ are reused for Om+V / Om1)
  long long ws_stride = 0;
  double* wo = nullptr;            // per sequence: OB, OBI, A, CL (ld*ld doubles each) then q5[ld]
  long long wo_stride = 0;
  const short* pt = nullptr;       // pair table of the design target: L+2 shorts, 1-based, 0 = unpaired
  double* edef = nullptr;          // R
  double* bpp = nullptr;           // optional: R x (L+1) x (L+1), P[i,j] at [i*(L+1)+j], i < j, 1-based
  const int32_t* pf_status = nullptr;   // R: status words written by pf_kernel
};

inline long long outside_ws_stride(int ld) { return (long long)4 * ld * ld + ld + 6; }   // doubles per sequence

struct OutSmem : PfSmem {
  double partM[PART_ITEMS * WAVE];
  double q3[MAXN + 3];
};

template <int NT>
__global__ __launch_bounds__(NT) void outside_kernel(OutArgs A) {
  __shared__ OutSmem sm;
  constexpr int NW = NT / WAVE;
  const PfTables& T = *A.T;
  const Plan& P = *A.plan;
  const int r = blockIdx.x;
  const int n = A.L, ld = A.ld;
  const int tid = threadIdx.x, lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(wave_id());
  const int segG = P.seg[PK_GENERIC];
  int seg[PK_NKINDS];                   // first entry of every kind (wave-uniform: the kind of an entry stays scalar)
  for (int k = 0; k < PK_NKINDS; k++) seg[k] = P.seg[k];

  double* base = A.ws + (long long)r * A.ws_stride;
  const long long tab = (long long)ld * ld;
  const double* QB = base;
  const double* QM = base + 2 * tab;
  const double* QM1 = base + 3 * tab;
  double* OMV = base + 4 * tab;        // Om + V
  double* OM1 = base + 5 * tab;
  const double* QEXT = base + 6 * tab;
  const unsigned char* INFO = reinterpret_cast<const unsigned char*>(base + 7 * tab);
  double* ob = A.wo + (long long)r * A.wo_stride;
  double* OB = ob;
  double* OBI = ob + tab;
  double* AT = ob + 2 * tab;
  double* CL = ob + 3 * tab;
  const double* q5g = ob + 4 * tab;

  for (int k = tid; k < 64; k += NT) sm.stack[k] = T.stack[k];
  for (int k = tid; k < 128; k += NT) {
    sm.mmI[k] = T.mmI[k]; sm.mm1n[k] = T.mm1n[k]; sm.mm23[k] = T.mm23[k];
    sm.mmM[k] = T.mmM[k]; sm.mmExt[k] = T.mmExt[k];
  }
  for (int k = tid; k < 1024; k += NT) sm.int11[k] = T.int11[k];
  for (int k = tid; k < 32; k += NT) { sm.d5[k] = T.d5[k]; sm.d3[k] = T.d3[k]; }
  const char* seq = A.seqs + (long long)r * n;
  for (int k = tid; k < n; k += NT) {
    const int c = enc_nt(seq[k]);
    sm.S[k + 1] = (unsigned char)(c < 0 ? 0 : c);
  }
  for (int k = tid; k <= n; k += NT) sm.q5[k] = q5g[k];
  for (int e = tid; e < NPLAN; e += NT) { sm.plan_u[e] = P.u1[e] | (P.u2[e] << 8) | (P.kind[e] << 16); sm.plan_W[e] = P.W[e]; }
  __syncthreads();
  if (tid == 0) { sm.S[0] = sm.S[n]; sm.S[n + 1] = sm.S[1]; }
  __syncthreads();
  if (A.pf_status[r] != ST_OK) {          // bad character / PF out of range: the host reports it
    if (tid == 0) A.edef[r] = 0.0;
    return;
  }
  const double b1 = A.eMLb[1];
  const double sc1 = A.scale[1], sc2 = A.scale[2];

  // q3[i] = Z of the suffix i..n: q3[i] = q3[i+1] scale + sum_j qb[i,j] Ext(i,j) q3[j+1]
  if (wave == 0) {
    sm.q3[n + 1] = 1.0;                   // every lane stores the same value
    for (int i = n; i >= 1; i--) {
      double s = 0.0;
      for (int j = i + TURN + 1 + lane; j <= n; j += WAVE) s += QEXT[j * ld + i] * sm.q3[j + 1];
      s = wave_sum_f64(s);
      sm.q3[i] = sm.q3[i + 1] * sc1 + s;  // every lane stores the same value
    }
  }
  __syncthreads();

  if (wave == 0 && n - 1 >= TURN + 1) pf_build_plist(sm, n - 1, n, lane);
  __syncthreads();

  for (int d = n - 1; d >= TURN + 1; d--) {
    const int ncell = n - d, par = d & 1;
    const int nblk = (ncell + WAVE - 1) / WAVE;
    int H = NW / nblk;
    if (H < 1) H = 1;
    // the interior-loop pull runs over the PAIRABLE cells only (compact list, as in pf_kernel)
    const int pc = sm.pcnt[par];
    const int nblkP = (pc + WAVE - 1) / WAVE;
    int HI = nblkP ? NW / nblkP : 1;
    if (HI < 1) HI = 1;
    const int nI = nblkP * HI, nK = nblk * H;

    for (int item = wave; item < nI + nK; item += NW) {
      if (item < nI) {
        const int cb = item / HI, h = item - cb * HI;
        const int q = cb * WAVE + lane;
        const int i = sm.plist[par][q < pc ? q : pc - 1];
        const int j = i + d;
        const int info = INFO[d * ld + i];
        const int t2 = info >> 4;                 // rtype of (i,j) seen as the inner pair
        double accI = 0.0;
        // four plan entries per pass in explicit stages (entries from LDS, the four table loads, the arithmetic), as in pf_kernel
        for (int e = h; e < segG; e += 4 * HI) {
          int pu[4], io4[4], jo4[4];
          double pw[4], o4[4];
          bool ok[4];
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int ee = as_vector(min(e + k * HI, segG - 1));
            pu[k] = sm.plan_u[ee]; pw[k] = sm.plan_W[ee];
          }
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int u1 = pu[k] & 255, u2 = (pu[k] >> 8) & 255;
            ok[k] = e + k * HI < segG && i - 1 - u1 >= 1 && j + 1 + u2 <= n;
            io4[k] = ok[k] ? i - 1 - u1 : 1; jo4[k] = ok[k] ? j + 1 + u2 : n;
            o4[k] = OB[(jo4[k] - io4[k]) * ld + io4[k]];
          }
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int io = io4[k], jo = jo4[k];
            const int t = pair_type(sm.S[io], sm.S[jo]);
            const int si1 = sm.S[io + 1], sj1 = sm.S[jo - 1];
            const int ij = t * 16 + si1 * 4 + sj1;
            const double tau = t > 2 ? T.TermAU : 1.0;
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
            if (ok[k]) accI += o4[k] * f * pw[k];
          }
        }
        double accG = 0.0;
        for (int e = segG + h; e < NPLAN; e += 8 * HI) {
          int pu[8];
          double pw[8], o8[8];
          bool ok[8];
#pragma unroll
          for (int k = 0; k < 8; k++) {
            const int ee = as_vector(min(e + k * HI, NPLAN - 1));
            pu[k] = sm.plan_u[ee]; pw[k] = sm.plan_W[ee];
          }
#pragma unroll
          for (int k = 0; k < 8; k++) {
            const int u1 = pu[k] & 255, u2 = (pu[k] >> 8) & 255;
            ok[k] = e + k * HI < NPLAN && i - 1 - u1 >= 1 && j + 1 + u2 <= n;
            const int io = ok[k] ? i - 1 - u1 : 1, jo = ok[k] ? j + 1 + u2 : n;
            o8[k] = OBI[(jo - io) * ld + io];
          }
#pragma unroll
          for (int k = 0; k < 8; k++) if (ok[k]) accG += o8[k] * pw[k];
        }
        accI += accG * sm.mmI[info];
        sm.partI[item * WAVE + lane] = accI;
      } else {
        const int it = item - nI;
        const int b = it / H, h = it - b * H;
        const int i0 = b * WAVE + lane + 1;
        const bool act = i0 <= ncell;
        const int i = act ? i0 : ncell;
        const int j = i + d;
        // Om[i,j] = sum_s A[i, j+1+s] qm1[j+1, j+1+s]
        double accA1 = 0.0;
        const int smax = act ? n - j - 1 : -1;
        int s = TURN + 1 + h;
        for (; s + 3 * H <= smax; s += 4 * H) {                 // four terms in flight, summed in the order of the plain loop
          double a[4], c[4];
#pragma unroll
          for (int k = 0; k < 4; k++) { a[k] = AT[(d + 1 + s + k * H) * ld + i]; c[k] = QM1[(s + k * H) * ld + j + 1]; }
#pragma unroll
          for (int k = 0; k < 4; k++) accA1 += a[k] * c[k];
        }
        for (; s <= smax; s += H) accA1 += AT[(d + 1 + s) * ld + i] * QM1[s * ld + j + 1];
        // sum_t A[i-t, j] qm[i-t, i-1]
        double accA2 = 0.0;
        const int tmax = act ? i - 1 : -1;
        int t = TURN + 2 + h;
        for (; t + 3 * H <= tmax; t += 4 * H) {
          double a[4], c[4];
#pragma unroll
          for (int k = 0; k < 4; k++) { a[k] = AT[(d + t + k * H) * ld + i - t - k * H]; c[k] = QM[(t + k * H - 1) * ld + i - t - k * H]; }
#pragma unroll
          for (int k = 0; k < 4; k++) accA2 += a[k] * c[k];
        }
        for (; t <= tmax; t += H) accA2 += AT[(d + t) * ld + i - t] * QM[(t - 1) * ld + i - t];
        sm.partK[it * WAVE + lane] = accA1;
        sm.partM[it * WAVE + lane] = accA2;
      }
    }
    __syncthreads();

    for (int i = tid + 1; i <= ncell; i += NT) {
      const int b = (i - 1) / WAVE, ln = (i - 1) % WAVE;
      double aI = 0.0, a1 = 0.0, a2 = 0.0;
      for (int h = 0; h < H; h++) {
        a1 += sm.partK[(b * H + h) * WAVE + ln];
        a2 += sm.partM[(b * H + h) * WAVE + ln];
      }
      const int j = i + d;
      const int at = d * ld + i;
      const int t = pair_type(sm.S[i], sm.S[j]);
      if (t) {
        const int pos = sm.cpos[par][i];
        for (int h = 0; h < HI; h++) aI += sm.partI[((pos >> 6) * HI + h) * WAVE + (pos & 63)];
      }
      const double tau = t > 2 ? T.TermAU : 1.0;
      const double om = a1;
      const double V = i > 1 ? b1 * OMV[(d + 1) * ld + i - 1] : 0.0;
      double om1 = om + a2 + V;
      if (j < n) om1 += OM1[(d + 1) * ld + i] * b1;
      double o = 0.0, cl = 0.0, obi = 0.0;
      if (t) {
        o = om1 * T.MLintern * tau * pf_endstem(sm.mmM, sm, t, i, j, n);
        o += sm.q5[i - 1] * sm.q3[j + 1] * tau * pf_endstem(sm.mmExt, sm, t, i, j, n);
        o += aI;
        obi = o * sm.mmI[t * 16 + sm.S[i + 1] * 4 + sm.S[j - 1]];
        cl = o * T.MLclosing * T.MLintern * tau * sm.mmM[rtype_of(t) * 16 + sm.S[j - 1] * 4 + sm.S[i + 1]] * sc2;
      }
      OB[at] = o;
      OBI[at] = obi;
      CL[at] = cl;
      AT[at] = om + ((i > 1 && j < n) ? CL[(d + 2) * ld + i - 1] : 0.0);
      OMV[at] = om + V;
      OM1[at] = om1;
    }
    if (wave == NW - 1 && d - 1 >= TURN + 1) pf_build_plist(sm, d - 1, n, lane);     // list of the next diagonal
    __syncthreads();
  }

  // probabilities and ensemble defect (ViennaRNA vrna_ensemble_defect: '(' ')' pairs of the target only)
  const double Z = sm.q5[n];
  const short* pt = A.pt;
  for (int k = tid + 1; k <= n; k += NT) {
    double val;
    const int m = pt[k];
    if (m == 0) {
      double pk = 0.0;
      for (int i = 1; i <= k - TURN - 1; i++) pk += OB[(k - i) * ld + i] * QB[(k - i) * ld + i] / Z;
      for (int j = k + TURN + 1; j <= n; j++) pk += OB[(j - k) * ld + k] * QB[(j - k) * ld + k] / Z;
      val = pk;
    } else {
      const int a = m < k ? m : k, c = m < k ? k : m;
      const double p = c - a > TURN ? OB[(c - a) * ld + a] * QB[(c - a) * ld + a] / Z : 0.0;
      val = 1.0 - p;
    }
    sm.partI[k] = val;
  }
  if (A.bpp) {
    double* B = A.bpp + (long long)r * (n + 1) * (n + 1);
    for (int d = TURN + 1; d < n; d++)
      for (int i = tid + 1; i <= n - d; i += NT)
        B[(long long)i * (n + 1) + i + d] = OB[d * ld + i] * QB[d * ld + i] / Z;
  }
  __syncthreads();
  if (tid == 0) {
    double ed = 0.0;
    for (int k = 1; k <= n; k++) ed += sm.partI[k];
    A.edef[r] = ed / (double)n;
  }
}

}  // namespace drna
