// fold_cofold.hpp -- two interacting strands: co-fold MFE (fill + traceback) and partition function for one
// sequence pair per workgroup on gfx950.  Replaces fc.mfe_dimer() and fc.pf_dimer() of the reference's two-strand
// branch (utils/energy_scores.py:154-158; utils/dimer_multichain_energy.py:36-50, :89-114) -- SURVEY 8(f)-2.
//
// The strands are concatenated (no '&'); cut = length of the first one.  What differs from one strand:
//   * hairpins, the unpaired stretches of interior loops and the backbone of a multiloop must stay inside a strand
//     (a helix INSIDE a multiloop may enclose the nick);
//   * a pair that joins the strands may close the loop whose backbone contains the nick: E_ExtLoop of the pair seen
//     from inside + the best exterior decompositions of [i+1..cut] and [cut+1..j-1] (fcA / fcB; qA3 / qB5 in the PF),
//     which are advanced one entry per diagonal;
//   * such pairs exist at any distance, so the sweep starts at diagonal 1;
//   * dangling neighbours count only inside a strand;
//   * MFE = min(f5[n] + DuplexInit, fcA[1] + fcB[n]); Q = (q5[n] - QA QB) expDuplexInit [/ 2 for two equal strands]
//     + QA QB.
// One wave per cell (the lanes share the interior-loop shapes and the split points, wave minimum / fixed-order wave sum),
// every table in HBM/L2 (diagonal-major like fold_mfe.hpp); the traceback is wave 0's, candidates in ViennaRNA's order (interior loops, nick, multiloop).
#pragma once
#include "fold_mfe.hpp"
#include "fold_pf.hpp"

namespace drna {

struct CoArgs {
  const MfeTables* T = nullptr;
  const PfTables* F = nullptr;
  const Plan* plan = nullptr;
  const int* hp_len = nullptr;
  const double* hp_w = nullptr;
  const double* scale = nullptr;
  const double* eMLb = nullptr;
  const char* seqs = nullptr;     // R x L ASCII, both strands, no '&'
  int L = 0, cut = 0, ld = 0;
  int DuplexInit = 0;
  double eDuplexInit = 1.0;
  int32_t* wsm = nullptr;         // MFE: Wc, FML, EXT (ld*ld int32 each)
  long long wsm_stride = 0;
  double* wsp = nullptr;          // PF: QB, QM, QM1 (ld*ld doubles each)
  long long wsp_stride = 0;
  int32_t* Emfe = nullptr;        // R
  char* ss = nullptr;             // R x L
  double* F4 = nullptr;           // R x 4: FA, FB, FcAB, FAB (kcal/mol)
  int32_t* status = nullptr;      // R: MFE kernel
  int32_t* status_pf = nullptr;   // R: PF kernel
};

__device__ __forceinline__ bool co_same(int a, int b, int cut) { return !(a <= cut && b > cut); }   // a <= b

struct CoMfeSmem : MfeSmemCore<MAXN> {
  int fcA[MAXN + 3], fcB[MAXN + 3];
};

// E_ExtLoop / E_MLstem neighbour rule with explicit presence flags
template <class SM>
__device__ __forceinline__ int co_endstem(const int* mm, const SM& sm, int t, bool h5, int s5, bool h3, int s3) {
  if (h5 && h3) return mm[t * 16 + s5 * 4 + s3];
  if (h5) return sm.d5[t * 4 + s5];
  if (h3) return sm.d3[t * 4 + s3];
  return 0;
}

template <int NT>
__global__ __launch_bounds__(NT) void cofold_mfe_kernel(CoArgs A) {
  __shared__ CoMfeSmem sm;
  const MfeTables& T = *A.T;
  const Plan& P = *A.plan;
  const int r = blockIdx.x;
  const int n = A.L, cut = A.cut, ld = A.ld;
  const int tid = threadIdx.x, lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(wave_id());
  const int INF = INF_DEV, HALF = INF_DEV / 2;
  int32_t* base = A.wsm + (long long)r * A.wsm_stride;
  const long long tab = (long long)ld * ld;
  int32_t* Wc = base;
  int32_t* FML = base + tab;
  int32_t* EXT = base + 2 * tab;

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
    sm.ssw[k] = '.';
  }
  for (int k = tid; k < ld; k += NT) FML[k] = INF;           // row 0: empty segments
  for (int k = tid; k <= n + 2; k += NT) { sm.fcA[k] = 0; sm.fcB[k] = 0; }
  __syncthreads();
  if (tid == 0) { sm.S[0] = sm.S[n]; sm.S[n + 1] = sm.S[1]; }
  __syncthreads();
  if (sm.flag) {
    if (tid == 0) { A.status[r] = ST_BAD_CHAR; A.Emfe[r] = 0; }
    for (int k = tid; k < n; k += NT) A.ss[(long long)r * n + k] = '.';
    return;
  }

  for (int d = 1; d < n; d++) {
    const int ncell = n - d;
    // one wave per cell: the lanes share the 496 interior-loop shapes and the split points, then fold with a wave minimum
    for (int i = wave + 1; i <= ncell; i += NT / WAVE) {
      const int j = i + d;
      const bool same = co_same(i, j, cut);
      const int t = (d > TURN || !same) ? pair_type(sm.S[i], sm.S[j]) : 0;
      const int tau = t > 2 ? T.TermAU : 0;
      const bool adj_i = co_same(i, i + 1, cut), adj_j = co_same(j - 1, j, cut);
      int c = INF, info = 0;
      if (t) {
        const int si1 = sm.S[i + 1], sj1 = sm.S[j - 1];
        if (same) c = mfe_hairpin_e(sm, T, A.hp_len[d - 1], i, j, t);
        else c = tau + co_endstem(sm.mmExt, sm, rtype_of(t), adj_j, sj1, adj_i, si1) + sm.fcA[i + 1] + sm.fcB[j - 1];
        int m = INF;
        for (int e = lane; e < NPLAN; e += WAVE) {
          const int u1 = P.u1[e], u2 = P.u2[e];
          const int dp = d - 2 - u1 - u2;
          if (dp < 1) continue;
          const int p = i + 1 + u1, q = j - 1 - u2;
          if (!co_same(i, p, cut) || !co_same(q, j, cut)) continue;
          const int w = Wc[dp * ld + p];
          const int cpq = w >> 8;
          if (cpq >= HALF) continue;
          m = min(m, cpq + mfe_intloop(sm, T, u1, u2, t, si1, sj1, w & 127));
        }
        int dec = INF;
        if (adj_i && adj_j) {
          for (int u = i + 2 + lane; u <= j - 2; u += WAVE) {
            if (u == cut) continue;                                   // u, u+1 must be neighbours
            const int a = FML[(u - i - 1) * ld + i + 1], b = FML[(j - u - 2) * ld + u + 1];
            if (a < HALF && b < HALF) dec = min(dec, a + b);
          }
        }
        m = wave_min_i32(m);
        dec = wave_min_i32(dec);
        c = min(c, m);
        if (dec < HALF) c = min(c, dec + T.MLclosing + T.MLintern + tau + sm.mmM[rtype_of(t) * 16 + sj1 * 4 + si1]);
        if (c >= HALF) c = INF;
        info = (rtype_of(t) << 4) | (sm.S[j + 1] << 2) | sm.S[i - 1];
      }
      const bool h5 = i > 1 && co_same(i - 1, i, cut), h3 = j < n && co_same(j, j + 1, cut);
      int f = INF;
      for (int u = i + 1 + lane; u <= j - 2; u += WAVE) {
        if (u == cut) continue;
        const int a = FML[(u - i) * ld + i], b = FML[(j - u - 1) * ld + u + 1];
        if (a < HALF && b < HALF) f = min(f, a + b);
      }
      f = wave_min_i32(f);
      if (adj_i) { const int fa = FML[(d - 1) * ld + i + 1]; if (fa < HALF) f = min(f, fa + T.MLbase); }
      if (adj_j) { const int fb = FML[(d - 1) * ld + i]; if (fb < HALF) f = min(f, fb + T.MLbase); }
      if (c < INF) f = min(f, c + T.MLintern + tau + co_endstem(sm.mmM, sm, t, h5, sm.S[i - 1], h3, sm.S[j + 1]));
      if (lane == 0) {
        Wc[d * ld + i] = c * 256 + info;
        EXT[j * ld + i] = c < INF ? c + tau + co_endstem(sm.mmExt, sm, t, h5, sm.S[i - 1], h3, sm.S[j + 1]) : INF;
        FML[d * ld + i] = f;
      }
    }
    __syncthreads();
    // exterior decompositions next to the nick: fcA[cut - d] of [cut-d .. cut], fcB[cut + 1 + d] of [cut+1 .. cut+1+d]
    if (wave == 0 && cut - d >= 1) {
      const int x = cut - d;
      int m = INF;
      for (int k = x + 1 + lane; k <= cut; k += WAVE) {
        const int e = EXT[k * ld + x];
        if (e < HALF) m = min(m, e + sm.fcA[k + 1]);
      }
      m = wave_min_i32(m);
      sm.fcA[x] = min(sm.fcA[x + 1], m);          // every lane stores the same value
    }
    if (wave == (NT > WAVE ? 1 : 0) && cut > 0 && cut + 1 + d <= n) {
      const int y = cut + 1 + d;
      int m = INF;
      for (int k = cut + 1 + lane; k < y; k += WAVE) {
        const int e = EXT[y * ld + k];
        if (e < HALF) m = min(m, sm.fcB[k - 1] + e);
      }
      m = wave_min_i32(m);
      sm.fcB[y] = min(sm.fcB[y - 1], m);
    }
    __syncthreads();
  }

  if (wave != 0) return;
  // ---- exterior loop over the concatenation
  sm.f5[0] = 0;
  for (int j = 1; j <= n; j++) {
    int m = INF;
    for (int i = lane + 1; i < j; i += WAVE) {
      const int x = EXT[j * ld + i];
      if (x < HALF) m = min(m, sm.f5[i - 1] + x);
    }
    m = wave_min_i32(m);
    const int prev = sm.f5[j - 1];
    sm.f5[j] = prev < m ? prev : m;
  }
  const int e_dimer = sm.f5[n] + A.DuplexInit, e_mono = sm.fcA[1] + sm.fcB[n];
  const bool dimer = cut > 0 ? e_dimer < e_mono : true;
  if (lane == 0) A.Emfe[r] = cut > 0 ? (dimer ? e_dimer : e_mono) : sm.f5[n];

  // ---- traceback (sectors: 0 = f5[1..j], 1 = fML[i..j], 2 = pair, 3 = fcA[i..cut], 4 = fcB[cut+1..j])
  int sp = 0;
  bool ok = true;
  if (dimer) { sm.sec_i[0] = 1; sm.sec_j[0] = (short)n; sm.sec_ml[0] = 0; sp = 1; }
  else {
    sm.sec_i[0] = 1; sm.sec_j[0] = (short)cut; sm.sec_ml[0] = 3;
    sm.sec_i[1] = (short)(cut + 1); sm.sec_j[1] = (short)n; sm.sec_ml[1] = 4; sp = 2;
  }
  while (sp > 0 && ok) {
    sp--;
    int i = sm.sec_i[sp], j = sm.sec_j[sp];
    const int ml = sm.sec_ml[sp];
    bool have_pair = false;
    if (ml == 0) {
      while (j > 0 && sm.f5[j] == sm.f5[j - 1]) j--;
      if (j < 2) continue;
      int u = -1;
      for (int b0 = j - 1; b0 >= 1 && u < 0; b0 -= WAVE) {
        const int x = b0 - lane;
        bool hit = false;
        if (x >= 1) { const int e = EXT[j * ld + x]; hit = e < HALF && sm.f5[j] == e + sm.f5[x - 1]; }
        const int fl = first_lane(__ballot(hit));
        if (fl >= 0) u = b0 - fl;
      }
      if (u < 0) { ok = false; break; }
      sm.sec_i[sp] = 1; sm.sec_j[sp] = (short)(u - 1); sm.sec_ml[sp] = 0; sp++;
      i = u; have_pair = true;
    } else if (ml == 3) {
      while (i <= cut && sm.fcA[i] == sm.fcA[i + 1]) i++;
      if (i > cut) continue;
      int k = -1;
      for (int b0 = i + 1; b0 <= cut && k < 0; b0 += WAVE) {
        const int x = b0 + lane;
        bool hit = false;
        if (x <= cut) { const int e = EXT[x * ld + i]; hit = e < HALF && sm.fcA[i] == e + sm.fcA[x + 1]; }
        const int fl = first_lane(__ballot(hit));
        if (fl >= 0) k = b0 + fl;
      }
      if (k < 0) { ok = false; break; }
      sm.sec_i[sp] = (short)(k + 1); sm.sec_j[sp] = (short)cut; sm.sec_ml[sp] = 3; sp++;
      j = k; have_pair = true;
    } else if (ml == 4) {
      while (j > cut && sm.fcB[j] == sm.fcB[j - 1]) j--;
      if (j <= cut) continue;
      int k = -1;
      for (int b0 = j - 1; b0 > cut && k < 0; b0 -= WAVE) {
        const int x = b0 - lane;
        bool hit = false;
        if (x > cut) { const int e = EXT[j * ld + x]; hit = e < HALF && sm.fcB[j] == e + sm.fcB[x - 1]; }
        const int fl = first_lane(__ballot(hit));
        if (fl >= 0) k = b0 - fl;
      }
      if (k < 0) { ok = false; break; }
      sm.sec_i[sp] = (short)(cut + 1); sm.sec_j[sp] = (short)(k - 1); sm.sec_ml[sp] = 4; sp++;
      i = k; have_pair = true;
    } else if (ml == 1) {
      for (;;) {                 // strip unpaired 3' then 5' ends (only inside a strand)
        if (j > i && co_same(j - 1, j, cut)) {
          const int a = FML[(j - i) * ld + i], b = FML[(j - 1 - i) * ld + i];
          if (b < HALF && a == b + T.MLbase) { j--; continue; }
        }
        break;
      }
      for (;;) {
        if (i < j && co_same(i, i + 1, cut)) {
          const int a = FML[(j - i) * ld + i], b = FML[(j - i - 1) * ld + i + 1];
          if (b < HALF && a == b + T.MLbase) { i++; continue; }
        }
        break;
      }
      const int d = j - i;
      const int fij = FML[d * ld + i];
      const int cij = Wc[d * ld + i] >> 8;
      const bool same = co_same(i, j, cut);
      const int t = (d > TURN || !same) ? pair_type(sm.S[i], sm.S[j]) : 0;
      const bool h5 = i > 1 && co_same(i - 1, i, cut), h3 = j < n && co_same(j, j + 1, cut);
      if (t && cij < HALF &&
          fij == cij + T.MLintern + (t > 2 ? T.TermAU : 0) + co_endstem(sm.mmM, sm, t, h5, sm.S[i - 1], h3, sm.S[j + 1])) {
        have_pair = true;
      } else {
        int u = -1;
        for (int b0 = i + 1; b0 <= j - 2 && u < 0; b0 += WAVE) {
          const int x = b0 + lane;
          bool hit = false;
          if (x <= j - 2 && x != cut) {
            const int a = FML[(x - i) * ld + i], b = FML[(j - x - 1) * ld + x + 1];
            hit = a < HALF && b < HALF && fij == a + b;
          }
          const int fl = first_lane(__ballot(hit));
          if (fl >= 0) u = b0 + fl;
        }
        if (u < 0) { ok = false; break; }
        sm.sec_i[sp] = (short)i; sm.sec_j[sp] = (short)u; sm.sec_ml[sp] = 1; sp++;
        sm.sec_i[sp] = (short)(u + 1); sm.sec_j[sp] = (short)j; sm.sec_ml[sp] = 1; sp++;
      }
    } else {
      have_pair = true;
    }
    while (have_pair) {
      if (lane == 0) { sm.ssw[i - 1] = '('; sm.ssw[j - 1] = ')'; }
      const int d = j - i;
      const bool same = co_same(i, j, cut);
      const int t = pair_type(sm.S[i], sm.S[j]);
      const int tau = t > 2 ? T.TermAU : 0;
      const int cij = Wc[d * ld + i] >> 8;
      const int si1 = sm.S[i + 1], sj1 = sm.S[j - 1];
      const bool adj_i = co_same(i, i + 1, cut), adj_j = co_same(j - 1, j, cut);
      if (same && cij == mfe_hairpin_e(sm, T, A.hp_len[d - 1], i, j, t)) break;
      // interior loops: p ascending, q descending
      int found = -1;
      for (int b0 = 0; b0 < NPLAN && found < 0; b0 += WAVE) {
        const int k = b0 + lane;
        bool hit = false;
        if (k < NPLAN) {
          const int u1 = P.tb_u1[k], u2 = P.tb_u2[k];
          const int dp = d - 2 - u1 - u2;
          if (dp >= 1 && co_same(i, i + 1 + u1, cut) && co_same(j - 1 - u2, j, cut)) {
            const int w = Wc[dp * ld + i + 1 + u1];
            const int cpq = w >> 8;
            if (cpq < HALF) hit = cij == cpq + mfe_intloop(sm, T, u1, u2, t, si1, sj1, w & 127);
          }
        }
        const int fl = first_lane(__ballot(hit));
        if (fl >= 0) found = b0 + fl;
      }
      if (found >= 0) {
        i = i + 1 + P.tb_u1[found];
        j = j - 1 - P.tb_u2[found];
        continue;
      }
      // the loop with the nick
      if (!same && cij == tau + co_endstem(sm.mmExt, sm, rtype_of(t), adj_j, sj1, adj_i, si1) + sm.fcA[i + 1] + sm.fcB[j - 1]) {
        sm.sec_i[sp] = (short)(i + 1); sm.sec_j[sp] = (short)cut; sm.sec_ml[sp] = 3; sp++;
        sm.sec_i[sp] = (short)(cut + 1); sm.sec_j[sp] = (short)(j - 1); sm.sec_ml[sp] = 4; sp++;
        break;
      }
      // multiloop
      if (!(adj_i && adj_j)) { ok = false; break; }
      const int e = cij - T.MLclosing - T.MLintern - tau - sm.mmM[rtype_of(t) * 16 + sj1 * 4 + si1];
      int u = -1;
      for (int b0 = i + 2; b0 <= j - 2 && u < 0; b0 += WAVE) {
        const int x = b0 + lane;
        bool hit = false;
        if (x <= j - 2 && x != cut) {
          const int a = FML[(x - i - 1) * ld + i + 1], b = FML[(j - x - 2) * ld + x + 1];
          hit = a < HALF && b < HALF && e == a + b;
        }
        const int fl = first_lane(__ballot(hit));
        if (fl >= 0) u = b0 + fl;
      }
      if (u < 0) { ok = false; break; }
      sm.sec_i[sp] = (short)(i + 1); sm.sec_j[sp] = (short)u; sm.sec_ml[sp] = 1; sp++;
      sm.sec_i[sp] = (short)(u + 1); sm.sec_j[sp] = (short)(j - 1); sm.sec_ml[sp] = 1; sp++;
      break;
    }
  }
  // lane 0 wrote the brackets: the wave meets before the other lanes read them (the hardware runs the wave in lockstep;
  // the CPU emulation of the kernels does not)
  if (__ballot(ok) == 0ull) ok = false;
  for (int k = lane; k < n; k += WAVE) A.ss[(long long)r * n + k] = sm.ssw[k];
  if (lane == 0) A.status[r] = ok ? ST_OK : ST_TRACEBACK;
}

// ---------------------------------------------------------------- partition function

struct CoPfSmem : PfSmem {
  double qA3[MAXN + 3], qB5[MAXN + 3];
};

// Boltzmann factor of the interior loop (u1,u2) between a pair of type t and the inner pair given by its info byte,
// scale[u1+u2+2] included
__device__ __forceinline__ double co_pf_intloop(const PfSmem& sm, const PfTables& T, const double* scale, int u1, int u2, int t,
                                                int si1, int sj1, int info) {
  const int t2 = info >> 4, sq1 = (info >> 2) & 3, sp1 = info & 3;
  const int nl = u1 > u2 ? u1 : u2, ns = u1 > u2 ? u2 : u1;
  const double sc = scale[u1 + u2 + 2];
  if (nl == 0) return sm.stack[t * 8 + t2] * sc;
  if (ns == 0) {
    const double e = T.bulge[nl] * sc;
    if (nl == 1) return e * sm.stack[t * 8 + t2];
    return e * (t > 2 ? T.TermAU : 1.0) * (t2 > 2 ? T.TermAU : 1.0);
  }
  if (ns == 1) {
    if (nl == 1) return sm.int11[(t * 8 + t2) * 16 + si1 * 4 + sj1] * sc;
    if (nl == 2)
      return ((u1 == 1) ? T.int21[(t * 8 + t2) * 64 + si1 * 16 + sq1 * 4 + sj1]
                        : T.int21[(t2 * 8 + t) * 64 + sq1 * 16 + si1 * 4 + sp1]) * sc;
    return T.interior[nl + 1] * T.eninio[nl - ns] * sm.mm1n[t * 16 + si1 * 4 + sj1] * sm.mm1n[info] * sc;
  }
  if (ns == 2) {
    if (nl == 2) return T.int22[(t * 8 + t2) * 256 + si1 * 64 + sp1 * 16 + sq1 * 4 + sj1] * sc;
    if (nl == 3) return T.interior[5] * T.eninio[1] * sm.mm23[t * 16 + si1 * 4 + sj1] * sm.mm23[info] * sc;
  }
  return T.interior[nl + ns] * T.eninio[nl - ns] * sm.mmI[t * 16 + si1 * 4 + sj1] * sm.mmI[info] * sc;
}

__device__ __forceinline__ double co_pf_endstem(const double* mm, const PfSmem& sm, int t, bool h5, int s5, bool h3, int s3) {
  if (h5 && h3) return mm[t * 16 + s5 * 4 + s3];
  if (h5) return sm.d5[t * 4 + s5];
  if (h3) return sm.d3[t * 4 + s3];
  return 1.0;
}

template <int NT>
__global__ __launch_bounds__(NT) void cofold_pf_kernel(CoArgs A) {
  __shared__ CoPfSmem sm;
  const PfTables& T = *A.F;
  const Plan& P = *A.plan;
  const int r = blockIdx.x;
  const int n = A.L, cut = A.cut, ld = A.ld;
  const int tid = threadIdx.x, lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(wave_id());
  double* base = A.wsp + (long long)r * A.wsp_stride;
  const long long tab = (long long)ld * ld;
  double* QB = base;
  double* QM = base + tab;
  double* QM1 = base + 2 * tab;
  unsigned char* INFO = reinterpret_cast<unsigned char*>(base + 3 * tab);
  int32_t* status = A.status_pf;

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
  for (int k = tid; k < ld; k += NT) { QM[k] = 0.0; QM1[k] = 0.0; QB[k] = 0.0; INFO[k] = 0; }   // row 0
  for (int k = tid; k <= n + 2; k += NT) { sm.qA3[k] = 1.0; sm.qB5[k] = 1.0; }
  __syncthreads();
  if (tid == 0) {
    sm.S[0] = sm.S[n]; sm.S[n + 1] = sm.S[1];
    // one-nucleotide segments next to the nick (the sweep advances these arrays from diagonal 1 on)
    if (cut >= 1) sm.qA3[cut] = A.scale[1];
    if (cut >= 1 && cut + 1 <= n) sm.qB5[cut + 1] = A.scale[1];
  }
  __syncthreads();
  if (sm.flag) {
    if (tid == 0) { status[r] = ST_BAD_CHAR; for (int k = 0; k < 4; k++) A.F4[r * 4 + k] = 0.0; }
    return;
  }
  PfArgs H;                        // for pf_hairpin()
  H.T = A.F; H.plan = A.plan; H.hp_w = A.hp_w; H.scale = A.scale; H.eMLb = A.eMLb; H.seqs = A.seqs; H.L = n; H.ld = ld;
  H.ws = nullptr; H.ws_stride = 0; H.Epf = nullptr; H.status = nullptr;
  const double b1 = A.eMLb[1], sc1 = A.scale[1], sc2 = A.scale[2];

  for (int d = 1; d < n; d++) {
    const int ncell = n - d;
    // one wave per cell: the lanes share the interior-loop shapes and the split points; fixed-order wave sums
    for (int i = wave + 1; i <= ncell; i += NT / WAVE) {
      const int j = i + d;
      const bool same = co_same(i, j, cut);
      const int t = (d > TURN || !same) ? pair_type(sm.S[i], sm.S[j]) : 0;
      const double tau = t > 2 ? T.TermAU : 1.0;
      const bool adj_i = co_same(i, i + 1, cut), adj_j = co_same(j - 1, j, cut);
      double qb = 0.0;
      int info = 0;
      if (t) {
        const int si1 = sm.S[i + 1], sj1 = sm.S[j - 1];
        if (same) qb = pf_hairpin(sm, H, i, j, t);
        else qb = sm.qA3[i + 1] * sm.qB5[j - 1] * sc2 * tau * co_pf_endstem(sm.mmExt, sm, rtype_of(t), adj_j, sj1, adj_i, si1);
        double acc = 0.0;
        for (int e = lane; e < NPLAN; e += WAVE) {
          const int u1 = P.tb_u1[e], u2 = P.tb_u2[e];
          const int dp = d - 2 - u1 - u2;
          if (dp < 1) continue;
          const int p = i + 1 + u1, q = j - 1 - u2;
          if (!co_same(i, p, cut) || !co_same(q, j, cut)) continue;
          const int fi = INFO[dp * ld + p];
          if (!fi) continue;
          acc += QB[dp * ld + p] * co_pf_intloop(sm, T, A.scale, u1, u2, t, si1, sj1, fi);
        }
        double tmp = 0.0;
        if (adj_i && adj_j) {
          for (int k = i + 3 + lane; k <= j - 2; k += WAVE) {
            if (k - 1 == cut) continue;                               // k-1, k must be neighbours
            tmp += QM[(k - i - 2) * ld + i + 1] * QM1[(j - 1 - k) * ld + k];
          }
        }
        acc = wave_sum_f64(acc);
        tmp = wave_sum_f64(tmp);
        qb += acc + tmp * T.MLclosing * T.MLintern * tau * sm.mmM[rtype_of(t) * 16 + sj1 * 4 + si1] * sc2;
        info = (rtype_of(t) << 4) | (sm.S[j + 1] << 2) | sm.S[i - 1];
      }
      const bool h5 = i > 1 && co_same(i - 1, i, cut), h3 = j < n && co_same(j, j + 1, cut);
      double m1 = adj_j ? QM1[(d - 1) * ld + i] * b1 : 0.0;
      if (t) m1 += qb * T.MLintern * tau * co_pf_endstem(sm.mmM, sm, t, h5, sm.S[i - 1], h3, sm.S[j + 1]);
      double m = 0.0;
      for (int k = i + 1 + lane; k <= j - 1; k += WAVE) {
        double left = (k - 1 != cut) ? QM[(k - 1 - i) * ld + i] : 0.0;
        if (co_same(i, k, cut)) left += A.eMLb[k - i];
        m += left * QM1[(j - k) * ld + k];
      }
      m = m1 + wave_sum_f64(m);
      if (lane == 0) {
        QB[d * ld + i] = qb;
        INFO[d * ld + i] = (unsigned char)info;
        QM1[d * ld + i] = m1;
        QM[d * ld + i] = m;
      }
    }
    __syncthreads();
    if (wave == 0 && cut - d >= 1) {
      const int x = cut - d;
      double s = 0.0;
      for (int k = x + 1 + lane; k <= cut; k += WAVE) {
        const int fi = INFO[(k - x) * ld + x];
        if (!fi) continue;
        const int t = rtype_of(fi >> 4);
        s += QB[(k - x) * ld + x] * (t > 2 ? T.TermAU : 1.0) *
             co_pf_endstem(sm.mmExt, sm, t, x > 1, sm.S[x - 1], k < cut, sm.S[k + 1]) * sm.qA3[k + 1];
      }
      s = wave_sum_f64(s);
      sm.qA3[x] = sm.qA3[x + 1] * sc1 + s;
    }
    if (wave == (NT > WAVE ? 1 : 0) && cut > 0 && cut + 1 + d <= n) {
      const int y = cut + 1 + d;
      double s = 0.0;
      for (int k = cut + 1 + lane; k < y; k += WAVE) {
        const int fi = INFO[(y - k) * ld + k];
        if (!fi) continue;
        const int t = rtype_of(fi >> 4);
        s += sm.qB5[k - 1] * QB[(y - k) * ld + k] * (t > 2 ? T.TermAU : 1.0) *
             co_pf_endstem(sm.mmExt, sm, t, k > cut + 1, sm.S[k - 1], y < n, sm.S[y + 1]);
      }
      s = wave_sum_f64(s);
      sm.qB5[y] = sm.qB5[y - 1] * sc1 + s;
    }
    __syncthreads();
  }
  if (wave != 0) return;
  sm.q5[0] = 1.0;
  for (int j = 1; j <= n; j++) {
    double s = 0.0;
    for (int i = lane + 1; i < j; i += WAVE) {
      const int fi = INFO[(j - i) * ld + i];
      if (!fi) continue;
      const int t = rtype_of(fi >> 4);
      const bool h5 = i > 1 && co_same(i - 1, i, cut), h3 = j < n && co_same(j, j + 1, cut);
      s += sm.q5[i - 1] * QB[(j - i) * ld + i] * (t > 2 ? T.TermAU : 1.0) * co_pf_endstem(sm.mmExt, sm, t, h5, sm.S[i - 1], h3, sm.S[j + 1]);
    }
    s = wave_sum_f64(s);
    sm.q5[j] = sm.q5[j - 1] * sc1 + s;
  }
  if (lane == 0) {
    const double kT = T.kT / 1000.0, lsc = log(T.pf_scale);
    const double Q0 = sm.q5[n];
    double* out = A.F4 + (long long)r * 4;
    if (!(Q0 > 0.0) || !(Q0 < 1.0e300)) {
      status[r] = ST_PF_RANGE;
      for (int k = 0; k < 4; k++) out[k] = 0.0;
    } else if (cut <= 0) {
      status[r] = ST_OK;
      out[0] = out[3] = -kT * (log(Q0) + n * lsc); out[1] = 0.0; out[2] = 999.0;
    } else {
      // strand partition functions: scale^len when a strand cannot fold at all
      const double QA = sm.qA3[1], QB_ = sm.qB5[n];
      double QAB = (Q0 - QA * QB_) * A.eDuplexInit;
      bool sym = n == 2 * cut;
      for (int k = 1; sym && k <= cut; k++) sym = sm.S[k] == sm.S[cut + k];
      if (sym) QAB *= 0.5;                                           // rotational symmetry of a homodimer
      status[r] = ST_OK;
      out[0] = -kT * (log(QA) + cut * lsc);
      out[1] = -kT * (log(QB_) + (n - cut) * lsc);
      out[2] = QAB > 1e-17 ? -kT * (log(QAB) + n * lsc) : 999.0;
      out[3] = -kT * (log(QA * QB_ + QAB) + n * lsc);
    }
  }
}

}  // namespace drna
