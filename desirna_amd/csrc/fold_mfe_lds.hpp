// fold_mfe_lds.hpp -- LDS-resident Zuker MFE fill for n <= MFE_FAST_NMAX: the production path for the
// headline workload (L=200).  Same recursions, same outputs, same traceback as fold_mfe.hpp
// (reference utils/energy_scores.py:151,354); what differs is where operands live
// and how the interior loops are enumerated:
//
//  * fML is kept whole in LDS as a compact diagonal-major triangle (77 KB at n=200), c only as a
//    32-diagonal ring (interior loops never look further back than MAXLOOP+2 diagonals): every operand
//    of both inner loops is an LDS read that is unit-stride across the wave.  c is also streamed to HBM
//    once per cell (coalesced) for the traceback.
//  * lanes own TOWERS of nested cells {(i+k, j-k)}: lane <-> i + (d >> 1).  The set of generic interior
//    loops (u1,u2 >= 2) of (i,j) with inner pair on diagonal d' contains the set of (i+1,j-1) for the
//    same d' with the same asymmetry penalties, plus two new boundary candidates (u1 = 2 or u2 = 2).
//    So each tower carries, IN REGISTERS, one running minimum per live inner diagonal d' (27 of them,
//    spread over the waves pinned to the tower block) and a diagonal step costs 2 LDS reads per entry
//    instead of enumerating all ~375 generic (u1,u2) candidates (the O(n^2 L) scheme of Lyngso et al.).
//    Stack, bulges, 1x1, 2x1, 1xn, 2x2 and 2x3 loops (121 candidates) are still enumerated from the
//    host-built plan with wave-uniform size terms.
//  * f5 is advanced one column per diagonal step by a spare wave, so no serial tail remains.
#pragma once
#include "fold_mfe.hpp"

namespace drna {

constexpr int MFE_FAST_NMAX = 200;
constexpr int GSLOTS = 10;         // register-resident running minima per lane and parity (28 residues over >= 3 waves)
constexpr int GRES = 28;           // residues of d' (27 live entries + 1 spare)

template <int NT>
struct MfeFastSmem : MfeSmemCore<MFE_FAST_NMAX> {
  static constexpr int NW = NT / WAVE;
  static constexpr int RS = MFE_FAST_NMAX + 2;                                   // ring row pitch
  static constexpr int TRI = (MFE_FAST_NMAX - 4) * (MFE_FAST_NMAX - 3) / 2;      // rows 4..n-1
  static constexpr int NSLOT = 4 * WAVE;                                         // tower slots (n <= 256)
  int fml[TRI + 8];
  int wring[32 * RS];            // (c + TermAU(inner type)) * 256 + info   of the last 32 diagonals
  int ciring[32 * RS];           // c + mismatchI(inner side)               of the last 32 diagonals
  int dml[4 * RS];               // decomposition minima of the last 4 diagonals
  int hpl[MFE_FAST_NMAX + 2];    // hairpin size term by loop size
  int rowoff[MFE_FAST_NMAX + 2]; // offset of row d in fml[]
  int accG[2][NSLOT], accI[2][NSLOT], accK[2][NSLOT];   // per-tower minima, double-buffered by diagonal parity (ds_min)
  // tables with the inner pair's terminal-AU term taken out (it is folded into wring)
  int stackp[64], int11p[1024], mm1np[128], mm23p[128];
};

// compact triangle: row d (4 <= d <= n-1) holds cells i = 1..n-d
__device__ __forceinline__ int fml_off(int d, int n) { return (d - 4) * n - (d * (d - 1) / 2 - 6); }

template <int NT>
struct FmlLds {
  const MfeFastSmem<NT>* sm;
  int n;
  __device__ __forceinline__ int operator()(int d, int i) const {
    return d < TURN + 1 ? INF_DEV : sm->fml[fml_off(d, n) + i - 1];
  }
};

// one diagonal step of the register-resident generic-interior minima of a tower; returns the generic
// candidate (without the outer mismatch term) for the cell at column i on diagonal d.
// xs[r]: (d - 6 - residue) mod 28 of slot r, maintained incrementally by the caller (no division).
// v_int / v_asym: lane tables (lane s -> interior[s], lane a -> min(max_ninio, a * ninio)).
template <int NT>
__device__ __forceinline__ int mfe_tower_step(const MfeFastSmem<NT>& sm, int (&G)[GSLOTS], const int (&xs)[GSLOTS], int d,
                                              int i, int v_int, int v_asym) {
  constexpr int RS = MfeFastSmem<NT>::RS;
  int acc = INF_DEV;
#pragma unroll
  for (int r = 0; r < GSLOTS; r++) {
    const int x = xs[r];
    if (x > 26) continue;                 // unused slot, or the one residue with no live entry on this diagonal
    const int s = x + 4, dp = d - 6 - x;  // loop size and inner diagonal of this entry
    if (dp <= TURN) { G[r] = INF_DEV; continue; }
    const int* row = sm.ciring + (dp & 31) * RS;
    const int as = lane_table(v_asym, s - 4);
    if (s == 4) {
      G[r] = row[i + 3];
    } else if (s == 5) {
      G[r] = min(row[i + 3], row[i + 4]) + as;
    } else {
      const int a = row[i + 3], b = row[i + s - 1];
      G[r] = min(G[r], min(a, b) + as);
      acc = min(acc, G[r] + lane_table(v_int, s));
    }
  }
  return acc;
}

// Diagnostic build only (-DDRNA_STAMPS): per-wave cycle totals of each phase of block 0, written to
// the tail of its workspace (never read by the kernel).
#ifdef DRNA_STAMPS
#define STAMP(k) do { long long _n = clock64(); st_acc[k] += _n - st_last; st_last = _n; } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

// exterior column j: f5[j] = min(f5[j-1], min_i f5[i-1] + c[i,j] + E_ExtLoop); one wave, every lane stores the same value
template <int NT>
__device__ __forceinline__ void mfe_f5_column(MfeFastSmem<NT>& sm, const int32_t* __restrict__ EXT, int ld, int j, int lane) {
  int m = INF_DEV;
  for (int i = lane + 1; i <= j - TURN - 1; i += WAVE) {
    const int x = EXT[j * ld + i];
    if (x < INF_DEV / 2) m = min(m, sm.f5[i - 1] + x);
  }
  m = wave_min_i32(m);
  const int prev = sm.f5[j - 1];
  sm.f5[j] = prev < m ? prev : m;
}

// ---- sweep of one diagonal by one sweep wave over NBLK live 64-tower blocks (branch-free over blocks)
struct SweepCtx {
  int d, par, tb_lo, aw, NA, ncand;
  int v_cand, v_candL;      // lane k -> k-th enumerated loop shape of this wave: u1 | u2 << 8 | kind << 16, size term
  int v_ro[4];              // lane l of v_ro[q] -> offset of fml row 64 q + l
  int TermAU, e_bulge1, e_int23;
};

__device__ __forceinline__ int row_offset(const SweepCtx& X, int r) {
  return r < 64 ? lane_table(X.v_ro[0], r) : r < 128 ? lane_table(X.v_ro[1], r - 64)
       : r < 192 ? lane_table(X.v_ro[2], r - 128) : lane_table(X.v_ro[3], r - 192);
}

template <int NT, int NBLK>
__device__ __forceinline__ void mfe_sweep_blocks(MfeFastSmem<NT>& sm, const MfeTables& T, const SweepCtx& X, int n, int sh,
                                                 int off0, int lane) {
  constexpr int RS = MfeFastSmem<NT>::RS;
  const int d = X.d, ncell = n - d, INF = INF_DEV, HALF = INF_DEV / 2;
  int ci[NBLK], cx[NBLK], acc[NBLK];
  bool any_pair = false;
#pragma unroll
  for (int b = 0; b < NBLK; b++) {
    int i = (X.tb_lo + b) * WAVE + lane + 1 - sh - off0;
    const bool act = i >= 1 && i <= ncell;
    i = i < 1 ? 1 : (i > ncell ? ncell : i);
    const int t = act ? pair_type(sm.Sp[i], sm.Sp[i + d]) : 0;
    ci[b] = i;
    cx[b] = t * 16 + sm.S[i + 1] * 4 + sm.S[i + d - 1];      // ij index; t == 0 <=> cx < 16
    acc[b] = INF;
    any_pair |= __ballot(t != 0) != 0ull;
  }
  // ---- enumerated loops (kinds: 0 fixed small shape, 1 bulge, 2 1xn).  Loads of all blocks are issued
  // before any is used: the LDS round trip is paid once per candidate, not once per block.
  if (any_pair) {
    int tq[NBLK], m1[NBLK];
#pragma unroll
    for (int b = 0; b < NBLK; b++) { tq[b] = (cx[b] >> 4) > 2 ? X.TermAU : 0; m1[b] = sm.mm1n[cx[b]]; }
    for (int k = 0; k < X.ncand; k++) {
      const int cd = lane_table(X.v_cand, k);
      const int u1 = cd & 255, u2 = (cd >> 8) & 255, kind = cd >> 16;
      const int dp = d - 2 - u1 - u2;
      if (dp <= TURN) continue;
      const int L = lane_table(X.v_candL, k);
      const int* row = sm.wring + (dp & 31) * RS + 1 + u1;
      int w[NBLK];
#pragma unroll
      for (int b = 0; b < NBLK; b++) w[b] = row[ci[b]];
      if (kind == 1) {
#pragma unroll
        for (int b = 0; b < NBLK; b++) acc[b] = min(acc[b], (w[b] >> 8) + L + tq[b]);
      } else if (kind == 2) {
        int m[NBLK];
#pragma unroll
        for (int b = 0; b < NBLK; b++) m[b] = sm.mm1np[w[b] & 127];
#pragma unroll
        for (int b = 0; b < NBLK; b++) acc[b] = min(acc[b], (w[b] >> 8) + L + m[b] + m1[b]);
      } else {
        const int shape = L;   // 0..8: (0,0) (0,1) (1,0) (1,1) (1,2) (2,1) (2,2) (2,3) (3,2)
#pragma unroll
        for (int b = 0; b < NBLK; b++) {
          const int cpq = w[b] >> 8, info = w[b] & 127, t2 = info >> 4, t = cx[b] >> 4;
          const int si1 = (cx[b] >> 2) & 3, sj1 = cx[b] & 3;
          int e;
          switch (shape) {
            case 0: e = sm.stackp[t * 8 + t2]; break;
            case 1: case 2: e = X.e_bulge1 + sm.stackp[t * 8 + t2]; break;
            case 3: e = sm.int11p[(t * 8 + t2) * 16 + si1 * 4 + sj1]; break;
            case 4: e = T.int21[(t * 8 + t2) * 64 + si1 * 16 + ((info >> 2) & 3) * 4 + sj1] - (t2 > 2 ? X.TermAU : 0); break;
            case 5: e = T.int21[(t2 * 8 + t) * 64 + ((info >> 2) & 3) * 16 + si1 * 4 + (info & 3)] - (t2 > 2 ? X.TermAU : 0); break;
            case 6: e = T.int22[(t * 8 + t2) * 256 + si1 * 64 + (info & 3) * 16 + ((info >> 2) & 3) * 4 + sj1] - (t2 > 2 ? X.TermAU : 0); break;
            default: e = X.e_int23 + sm.mm23[cx[b]] + sm.mm23p[info]; break;   // (2,3), (3,2)
          }
          acc[b] = min(acc[b], cpq + e);
        }
      }
    }
#pragma unroll
    for (int b = 0; b < NBLK; b++)
      if (acc[b] < HALF) atomicMin(&sm.accI[X.par][(X.tb_lo + b) * WAVE + lane], acc[b]);
  }
  // ---- multiloop splits fML[i,u] + fML[u+1,j], u = i + tt, tt = 4 + aw mod NA
#pragma unroll
  for (int b = 0; b < NBLK; b++) acc[b] = INF;
  for (int tt = TURN + 1 + X.aw; tt <= d - TURN - 2; tt += X.NA) {
    const int* ra = sm.fml + row_offset(X, tt) - 1;
    const int* rb = sm.fml + row_offset(X, d - tt - 1) + tt;
    int fa[NBLK], fb[NBLK];
#pragma unroll
    for (int b = 0; b < NBLK; b++) { fa[b] = ra[ci[b]]; fb[b] = rb[ci[b]]; }
#pragma unroll
    for (int b = 0; b < NBLK; b++) acc[b] = min(acc[b], fa[b] + fb[b]);
  }
#pragma unroll
  for (int b = 0; b < NBLK; b++)
    if (acc[b] < HALF) atomicMin(&sm.accK[X.par][(X.tb_lo + b) * WAVE + lane], acc[b]);
}

// Structure of one diagonal step k (ONE workgroup barrier per diagonal):
//   * waves 0..NB-1 ("finalize waves", one lane per tower slot) turn the minima gathered for diagonal
//     k-1 into c / fML / ring rows, stream c to HBM and advance f5;
//   * waves NB..15 ("sweep waves") gather the candidate minima of diagonal k -- tower step, enumerated
//     small loops, multiloop splits.  None of those reads anything diagonal k-1 produces (interior loops
//     look at diagonals <= k-2, splits at diagonals <= k-5), so the two halves run concurrently.
// The scalar unit is shared by all 16 waves, so wave-uniform bookkeeping is kept off the inner loops:
// no division, no per-term offset arithmetic, per-wave candidate lists and row offsets held one entry
// per lane and fetched with v_readlane; a sweep wave takes the candidates / split points congruent to
// its index and sweeps every live 64-tower block with them; minima meet in LDS (ds_min).
template <int NT>
__device__ void mfe_fill_lds(MfeFastSmem<NT>& sm, const MfeArgs& A, int32_t* __restrict__ Wc, int32_t* __restrict__ EXT) {
  constexpr int NW = NT / WAVE;
  constexpr int RS = MfeFastSmem<NT>::RS;
  const MfeTables& T = *A.T;
  const int n = A.L, ld = A.ld;
  const int tid = threadIdx.x, lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(wave_id());
  const int INF = INF_DEV, HALF = INF_DEV / 2;
  // every scalar / small table the inner loops need is taken out of global memory ONCE
  const int ninio = T.ninio, max_ninio = T.max_ninio, MLbase = T.MLbase, MLclosing = T.MLclosing,
            MLintern = T.MLintern, TermAU = T.TermAU;
  const int lt = lane <= 30 ? lane : 30;
  const int v_int = T.interior[lt];                                          // lane s -> interior[s]
  const int v_asym = min(max_ninio, lane * ninio);                           // lane a -> asymmetry penalty
  // tower blocks, centred on the sequence
  const int NB = (n + WAVE - 1) / WAVE;
  const int off0 = (NB * WAVE - n) / 2;
  const int NA = NW - NB;                  // sweep waves
  const int aw = wave - NB;                // index among the sweep waves (< 0: finalize wave)
  const int NG = NA / NB;                  // sweep waves pinned to one tower block (>= 3 for n <= 256, NT = 1024)
  const int my_tb = aw >= 0 ? aw / NG : NB, my_g = aw >= 0 ? aw - my_tb * NG : 0;
  const bool pinned = aw >= 0 && my_tb < NB;
  int GE[GSLOTS], GO[GSLOTS], xs[GSLOTS];
#pragma unroll
  for (int r = 0; r < GSLOTS; r++) {
    GE[r] = INF; GO[r] = INF;
    const int rho = r * NG + my_g;
    xs[r] = (pinned && rho < GRES) ? (int)((unsigned)(TURN + 1 + 50 - rho) % (unsigned)GRES) : 99;   // value for d = TURN + 1
  }
  SweepCtx X;
  X.aw = aw; X.NA = NA; X.TermAU = TermAU; X.e_bulge1 = T.bulge[1]; X.e_int23 = T.interior[5] + ninio;
  {
    // this wave's enumerated loop shapes: the 121 shapes (9 fixed small loops, (0,u) (u,0) bulges u=2..30,
    // (1,u) (u,1) loops u=3..29) dealt round-robin over the sweep waves; lane k holds the k-th of this wave
    const int c = aw + lane * NA;
    int u1 = 0, u2 = 0, kind = 0, L = 0;
    if (c < 9) {
      u1 = (int)((0x322211100ull >> (4 * c)) & 15ull);
      u2 = (int)((0x232121010ull >> (4 * c)) & 15ull);
      kind = 0; L = c;
    } else if (c < 67) {
      if (c < 38) { u1 = 0; u2 = c - 7; } else { u1 = c - 36; u2 = 0; }
      kind = 1; L = T.bulge[u1 + u2 <= 30 ? u1 + u2 : 30];
    } else if (c < 121) {
      if (c < 94) { u1 = 1; u2 = c - 64; } else { u1 = c - 91; u2 = 1; }
      const int nl = u1 + u2 - 1;
      kind = 2; L = T.interior[nl + 1 <= 30 ? nl + 1 : 30] + min(max_ninio, (nl - 1) * ninio);
    }
    X.v_cand = u1 | (u2 << 8) | (kind << 16);
    X.v_candL = L;
    X.ncand = aw >= 0 ? (121 - aw + NA - 1) / NA : 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int r = 64 * q + lane;
      X.v_ro[q] = (r >= TURN + 1 && r < n) ? fml_off(r, n) : 0;
    }
  }

  for (int k = tid; k < 4 * RS; k += NT) sm.dml[k] = INF;
  for (int k = tid; k <= n; k += NT) { sm.hpl[k] = A.hp_len[k]; sm.rowoff[k] = k >= TURN + 1 ? fml_off(k, n) : 0; }
  for (int k = tid; k < MfeFastSmem<NT>::NSLOT; k += NT)
    for (int p = 0; p < 2; p++) { sm.accG[p][k] = INF; sm.accI[p][k] = INF; sm.accK[p][k] = INF; }
  for (int k = tid; k < 64; k += NT) sm.stackp[k] = sm.stack[k] - ((k & 7) > 2 ? TermAU : 0);
  for (int k = tid; k < 1024; k += NT) sm.int11p[k] = sm.int11[k] - (((k >> 4) & 7) > 2 ? TermAU : 0);
  for (int k = tid; k < 128; k += NT) {
    sm.mm1np[k] = sm.mm1n[k] - ((k >> 4) > 2 ? TermAU : 0);
    sm.mm23p[k] = sm.mm23[k] - ((k >> 4) > 2 ? TermAU : 0);
  }
  for (int j = tid; j <= n && j <= TURN + 1; j += NT) sm.f5[j] = 0;
  __syncthreads();

#ifdef DRNA_STAMPS
  long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long st_last = clock64();
#endif
  // steps k = TURN+1 .. n: sweep diagonal k (k < n) while finalizing diagonal k-1 (k-1 > TURN).  The two
  // roles run separate copies of the loop (same trip count, one barrier per trip) so that neither
  // carries the other's registers.
  if (aw < 0) {
    // ================= finalize waves
    for (int k = TURN + 1; k <= n; k++) {
      const int d = k - 1;
      if (d > TURN) {
        const int ncell = n - d, sh = d >> 1, par = d & 1;
        const int i = tid + 1 - sh - off0;
        if (i >= 1 && i <= ncell) {
          const int aG = sm.accG[par][tid], aI = sm.accI[par][tid], aK = sm.accK[par][tid];
          sm.accG[par][tid] = INF; sm.accI[par][tid] = INF; sm.accK[par][tid] = INF;
          const int j = i + d;
          const int t = pair_type(sm.Sp[i], sm.Sp[j]);
          const int tau = t > 2 ? TermAU : 0;
          int c = INF, info = 0, cb = INF;
          if (t) {
            const int ij = t * 16 + sm.S[i + 1] * 4 + sm.S[j - 1];
            c = mfe_hairpin_e(sm, T, sm.hpl[d - 1], i, j, t);
            c = min(c, aI);
            c = min(c, aG + sm.mmI[ij]);
            const int dml = sm.dml[((d - 2) & 3) * RS + i + 1];
            if (dml < HALF)
              c = min(c, dml + MLclosing + MLintern + tau + sm.mmM[rtype_of(t) * 16 + sm.S[j - 1] * 4 + sm.S[i + 1]]);
            if (c >= HALF) c = INF;
            info = (rtype_of(t) << 4) | (sm.S[j + 1] << 2) | sm.S[i - 1];
            cb = c < INF ? c + tau : INF;    // TermAU of this pair seen as the inner pair of a loop (rtype keeps > 2)
          }
          sm.wring[(d & 31) * RS + i] = cb * 256 + info;
          sm.ciring[(d & 31) * RS + i] = c < INF ? c + sm.mmI[info] : INF;
          Wc[d * ld + i] = c * 256 + info;
          EXT[j * ld + i] = c < INF ? c + tau + mfe_extstem(sm, t, i, j, n) : INF;
          int f = INF;
          if (d - 1 > TURN) {
            const int fa = sm.fml[sm.rowoff[d - 1] + i], fb = sm.fml[sm.rowoff[d - 1] + i - 1];
            if (fa < HALF) f = fa + MLbase;
            if (fb < HALF) f = min(f, fb + MLbase);
          }
          if (c < INF) f = min(f, c + MLintern + tau + sm.mmM[t * 16 + sm.S[i - 1] * 4 + sm.S[j + 1]]);
          const int dec = aK >= HALF ? INF : aK;
          sm.dml[(d & 3) * RS + i] = dec;
          sm.fml[sm.rowoff[d] + i - 1] = min(f, dec);
        }
      }
      // exterior column j = k-3 by wave 0: its cells (diagonals <= k-4) were stored in step <= k-3 and
      // drained by the barrier that ended that step
      if (wave == 0 && k - 3 >= TURN + 2) mfe_f5_column<NT>(sm, EXT, ld, k - 3, lane);
      STAMP(4);
      __syncthreads();                     // one barrier per diagonal (drains vmcnt: c / EXT stores of this step)
      STAMP(3);
    }
  } else {
    // ================= sweep waves
    for (int k = TURN + 1; k <= n; k++) {
      if (k < n) {
        const int d = k;
        const int ncell = n - d, sh = d >> 1, par = d & 1;
        const int lo = sh + off0, hi = ncell + sh + off0 - 1;
        const int tb_lo = lo >> 6, tb_hi = hi >> 6;
        // ---- tower step: generic interior loops, towers pinned to their waves
        if (pinned && my_tb >= tb_lo && my_tb <= tb_hi) {
          int i = my_tb * WAVE + lane + 1 - sh - off0;
          i = i < 1 ? 1 : (i > ncell ? ncell : i);
          const int accG = par ? mfe_tower_step<NT>(sm, GO, xs, d, i, v_int, v_asym)
                               : mfe_tower_step<NT>(sm, GE, xs, d, i, v_int, v_asym);
          atomicMin(&sm.accG[par][my_tb * WAVE + lane], accG);
        }
        STAMP(0);
        X.d = d; X.par = par; X.tb_lo = tb_lo;
        switch (tb_hi - tb_lo) {
          case 0: mfe_sweep_blocks<NT, 1>(sm, T, X, n, sh, off0, lane); break;
          case 1: mfe_sweep_blocks<NT, 2>(sm, T, X, n, sh, off0, lane); break;
          case 2: mfe_sweep_blocks<NT, 3>(sm, T, X, n, sh, off0, lane); break;
          default: mfe_sweep_blocks<NT, 4>(sm, T, X, n, sh, off0, lane); break;
        }
        STAMP(1);
      }
#pragma unroll
      for (int r = 0; r < GSLOTS; r++)
        if (xs[r] < 99) xs[r] = xs[r] == GRES - 1 ? 0 : xs[r] + 1;
      __syncthreads();
      STAMP(3);
    }
  }
#ifdef DRNA_STAMPS
  if (blockIdx.x == 0 && lane == 0) {
    long long* dbg = reinterpret_cast<long long*>(Wc + 3ll * ld * ld);
    for (int k = 0; k < 8; k++) dbg[wave * 8 + k] = st_acc[k];
  }
#endif
  // the remaining exterior columns (every store has landed: the loop ended with a draining barrier)
  if (wave == 0) {
    for (int j = max(TURN + 2, n - 2); j <= n; j++) mfe_f5_column<NT>(sm, EXT, ld, j, lane);
  }
  __syncthreads();
}

template <int NT>
__global__ __launch_bounds__(NT) void mfe_lds_kernel(MfeArgs A) {
  __shared__ MfeFastSmem<NT> sm;
  const int r = blockIdx.x;
  const int n = A.L, ld = A.ld, tid = threadIdx.x;
  const MfeTables& T = *A.T;
  int32_t* base = A.ws + (long long)r * A.ws_stride;
  const long long tab = (long long)ld * ld;
  int32_t* Wc = base;
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
  const char* seq = A.seqs + (long long)r * n;
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
    for (int k = tid; k < n; k += NT) A.ss[(long long)r * n + k] = '.';
    return;
  }

  int status = ST_OK;
  for (int round = 0; round <= A.pk_rounds; round++) {
    for (int k = tid; k < n; k += NT) sm.ssw[k] = '.';
    mfe_fill_lds<NT>(sm, A, Wc, EXT);               // ends with a barrier
    if (wave_id() == 0) {
      const bool ok = mfe_traceback(sm, A, Wc, FmlLds<NT>{&sm, n}, EXT);
      if (lane_id() == 0) {
        if (round == 0) A.Emfe[r] = sm.f5[n];
        sm.flag = ok ? 0 : 1;
      }
    }
    __syncthreads();
    if (sm.flag) { status = ST_TRACEBACK; break; }
    const char op = round == 0 ? '(' : round == 1 ? '[' : round == 2 ? '<' : '{';
    const char cl = round == 0 ? ')' : round == 1 ? ']' : round == 2 ? '>' : '}';
    __syncthreads();
    int any = 0;
    for (int k = tid; k < n; k += NT) {
      const char ch = sm.ssw[k];
      if (ch == '(') { sm.sspk[k] = op; any = 1; }
      else if (ch == ')') sm.sspk[k] = cl;
      if (sm.sspk[k] != '.') sm.Sp[k + 1] = 4;
    }
    if (any) sm.flag = 2;
    __syncthreads();
    const bool more = (round == 0) || (sm.flag == 2);
    __syncthreads();
    if (tid == 0) sm.flag = 0;
    __syncthreads();
    if (!more) break;
  }
  for (int k = tid; k < n; k += NT) A.ss[(long long)r * n + k] = sm.sspk[k];
  if (tid == 0) A.status[r] = status;
}

}  // namespace drna
