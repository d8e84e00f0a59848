// fold_mfe_dual.hpp -- two-workgroup form of the LDS-resident MFE fill (n <= MFE_FAST_NMAX), used when the batch is small
// enough that half of the chip would otherwise idle (R = 64 replicas: 2 folds x 64 workgroups on 256 CUs).  Same recursions
// and results, bit for bit, as mfe_lds_kernel (reference utils/energy_scores.py:151).
//
// Roles (two workgroups of the grid fold sequence r, 8 blocks apart = on one XCD: pair_block, fold_common.hpp):
//   MAIN   (mfe_lds_body<NT, true>, fold_mfe_lds.hpp): finalize, tower step (generic interior loops), the six NEAR shapes
//          whose inner pair is at most four diagonals back, exterior column, traceback, pseudoknot rounds.  After every
//          diagonal it publishes the ring word and the fML value of each cell.
//   HELPER (mfe_helper, here): multiloop splits (K) and the 115 FAR bulge / 1xn / small shapes (E) of diagonal D, which need
//          nothing newer than diagonal D-5: it mirrors the published rows into its own LDS (fML triangle, 32-row ring),
//          runs up to five diagonals behind the main workgroup's finalize and AHEAD of its need (the results of D are
//          consumed when D is finalized), and publishes two minima per cell and diagonal.
// Wave roles in the helper: wave 0 INBOUND (waits for the rows of diagonal D-4 and copies them to LDS), wave 1 OUTBOUND (ships the minima of diagonal D-1,
// drains its stores, raises flagB), wave 2 TABLES (pairable list / staged small-loop energies / shape table of diagonal D+1),
// waves 3..15 work through the items of diagonal D from a queue.  One workgroup barrier
// per diagonal; the tables are double-buffered by diagonal parity.  Handshake: DualLink in fold_common.hpp.
#pragma once
#include "fold_mfe_lds.hpp"

namespace drna {

template <int NT>
struct MfeHelperSmem {
  static constexpr int RS = MFE_FAST_NMAX + 2;
  static constexpr int TRI = (MFE_FAST_NMAX - 4) * (MFE_FAST_NMAX - 3) / 2;
  static constexpr int NSLOT = 4 * WAVE;
  static constexpr int NL = MFE_FAST_NMAX + 8;
  static constexpr int XT_STACK = 0, XT_INT11 = 64, XT_MM1N = 64 + 1024, XT_MM23 = 64 + 1024 + 128;
  int fml[TRI + 8];
  int wring[33 * RS];
  int accK[2][NSLOT], accI[2][NSLOT];
  int xtab[64 + 1024 + 128 + 128];
  int mm1n[128], mm23[128];
  int eshape[128], etab[2][128];
  int plist[2][NL], xe[2][1];       // xe: only named by the shared item code's small-shape branch, which the helper never takes
  int pcnt[2], qhead[2];
  int flag, fail, failr[2];       // failr[D & 1]: set by the inbound wave during step D, read by everybody after the step's barrier
  unsigned char S[MFE_FAST_NMAX + 4], Sp[MFE_FAST_NMAX + 4];
};

template <int NT>
__device__ __forceinline__ void mfe_helper(MfeHelperSmem<NT>& sm, MfeArgs A, int r, DualLink lk) {
  using SM = MfeHelperSmem<NT>;
  constexpr int RS = SM::RS;
  const MfeTables& T = *A.T;
  const int n = A.L, tid = threadIdx.x, lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(wave_id());
  const int INF = INF_DEV, TermAU = T.TermAU;
  int32_t* const PL = A.ws + (long long)r * A.ws_stride + 3LL * A.ld * A.ld;      // the main role's list tables (mfe_lds_body)
  int32_t* const PLX = A.ws + (long long)r * A.ws_stride + 1LL * A.ld * A.ld;
  const int32_t* const xw = reinterpret_cast<const int32_t*>(lk.xa);
  const int32_t* const xf = xw + (MFE_FAST_NMAX + 2) * XP;
  int32_t* const xk = reinterpret_cast<int32_t*>(lk.xb);
  int32_t* const xi = xk + (MFE_FAST_NMAX + 2) * XP;

  // ---- constant tables (same contents as the main workgroup's) and the sequence
  for (int k = tid; k < 64; k += NT) sm.xtab[SM::XT_STACK + k] = T.stack[k] - ((k & 7) > 2 ? TermAU : 0);
  for (int k = tid; k < 1024; k += NT) sm.xtab[SM::XT_INT11 + k] = T.int11[k] - (((k >> 4) & 7) > 2 ? TermAU : 0);
  for (int k = tid; k < 128; k += NT) {
    sm.mm1n[k] = T.mm1n[k]; sm.mm23[k] = T.mm23[k];
    sm.xtab[SM::XT_MM1N + k] = T.mm1n[k] - ((k >> 4) > 2 ? TermAU : 0);
    sm.xtab[SM::XT_MM23 + k] = T.mm23[k] - ((k >> 4) > 2 ? TermAU : 0);
  }
  mfe_init_eshape(sm, T, tid, NT);
  for (int k = tid; k < 33 * RS; k += NT) sm.wring[k] = INF * 256;
  for (int k = tid; k < SM::NSLOT; k += NT)
    for (int p = 0; p < 2; p++) { sm.accK[p][k] = INF; sm.accI[p][k] = INF; }
  if (tid == 0) { sm.flag = 0; sm.fail = 0; sm.failr[0] = 0; sm.failr[1] = 0; }
  __syncthreads();
  const char* seq = A.seqs + (long long)r * n;
  for (int k = tid; k < n; k += NT) {
    const int c = enc_nt(seq[k]);
    if (c < 0) sm.flag = 1;
    sm.S[k + 1] = (unsigned char)(c < 0 ? 0 : c);
  }
  __syncthreads();
  if (sm.flag) return;                               // the main workgroup reports the bad character
  if (tid == 0) { sm.S[0] = sm.S[n]; sm.S[n + 1] = sm.S[1]; }
  const int e_bulge1 = keep_i32(T.bulge[1]), e_int23 = keep_i32(T.interior[5] + T.ninio);

  // pairable cells of diagonal dn (i | pair info << 8, as in the main workgroup's list) and the far-shape table, rebuilt here
  // from the sequence; the staged small-loop energies are not needed: every small shape is a NEAR shape
  auto prepare = [&](const int dn) {
    const int par = dn & 1, ncell = n - dn;
    int t[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int i = c * WAVE + lane + 1;
      t[c] = i <= ncell ? pair_type(sm.Sp[i], sm.Sp[i + dn]) : 0;
    }
    int cnt = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const unsigned long long m = __ballot(t[c] != 0);
      if (t[c]) {
        const int i = c * WAVE + lane + 1, j = i + dn;
        const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
        if (pos < SM::NL) sm.plist[par][pos] = i | ((t[c] * 16 + sm.S[i + 1] * 4 + sm.S[j - 1]) << 8);
      }
      cnt += __popcll(m);
    }
    mfe_prepare_etab(sm, dn, lane, E_FAR);
    if (lane == 0) { sm.pcnt[par] = cnt; sm.qhead[par] = 0; }
  };

  // the main role's pairable lists from diagonal PL_D1 on (mfe_pl_row): it reads the first of them after it has seen a flag of the
  // round, which is raised behind a drained barrier.  The first round's pairing codes are the sequence itself: its lists are
  // built at once, while the main role is still filling its tables (the later rounds' codes come from the main role)
  auto build_lists = [&]() {
    for (int d = PL_D1 + wave; d < n; d += NT / WAVE) mfe_pl_row<true>(sm, T, PL, PLX, A.ld, n, d, lane, TermAU);
  };
  for (int k = tid; k <= n + 1; k += NT) sm.Sp[k] = k >= 1 && k <= n ? sm.S[k] : (unsigned char)4;
  __syncthreads();
  build_lists();

#ifdef DRNA_TL
  // timeline of sequence 0's helper (tools/timeline.py mfe): per step, clocks of wave 0 at the top and after the inbound copy, of
  // wave 1 after the outbound stores, of the first and the last worker wave after their items (rows 49 .. 53 of the main role's mark table)
  long long* htl = reinterpret_cast<long long*>(A.ws + (long long)r * A.ws_stride + 2LL * A.ld * A.ld);
  const bool htl_on = r == 0 && lane == 0;
#define HTL(ev, D) do { if (htl_on) htl[((49 + (ev)) << 8) + (D)] = (long long)wall_clock64(); } while (0)
#else
#define HTL(ev, D) do { } while (0)
#endif
  bool failed = false;
  for (int round = 0; round <= A.pk_rounds; round++) {
    const int base = dual_base(lk.epoch, round);
    // ---- the round starts when the main workgroup has published its pairing codes (or ends the call)
    if (wave == 0) {
      if (!wait_flag_wave(lk.flagA, base + TURN)) sm.fail = 1;
      else if (flag_ge(__builtin_amdgcn_readfirstlane(ld_agent(lk.flagA)), dual_done(lk.epoch))) sm.fail = 2;    // no further round
    }
    __syncthreads();
    if (sm.fail) break;
    if (round > 0) {
      for (int k = tid; k <= n + 1; k += NT) sm.Sp[k] = (unsigned char)ld_agent(lk.xs + k);
      __syncthreads();
      build_lists();
    }
    drain_vmem();
    __syncthreads();
    if (wave == 0 && DUAL_D0 < n) prepare(DUAL_D0);
    __syncthreads();

    int fa = 0;
    if (wave == 0) fa = ld_agent(lk.flagA);
    for (int D = DUAL_D0; D <= n + 1; D++) {         // steps D = n, n+1 only ship the last diagonal and raise its flag
      const int par = D & 1, ncell = n - D;
      if (wave == 0) {
        HTL(0, D);
        // ---- inbound: rows of diagonal D+1-DLAG (needed from diagonal D+1 on); the flag was read at the end of the
        // previous step, so the only latency in the step is one round of loads
        if (D + 1 < n) {
          const int dr = D + 1 - DLAG;
          if (dr > TURN) {
            if (!flag_ge(__builtin_amdgcn_readfirstlane(fa), base + dr) && !wait_flag_wave(lk.flagA, base + dr)) sm.failr[D & 1] = 1;
            const int ro = fml_off(dr, n);
            int vw[4], vf[4];                                    // all eight loads in flight, then the LDS stores
#pragma unroll
            for (int c = 0; c < 4; c++) {
              const int i = c * WAVE + lane + 1;
              const bool on = i <= n - dr;
              vw[c] = on ? ld_agent(xw + dr * XP + i) : 0;
              vf[c] = on ? ld_agent(xf + dr * XP + i) : 0;
            }
#pragma unroll
            for (int c = 0; c < 4; c++) {
              const int i = c * WAVE + lane + 1;
              if (i <= n - dr) { sm.wring[(dr & 31) * RS + i] = vw[c]; sm.fml[ro + i - 1] = vf[c]; }
            }
          }
        }
        HTL(1, D);
        fa = ld_agent(lk.flagA);                       // for the next step (left in flight across the barrier)
      } else if (wave == 2) {
        if (D + 1 < n) prepare(D + 1);               // tables of the next diagonal (sequence only: no wait)
        HTL(5, D);
      } else if (wave == 1) {
        // ---- outbound: the flag of diagonal D-2 (its minima were stored during the previous step: they have landed by
        // now, so the wait is short), then the minima of diagonal D-1 (reset for diagonal D+1)
        drain_vmem();
        if (lane == 0 && D - 2 >= DUAL_D0) st_agent(lk.flagB, base + D - 2);
        const int ds = D - 1;
        if (ds >= DUAL_D0 && ds < n) {
          const int ps = ds & 1;
          for (int i = lane + 1; i <= n - ds; i += WAVE) {
            st_agent(xk + ds * XP + i, (int32_t)sm.accK[ps][i]);
            st_agent(xi + ds * XP + i, (int32_t)sm.accI[ps][i]);
            sm.accK[ps][i] = INF; sm.accI[ps][i] = INF;
          }
        }
        HTL(2, D);
      } else if (D < n) {
        // ---- workers: K items (32-cell blocks of split sweeps), then the far-shape items of the pairable cells
        const int pcnt = __builtin_amdgcn_readfirstlane(sm.pcnt[par]);
        // a late diagonal has one to three 32-cell blocks and up to 180 split points per cell: as one item per block one worker wave
        // walked a 1.5 - 1.9 us chain while twelve idled, and from diagonal ~125 on the main role waited for its helper in every step
        // (tools/timeline.py mfe).  The split points of a block go to 1, 2, 4 or 8 items as the cells get fewer (minima are order-free)
        const int kssh = ncell > 96 ? 0 : ncell > 64 ? 1 : 2;
        const int k_lo = TURN + 1 + KEDGE, k_hi = D - TURN - 2 - KEDGE;
        const int k_per = (((k_hi - k_lo + 1 + (1 << kssh) - 1) >> kssh) + 3) & ~3;       // split points per item
        const int nK = (DRNA_SKIP & 8) ? 0 : ((ncell + 31) >> 5) << kssh, nE = (DRNA_SKIP & 2) ? 0 : e_items_per_block<E_FAR>() * ((pcnt + WAVE - 1) >> 6);      // (DRNA_SKIP: timing builds)
        const int nItems = __builtin_amdgcn_readfirstlane(nK + nE);
        for (int it = queue_pop(&sm.qhead[par], lane); it < nItems; it = queue_pop(&sm.qhead[par], lane)) {
          if (it < nK) {
            const int lo = k_lo + (it & ((1 << kssh) - 1)) * k_per;
            mfe_k_item(sm, it >> kssh, D, n, ncell, par, 0, lane, lo, min(k_hi, lo + k_per - 1));
          } else mfe_e_item<E_FAR>(sm, it - nK, D, par, pcnt, 0, lane, TermAU, e_bulge1, e_int23);
        }
        if (wave == 3) HTL(3, D);
        if (wave == NT / WAVE - 1) HTL(4, D);
        HTL(6 + wave, D);
      }
      // an LDS-only barrier: the inbound wave's flag load (for the next step) really stays in flight across it -- behind a
      // draining barrier that cross-XCD round trip was the helper's step (tools/timeline.py mfe: everything done at +1.1 us, step
      // 1.9 us, and from diagonal ~125 on the main role waited for its helper in every step).  Nothing else needs the drain:
      // the outbound wave drains its own stores before it raises their flag, the inbound rows are in LDS, the workers store to LDS
      lds_barrier();
      if (sm.failr[D & 1]) { failed = true; break; }         // (the other parity's word is the one step D+1 may write)
    }
    if (failed) break;
  }
}

// grid = pair_grid(R) workgroups: main and helper of a sequence 8 blocks apart (pair_block, fold_common.hpp).  LDS is one buffer
// used as either role's struct.
template <int NT>
__global__ __launch_bounds__(NT) void mfe_dual_kernel(MfeArgs A, DualLink lk, int R) {
  constexpr size_t BYTES = sizeof(MfeFastSmem<NT>) > sizeof(MfeHelperSmem<NT>) ? sizeof(MfeFastSmem<NT>) : sizeof(MfeHelperSmem<NT>);
  __shared__ __attribute__((aligned(16))) unsigned char raw[BYTES];
  int r, is_helper;
  pair_block(blockIdx.x, r, is_helper);
  if (r >= R) return;
  // per-sequence slices of the link
  lk.flagA += r * 64; lk.flagB += r * 64 + 32;
  lk.xs += (long long)r * 256;
  lk.xa = reinterpret_cast<int32_t*>(lk.xa) + (long long)r * 2 * (MFE_FAST_NMAX + 2) * XP;
  lk.xb = reinterpret_cast<int32_t*>(lk.xb) + (long long)r * 2 * (MFE_FAST_NMAX + 2) * XP;
  if (is_helper) mfe_helper<NT>(*reinterpret_cast<MfeHelperSmem<NT>*>(raw), A, r, lk);
  else mfe_lds_body<NT, true>(*reinterpret_cast<MfeFastSmem<NT>*>(raw), A, r, lk);
}

}  // namespace drna
