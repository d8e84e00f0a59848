// fold_pf_strip.hpp -- McCaskill partition function (inside) of ONE sequence by SEVERAL workgroups, each keeping its share
// of the rings in LDS like fold_pf_lds.hpp: the path for 200 < n <= 2046, where one workgroup's LDS cannot hold a 32-diagonal
// fp64 ring of the whole sequence.  Same recursions and outputs as fold_pf.hpp / fold_pf_lds.hpp (reference
// utils/energy_scores.py:150 with compute_bpp = 0).
//
// The triangle is cut into STRIPS OF COLUMNS i (5' ends): strip s owns the cells (i, j) with c0 <= i <= c1, all j.  Every
// cell depends only on cells with i' >= i, so the dependencies between strips run ONE WAY: the strip with the largest i (s = 0)
// needs nobody, strip s needs strips < s.  No workgroup ever waits for a workgroup that waits for it: an upstream strip simply
// runs a few diagonals ahead, publishes what its neighbour needs, and the neighbour finds it there.  Per diagonal d an
// upstream strip publishes ONE record (88 doubles) for the strip below it:
//     [0,32)  qb * expMismatchI of its first 32 columns   (the halo of the neighbour's ring: interior loops reach <= 31 columns)
//     [32,48) their info bytes (one int32 each)
//     [48,78) the tower sums (3 waves x 10 entries) of the tower that leaves the strip after diagonal d: the generic-interior
//             recurrence follows (i, d) -> (i-1, d+2), i.e. a tower walks from strip to strip
//     78..80  qm1, U, D of its first column (the multiloop recurrences look one column to the right)
// and the full tables the multiloop sums and the exterior column read anyway (QM1, QEXT) are stored write-through.  In strip-
// local coordinates (i_loc = i - c0 + 1, n_loc = n - c0 + 1) a strip is the LDS kernel on the suffix S[c0..n] that computes
// only its first `wid` columns and finds columns wid+1 .. wid+31 in the halo.  Tower sums stay in registers: physical tower
// lane = (i_loc + d/2) mod P, so that a tower keeps its lane while it is inside the strip.
//
// Cross-workgroup visibility follows the CDNA4 guide (R1): every handed-off byte is stored sc1 (write-through) and loaded
// sc1 (L1-bypassing), every storing wave drains its stores before the workgroup barrier that ends the step, ONE lane then
// stores the strip's flag (epoch << 12 | diagonal, monotone over calls, compared wrap-safe); the consumer's service wave polls
// it with sc1 loads, bounded (ST_SYNC on expiry).  Deadlock freedom needs only that the strips of a sequence are dispatched in
// order (upstream strips have the lower block index; a strip only ever waits for lower block indices); strips of one sequence
// get block indices that are equal mod 8, i.e. the same XCD under round-robin placement (speed only).
#pragma once
#include "fold_pf_lds.hpp"
#include <type_traits>

namespace drna {

// ---- blocked multiloop sums (round 3).  With m = i + tt + 2 the multiloop sum of cell (i, j) is sum_m QM(i, m-1) * QM1(m, j),
// m = i + TURN + 2 .. j - TURN - 1: a matrix product.  Walked per cell and per diagonal (rounds 1-2) it reads two operands per
// term out of tables that no longer fit an L2 once a chip is full of strips: 171 MB per 400-nt fold, 43 GB per launch at R = 256,
// 0.67 of the HBM peak (profiles/r2/strips_pmc.json).  Now cells are grouped in TILES of 16 columns i x 16 columns j (strip-local
// coordinates, tile row t = (i-1) >> 4, tile column bj = (j-1) >> 4, block distance B = bj - t; first cell due at diagonal
// d_min = 16 B - 15).  The split points
//        m_lo = 16 t + 31 + PKT_L  <=  m  <=  16 bj - 13 - PKT_L = m_hi                                    (the FAR range of the tile)
// have, for EVERY cell of the tile, both operands on diagonals <= d_min - PKT_L, the ones further inside earlier still (one
// diagonal per split point).  Their part of the product is a dense 16 x (m_hi - m_lo + 1) x 16 product: a TILE WAVE computes it
// with v_mfma_f64_16x16x4_f64 in chunks of 4 split points (one operand double per lane: 2 loads per 1024 terms instead of 2 per
// term), spread over the PKT_W steps before d_min FROM THE MIDDLE OUTWARD -- step d_min - 1 - e takes the chunks e cl .. (e+1) cl - 1
// counted from the low end and e ch .. (e+1) ch - 1 from the high end, so every chunk is taken when its operands are final and
// visible (diagonals <= step - 2; PKT_L >= 3 suffices: tools/pkt_schedule.py checks every block distance) -- accumulators in
// registers across the steps, the 256 sums stored to the table DFAR at the end of the window.  The per-diagonal multiloop items
// keep only the NEAR split points of their cells (m < m_lo or m > m_hi: at most 24 + PKT_L on either side, from rows written or
// read a few steps ago) and start from the DFAR entry.  Every sum has one fixed order of additions: Epf is reproducible.
#ifndef PSTRIP_FARK
#define PSTRIP_FARK 1
#endif
#ifndef PSTRIP_SKIP
#define PSTRIP_SKIP 0        // diagnostic builds only (timing; results wrong): 1 no multiloop items, 2 no bulge / 1xn items, 4 no small shapes, 8 no towers, 16 no tile products
#endif
// (measured and dropped, DESIGN 3.8: tile rows on the tower waves with the tower step under the operands' round trip -- 100
// spilled VGPRs, 2.5 vs 1.57 ms at 400 nt; the step's first item between a floating wave's tile loads and its products -- 36
// spills, 1.72 vs 1.59 ms; tile rows on the floating waves only -- 1.61 vs 1.58 ms)
constexpr int PKT_DEPTH = 8;                            // chunks of a tile step in flight (two loads each)
constexpr int PKT_NB = 4;                               // near split points per lane and round trip
#ifndef DRNA_PKT_W
#define DRNA_PKT_W 16
#endif
#ifndef DRNA_PKT_L
#define DRNA_PKT_L 4
#endif
constexpr int PKT_W = DRNA_PKT_W;                       // steps a tile product is spread over (<= 16: the windows of consecutive block distances do not overlap)
constexpr int PKT_L = DRNA_PKT_L;                       // the far range's operands are final this many diagonals before d_min
constexpr int PKT_BMIN = PSTRIP_FARK ? (44 + 2 * PKT_L + 15) / 16 : (1 << 20);   // smallest block distance with a far range
static_assert(PKT_W >= 1 && PKT_W <= 16, "tile windows must not overlap");
static_assert(PKT_L >= 3, "a chunk must be final and visible when its step comes");

template <int NT>
struct PfStripSmem {
  static constexpr int NW = NT / WAVE;
  static constexpr int P = NT >= 1024 ? 128 : 64;        // physical tower lanes
  static constexpr int WMAX = P - 8;                     // widest strip
  static constexpr int RS = WMAX + 34;                   // ring row pitch: own columns 1..wid, halo wid+1..wid+32
  static constexpr int NL = WMAX + 8;
  static constexpr int NFIN = P / WAVE;                  // finalize waves
  static constexpr int NSVC = NT >= 1024 ? 2 : 0;        // service waves (halo + tables, list + exterior column); else finalize wave 0
  double qbi[33 * RS];
  double dring[4 * RS];
  double qm1row[2][RS];
  double urow[2][RS];
  double hpw[STRIP_NMAX + 2];
  double q5[STRIP_NMAX + 2];
  double partG[2][PNG][P];
  double partK[2][8][P];             // multiloop sums, one slice per split-point group (up to 8)
  double accE[2][P], accX[2][3][P];
  double gimp[2][PNG][PGSLOTS + 2];  // tower sums of the tower that enters the strip, staged by the service wave
  double stack[64], mmH[128], mmI[128], mm1n[128], mm23[128], mmM[128], mmExt[128], int11[1024], d5[32], d3[32];
  double rinv[128], rbul[128], r1n[128], r23[128];
  double eW[128];
  int eshape[128];
  double xc[8];
  double tw_as[32], tw_W[32];
  double tw_d[2][32][3];
  int tw_i[2][32][2];
  int plist[2][NL];
  int pcnt[2];
  int qk[2], qe[2];              // work-queue heads of the diagonal: multiloop items | shape items
  unsigned char info[33 * RS];
  unsigned char S[STRIP_NMAX + 4];
  int flag;
  int sync_fail[2];      // by step parity: set by the service wave during step k, read by everybody after the barrier of step k
};

// one diagonal step of a tower wave: import of the tower that enters the strip, the recurrence, export of the one that leaves
template <class SM>
__device__ __forceinline__ void strip_tower(SM& sm, double (&G)[PGSLOTS], int d, int wid, int n_loc, int phys, int my_tb, int my_g,
                                            int lane, bool has_up, bool has_down, double* rec_out) {
  constexpr int P = SM::P;
  const int ncell = min(wid, n_loc - d), sh = d >> 1, par = d & 1;
  const int iraw = ((phys - sh - 1) & (P - 1)) + 1;
  const bool live = iraw <= ncell;
  const int i = live ? iraw : 1;
  // the tower that enters the strip at its last column continues the one that left the strip above two diagonals ago
  const int pe = (wid + sh) & (P - 1);
  if (has_up && ncell == wid && d - 2 > TURN && (pe >> 6) == my_tb) {
    const bool mine = lane == (pe & (WAVE - 1));
#pragma unroll
    for (int qx = 0; qx < PGSLOTS; qx++) {
      const double v = sm.gimp[par][my_g][qx];
      G[qx] = mine ? v : G[qx];
    }
  }
  const double accG = (PSTRIP_SKIP & 8) ? 0.0 : pf_tower_step(sm, G, par, i * 8, my_g, lane);
  if (live) sm.partG[par][my_g][phys] = accG;
  if (has_down && live && iraw == 1) {               // the tower leaves the strip: its sums go into the record
    double* rec = rec_out + (long long)d * STRIP_REC + 48 + my_g * PGSLOTS;
#pragma unroll
    for (int qx = 0; qx < PGSLOTS; qx++) st_agent(rec + qx, G[qx]);
  }
}

// one strip of one sequence.  q = sequence slot of the launch, s = strip (0 = highest columns)
template <int NT>
__device__ void pf_strip_body(PfStripSmem<NT>& sm, PfArgs A, StripLink lk, int q, int s) {
  using SM = PfStripSmem<NT>;
  constexpr int NW = SM::NW, RS = SM::RS, P = SM::P, NFIN = SM::NFIN, NSVC = SM::NSVC;
  const PfTables& T = *A.T;
  const int r = lk.idx ? lk.idx[q] : q + lk.r0;
  if (A.rg.len) A.L = A.rg.len[r];
  const int n = A.L, ld = A.ld;
  const int tid = threadIdx.x, lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(wave_id());
  // strips of this sequence: a launch is made for S strips; shorter sequences of a ragged batch may need fewer
  const int S = lk.S;
  int c0, c1;
  strip_bounds(n, S, s, c0, c1);
  if (c0 > n) return;
  const int wid = c1 - c0 + 1, n_loc = n - c0 + 1;
  const bool has_up = c1 < n, has_down = c0 > 1;
  const int n_loc_up = n_loc - wid;                       // suffix length of the strip above
  int* const my_flag = lk.flags + ((long long)q * STRIP_MAXS + s) * 32;
  if (lk.fault && s == 0 && c1 == n && c0 > 1) {        // injected fault: the strips below see FAIL, the engine falls back
    if (threadIdx.x == 0) st_agent(my_flag, lk.base + STRIP_FAIL);
    return;
  }
  const int* const up_flag = lk.flags + ((long long)q * STRIP_MAXS + (s > 0 ? s - 1 : 0)) * 32;

  double* base = A.ws + (long long)r * A.ws_stride;
  const long long tab = (long long)ld * ld;
  double* XR = base;                                                     // exchange records (tables 0 and 1)
  double* QM = base + 2 * tab;
  double* QM1 = base + 3 * tab;
  int32_t* PL = reinterpret_cast<int32_t*>(base + 4 * tab);              // pairable lists: row d, columns of the strip
  int32_t* PLC = PL + tab;                                               // their counts: [d * 8 + s]
  double* QEXT = base + 6 * tab;
  double* const rec_out = XR + (long long)s * ld * STRIP_REC;            // records this strip writes (row d)
  const double* const rec_in = XR + (long long)(s > 0 ? s - 1 : 0) * ld * STRIP_REC;

  const double eTau = T.TermAU, eMLc = T.MLclosing, eMLi = T.MLintern;
  const double b1 = A.eMLb[1], sc1 = A.scale[1], sc2 = A.scale[2];

  // ---- prologue (as in pf_lds_kernel)
  for (int k = tid; k < 64; k += NT) sm.stack[k] = T.stack[k];
  for (int k = tid; k < 128; k += NT) {
    sm.mmH[k] = T.mmH[k]; sm.mmI[k] = T.mmI[k]; sm.mm1n[k] = T.mm1n[k];
    sm.mm23[k] = T.mm23[k]; sm.mmM[k] = T.mmM[k]; sm.mmExt[k] = T.mmExt[k];
    const double inv = 1.0 / T.mmI[k];
    sm.rinv[k] = inv;
    sm.rbul[k] = ((k >> 4) > 2 ? eTau : 1.0) * inv;
    sm.r1n[k] = T.mm1n[k] * inv;
    sm.r23[k] = T.mm23[k] * inv;
  }
  for (int k = tid; k < 1024; k += NT) sm.int11[k] = T.int11[k];
  for (int k = tid; k < 32; k += NT) { sm.d5[k] = T.d5[k]; sm.d3[k] = T.d3[k]; }
  for (int k = tid; k <= n; k += NT) sm.hpw[k] = A.hp_w[k];
  for (int x = tid; x < 128; x += NT) {
    int s_, u1_;
    double W = 0.0;
    if (x < 64) {
      const bool on = x < 58;
      u1_ = (x < 29 || !on) ? 0 : x - 27;
      s_ = !on ? 2 : x < 29 ? x + 2 : x - 27;
      if (on) W = T.bulge[s_] * A.scale[s_ + 2];
    } else {
      const int y = x - 64;
      const bool on = y < 54;
      u1_ = (y < 27 || !on) ? 1 : y - 24;
      s_ = !on ? 4 : y < 27 ? y + 4 : y - 23;
      if (on) W = T.interior[s_] * T.eninio[s_ - 2] * A.scale[s_ + 2];
    }
    sm.eshape[x] = s_ | (u1_ << 8);
    sm.eW[x] = W;
  }
  if (tid == 0) {
    sm.xc[0] = T.bulge[1] * A.scale[3]; sm.xc[1] = T.interior[5] * T.eninio[1] * A.scale[7];
    sm.xc[2] = A.scale[4]; sm.xc[3] = A.scale[5]; sm.xc[4] = A.scale[6];
  }
  for (int k = tid; k < 32; k += NT) {
    sm.tw_as[k] = k >= 4 && k <= 30 ? T.eninio[k - 4] : 0.0;
    sm.tw_W[k] = k >= 6 && k <= 30 ? T.interior[k] * A.scale[k + 2] : 0.0;
  }
  for (int k = tid; k < 4 * RS; k += NT) sm.dring[k] = 0.0;
  for (int k = tid; k < 33 * RS; k += NT) { sm.qbi[k] = 0.0; sm.info[k] = 0; }
  for (int k = tid; k < 2 * RS; k += NT) { (&sm.qm1row[0][0])[k] = 0.0; (&sm.urow[0][0])[k] = 0.0; }
  for (int k = tid; k < 2 * PNG * P; k += NT) (&sm.partG[0][0][0])[k] = 0.0;
  for (int k = tid; k < 2 * 8 * P; k += NT) (&sm.partK[0][0][0])[k] = 0.0;
  for (int k = tid; k < 2 * P; k += NT) (&sm.accE[0][0])[k] = 0.0;
  for (int k = tid; k < 6 * P; k += NT) (&sm.accX[0][0][0])[k] = 0.0;
  for (int k = tid; k < 2 * PNG * (PGSLOTS + 2); k += NT) (&sm.gimp[0][0][0])[k] = 0.0;
  if (tid == 0) { sm.flag = 0; sm.sync_fail[0] = 0; sm.sync_fail[1] = 0; sm.q5[0] = 1.0; }
  __syncthreads();
  // local sequence: S[k] = residue c0 - 1 + k, k = 0 .. n_loc + 1 (the ends wrap as in the one-workgroup kernels)
  const char* seq = A.seqs + (A.rg.off ? (long long)A.rg.off[r] : (long long)r * n);
  for (int k = tid; k <= n_loc + 1; k += NT) {
    int g = c0 - 1 + k;                     // 1-based global position
    g = g < 1 ? n : (g > n ? 1 : g);
    const int c = enc_nt(seq[g - 1]);
    if (c < 0) sm.flag = 1;
    sm.S[k] = (unsigned char)(c < 0 ? 0 : c);
  }
  __syncthreads();
  const bool last = !has_down;              // the strip that owns column 1: exterior column, Z, status
  if (last && sm.flag) {
    if (tid == 0) { A.status[r] = ST_BAD_CHAR; A.Epf[r] = 0.0; }
    return;
  }
  if (wave == 0 && last) {
    for (int j = 1; j <= n && j <= TURN + 1; j++) sm.q5[j] = sm.q5[j - 1] * sc1;
  }
  // compacted list of the strip's pairable cells of every diagonal
  for (int d = TURN + 1 + wave; d < n_loc; d += NW) {
    const int nc = min(wid, n_loc - d);
    int cntb = 0;
    for (int i0 = 1; i0 <= nc; i0 += WAVE) {
      const int i = i0 + lane;
      int t = 0;
      if (i <= nc) t = pair_type(sm.S[i], sm.S[i + d]);
      const unsigned long long m = __ballot(t != 0);
      if (t) {
        const int pos = cntb + __popcll(m & ((1ull << lane) - 1ull));
        PL[d * ld + c0 - 1 + pos] = i | ((t * 16 + sm.S[i + 1] * 4 + sm.S[i + d - 1]) << 8);
      }
      cntb += __popcll(m);
    }
    if (lane == 0) PLC[d * STRIP_MAXS + s] = cntb;
  }
  __syncthreads();

  // roles
  const int w_svcA = NSVC ? NFIN : 0, w_svcB = NSVC ? NFIN + 1 : 0;       // halo + tower tables | pairable list + exterior column
  const int aw = wave - NFIN - NSVC;                                     // index among the sweep waves
  const int my_tb = aw >= 0 ? aw / PNG : NFIN, my_g = aw >= 0 ? aw - my_tb * PNG : 0;
  const bool pinned = aw >= 0 && my_tb < NFIN;
  const bool fin = wave < NFIN;

  if (wave == w_svcA) {
    const int d = TURN + 1;
    if (d < n_loc) pf_prepare_tables(sm, d, lane);
  }
  if (wave == w_svcB) {
    const int d = TURN + 1;
    if (d < n_loc) {
      const int cnt = PLC[d * STRIP_MAXS + s];
      for (int k = lane; k < cnt; k += WAVE) sm.plist[d & 1][k] = PL[d * ld + c0 - 1 + k];
      if (lane == 0) { sm.pcnt[d & 1] = cnt; sm.qk[0] = 0; sm.qk[1] = 0; sm.qe[0] = 0; sm.qe[1] = 0; }
    }
  }
  __syncthreads();

  const auto rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)QM, (short)0, (int)(2 * tab * 8), 0x00020000);
  const auto rsX = __builtin_amdgcn_make_buffer_rsrc((void*)QEXT, (short)0, (int)(tab * 8), 0x00020000);
  double* const DFAR = base + 5 * tab;                                   // far parts of the multiloop sums, [d][global column]
  const auto rsF = __builtin_amdgcn_make_buffer_rsrc((void*)DFAR, (short)0, (int)(tab * 8), 0x00020000);

  // floating work items of diagonal d (see pf_lds_kernel): multiloop sums from L2 (the qm1 operand may be another strip's: sc1),
  // bulge / 1xn shapes, fixed small shapes.  Output slots are physical tower lanes (i_loc + d/2) mod P.
  auto run_items = [&](const int d, auto with_k) {
    const int ncell = min(wid, n_loc - d), sh = d >> 1, par = d & 1;
    const int pcnt = __builtin_amdgcn_readfirstlane(sm.pcnt[par]);
    // near split points of the diagonal (tile geometry: see PKT_L): tt = m - 1 - i runs over [TURN+1, 28+PKT_L] and
    // [d-29-PKT_L, d-TURN-2], every cell masks what belongs to its tile's far range; one range while cells without a far range
    // exist on the diagonal (d < 16 PKT_BMIN) or the two meet
    const int tt_end = d - TURN - 2;
    const bool whole = d < 16 * PKT_BMIN || 28 + PKT_L + 1 >= d - 29 - PKT_L;
    const int n1 = whole ? tt_end - TURN : 28 + PKT_L - TURN;                 // terms of the first range
    const int s2 = d - 29 - PKT_L;                                            // first tt of the second
    const int nterm = whole ? n1 : n1 + tt_end - s2 + 1;
    // a block of 32 cells is dealt to KS = 1, 2, 4 or 8 items by split point, so that no wave walks a long sum as one chain of
    // dependent L2 round trips while the others idle
    int kssh = nterm > 192 ? 3 : nterm > 96 ? 2 : nterm > 48 ? 1 : 0;
    if (ncell <= 32 && kssh < 2) kssh = 2;
    const int KS = 1 << kssh, KG = 4 << kssh;
    const int nK = (PSTRIP_SKIP & 1) ? 0 : ((ncell + 31) >> 5) << kssh, nE = (PSTRIP_SKIP & 2) ? 0 : (pcnt + 3) >> 2,
              nX = (PSTRIP_SKIP & 4) ? 0 : 3 * ((pcnt + WAVE - 1) / WAVE);
    const int nItems = __builtin_amdgcn_readfirstlane(nK + nE + nX);
    // two queues: the multiloop items (16 loads in flight per lane: only waves that hold no tower sums take them), then the shape items
    auto pop = [&]() -> int {
      if (decltype(with_k)::value) {
        const int it = queue_pop(&sm.qk[par], lane);
        if (it < nK) return it;
      }
      return nK + queue_pop(&sm.qe[par], lane);
    };
    for (int it = pop(); it < nItems; it = pop()) {
      if (decltype(with_k)::value && it < nK) {
        const int g = (it & (KS - 1)) * 4 + (lane >> 4), cl = lane & 15;
        int i = ((it >> kssh) << 5) + 2 * cl + 1;
        const bool act0 = i <= ncell, act1 = i + 1 <= ncell;
        i = act0 ? i : 1;
        const int ig = i + c0 - 1;                                         // global column
        // the lane's two cells (i, i + d) and (i + 1, i + 1 + d): same tile row (i is odd), maybe different tile columns
        const int t16 = (i - 1) & ~15, bj0 = (i + d - 1) & ~15, bj1 = (i + d) & ~15;
        const bool far0 = bj0 - t16 >= 16 * PKT_BMIN, far1 = bj1 - t16 >= 16 * PKT_BMIN;
        const int tl0 = far0 ? t16 + 29 + PKT_L - i : d, th0 = far0 ? bj0 - 13 - PKT_L - i : d + 1;
        const int tl1 = far1 ? t16 + 28 + PKT_L - i : d, th1 = far1 ? bj1 - 14 - PKT_L - i : d + 1;
        const bool head = (it & (KS - 1)) == 0 && lane < 16;              // the lanes that write slice 0 add the far part
        f64x2 fv{0.0, 0.0};
        if (head && (far0 || far1)) fv = buf_load_f64x2_sc1(rsF, (d * ld + ig) * 8, 0);
        double p0 = 0.0, p1 = 0.0, q0 = 0.0, q1 = 0.0;
        constexpr int NB = PKT_NB;        // terms per lane and round trip
        for (int x = g; x < nterm; x += NB * KG) {
          f64x2 a[NB], c[NB];
          int tt[NB];
#pragma unroll
          for (int u = 0; u < NB; u++) {
            const int xu = min(x + u * KG, nterm - 1);
            tt[u] = xu < n1 ? TURN + 1 + xu : s2 + (xu - n1);
            a[u] = buf_load_f64x2(rsQ, (tt[u] * ld + ig) * 8, 0);
            c[u] = buf_load_f64x2_sc1(rsQ, (int)tab * 8 + ((d - tt[u] - 1) * ld + ig + tt[u] + 1) * 8, 0);
          }
#pragma unroll
          for (int u = 0; u < NB; u++) {
            const bool in = x + u * KG < nterm;
            const bool k0 = in && (tt[u] <= tl0 || tt[u] >= th0), k1 = in && (tt[u] <= tl1 || tt[u] >= th1);
            const double e0 = k0 ? a[u].x * c[u].x : 0.0, e1 = k1 ? a[u].y * c[u].y : 0.0;
            if (u & 1) { p1 += e0; q1 += e1; } else { p0 += e0; q0 += e1; }
          }
        }
        double v0 = p0 + p1, v1 = q0 + q1;
        int slice = lane >> 4;
        bool writer = true;
        if (kssh >= 1) {                       // rows 0+1 and 2+3
          v0 += __shfl_xor(v0, 16); v1 += __shfl_xor(v1, 16);
          slice = (it & 1) * 2 + (lane >> 5);
          writer = (lane & 16) == 0;
        }
        if (kssh >= 2) {                       // all four rows
          v0 += __shfl_xor(v0, 32); v1 += __shfl_xor(v1, 32);
          slice = it & (KS - 1);
          writer = lane < 16;
        }
        if (head) { v0 += far0 ? fv.x : 0.0; v1 += far1 ? fv.y : 0.0; }
        if (writer) {
          if (act0) sm.partK[par][slice][(i + sh) & (P - 1)] = v0;
          if (act1) sm.partK[par][slice][(i + 1 + sh) & (P - 1)] = v1;
        }
      } else if (it < nK + nE) {
        const int qq = 4 * (it - nK) + (lane >> 4);
        const int pe = sm.plist[par][qq < pcnt ? qq : pcnt - 1];
        const int i0 = pe & 255, ij = pe >> 8;
        double acc[2];
#pragma unroll
        for (int half = 0; half < 2; half++) {
          int shp[4], off[4], f[4];
          double w[4], rr[4], ww[4];
#pragma unroll
          for (int k = 0; k < 4; k++) { shp[k] = sm.eshape[half * 64 + k * 16 + (lane & 15)]; ww[k] = sm.eW[half * 64 + k * 16 + (lane & 15)]; }
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int dp = d - 2 - (shp[k] & 255);
            off[k] = (dp > TURN ? (dp & 31) * RS + 1 + (shp[k] >> 8) : 32 * RS) + i0;
          }
#pragma unroll
          for (int k = 0; k < 4; k++) { w[k] = sm.qbi[off[k]]; f[k] = sm.info[off[k]]; }
#pragma unroll
          for (int k = 0; k < 4; k++) rr[k] = half ? sm.r1n[f[k]] : sm.rbul[f[k]];
          double a = 0.0;
#pragma unroll
          for (int k = 0; k < 4; k++) a += w[k] * rr[k] * ww[k];
          acc[half] = a;
        }
        double v = acc[0] * ((ij >> 4) > 2 ? eTau : 1.0) + acc[1] * sm.mm1n[ij];
        v = dpp_add_f64<0x111, 0xF>(v);
        v = dpp_add_f64<0x112, 0xF>(v);
        v = dpp_add_f64<0x114, 0xF>(v);
        v = dpp_add_f64<0x118, 0xF>(v);
        if ((lane & 15) == 15 && qq < pcnt) sm.accE[par][(i0 + sh) & (P - 1)] = v;
      } else {
        const int xi = it - nK - nE;
        const int ch = xi / 3, grp = xi - 3 * ch;
        const int qq = ch * WAVE + lane;
        const int pe = sm.plist[par][qq < pcnt ? qq : pcnt - 1];
        const int i = pe & 255, cxv = pe >> 8, t = cxv >> 4, si1 = (cxv >> 2) & 3, sj1 = cxv & 3;
        double sum;
        if (grp == 0) {
          double w[4];
          int f[4];
#pragma unroll
          for (int shp = 0; shp < 4; shp++) {
            const int u1 = shp >> 1, u2 = shp == 1 || shp == 3 ? 1 : 0;
            const int dp = d - 2 - u1 - u2;
            const int off = (dp & 31) * RS + 1 + u1 + i;
            w[shp] = dp > TURN ? sm.qbi[off] : 0.0;
            f[shp] = dp > TURN ? sm.info[off] : 0;
          }
          sum = w[0] * sm.rinv[f[0]] * sm.stack[t * 8 + (f[0] >> 4)] * sc2;
          sum += (w[1] * sm.rinv[f[1]] * sm.stack[t * 8 + (f[1] >> 4)] + w[2] * sm.rinv[f[2]] * sm.stack[t * 8 + (f[2] >> 4)]) * sm.xc[as_vector(0)];
          sum += w[3] * sm.rinv[f[3]] * sm.int11[(t * 8 + (f[3] >> 4)) * 16 + si1 * 4 + sj1] * sm.xc[as_vector(2)];
        } else if (grp == 1) {
          const int dpa = d - 5, dpb = d - 6;
          const int oa = (dpa & 31) * RS + 2 + i, ob = (dpa & 31) * RS + 3 + i, oc = (dpb & 31) * RS + 3 + i;
          const double wa = dpa > TURN ? sm.qbi[oa] : 0.0, wb = dpa > TURN ? sm.qbi[ob] : 0.0, wc = dpb > TURN ? sm.qbi[oc] : 0.0;
          const int fa = dpa > TURN ? sm.info[oa] : 0, fb = dpa > TURN ? sm.info[ob] : 0, fc = dpb > TURN ? sm.info[oc] : 0;
          const double ga = T.int21[(t * 8 + (fa >> 4)) * 64 + si1 * 16 + ((fa >> 2) & 3) * 4 + sj1];
          const double gb = T.int21[((fb >> 4) * 8 + t) * 64 + ((fb >> 2) & 3) * 16 + si1 * 4 + (fb & 3)];
          const double gc = T.int22[(t * 8 + (fc >> 4)) * 256 + si1 * 64 + (fc & 3) * 16 + ((fc >> 2) & 3) * 4 + sj1];
          sum = (wa * sm.rinv[fa] * ga + wb * sm.rinv[fb] * gb) * sm.xc[as_vector(3)] + wc * sm.rinv[fc] * gc * sm.xc[as_vector(4)];
        } else {
          const int dp = d - 7;
          const int oa = (dp & 31) * RS + 3 + i, ob = (dp & 31) * RS + 4 + i;
          const double wa = dp > TURN ? sm.qbi[oa] : 0.0, wb = dp > TURN ? sm.qbi[ob] : 0.0;
          const int fa = dp > TURN ? sm.info[oa] : 0, fb = dp > TURN ? sm.info[ob] : 0;
          sum = (wa * sm.r23[fa] + wb * sm.r23[fb]) * sm.mm23[cxv] * sm.xc[as_vector(1)];
        }
        if (qq < pcnt) sm.accX[par][grp][(i + sh) & (P - 1)] = sum;
      }
    }
  };

  // ---- service jobs of step k (one wave each when the workgroup has service waves, else finalize wave 0 after its cells)
  // A: wait for the strip above to have published diagonal k-1, stage its record (ring halo of row k-1, qm1 / U / D of its
  //    first column, the sums of the tower that enters at diagonal k+1); tower table of diagonal k+1
  auto service_a = [&](const int k) {
    if (has_up && k - 1 > TURN && k - 1 <= n_loc_up - 1) {
      int seen = 0;
      if (!strip_wait(up_flag, lk.base, k - 1, seen)) {
        sm.sync_fail[k & 1] = 1;
        if (lane == 0 && lk.dbg) { int* g = lk.dbg + q * 8; g[0] = s + 1; g[1] = k; g[2] = seen; g[3] = lk.base; g[4] = (int)blockIdx.x; g[5] = n; }
      }
      else {
        const int dd = k - 1;
        const double* rec = rec_in + (long long)dd * STRIP_REC;
        const double h = lane < 32 ? ld_agent(rec + lane) : 0.0;
        const int fi = lane < 32 ? ld_agent(reinterpret_cast<const int*>(rec + 32) + lane) : 0;
        const double gv = lane < PNG * PGSLOTS ? ld_agent(rec + 48 + lane) : 0.0;
        const double sv = lane < 3 ? ld_agent(rec + 78 + lane) : 0.0;
        if (lane < 32) { sm.qbi[(dd & 31) * RS + wid + 1 + lane] = h; sm.info[(dd & 31) * RS + wid + 1 + lane] = (unsigned char)fi; }
        if (lane < PNG * PGSLOTS) sm.gimp[(k + 1) & 1][lane / PGSLOTS][lane % PGSLOTS] = gv;
        if (lane == 0) sm.qm1row[dd & 1][wid + 1] = sv;
        if (lane == 1) sm.urow[dd & 1][wid + 1] = sv;
        if (lane == 2) sm.dring[(dd & 3) * RS + wid + 1] = sv;
      }
    }
    if (k + 1 < n_loc) pf_prepare_tables(sm, k + 1, lane);
  };
  // B: pairable list of diagonal k+1; exterior column j = k-3 (last strip)
  // (every load of the job is issued before the first is used, the column's cells first; the list of diagonal k+2 -- rows
  // written by the prologue long ago, i.e. from HBM -- is requested one step ahead and stays in flight behind them)
  int lp_cnt = 0, lp_p0 = 0, lp_p1 = 0;
  if (wave == w_svcB && TURN + 2 < n_loc) {
    const int32_t* row = PL + (TURN + 2) * ld + c0 - 1;
    lp_cnt = PLC[(TURN + 2) * STRIP_MAXS + s];
    lp_p0 = row[lane]; lp_p1 = row[min(lane + WAVE, wid)];
  }
  auto service_b = [&](const int k) {
    constexpr int NFX = 15;       // chunks of the column held at once (960 cells); longer columns continue in batches of 8
    const bool do_list = k + 1 < n_loc, do_q5 = last && k - 3 >= TURN + 2;
    const int dn = k + 1, j = k - 3;
    const int cnt = lp_cnt, p0 = lp_p0, p1 = lp_p1;
    const int fcnt = j - TURN - 1, nch = do_q5 ? (fcnt + WAVE - 1) >> 6 : 0;
    double fx[NFX];
#pragma unroll
    for (int c = 0; c < NFX; c++) {
      fx[c] = 0.0;
      if (c < nch) fx[c] = buf_load_f64_aux(rsX, (j * ld + min(lane + 1 + c * WAVE, fcnt)) * 8, 0);
    }
    if (k + 2 < n_loc) {
      const int32_t* row = PL + (k + 2) * ld + c0 - 1;
      lp_cnt = PLC[(k + 2) * STRIP_MAXS + s];
      lp_p0 = row[lane]; lp_p1 = row[min(lane + WAVE, wid)];
    }
    if (do_list) {
      int* dst = sm.plist[dn & 1];
      dst[lane] = p0;
      if (lane + WAVE < SM::NL) dst[lane + WAVE] = p1;
      if (lane == 0) { sm.pcnt[dn & 1] = cnt; sm.qk[dn & 1] = 0; sm.qe[dn & 1] = 0; }
    }
    if (do_q5) {
      double sacc = 0.0;
#pragma unroll
      for (int c = 0; c < NFX; c++) {
        const int i = lane + 1 + c * WAVE;
        if (c < nch && i <= fcnt) sacc += sm.q5[i - 1] * fx[c];
      }
      for (int cb = NFX; cb < nch; cb += 8) {                     // (columns beyond 960 cells; same order of additions)
        double gx[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          gx[u] = 0.0;
          if (cb + u < nch) gx[u] = buf_load_f64_aux(rsX, (j * ld + min(lane + 1 + (cb + u) * WAVE, fcnt)) * 8, 0);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int i = lane + 1 + (cb + u) * WAVE;
          if (cb + u < nch && i <= fcnt) sacc += sm.q5[i - 1] * gx[u];
        }
      }
      sacc = wave_sum_f64(sacc);
      sm.q5[j] = sm.q5[j - 1] * sc1 + sacc;
    }
  };

  // ---- tile products (see PKT_L above).  Tile waves: the floating and the service waves, one tile row each (the finalize waves of a
  // workgroup that has neither); a tile wave requests its chunks' operands (tile_issue) and multiplies (tile_finish).  Tile wave
  // f owns the tile rows f, f + NTW, ...  At step k the tiles of block distance B = (k + 15 + PKT_W) >> 4 are in step
  // g = (k + 15 + PKT_W) & 15 of their window (g < PKT_W; k = d_min - PKT_W + g): operands on diagonals <= d_min - PKT_L <= k - 2.
  constexpr int NFLOAT = NW - NFIN - NSVC - NFIN * PNG;
  constexpr int NTW = NFLOAT > 0 ? NFLOAT + NSVC : NFIN;
  constexpr int TOWN = ((SM::WMAX + 15) / 16 + NTW - 1) / NTW;
  const int tf = NFLOAT > 0 ? (aw >= 0 ? aw - NFIN * PNG : NFLOAT + wave - NFIN) : wave;
  f64x4 tacc[TOWN];
#pragma unroll
  for (int o = 0; o < TOWN; o++) tacc[o] = f64x4{0.0, 0.0, 0.0, 0.0};
  double tq_a[TOWN][PKT_DEPTH], tq_b[TOWN][PKT_DEPTH];   // operands requested by tile_issue
  struct TileStep { int t, bj, m_lo, m_hi, nch, lo0, nlo, hi0, ncs, oA, oB; bool on; };
  auto tile_step = [&](const int k, const int o, int& g) -> TileStep {
    TileStep q;
    const int x = k + 15 + PKT_W, B = x >> 4;
    g = x & 15;
    q.t = tf + NTW * o; q.bj = q.t + B;
    q.on = PSTRIP_FARK && !(PSTRIP_SKIP & 16) && g < PKT_W && B >= PKT_BMIN && 16 * q.t < wid && 16 * q.bj + 1 <= n_loc;   // (wave-uniform)
    q.m_lo = 16 * q.t + 31 + PKT_L; q.m_hi = 16 * q.bj - 13 - PKT_L;
    q.nch = (q.m_hi - q.m_lo + 4) >> 2;
    const int nl = (q.nch + 1) >> 1, nh = q.nch >> 1;
    const int cl = (nl + PKT_W - 1) / PKT_W, ch = (nh + PKT_W - 1) / PKT_W, e = PKT_W - 1 - g;
    q.lo0 = e * cl; q.nlo = max(0, min(nl, q.lo0 + cl) - q.lo0); q.hi0 = e * ch;
    const int nhi = max(0, min(nh, q.hi0 + ch) - q.hi0);
    q.ncs = q.nlo + nhi;                                                      // chunks of this step: low side first
    // rows / columns beyond the strip or the sequence repeat the last one: their sums are never stored
    const int r = lane & 15;
    const int il = min(16 * q.t + 1 + r, wid), jl = min(16 * q.bj + 1 + r, n_loc);
    q.oA = ((-1 - il) * ld + c0 - 1 + il) * 8; q.oB = (int)tab * 8 + (jl * ld + c0 - 1) * 8;     // + m * (ld * 8)  |  - m * (ld - 1) * 8
    return q;
  };
  auto tile_load = [&](const TileStep& q, const int c, double& a, double& b) {
    const int cid = c < q.nlo ? q.lo0 + c : q.nch - 1 - (q.hi0 + c - q.nlo);
    const int m = q.m_lo + 4 * cid + (lane >> 4), mc = min(m, q.m_hi);
    a = buf_load_f64(rsQ, q.oA + mc * (ld * 8), 0);                 // QM(i, m - 1): own columns
    b = buf_load_f64_aux(rsQ, q.oB - mc * ((ld - 1) * 8), 0);       // QM1(m, j): maybe another strip's
    if (m > q.m_hi) a = 0.0;
  };
  auto tile_issue = [&](const int k) {
#pragma unroll
    for (int o = 0; o < TOWN; o++) {
      int g;
      const TileStep q = tile_step(k, o, g);
#pragma unroll
      for (int u = 0; u < PKT_DEPTH; u++) {
        tq_a[o][u] = 0.0; tq_b[o][u] = 0.0;
        if (q.on && u < q.ncs) tile_load(q, u, tq_a[o][u], tq_b[o][u]);
      }
    }
  };
  auto tile_finish = [&](const int k) {
#pragma unroll
    for (int o = 0; o < TOWN; o++) {
      int g;
      const TileStep q = tile_step(k, o, g);
      if (!q.on) continue;
      f64x4 acc = tacc[o];
      if (g == 0) acc = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int u = 0; u < PKT_DEPTH; u++)
        if (u < q.ncs) acc = mfma_f64_16x16x4(tq_a[o][u], tq_b[o][u], acc);
      for (int c = PKT_DEPTH; c < q.ncs; c += PKT_DEPTH) {                    // (long folds: more chunks per step than ride in registers)
        double a[PKT_DEPTH], b[PKT_DEPTH];
#pragma unroll
        for (int u = 0; u < PKT_DEPTH; u++) {
          a[u] = 0.0; b[u] = 0.0;
          if (c + u < q.ncs) tile_load(q, c + u, a[u], b[u]);
        }
#pragma unroll
        for (int u = 0; u < PKT_DEPTH; u++)
          if (c + u < q.ncs) acc = mfma_f64_16x16x4(a[u], b[u], acc);
      }
      tacc[o] = acc;
      if (g == PKT_W - 1) {
        const int r = lane & 15, kk = lane >> 4, jj = 16 * q.bj + 1 + r;
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          const int ii = 16 * q.t + 1 + kk + 4 * rr;
          if (ii <= wid && jj <= n_loc) DFAR[(long long)(jj - ii) * ld + c0 - 1 + ii] = acc[rr];
        }
      }
    }
  };
  auto tile_job = [&](const int k) { tile_issue(k); tile_finish(k); };

  bool failed = false;
  if (fin) {
    // ================= finalize waves: diagonal d = k-1 at step k
    for (int k = TURN + 1; k <= n_loc; k++) {
      const int d = k - 1;
      // everything this strip stored up to step k-1 has landed (the barrier that ended it drained every wave): diagonal k-2
      // (cells finalized in step k-1, tower exported in step k-2) is complete
      if (tid == 0 && has_down && k - 2 > TURN) st_agent(my_flag, lk.base + k - 2);
      if (d > TURN) {
        const int ncell = min(wid, n_loc - d), sh = d >> 1, par = d & 1;
        const int i = ((tid - sh - 1) & (P - 1)) + 1;
        if (i <= ncell) {
          const double aG = (sm.partG[par][0][tid] + sm.partG[par][1][tid]) + sm.partG[par][2][tid];
          const double aK = ((sm.partK[par][0][tid] + sm.partK[par][1][tid]) + (sm.partK[par][2][tid] + sm.partK[par][3][tid])) +
                            ((sm.partK[par][4][tid] + sm.partK[par][5][tid]) + (sm.partK[par][6][tid] + sm.partK[par][7][tid]));
          const double aE = sm.accE[par][tid], aX = (sm.accX[par][0][tid] + sm.accX[par][1][tid]) + sm.accX[par][2][tid];
          sm.partG[par][0][tid] = 0.0; sm.partG[par][1][tid] = 0.0; sm.partG[par][2][tid] = 0.0;
          sm.partK[par][0][tid] = 0.0; sm.partK[par][1][tid] = 0.0; sm.partK[par][2][tid] = 0.0; sm.partK[par][3][tid] = 0.0;
          sm.partK[par][4][tid] = 0.0; sm.partK[par][5][tid] = 0.0; sm.partK[par][6][tid] = 0.0; sm.partK[par][7][tid] = 0.0;
          sm.accE[par][tid] = 0.0; sm.accX[par][0][tid] = 0.0; sm.accX[par][1][tid] = 0.0; sm.accX[par][2][tid] = 0.0;
          const int j = i + d;                               // local
          const int ig = i + c0 - 1, jg = ig + d;            // global
          const int t = pair_type(sm.S[i], sm.S[j]);
          const double tau = t > 2 ? eTau : 1.0;
          double qb = 0.0;
          int info = 0;
          if (t) {
            const int u = d - 1;
            const int ij = t * 16 + sm.S[i + 1] * 4 + sm.S[j - 1];
            double hp;
            if (u == 3 || u == 4 || u == 6) {
              int code = 0;
              const int len = u + 2;
              for (int qx = 0; qx < len; qx++) code |= sm.S[i + qx] << (2 * qx);
              hp = -1.0;
              if (u == 3) { for (int qx = 0; qx < T.n_tri; qx++) if (T.tri_code[qx] == code) hp = T.tri_w[qx] * A.scale[u + 2]; if (hp < 0.0) hp = sm.hpw[u] * tau; }
              else if (u == 4) { for (int qx = 0; qx < T.n_tetra; qx++) if (T.tetra_code[qx] == code) hp = T.tetra_w[qx] * A.scale[u + 2]; }
              else { for (int qx = 0; qx < T.n_hexa; qx++) if (T.hexa_code[qx] == code) hp = T.hexa_w[qx] * A.scale[u + 2]; }
              if (hp < 0.0) hp = sm.hpw[u] * sm.mmH[ij];
            } else {
              hp = sm.hpw[u] * sm.mmH[ij];
            }
            qb = hp + aE + aX + aG * sm.mmI[ij];
            qb += sm.dring[((d - 2) & 3) * RS + i + 1] * eMLc * eMLi * tau *
                  sm.mmM[rtype_of(t) * 16 + sm.S[j - 1] * 4 + sm.S[i + 1]] * sc2;
            info = (rtype_of(t) << 4) | (sm.S[j + 1] << 2) | sm.S[i - 1];
          }
          const double qv = qb * sm.mmI[info];
          sm.qbi[(d & 31) * RS + i] = qv;
          sm.info[(d & 31) * RS + i] = (unsigned char)info;
          double ext = 0.0, stem = 0.0;
          if (t) {
            double me, mm;
            if (ig > 1 && jg < n) { me = sm.mmExt[t * 16 + sm.S[i - 1] * 4 + sm.S[j + 1]]; mm = sm.mmM[t * 16 + sm.S[i - 1] * 4 + sm.S[j + 1]]; }
            else if (ig > 1) { me = mm = sm.d5[t * 4 + sm.S[i - 1]]; }
            else if (jg < n) { me = mm = sm.d3[t * 4 + sm.S[j + 1]]; }
            else { me = mm = 1.0; }
            ext = qb * tau * me;
            stem = qb * eMLi * tau * mm;
          }
          strip_store(&QEXT[jg * ld + ig], ext);
          const int pp = (d - 1) & 1;
          const double m1 = sm.qm1row[pp][i] * b1 + stem;
          const double U = b1 * (sm.qm1row[pp][i + 1] + sm.urow[pp][i + 1]);
          sm.qm1row[par][i] = m1;
          sm.urow[par][i] = U;
          sm.dring[(d & 3) * RS + i] = aK;
          strip_store(&QM1[d * ld + ig], m1);
          QM[d * ld + ig] = m1 + aK + U;
          if (has_down && i <= 32) {                          // the record for the strip below
            double* rec = rec_out + (long long)d * STRIP_REC;
            st_agent(rec + (i - 1), qv);
            st_agent(reinterpret_cast<int*>(rec + 32) + (i - 1), info);
            if (i == 1) { st_agent(rec + 78, m1); st_agent(rec + 79, U); st_agent(rec + 80, aK); }
          }
        }
      }
      if (!NSVC && wave == 0) { service_a(k); service_b(k); }
      if (NFLOAT == 0 && k < n_loc) tile_job(k);
      if (k < n_loc) run_items(k, std::true_type{});            // help the sweep of diagonal k
      STRIP_BARRIER();
      if (sm.sync_fail[k & 1]) { failed = true; break; }
    }
  } else if (NSVC && wave < NFIN + NSVC) {
    // ================= service waves
    for (int k = TURN + 1; k <= n_loc; k++) {
      if (wave == w_svcA) service_a(k); else service_b(k);
      if (k < n_loc) tile_job(k);
      if (k < n_loc) run_items(k, std::true_type{});
      STRIP_BARRIER();
      if (sm.sync_fail[k & 1]) { failed = true; break; }
    }
  } else if (!pinned) {
    // ================= floating waves: items only
    for (int k = TURN + 1; k <= n_loc; k++) {
      if (k < n_loc) {
        tile_job(k);
        run_items(k, std::true_type{});
      }
      STRIP_BARRIER();
      if (sm.sync_fail[k & 1]) { failed = true; break; }
    }
  } else {
    // ================= tower waves (a block of 64 physical tower lanes each): diagonal d = k at step k, then shape items
    double GE[PGSLOTS], GO[PGSLOTS];
#pragma unroll
    for (int qx = 0; qx < PGSLOTS; qx++) { GE[qx] = 0.0; GO[qx] = 0.0; }
    const int phys = my_tb * WAVE + lane;
    auto tower = [&](const int d, double (&G)[PGSLOTS]) __attribute__((always_inline)) {
      strip_tower(sm, G, d, wid, n_loc, phys, my_tb, my_g, lane, has_up, has_down, rec_out);
    };
    // two steps per trip (TURN + 1 is even): each parity's sums are named at their own call site and stay in registers
    static_assert(((TURN + 1) & 1) == 0, "the loop below starts on an even diagonal");
    for (int k = TURN + 1; k <= n_loc; k += 2) {
      if (k < n_loc) {
        tower(k, GE);
        run_items(k, std::false_type{});
      }
      STRIP_BARRIER();
      if (sm.sync_fail[k & 1]) { failed = true; break; }
      if (k + 1 > n_loc) break;
      if (k + 1 < n_loc) {
        tower(k + 1, GO);
        run_items(k + 1, std::false_type{});
      }
      STRIP_BARRIER();
      if (sm.sync_fail[(k + 1) & 1]) { failed = true; break; }
    }
  }

  if (failed) {
    if (tid == 0) {
      if (has_down) st_agent(my_flag, lk.base + STRIP_FAIL);
      else { A.status[r] = ST_SYNC; A.Epf[r] = 0.0; }
    }
    return;
  }
  if (has_down) {
    if (tid == 0) st_agent(my_flag, lk.base + STRIP_DONE);
    return;
  }
  // last strip: the remaining exterior columns, then Z
  if (wave == 0) {
    for (int j = max(TURN + 2, n - 2); j <= n; j++) {
      double sacc = 0.0;
      for (int i = lane + 1; i <= j - TURN - 1; i += WAVE) sacc += sm.q5[i - 1] * ld_agent(&QEXT[j * ld + i]);
      sacc = wave_sum_f64(sacc);
      sm.q5[j] = sm.q5[j - 1] * sc1 + sacc;
    }
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

// grid: ceil(nseq / 8) groups of 8 S blocks; inside a group block x is strip x / 8 of sequence slot group * 8 + x % 8, so
// that the strips of a sequence are dispatched upstream first and share blockIdx mod 8
template <int NT>
__global__ __launch_bounds__(NT) void pf_strip_kernel(PfArgs A, StripLink lk) {
  __shared__ PfStripSmem<NT> sm;
  const int b = blockIdx.x, per = 8 * (lk.S + lk.pad);
  const int grp = b / per, x = b - grp * per;
  const int q = grp * 8 + (x & 7), s = x >> 3;
  if (q >= lk.nseq || s >= lk.S) return;          // (padding blocks: see STRIP_PAD)
  pf_strip_body<NT>(sm, A, lk, q, s);
}

}  // namespace drna
