// fold_mfe_strip.hpp -- Zuker MFE fill of ONE sequence by SEVERAL workgroups, each keeping its share of the rings in LDS
// like fold_mfe_lds.hpp: the path for 200 < n <= 2046.  Same recursions, same outputs and the same traceback as fold_mfe.hpp
// (reference utils/energy_scores.py:151,354).  The decomposition is the one of fold_pf_strip.hpp --
// strips of columns i, dependencies one way, one exchange record per diagonal and strip boundary, tower minima that walk from
// strip to strip -- with the min-plus algebra of fold_mfe_lds.hpp:
//   record of diagonal d (96 int32): [0,32) ring words (c + TermAU) * 256 + info of the strip's first 32 columns,
//   [32,64) their c + mismatchI words, [64,94) the tower minima (3 waves x 10 entries) of the tower that leaves the strip,
//   94 fML and 95 the decomposition minimum of the first column.
// fML cannot stay in LDS here (the multiloop splits read all of it): it goes to the sequence's table 2 in HBM/L2 like the
// general kernel's, stored write-through, and the split items read it from there (16 bytes = four adjacent cells per lane).
// A pseudoknot round is ONE launch of the fill (mfe_strip_kernel) followed by ONE launch of the traceback
// (mfe_strip_trace_kernel, eight waves per sequence, on the tables the strips left in HBM): the kernel boundary is the hand-over,
// the structure found so far (the output string itself) is the state that masks the next round.
#pragma once
#include "fold_mfe_lds.hpp"
#include <type_traits>

namespace drna {

constexpr int MSTRIP_REC = 96;        // int32 per exchange record
#ifdef MSTRIP_STAMPS
#define MST(k) do { const long long _n = clock64(); st_acc[k] += _n - st_last; st_last = _n; } while (0)
#else
#define MST(k) do { } while (0)
#endif
#ifndef MSTRIP_SKIP
#define MSTRIP_SKIP 0     // diagnostic builds only (results wrong): 1 no multiloop items, 2 no shape items, 4 no tower step, 8 no cell finalize, 128 split items without their second operand's loads, 256 those loads plain
#endif

constexpr int MSTRIP_KU = 8;       // split points per lane and round trip in the big batches (2 x 16-byte loads each)
// ---- blocked multiloop splits (StripLink::fark, chosen per launch by the engine: long folds).  With m = i + tt + 2 the split
// minimum of cell (i, j) is min_m fML(i, m-1) + fML(m, j), m = i + TURN + 2 .. j - TURN - 1: a (min,+) matrix product.  Same
// geometry and schedule as the blocked sums of the partition function (fold_pf_strip.hpp, which has the derivation): tiles of
// 16 x 16 cells in strip-local coordinates, the FAR range m_lo = 16 t + 31 + MKT_L .. 16 bj - 13 - MKT_L = m_hi of tile (t, bj)
// computed by a tile wave in chunks of 4 split points over the MKT_W steps before the tile's first cell is due, from the middle
// outward; near split points masked per cell in the per-diagonal items.  There is no matrix instruction for (min,+): a chunk is
// 2 loads (lane (h, x): fML(i_min + h + 4 (x >> 2), m0 + (x & 3) - 1) and fML(m0 + h, j_min + x)), 4 ds_bpermute that bring a
// lane the four second operands of its column, and 16 add + min pairs whose first operand is a DPP row broadcast
// (row_newbcast) -- 1024 terms for ~60 instructions and 2 loads, against 8 16-byte loads per 1024 terms per diagonal.  The
// accumulators (4 per lane) stay in registers over the window; the 256 minima of a tile go to an LDS slot (two per tile row:
// the tile is read for 31 diagonals, the next one of its row is due 16 later).  Round 2's form of this (-DMSTRIP_FARK: tile
// products through LDS staging, far items in the queue, lead 4 + 2 t) was slower than the plain items at every length and is gone.
#ifndef DRNA_MKT_W
#define DRNA_MKT_W 16
#endif
#ifndef DRNA_MKT_L
#define DRNA_MKT_L 4
#endif
constexpr int MKT_W = DRNA_MKT_W, MKT_L = DRNA_MKT_L;
constexpr int MKT_DEPTH = 2;          // chunks of a tile step in flight (8 / 4 / 3 / 2 / 1 at 400 nt x 256: 4.80 / 4.68 / 4.52 / 4.46 / 4.58 ms)
constexpr int MKT_BMIN = (44 + 2 * MKT_L + 15) / 16;          // smallest block distance with a far range
static_assert(MKT_W >= 1 && MKT_W <= 16 && MKT_L >= 3, "see tools/pkt_schedule.py");

// lane (row, n) of the 16-lane row: the value lane n of the same row holds (DPP row_newbcast on gfx90a and later)
template <int N>
__device__ __forceinline__ int row_bcast_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + N, 0xF, 0xF, false); }

template <int NT>
struct MfeStripSmem {
  static constexpr int NW = NT / WAVE;
  static constexpr int P = NT >= 1024 ? 128 : 64;        // physical tower lanes
  static constexpr int WMAX = P - 8;                     // widest strip
  static constexpr int RS = WMAX + 34;                   // ring row pitch: own columns 1..wid, halo wid+1..wid+32
  static constexpr int NL = WMAX + 8;
  static constexpr int NFIN = P / WAVE;                  // finalize waves
  static constexpr int NSVC = NT >= 1024 ? 2 : 0;        // service waves; else finalize wave 0 does their jobs
  static constexpr int NG = 3;                           // tower waves per block of 64 towers
  static constexpr int XT_STACK = 0, XT_INT11 = 64, XT_MM1N = 64 + 1024, XT_MM23 = 64 + 1024 + 128;
  int stack[64];
  int mmH[128], mmI[128], mm1n[128], mm23[128], mmM[128], mmExt[128];
  int int11[1024];
  int d5[32], d3[32];
  int f5[STRIP_NMAX + 2];
  int hpl[STRIP_NMAX + 2];
  int wring[33 * RS];            // (c + TermAU(inner type)) * 256 + info of the last 32 diagonals; row 32 stays INF
  int ciring[32 * RS];           // c + mismatchI(inner side)
  int dml[4 * RS];               // decomposition minima of the last 4 diagonals
  int fmlrow[2][RS];             // fML of the last two diagonals
  int accG[2][P], accI[2][P], accK[2][P];
  int dfar[(P + 15) / 16][2][256];   // far part of the split minimum of a tile: [tile row][tile column & 1][16 x 16 cells]
  int gimp[2][NG][GSLOTS + 2];   // minima of the tower that enters the strip, staged by the service wave
  int xtab[64 + 1024 + 128 + 128];
  int plist[2][NL];
  int xe[2][NL];
  int pcnt[2];
  int qk[2], qe[2];
  int eshape_rows[128];
  int tw_L[32];
  int tower_tab[2][32][6];
  unsigned char S[STRIP_NMAX + 4];
  unsigned char Sp[STRIP_NMAX + 4];
  int flag;
  int sync_fail[2];
};

template <typename RSRC>
__device__ __forceinline__ i32x4 buf_load_i32x4_sc1(RSRC rsrc, int voff, int soff) {
  const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 16);
  return i32x4{(int)v[0], (int)v[1], (int)v[2], (int)v[3]};
}

// exterior column j: f5[j] = min(f5[j-1], min_i f5[i-1] + EXT[j][i]); one wave.  The column's cells come from every strip: sc1
// loads, all of them issued before the first is used (a chain of dependent round trips here was the whole step's floor)
template <class SM, typename RSRC>
__device__ __forceinline__ void mstrip_f5_column(SM& sm, RSRC rsE, int ld, int j, int lane) {
  constexpr int NFX = 15;                             // chunks held in registers at once; longer columns continue in batches of 8
  const int cnt = j - TURN - 1;                       // cells i = 1 .. cnt
  const int nch = (cnt + WAVE - 1) >> 6;
  int fx[NFX];
#pragma unroll
  for (int c = 0; c < NFX; c++) {
    fx[c] = INF_DEV;
    if (c < nch) fx[c] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsE, (j * ld + lane + 1 + c * WAVE) * 4, 0, 16);
  }
  int m = INF_DEV;
#pragma unroll
  for (int c = 0; c < NFX; c++) {
    const int i = lane + 1 + c * WAVE;
    if (c < nch && i <= cnt && fx[c] < INF_DEV / 2) m = min(m, sm.f5[i - 1] + fx[c]);
  }
  for (int cb = NFX; cb < nch; cb += 8) {             // (columns beyond 960 cells)
    int gx[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      gx[u] = INF_DEV;
      if (cb + u < nch) gx[u] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsE, (j * ld + lane + 1 + (cb + u) * WAVE) * 4, 0, 16);
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int i = lane + 1 + (cb + u) * WAVE;
      if (cb + u < nch && i <= cnt && gx[u] < INF_DEV / 2) m = min(m, sm.f5[i - 1] + gx[u]);
    }
  }
  m = wave_min_i32(m);
  const int prev = sm.f5[j - 1];
  sm.f5[j] = prev < m ? prev : m;
}

// one diagonal step of a tower wave: import of the tower that enters the strip, the recurrence, export of the one that leaves
template <class SM>
__device__ __forceinline__ void mstrip_tower(SM& sm, int (&G)[GSLOTS], int d, int wid, int n_loc, int phys, int my_tb, int my_g,
                                             int lane, bool has_up, bool has_down, int32_t* rec_out) {
  constexpr int P = SM::P;
  const int ncell = min(wid, n_loc - d), sh = d >> 1, par = d & 1;
  const int iraw = ((phys - sh - 1) & (P - 1)) + 1;
  const bool live = iraw <= ncell;
  const int i = live ? iraw : 1;
  const int pe = (wid + sh) & (P - 1);
  if (has_up && ncell == wid && d - 2 > TURN && (pe >> 6) == my_tb) {
    const bool mine = lane == (pe & (WAVE - 1));
#pragma unroll
    for (int qx = 0; qx < GSLOTS; qx++) {
      const int v = sm.gimp[par][my_g][qx];
      G[qx] = mine ? v : G[qx];
    }
  }
  const int accG = (MSTRIP_SKIP & 4) ? INF_DEV : mfe_tower_step(sm, G, par, i * 4, my_g, SM::NG, lane);
  if (live) atomicMin(&sm.accG[par][phys], accG);
  if (has_down && live && iraw == 1) {
    int32_t* rec = rec_out + (long long)d * MSTRIP_REC + 64 + my_g * GSLOTS;
#pragma unroll
    for (int qx = 0; qx < GSLOTS; qx++) st_agent(rec + qx, (int32_t)G[qx]);
  }
}

// one strip of one sequence, one round.  q = sequence slot of the launch, s = strip (0 = highest columns)
template <int NT, bool FARK>
__device__ void mfe_strip_body(MfeStripSmem<NT>& sm, MfeArgs A, StripLink lk, StripRec xr, int q, int s, int round) {
  using SM = MfeStripSmem<NT>;
  constexpr int NW = SM::NW, RS = SM::RS, P = SM::P, NFIN = SM::NFIN, NSVC = SM::NSVC, NG = SM::NG;
  const MfeTables& T = *A.T;
  const int r = lk.idx ? lk.idx[q] : q + lk.r0;
  if (A.rg.len) A.L = A.rg.len[r];
  const long long so = A.rg.off ? (long long)A.rg.off[r] : (long long)r * A.L;
  const int n = A.L, ld = A.ld;
  const int tid = threadIdx.x, lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(wave_id());
  const int INF = INF_DEV, HALF = INF_DEV / 2;
  const int ninio = T.ninio, max_ninio = T.max_ninio, MLbase = T.MLbase, MLclosing = T.MLclosing,
            MLintern = T.MLintern, TermAU = T.TermAU;
  const int S = lk.S;
  int c0, c1;
  strip_bounds(n, S, s, c0, c1);
  if (c0 > n) return;
  const int wid = c1 - c0 + 1, n_loc = n - c0 + 1;
  const bool has_up = c1 < n, has_down = c0 > 1, last = !has_down;
  const int n_loc_up = n_loc - wid;
  int* const my_flag = lk.flags + ((long long)q * STRIP_MAXS + s) * 32;
  if (lk.fault && s == 0 && c1 == n && c0 > 1) {        // injected fault: the strips below see FAIL, the engine falls back
    if (threadIdx.x == 0) st_agent(my_flag, lk.base + STRIP_FAIL);
    return;
  }
  const int* const up_flag = lk.flags + ((long long)q * STRIP_MAXS + (s > 0 ? s - 1 : 0)) * 32;

  int32_t* base = A.ws + (long long)r * A.ws_stride;
  const long long tab = (long long)ld * ld;
  int32_t* Wc = base;                 // row 0: state word (see mfe_strip_trace_kernel), row 1: f5
  int32_t* PLX = base + 1 * tab;
  int32_t* FML = base + 2 * tab;
  int32_t* PL = base + 3 * tab;
  int32_t* EXT = base + 4 * tab;
  int32_t* const xbase = xr.rec + (long long)r * xr.stride;
  int32_t* const rec_out = xbase + (long long)s * ld * MSTRIP_REC;
  const int32_t* const rec_in = xbase + (long long)(s > 0 ? s - 1 : 0) * ld * MSTRIP_REC;
  int32_t* const PLC = xbase + (long long)(S - 1) * ld * MSTRIP_REC;      // list counts: [d * 8 + s]

  // later rounds run only for sequences whose previous round found a pair (state word 1, written by the traceback kernel)
  if (round > 0 && Wc[0] != 1) return;
  if (lk.clk && tid == 0) lk.clk[((long long)q * STRIP_MAXS + s) * 2] = wall_clock_100mhz();

  // ---- prologue: tables
  for (int k = tid; k < 64; k += NT) sm.stack[k] = T.stack[k];
  for (int k = tid; k < 128; k += NT) {
    sm.mmH[k] = T.mmH[k]; sm.mmI[k] = T.mmI[k]; sm.mm1n[k] = T.mm1n[k];
    sm.mm23[k] = T.mm23[k]; sm.mmM[k] = T.mmM[k]; sm.mmExt[k] = T.mmExt[k];
  }
  for (int k = tid; k < 1024; k += NT) sm.int11[k] = T.int11[k];
  for (int k = tid; k < 32; k += NT) { sm.d5[k] = T.d5[k]; sm.d3[k] = T.d3[k]; }
  if (tid == 0) { sm.flag = 0; sm.sync_fail[0] = 0; sm.sync_fail[1] = 0; }
  __syncthreads();
  // local sequence and pairing codes (4 = may not pair: positions paired in an earlier round, and both ends)
  const char* seq = A.seqs + so;
  for (int k = tid; k <= n_loc + 1; k += NT) {
    int g = c0 - 1 + k;
    const bool inside = g >= 1 && g <= n;
    g = g < 1 ? n : (g > n ? 1 : g);
    const int c = enc_nt(seq[g - 1]);
    sm.S[k] = (unsigned char)(c < 0 ? 0 : c);
    int p = c < 0 ? 4 : c;
    if (!inside) p = 4;
    else if (round > 0 && A.ss[so + g - 1] != '.') p = 4;
    sm.Sp[k] = (unsigned char)p;
  }
  for (int k = tid; k < 4 * RS; k += NT) sm.dml[k] = INF;
  for (int k = tid; k < 2 * RS; k += NT) (&sm.fmlrow[0][0])[k] = INF;
  for (int k = tid; k < 32 * RS; k += NT) { sm.wring[k] = INF * 256; sm.ciring[k] = INF; }
  for (int k = tid; k < RS; k += NT) sm.wring[32 * RS + k] = INF * 256;
  for (int k = tid; k <= n; k += NT) sm.hpl[k] = A.hp_len[k];
  for (int k = tid; k < P; k += NT)
    for (int p = 0; p < 2; p++) { sm.accG[p][k] = INF; sm.accI[p][k] = INF; sm.accK[p][k] = INF; }
  for (int k = tid; k < 2 * NG * (GSLOTS + 2); k += NT) (&sm.gimp[0][0][0])[k] = INF;
  for (int x = tid; x < 128; x += NT) {
    // shape slots of the 16-lane-row E items (see mfe_fill_lds)
    int s_, u1_, L_, kind_ = 0;
    if (x < 64) {
      const bool on = x < 58;
      u1_ = (x < 29 || !on) ? 0 : x - 27;
      s_ = !on ? 2 : x < 29 ? x + 2 : x - 27;
      L_ = on ? T.bulge[s_] : 0x3fff;
      if (x >= 58 && x <= 60) { kind_ = x - 53; s_ = x == 60 ? 4 : 3; u1_ = x == 58 ? 1 : 2; L_ = 0; }
    } else {
      const int y = x - 64;
      const bool on = y < 54;
      u1_ = (y < 27 || !on) ? 1 : y - 24;
      s_ = !on ? 4 : y < 27 ? y + 4 : y - 23;
      const int nl = s_ - 1;
      L_ = on ? T.interior[nl + 1] + min(max_ninio, (nl - 1) * ninio) : 0x3fff;
      if (y >= 54 && y <= 59) {
        const int z = y - 54;
        kind_ = z == 0 ? 1 : z <= 2 ? 2 : z == 3 ? 3 : 4;
        s_ = z == 0 ? 0 : z <= 2 ? 1 : z == 3 ? 2 : 5;
        u1_ = z <= 1 ? 0 : z <= 3 ? 1 : z - 2;
        L_ = 0;
      }
    }
    sm.eshape_rows[x] = s_ | (kind_ << 5) | (u1_ << 8) | (L_ << 16);
  }
  for (int k = tid; k < 32; k += NT) sm.tw_L[k] = k >= 6 && k <= 30 ? T.interior[k] : INF;
  __syncthreads();
  for (int k = tid; k < 64; k += NT) sm.xtab[SM::XT_STACK + k] = sm.stack[k] - ((k & 7) > 2 ? TermAU : 0);
  for (int k = tid; k < 1024; k += NT) sm.xtab[SM::XT_INT11 + k] = sm.int11[k] - (((k >> 4) & 7) > 2 ? TermAU : 0);
  for (int k = tid; k < 128; k += NT) {
    sm.xtab[SM::XT_MM1N + k] = sm.mm1n[k] - ((k >> 4) > 2 ? TermAU : 0);
    sm.xtab[SM::XT_MM23 + k] = sm.mm23[k] - ((k >> 4) > 2 ? TermAU : 0);
  }
  if (last) for (int j = tid; j <= n && j <= TURN + 1; j += NT) sm.f5[j] = 0;
  // compacted list of the strip's pairable cells of every diagonal, with the staged (1,2) / (2,1) / (2,2) loop energies
  for (int d = TURN + 1 + wave; d < n_loc; d += NW) {
    const int nc = min(wid, n_loc - d);
    int cntb = 0;
    for (int i0 = 1; i0 <= nc; i0 += WAVE) {
      const int i = i0 + lane;
      int t = 0;
      if (i <= nc) t = pair_type(sm.Sp[i], sm.Sp[i + d]);
      const unsigned long long m = __ballot(t != 0);
      if (t) {
        const int pos = cntb + __popcll(m & ((1ull << lane) - 1ull));
        const int j = i + d, si1 = sm.S[i + 1], sj1 = sm.S[j - 1];
        const bool va = d - 5 > TURN, vc = d - 6 > TURN;
        const int ta = va ? pair_type(sm.Sp[i + 2], sm.Sp[j - 3]) : 0, tb = va ? pair_type(sm.Sp[i + 3], sm.Sp[j - 2]) : 0,
                  tc = vc ? pair_type(sm.Sp[i + 3], sm.Sp[j - 3]) : 0;
        const int ra = rtype_of(ta), rb = rtype_of(tb), rc = rtype_of(tc);
        const int s_ip2 = sm.S[i + 2], s_jm2 = sm.S[j - 2];
        const int ea = T.int21[ta ? (t * 8 + ra) * 64 + si1 * 16 + s_jm2 * 4 + sj1 : 0];
        const int eb = T.int21[tb ? (rb * 8 + t) * 64 + sj1 * 16 + si1 * 4 + s_ip2 : 0];
        const int ec = T.int22[tc ? (t * 8 + rc) * 256 + si1 * 64 + s_ip2 * 16 + s_jm2 * 4 + sj1 : 0];
        const int a = ta ? ea - (ra > 2 ? TermAU : 0) : 0, b = tb ? eb - (rb > 2 ? TermAU : 0) : 0,
                  cc = tc ? ec - (rc > 2 ? TermAU : 0) : 0;
        PL[d * ld + c0 - 1 + pos] = (i | ((t * 16 + si1 * 4 + sj1) << 8)) | (cc << 15);
        PLX[d * ld + c0 - 1 + pos] = (a & 0xffff) | (b << 16);
      }
      cntb += __popcll(m);
    }
    if (lane == 0) PLC[d * STRIP_MAXS + s] = cntb;
  }
  __syncthreads();

  // roles
  const int w_svcA = NSVC ? NFIN : 0, w_svcB = NSVC ? NFIN + 1 : 0;
  const int aw = wave - NFIN - NSVC;
  const int my_tb = aw >= 0 ? aw / NG : NFIN, my_g = aw >= 0 ? aw - my_tb * NG : 0;
  const bool pinned = aw >= 0 && my_tb < NFIN;
  const bool fin = wave < NFIN;

  if (wave == w_svcA) {
    const int d = TURN + 1;
    if (d < n_loc) mfe_prepare_tower_tab(sm, d, lane, ninio, max_ninio);
  }
  if (wave == w_svcB) {
    const int d = TURN + 1;
    if (d < n_loc) {
      const int cnt = PLC[d * STRIP_MAXS + s];
      for (int k = lane; k < cnt; k += WAVE) { sm.plist[d & 1][k] = PL[d * ld + c0 - 1 + k]; sm.xe[d & 1][k] = PLX[d * ld + c0 - 1 + k]; }
      if (lane == 0) { sm.pcnt[d & 1] = cnt; sm.qk[0] = 0; sm.qk[1] = 0; sm.qe[0] = 0; sm.qe[1] = 0; }
    }
  }
  __syncthreads();

  const auto rsF = __builtin_amdgcn_make_buffer_rsrc((void*)FML, (short)0, (int)(tab * 4), 0x00020000);
  const auto rsE = __builtin_amdgcn_make_buffer_rsrc((void*)EXT, (short)0, (int)(tab * 4), 0x00020000);
  constexpr bool fark = FARK;                           // blocked multiloop splits (see MKT_L): a kernel of its own, so that the plain items keep their registers
  const int e_bulge1 = keep_i32(T.bulge[1]), e_int23 = keep_i32(T.interior[5] + ninio);

  // floating work items of diagonal d: multiloop splits from L2 (64 cells x 4 split-point groups per item, four adjacent cells
  // per lane; a block's split points are dealt to 1, 2, 4 or 8 items as the sums grow), then the bulge / 1xn / small shapes
  auto run_items = [&](const int d, auto with_k) {
    const int ncell = min(wid, n_loc - d), sh = d >> 1, par = d & 1;
    const int pcnt = __builtin_amdgcn_readfirstlane(sm.pcnt[par]);
    const int tmax = d - TURN - 2;
    // blocked form (fark): the near split points of the diagonal -- tt = m - 1 - i over [TURN+1, 28+MKT_L] and [d-29-MKT_L,
    // d-TURN-2], every cell masks what belongs to its tile's far range; one range while cells without a far range exist on the
    // diagonal (d < 16 MKT_BMIN) or the two meet
    const bool whole = !fark || d < 16 * MKT_BMIN || 28 + MKT_L + 1 >= d - 29 - MKT_L;
    const int n1 = whole ? tmax - TURN : 28 + MKT_L - TURN, s2 = d - 29 - MKT_L;
    const int nterm = whole ? n1 : n1 + tmax - s2 + 1;
    const int kssh = fark ? (nterm > 96 ? 3 : nterm > 48 ? 2 : nterm > 24 ? 1 : 0) : (d > 96 ? 3 : d > 48 ? 2 : d > 24 ? 1 : 0);
    const int KS = 1 << kssh, KG = 4 << kssh;
    const int astep = 4 * KG * ld, cstep = 4 * KG * (ld - 1);
    const int nK = (MSTRIP_SKIP & 1) ? 0 : ((ncell + 63) >> 6) << kssh, nE = (MSTRIP_SKIP & 2) ? 0 : (pcnt + 3) >> 2;
    const int nKF = nK;
    const int nItems = __builtin_amdgcn_readfirstlane(nKF + nE);
    auto pop = [&]() -> int {
      if (decltype(with_k)::value) {
        const int it = queue_pop(&sm.qk[par], lane);
        if (it < nKF) return it;
      }
      return nKF + queue_pop(&sm.qe[par], lane);
    };
    for (int it = pop(); it < nItems; it = pop()) {
      if (decltype(with_k)::value && it < nKF) {
        const int g = (it & (KS - 1)) * 4 + (lane >> 4), cl = lane & 15;
        int i = ((it >> kssh) << 6) + 4 * cl + 1;
        const bool act = i <= ncell;
        i = act ? i : 1;
        const int ig = i + c0 - 1;
        int m0 = INF, m1 = INF, m2 = INF, m3 = INF;
       if (fark) {
        // the lane's four cells (i + x, i + x + d): same tile row (i - 1 is a multiple of 4), maybe different tile columns; split
        // points tt <= lb or tt >= rb are near (all of them when the cell's tile has no far range)
        int lb[4], rb[4], fv[4];
        const int t16 = (i - 1) & ~15;
        const bool head = (it & (KS - 1)) == 0 && lane < 16;               // these lanes fold the tile's far minimum in
#pragma unroll
        for (int x = 0; x < 4; x++) {
          const int b16 = (i + x + d - 1) & ~15;
          const bool hf = b16 - t16 >= 16 * MKT_BMIN;
          lb[x] = hf ? t16 + 29 + MKT_L - (i + x) : 0x3fffffff;
          rb[x] = hf ? b16 - 13 - MKT_L - (i + x) : 0x3fffffff;
          fv[x] = (head && hf) ? sm.dfar[t16 >> 4][(b16 >> 4) & 1][((i + x - 1) & 15) * 16 + ((i + x + d - 1) & 15)] : INF;
        }
        // the near split points as ONE index range v = 0 .. nv-1 (v < n1: tt = TURN + 1 + v; else tt = s2 + v - n1), dealt to the
        // lanes' groups by v: four of them per trip, eight loads in flight
        const int r0 = s2, nv = nterm;
        auto tt_of = [&](const int v) { return v < n1 ? TURN + 1 + v : r0 + (v - n1); };
#define NEARV(x, tq) ((tq) <= lb[x] || (tq) >= rb[x])
        for (int v = g; v < nv; v += 4 * KG) {
          i32x4 a[4], c[4];
          int tq[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int vv = v + u * KG;
            tq[u] = vv < nv ? tt_of(vv) : -1;
            const int tl = tq[u] < 0 ? TURN + 1 : tq[u];
            a[u] = buf_load_i32x4(rsF, (tl * ld + ig) * 4, 0);
            c[u] = buf_load_i32x4_sc1(rsF, ((d - tl - 1) * ld + ig + tl + 1) * 4, 0);
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const bool on = tq[u] >= 0;
            m0 = min(m0, on && NEARV(0, tq[u]) ? a[u].x + c[u].x : INF); m1 = min(m1, on && NEARV(1, tq[u]) ? a[u].y + c[u].y : INF);
            m2 = min(m2, on && NEARV(2, tq[u]) ? a[u].z + c[u].z : INF); m3 = min(m3, on && NEARV(3, tq[u]) ? a[u].w + c[u].w : INF);
          }
        }
#undef NEARV
        m0 = min(m0, fv[0]); m1 = min(m1, fv[1]); m2 = min(m2, fv[2]); m3 = min(m3, fv[3]);
       } else {
        int tt = TURN + 1 + g;
        int vA = (tt * ld + ig) * 4;                               // fML[i .. i+3, . + tt]
        int vC = ((d - tt - 1) * ld + ig + tt + 1) * 4;            // fML[i+tt+1 .. , j ..]
        // split points in batches of KU (16 loads in flight), then 4, then the last <= 3 at once (masked): every batch is ONE
        // round trip to L2, and the round trips are what a split item costs
        auto batch = [&](auto nu, const int nvalid) {
          constexpr int U = decltype(nu)::value;
          i32x4 a[U], c[U];
#pragma unroll
          for (int u = 0; u < U; u++) {
            const bool on = u < nvalid;
            a[u] = buf_load_i32x4(rsF, on ? vA + u * astep : vA, 0);
            c[u] = (MSTRIP_SKIP & 128) ? a[u] : (MSTRIP_SKIP & 256) ? buf_load_i32x4(rsF, on ? vC - u * cstep : vC, 0) : buf_load_i32x4_sc1(rsF, on ? vC - u * cstep : vC, 0);
          }
#pragma unroll
          for (int u = 0; u < U; u++) {
            const bool on = u < nvalid;
            m0 = min(m0, on ? a[u].x + c[u].x : INF); m1 = min(m1, on ? a[u].y + c[u].y : INF);
            m2 = min(m2, on ? a[u].z + c[u].z : INF); m3 = min(m3, on ? a[u].w + c[u].w : INF);
          }
          vA += U * astep; vC -= U * cstep; tt += U * KG;
        };
        constexpr int KU = MSTRIP_KU;
        for (; tt + (KU - 1) * KG <= tmax;) batch(std::integral_constant<int, KU>{}, KU);
        for (; tt + 3 * KG <= tmax;) batch(std::integral_constant<int, 4>{}, 4);
        if (tt <= tmax) batch(std::integral_constant<int, 3>{}, (tmax - tt) / KG + 1);
       }
        // the four 16-lane rows hold different split points of the same cells
        m0 = min(m0, __shfl_xor(m0, 16)); m1 = min(m1, __shfl_xor(m1, 16)); m2 = min(m2, __shfl_xor(m2, 16)); m3 = min(m3, __shfl_xor(m3, 16));
        m0 = min(m0, __shfl_xor(m0, 32)); m1 = min(m1, __shfl_xor(m1, 32)); m2 = min(m2, __shfl_xor(m2, 32)); m3 = min(m3, __shfl_xor(m3, 32));
        if (lane < 16 && act) {
          if (m0 < HALF) atomicMin(&sm.accK[par][(i + sh) & (P - 1)], m0);
          if (i + 1 <= ncell && m1 < HALF) atomicMin(&sm.accK[par][(i + 1 + sh) & (P - 1)], m1);
          if (i + 2 <= ncell && m2 < HALF) atomicMin(&sm.accK[par][(i + 2 + sh) & (P - 1)], m2);
          if (i + 3 <= ncell && m3 < HALF) atomicMin(&sm.accK[par][(i + 3 + sh) & (P - 1)], m3);
        }
      } else {
        mfe_e_item_rows(sm, it - nKF, d, par, pcnt, sh, lane, TermAU, e_bulge1, e_int23, P - 1);
      }
    }
  };

  // ---- service jobs of step k.  What they read was stored write-through by other workgroups (or long ago by the prologue):
  // it comes from beyond the L2, ~1.5 us a trip -- with the loads inside the step the service waves WERE the floor of a step
  // (3.4 k of its 4.2 k cycles, tools/strip_stamps.py).  So every load is requested one step AHEAD and consumed from registers
  // at the top of the next step; the registers ride across the items and the barrier.
  // A: the record of diagonal k-1 of the strip above (ring halo, fML / decomposition minimum of its first column, the minima of
  //    the tower that enters at diagonal k+1).  It is requested as soon as a flag value READ EARLIER covers it (the flag itself
  //    is re-read every step, one step ahead as well); only when the strip above is less than two diagonals ahead does the
  //    wave wait (bounded) and load inside the step.  Tower table of diagonal k+1.
  int sa_w0 = 0, sa_g0 = 0, sa_f = lk.base;
  bool sa_pend = false;
  auto sa_request = [&](const int dd) {                               // record of diagonal dd -> registers
    const int32_t* rec = rec_in + (long long)dd * MSTRIP_REC;
    sa_w0 = ld_agent(rec + lane);                                     // lanes 0..31 ring words, 32..63 c + mismatchI words
    sa_g0 = ld_agent(rec + 64 + (lane & 31));                         // tower minima (30), fML, decomposition minimum
  };
  auto service_a = [&](const int k) {
    const bool need = has_up && k - 1 > TURN && k - 1 <= n_loc_up - 1;
    if (need) {
      if (!sa_pend) {
        int seen = sa_f;
        if (!flag_ge(sa_f, lk.base + k - 1)) (void)strip_wait(up_flag, lk.base, k - 1, seen);
        sa_f = seen;
        if (flag_ge(sa_f, lk.base + k - 1) && sa_f != lk.base + STRIP_FAIL) { sa_request(k - 1); sa_pend = true; }
      }
      if (!sa_pend) {
        sm.sync_fail[k & 1] = 1;
        if (lane == 0 && lk.dbg) { int* g = lk.dbg + q * 8; g[0] = s + 1; g[1] = k; g[2] = sa_f; g[3] = lk.base; g[4] = (int)blockIdx.x; g[5] = n; }
      } else {
        const int dd = k - 1;
        if (lane < 32) sm.wring[(dd & 31) * RS + wid + 1 + lane] = sa_w0;
        else sm.ciring[(dd & 31) * RS + wid + 1 + lane - 32] = sa_w0;
        if (lane < NG * GSLOTS) sm.gimp[(k + 1) & 1][lane / GSLOTS][lane % GSLOTS] = sa_g0;
        if (lane == 30) sm.fmlrow[dd & 1][wid + 1] = sa_g0;
        if (lane == 31) sm.dml[(dd & 3) * RS + wid + 1] = sa_g0;
      }
    }
    sa_pend = false;
    // one step ahead: the record of diagonal k, if the flag as last seen covers it; and the flag again
    const bool next = has_up && k > TURN && k <= n_loc_up - 1;
    if (next) {
      if (flag_ge(sa_f, lk.base + k) && sa_f != lk.base + STRIP_FAIL) { sa_request(k); sa_pend = true; }
      sa_f = __builtin_amdgcn_readfirstlane(ld_agent(up_flag));
    }
    if (k + 1 < n_loc) mfe_prepare_tower_tab(sm, k + 1, lane, ninio, max_ninio);
  };
  // B: pairable list of diagonal k+1 (entries beyond the count are never read, so the rows do not wait for it); exterior
  //    column j = k-3 (last strip).  Both requested one step ahead: lp_* / fx_*.
  constexpr int NFX = 15;       // chunks of the exterior column requested ahead (960 cells); longer columns continue inside the step
  int lp_cnt = 0, lp_p0 = 0, lp_p1 = 0, lp_x0 = 0, lp_x1 = 0, fx[NFX];
  auto sb_request = [&](const int k) {                                // for step k: list of diagonal k+1, column k-3
    if (!(MSTRIP_SKIP & 64) && k + 1 < n_loc) {
      const int32_t* row = PL + (k + 1) * ld + c0 - 1;
      const int32_t* rowx = PLX + (k + 1) * ld + c0 - 1;
      lp_cnt = PLC[(k + 1) * STRIP_MAXS + s];
      lp_p0 = row[lane]; lp_p1 = row[min(lane + WAVE, wid)];
      lp_x0 = rowx[lane]; lp_x1 = rowx[min(lane + WAVE, wid)];
    }
    const int j = k - 3, fcnt = j - TURN - 1;
    const int nch = (!(MSTRIP_SKIP & 32) && last && j >= TURN + 2) ? (fcnt + WAVE - 1) >> 6 : 0;
#pragma unroll
    for (int c = 0; c < NFX; c++) {
      fx[c] = INF;
      if (c < nch) fx[c] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsE, (j * ld + lane + 1 + c * WAVE) * 4, 0, 16);
    }
  };
  if (wave == w_svcB) sb_request(TURN + 1);
  auto service_b = [&](const int k) {
    const int dn = k + 1, j = k - 3, fcnt = j - TURN - 1;
    if (!(MSTRIP_SKIP & 64) && k + 1 < n_loc) {
      sm.plist[dn & 1][lane] = lp_p0; sm.xe[dn & 1][lane] = lp_x0;
      if (lane + WAVE < SM::NL) { sm.plist[dn & 1][lane + WAVE] = lp_p1; sm.xe[dn & 1][lane + WAVE] = lp_x1; }
      if (lane == 0) { sm.pcnt[dn & 1] = lp_cnt; sm.qk[dn & 1] = 0; sm.qe[dn & 1] = 0; }
    }
    if (!(MSTRIP_SKIP & 32) && last && j >= TURN + 2) {
      int m = INF;
#pragma unroll
      for (int c = 0; c < NFX; c++) {
        const int i = lane + 1 + c * WAVE;
        if (c * WAVE < fcnt && i <= fcnt && fx[c] < HALF) m = min(m, sm.f5[i - 1] + fx[c]);
      }
      for (int cb = NFX; cb * WAVE < fcnt; cb += 8) {             // (columns beyond 960 cells)
        int gx[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          gx[u] = INF;
          if ((cb + u) * WAVE < fcnt) gx[u] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsE, (j * ld + lane + 1 + (cb + u) * WAVE) * 4, 0, 16);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int i = lane + 1 + (cb + u) * WAVE;
          if ((cb + u) * WAVE < fcnt && i <= fcnt && gx[u] < HALF) m = min(m, sm.f5[i - 1] + gx[u]);
        }
      }
      m = wave_min_i32(m);
      const int prev = sm.f5[j - 1];
      sm.f5[j] = prev < m ? prev : m;
    }
    sb_request(k + 1);              // column k-2: its cells (diagonals <= k-3) were stored in step k-2 and drained by its barrier
  };

  // ---- tile products of the blocked form (see MKT_L above and fold_pf_strip.hpp).  Tile waves: the tower waves and the service
  // waves, one tile row each -- a tower wave requests its chunks' operands, runs its tower step (LDS work) under their round trip
  // and folds them in afterwards (tile_issue / tile_finish); on the floating waves instead: 4.84 vs 4.46 ms at 400 nt x 256.  At step k the tiles of block distance B = (k + 15 + MKT_W) >> 4 are in step
  // g = (k + 15 + MKT_W) & 15 of their window (g < MKT_W).
  constexpr int NTW = NFIN * NG + NSVC;
  constexpr int TOWN = ((SM::WMAX + 15) / 16 + NTW - 1) / NTW;
  const int tf = aw >= 0 ? aw : NFIN * NG + wave - NFIN;
  int tacc[TOWN][4];
#pragma unroll
  for (int o = 0; o < TOWN; o++) { tacc[o][0] = INF; tacc[o][1] = INF; tacc[o][2] = INF; tacc[o][3] = INF; }
  constexpr int DEPTH = MKT_DEPTH;                                           // chunks in flight (two loads each)
  int tq_a[TOWN][DEPTH], tq_b[TOWN][DEPTH];                                  // operands requested by tile_issue
  struct TileStep { int t, bj, m_lo, m_hi, nch, lo0, nlo, hi0, ncs, oA, oB; bool on; };
  auto tile_step = [&](const int k, const int o, int& g) -> TileStep {
    TileStep q;
    const int xk = k + 15 + MKT_W, B = xk >> 4;
    g = xk & 15;
    q.t = tf + NTW * o; q.bj = q.t + B;
    q.on = fark && !(MSTRIP_SKIP & 16) && g < MKT_W && B >= MKT_BMIN && 16 * q.t < wid && 16 * q.bj + 1 <= n_loc;   // (wave-uniform)
    q.m_lo = 16 * q.t + 31 + MKT_L; q.m_hi = 16 * q.bj - 13 - MKT_L;
    q.nch = (q.m_hi - q.m_lo + 4) >> 2;
    const int nl = (q.nch + 1) >> 1, nh = q.nch >> 1;
    const int cl = (nl + MKT_W - 1) / MKT_W, ch = (nh + MKT_W - 1) / MKT_W, e = MKT_W - 1 - g;
    q.lo0 = e * cl; q.nlo = max(0, min(nl, q.lo0 + cl) - q.lo0); q.hi0 = e * ch;
    const int nhi = max(0, min(nh, q.hi0 + ch) - q.hi0);
    q.ncs = q.nlo + nhi;                                                      // chunks of this step: low side first
    // lane (h, x) loads fML(i_min + h + 4 (x >> 2), m - 1) for m = m0 + (x & 3) and fML(m0 + h, j_min + x); rows / columns
    // beyond the strip or the sequence repeat the last one (their minima are never read)
    const int h = lane >> 4, x = lane & 15;
    const int il = min(16 * q.t + 1 + h + 4 * (x >> 2), wid), jl = min(16 * q.bj + 1 + x, n_loc);
    q.oA = ((-1 - il) * ld + c0 - 1 + il) * 4; q.oB = (jl * ld + c0 - 1) * 4;       // + m * (ld * 4)  |  - m * (ld - 1) * 4
    return q;
  };
  auto tile_load = [&](const TileStep& q, const int c, int& a, int& b) {
    const int cid = c < q.nlo ? q.lo0 + c : q.nch - 1 - (q.hi0 + c - q.nlo);
    const int m0 = q.m_lo + 4 * cid, ma = m0 + (lane & 3), mb = m0 + (lane >> 4);
    a = (int)__builtin_amdgcn_raw_buffer_load_b32(rsF, q.oA + min(ma, q.m_hi) * (ld * 4), 0, 0);
    b = (int)__builtin_amdgcn_raw_buffer_load_b32(rsF, q.oB - min(mb, q.m_hi) * ((ld - 1) * 4), 0, 16);
    if (ma > q.m_hi) a = INF;
    if (mb > q.m_hi) b = INF;
  };
  // one chunk into the four accumulators of lane (h, x): rows h + 4 rr, column x
  auto tile_fold = [&](const int a, const int b, int& acc0, int& acc1, int& acc2, int& acc3) {
    const int x = lane & 15;
    const int b0 = lane_fetch_i32(b, x), b1 = lane_fetch_i32(b, 16 + x), b2 = lane_fetch_i32(b, 32 + x), b3 = lane_fetch_i32(b, 48 + x);
    acc0 = min(acc0, min(min(row_bcast_i32<0>(a) + b0, row_bcast_i32<1>(a) + b1), min(row_bcast_i32<2>(a) + b2, row_bcast_i32<3>(a) + b3)));
    acc1 = min(acc1, min(min(row_bcast_i32<4>(a) + b0, row_bcast_i32<5>(a) + b1), min(row_bcast_i32<6>(a) + b2, row_bcast_i32<7>(a) + b3)));
    acc2 = min(acc2, min(min(row_bcast_i32<8>(a) + b0, row_bcast_i32<9>(a) + b1), min(row_bcast_i32<10>(a) + b2, row_bcast_i32<11>(a) + b3)));
    acc3 = min(acc3, min(min(row_bcast_i32<12>(a) + b0, row_bcast_i32<13>(a) + b1), min(row_bcast_i32<14>(a) + b2, row_bcast_i32<15>(a) + b3)));
  };
  auto tile_issue = [&](const int k) {
#pragma unroll
    for (int o = 0; o < TOWN; o++) {
      int g;
      const TileStep q = tile_step(k, o, g);
#pragma unroll
      for (int u = 0; u < DEPTH; u++) {
        tq_a[o][u] = INF; tq_b[o][u] = INF;
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
      int acc0 = tacc[o][0], acc1 = tacc[o][1], acc2 = tacc[o][2], acc3 = tacc[o][3];
      if (g == 0) { acc0 = INF; acc1 = INF; acc2 = INF; acc3 = INF; }
#pragma unroll
      for (int u = 0; u < DEPTH; u++)
        if (u < q.ncs) tile_fold(tq_a[o][u], tq_b[o][u], acc0, acc1, acc2, acc3);
      for (int c = DEPTH; c < q.ncs; c += DEPTH) {                            // (long folds: more chunks per step than ride in registers)
        int av[DEPTH], bv[DEPTH];
#pragma unroll
        for (int u = 0; u < DEPTH; u++) {
          av[u] = INF; bv[u] = INF;
          if (c + u < q.ncs) tile_load(q, c + u, av[u], bv[u]);
        }
#pragma unroll
        for (int u = 0; u < DEPTH; u++)
          if (c + u < q.ncs) tile_fold(av[u], bv[u], acc0, acc1, acc2, acc3);
      }
      tacc[o][0] = acc0; tacc[o][1] = acc1; tacc[o][2] = acc2; tacc[o][3] = acc3;
      if (g == MKT_W - 1) {                    // accumulator rr of lane (h, x): cell (row h + 4 rr, column x) of the tile
        int* slot = sm.dfar[q.t][q.bj & 1] + (lane >> 4) * 16 + (lane & 15);
        slot[0] = acc0; slot[64] = acc1; slot[128] = acc2; slot[192] = acc3;
      }
    }
  };
  auto tile_job = [&](const int k) { tile_issue(k); tile_finish(k); };

  bool failed = false;
#ifdef MSTRIP_STAMPS
  long long st_acc[4] = {0, 0, 0, 0}, st_last = clock64();
#endif
  if (fin) {
    // ================= finalize waves: diagonal d = k-1 at step k
    for (int k = TURN + 1; k <= n_loc; k++) {
      const int d = k - 1;
      if (tid == 0 && has_down && k - 2 > TURN) st_agent(my_flag, lk.base + k - 2);
      if (d > TURN) {
        const int ncell = min(wid, n_loc - d), sh = d >> 1, par = d & 1;
        const int i = ((tid - sh - 1) & (P - 1)) + 1;
        const int dm1v = as_vector(d - 1);
        if (!(MSTRIP_SKIP & 8) && i <= ncell) {
          const int aG = sm.accG[par][tid], aI = sm.accI[par][tid], aK = sm.accK[par][tid];
          sm.accG[par][tid] = INF; sm.accI[par][tid] = INF; sm.accK[par][tid] = INF;
          const int j = i + d;
          const int ig = i + c0 - 1, jg = ig + d;
          const int t = pair_type(sm.Sp[i], sm.Sp[j]);
          const int tau = t > 2 ? TermAU : 0;
          int c = INF, info = 0, cb = INF;
          if (t) {
            const int ij = t * 16 + sm.S[i + 1] * 4 + sm.S[j - 1];
            c = mfe_hairpin_e(sm, T, sm.hpl[dm1v], i, j, t);
            c = min(c, aI);
            c = min(c, aG + sm.mmI[ij]);
            const int dmlv = sm.dml[((d - 2) & 3) * RS + i + 1];
            if (dmlv < HALF)
              c = min(c, dmlv + MLclosing + MLintern + tau + sm.mmM[rtype_of(t) * 16 + sm.S[j - 1] * 4 + sm.S[i + 1]]);
            if (c >= HALF) c = INF;
            info = (rtype_of(t) << 4) | (sm.S[j + 1] << 2) | sm.S[i - 1];
            cb = c < INF ? c + tau : INF;
          }
          const int ww = cb * 256 + info, cw = c < INF ? c + sm.mmI[info] : INF;
          sm.wring[(d & 31) * RS + i] = ww;
          sm.ciring[(d & 31) * RS + i] = cw;
          Wc[d * ld + ig] = c * 256 + info;
          int est = 0;                                    // E_ExtLoop(type, i > 1 ? S[i-1] : -1, j < n ? S[j+1] : -1), dangles = 2
          if (t) {
            if (ig > 1 && jg < n) est = sm.mmExt[t * 16 + sm.S[i - 1] * 4 + sm.S[j + 1]];
            else if (ig > 1) est = sm.d5[t * 4 + sm.S[i - 1]];
            else if (jg < n) est = sm.d3[t * 4 + sm.S[j + 1]];
          }
          strip_store(&EXT[jg * ld + ig], (int32_t)(c < INF ? c + tau + est : INF));
          int f = INF;
          if (d - 1 > TURN) {
            const int fa = sm.fmlrow[(d - 1) & 1][i + 1], fb = sm.fmlrow[(d - 1) & 1][i];
            if (fa < HALF) f = fa + MLbase;
            if (fb < HALF) f = min(f, fb + MLbase);
          }
          if (c < INF) f = min(f, c + MLintern + tau + sm.mmM[t * 16 + sm.S[i - 1] * 4 + sm.S[j + 1]]);
          int aKf = aK;
          const int dec = aKf >= HALF ? INF : aKf;
          sm.dml[(d & 3) * RS + i] = dec;
          const int fv = min(f, dec);
          sm.fmlrow[d & 1][i] = fv;
          strip_store(&FML[d * ld + ig], (int32_t)fv);
          if (has_down && i <= 32) {
            int32_t* rec = rec_out + (long long)d * MSTRIP_REC;
            st_agent(rec + (i - 1), (int32_t)ww);
            st_agent(rec + 32 + (i - 1), (int32_t)cw);
            if (i == 1) { st_agent(rec + 94, (int32_t)fv); st_agent(rec + 95, (int32_t)dec); }
          }
        }
      }
      if (!NSVC && wave == 0) { service_a(k); service_b(k); }
      MST(0);
      if (k < n_loc) run_items(k, std::true_type{});
      MST(1);
      STRIP_BARRIER();
      MST(2);
      if (sm.sync_fail[k & 1]) { failed = true; break; }
    }
  } else if (NSVC && wave < NFIN + NSVC) {
    // ================= service waves
    for (int k = TURN + 1; k <= n_loc; k++) {
      if (wave == w_svcA) service_a(k); else service_b(k);
      MST(0);
      if (k < n_loc) { tile_job(k); run_items(k, std::true_type{}); }
      MST(1);
      STRIP_BARRIER();
      MST(2);
      if (sm.sync_fail[k & 1]) { failed = true; break; }
    }
  } else if (!pinned) {
    // ================= floating waves: tile products and items
    for (int k = TURN + 1; k <= n_loc; k++) {
      if (k < n_loc) run_items(k, std::true_type{});
      MST(1);
      STRIP_BARRIER();
      MST(2);
      if (sm.sync_fail[k & 1]) { failed = true; break; }
    }
  } else {
    // ================= tower waves: diagonal d = k at step k, then shape items
    int GE[GSLOTS], GO[GSLOTS];
#pragma unroll
    for (int qx = 0; qx < GSLOTS; qx++) { GE[qx] = INF; GO[qx] = INF; }
    const int phys = my_tb * WAVE + lane;
    static_assert(((TURN + 1) & 1) == 0, "the loop below starts on an even diagonal");
    for (int k = TURN + 1; k <= n_loc; k += 2) {
      if (k < n_loc) {
        tile_issue(k);
        mstrip_tower(sm, GE, k, wid, n_loc, phys, my_tb, my_g, lane, has_up, has_down, rec_out);
        tile_finish(k);
        MST(0);
        run_items(k, std::false_type{});
        MST(1);
      }
      STRIP_BARRIER();
      MST(2);
      if (sm.sync_fail[k & 1]) { failed = true; break; }
      if (k + 1 > n_loc) break;
      if (k + 1 < n_loc) {
        tile_issue(k + 1);
        mstrip_tower(sm, GO, k + 1, wid, n_loc, phys, my_tb, my_g, lane, has_up, has_down, rec_out);
        tile_finish(k + 1);
        MST(0);
        run_items(k + 1, std::false_type{});
        MST(1);
      }
      STRIP_BARRIER();
      MST(2);
      if (sm.sync_fail[(k + 1) & 1]) { failed = true; break; }
    }
  }

#ifdef MSTRIP_STAMPS
  if (lk.clk && q == 0 && last && lane == 0)
    for (int k = 0; k < 4; k++) lk.clk[16 + wave * 4 + k] = st_acc[k];
#endif
  if (failed) {
    if (tid == 0) {
      if (has_down) st_agent(my_flag, lk.base + STRIP_FAIL);
      else { A.status[r] = ST_SYNC; Wc[0] = -1; }
    }
    return;
  }
  if (lk.clk && tid == 0) lk.clk[((long long)q * STRIP_MAXS + s) * 2 + 1] = wall_clock_100mhz();
  if (has_down) {
    if (tid == 0) st_agent(my_flag, lk.base + STRIP_DONE);
    return;
  }
  // last strip: the remaining exterior columns; f5 and the state word go to the traceback kernel
  if (wave == 0) {
    for (int j = max(TURN + 2, n - 2); j <= n; j++) mstrip_f5_column(sm, rsE, ld, j, lane);
  }
  __syncthreads();
  for (int k = tid; k <= n; k += NT) Wc[ld + k] = sm.f5[k];
  if (tid == 0) Wc[0] = 2;
}

template <int NT, bool FARK>
__global__ __launch_bounds__(NT) void mfe_strip_kernel(MfeArgs A, StripLink lk, StripRec xr, int round) {
  __shared__ MfeStripSmem<NT> sm;
  const int b = blockIdx.x, per = 8 * (lk.S + lk.pad);
  const int grp = b / per, x = b - grp * per;
  const int q = grp * 8 + (x & 7), s = x >> 3;
  if (q >= lk.nseq || s >= lk.S) return;          // (padding blocks: see STRIP_PAD)
  mfe_strip_body<NT, FARK>(sm, A, lk, xr, q, s, round);
}

// traceback of one round, one workgroup of TRACE_WAVES waves per sequence, on the tables the strips of that round left in HBM (a
// launch of its own: the kernel boundary makes them visible).  The waves work from one queue of sectors in LDS (TbShared,
// fold_mfe.hpp: exterior stems and multiloop branches are independent), as the LDS-resident kernels' traceback does since
// round 3; round 3's one wave per sequence took 0.18-0.22 ms per round at 400 nt.  State word Wc[0]: 2 = the fill of this round
// is done (written by the last strip), -1 = a strip lost its neighbour; this kernel leaves 1 if another round follows for the
// sequence, else 0.
constexpr int TRACE_WAVES = 8;
struct MfeTraceSmem : MfeSmemCore<STRIP_NMAX> {
  int tbq[4];                    // sector queue of the traceback (TbShared)
};

__device__ inline void mfe_strip_trace_body(MfeTraceSmem& sm, MfeArgs A, const int* idx, int q, int round, int r0 = 0) {
  constexpr int NT = TRACE_WAVES * WAVE;
  const int r = idx ? idx[q] : q + r0;
  if (A.rg.len) A.L = A.rg.len[r];
  const long long so = A.rg.off ? (long long)A.rg.off[r] : (long long)r * A.L;
  const int n = A.L, ld = A.ld, tid = threadIdx.x;
  const MfeTables& T = *A.T;
  int32_t* base = A.ws + (long long)r * A.ws_stride;
  const long long tab = (long long)ld * ld;
  int32_t* Wc = base;
  const int32_t* FML = base + 2 * tab;
  const int32_t* EXT = base + 4 * tab;
  const int state = Wc[0];
  if (round > 0 && state != 2) return;                 // no fill this round: the sequence was finished earlier (workgroup-uniform)
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
    const char prev = round > 0 ? A.ss[so + k] : '.';
    sm.sspk[k] = prev;
    sm.Sp[k + 1] = (unsigned char)(c < 0 || prev != '.' ? 4 : c);
    sm.ssw[k] = '.';
  }
  for (int k = tid; k <= n; k += NT) sm.f5[k] = Wc[ld + k];
  for (int k = tid; k < (int)(sizeof(sm.sec_ml) / sizeof(sm.sec_ml[0])); k += NT) sm.sec_ml[k] = 0;
  __syncthreads();
  if (tid == 0) {
    sm.S[0] = sm.S[n]; sm.S[n + 1] = sm.S[1]; sm.Sp[0] = 4; sm.Sp[n + 1] = 4;
    sm.sec_i[0] = 1; sm.sec_j[0] = (short)n; sm.sec_ml[0] = 1;          // entry 0 = the whole exterior interval (ml 0), published
    sm.tbq[0] = 0; sm.tbq[1] = 1; sm.tbq[2] = 1; sm.tbq[3] = 0;
  }
  __syncthreads();
  if (sm.flag) {
    if (tid == 0) { A.status[r] = ST_BAD_CHAR; A.Emfe[r] = 0; Wc[0] = 0; }
    for (int k = tid; k < n; k += NT) A.ss[so + k] = '.';
    return;
  }
  if (state != 2) {                                    // round 0 and the fill failed (ST_SYNC is already in the status word)
    if (tid == 0) { A.Emfe[r] = 0; Wc[0] = 0; }
    for (int k = tid; k < n; k += NT) A.ss[so + k] = '.';
    return;
  }
  (void)mfe_traceback_q(sm, A, Wc, FmlGlobal{FML, ld}, EXT, TbShared<MfeTraceSmem>{sm});
  __syncthreads();
  if (sm.tbq[3] == 2) {                                // a wave could not reproduce a table value
    if (tid == 0) { A.status[r] = ST_TRACEBACK; Wc[0] = 0; if (round == 0) A.Emfe[r] = sm.f5[n]; }
    for (int k = tid; k < n; k += NT) A.ss[so + k] = sm.sspk[k];
    return;
  }
  const char op = round == 0 ? '(' : round == 1 ? '[' : round == 2 ? '<' : '{';
  const char cl = round == 0 ? ')' : round == 1 ? ']' : round == 2 ? '>' : '}';
  int any = 0;
  for (int k = tid; k < n; k += NT) {
    const char ch = sm.ssw[k];
    if (ch == '(') { sm.sspk[k] = op; any = 1; }
    else if (ch == ')') sm.sspk[k] = cl;
    A.ss[so + k] = sm.sspk[k];
  }
  if (any) sm.flag = 2;
  __syncthreads();
  if (tid == 0) {
    if (round == 0) A.Emfe[r] = sm.f5[n];
    A.status[r] = ST_OK;
    // reference sequence_utils.py:1194,1210: the next re-fold happens only if this one found a pair
    Wc[0] = ((round == 0 || sm.flag == 2) && round < A.pk_rounds) ? 1 : 0;
  }
}

__global__ __launch_bounds__(TRACE_WAVES * WAVE) void mfe_strip_trace_kernel(MfeArgs A, const int* idx, int nseq, int round, int r0) {
  __shared__ MfeTraceSmem sm;
  if ((int)blockIdx.x >= nseq) return;
  mfe_strip_trace_body(sm, A, idx, blockIdx.x, round, r0);
}

}  // namespace drna
