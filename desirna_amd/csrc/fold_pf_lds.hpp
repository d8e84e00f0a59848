// fold_pf_lds.hpp -- LDS-resident McCaskill partition function (inside) for n <= PF_FAST_NMAX: the
// production path of the headline workload.  Same recursions and outputs as fold_pf.hpp (reference
// utils/energy_scores.py:150 with compute_bpp = 0); design shared with
// fold_mfe_lds.hpp:
//   * qb is kept as a 32-diagonal LDS ring (stored as qb * expMismatchI(inner side), plus one info byte
//     per cell), qm / qm1 go to HBM/L2 (both full triangles are needed by the multiloop sums and do not
//     fit in LDS in fp64), D = sum_k qm qm1 of the last four diagonals and the previous diagonal of qm1
//     and U stay in LDS;
//   * generic interior loops by register-resident tower sums (2 LDS reads per live inner diagonal);
//   * bulges and 1xn loops one PAIRABLE cell at a time with the 112 shapes spread over the lanes and a
//     DPP wave sum; the nine fixed small shapes by one wave per 64 pairable cells;
//   * multiloop sums: the far split points of a 4 x 4 tile of cells as one matrix product (v_mfma_f64_4x4x4_4b_f64),
//     the near ones per cell from L2, a wave takes 32 cells x 4 rows of slots;
//   * finalize waves (one lane per cell) run one diagonal behind the sweep waves: one barrier per
//     diagonal.  Every partial sum has exactly one writer and is combined in a fixed order, so the
//     result is bit-reproducible from run to run and independent of the batch composition.
#pragma once
#include <cstddef>
#include "eval_structure.hpp"
#include "fold_pf.hpp"

namespace drna {

constexpr int PF_FAST_NMAX = 200;
constexpr int PGSLOTS = 10;       // strip kernels (fold_pf_strip.hpp): tower entries per pinned wave, 28 residues over 3 waves
constexpr int PNG = 3;            // strip kernels: sweep waves pinned to one 64-tower block
constexpr int TSL = 14;           // tower entries per tower wave: loop sizes s = 4 + sigma + 2 q (q = 0..13, s <= 30)

template <int NT>
struct PfFastSmem {
  static constexpr int NW = NT / WAVE;
  static constexpr int RS = PF_FAST_NMAX + 2;
  static constexpr int NSLOT = 4 * WAVE;
  static constexpr int NL = PF_FAST_NMAX + 8;
  double qbi[32 * RS];            // qb * expMismatchI(inner side) of the last 32 diagonals.  Zero-initialised: rows of diagonals
                                  // that do not exist yet (d' <= TURN or negative, wrapped) read as zero until their first tenant
                                  // is written, which is after their last such read.  (A pitch of 2 KB would let a 16-bit add wrap
                                  // the row for free, but then equal columns of all rows share a bank and the bulge items, whose
                                  // lanes read one column in sixteen rows, run into 16-way conflicts: measured +0.08 ms.)
  double dring[4 * RS];           // D[i,j] = sum_k qm[i,k-1] qm1[k,j] of the last 4 diagonals
  double qm1row[2][RS];           // qm1 of the previous diagonal
  double urow[2][RS];             // U of the previous diagonal
  double hpw[PF_FAST_NMAX + 2];   // hairpin weight by loop size (scale folded in)
  double q5[PF_FAST_NMAX + 2];
  double partG[2][2][NSLOT];      // tower sums, one slice per parity of the loop size
  double partK[2][4][NSLOT];      // multiloop sums, near split points: one slice per row
  double accE[2][NSLOT], accX[2][3][NSLOT];   // accX: one slice per group of fixed shapes
  // Boltzmann tables
  double stack[64], mmH[128], mmI[128], mm1n[128], mm23[128], mmM[128], mmExt[128], int11[1024], d5[32], d3[32];
  double rinv[128];               // 1 / expMismatchI(info)
  double rbul[128];               // expTermAU(inner type) / expMismatchI(info)
  double r1n[128];                // expMismatch1nI(info) / expMismatchI(info)
  double r23[128];                // expMismatch23I(info) / expMismatchI(info)
  double eWt[36][2];              // E items, by loop size t: {bulge[t] scale[t+2], interior[t] eninio[t-2] scale[t+2] (1xn loops, t >= 4)}; 0 beyond 30
  double xc[16];                  // weights of the fixed small shapes: bulge-1, 2x3, scale^4, scale^5, scale^6 (read by the X items);
                                  // 8..: TermAU, MLclosing, MLintern, expMLbase, scale^2: the cell finalize reads its constants here
                                  // (held in registers over the loop they cost spills on the finalize waves' critical path)
  double twc[32][2];              // by total size s of a generic loop: {eninio[s - 4], interior[s] scale[s + 2] (0 below s = 6)}
  int plist[2][NL];               // pairable cells of the diagonal: i | ij << 8
  int pcnt[2];
  int qhead[2];                   // work-queue head of the diagonal's floating items
  int qtile[2];                   // ... of its tile products (sweep waves only)
  unsigned char info[32 * RS];    // zero-initialised like qbi
  unsigned char S[PF_FAST_NMAX + 4];
  int flag;
};

// exterior column j: q5[j] = q5[j-1] scale[1] + sum_i q5[i-1] qb[i,j] expExt(i,j); one wave
template <int NT>
__device__ __forceinline__ void pf_q5_column(PfFastSmem<NT>& sm, const double* __restrict__ QEXT, int ld, int j, int lane,
                                             double sc1) {
  double s = 0.0;
  for (int i = lane + 1; i <= j - TURN - 1; i += WAVE) s += sm.q5[i - 1] * QEXT[j * ld + i];
  s = wave_total_f64_lane63(s);            // DPP scan, fixed order; lane 63 holds the total
  if (lane == WAVE - 1) sm.q5[j] = sm.q5[j - 1] * sc1 + s;
  wave_lds_sync();                         // the next column of the same wave reads q5[j]
}

template <class SM>
__device__ __forceinline__ void pf_prepare_tables(SM& sm, int d, int tid) {
  constexpr int RS = SM::RS;
  if (tid >= 0 && tid < GRES) {
    // entry of the tower slot whose inner diagonal is congruent to tid (mod 28), as seen from diagonal d:
    //   G <- G * keep + (ring[A + i] + ring[B + i]) * asym;   contribution = G * size
    // dead entries: asym = 0, keep = 1, size = 0; entries without an inner pair yet: keep = 0, asym = 0
    const int par = d & 1;
    const int x = (int)((unsigned)(d + 50 - tid) % (unsigned)GRES);     // (d - 6 - rho) mod 28
    const int zero_row = 32 * RS * 8;
    int offA = zero_row, offB = zero_row;
    double eas = 0.0, keep = 1.0, W = 0.0;
    if (x <= 26) {
      const int s = x + 4, dp = d - 6 - x;
      if (dp <= TURN) keep = 0.0;
      else {
        const int base = (dp & 31) * RS * 8;
        offA = base + 3 * 8;
        offB = s == 4 ? zero_row : base + (s - 1) * 8;       // s = 4 has the single shape (2,2)
        eas = sm.tw_as[s];                                   // eninio[s - 4]
        keep = s <= 5 ? 0.0 : 1.0;
        W = sm.tw_W[s];                                      // s >= 6 ? interior[s] scale[s+2] : 0
      }
    }
    sm.tw_i[par][tid][0] = offA; sm.tw_i[par][tid][1] = offB;
    sm.tw_d[par][tid][0] = eas; sm.tw_d[par][tid][1] = keep; sm.tw_d[par][tid][2] = W;
  }
}

__device__ __forceinline__ double lane_table_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

template <class SM, int R0, int R1>
__device__ __forceinline__ double pf_tower_part(const SM& sm, double (&G)[PGSLOTS], int i8, int eA, int eB, double eas,
                                                double ekeep, double eW) {
  const char* ring = reinterpret_cast<const char*>(sm.qbi);
  double a[R1 - R0], b[R1 - R0];
#pragma unroll
  for (int r = R0; r < R1; r++) {
    a[r - R0] = *reinterpret_cast<const double*>(ring + lane_table(eA, r) + i8);
    b[r - R0] = *reinterpret_cast<const double*>(ring + lane_table(eB, r) + i8);
  }
  double acc = 0.0;
#pragma unroll
  for (int r = R0; r < R1; r++) {
    G[r] = G[r] * lane_table_f64(ekeep, r) + (a[r - R0] + b[r - R0]) * lane_table_f64(eas, r);
    acc += G[r] * lane_table_f64(eW, r);
  }
  return acc;
}

template <class SM>
__device__ __forceinline__ double pf_tower_step(const SM& sm, double (&G)[PGSLOTS], int par, int i8, int g, int lane) {
  // lane r fetches the table entry of slot r (one LDS round trip for the wave); words are broadcast with v_readlane
  const int rr = lane * PNG + g;
  const bool on = lane < PGSLOTS && rr < GRES;
  const int zero_row = 32 * SM::RS * 8;
  const int eA = on ? sm.tw_i[par][rr][0] : zero_row, eB = on ? sm.tw_i[par][rr][1] : zero_row;
  const double eas = on ? sm.tw_d[par][rr][0] : 0.0, ekeep = on ? sm.tw_d[par][rr][1] : 1.0, eW = on ? sm.tw_d[par][rr][2] : 0.0;
  const double lo = pf_tower_part<SM, 0, PGSLOTS / 2>(sm, G, i8, eA, eB, eas, ekeep, eW);
  const double hi = pf_tower_part<SM, PGSLOTS / 2, PGSLOTS>(sm, G, i8, eA, eB, eas, ekeep, eW);
  return lo + hi;
}


// ---- tower step of pf_lds_kernel (round 3): entries indexed by the LOOP SIZE s, not by the inner diagonal.
// G_s(i,j) = G_{s-2}(i+1,j-1) + (X[d'][i+3] + X[d'][i+s-1]) eninio[s-4] with d' = d - 2 - s, and the step of a tower from
// diagonal d-2 to d moves every sum from size s-2 to s: in descending order of s that is G[q] <- G[q-1] + ..., a shift that
// the FMA's separate destination register does for free.  Everything per entry is then a compile-time constant except the
// ring row (d - 2 - s) & 31: three VALU for the address, one ds_read2_b64 with immediate column offsets, one broadcast read
// of the entry's two constants, one add and two FMAs (the table-driven form this replaces: eight v_readlane, two adds and
// four fp64 operations per entry, plus a table a finalize wave rebuilt every diagonal).  A wave owns the sizes of one parity
// (s and s-2 must live in the same registers); rows of diagonals that do not exist yet read as zero (see PfFastSmem::qbi).
template <class SM, int SIG, int Q0, int Q1>
__device__ __forceinline__ void pf_tower2_part(const SM& sm, double (&G)[TSL], int dv, int i8, int cbase, double& acc) {
  const char* ring = reinterpret_cast<const char*>(sm.qbi);
  double a[Q1 - Q0], b[Q1 - Q0];
  f64x2 c[Q1 - Q0];
#pragma unroll
  for (int q = Q1 - 1; q >= Q0; q--) {
    const int s = 4 + SIG + 2 * q;
    if (s > 30) continue;
    const int va = ((dv - s) & 31) * (SM::RS * 8) + i8;        // dv = d + 30 in a VGPR: row (d - 2 - s) & 31
    a[q - Q0] = *reinterpret_cast<const double*>(ring + va + 24);
    b[q - Q0] = s == 4 ? 0.0 : *reinterpret_cast<const double*>(ring + va + (s - 1) * 8);     // s = 4 has the single shape (2,2)
    c[q - Q0] = *reinterpret_cast<const f64x2*>(ring + cbase + s * 16);                       // the entry's two constants (same address in every lane)
  }
#pragma unroll
  for (int q = Q1 - 1; q >= Q0; q--) {
    const int s = 4 + SIG + 2 * q;
    if (s > 30) continue;
    G[q] = q > 0 ? fma3_f64(a[q - Q0] + b[q - Q0], c[q - Q0].x, G[q - 1]) : (a[q - Q0] + b[q - Q0]) * c[q - Q0].x;
    acc += G[q] * c[q - Q0].y;
  }
}

template <class SM, int SIG>
__device__ __forceinline__ double pf_tower2_step(const SM& sm, double (&G)[TSL], int dv, int i8) {
  // byte offset of the constants from the ring's start, kept out of the compiler's sight (a literal address costs a v_mov per read)
  const int cbase = as_vector((int)(reinterpret_cast<const char*>(&sm.twc[0][0]) - reinterpret_cast<const char*>(sm.qbi)));
  double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
  pf_tower2_part<SM, SIG, 9, 14>(sm, G, dv, i8, cbase, acc0);     // descending: the upper entries read their predecessors first
  pf_tower2_part<SM, SIG, 4, 9>(sm, G, dv, i8, cbase, acc1);
  pf_tower2_part<SM, SIG, 0, 4>(sm, G, dv, i8, cbase, acc2);
  return (acc0 + acc1) + acc2;
}


// ---- multiloop sums D[i,j] = sum_m qm[i,m-1] qm1[m,j], m = i+TURN+2 .. j-TURN-1, in ONE canonical order (round 4).
// Cells are grouped in TILES of 4 columns i x 4 columns j: tile row a = (i-1) >> 2, tile column c = (j-1) >> 2, block distance
// B = c - a; the tile's cells lie on the diagonals 4B-3 .. 4B+3.  The split points
//        m_lo = 4a + 9 + PKE  <=  m  <=  4c - 3 - PKE = m_hi                       (the FAR range of the tile: 4B - 11 - 2 PKE of them)
// have, for EVERY cell of the tile, both operands on diagonals <= 4B - 5 - PKE, i.e. final PKE + 2 diagonals before the tile's
// first cell is due.  Their part of the sum is a 4 x (m_hi - m_lo + 1) x 4 matrix product: ONE wave computes it with
// v_mfma_f64_4x4x4_4b_f64 -- the instruction's four independent blocks take the split points m_lo + 16 u + 4 block + k, so one
// instruction consumes sixteen of them for all sixteen cells (two 8-byte loads per 256 terms; per-cell sums need two 16-byte
// loads per 2 terms) --, two alternating accumulator chains over u, the four blocks added by two DPP steps:
// far = ((D3 + D2) + (D1 + D0)) of (even u) + (odd u).  What is left per cell is the NEAR part: at most PKE + 3 split points at
// either end of its range (all of them in tiles without a far range, B < KT_BMIN), in 4 rows of KT_U slots -- slot s of the
// cell is its s-th split point from below (s < KT_NL) or its (s - KT_NL)-th from above; row s & 3 walks its slots with two
// alternating fma chains; near = (r0 + r1) + (r2 + r3); D = near + far.
// The far part needs nothing recent, so in small batches a HELPER workgroup on an idle CU computes it from rows the main
// workgroup publishes (pf_kfar_helper), one block distance per round; without a helper (large batches: every CU holds a fold)
// the main workgroup's own item queue takes the tiles of block distance B during the four steps before their first cell is due.
// Either way the same device function runs the same instructions on the same operands: Epf does not depend on how a batch was run.
// (Round 3 walked the far part per cell: 16 bytes per term, which on a full chip is 4.4 GB per launch from beyond the L2 --
// R = 256 ran the fold in 0.83 ms against 0.60 on a quarter of the chip.)
#ifndef DRNA_PKE
#define DRNA_PKE 5
#endif
constexpr int PKE = DRNA_PKE;                       // slack of the far range, in diagonals (>= 4: the main workgroup's own window)
constexpr int KT_NL = PKE + 3;                      // near split points at either end of a cell's range, at most
constexpr int KT_U = (2 * KT_NL + 3) / 4;           // near slots per row
constexpr int KT_BMIN = (12 + 2 * PKE + 3) / 4;     // smallest block distance with a far range
constexpr int KT_D0 = 4 * KT_BMIN - 3;              // first diagonal with a far split point
constexpr int KT_MF = (4 * ((PF_FAST_NMAX - 1) / 4) - 11 - 2 * PKE + 15) / 16;   // products of the longest far range
constexpr int KT_DEPTH = 3;                         // products of a tile in flight (3 / 5 / 11 measured: 0.551 / 0.553 / 0.563 ms at R = 128)
static_assert(PKE >= 4, "a tile's operands must be final when the first of its four steps comes");

// near split points of cell (i, i+d): nl from below, nh from above
__device__ __forceinline__ void kt_near_counts(int i, int d, int& nl, int& nh) {
  const int j = i + d, a = (i - 1) >> 2, c = (j - 1) >> 2, tot = max(d - 2 * TURN - 2, 0);
  if (c - a >= KT_BMIN) { nl = 4 * a + 4 + PKE - i; nh = j - 4 * c - 1 + PKE; }
  else { nl = min(tot, KT_NL); nh = tot - nl; }
}
__device__ __forceinline__ bool kt_has_far(int i, int d) { return ((i + d - 1) >> 2) - ((i - 1) >> 2) >= KT_BMIN; }

// near part: row `row` of four, cells i and i+1 of diagonal d (one 16-byte load fetches an operand of both: they share the
// split point's distance tt = m - 1 - i; a slot that only one of them owns counts as zero for the other)
template <typename RS>
__device__ __forceinline__ void k_near_row(RS rs, int tab8, int ld, int d, int i, int row, double& v0, double& v1) {
  int nl0, nh0, nl1, nh1;
  kt_near_counts(i, d, nl0, nh0);
  kt_near_counts(i + 1, d, nl1, nh1);
  f64x2 a[KT_U], c[KT_U];
#pragma unroll
  for (int u = 0; u < KT_U; u++) {
    const int s = row + 4 * u, y = s - KT_NL;
    const bool low = s < KT_NL;
    const bool ok0 = low ? s < nl0 : y < nh0, ok1 = low ? s < nl1 : y < nh1;
    int tt = low ? TURN + 1 + s : d - TURN - 2 - y;
    tt = (ok0 || ok1) ? tt : TURN + 1;
    a[u] = buf_load_f64x2(rs, (tt * ld + i) * 8, 0);
    c[u] = buf_load_f64x2(rs, tab8 + ((d - tt - 1) * ld + i + tt + 1) * 8, 0);
    if (!ok0) a[u].x = 0.0;
    if (!ok1) a[u].y = 0.0;
  }
  double p0 = 0.0, p1 = 0.0, q0 = 0.0, q1 = 0.0;
#pragma unroll
  for (int u = 0; u < KT_U; u++) {
    if (u & 1) { p1 = fma(a[u].x, c[u].x, p1); q1 = fma(a[u].y, c[u].y, q1); }
    else { p0 = fma(a[u].x, c[u].x, p0); q0 = fma(a[u].y, c[u].y, q0); }
  }
  v0 = p0 + p1; v1 = q0 + q1;
}

// far part of tile (a, a + B): operands from the table at byte 0 (qm) and at tab8 (qm1) of `rs`.  Two halves: k_tile_issue
// requests the operands of the last product and of the first KT_PRE (two 8-byte loads per product: a lane's addresses advance
// by wave-uniform strides of 16 split points, which the buffer instructions take as a scalar offset; only the last product
// clamps and masks), k_tile_finish multiplies, fetches what a long far range has beyond that KT_DEPTH products at a time, and
// hands the sixteen sums to store(d, i, v).
constexpr int KT_PRE = 3;                           // products k_tile_issue requests besides the last one
struct KTile {
  double al, bl, av[KT_PRE], bv[KT_PRE];
  int vA, vB;                // a lane's operand addresses (product 0 of qm, the last but one of qm1)
  int a, c, last;            // (wave-uniform)
  bool mask_last;
};
template <bool SC1, typename RS>
__device__ __forceinline__ void k_tile_issue(RS rs, int tab8, int ld, int n, int a, int B, int lane, KTile& t) {
  const int c = a + B, m_lo = 4 * a + 9 + PKE, m_hi = 4 * c - 3 - PKE;
  const int last = __builtin_amdgcn_readfirstlane((m_hi - m_lo + 16) >> 4) - 1;
  const int q = lane & 3, moff = (lane & 12) + (lane >> 4);
  const int i = 4 * a + 1 + q, jc = min(4 * c + 1 + q, n);         // a column beyond the sequence repeats the last one: its sums are not stored
  // qm[i, m-1] at ((m - 1 - i) ld + i) 8, qm1[m, j] at tab8 + ((j - m) ld + m) 8; product u takes m = m_lo + moff + 16 u
  const int sA = 16 * ld * 8, sB = 16 * (ld - 1) * 8;                                      // bytes per product (qm1 walks downwards)
  const int m0 = m_lo + moff, ml = min(m0 + 16 * last, m_hi);                              // the last product's split point, clamped
  const int vA = (i - (1 + i) * ld + m0 * ld) * 8;                                         // + u sA
  const int vB = tab8 + (jc * ld - (m0 + 16 * last) * (ld - 1)) * 8;                        // + (last - u) sB
  const int vAl = (i - (1 + i) * ld + ml * ld) * 8, vBl = tab8 + (jc * ld - ml * (ld - 1)) * 8;
  t.a = a; t.c = c; t.last = last; t.mask_last = m0 + 16 * last > m_hi; t.vA = vA; t.vB = vB;
#pragma unroll
  for (int u = 0; u < KT_PRE; u++) {
    if (u < last) {
      t.av[u] = SC1 ? buf_load_f64_aux(rs, vA, u * sA) : buf_load_f64(rs, vA, u * sA);
      t.bv[u] = SC1 ? buf_load_f64_aux(rs, vB, (last - u) * sB) : buf_load_f64(rs, vB, (last - u) * sB);
    }
  }
  t.al = SC1 ? buf_load_f64_aux(rs, vAl, 0) : buf_load_f64(rs, vAl, 0);
  t.bl = SC1 ? buf_load_f64_aux(rs, vBl, 0) : buf_load_f64(rs, vBl, 0);
}
template <bool SC1, typename RS, typename ST>
__device__ __forceinline__ void k_tile_finish(RS rs, int ld, const KTile& t, int n, int lane, ST store) {
  const int sA = 16 * ld * 8, sB = 16 * (ld - 1) * 8, last = t.last;
  double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
  for (int u = 0; u < KT_PRE; u++) {
    if (u < last) {
      if (u & 1) acc1 = mfma_f64_4x4x4_4b(t.av[u], t.bv[u], acc1);
      else acc0 = mfma_f64_4x4x4_4b(t.av[u], t.bv[u], acc0);
    }
  }
#pragma unroll
  for (int u0 = KT_PRE; u0 < KT_MF - 1; u0 += KT_DEPTH) {
    if (u0 < last) {
      double av[KT_DEPTH], bv[KT_DEPTH];
#pragma unroll
      for (int w = 0; w < KT_DEPTH; w++) {
        const int u = u0 + w;
        if (u < last && u < KT_MF - 1) {
          av[w] = SC1 ? buf_load_f64_aux(rs, t.vA, u * sA) : buf_load_f64(rs, t.vA, u * sA);
          bv[w] = SC1 ? buf_load_f64_aux(rs, t.vB, (last - u) * sB) : buf_load_f64(rs, t.vB, (last - u) * sB);
        }
      }
#pragma unroll
      for (int w = 0; w < KT_DEPTH; w++) {
        const int u = u0 + w;
        if (u < last && u < KT_MF - 1) {
          if (u & 1) acc1 = mfma_f64_4x4x4_4b(av[w], bv[w], acc1);
          else acc0 = mfma_f64_4x4x4_4b(av[w], bv[w], acc0);
        }
      }
    }
  }
  // (the last product joins the chain its index belongs to: the order of the sums does not depend on the cut)
  const double al = t.mask_last ? 0.0 : t.al;
  if (last & 1) acc1 = mfma_f64_4x4x4_4b(al, t.bl, acc1);
  else acc0 = mfma_f64_4x4x4_4b(al, t.bl, acc0);
  double v = acc0 + acc1;
  v = dpp_add_f64<0x114, 0xF>(v);          // row_shr:4, row_shr:8: the lanes of block 3 end with ((D3 + D2) + (D1 + D0))
  v = dpp_add_f64<0x118, 0xF>(v);
  const int ii = 4 * t.a + 1 + (lane >> 4), jj = 4 * t.c + 1 + (lane & 3);
  if ((lane & 12) == 12 && jj <= n) store(jj - ii, ii, v);
}
// Diagnostic builds only (-DDRNA_SKIP=mask, tools/phase_cost.py): leave out a sweep phase to read its marginal cost
// from the kernel time (results are wrong by construction).  1 = T, 2 = E, 4 = X, 8 = K, 16 = cell finalize, 32 = table / pairable-list
// preparation, 64 = exterior column.
#ifndef DRNA_SKIP
#define DRNA_SKIP 0
#endif
// -DDRNA_TL (tools/timeline.py): every wave of sequence 0's main workgroup stores a raw clock at three points of every step
// (0 after the barrier, 1 after the finalize / tower step, 2 after its last item) into the unused upper half of table 4
#ifdef DRNA_TL
#define TLMARK(ev, k) do { if (tl_on && lane == 0) tl[((wave * 3 + (ev)) << 8) + (k)] = (long long)wall_clock64(); } while (0)
#define TLMARK2(ev, k) do { if (tl_on && lane == 0) tl[((48 + wave * 3 + (ev)) << 8) + (k)] = (long long)wall_clock64(); } while (0)
#define TLMARK3(k) do { if (tl_on && lane == 0) tl[((60 + wave) << 8) + (k)] = (long long)wall_clock64(); } while (0)     // sweep waves: tile products done
#else
#define TLMARK(ev, k) do { } while (0)
#define TLMARK2(ev, k) do { } while (0)
#define TLMARK3(k) do { } while (0)
#endif
#ifndef STAMP
#ifdef DRNA_STAMPS
#define STAMP(k) do { long long _n = clock64(); st_acc[k] += _n - st_last; st_last = _n; } while (0)
#else
#define STAMP(k) do { } while (0)
#endif
#endif

// ---- compacted list of the pairable cells of diagonal d (one wave): i | (pair type, inner neighbours) << 8, the count in the
// row's last word.  With a helper workgroup the main workgroup builds the rows below PFL_D1 only; the helper, idle until the first
// far split point, builds the rest while the main workgroup is on its first diagonals (AGENT: stored write-through, read sc1)
// and says so with its first flag: 7 us less prologue for the main workgroup.
#ifndef DRNA_PFL_D1
#define DRNA_PFL_D1 24
#endif
constexpr int PFL_D1 = DRNA_PFL_D1;
template <bool AGENT, class SM>
__device__ __forceinline__ void pf_pl_row(const SM& sm, int32_t* PL, int ld, int n, int d, int lane) {
  int cntb = 0;
  for (int i0 = 1; i0 <= n - d; i0 += WAVE) {
    const int i = i0 + lane;
    int t = 0;
    if (i <= n - d) t = pair_type(sm.S[i], sm.S[i + d]);
    const unsigned long long m = __ballot(t != 0);
    if (t) {
      const int pos = cntb + __popcll(m & ((1ull << lane) - 1ull));
      const int32_t w = i | ((t * 16 + sm.S[i + 1] * 4 + sm.S[i + d - 1]) << 8);
      if (AGENT) st_agent(&PL[d * ld + pos], w); else PL[d * ld + pos] = w;
    }
    cntb += __popcll(m);
  }
  if (lane == 0) { if (AGENT) st_agent(&PL[d * ld + ld - 1], (int32_t)cntb); else PL[d * ld + ld - 1] = cntb; }
}

// ---- helper workgroup of pf_lds_kernel (small batches: idle CUs): the FAR multiloop split points of every tile, from the
// rows the main workgroup publishes.  Hand-over as the CDNA4 guide prescribes (and as fold_mfe_dual.hpp does it): payload by
// sc1 stores into tables of its own (XQM, XQM1: tables 0 and 1 of the sequence's workspace, which this kernel does not use
// otherwise -- the main workgroup's own reads stay on plain-stored QM / QM1, whose lines remain in its L2), every storing wave
// drained, workgroup barrier, ONE lane stores the flag; the consumer polls the flag from one wave and loads the payload
// only afterwards, sc1.  One-way slack instead of a per-step round trip: the tiles of block distance B need rows <= 4B - 5 - PKE
// only and their first cell is due at diagonal 4B - 3, so a round (one block distance: up to 44 tiles for sixteen waves) starts
// PKE steps before its results are asked for.  Results travel back through DFAR (table 5); the flag says up to which diagonal
// every cell has its far sum.
template <int NT>
__device__ __forceinline__ void pf_kfar_helper(const PfArgs& A, const EvalArgs& EV, PfFastSmem<NT>& sm, int r, int n) {
  constexpr int NW = NT / WAVE;
  const int ld = A.ld, tid = threadIdx.x, lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(wave_id());
  double* base = A.ws + (long long)r * A.ws_stride;
  const long long tab = (long long)ld * ld;
  double* DFAR = base + 5 * tab;
  const auto rsX = __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, (int)(2 * tab * 8), 0x00020000);     // XQM, XQM1
  const int* flagA = A.hflags + (long long)r * 64;
  int* flagB = A.hflags + (long long)r * 64 + 32;
  int* ctl = reinterpret_cast<int*>(&sm);                     // (the ring is not used here)
  if (tid == 0) ctl[0] = 0;
  // E(target structures) of this sequence (eval_structure.hpp), while the main workgroup works towards the first far split point:
  // a launch of its own would need CUs of its own, and in the batches that get a helper every CU holds a fold workgroup
  if (EV.n_targets > 0) {
    constexpr int NEV = NW < 8 ? NW : 8;
    EvalSmem* es = reinterpret_cast<EvalSmem*>(ctl + 32);
    if (wave < NEV)
      for (int k = wave; k < EV.n_targets; k += NEV) eval_one(es[wave], EV, r, k, lane);
  }
  // the main workgroup's pairable lists from diagonal PFL_D1 on (pf_pl_row), announced by the first flag: far sums complete below
  // the first diagonal that has any
  {
    const char* seq = A.seqs + (A.rg.off ? (long long)A.rg.off[r] : (long long)r * n);
    for (int k = tid; k < n; k += NT) { const int c = enc_nt(seq[k]); sm.S[k + 1] = (unsigned char)(c < 0 ? 0 : c); }
    if (tid == 0) { sm.S[0] = 0; sm.S[n + 1] = 0; }
    __syncthreads();
    int32_t* PL = reinterpret_cast<int32_t*>(base + 4 * tab);
    for (int d = PFL_D1 + wave; d < n; d += NW) pf_pl_row<true>(sm, PL, ld, n, d, lane);
    drain_vmem();
  }
  __syncthreads();
  if (tid == 0) st_agent(flagB, A.hbase + KT_D0 - 1);
  const int Bmax = (n - 1) >> 2;
  for (int B = KT_BMIN; B <= Bmax; B++) {
    if (wave == 0) {
      int seen = 0;
      const bool ok = strip_wait(flagA, A.hbase, 4 * B - 5 - PKE, seen);
      if (lane == 0 && (!ok || seen == A.hbase + STRIP_DONE)) ctl[0] = 1;        // the main workgroup is gone (bad character) or lost
    }
    __syncthreads();
    if (ctl[0]) break;
    if (!(DRNA_SKIP & 8))                                                         // (DRNA_SKIP: timing builds)
      for (int a = wave; a <= Bmax - B; a += NW) {
        KTile t;
        k_tile_issue<true>(rsX, (int)tab * 8, ld, n, a, B, lane, t);
        k_tile_finish<true>(rsX, ld, t, n, lane, [&](int d, int i, double v) { st_agent(&DFAR[d * ld + i], v); });
      }
    drain_vmem();
    __syncthreads();
    if (tid == 0) st_agent(flagB, A.hbase + (B == Bmax ? n : 4 * B));           // every diagonal below the next round's first cell is complete
  }
}

// bx_in / helper_in: the caller's own block -> (sequence slot, role) mapping (fold_fused.hpp); -1 = this kernel's grid (2 bx + role
// with a helper workgroup per sequence, bx without).  TILES = false: an instance for launches WITH helper workgroups only -- it
// holds no tile code in its sweep loop, so the cell finalize can keep its five constants in registers (with the tile code the
// kernel is at the register limit and they spill into the finalize chain; read from LDS they cost the floor build 12 us)
template <int NT, bool TILES = true>
__device__ __forceinline__ void pf_lds_body(PfFastSmem<NT>& sm, PfArgs A, const EvalArgs& EV, int bx_in = -1, int helper_in = -1) {
  constexpr int NW = NT / WAVE;
  constexpr int RS = PfFastSmem<NT>::RS;
  const PfTables& T = *A.T;
  const bool hm = A.helper != 0;                     // a helper workgroup per sequence computes the far multiloop split points (emulator: odd blocks; kernel: pair_block)
  const int bx = bx_in >= 0 ? bx_in : hm ? blockIdx.x >> 1 : blockIdx.x;
  const bool is_helper = hm && (helper_in >= 0 ? helper_in != 0 : (blockIdx.x & 1) != 0);
  const int r = A.rg.idx ? A.rg.idx[bx] : bx;
  if (A.rg.len) A.L = A.rg.len[r];
  const int n = A.L, ld = A.ld;
  if (is_helper) {
    if (A.helper == 2) return;                        // fault injection (tests): the helper never shows up, the main workgroup's wait expires
    pf_kfar_helper<NT>(A, EV, sm, r, n);
    return;
  }
  const int tid = threadIdx.x, lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(wave_id());

  double* base = A.ws + (long long)r * A.ws_stride;
  const long long tab = (long long)ld * ld;
  double* QM = base + 2 * tab;
  double* QM1 = base + 3 * tab;
  int32_t* PL = reinterpret_cast<int32_t*>(base + 4 * tab);   // compacted pairable-cell lists, one row per diagonal
  double* QEXT = base + 6 * tab;

#ifdef DRNA_TL
  long long* tl = reinterpret_cast<long long*>(base + 4 * tab + tab / 2);
  const bool tl_on = r == 0;
  TLMARK(0, 0);                       // kernel entry (steps start at TURN + 1: slots 0 .. 3 are free); 1: prologue done; 2, 3: epilogue
#endif
  const double eTau = T.TermAU, eMLc = T.MLclosing, eMLi = T.MLintern;
  const double b1 = A.eMLb[1], sc1 = A.scale[1], sc2 = A.scale[2];

  // ---- prologue
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
  for (int t = tid; t < 36; t += NT) {
    sm.eWt[t][0] = t >= 2 && t <= 30 ? T.bulge[t] * A.scale[t + 2] : 0.0;
    sm.eWt[t][1] = t >= 4 && t <= 30 ? T.interior[t] * T.eninio[t - 2] * A.scale[t + 2] : 0.0;
  }
  if (tid == 0) {
    sm.xc[0] = T.bulge[1] * A.scale[3]; sm.xc[1] = T.interior[5] * T.eninio[1] * A.scale[7];
    sm.xc[2] = A.scale[4]; sm.xc[3] = A.scale[5]; sm.xc[4] = A.scale[6];
    sm.xc[8] = eTau; sm.xc[9] = eMLc; sm.xc[10] = eMLi; sm.xc[11] = b1; sm.xc[12] = sc2;
  }
  for (int k = tid; k < 32; k += NT) {
    sm.twc[k][0] = k >= 4 && k <= 30 ? T.eninio[k - 4] : 0.0;
    sm.twc[k][1] = k >= 6 && k <= 30 ? T.interior[k] * A.scale[k + 2] : 0.0;
  }
  for (int k = tid; k < 4 * RS; k += NT) sm.dring[k] = 0.0;
  for (int k = tid; k < 32 * RS; k += NT) { sm.qbi[k] = 0.0; sm.info[k] = 0; }
  for (int k = tid; k < 2 * RS; k += NT) { (&sm.qm1row[0][0])[k] = 0.0; (&sm.urow[0][0])[k] = 0.0; }
  for (int k = tid; k < 2 * 2 * PfFastSmem<NT>::NSLOT; k += NT) (&sm.partG[0][0][0])[k] = 0.0;
  for (int k = tid; k < 2 * 4 * PfFastSmem<NT>::NSLOT; k += NT) (&sm.partK[0][0][0])[k] = 0.0;
  for (int k = tid; k < 2 * PfFastSmem<NT>::NSLOT; k += NT) (&sm.accE[0][0])[k] = 0.0;
  for (int k = tid; k < 6 * PfFastSmem<NT>::NSLOT; k += NT) (&sm.accX[0][0][0])[k] = 0.0;
  if (tid == 0) { sm.flag = 0; sm.q5[0] = 1.0; }
  __syncthreads();
  const char* seq = A.seqs + (A.rg.off ? (long long)A.rg.off[r] : (long long)r * n);
  for (int k = tid; k < n; k += NT) {
    const int c = enc_nt(seq[k]);
    if (c < 0) sm.flag = 1;
    sm.S[k + 1] = (unsigned char)(c < 0 ? 0 : c);
  }
  __syncthreads();
  if (tid == 0) { sm.S[0] = sm.S[n]; sm.S[n + 1] = sm.S[1]; }
  __syncthreads();
  if (sm.flag) {
    if (tid == 0) { A.status[r] = ST_BAD_CHAR; A.Epf[r] = 0.0; if (hm) st_agent(A.hflags + (long long)r * 64, A.hbase + STRIP_DONE); }
    return;
  }
  if (wave == 0) {                                   // q5[j] = scale^j while no pair fits (j <= TURN + 1)
    for (int j = 1; j <= n && j <= TURN + 1; j++) sm.q5[j] = sm.q5[j - 1] * sc1;
  }
  // compacted list of pairable cells of every diagonal (HBM/L2); with a helper workgroup the rows from PFL_D1 on are the helper's
  for (int d = TURN + 1 + wave; d < (hm ? min(n, PFL_D1) : n); d += NW) pf_pl_row<false>(sm, PL, ld, n, d, lane);
  __syncthreads();

  // tower blocks, centred on the sequence
  const int NB = (n + WAVE - 1) / WAVE;
  const int off0 = (NB * WAVE - n) / 2;
  const int aw = wave - NB;                // index among the sweep waves (< 0: finalize wave)
  // Tower roles.  Generic loops exist from diagonal 10 on, where n - 10 cells are left: NBT = ceil((n - 10) / 64) blocks of 64 tower
  // slots starting at slot T0 cover them for the rest of the fill (the slot range only shrinks).  1024 threads: a block gets
  // four waves, parity of the loop size x parity of the diagonal -- such a wave works every other step and carries 14 sums;
  // n = 200: three blocks, twelve roles, every sweep wave has one.  Smaller workgroups (CPU emulation of n <= 64): two waves
  // per block, both diagonal parities each.
  constexpr bool TWO_PAR = NT < 1024;
  const int nT = n - 10, NBT = nT > 0 ? (nT + WAVE - 1) / WAVE : 0;
  const int T0 = max(0, off0 + 5 - (NBT * WAVE - nT) / 2);
  int my_tb = -1, my_sig = 0, my_pm = 0;             // block, parity of s, mask of the diagonal parities taken (0: no tower role)
  if (aw >= 0) {
    if (TWO_PAR) { if (aw < 2 * NBT) { my_tb = aw >> 1; my_sig = aw & 1; my_pm = 3; } }
    else if (aw < 4 * NBT) { my_tb = aw >> 2; const int x = (aw + my_tb) & 3; my_sig = x & 1; my_pm = 1 << (x >> 1); }
  }
  const int w_pl = NB > 2 ? 2 : 0, w_q5 = NB - 1;     // finalize waves that take the side jobs
  // (The list staging by two sweep waves taking turns, which fold_mfe_lds.hpp has, was measured here and dropped: this kernel's early
  // steps are item-bound on exactly those waves -- 0.430 -> 0.438 ms at R = 64, 0.531 -> 0.542 at R = 128.)

  if (aw < 0) {
    const int d = TURN + 1;
    if (d < n) {
      const int cnt = PL[d * ld + ld - 1];
      if (tid < cnt) sm.plist[d & 1][tid] = PL[d * ld + tid];
      if (tid == 0) { sm.pcnt[d & 1] = cnt; sm.qhead[0] = 0; sm.qhead[1] = 0; sm.qtile[0] = 0; sm.qtile[1] = 0; }
    }
  }
  __syncthreads();

#ifdef DRNA_STAMPS
  long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long st_last = clock64();
  long long* dbg = reinterpret_cast<long long*>(base + 5 * tab);     // the U table is unused by this kernel
#endif
  // X: weights of the nine fixed shapes (0,0) (0,1) (1,0) (1,1) (1,2) (2,1) (2,2) (2,3) (3,2)

  // QM and QM1 (adjacent tables) through one buffer descriptor
  const auto rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)QM, (short)0, (int)(2 * tab * 8), 0x00020000);


  double* XQM = base, *XQM1 = base + tab, *DFAR = base + 5 * tab;      // exchange tables of the helper hand-over (tables 0, 1, 5)
  int* flagA = hm ? A.hflags + (long long)r * 64 : nullptr;
  const int* flagB = hm ? A.hflags + (long long)r * 64 + 32 : nullptr;
  double dfar_cur = 0.0, dfar_next = 0.0;
  int fb_last = 0;                  // (a zero flag never compares as published: epochs start at 1)
  bool helper_lost = false, pl_seen = false;
  // Floating work items of diagonal d, taken from a work queue (LDS counter).  The sweep waves run this after their tower
  // step; the finalize waves, which are done with diagonal d-1 long before the sweep of d ends, join in: every item owns
  // its output slots and reads nothing the current step writes, so the result does not depend on who takes it.
  auto run_items = [&](const int d) {
    const int ncell = n - d, sh = d >> 1, par = d & 1;
    const int slot0 = sh + off0 - 1;          // tower slot of column i is i + slot0
    const int pcnt = __builtin_amdgcn_readfirstlane(sm.pcnt[par]);
    // ---- floating items of the diagonal, taken from a work queue (LDS counter): 16-cell multiloop
    // sub-blocks (K, from L2), pairs of pairable cells for the 112 bulge / 1xn shapes (E), three groups of
    // fixed small shapes per 64 pairable cells (X).  Every item owns its output slot(s), so the result does
    // not depend on which wave takes it.
    // K items: per 32-cell block one item for the NEAR split points (the far ones: run_tiles, or the helper workgroup)
    const int nblk = (ncell + 31) >> 5;
    const int nK = ((DRNA_SKIP & 8) || d < 2 * TURN + 3) ? 0 : nblk;
    const int nE = (DRNA_SKIP & 2) ? 0 : (pcnt + 3) >> 2, nX = (DRNA_SKIP & 4) ? 0 : 3 * ((pcnt + WAVE - 1) / WAVE);
    const int nItems = __builtin_amdgcn_readfirstlane(nK + nE + nX);
    for (int it0 = queue_pop(&sm.qhead[par], lane); it0 < nItems; it0 = queue_pop(&sm.qhead[par], lane)) {
      // The item index below is in the order K, E, X; the QUEUE hands them out as X, K, E: the items with global loads (tables of
      // the 1x2 / 2x2 loops, multiloop operands) first, the bulge / 1xn items -- LDS only -- last, for the finalize waves, which join
      // late and whose table stores a later global load would have to wait for (K, E, X: 0.428 ms; X, K, E: 0.417; K, X, E: 0.420; X, E, K: 0.457)
      const int it = it0 < nX ? nK + nE + it0 : it0 - nX;
      if (it < nK) {
        // ---- K near: a lane owns two adjacent cells and fetches both operands of both with one 16-byte load each (the
        // vector-memory pipe, not the ALU, bounds this sweep: half the instructions, whole 128-byte lines per row)
        const int blk = it, cl = lane & 15, row = lane >> 4;
        int i = (blk << 5) + 2 * cl + 1;
        const bool act0 = i <= ncell, act1 = i + 1 <= ncell;
        i = act0 ? i : 1;
        double v0, v1;
        k_near_row(rsQ, (int)tab * 8, ld, d, i, row, v0, v1);
        if (act0) sm.partK[par][row][i + slot0] = v0;
        if (act1) sm.partK[par][row][i + 1 + slot0] = v1;
      } else if (it < nK + nE) {
        // ---- E: four pairable cells per item, one per 16-lane row.  Lane l of a row takes the loop SIZES t = l + 2 and l + 18
        // (2 .. 30): the two bulges (0,t) (t,0) and the two 1xn loops (1,t-1) (t-1,1) of one size have their inner pairs on
        // ONE ring row, d - 2 - t, at columns i+1, i+t+1 and i+2, i+t -- one row address and two pairs of adjacent cells per
        // size instead of four separately decoded shape slots, and the size weights come straight from a table indexed by
        // the lane (no per-slot shape words).  Row sums by four DPP steps for all four cells at once, lane 15 of a row writes.
        const int q = 4 * (it - nK) + (lane >> 4);
        const int pe = sm.plist[par][q < pcnt ? q : pcnt - 1];
        const int i0 = pe & 255, ij = pe >> 8;
        const int tA = (lane & 15) + 2, tBr = tA + 16, tB = tBr > 30 ? 30 : tBr;      // sizes beyond 30 weigh nothing (eWt) and read size 30's cells
        const f64x2 wA = *reinterpret_cast<const f64x2*>(&sm.eWt[tA][0]), wB = *reinterpret_cast<const f64x2*>(&sm.eWt[tBr][0]);
        const int oA = ((d - 2 - tA) & 31) * RS + i0, oB = ((d - 2 - tB) & 31) * RS + i0;   // rows of diagonals <= TURN read as zero
        const double a1 = sm.qbi[oA + 1], a2 = sm.qbi[oA + 2], a3 = sm.qbi[oA + tA], a4 = sm.qbi[oA + tA + 1];
        const double b1_ = sm.qbi[oB + 1], b2_ = sm.qbi[oB + 2], b3_ = sm.qbi[oB + tB], b4_ = sm.qbi[oB + tB + 1];
        const int fa1 = sm.info[oA + 1], fa2 = sm.info[oA + 2], fa3 = sm.info[oA + tA], fa4 = sm.info[oA + tA + 1];
        const int fb1 = sm.info[oB + 1], fb2 = sm.info[oB + 2], fb3 = sm.info[oB + tB], fb4 = sm.info[oB + tB + 1];
        const double ra1 = sm.rbul[fa1], ra4 = sm.rbul[fa4], ra2 = sm.r1n[fa2], ra3 = sm.r1n[fa3];
        const double rb1 = sm.rbul[fb1], rb4 = sm.rbul[fb4], rb2 = sm.r1n[fb2], rb3 = sm.r1n[fb3];
        const double bul = (a1 * ra1 + a4 * ra4) * wA.x + (b1_ * rb1 + b4_ * rb4) * wB.x;
        const double one = (a2 * ra2 + a3 * ra3) * wA.y + (b2_ * rb2 + b3_ * rb3) * wB.y;
        double v = bul * ((ij >> 4) > 2 ? sm.xc[as_vector(8)] : 1.0) + one * sm.mm1n[ij];
        v = dpp_add_f64<0x111, 0xF>(v);
        v = dpp_add_f64<0x112, 0xF>(v);
        v = dpp_add_f64<0x114, 0xF>(v);
        v = dpp_add_f64<0x118, 0xF>(v);
        if ((lane & 15) == 15 && q < pcnt) sm.accE[par][i0 + slot0] = v;
      } else {
        const int xi = it - nK - nE;
        const int ch = xi / 3, grp = xi - 3 * ch;
        const int q = ch * WAVE + lane;
        const int pe = sm.plist[par][q < pcnt ? q : pcnt - 1];
        const int i = pe & 255, cxv = pe >> 8, t = cxv >> 4, si1 = (cxv >> 2) & 3, sj1 = cxv & 3;
        double sum;
        if (grp == 0) {
          // (0,0) (0,1) (1,0) (1,1): LDS tables
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
          sum = w[0] * sm.rinv[f[0]] * sm.stack[t * 8 + (f[0] >> 4)] * sm.xc[as_vector(12)];
          sum += (w[1] * sm.rinv[f[1]] * sm.stack[t * 8 + (f[1] >> 4)] + w[2] * sm.rinv[f[2]] * sm.stack[t * 8 + (f[2] >> 4)]) * sm.xc[as_vector(0)];
          sum += w[3] * sm.rinv[f[3]] * sm.int11[(t * 8 + (f[3] >> 4)) * 16 + si1 * 4 + sj1] * sm.xc[as_vector(2)];
        } else if (grp == 1) {
          // (1,2) (2,1) (2,2): tables in global memory (L2)
          const int dpa = d - 5, dpb = d - 6;
          const int oa = (dpa & 31) * RS + 2 + i, ob = (dpa & 31) * RS + 3 + i, oc = (dpb & 31) * RS + 3 + i;
          const double wa = dpa > TURN ? sm.qbi[oa] : 0.0, wb = dpa > TURN ? sm.qbi[ob] : 0.0, wc = dpb > TURN ? sm.qbi[oc] : 0.0;
          const int fa = dpa > TURN ? sm.info[oa] : 0, fb = dpa > TURN ? sm.info[ob] : 0, fc = dpb > TURN ? sm.info[oc] : 0;
          const double ga = T.int21[(t * 8 + (fa >> 4)) * 64 + si1 * 16 + ((fa >> 2) & 3) * 4 + sj1];
          const double gb = T.int21[((fb >> 4) * 8 + t) * 64 + ((fb >> 2) & 3) * 16 + si1 * 4 + (fb & 3)];
          const double gc = T.int22[(t * 8 + (fc >> 4)) * 256 + si1 * 64 + (fc & 3) * 16 + ((fc >> 2) & 3) * 4 + sj1];
          sum = (wa * sm.rinv[fa] * ga + wb * sm.rinv[fb] * gb) * sm.xc[as_vector(3)] + wc * sm.rinv[fc] * gc * sm.xc[as_vector(4)];
        } else {
          // (2,3) (3,2)
          const int dp = d - 7;
          const int oa = (dp & 31) * RS + 3 + i, ob = (dp & 31) * RS + 4 + i;
          const double wa = dp > TURN ? sm.qbi[oa] : 0.0, wb = dp > TURN ? sm.qbi[ob] : 0.0;
          const int fa = dp > TURN ? sm.info[oa] : 0, fb = dp > TURN ? sm.info[ob] : 0;
          sum = (wa * sm.r23[fa] + wb * sm.r23[fb]) * sm.mm23[cxv] * sm.xc[as_vector(1)];
        }
        if (q < pcnt) sm.accX[par][grp][i + slot0] = sum;
      }
#ifdef DRNA_STAMPS
      STAMP(it < nK ? 5 : it < nK + nE ? 1 : 2);
#endif
    }
  };

  // Tile products (no helper workgroup: large batches).  During step d the tiles of block distance Bt, whose first cell is due
  // 4 - ph diagonals from now, a quarter of them per step (neighbours in one step: they share cache lines); operands on
  // diagonals <= 4 Bt - 5 - PKE <= d - 3, sums stored (DFAR) a step before the finalize waves ask for them.  A queue of its own
  // that only the sweep waves serve, first thing after their tower step: in the finalize waves' loop the tile code costs spilled
  // registers on their critical path (0.50 instead of 0.46 ms at R = 64).  (Measured and dropped: the operands requested as the
  // last thing of the step before and left in flight across the barrier, so that a tile costs its wave the matrix instructions
  // and not an L2 round trip -- 4 to 24 operand registers carried over the loop's back edge are 6 to 77 spilled ones:
  // 0.56 / 0.61 / 0.63 / 0.65 ms for 0 / 1 / 2 / 4 products carried, against 0.54.)
  auto run_tiles = [&](const int d) {
    const int Bmax = (n - 1) >> 2, Bt = ((d + 3) >> 2) + 1, ph = (d + 3) & 3;
    if (!TILES || (DRNA_SKIP & (8 | 512)) || hm || Bt < KT_BMIN || Bt > Bmax) return;         // (512: timing build without the far split points)
    const int cnt = Bmax - Bt + 1, per = (cnt + 3) >> 2, a0 = ph * per;
    const int nT = __builtin_amdgcn_readfirstlane(max(0, min(cnt, a0 + per) - a0));
    for (int it = queue_pop(&sm.qtile[d & 1], lane); it < nT; it = queue_pop(&sm.qtile[d & 1], lane)) {
      KTile t;
      k_tile_issue<false>(rsQ, (int)tab * 8, ld, n, a0 + it, Bt, lane, t);
      k_tile_finish<false>(rsQ, ld, t, n, lane, [&](int dd, int ii, double v) { DFAR[dd * ld + ii] = v; });
    }
  };

  if (aw < 0) {
    // ================= finalize waves: diagonal d = k-1 at step k
    // (the cell finalize's constants in registers: values of THIS branch only, read from LDS once, so that they do not occupy
    // registers -- or spill slots reloaded inside the finalize chain -- through the sweep waves' loop, tile code or not)
    double fTau = sm.xc[as_vector(8)], fMLc = sm.xc[as_vector(9)], fMLi = sm.xc[as_vector(10)], fb1 = sm.xc[as_vector(11)], fsc2 = sm.xc[as_vector(12)];
    // the staged list row travels in registers from the step that requests it to the next one (wave w_pl)
    int pl_cnt = 0, pl0 = 0, pl1 = 0, pl2 = 0, pl3 = 0;
    auto pl_request = [&](const int dn) {
      const int32_t* row = PL + dn * ld;
      if (hm && dn >= PFL_D1) {                                       // the helper's rows: after its first flag, sc1
        if (!pl_seen) {
          int seen = 0;
          if (!helper_lost && !flag_ge(fb_last, A.hbase + KT_D0 - 1) && !strip_wait(flagB, A.hbase, KT_D0 - 1, seen)) { sm.flag = 2; helper_lost = true; }
          pl_seen = true;
        }
        pl_cnt = ld_agent(row + ld - 1);
        pl0 = ld_agent(row + lane); pl1 = ld_agent(row + lane + WAVE); pl2 = ld_agent(row + lane + 2 * WAVE);
        pl3 = ld_agent(row + min(lane + 3 * WAVE, ld - 1));
      } else {
        pl_cnt = row[ld - 1];
        pl0 = row[lane]; pl1 = row[lane + WAVE]; pl2 = row[lane + 2 * WAVE]; pl3 = row[min(lane + 3 * WAVE, ld - 1)];
      }
    };
    if (!(DRNA_SKIP & 32) && wave == w_pl && TURN + 2 < n) pl_request(TURN + 2);
    for (int k = TURN + 1; k <= n; k++) {
      const int d = k - 1;
      TLMARK(0, k);
      // side jobs of the step, one finalize wave each (when there are that many): pairable list of diagonal k+1, exterior
      // column j = k-3 (its cells were stored in step <= k-3 and drained by that step's barrier).  Their global loads are
      // REQUESTED here, before this step's stores, and consumed after the cell finalize: vector-memory operations retire in
      // order, so a wait for them does not wait for the (write-through) stores behind them.
      // The list row of diagonal k+1 was requested a whole step ago (with a helper workgroup the late rows are the helper's: they
      // come from beyond this XCD's L2, ~2 us, and with the request at the top of the SAME step the staging wave reached the
      // barrier last once the steps were down to the finalize chain): it goes into LDS here -- the items of this step read the
      // other parity's buffer -- and the row of diagonal k+2 is requested into the same registers.
      if (!(DRNA_SKIP & 32) && wave == w_pl) {
        if (k + 1 < n) {
          int* dst = sm.plist[(k + 1) & 1];
          dst[lane] = pl0; dst[lane + WAVE] = pl1; dst[lane + 2 * WAVE] = pl2;
          if (lane + 3 * WAVE < PfFastSmem<NT>::NL) dst[lane + 3 * WAVE] = pl3;
          if (lane == 0) { sm.pcnt[(k + 1) & 1] = pl_cnt; sm.qhead[(k + 1) & 1] = 0; sm.qtile[(k + 1) & 1] = 0; }
        }
        if (k + 2 < n) pl_request(k + 2);
      }
      const bool job_q5 = !(DRNA_SKIP & 64) && wave == w_q5 && k - 3 >= TURN + 2;
      double qx0 = 0.0, qx1 = 0.0, qx2 = 0.0, qx3 = 0.0;
      if (job_q5) {
        const int j = k - 3, top = j - TURN - 1;                   // i = 1 .. top
        const double* col = QEXT + j * ld;
        qx0 = lane + 1 <= top ? col[lane + 1] : 0.0;
        qx1 = lane + 1 + WAVE <= top ? col[lane + 1 + WAVE] : 0.0;
        qx2 = lane + 1 + 2 * WAVE <= top ? col[lane + 1 + 2 * WAVE] : 0.0;
        qx3 = lane + 1 + 3 * WAVE <= top ? col[lane + 1 + 3 * WAVE] : 0.0;
      }
      int f_req = 0;
      double d_req = 0.0;
      dfar_cur = dfar_next;
      if (hm) {
        // diagonal k-2 was finalized in step k-1 and every store of it drained by that step's barrier: publish it
        if (tid == 0 && k - 2 > TURN) st_agent(flagA, A.hbase + (k - 2));
        // The helper's flag and the far sums of the next diagonal are REQUESTED here and taken over at the end of the step
        // (after the items): assigning the loop-carried variables here would make the compiler wait for the loads at once.
        f_req = ld_agent(flagB);
        if (k >= KT_D0 && k < n && !helper_lost && !flag_ge(fb_last, A.hbase + k)) {
          // the flag value of the previous step decides (the helper is ahead as a rule; an old value will do); else wait here
          int seen = 0;
          if (!strip_wait(flagB, A.hbase, k, seen)) { sm.flag = 2; helper_lost = true; }     // one expired wait per wave, then no more
        }
      }
      if (k >= KT_D0 && k < n) {
        // far sums of diagonal k: the helper's (sc1) or this workgroup's own tile items', stored in a step before this one
        const int ic = tid + 1 - (k >> 1) - off0;
        if (ic >= 1 && ic <= n - k && kt_has_far(ic, k)) d_req = hm ? ld_agent(&DFAR[k * ld + ic]) : DFAR[k * ld + ic];
      }
      TLMARK2(0, k);
      if (d > TURN) {
        const int ncell = n - d, sh = d >> 1, par = d & 1;
        const int i = tid + 1 - sh - off0;
        if (!(DRNA_SKIP & 16) && i >= 1 && i <= ncell) {
          // every producer writes its slot of every live cell on every diagonal it exists for, so nothing is zeroed here:
          // a family that does not exist yet (or a cell that cannot pair) is simply not read
          const int j = i + d;
          const int si = sm.S[i], sj = sm.S[j], si1 = sm.S[i + 1],
                    sj1 = sm.S[j - 1], sim = sm.S[i - 1], sjp = sm.S[j + 1];
          const double aG = d >= 10 ? sm.partG[par][0][tid] + sm.partG[par][1][tid] : 0.0;
          const double aKn = d >= 2 * TURN + 3 ? (sm.partK[par][0][tid] + sm.partK[par][1][tid]) + (sm.partK[par][2][tid] + sm.partK[par][3][tid]) : 0.0;
          const double aK = aKn + dfar_cur;        // (zero for a cell without far split points)
          const int t = pair_type(si, sj);
          const double cTau = fTau, cMLc = fMLc, cMLi = fMLi, cb1 = fb1, csc2 = fsc2;
          const double tau = t > 2 ? cTau : 1.0;
          const int ij = t * 16 + si1 * 4 + sj1, rt = rtype_of(t);
          const int info = t ? (rt << 4) | (sjp << 2) | sim : 0;
          const double aE = sm.accE[par][tid], aX = (sm.accX[par][0][tid] + sm.accX[par][1][tid]) + sm.accX[par][2][tid];
          const double wH = sm.mmH[ij], wI = sm.mmI[ij], wMc = sm.mmM[rt * 16 + sj1 * 4 + si1], wInfo = sm.mmI[info];
          const double dprev = sm.dring[((d - 2) & 3) * RS + i + 1];
          const int ex = t * 16 + sim * 4 + sjp;
          const double wExt = sm.mmExt[ex], wMs = sm.mmM[ex], w5 = sm.d5[t * 4 + sim], w3 = sm.d3[t * 4 + sjp];
          const int pp = (d - 1) & 1;
          const double m1p = sm.qm1row[pp][i], m1q = sm.qm1row[pp][i + 1], up = sm.urow[pp][i + 1];
          double qb = 0.0;
          if (t) {
            const int u = d - 1;
            double hp;
            if (u == 3 || u == 4 || u == 6) {
              // special hairpins (tri / tetra / hexa loops) go through the general routine
              int code = 0;
              const int len = u + 2;
              for (int q = 0; q < len; q++) code |= sm.S[i + q] << (2 * q);
              hp = -1.0;
              if (u == 3) { for (int q = 0; q < T.n_tri; q++) if (T.tri_code[q] == code) hp = T.tri_w[q] * A.scale[u + 2]; if (hp < 0.0) hp = sm.hpw[u] * tau; }
              else if (u == 4) { for (int q = 0; q < T.n_tetra; q++) if (T.tetra_code[q] == code) hp = T.tetra_w[q] * A.scale[u + 2]; }
              else { for (int q = 0; q < T.n_hexa; q++) if (T.hexa_code[q] == code) hp = T.hexa_w[q] * A.scale[u + 2]; }
              if (hp < 0.0) hp = sm.hpw[u] * wH;
            } else {
              hp = sm.hpw[u] * wH;
            }
            qb = hp + aE + aX + aG * wI;
            qb += dprev * cMLc * cMLi * tau * wMc * csc2;
          }
          sm.qbi[(d & 31) * RS + i] = qb * wInfo;
          sm.info[(d & 31) * RS + i] = (unsigned char)info;
          const double me = (i > 1 && j < n) ? wExt : i > 1 ? w5 : j < n ? w3 : 1.0;
          const double mm = (i > 1 && j < n) ? wMs : i > 1 ? w5 : j < n ? w3 : 1.0;
          const double ext = t ? qb * tau * me : 0.0, stem = t ? qb * cMLi * tau * mm : 0.0;
          const double m1 = m1p * cb1 + stem;
          const double U = cb1 * (m1q + up);
          sm.qm1row[par][i] = m1;
          sm.urow[par][i] = U;
          sm.dring[(d & 3) * RS + i] = aK;
          {
            QEXT[j * ld + i] = ext;
            QM1[d * ld + i] = m1;
            QM[d * ld + i] = m1 + aK + U;
            if (hm) { st_agent(&XQM1[d * ld + i], m1); st_agent(&XQM[d * ld + i], m1 + aK + U); }
          }
        }
      }
      TLMARK2(1, k);
      if (job_q5) {
        // q5[j] = q5[j-1] scale[1] + sum_i q5[i-1] qb[i,j] expExt(i,j): four strided terms per lane, fixed-order wave sum
        const int j = k - 3, top = j - TURN - 1;
        double sq = 0.0;
        if (lane + 1 <= top) sq += sm.q5[lane] * qx0;
        if (lane + 1 + WAVE <= top) sq += sm.q5[lane + WAVE] * qx1;
        if (lane + 1 + 2 * WAVE <= top) sq += sm.q5[lane + 2 * WAVE] * qx2;
        if (lane + 1 + 3 * WAVE <= top) sq += sm.q5[lane + 3 * WAVE] * qx3;
        sq = wave_total_f64_lane63(sq);          // DPP scan (no LDS traffic); lane 63 holds the total
        if (lane == WAVE - 1) sm.q5[j] = sm.q5[j - 1] * sc1 + sq;
      }
      STAMP(4);
      TLMARK(1, k);
      if (!(DRNA_SKIP & 128) && k < n) run_items(k);          // help the sweep of diagonal k
      TLMARK(2, k);
      dfar_next = d_req;
      if (hm) fb_last = __builtin_amdgcn_readfirstlane(f_req);
      __syncthreads();
      STAMP(3);
#ifdef DRNA_STAMPS
      if (blockIdx.x == 0 && tid == 0) dbg[256 + k] = st_acc[4];
#endif
    }
  } else {
    // ================= sweep waves: diagonal d = k at step k
    double GE[TSL], GO[TSL];
#pragma unroll
    for (int q = 0; q < TSL; q++) { GE[q] = 0.0; if (TWO_PAR) GO[q] = 0.0; }
    for (int k = TURN + 1; k <= n; k++) {
      TLMARK(0, k);
      if (k < n) {
        const int d = k;
        const int ncell = n - d, sh = d >> 1, par = d & 1;
        const int lo = sh + off0, hi = ncell + sh + off0 - 1;

        // ---- T: tower step
        if (!(DRNA_SKIP & 1) && ((my_pm >> par) & 1) && d >= 10) {
          const int blo = T0 + my_tb * WAVE;                        // the block's slots blo .. blo + 63; the diagonal's cells lo .. hi
          int i = blo + lane + 1 - sh - off0;
          i = i < 1 ? 1 : (i > ncell ? ncell : i);
          const int dv = as_vector(d + 30), i8 = i * 8;
          if (blo <= hi && blo + WAVE - 1 >= lo) {
            double accG = 0.0;
            if constexpr (TWO_PAR) {
              if (par) accG = my_sig ? pf_tower2_step<PfFastSmem<NT>, 1>(sm, GO, dv, i8) : pf_tower2_step<PfFastSmem<NT>, 0>(sm, GO, dv, i8);
            }
            if (!TWO_PAR || !par) accG = my_sig ? pf_tower2_step<PfFastSmem<NT>, 1>(sm, GE, dv, i8) : pf_tower2_step<PfFastSmem<NT>, 0>(sm, GE, dv, i8);
            sm.partG[par][my_sig][T0 + my_tb * WAVE + lane] = accG;
          }
        }
        STAMP(0);
        TLMARK(1, k);
        run_tiles(d);
        TLMARK3(k);
        run_items(d);
        STAMP(6);
        TLMARK(2, k);
      }
      __syncthreads();
      STAMP(3);
#ifdef DRNA_STAMPS
      if (blockIdx.x == 0 && aw == 0 && lane == 0) dbg[512 + k] = st_last;
      if (blockIdx.x == 0 && lane == 0) dbg[1024 + (wave - NB) * 256 + k] = st_acc[3];
#endif
    }
  }

#ifdef DRNA_STAMPS
  if (blockIdx.x == 0 && lane == 0)
    for (int k = 0; k < 8; k++) dbg[wave * 8 + k] = st_acc[k];
#endif
  TLMARK(0, 2);
  // the remaining exterior columns -- their sums by one wave each, side by side (column j reads q5 up to j - 5, which the loop has
  // left final; the same lanes add the same terms in the same order as pf_q5_column), the three-step recurrence by one lane -- then Z
  const int jq0 = max(TURN + 2, n - 2);
  if (NW >= 3) {
    if (wave < 3 && jq0 + wave <= n) {
      const int j = jq0 + wave;
      double sq = 0.0;
      for (int i = lane + 1; i <= j - TURN - 1; i += WAVE) sq += sm.q5[i - 1] * QEXT[j * ld + i];
      sq = wave_total_f64_lane63(sq);
      if (lane == WAVE - 1) sm.xc[13 + wave] = sq;
    }
    __syncthreads();
    if (tid == 0)
      for (int j = jq0; j <= n; j++) sm.q5[j] = sm.q5[j - 1] * sc1 + sm.xc[13 + j - jq0];
    __syncthreads();
  } else if (wave == 0) {
    for (int j = jq0; j <= n; j++) pf_q5_column<NT>(sm, QEXT, ld, j, lane, sc1);
  }
  if (wave == 0) {
    if (lane == 0) {
      const double Z = sm.q5[n];
      if (!(Z > 0.0) || !(Z < 1.0e300)) {
        A.status[r] = ST_PF_RANGE;
        A.Epf[r] = 0.0;
      } else {
        A.status[r] = sm.flag == 2 ? ST_SYNC : ST_OK;             // 2: a wait for the helper workgroup expired (the engine redoes the call without one)
        A.Epf[r] = (-log(Z) - (double)n * log(T.pf_scale)) * T.kT / 1000.0;
      }
      if (hm) st_agent(flagA, A.hbase + STRIP_DONE);
    }
    TLMARK(0, 3);
  }
}

template <int NT, bool TILES = true>
__global__ __launch_bounds__(NT) void pf_lds_kernel(PfArgs A, EvalArgs EV, int R) {      // EV.n_targets > 0 (helper launches only): E(targets) too
  __shared__ PfFastSmem<NT> sm;
  if (!TILES && !A.helper) return;               // (the engine launches this instance with helper workgroups only)
  int bx = -1, is_helper = -1;                   // (no helpers: grid = R or the index list's length, block = sequence)
  if (A.helper) {                                // grid = pair_grid(R): main and helper of a sequence 8 blocks apart (pair_block)
    pair_block(blockIdx.x, bx, is_helper);
    if (bx >= R) return;
  }
  pf_lds_body<NT, TILES>(sm, A, EV, bx, is_helper);
}

}  // namespace drna
