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
//    (Since round 4 the entries of this kernel are indexed by the loop SIZE and a wave owns one parity of the size and one of the
//    diagonal: mfe_tower2_step; the form indexed by the inner diagonal, mfe_tower_step with its per-diagonal table, serves
//    the strip kernels, whose towers travel from strip to strip.)
//    Stack, bulges, 1x1, 2x1, 1xn, 2x2 and 2x3 loops (121 candidates) are still enumerated from the
//    host-built plan with wave-uniform size terms.
//  * f5 is advanced one column per diagonal step by a spare wave, so no serial tail remains.
#pragma once
#include "fold_mfe.hpp"

namespace drna {

constexpr int MFE_FAST_NMAX = 200;
enum : int { E_ALL = 0, E_NEAR = 1, E_FAR = 2, E_COARSE = 3 };   // which shapes an E item / shape table covers (see mfe_e_item)
constexpr int GSLOTS = 10;         // register-resident running minima per lane and parity (28 residues over >= 3 waves)

template <int NT>
struct MfeFastSmem : MfeSmemCore<MFE_FAST_NMAX> {
  static constexpr int NW = NT / WAVE;
  static constexpr int RS = MFE_FAST_NMAX + 2;                                   // ring row pitch
  static constexpr int TRI = (MFE_FAST_NMAX - 4) * (MFE_FAST_NMAX - 3) / 2;      // rows 4..n-1
  static constexpr int NSLOT = 4 * WAVE;                                         // tower slots (n <= 256)
  static constexpr int NL = MFE_FAST_NMAX + 8;
  int fml[TRI + 8];
  int wring[33 * RS];            // (c + TermAU(inner type)) * 256 + info   of the last 32 diagonals; row 32 stays INF
  int ciring[32 * RS];           // c + mismatchI(inner side)               of the last 32 diagonals
  int dml[4 * RS];               // decomposition minima of the last 4 diagonals
  int hpl[MFE_FAST_NMAX + 2];    // hairpin size term by loop size
  int rowoff[MFE_FAST_NMAX + 2]; // offset of row d in fml[]
  int accG[2][NSLOT], accI[2][NSLOT], accK[2][NSLOT];   // per-tower minima, double-buffered by diagonal parity (ds_min)
  // tables with the inner pair's terminal-AU term taken out (it is folded into wring)
  // one array, so that a shape slot can pick its table by offset: stack (XT_STACK), 1x1 (XT_INT11), 1xn mismatch
  // (XT_MM1N), 2x3 mismatch (XT_MM23)
  static constexpr int XT_STACK = 0, XT_INT11 = 64, XT_MM1N = 64 + 1024, XT_MM23 = 64 + 1024 + 128;
  int xtab[64 + 1024 + 128 + 128];
  // per-diagonal tables prepared one step ahead by the finalize waves (double-buffered by diagonal parity)
  int plist[2][NL];              // pairable cells of the diagonal: i | ij << 8 | e(2x2 loop) << 15
  int xe[2][NL];                 // their 1x2 / 2x1 loop energies (two signed halves), inner TermAU taken out
  int pcnt[2];
  int qhead[2];                  // work-queue head of the diagonal's floating items (K sub-blocks, E cell pairs, X groups)
  int eshape[128];               // bulge / 1xn shapes of the shape-uniform E items: s = u1+u2 | u1 << 8 | size term << 16 (static)
  int eshape_rows[128];          // shape slots of the 16-lane-row E items (one-workgroup kernel): s | kind << 5 | u1 << 8 | size term << 16
  int sync_fail;                 // two-workgroup kernel: a wait for the helper expired
  int tbq[4];                    // sector queue of the traceback (TbShared, fold_mfe.hpp)
  int etab[2][128];              // the same shapes as seen from one diagonal: byte offset of the inner pair's ring cell
                                 // for column 0 | size term << 16 (shapes without an inner pair yet point at the INF row)
  int twc[32][2];                // by total loop size s: {asymmetry term of the two new shapes (2, s-2) (s-2, 2), size term (INF below 6)} (mfe_tower2_step)
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

// Diagnostic build only (-DDRNA_STAMPS): per-wave cycle totals of each phase of block 0, written to
// an unused part of its workspace (never read by the kernel).
#ifndef DRNA_SKIP
#define DRNA_SKIP 0          // diagnostic builds only: see fold_pf_lds.hpp
#endif
#ifdef DRNA_STAMPS
#define STAMP(k) do { long long _n = clock64(); st_acc[k] += _n - st_last; st_last = _n; } while (0)
#else
#define STAMP(k) do { } while (0)
#endif
// -DDRNA_TL (tools/timeline.py mfe): raw 100 MHz clocks per wave and step of sequence 0's main workgroup, into
// the unused table 2 of its workspace: 0 after the barrier, 1 after the finalize / tower step, 2 after the last item
#ifdef DRNA_TL
#define MTLMARK(ev, k) do { if (mtl_on && lane == 0) mtl[((wave * 3 + (ev)) << 8) + (k)] = (long long)wall_clock64(); } while (0)
#else
#define MTLMARK(ev, k) do { } while (0)
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

// Per-diagonal table for diagonal d (strip kernels, fold_mfe_strip.hpp; written by a service wave one step ahead, so the tower
// waves spend no scalar instructions on offsets): residue rho of an inner diagonal -> ring offsets and size terms of
// the tower entry that lives there on diagonal d.
template <class SM>
__device__ __forceinline__ void mfe_prepare_etab(SM& sm, int d, int lane, int mode);
template <class SM>
__device__ __forceinline__ void mfe_prepare_tower_tab(SM& sm, int d, int tid, int ninio, int max_ninio) {
  constexpr int RS = SM::RS;
  const int par = d & 1;
  if (tid >= 0 && tid < GRES) {
    // entry of the tower slot whose inner diagonal is congruent to tid (mod 28), as seen from diagonal d:
    //   G <- min(max(G, floor), min(ring[A + i], ring[B + i]) + asym);  candidate = G + size
    // dead / not yet possible entries: asym = INF (G unchanged or INF), size = INF (no candidate)
    const int x = (int)((unsigned)(d + 50 - tid) % (unsigned)GRES);     // (d - 6 - rho) mod 28
    int offA = 0, offB = 0, as = INF_DEV, fl = -INF_DEV, L = INF_DEV;
    if (x <= 26) {
      const int s = x + 4, dp = d - 6 - x;
      if (dp <= TURN) fl = INF_DEV;                                      // entry exists but has no inner pair yet: G = INF
      else {
        const int base = (dp & 31) * RS * 4;
        offA = base + 3 * 4;
        offB = base + (s - 1) * 4;
        as = min(max_ninio, (s - 4) * ninio);
        fl = s <= 5 ? INF_DEV : -INF_DEV;                                // first appearance: forget the previous tenant
        L = sm.tw_L[s];
      }
    }
    int* e = sm.tower_tab[par][tid];
    e[0] = offA; e[1] = offB; e[2] = as; e[3] = fl; e[4] = L; e[5] = 0;
  }
}
template <int NT>
__device__ __forceinline__ void mfe_prepare_tables(MfeFastSmem<NT>& sm, int d, int tid, int ninio, int max_ninio, int emode = E_ALL) {
  (void)ninio; (void)max_ninio;      // (the tower step needs no per-diagonal table any more: mfe_tower2_step)
  if (tid >= 0 && tid < WAVE) mfe_prepare_etab(sm, d, tid, emode);
}

// one diagonal step of the register-resident generic-interior minima of a tower (branch-free, table-driven);
// returns the generic candidate (without the outer mismatch term) for the cell at column i
template <class SM>
__device__ __forceinline__ int mfe_tower_step(const SM& sm, int (&G)[GSLOTS], int par, int i4, int g, int NG, int lane) {
  const char* ring = reinterpret_cast<const char*>(sm.ciring);
  // lane r fetches the table entry of slot r (one LDS round trip for the whole wave); the words are then
  // broadcast with v_readlane as they are needed -- no per-slot LDS read, no scalarised loads
  const int rr = lane * NG + g;
  const bool on = lane < GSLOTS && rr < GRES;
  const int* e = sm.tower_tab[par][on ? rr : 0];
  const int eA = e[0], eB = e[1];
  const int eas = on ? e[2] : INF_DEV, efl = on ? e[3] : -INF_DEV, eL = on ? e[4] : INF_DEV;
  int a[GSLOTS], b[GSLOTS];
#pragma unroll
  for (int r = 0; r < GSLOTS; r++) {
    a[r] = *reinterpret_cast<const int*>(ring + lane_table(eA, r) + i4);
    b[r] = *reinterpret_cast<const int*>(ring + lane_table(eB, r) + i4);
  }
  int acc = INF_DEV;
#pragma unroll
  for (int r = 0; r < GSLOTS; r++) {
    const int v = min(a[r], b[r]) + lane_table(eas, r);
    G[r] = min(max(G[r], lane_table(efl, r)), v);
    acc = min(acc, G[r] + lane_table(eL, r));
  }
  return acc;
}

// ---- tower step with entries indexed by the LOOP SIZE s, not by the inner diagonal (the form fold_pf_lds.hpp has since round 3).
// G_s(i,j) = min over the generic shapes of size s of (inner pair's ring word + asymmetry term) obeys
//     G_s(i,j) = min( G_{s-2}(i+1,j-1), min(X[d'][i+3], X[d'][i+s-1]) + asym[s-4] ),   d' = d - 2 - s,
// and the step of a tower from diagonal d-2 to d moves every minimum from size s-2 to s: in descending order of s that is
// G[q] <- min(G[q-1], ...), a shift.  Per entry everything is a compile-time constant except the ring row (d - 2 - s) & 31: three
// VALU for the address, two ring reads with immediate column offsets, one broadcast read of {asym, size term}, an add, two mins
// and an add (the table-driven form: five v_readlane, two reads, six VALU per entry, and a table a finalize wave rebuilt every
// diagonal).  A wave owns the sizes of ONE parity (s and s-2 share registers) and ONE diagonal parity (a tower has one): it
// works every other step and carries TSL_M = 14 minima; from diagonal 10 on (the first generic shape) three 64-slot blocks cover
// the cells of a 200-nt fold: twelve roles for its twelve sweep waves.  Rows of diagonals that do not exist yet read as INF
// (the ring is INF-initialised and a row's first tenant is written after its last such read).
constexpr int TSL_M = 14;          // entries per tower wave: s = 4 + sigma + 2 q, q = 0 .. 13 (s <= 30)
template <class SM, int SIG, int Q0, int Q1>
__device__ __forceinline__ void mfe_tower2_part(const SM& sm, int (&G)[TSL_M], int dv, int i4, int cbase, int& acc) {
  const char* ring = reinterpret_cast<const char*>(sm.ciring);
  int a[Q1 - Q0], b[Q1 - Q0], cx[Q1 - Q0], cy[Q1 - Q0];
#pragma unroll
  for (int q = Q1 - 1; q >= Q0; q--) {
    const int s = 4 + SIG + 2 * q;
    if (s > 30) continue;
    const int va = ((dv - s) & 31) * (SM::RS * 4) + i4;          // dv = d + 30 in a VGPR: row (d - 2 - s) & 31
    a[q - Q0] = *reinterpret_cast<const int*>(ring + va + 12);
    b[q - Q0] = *reinterpret_cast<const int*>(ring + va + (s - 1) * 4);
    cx[q - Q0] = *reinterpret_cast<const int*>(ring + cbase + s * 8);      // the entry's two constants (same address in every lane)
    cy[q - Q0] = *reinterpret_cast<const int*>(ring + cbase + s * 8 + 4);
  }
#pragma unroll
  for (int q = Q1 - 1; q >= Q0; q--) {
    const int s = 4 + SIG + 2 * q;
    if (s > 30) continue;
    const int v = min(a[q - Q0], b[q - Q0]) + cx[q - Q0];
    G[q] = q > 0 ? min(G[q - 1], v) : v;
    acc = min(acc, G[q] + cy[q - Q0]);
  }
}
template <class SM, int SIG>
__device__ __forceinline__ int mfe_tower2_step(const SM& sm, int (&G)[TSL_M], int dv, int i4) {
  // byte offset of the constants from the ring's start, kept out of the compiler's sight (a literal address costs a v_mov per read)
  const int cbase = as_vector((int)(reinterpret_cast<const char*>(&sm.twc[0][0]) - reinterpret_cast<const char*>(sm.ciring)));
  int acc0 = INF_DEV, acc1 = INF_DEV, acc2 = INF_DEV;
  mfe_tower2_part<SM, SIG, 9, 14>(sm, G, dv, i4, cbase, acc0);     // descending: the upper entries read their predecessors first
  mfe_tower2_part<SM, SIG, 4, 9>(sm, G, dv, i4, cbase, acc1);
  mfe_tower2_part<SM, SIG, 0, 4>(sm, G, dv, i4, cbase, acc2);
  return min(min(acc0, acc1), acc2);
}

// ---- work items shared by the one-workgroup kernel and by the two roles of the two-workgroup kernel (fold_mfe_dual.hpp)

// K: multiloop splits of 32 cells x 4 interleaved split-point groups, tt = tt_lo .. tmax (all of them: TURN+1 .. d-TURN-2;
// the helper of the two-workgroup kernel leaves KEDGE at either end to the main workgroup).  A lane owns two adjacent
// cells (one ds_read2 per operand pair) and walks the compact triangle with running offsets: row tt starts
// rowoff[tt+4] - rowoff[tt] = 4n - 4tt - 6 words after row tt-4... so offset += delta, delta -= 16 per step instead of a
// table lookup per operand.  Minima go to accK by ds_min (order-free).
template <class SM>
__device__ __forceinline__ void mfe_k_item(SM& sm, int it, int d, int n, int ncell, int par, int slot0, int lane, int tt_lo, int tmax) {
  const int INF = INF_DEV, HALF = INF_DEV / 2;
  const int g = lane >> 4, cl = lane & 15;
  int i = (it << 5) + 2 * cl + 1;
  const bool act0 = i <= ncell, act1 = i + 1 <= ncell;
  i = act0 ? i : 1;
  int tt = tt_lo + g;
  const int rr = tt <= tmax ? d - tt - 1 : TURN + 1;
  int offA = fml_off(tt, n) + i - 1, dA = 4 * n - 4 * tt - 6;            // fML[i, i+tt]
  int offC = fml_off(rr, n) + i + tt, dC = -4 * n + 4 * rr - 6;          // fML[i+tt+1, j]
  int m0 = INF, m1 = INF, m2 = INF, m3 = INF;                            // cell i: m0, m2; cell i+1: m1, m3
  for (; tt + 12 <= tmax; tt += 16) {
    const int oA1 = offA + dA, oA2 = oA1 + dA - 16, oA3 = oA2 + dA - 32;
    const int oC1 = offC + dC, oC2 = oC1 + dC - 16, oC3 = oC2 + dC - 32;
    const int a00 = sm.fml[offA], a01 = sm.fml[offA + 1], c00 = sm.fml[offC], c01 = sm.fml[offC + 1];
    const int a10 = sm.fml[oA1], a11 = sm.fml[oA1 + 1], c10 = sm.fml[oC1], c11 = sm.fml[oC1 + 1];
    const int a20 = sm.fml[oA2], a21 = sm.fml[oA2 + 1], c20 = sm.fml[oC2], c21 = sm.fml[oC2 + 1];
    const int a30 = sm.fml[oA3], a31 = sm.fml[oA3 + 1], c30 = sm.fml[oC3], c31 = sm.fml[oC3 + 1];
    offA = oA3 + dA - 48; offC = oC3 + dC - 48; dA -= 64; dC -= 64;
    m0 = min(m0, min(a00 + c00, a20 + c20)); m1 = min(m1, min(a01 + c01, a21 + c21));
    m2 = min(m2, min(a10 + c10, a30 + c30)); m3 = min(m3, min(a11 + c11, a31 + c31));
  }
  for (; tt <= tmax; tt += 4) {
    m0 = min(m0, sm.fml[offA] + sm.fml[offC]);
    m1 = min(m1, sm.fml[offA + 1] + sm.fml[offC + 1]);
    offA += dA; offC += dC; dA -= 16; dC -= 16;
  }
  m0 = min(m0, m2); m1 = min(m1, m3);
  if (act0 && m0 < HALF) atomicMin(&sm.accK[par][i + slot0], m0);
  if (act1 && m1 < HALF) atomicMin(&sm.accK[par][i + 1 + slot0], m1);
}

// The 2 KEDGE split points next to either end of the range, tt = 4 .. 4+KEDGE-1 and d-4-KEDGE .. d-5: the ones that need the
// KEDGE newest finished rows of fML, which the helper workgroup of the two-workgroup kernel does not have yet.  32 cells
// per item, two adjacent cells per lane, the (at most 2 KEDGE) split points dealt to the four 16-lane groups.
template <class SM>
__device__ __forceinline__ void mfe_k_edge_item(SM& sm, int it, int d, int n, int ncell, int par, int slot0, int lane) {
  const int INF = INF_DEV, HALF = INF_DEV / 2;
  const int g = lane >> 4, cl = lane & 15;
  int i = (it << 5) + 2 * cl + 1;
  const bool act0 = i <= ncell, act1 = i + 1 <= ncell;
  i = act0 ? i : 1;
  const int tmax = d - TURN - 2;
  int m0 = INF, m1 = INF;
#pragma unroll
  for (int h = 0; h < (2 * KEDGE + 3) / 4; h++) {
    const int e = g + 4 * h;                                       // 0 .. 2 KEDGE - 1
    const int tt = e < KEDGE ? TURN + 1 + e : tmax - (2 * KEDGE - 1 - e);
    const bool on = e < 2 * KEDGE && tt <= tmax && (e < KEDGE || tt >= TURN + 1 + KEDGE);
    if (on) {
      const int offA = fml_off(tt, n) + i - 1, offC = fml_off(d - tt - 1, n) + i + tt;
      m0 = min(m0, sm.fml[offA] + sm.fml[offC]);
      m1 = min(m1, sm.fml[offA + 1] + sm.fml[offC + 1]);
    }
  }
  if (act0 && m0 < HALF) atomicMin(&sm.accK[par][i + slot0], m0);
  if (act1 && m1 < HALF) atomicMin(&sm.accK[par][i + 1 + slot0], m1);
}

// E: 64 pairable cells per item (lane = compacted cell), one shape CLASS per item: bulges (0,u), bulges (u,0), loops (1,u),
// loops (u,1) in parts of ESH shapes, or the fixed small shapes.  Everything about a shape is wave-uniform (ring row and
// column offset of the inner pair, size term): lane k fetched the diagonal's table word of shape k once and the words are
// broadcast with v_readlane, so a candidate costs one address add, one ring read, a shift, an add and a min -- instead
// of a per-lane shape decode.  MODE: E_ALL = every shape (one-workgroup kernel); E_NEAR / E_FAR = the split of the
// two-workgroup kernel: NEAR are the shapes whose inner pair sits at most four diagonals back ((0,0) (0,1) (1,0) (1,1)
// (0,2) (2,0): one item per block), FAR everything else (its etab marks the two near bulges as padding).
constexpr int ECOARSE = 32;   // at most this many pairable cells on the diagonal: coarse E items
constexpr int ESH = 10, EPB = 13;   // one-workgroup kernel: shapes per E item; E items per block of 64 pairable cells (4 classes x 3 parts + small shapes)
constexpr int NEAR_B = DLAG - 4, NEAR_I = DLAG - 6;   // near shapes per bulge class (u = 2 .. DLAG-3) and per 1xn class (u = 3 .. DLAG-4)
constexpr int EFAR_PARTS = 2;        // items per class of far shapes (helper workgroup): 23 live shapes in 1 x 29 / 2 x 12 / 3 x 8
constexpr int EFAR_NSH = EFAR_PARTS == 1 ? 29 : (33 - DLAG + EFAR_PARTS - 1) / EFAR_PARTS;   // 33 - DLAG far shapes per class
template <int MODE>
__device__ __forceinline__ constexpr int e_items_per_block() { return MODE == E_NEAR ? 2 : MODE == E_FAR ? 4 * EFAR_PARTS : MODE == E_COARSE ? 5 : EPB; }

// candidates of NSH consecutive shapes of one class, starting at table entry `first` (entries that are padding in this
// table read the INF row); bulges: ring word >> 8 + size term; 1xn loops: + the inner pair's 1xn mismatch
template <int NSH, bool ONE_N, class SM>
__device__ __forceinline__ int mfe_e_class(const SM& sm, const char* ring, int par, int first, int lane) {
  const int tw = sm.etab[par][first + (lane & 31)];
  int w[NSH], v = INF_DEV;
#pragma unroll
  for (int k = 0; k < NSH; k++) w[k] = *reinterpret_cast<const int*>(ring + (lane_table(tw, k) & 0xffff));
  if (!ONE_N) {
#pragma unroll
    for (int k = 0; k < NSH; k++) v = min(v, (w[k] >> 8) + (lane_table(tw, k) >> 16));
  } else {
    int tb[NSH];
#pragma unroll
    for (int k = 0; k < NSH; k++) tb[k] = sm.xtab[SM::XT_MM1N + (w[k] & 127)];
#pragma unroll
    for (int k = 0; k < NSH; k++) v = min(v, (w[k] >> 8) + tb[k] + (lane_table(tw, k) >> 16));
  }
  return v;
}

// the nine fixed small shapes: (0,0) | (0,1) (1,0) | (1,1) | (1,2) (2,1) | (2,2) | (2,3) (3,2)
template <class SM>
__device__ __forceinline__ int mfe_e_small(const SM& sm, const char* ring, int d, int par, int qc, int pe, int ij, int e_bulge1,
                                           int e_int23) {
  constexpr int RS = SM::RS;
  const int t8 = (ij >> 4) * 8;
  auto cell = [&](int s_tot, int u1) -> int {          // ring word of the inner pair (uniform row, INF row if none)
    const int dp = d - 2 - s_tot;
    const int off = dp > TURN ? ((dp & 31) * RS + 1 + u1) * 4 : 32 * RS * 4;
    return *reinterpret_cast<const int*>(ring + off);
  };
  const int xv = sm.xe[par][qc];
  const int w00 = cell(0, 0), w01 = cell(1, 0), w10 = cell(1, 1), w11 = cell(2, 1), w12 = cell(3, 1), w21 = cell(3, 2),
            w22 = cell(4, 2), w23 = cell(5, 2), w32 = cell(5, 3);
  const int s00 = sm.xtab[SM::XT_STACK + t8 + ((w00 & 127) >> 4)], s01 = sm.xtab[SM::XT_STACK + t8 + ((w01 & 127) >> 4)],
            s10 = sm.xtab[SM::XT_STACK + t8 + ((w10 & 127) >> 4)],
            s11 = sm.xtab[SM::XT_INT11 + (t8 + ((w11 & 127) >> 4)) * 16 + (ij & 15)],
            s23 = sm.xtab[SM::XT_MM23 + (w23 & 127)], s32 = sm.xtab[SM::XT_MM23 + (w32 & 127)];
  int v = (w00 >> 8) + s00;
  v = min(v, min((w01 >> 8) + s01, (w10 >> 8) + s10) + e_bulge1);
  v = min(v, (w11 >> 8) + s11);
  v = min(v, min((w12 >> 8) + ((xv << 16) >> 16), (w21 >> 8) + (xv >> 16)));
  v = min(v, (w22 >> 8) + (pe >> 15));
  v = min(v, min((w23 >> 8) + s23, (w32 >> 8) + s32) + e_int23 + sm.mm23[ij]);
  return v;
}

// One E item.  E_ALL: item x of the block = part x % 3 (ESH shapes) of class x / 3, x = 12 the small shapes.  E_FAR (helper of
// the two-workgroup kernel): item x = the whole class x, 29 / 27 table entries of which the near ones are padding.  E_NEAR
// (main workgroup): item 0 = the near bulges of both classes, item 1 = the near 1xn loops of both classes and the small shapes.
template <int MODE, class SM>
__device__ __forceinline__ void mfe_e_item(SM& sm, int e, int d, int par, int pcnt, int slot0, int lane, int TermAU, int e_bulge1,
                                           int e_int23) {
  constexpr int IPB = e_items_per_block<MODE>();
  const int INF = INF_DEV, HALF = INF_DEV / 2;
  const int blk = e / IPB, x = e - IPB * blk;
  const int q = blk * WAVE + lane;
  const int qc = q < pcnt ? q : pcnt - 1;
  const int pe = sm.plist[par][qc];
  const int i0 = pe & 255, ij = (pe >> 8) & 127;
  const char* ring = reinterpret_cast<const char*>(sm.wring) + i0 * 4;
  const int outer_b = (ij >> 4) > 2 ? TermAU : 0;
  int v = INF;
  if (MODE == E_ALL) {
    const int cls = x / 3, part = x - 3 * cls;
    if (cls < 2) v = mfe_e_class<ESH, false>(sm, ring, par, cls * 32 + part * ESH, lane) + outer_b;
    else if (cls < 4) v = mfe_e_class<ESH, true>(sm, ring, par, cls * 32 + part * ESH, lane) + sm.mm1n[ij];
    else v = mfe_e_small(sm, ring, d, par, qc, pe, ij, e_bulge1, e_int23);
  } else if (MODE == E_COARSE) {
    // few pairable cells (short sequences, late diagonals): one item per class, its three parts one after the other, so that
    // the per-item costs (queue pop, list and table reads) are paid 5 instead of 13 times
    if (x < 4) {
      const int first = x * 32;
      if (x < 2) v = min(min(mfe_e_class<ESH, false>(sm, ring, par, first, lane), mfe_e_class<ESH, false>(sm, ring, par, first + ESH, lane)),
                         mfe_e_class<ESH, false>(sm, ring, par, first + 2 * ESH, lane)) + outer_b;
      else v = min(min(mfe_e_class<ESH, true>(sm, ring, par, first, lane), mfe_e_class<ESH, true>(sm, ring, par, first + ESH, lane)),
                   mfe_e_class<ESH, true>(sm, ring, par, first + 2 * ESH, lane)) + sm.mm1n[ij];
    } else v = mfe_e_small(sm, ring, d, par, qc, pe, ij, e_bulge1, e_int23);
  } else if (MODE == E_FAR) {
    // the 23 far shapes of a class (bulges u = DLAG-2 .. 30, 1xn loops u = DLAG-3 .. 29) in EFAR_PARTS items: as one item per class
    // (29 table entries, the near ones reading the INF row) an item was a 1.3 - 1.8 us chain, and at late diagonals, where a step has
    // a dozen items for thirteen worker waves, the helper's step was its longest item (tools/timeline.py mfe)
    const int cls = EFAR_PARTS == 1 ? x : x / EFAR_PARTS, part = EFAR_PARTS == 1 ? 0 : x - EFAR_PARTS * cls;
    const int first = cls * 32 + (EFAR_PARTS == 1 ? 0 : (cls < 2 ? NEAR_B : NEAR_I) + part * EFAR_NSH);
    if (cls < 2) v = mfe_e_class<EFAR_NSH, false>(sm, ring, par, first, lane) + outer_b;
    else v = mfe_e_class<EFAR_NSH, true>(sm, ring, par, first, lane) + sm.mm1n[ij];
  } else {
    if (x == 0) v = min(mfe_e_class<NEAR_B, false>(sm, ring, par, 0, lane), mfe_e_class<NEAR_B, false>(sm, ring, par, 32, lane)) + outer_b;
    else v = min(min(mfe_e_class<NEAR_I, true>(sm, ring, par, 64, lane), mfe_e_class<NEAR_I, true>(sm, ring, par, 96, lane)) + sm.mm1n[ij],
                 mfe_e_small(sm, ring, d, par, qc, pe, ij, e_bulge1, e_int23));
  }
  if (q < pcnt && v < HALF) atomicMin(&sm.accI[par][i0 + slot0], v);
}

// bulge / 1xn shapes as seen from diagonal d: where the inner pair of column 0 sits in the ring (bytes), and the size term;
// mode E_NEAR / E_FAR keeps only the shapes of that side of the two-workgroup split
template <class SM>
__device__ __forceinline__ void mfe_prepare_etab(SM& sm, int d, int lane, int mode) {
  constexpr int RS = SM::RS;
  const int par = d & 1;
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int x = lane + h * WAVE;
    const int es = sm.eshape[x];
    const int dp = d - 2 - (es & 255);
    const bool near = (es & 255) <= DLAG - 3;                      // inner pair at most DLAG - 1 diagonals back
    const bool live = dp > TURN && (es >> 16) != 0x3fff &&          // padding entries read the INF row too
                      !(mode == E_FAR && near) && !(mode == E_NEAR && !near);
    const int off = live ? ((dp & 31) * RS + 1 + ((es >> 8) & 255)) * 4 : 32 * RS * 4;
    sm.etab[par][x] = off | (es & 0xffff0000);
  }
}

// static shape table of the E items: shape x of class x >> 5: class 0 bulges (0,u) u = y+2 (y = x & 31 < 29), class 1 bulges
// (u,0), class 2 1xn loops (1,u) u = y+3 (y < 27), class 3 (u,1); the rest is padding
template <class SM>
__device__ __forceinline__ void mfe_init_eshape(SM& sm, const MfeTables& T, int tid, int nthreads) {
  for (int x = tid; x < 128; x += nthreads) {
    const int c = x >> 5, y = x & 31;
    int s_ = 2, u1_ = 0, L_ = 0x3fff;
    if (c < 2 && y < 29) { s_ = y + 2; u1_ = c == 0 ? 0 : s_; L_ = T.bulge[s_]; }
    if (c >= 2 && y < 27) {
      const int u = y + 3;
      s_ = u + 1; u1_ = c == 2 ? 1 : u;
      L_ = T.interior[s_] + min(T.max_ninio, (u - 1) * T.ninio);
    }
    sm.eshape[x] = s_ | (u1_ << 8) | (L_ << 16);
  }
}

// E item of the one-workgroup kernel: four pairable cells per item, one per 16-lane row; a lane folds its eight shape slots in
// registers, the row minimum takes four DPP steps for all four cells at once, lane 15 of each row is the only writer.  Slots
// 3 and 7 of some lanes are the nine small shapes (see eshape_rows): same ring read, but the energy comes from the cell's
// staged values or from another table of xtab.  (The shape-uniform items of mfe_e_item need a third of the instructions but
// more items per diagonal; with sixteen waves sharing one work queue this form is faster up to n = 200: 0.245 vs 0.281 ms at
// n = 100, equal at 200.)
template <class SM>
__device__ __forceinline__ void mfe_e_item_rows(SM& sm, int e, int d, int par, int pcnt, int slot0, int lane, int TermAU,
                                                int e_bulge1, int e_int23, int slot_mask = -1) {
  constexpr int RS = SM::RS;
  const int INF = INF_DEV, HALF = INF_DEV / 2;
  const int it = e, nK = 0;
  // ---- E: four pairable cells per item, one per 16-lane row; a lane folds its eight shape slots in registers, the
  // row minimum takes four DPP steps for all four cells at once, lane 15 of each row is the only writer.
  // Slots 3 and 7 of some lanes are the nine small shapes (see the eshape table): same ring read, but the
  // energy comes from the cell's staged values or from another table of xtab.
  const int q = 4 * (it - nK) + (lane >> 4);
  const int qc = q < pcnt ? q : pcnt - 1;
  const int pe = sm.plist[par][qc], xv = sm.xe[par][qc];
  const int i0 = pe & 255, ij = (pe >> 8) & 127;
  int w[8], e_shape[8];
  bool ok[8];
#pragma unroll
  for (int k = 0; k < 8; k++) e_shape[k] = sm.eshape_rows[k * 16 + (lane & 15)];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int dp = d - 2 - (e_shape[k] & 31);          // diagonal of the inner pair
    ok[k] = dp > TURN;
    w[k] = sm.wring[(dp & 31) * RS + 1 + ((e_shape[k] >> 8) & 31) + i0];
  }
  const int outer_b = (ij >> 4) > 2 ? TermAU : 0, outer_o = sm.mm1n[ij], outer_23 = sm.mm23[ij];
  const int kind3 = (e_shape[3] >> 5) & 7, kind7 = (e_shape[7] >> 5) & 7;
  const int f7 = w[7] & 127, tq = (ij >> 4) * 8 + (f7 >> 4);
  int idx7 = SM::XT_MM1N + f7;
  idx7 = kind7 == 4 ? SM::XT_MM23 + f7 : idx7;
  idx7 = kind7 == 3 ? SM::XT_INT11 + tq * 16 + (ij & 15) : idx7;
  idx7 = (kind7 == 1 || kind7 == 2) ? SM::XT_STACK + tq : idx7;
  int tb[4];                                           // inner-side terms: one more LDS stage
#pragma unroll
  for (int k = 0; k < 3; k++) tb[k] = sm.xtab[SM::XT_MM1N + (w[k + 4] & 127)];
  tb[3] = sm.xtab[idx7];
  int add3 = (e_shape[3] >> 16) + outer_b;
  add3 = kind3 == 5 ? (xv << 16) >> 16 : add3;
  add3 = kind3 == 6 ? xv >> 16 : add3;
  add3 = kind3 == 7 ? pe >> 15 : add3;
  int add7 = (e_shape[7] >> 16) + outer_o;
  add7 = (kind7 == 1 || kind7 == 3) ? 0 : add7;
  add7 = kind7 == 2 ? e_bulge1 : add7;
  add7 = kind7 == 4 ? e_int23 + outer_23 : add7;
  int v = INF;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int e = (w[k] >> 8) + (e_shape[k] >> 16) + outer_b;
    v = min(v, ok[k] ? e : INF);
  }
  v = min(v, ok[3] ? (w[3] >> 8) + add3 : INF);
#pragma unroll
  for (int k = 4; k < 7; k++) {
    const int e = (w[k] >> 8) + (e_shape[k] >> 16) + tb[k - 4] + outer_o;
    v = min(v, ok[k] ? e : INF);
  }
  v = min(v, ok[7] ? (w[7] >> 8) + tb[3] + add7 : INF);
  v = dpp_min_i32<0x111, 0xF>(v);
  v = dpp_min_i32<0x112, 0xF>(v);
  v = dpp_min_i32<0x114, 0xF>(v);
  v = dpp_min_i32<0x118, 0xF>(v);
  if ((lane & 15) == 15 && q < pcnt && v < HALF) atomicMin(&sm.accI[par][(i0 + slot0) & slot_mask], v);
}

// Structure of one diagonal step k (ONE workgroup barrier per diagonal):
//   * waves 0..NB-1 ("finalize waves", one lane per tower slot) turn the minima gathered for diagonal
//     k-1 into c / fML / ring rows, stream c to HBM, advance f5, and prepare the offset tables and the
//     list of pairable cells of diagonal k+1;
//   * waves NB..15 ("sweep waves") gather the candidate minima of diagonal k.  None of that reads
//     anything diagonal k-1 produces (interior loops look at diagonals <= k-2, splits at <= k-5), so the
//     two halves run concurrently.  A sweep wave does, per diagonal:
//       T  tower step of its pinned towers (generic interior loops, 2 LDS reads per live entry);
//       E  bulges and 1xn loops of its share of the PAIRABLE cells, one cell at a time with the 112
//          loop shapes spread over the lanes (two passes of one wave), minima into LDS by ds_min;
//       X  one of the nine fixed small shapes (stack, 1-bulges, 1x1, 2x1, 1x2, 2x2, 2x3, 3x2) for all
//          pairable cells, lane = compacted cell;
//       K  multiloop splits tt = 4 + aw (mod NA) for all cells, lane = cell.
// Uniform bookkeeping comes from LDS tables, not from scalar arithmetic: the scalar unit is shared by
// the 16 waves and was the bottleneck of earlier versions of this kernel.
// ---- compacted list of the pairable cells of diagonal d (one wave): list word i | pair info << 8 | e(2,2) << 15 and the staged
// (1,2) / (2,1) loop energies.  These three small loops depend on the sequence alone, so their table energies (L2) are fetched
// here, once per fill, instead of on the per-diagonal critical path (inner TermAU taken out: it is folded into the ring word).
// A row costs a wave ~0.8 us per 64 cells (the table loads' round trip): 20 us of prologue for 196 rows on sixteen waves.  In
// the two-workgroup kernel the main role builds the rows below PL_D1 only and its HELPER, which has nothing to do until the main
// role's tenth diagonal, the rest (AGENT: stored write-through, read sc1): the main role's prologue is 16 us shorter.
#ifndef DRNA_PL_D1
#define DRNA_PL_D1 16
#endif
constexpr int PL_D1 = DRNA_PL_D1;
constexpr int DUAL_D0 = 2 * TURN + 3;      // first diagonal the helper contributes to (a far shape's inner pair is >= 5 diagonals back)
static_assert(PL_D1 >= DUAL_D0 + 3, "the main role reads a helper-built row only after it has seen the helper's first flag");
template <bool AGENT, class SM>
__device__ __forceinline__ void mfe_pl_row(const SM& sm, const MfeTables& T, int32_t* PL, int32_t* PLX, int ld, int n, int d, int lane,
                                           int TermAU) {
  int base = 0;
  for (int i0 = 1; i0 <= n - d; i0 += WAVE) {
    const int i = i0 + lane;
    int t = 0;
    if (i <= n - d) t = pair_type(sm.Sp[i], sm.Sp[i + d]);
    const unsigned long long m = __ballot(t != 0);
    if (t) {
      const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
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
      const int32_t w = (i | ((t * 16 + si1 * 4 + sj1) << 8)) | (cc << 15), x = (a & 0xffff) | (b << 16);
      if (AGENT) { st_agent(&PL[d * ld + pos], w); st_agent(&PLX[d * ld + pos], x); }
      else { PL[d * ld + pos] = w; PLX[d * ld + pos] = x; }
    }
    base += __popcll(m);
  }
  if (lane == 0) {                              // count kept in the last word of the row
    if (AGENT) st_agent(&PL[d * ld + ld - 1], (int32_t)base);
    else PL[d * ld + ld - 1] = base;
  }
}

template <int NT, bool DUAL = false>
__device__ __forceinline__ void mfe_fill_lds(MfeFastSmem<NT>& sm, const MfeArgs& A, int32_t* __restrict__ Wc, int32_t* __restrict__ EXT,
                             int32_t* __restrict__ PL, int32_t* __restrict__ PLX, DualLink lk = DualLink{}) {
  constexpr int NW = NT / WAVE;
  constexpr int RS = MfeFastSmem<NT>::RS;
  const MfeTables& T = *A.T;
  const int n = A.L, ld = A.ld;
  const int tid = threadIdx.x, lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(wave_id());
  const int INF = INF_DEV, HALF = INF_DEV / 2;
  const int ninio = T.ninio, max_ninio = T.max_ninio, MLbase = T.MLbase, MLclosing = T.MLclosing,
            MLintern = T.MLintern, TermAU = T.TermAU;
  // tower blocks, centred on the sequence
  const int NB = (n + WAVE - 1) / WAVE;
  const int off0 = (NB * WAVE - n) / 2;
  const int NA = NW - NB;                  // sweep waves
  const int aw = wave - NB;                // index among the sweep waves (< 0: finalize wave)
  // Tower roles (mfe_tower2_step).  Generic loops exist from diagonal 10 on, where n - 10 cells are left: NBT = ceil((n - 10) / 64)
  // blocks of 64 tower slots starting at slot T0 cover them for the rest of the fill (the slot range only shrinks).  1024 threads:
  // a block gets four waves, parity of the loop size x parity of the diagonal -- such a wave works every other step; n = 200:
  // three blocks, twelve roles, every sweep wave has one.  Smaller workgroups (CPU emulation, n <= 64): two waves per block, both
  // diagonal parities each.
  constexpr bool TWO_PAR = NT < 1024;
  const int nT = n - 10, NBT = nT > 0 ? (nT + WAVE - 1) / WAVE : 0;
  const int T0 = max(0, off0 + 5 - (NBT * WAVE - nT) / 2);
  int my_tb = -1, my_sig = 0, my_pm = 0;             // block, parity of s, mask of the diagonal parities taken (0: no tower role)
  if (aw >= 0) {
    if (TWO_PAR) { if (aw < 2 * NBT) { my_tb = aw >> 1; my_sig = aw & 1; my_pm = 3; } }
    else if (aw < 4 * NBT) { my_tb = aw >> 2; const int x = (aw + my_tb) & 3; my_sig = x & 1; my_pm = 1 << (x >> 1); }
  }
  // The pairable-list row of the next diagonal is staged into LDS by two SWEEP waves taking turns (block 0's even-size waves: the
  // one whose towers rest in a step requests the row after next, and writes it first thing in the following step, in which it is
  // active).  On a finalize wave the staging was +0.5 us on the cell finalize -- the pole of every step in which the outermost
  // block still holds cells (tools/timeline.py mfe: +2.04 us against +1.27).  Small workgroups (emulation) keep it on a finalize wave.
  const bool sweep_stage = !TWO_PAR && NBT >= 1;
  const bool stager = sweep_stage && my_tb == 0 && my_sig == 0;
  // ... and the exterior column j = k - 3 by block 0's two odd-size waves, the one that rests in the step (its cells, diagonals
  // <= k-4, were stored in step <= k-3): on a finalize wave the column was +0.5 us on the cell finalize of the last block
  const bool sweep_q5 = sweep_stage;
  const bool q5er = sweep_q5 && my_tb == 0 && my_sig == 1;
  // finalize waves that take the side jobs; wave 0 owns the outermost tower block, which has the fewest live cells
  const int w_tab = NB > 1 ? 1 : 0, w_pl = 0, w_q5 = NB - 1;
  // the shape table of the next diagonal goes to a wave of its own when there is one without a side job (n > 128: wave 2): with
  // both tables on wave 1 that wave was the last to reach the barrier in two steps of three (tools/timeline.py: +1.84 us against
  // +1.33 for the cell finalize alone)
  const int w_et = NB > 3 ? 2 : w_tab;

#ifdef DRNA_TL
  long long* ptl2 = reinterpret_cast<long long*>(Wc + 2ll * ld * ld);
  const bool ptl2_on = Wc == A.ws && tid == 0;            // (sequence 0's main workgroup, whatever the block mapping)
#define PTL2(x) do { if (ptl2_on) ptl2[256 + (x)] = (long long)wall_clock64(); } while (0)
#else
#define PTL2(x) do { } while (0)
#endif
  PTL2(0);
  if (DUAL) {
    // round prologue for the helper workgroup, first thing: the pairing codes of this round (masked positions = 4), then the
    // flag -- the helper builds the pairable lists of the diagonals from PL_D1 on while this workgroup fills its tables
    // (the stores travel while this workgroup fills its LDS tables; the flag follows once they have landed)
    int32_t* xs = lk.xs;
    for (int k = tid; k <= n + 1; k += NT) st_agent(xs + k, (int32_t)sm.Sp[k]);
    if (tid == 0) sm.sync_fail = 0;
  }
  PTL2(1);
  // ---- prologue: constant tables, and the compacted list of pairable cells of every diagonal (HBM/L2)
  for (int k = tid; k < 4 * RS; k += NT) sm.dml[k] = INF;
  for (int k = tid; k < 32 * RS; k += NT) { sm.wring[k] = INF * 256; sm.ciring[k] = INF; }   // idle tower entries read row 0
  for (int k = tid; k < RS; k += NT) sm.wring[32 * RS + k] = INF * 256;
  for (int k = tid; k <= n; k += NT) { sm.hpl[k] = A.hp_len[k]; sm.rowoff[k] = k >= TURN + 1 ? fml_off(k, n) : 0; }
  for (int k = tid; k < MfeFastSmem<NT>::NSLOT; k += NT)
    for (int p = 0; p < 2; p++) { sm.accG[p][k] = INF; sm.accI[p][k] = INF; sm.accK[p][k] = INF; }
  mfe_init_eshape(sm, T, tid, NT);
  for (int x = tid; x < 128; x += NT) {
    // E items: a 16-lane row works on one pairable cell; lane l of the row takes the eight shape slots 16 k + l:
    // slots < 64 bulges (x < 29: (0,u) u = x+2; x < 58: (u,0) u = x-27), slots >= 64 1xn loops (y = x-64 < 27: (1,u) u = y+3;
    // y < 54: (u,1) u = y-24).  The spare slots carry the nine small shapes with tables of their own, told apart by a
    // kind field: x = 58..60 the (1,2) (2,1) (2,2) loops (kinds 5..7: energy staged per cell), y = 54..59 stack (1),
    // the two 1-bulges (2), 1x1 (3), (2,3) and (3,2) (4); what is left is padding with an unreachable size term
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
        const int z = y - 54;                 // (0,0) (0,1) (1,0) (1,1) (2,3) (3,2)
        kind_ = z == 0 ? 1 : z <= 2 ? 2 : z == 3 ? 3 : 4;
        s_ = z == 0 ? 0 : z <= 2 ? 1 : z == 3 ? 2 : 5;
        u1_ = z <= 1 ? 0 : z <= 3 ? 1 : z - 2;
        L_ = 0;
      }
    }
    sm.eshape_rows[x] = s_ | (kind_ << 5) | (u1_ << 8) | (L_ << 16);
  }
  for (int k = tid; k < 32; k += NT) {
    sm.twc[k][0] = k >= 4 && k <= 30 ? min(max_ninio, (k - 4) * ninio) : INF;
    sm.twc[k][1] = k >= 6 && k <= 30 ? T.interior[k] : INF;
  }
  using SM = MfeFastSmem<NT>;
  for (int k = tid; k < 64; k += NT) sm.xtab[SM::XT_STACK + k] = sm.stack[k] - ((k & 7) > 2 ? TermAU : 0);
  for (int k = tid; k < 1024; k += NT) sm.xtab[SM::XT_INT11 + k] = sm.int11[k] - (((k >> 4) & 7) > 2 ? TermAU : 0);
  for (int k = tid; k < 128; k += NT) {
    sm.xtab[SM::XT_MM1N + k] = sm.mm1n[k] - ((k >> 4) > 2 ? TermAU : 0);
    sm.xtab[SM::XT_MM23 + k] = sm.mm23[k] - ((k >> 4) > 2 ? TermAU : 0);
  }
  for (int j = tid; j <= n && j <= TURN + 1; j += NT) sm.f5[j] = 0;
  if (DUAL) {
    drain_vmem();
    __syncthreads();
    if (tid == 0) st_agent(lk.flagA, lk.base + TURN);
  }
  PTL2(2);
  // (two-workgroup kernel: the rows from PL_D1 on are the helper's, see mfe_pl_row; this workgroup's own rows stay in its L2)
  for (int d = TURN + 1 + wave; d < (DUAL ? min(n, PL_D1) : n); d += NW) mfe_pl_row<false>(sm, T, PL, PLX, ld, n, d, lane, TermAU);
  __syncthreads();
  PTL2(3);
  // tables and pairable list of the first diagonal
  if (aw < 0) {
    const int d = TURN + 1;
    if (d < n) {
      mfe_prepare_tables<NT>(sm, d, tid, ninio, max_ninio, DUAL ? E_NEAR : E_ALL);
      const int cnt = PL[d * ld + ld - 1];
      if (tid < cnt) { sm.plist[d & 1][tid] = PL[d * ld + tid]; sm.xe[d & 1][tid] = PLX[d * ld + tid]; }
      if (tid == 0) { sm.pcnt[d & 1] = cnt; sm.qhead[0] = 0; sm.qhead[1] = 0; }
    }
  }
  __syncthreads();

#ifdef DRNA_STAMPS
  long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long st_last = clock64();
#endif
#ifdef DRNA_TL
  long long* mtl = reinterpret_cast<long long*>(Wc + 2ll * ld * ld);
  const bool mtl_on = Wc == A.ws;
#endif
  if (aw < 0) {
    // ================= finalize waves
    // two-workgroup kernel: the helper's results for diagonal d (split minima, far-shape minima) are fetched one step AHEAD
    // into registers (pfK, pfI) whenever the flag read during the previous step shows them published, so that in the steady
    // state -- helper a few diagonals ahead -- neither the flag nor the payload latency sits in the step
    int32_t* const xw = reinterpret_cast<int32_t*>(lk.xa);
    int32_t* const xf = xw + (MFE_FAST_NMAX + 2) * XP;
    const int32_t* const xk = reinterpret_cast<const int32_t*>(lk.xb);
    const int32_t* const xi = xk + (MFE_FAST_NMAX + 2) * XP;
    int pfK = INF, pfI = INF, fb_s = 0;
    bool have = false;
    if (DUAL) fb_s = __builtin_amdgcn_readfirstlane(ld_agent(lk.flagB));
    // the staged list row travels in registers from the step that requests it to the next one (wave w_pl)
    int pw[4] = {0, 0, 0, 0}, px[4] = {0, 0, 0, 0}, pw_cnt = 0;
    auto pl_request = [&](const int dn) {
      const int32_t* row = PL + dn * ld;
      const int32_t* rowx = PLX + dn * ld;
      const bool ag = DUAL && dn >= PL_D1;                        // (two-workgroup kernel: the helper's rows)
      pw_cnt = ag ? ld_agent(row + ld - 1) : row[ld - 1];
      const int nch = (n - dn + WAVE - 1) >> 6;                 // chunks the diagonal's cells can fill (uniform, known without the count)
#pragma unroll
      for (int c = 0; c < 4; c++)
        if (c < nch) {
          const int o = min(lane + c * WAVE, ld - 1);
          pw[c] = ag ? ld_agent(row + o) : row[o];
          px[c] = ag ? ld_agent(rowx + o) : rowx[o];
        }
    };
    if (!sweep_stage && wave == w_pl && TURN + 2 < n) pl_request(TURN + 2);
    for (int k = TURN + 1; k <= n; k++) {
      const int d = k - 1;
      MTLMARK(0, k);
      // rows of diagonal d-2: stored in step k-2, and a step's stores have landed by the NEXT step's barrier (the counted wait
      // at the end of a step leaves that step's own stores in flight: no store latency inside a step)
      if (DUAL && tid == 0 && d - 2 > TURN) st_agent(lk.flagA, lk.base + d - 2);
      // two-workgroup kernel: the helper's results for diagonal d+1 and its flag are REQUESTED here, before this step's stores
      // (vector-memory operations retire in order), and taken over at the end of the step
      int rqK = INF, rqI = INF, rqF = 0;
      bool rq_have = false, rq_on = false;
      if (DUAL && d > TURN) {
        const int i2 = tid + 1 - ((d + 1) >> 1) - off0;
        rq_have = d + 1 < n && d + 1 >= DUAL_D0 && flag_ge(fb_s, lk.base + d + 1);
        rq_on = rq_have && i2 >= 1 && i2 <= n - d - 1;
        rqK = ld_agent(rq_on ? xk + (d + 1) * XP + i2 : lk.flagB);
        rqI = ld_agent(rq_on ? xi + (d + 1) * XP + i2 : lk.flagB);
        rqF = ld_agent(lk.flagB);
      }
      // ---- top of the step: the pipelined side jobs.  The list row of diagonal k+1 was requested a whole step ago (the rows were
      // written by the prologue and have left the L2 by now: they come from HBM, and with the request at the top of the SAME step
      // the wave that stages them reached the barrier last in every early step -- tools/timeline.py: +2.46 us against +1.66 for
      // the cell finalize); it goes into LDS here -- the sweep waves read the other parity's buffer -- and the row of diagonal
      // k+2 is requested into the same registers.  The exterior column's cells are requested here and consumed after the cell
      // finalize.
      int fx[4];
      if (!sweep_stage && wave == w_pl) {
        if (k + 1 < n) {
          const int dn = k + 1;
          int* dst = sm.plist[dn & 1];
          int* dxe = sm.xe[dn & 1];
#pragma unroll
          for (int c = 0; c < 4; c++)
            if (lane + c * WAVE < MfeFastSmem<NT>::NL) { dst[lane + c * WAVE] = pw[c]; dxe[lane + c * WAVE] = px[c]; }
          if (lane == 0) { sm.pcnt[dn & 1] = pw_cnt; sm.qhead[dn & 1] = 0; }
        }
        if (k + 2 < n) pl_request(k + 2);
      }
      if (!sweep_q5 && wave == w_q5 && k - 3 >= TURN + 2) {
        const int j = k - 3;
        const int nch = (j - TURN - 1 + WAVE - 1) >> 6;
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const int i = lane + 1 + c * WAVE;
          fx[c] = INF;
          if (c < nch && i <= j - TURN - 1) fx[c] = EXT[j * ld + i];
        }
      }
      if (d > TURN) {
        const int ncell = n - d, sh = d >> 1, par = d & 1;
        const int i = tid + 1 - sh - off0;
        const int dv = as_vector(d), dm1v = as_vector(d - 1);   // uniform LDS indices kept in VGPRs (outside divergent code)
        int bK = pfK, bI = pfI;
        if (DUAL && d >= DUAL_D0) {                              // (before that diagonal the helper has nothing to add)
          if (!have) {                                           // not fetched ahead: wait for the helper here
#ifdef DRNA_TL
            if (mtl_on && tid == 0) { long long* cnt = mtl + 12287; cnt[0] += 1ll << (8 * (k / 25)); }      // (timeline builds: steps that met the helper late)
#endif
            if (!sm.sync_fail && !wait_flag_wave(lk.flagB, lk.base + d)) sm.sync_fail = 1;
            const bool on = i >= 1 && i <= ncell;
            bK = on ? ld_agent(xk + d * XP + i) : INF;
            bI = on ? ld_agent(xi + d * XP + i) : INF;
          }
        }
        if (i >= 1 && i <= ncell) {
          const int aG = sm.accG[par][tid];
          const int aI = DUAL ? min(sm.accI[par][tid], bI) : sm.accI[par][tid];
          const int aK = DUAL ? min(sm.accK[par][tid], bK) : sm.accK[par][tid];
          sm.accG[par][tid] = INF; sm.accI[par][tid] = INF; sm.accK[par][tid] = INF;
          const int j = i + d;
          const int t = pair_type(sm.Sp[i], sm.Sp[j]);
          const int tau = t > 2 ? TermAU : 0;
          int c = INF, info = 0, cb = INF;
          if (t) {
            const int ij = t * 16 + sm.S[i + 1] * 4 + sm.S[j - 1];
            c = mfe_hairpin_e(sm, T, sm.hpl[dm1v], i, j, t);
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
          {
            Wc[d * ld + i] = c * 256 + info;
            EXT[j * ld + i] = c < INF ? c + tau + mfe_extstem(sm, t, i, j, n) : INF;
          }
          int f = INF;
          if (d - 1 > TURN) {
            const int ro1 = sm.rowoff[dm1v];
            const int fa = sm.fml[ro1 + i], fb = sm.fml[ro1 + i - 1];
            if (fa < HALF) f = fa + MLbase;
            if (fb < HALF) f = min(f, fb + MLbase);
          }
          if (c < INF) f = min(f, c + MLintern + tau + sm.mmM[t * 16 + sm.S[i - 1] * 4 + sm.S[j + 1]]);
          const int dec = aK >= HALF ? INF : aK;
          sm.dml[(d & 3) * RS + i] = dec;
          const int fv = min(f, dec);
          sm.fml[sm.rowoff[dv] + i - 1] = fv;
          if (DUAL) {                                            // the helper's copies of the ring word and of fML
            st_agent(xw + d * XP + i, cb * 256 + info);
            st_agent(xf + d * XP + i, fv);
          }
        }
      }
      MTLMARK(1, k);
      // side jobs of the step, one finalize wave each (when there are that many): tower table and pairable list of
      // diagonal k+1 (the sweep waves are reading those of diagonal k), exterior column j = k-3 (its cells, diagonals
      // <= k-4, were stored in step <= k-3 and had landed by the end of step k-2)
      if (k + 1 < n) {
        // once the outermost tower blocks hold no cell any more (n = 200: from diagonal 72 on) their finalize waves, which keep the
        // list staging and the exterior column, are done long before the centre blocks' (+0.9 against +1.6 us): the tables go to them
        const bool outer_idle = NB > 3 && (d >> 1) + off0 >= WAVE;
        const int we = outer_idle ? NB - 1 : w_et;
        if (wave == we) mfe_prepare_etab(sm, k + 1, lane, DUAL ? E_NEAR : E_ALL);
      }
      if (!sweep_q5 && wave == w_q5 && k - 3 >= TURN + 2) {
        const int j = k - 3;
        int m = INF;
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const int i = lane + 1 + c * WAVE;
          if (i <= j - TURN - 1 && fx[c] < HALF) m = min(m, sm.f5[i - 1] + fx[c]);
        }
        m = wave_min_i32_lane63(m);        // DPP (no LDS traffic); lane 63 holds the minimum
        const int prev = sm.f5[j - 1];
        if (lane == WAVE - 1) sm.f5[j] = prev < m ? prev : m;
      }
      STAMP(4);
      MTLMARK(2, k);
      // every global load of the step has been consumed; what is still in flight are this step's stores (c, exterior term,
      // and the two published words of the two-workgroup kernel): the wait lets exactly those stay in flight across the
      // barrier, so the stores of the PREVIOUS step have landed -- which is what their readers rely on (exterior column
      // three steps later, flagA two steps later) -- and no store latency sits in the step
      bool stored = false;
      if (d > TURN) { const int i_ = tid + 1 - (d >> 1) - off0; stored = __ballot(i_ >= 1 && i_ <= n - d) != 0ull; }
      if (DUAL && d > TURN) {
        pfK = rq_on ? rqK : INF; pfI = rq_on ? rqI : INF;
        have = rq_have;
        fb_s = __builtin_amdgcn_readfirstlane(rqF);
      }
      // the requests above are older than the stores, so once they are in, everything of the previous steps has landed; this
      // step's own stores (2, or 4 in the two-workgroup kernel) stay in flight across the barrier
      if (stored) stores_in_flight<DUAL ? 4 : 2>(); else stores_in_flight<0>();
      lds_barrier();                       // one barrier per diagonal
      STAMP(3);
    }
  } else {
    // ================= sweep waves
    int GE[TSL_M], GO[TSL_M];
#pragma unroll
    for (int r = 0; r < TSL_M; r++) { GE[r] = INF; if (TWO_PAR) GO[r] = INF; }
    const int e_bulge1 = keep_i32(T.bulge[1]), e_int23 = keep_i32(T.interior[5] + ninio);
    // Static dealing of the main role's items (two-workgroup kernel).  A tower wave works every other step: the waves whose towers
    // rest in this step come first, among them the waves without a tower role, then the outer blocks' (whose towers die first)
    const int blk_ord = my_tb < 0 ? 0 : my_tb == 0 ? 0 : my_tb == NBT - 1 ? min(1, NBT - 1) : my_tb + 1;      // 0, 2 .. NBT-1, 1 -> outer blocks first
    const int n_free = NA - (TWO_PAR ? 2 : 4) * NBT;          // sweep waves without a tower role
    auto item_rank_of = [&](const int par) -> int {
      if (TWO_PAR) return aw;
      if (my_tb < 0) return aw - 4 * NBT;
      const bool rests = !((my_pm >> par) & 1);
      return n_free + (rests ? 0 : 2 * NBT) + 2 * blk_ord + my_sig;
    };

    int sw[4] = {0, 0, 0, 0}, sx[4] = {0, 0, 0, 0}, sw_cnt = 0;            // the staged list row (stager waves)
    auto st_request = [&](const int dn) {
      const int32_t* row = PL + dn * ld;
      const int32_t* rowx = PLX + dn * ld;
      const bool ag = DUAL && dn >= PL_D1;                        // (two-workgroup kernel: the helper's rows)
      sw_cnt = ag ? ld_agent(row + ld - 1) : row[ld - 1];
      const int nch = (n - dn + WAVE - 1) >> 6;
#pragma unroll
      for (int c = 0; c < 4; c++)
        if (c < nch) {
          const int o = min(lane + c * WAVE, ld - 1);
          sw[c] = ag ? ld_agent(row + o) : row[o];
          sx[c] = ag ? ld_agent(rowx + o) : rowx[o];
        }
    };
    // (the wave that is active in the first step writes the second diagonal's row at its top: requested here)
    if (stager && ((my_pm >> ((TURN + 1) & 1)) & 1) && TURN + 2 < n) st_request(TURN + 2);
    for (int k = TURN + 1; k <= n; k++) {
      MTLMARK(0, k);
      if (q5er && !((my_pm >> (k & 1)) & 1) && k - 3 >= TURN + 2) {
        const int j = k - 3;
        const int nch = (j - TURN - 1 + WAVE - 1) >> 6;
        int fx[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const int i = lane + 1 + c * WAVE;
          fx[c] = INF;
          if (c < nch && i <= j - TURN - 1) fx[c] = EXT[j * ld + i];
        }
        int m = INF;
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const int i = lane + 1 + c * WAVE;
          if (i <= j - TURN - 1 && fx[c] < HALF) m = min(m, sm.f5[i - 1] + fx[c]);
        }
        m = wave_min_i32_lane63(m);        // DPP (no LDS traffic); lane 63 holds the minimum
        const int prev = sm.f5[j - 1];
        if (lane == WAVE - 1) sm.f5[j] = prev < m ? prev : m;
      }
      if (stager && k < n) {
        if ((my_pm >> (k & 1)) & 1) {                              // active in this step: the row requested a step ago goes into LDS
          if (k + 1 < n) {
            const int dn = k + 1;
            int* dst = sm.plist[dn & 1];
            int* dxe = sm.xe[dn & 1];
#pragma unroll
            for (int c = 0; c < 4; c++)
              if (lane + c * WAVE < MfeFastSmem<NT>::NL) { dst[lane + c * WAVE] = sw[c]; dxe[lane + c * WAVE] = sx[c]; }
            if (lane == 0) { sm.pcnt[dn & 1] = sw_cnt; sm.qhead[dn & 1] = 0; }
          }
        } else if (k + 2 < n) st_request(k + 2);                   // resting: request the row after next
      }
      if (k < n) {
        const int d = k;
        const int ncell = n - d, sh = d >> 1, par = d & 1;
        const int lo = sh + off0, hi = ncell + sh + off0 - 1;
        // ---- T: tower step
        if (!(DRNA_SKIP & 1) && ((my_pm >> par) & 1) && d >= 10) {
          const int blo = T0 + my_tb * WAVE;                        // the block's slots blo .. blo + 63; the diagonal's cells lo .. hi
          int i = blo + lane + 1 - sh - off0;
          i = i < 1 ? 1 : (i > ncell ? ncell : i);
          const int dv = as_vector(d + 30), i4 = i * 4;
          if (blo <= hi && blo + WAVE - 1 >= lo) {
            int accG = INF;
            if constexpr (TWO_PAR) {
              if (par) accG = my_sig ? mfe_tower2_step<MfeFastSmem<NT>, 1>(sm, GO, dv, i4) : mfe_tower2_step<MfeFastSmem<NT>, 0>(sm, GO, dv, i4);
            }
            if (!TWO_PAR || !par) accG = my_sig ? mfe_tower2_step<MfeFastSmem<NT>, 1>(sm, GE, dv, i4) : mfe_tower2_step<MfeFastSmem<NT>, 0>(sm, GE, dv, i4);
            atomicMin(&sm.accG[par][blo + lane], accG);
          }
        }
        STAMP(0);
        MTLMARK(1, k);
        const int pcnt = __builtin_amdgcn_readfirstlane(sm.pcnt[par]);
        const int slot0 = sh + off0 - 1;          // tower slot of column i is i + slot0
        // ---- floating items of the diagonal, taken from a work queue (LDS counter) so the sweep waves stay
        // balanced whatever their tower load: first the 16-cell multiloop sub-blocks (K), then pairs of
        // pairable cells for the 112 bulge / 1xn shapes and the nine small fixed shapes (E).  Minima are order-free,
        // so who takes what does not matter.
        // one-workgroup kernel: a 32-cell block's split points go to 1, 2 or 4 items as the cells get fewer and the sums longer (a late
        // diagonal has two or three blocks and 190 split points: as one item per block three waves walked ~1.5 us chains while nine
        // idled; minima are order-free, so the split does not show in the results: 0.566 -> 0.555 ms at R = 64, 0.579 -> 0.565 at R = 128).
        // In the helper of the two-workgroup kernel the same split gains nothing and finer ones lose (0.477 -> 0.487 / 0.500 ms).
        // Two-workgroup kernel, main role: the edge split points only, one item per block
        const int kssh = DUAL ? 0 : ncell > 128 ? 0 : ncell > 64 ? 1 : 2;
        const int k_per = (((d - 2 * TURN - 2 + (1 << kssh) - 1) >> kssh) + 3) & ~3;      // split points per item (TURN+1 .. d-TURN-2)
        const int nK = (DRNA_SKIP & 8) ? 0 : ((ncell + 31) >> 5) << kssh,
                  nE = (DRNA_SKIP & 2) ? 0 : DUAL ? e_items_per_block<E_NEAR>() * ((pcnt + WAVE - 1) >> 6) : (pcnt + 3) >> 2;
        const int nItems = __builtin_amdgcn_readfirstlane(nK + nE);
        // one-workgroup kernel: items from the work queue (LDS counter).  Main role of the two-workgroup kernel: the few
        // items left (edge split points, near shapes) are dealt statically, waves of the outer tower blocks -- whose towers
        // die first -- before those of the centre blocks: no queue pops (a pop is an LDS atomic round trip of ~400 cycles)
        int it = DUAL ? item_rank_of(par) : queue_pop(&sm.qhead[par], lane);
        for (; it < nItems; it = DUAL ? it + NA : queue_pop(&sm.qhead[par], lane)) {
          STAMP(6);
          if (it < nK) {
            if (DUAL) mfe_k_edge_item(sm, it, d, n, ncell, par, slot0, lane);
            else {
              const int lo = TURN + 1 + (it & ((1 << kssh) - 1)) * k_per;
              mfe_k_item(sm, it >> kssh, d, n, ncell, par, slot0, lane, lo, min(d - TURN - 2, lo + k_per - 1));
            }
            STAMP(5);
#ifdef DRNA_STAMPS
            st_acc[7]++;
#endif
          } else {
            if (DUAL) mfe_e_item<E_NEAR>(sm, it - nK, d, par, pcnt, slot0, lane, TermAU, e_bulge1, e_int23);
            else mfe_e_item_rows(sm, it - nK, d, par, pcnt, slot0, lane, TermAU, e_bulge1, e_int23);
            STAMP(1);
#ifdef DRNA_STAMPS
            st_acc[2]++;
#endif
          }
        }
        STAMP(6);
        MTLMARK(2, k);
      }
      __syncthreads();
      STAMP(3);
    }
  }
#ifdef DRNA_STAMPS
  if (blockIdx.x == 0 && lane == 0) {
    long long* dbg = reinterpret_cast<long long*>(Wc + 2ll * ld * ld);
    for (int k = 0; k < 8; k++) dbg[wave * 8 + k] = st_acc[k];
  }
#endif
  // the remaining exterior columns (every store has landed: the loop ended with a draining barrier): their minima by one wave
  // each, side by side (column j reads f5 up to j - 5, which the loop has left final), then the three-step recurrence by one lane
  {
    const int j0 = max(TURN + 2, n - 2);
    if (NW >= 3) {
      if (wave < 3 && j0 + wave <= n) {
        const int j = j0 + wave;
        int m = INF_DEV;
        for (int i = lane + 1; i <= j - TURN - 1; i += WAVE) {
          const int x = EXT[j * ld + i];
          if (x < INF_DEV / 2) m = min(m, sm.f5[i - 1] + x);
        }
        m = wave_min_i32(m);
        if (lane == 0) sm.tbq[wave] = m;                 // (the traceback's queue words: not in use yet)
      }
      __syncthreads();
      if (tid == 0)
        for (int j = j0; j <= n; j++) { const int prev = sm.f5[j - 1], m = sm.tbq[j - j0]; sm.f5[j] = prev < m ? prev : m; }
    } else if (wave == 0) {
      for (int j = j0; j <= n; j++) mfe_f5_column<NT>(sm, EXT, ld, j, lane);
    }
  }
  __syncthreads();
}

// one sequence, all pseudoknot rounds: prologue, fill, traceback.  DUAL: the main role of the two-workgroup kernel (lk links
// it to its helper); every exit publishes DONE so that the helper leaves too
template <int NT, bool DUAL>
__device__ __forceinline__ void mfe_lds_body(MfeFastSmem<NT>& sm, MfeArgs A, int r, DualLink lk) {
  if (A.rg.len) A.L = A.rg.len[r];
  const long long so = A.rg.off ? (long long)A.rg.off[r] : (long long)r * A.L;      // offset in seqs / ss
  const int n = A.L, ld = A.ld, tid = threadIdx.x;
  const MfeTables& T = *A.T;
  int32_t* base = A.ws + (long long)r * A.ws_stride;
  const long long tab = (long long)ld * ld;
  int32_t* Wc = base;
  int32_t* EXT = base + 4 * tab;
  int32_t* PL = base + 3 * tab;      // compacted pairable-cell lists, one row per diagonal
  int32_t* PLX = base + 1 * tab;     // their staged (1,2) / (2,1) loop energies

#ifdef DRNA_TL
  // phase marks of sequence 0's (main) workgroup (tools/timeline.py mfe): kernel entry, fill done, traceback done (steps start at
  // TURN + 1: slots 0 .. 3 of every event row are free); PTL2 in mfe_fill_lds: the parts of the first round's prologue
  long long* ptl = reinterpret_cast<long long*>(Wc + 2ll * ld * ld);
  const bool ptl_on = Wc == A.ws && tid == 0;
  if (ptl_on) { ptl[0] = (long long)wall_clock64(); ptl[12287] = 0; }
#endif
  for (int k = tid; k < 64; k += NT) sm.stack[k] = T.stack[k];
  for (int k = tid; k < 128; k += NT) {
    sm.mmH[k] = T.mmH[k]; sm.mmI[k] = T.mmI[k]; sm.mm1n[k] = T.mm1n[k];
    sm.mm23[k] = T.mm23[k]; sm.mmM[k] = T.mmM[k]; sm.mmExt[k] = T.mmExt[k];
  }
  for (int k = tid; k < 1024; k += NT) sm.int11[k] = T.int11[k];
  for (int k = tid; k < 32; k += NT) { sm.d5[k] = T.d5[k]; sm.d3[k] = T.d3[k]; }
  if (tid == 0) { sm.flag = 0; sm.sync_fail = 0; }
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
    if (tid == 0) { A.status[r] = ST_BAD_CHAR; A.Emfe[r] = 0; if (DUAL) st_agent(lk.flagA, dual_done(lk.epoch)); }
    for (int k = tid; k < n; k += NT) A.ss[so + k] = '.';
    return;
  }

  int status = ST_OK;
  for (int round = 0; round <= A.pk_rounds; round++) {
    for (int k = tid; k < n; k += NT) sm.ssw[k] = '.';
    lk.base = dual_base(lk.epoch, round);
    mfe_fill_lds<NT, DUAL>(sm, A, Wc, EXT, PL, PLX, lk);           // ends with a barrier
#ifdef DRNA_TL
    if (ptl_on && round == 0) ptl[1] = (long long)wall_clock64();
#endif
    if (DUAL && sm.sync_fail) { status = ST_SYNC; break; }
    // traceback by TB_WAVES waves working from one queue of sectors in LDS (TbShared): entry 0 = the whole exterior interval
    constexpr int TB_WAVES = NT / WAVE < 8 ? NT / WAVE : 8;
    for (int k = tid; k < (int)(sizeof(sm.sec_ml) / sizeof(sm.sec_ml[0])); k += NT) sm.sec_ml[k] = 0;
    __syncthreads();
    if (tid == 0) {
      sm.sec_i[0] = 1; sm.sec_j[0] = (short)n; sm.sec_ml[0] = 1;          // ml 0, published
      sm.tbq[0] = 0; sm.tbq[1] = 1; sm.tbq[2] = 1; sm.tbq[3] = 0;
    }
    __syncthreads();
#ifdef DRNA_TL_TB
    if (blockIdx.x == 0 && tid == 0) drna_tl_tb_ptr = round == 0 ? ptl : nullptr;
    __syncthreads();
#endif
    if (!(DRNA_SKIP & 256) && wave_id() < TB_WAVES) (void)mfe_traceback_q(sm, A, Wc, FmlLds<NT>{&sm, n}, EXT, TbShared<MfeFastSmem<NT>>{sm});
    __syncthreads();
#ifdef DRNA_TL
    if (ptl_on && round == 0) ptl[2] = (long long)wall_clock64();
#endif
    if (tid == 0) {
      if (round == 0) A.Emfe[r] = sm.f5[n];
      sm.flag = (!(DRNA_SKIP & 256) && sm.tbq[3] == 2) ? 1 : 0;
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
  if (DUAL && tid == 0) st_agent(lk.flagA, dual_done(lk.epoch));
  for (int k = tid; k < n; k += NT) A.ss[so + k] = sm.sspk[k];
  if (tid == 0) A.status[r] = status;
}

template <int NT>
__global__ __launch_bounds__(NT) void mfe_lds_kernel(MfeArgs A) {
  __shared__ MfeFastSmem<NT> sm;
  const int r = A.rg.idx ? A.rg.idx[blockIdx.x] : blockIdx.x;
  mfe_lds_body<NT, false>(sm, A, r, DualLink{});
}

}  // namespace drna
