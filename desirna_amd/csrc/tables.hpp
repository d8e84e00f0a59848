// tables.hpp -- host-side expansion of the parameter blob (desirna_amd/params.py) into the compact
// lookup tables the gfx950 kernels index, plus the interior-loop "plan".
//
// Replaces what RNA.params_load + RNA.fold_compound do on the host in the reference
// (DesiRNA.py:455-456, utils/energy_scores.py:147): ViennaRNA copies its parameter set into every
// fold compound; here one immutable table block is built once per engine and shared by all launches.
//
// Index conventions on the device (all tables are flat arrays):
//   nucleotide code  A=0 C=1 G=2 U=3            (2 bits)
//   pair type        CG=1 GC=2 GU=3 UG=4 AU=5 UA=6, 0 = cannot pair, 7 = non-standard (eval only)
//   mm*[t*16 + a*4 + b]           mismatch tables, a = first neighbour arg, b = second
//   stack[t*8 + t2]               t2 = rtype of the inner pair
//   int11[(t*8+t2)*16 + a*4+b], int21[(t*8+t2)*64 + a*16+b*4+c], int22[(t*8+t2)*256 + a*64+b*16+c*4+d]
//   "info" byte of a pair (p,q) seen as the INNER pair of a loop: (rtype(p,q) << 4) | (S[q+1] << 2) | S[p-1]
//   so that mmX[info] is exactly the inner-side mismatch term of ViennaRNA's E_IntLoop.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace drna {

constexpr int INF_REF = 10000000;   // ViennaRNA's INF (what the ABI reports)
constexpr int INF_DEV = 1 << 22;    // device-side "infinity": (INF_DEV << 8) still fits an int32
constexpr int TURN = 3;
constexpr int MAXLOOP = 30;
constexpr int NPLAN = 496;          // #(u1,u2) with u1+u2 <= 30
constexpr int NPAIR_MAX = 256;      // slots of consecutive generic entries (200 pairs / 109 quads with MAXLOOP = 30)
constexpr int MAX_SPECIAL = 64;

// plan kinds (wave-uniform switch in the fill kernels)
enum PlanKind : int {
  PK_STACK = 0, PK_BULGE1 = 1, PK_BULGEN = 2, PK_INT11 = 3, PK_INT21 = 4, PK_INT12 = 5,
  PK_1XN = 6, PK_INT22 = 7, PK_INT23 = 8, PK_GENERIC = 9, PK_NKINDS = 10
};

struct Plan {
  int u1[NPLAN], u2[NPLAN], kind[NPLAN];
  int L[NPLAN];          // MFE: size-dependent integer term
  double W[NPLAN];       // PF : size-dependent Boltzmann factor INCLUDING scale[u1+u2+2]
  int seg[PK_NKINDS + 1];  // entries of kind k are [seg[k], seg[k+1])
  // canonical traceback order (p ascending, q descending): SURVEY App. A.4
  int tb_u1[NPLAN], tb_u2[NPLAN];
  int tb_shape[NPLAN], tb_L[NPLAN];   // ... packed for the traceback's scan: u1 | u2 << 8 | kind << 16, and the candidate's size term
  // The generic entries are ordered by (u1+u2, u1): entries of one loop size read CONSECUTIVE cells of one table row, so a
  // lane can fetch them 16 bytes at a time.  Slots of two (fp64 tables) and of four (int32 tables) consecutive entries:
  // first entry and number of entries (a slot never crosses a loop size).
  int n_pair, n_quad;
  int pair_e[NPAIR_MAX], pair_n[NPAIR_MAX];
  int quad_e[NPAIR_MAX], quad_n[NPAIR_MAX];
};

struct MfeTables {
  int stack[64];
  int mmH[128], mmI[128], mm1n[128], mm23[128], mmM[128], mmExt[128];
  int d5[32], d3[32];
  int int11[1024];
  int int21[4096];
  int int22[16384];
  int bulge[31], interior[31];
  int ninio, max_ninio, MLbase, MLclosing, MLintern, TermAU;
  int n_tri, n_tetra, n_hexa;
  int tri_code[MAX_SPECIAL], tri_e[MAX_SPECIAL];
  int tetra_code[MAX_SPECIAL], tetra_e[MAX_SPECIAL];
  int hexa_code[MAX_SPECIAL], hexa_e[MAX_SPECIAL];
};

struct PfTables {
  double stack[64];
  double mmH[128], mmI[128], mm1n[128], mm23[128], mmM[128], mmExt[128];
  double d5[32], d3[32];
  double int11[1024];
  double int21[4096];
  double int22[16384];
  double bulge[31], interior[31], eninio[31];
  double MLbase, MLclosing, MLintern, TermAU;   // Boltzmann factors
  double kT, pf_scale;
  int n_tri, n_tetra, n_hexa;
  int tri_code[MAX_SPECIAL], tetra_code[MAX_SPECIAL], hexa_code[MAX_SPECIAL];
  double tri_w[MAX_SPECIAL], tetra_w[MAX_SPECIAL], hexa_w[MAX_SPECIAL];
};

struct HostTables {
  MfeTables mfe;
  PfTables pf;
  Plan plan;
  int DuplexInit = 0;
  double lxc = 0;
  int hairpin[31];
  double ehairpin[31];
  // length-indexed terms, filled by size_tables(max_L)
  std::vector<int> hp_len;        // hairpin energy by loop size u (log extrapolation beyond 30)
  std::vector<double> hp_w;       // PF: hairpin Boltzmann factor by size, times scale[u+2]
  std::vector<double> scale;      // pf_scale^-k
  std::vector<double> eMLb;       // (expMLbase/pf_scale)^k
  std::vector<int> bulge_len, int_len;  // eval_structure: bulge / interior size terms beyond MAXLOOP
};

namespace detail {
inline int nt_code(char c) {
  switch (c) {
    case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'U': return 3;
    default: return -1;
  }
}
inline double smooth(double x) {  // ViennaRNA SMOOTH(), pf_smooth = 1
  const double SC = 10.0;
  if (x / SC < -1.2283697) return 0.0;
  if (x / SC > 0.8660254) return x;
  double s = std::sin(x / SC - 0.34242663) + 1.0;
  return SC * 0.38490018 * s * s;
}
inline int clampdev(int e) { return e >= INF_REF / 2 ? INF_DEV : e; }
}  // namespace detail

// returns empty string on success, else an error message
inline std::string build_tables(const int32_t* b, int n_int32, HostTables& T) {
  using namespace detail;
  if (n_int32 < 3 || b[0] != 0x504E5244 || b[1] != 1 || b[2] != n_int32) return "bad parameter blob header";
  const int32_t* p = b + 3;
  const int32_t* stack = p; p += 64;
  const int32_t* mm[6];
  for (int k = 0; k < 6; k++) { mm[k] = p; p += 200; }
  const int32_t* d5 = p; p += 40;
  const int32_t* d3 = p; p += 40;
  const int32_t* i11 = p; p += 1600;
  const int32_t* i21 = p; p += 8000;
  const int32_t* i22 = p; p += 40000;
  const int32_t* hairpin = p; p += 31;
  const int32_t* bulge = p; p += 31;
  const int32_t* interior = p; p += 31;
  int ninio = *p++, max_ninio = *p++, MLbase = *p++, MLclosing = *p++, MLintern = *p++;
  int DuplexInit = *p++, TermAU = *p++;
  double lxc; std::memcpy(&lxc, p, 8); p += 2;
  int n_tri = *p++, n_tetra = *p++, n_hexa = *p++;
  if (n_tri > MAX_SPECIAL || n_tetra > MAX_SPECIAL || n_hexa > MAX_SPECIAL) return "too many special loops";
  if ((p - b) + 3 * (n_tri + n_tetra + n_hexa) != n_int32) return "parameter blob length mismatch";

  const double kT = (37.0 + 273.15) * 1.98717;  // cal/mol
  auto BF = [&](double e) { return std::exp(-e * 10.0 / kT); };
  MfeTables& M = T.mfe;
  PfTables& F = T.pf;
  std::memset(&M, 0, sizeof(M));
  std::memset(&F, 0, sizeof(F));
  T.DuplexInit = DuplexInit;
  T.lxc = lxc;
  F.kT = kT;
  F.pf_scale = std::exp(1.07 * 185.0 / kT);
  if (F.pf_scale < 1.0) F.pf_scale = 1.0;

  for (int t = 0; t < 8; t++)
    for (int t2 = 0; t2 < 8; t2++) {
      M.stack[t * 8 + t2] = clampdev(stack[t * 8 + t2]);
      F.stack[t * 8 + t2] = BF(stack[t * 8 + t2]);
    }
  int* mdst[6] = {M.mmH, M.mmI, M.mm1n, M.mm23, M.mmM, M.mmExt};
  double* fdst[6] = {F.mmH, F.mmI, F.mm1n, F.mm23, F.mmM, F.mmExt};
  for (int k = 0; k < 6; k++)
    for (int t = 0; t < 8; t++)
      for (int a = 0; a < 4; a++)
        for (int c = 0; c < 4; c++) {
          int raw = mm[k][t * 25 + (a + 1) * 5 + (c + 1)];
          bool dangle_like = (k >= 4);  // multi / exterior: clamp for MFE, smooth for PF (App. A.2/A.5)
          mdst[k][t * 16 + a * 4 + c] = dangle_like ? std::min(0, raw) : clampdev(raw);
          fdst[k][t * 16 + a * 4 + c] = dangle_like ? std::exp(smooth(-(double)raw) * 10.0 / kT) : BF(raw);
        }
  for (int t = 0; t < 8; t++)
    for (int a = 0; a < 4; a++) {
      int r5 = d5[t * 5 + a + 1], r3 = d3[t * 5 + a + 1];
      M.d5[t * 4 + a] = std::min(0, r5);
      M.d3[t * 4 + a] = std::min(0, r3);
      F.d5[t * 4 + a] = std::exp(smooth(-(double)r5) * 10.0 / kT);
      F.d3[t * 4 + a] = std::exp(smooth(-(double)r3) * 10.0 / kT);
    }
  for (int t = 0; t < 8; t++)
    for (int t2 = 0; t2 < 8; t2++) {
      for (int a = 0; a < 4; a++)
        for (int c = 0; c < 4; c++) {
          int v = i11[((t * 8 + t2) * 5 + a + 1) * 5 + c + 1];
          M.int11[(t * 8 + t2) * 16 + a * 4 + c] = clampdev(v);
          F.int11[(t * 8 + t2) * 16 + a * 4 + c] = BF(v);
          for (int d = 0; d < 4; d++) {
            int v21 = i21[(((t * 8 + t2) * 5 + a + 1) * 5 + c + 1) * 5 + d + 1];
            M.int21[(t * 8 + t2) * 64 + a * 16 + c * 4 + d] = clampdev(v21);
            F.int21[(t * 8 + t2) * 64 + a * 16 + c * 4 + d] = BF(v21);
            for (int e = 0; e < 4; e++) {
              int v22 = i22[((((t * 8 + t2) * 5 + a + 1) * 5 + c + 1) * 5 + d + 1) * 5 + e + 1];
              M.int22[(t * 8 + t2) * 256 + a * 64 + c * 16 + d * 4 + e] = clampdev(v22);
              F.int22[(t * 8 + t2) * 256 + a * 64 + c * 16 + d * 4 + e] = BF(v22);
            }
          }
        }
    }
  for (int k = 0; k <= 30; k++) {
    T.hairpin[k] = clampdev(hairpin[k]);
    T.ehairpin[k] = BF(hairpin[k]);
    M.bulge[k] = clampdev(bulge[k]);
    M.interior[k] = clampdev(interior[k]);
    F.bulge[k] = BF(bulge[k]);
    F.interior[k] = BF(interior[k]);
    F.eninio[k] = BF(std::min(max_ninio, k * ninio));
  }
  M.ninio = ninio; M.max_ninio = max_ninio; M.MLbase = MLbase; M.MLclosing = MLclosing;
  M.MLintern = MLintern; M.TermAU = TermAU;
  F.MLbase = BF(MLbase); F.MLclosing = BF(MLclosing); F.MLintern = BF(MLintern); F.TermAU = BF(TermAU);

  auto take_special = [&](int cnt, int len, int* codes, int* es, double* ws, int& n_out) {
    n_out = 0;
    for (int k = 0; k < cnt; k++, p += 3) {
      char s[9] = {0};
      std::memcpy(s, p, 8);
      int code = 0;
      bool ok = true;
      for (int c = 0; c < len; c++) {
        int nc = nt_code(s[c]);
        if (nc < 0) { ok = false; break; }
        code |= nc << (2 * c);
      }
      if (!ok) continue;
      codes[n_out] = code; es[n_out] = p[2]; ws[n_out] = BF(p[2]);
      n_out++;
    }
  };
  int tri_code2[MAX_SPECIAL], tetra_code2[MAX_SPECIAL], hexa_code2[MAX_SPECIAL];
  (void)tri_code2; (void)tetra_code2; (void)hexa_code2;
  take_special(n_tri, 5, M.tri_code, M.tri_e, F.tri_w, M.n_tri);
  take_special(n_tetra, 6, M.tetra_code, M.tetra_e, F.tetra_w, M.n_tetra);
  take_special(n_hexa, 8, M.hexa_code, M.hexa_e, F.hexa_w, M.n_hexa);
  F.n_tri = M.n_tri; F.n_tetra = M.n_tetra; F.n_hexa = M.n_hexa;
  std::memcpy(F.tri_code, M.tri_code, sizeof(M.tri_code));
  std::memcpy(F.tetra_code, M.tetra_code, sizeof(M.tetra_code));
  std::memcpy(F.hexa_code, M.hexa_code, sizeof(M.hexa_code));

  // ---- interior-loop plan: every (u1,u2), u1+u2 <= MAXLOOP, classified as ViennaRNA's E_IntLoop does
  Plan& P = T.plan;
  std::vector<int> order[PK_NKINDS];
  auto kind_of = [](int u1, int u2) {
    int nl = std::max(u1, u2), ns = std::min(u1, u2);
    if (nl == 0) return (int)PK_STACK;
    if (ns == 0) return nl == 1 ? (int)PK_BULGE1 : (int)PK_BULGEN;
    if (ns == 1) {
      if (nl == 1) return (int)PK_INT11;
      if (nl == 2) return u1 == 1 ? (int)PK_INT21 : (int)PK_INT12;
      return (int)PK_1XN;
    }
    if (ns == 2 && nl == 2) return (int)PK_INT22;
    if (ns == 2 && nl == 3) return (int)PK_INT23;
    return (int)PK_GENERIC;
  };
  for (int u1 = 0; u1 <= MAXLOOP; u1++)
    for (int u2 = 0; u1 + u2 <= MAXLOOP; u2++)
      if (kind_of(u1, u2) != PK_GENERIC) order[kind_of(u1, u2)].push_back(u1 * 64 + u2);
  for (int sz = 0; sz <= MAXLOOP; sz++)           // generic entries by (size, u1)
    for (int u1 = 0; u1 <= sz; u1++)
      if (kind_of(u1, sz - u1) == PK_GENERIC) order[PK_GENERIC].push_back(u1 * 64 + (sz - u1));
  int e = 0;
  for (int k = 0; k < PK_NKINDS; k++) {
    P.seg[k] = e;
    for (int code : order[k]) {
      int u1 = code / 64, u2 = code % 64, nl = std::max(u1, u2), ns = std::min(u1, u2), s = u1 + u2;
      P.u1[e] = u1; P.u2[e] = u2; P.kind[e] = k;
      int L = 0; double W = 1.0;
      switch (k) {
        case PK_BULGE1: case PK_BULGEN: L = bulge[nl]; W = BF(bulge[nl]); break;
        case PK_1XN:
          L = interior[nl + 1] + std::min(max_ninio, (nl - ns) * ninio);
          W = BF(interior[nl + 1]) * BF(std::min(max_ninio, (nl - ns) * ninio)); break;
        case PK_INT23:
          L = interior[5] + ninio; W = BF(interior[5]) * BF(std::min(max_ninio, ninio)); break;
        case PK_GENERIC:
          L = interior[s] + std::min(max_ninio, (nl - ns) * ninio);
          W = BF(interior[s]) * BF(std::min(max_ninio, (nl - ns) * ninio)); break;
        default: break;
      }
      P.L[e] = L;
      P.W[e] = W * std::pow(F.pf_scale, -(double)(s + 2));
      e++;
    }
  }
  P.seg[PK_NKINDS] = e;
  P.n_pair = P.n_quad = 0;
  for (int g = P.seg[PK_GENERIC]; g < NPLAN;) {
    int run = 1;                                 // entries of the same size with consecutive u1
    while (g + run < NPLAN && P.u1[g + run] + P.u2[g + run] == P.u1[g] + P.u2[g] && P.u1[g + run] == P.u1[g] + run) run++;
    for (int x = 0; x < run; x += 2) { P.pair_e[P.n_pair] = g + x; P.pair_n[P.n_pair] = std::min(2, run - x); P.n_pair++; }
    for (int x = 0; x < run; x += 4) { P.quad_e[P.n_quad] = g + x; P.quad_n[P.n_quad] = std::min(4, run - x); P.n_quad++; }
    g += run;
  }
  e = 0;
  for (int u1 = 0; u1 <= MAXLOOP; u1++)       // p ascending
    for (int u2 = 0; u1 + u2 <= MAXLOOP; u2++) {  // q descending
      const int k = kind_of(u1, u2), nl = std::max(u1, u2), ns = std::min(u1, u2);
      int L = 0;
      if (k == PK_BULGE1 || k == PK_BULGEN) L = M.bulge[nl];           // (the device tables' values: what the fill added)
      else if (k == PK_1XN) L = M.interior[nl + 1] + std::min(M.max_ninio, (nl - ns) * M.ninio);
      else if (k == PK_INT23) L = M.interior[5] + M.ninio;
      else if (k == PK_GENERIC) L = M.interior[nl + ns] + std::min(M.max_ninio, (nl - ns) * M.ninio);
      P.tb_u1[e] = u1; P.tb_u2[e] = u2; P.tb_shape[e] = u1 | (u2 << 8) | (k << 16); P.tb_L[e] = L; e++;
    }
  return "";
}

inline void size_tables(HostTables& T, int max_L) {
  const double kT = T.pf.kT;
  // the loop-size weights scale[u1+u2+2] are tabulated for every loop size up to MAXLOOP whatever the sequence length
  int N = (max_L > MAXLOOP + 4 ? max_L : MAXLOOP + 4) + 4;
  T.hp_len.assign(N, INF_DEV);
  T.hp_w.assign(N, 0.0);
  T.scale.assign(N, 1.0);
  T.eMLb.assign(N, 1.0);
  T.bulge_len.assign(N, INF_DEV);
  T.int_len.assign(N, INF_DEV);
  for (int k = 1; k < N; k++) {
    T.scale[k] = T.scale[k - 1] / T.pf.pf_scale;
    T.eMLb[k] = T.eMLb[k - 1] * T.pf.MLbase / T.pf.pf_scale;
  }
  for (int u = 0; u < N; u++) {
    if (u <= 30) {
      T.hp_len[u] = T.hairpin[u];
      T.hp_w[u] = T.ehairpin[u];
      T.bulge_len[u] = T.mfe.bulge[u];
      T.int_len[u] = T.mfe.interior[u];
    } else {
      // ViennaRNA: X[30] + (int)(lxc * log(u / 30.)); PF uses the untruncated value (App. A.3/A.5)
      int ext = (int)(T.lxc * std::log(u / 30.0));
      T.hp_len[u] = T.hairpin[30] + ext;
      T.hp_w[u] = T.ehairpin[30] * std::exp(-(T.lxc * std::log(u / 30.0)) * 10.0 / kT);
      T.bulge_len[u] = T.mfe.bulge[30] + ext;
      T.int_len[u] = T.mfe.interior[30] + ext;
    }
    if (u + 2 < N) T.hp_w[u] *= T.scale[u + 2];
  }
}

}  // namespace drna
