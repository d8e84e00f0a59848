/*
 * oracle.c -- CPU restatement (plain C) of the scoring path; see oracle.h for scope and pinning.
 * TEST INFRASTRUCTURE ONLY: never linked into the product library.
 *
 * Each function names the reference call it stands for (reference = /root/reference, DesiRNA)
 * and the ViennaRNA 2.6.4 routine whose published algorithm it restates.
 */
#define _GNU_SOURCE
#include "oracle.h"

#include <math.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define INF 10000000
#define TURN 3
#define MAXLOOP 30
#define NBP 8
#define K0 273.15
#define GASCONST 1.98717
#define MIN2(a, b) ((a) < (b) ? (a) : (b))
#define MAX2(a, b) ((a) > (b) ? (a) : (b))

struct orc_params {
  /* integer model (dcal/mol), App. A.2 */
  int stack[NBP][NBP];
  int mmH[NBP][5][5], mmI[NBP][5][5], mm1nI[NBP][5][5], mm23I[NBP][5][5];
  int mmM[NBP][5][5], mmExt[NBP][5][5], d5[NBP][5], d3[NBP][5];             /* clamped <= 0 */
  int mmM_raw[NBP][5][5], mmExt_raw[NBP][5][5], d5_raw[NBP][5], d3_raw[NBP][5];
  int int11[NBP][NBP][5][5];
  int int21[NBP][NBP][5][5][5];
  int int22[NBP][NBP][5][5][5][5];
  int hairpin[31], bulge[31], interior[31];
  int ninio, max_ninio, MLbase, MLclosing, MLintern, DuplexInit, TerminalAU;
  double lxc;
  int n_tri, n_tetra, n_hexa;
  char tri[64][8], tetra[128][8], hexa[64][12];
  int tri_e[64], tetra_e[128], hexa_e[64];
  /* Boltzmann model, App. A.5 */
  double kT, pf_scale;
  double estack[NBP][NBP];
  double emmH[NBP][5][5], emmI[NBP][5][5], emm1nI[NBP][5][5], emm23I[NBP][5][5];
  double emmM[NBP][5][5], emmExt[NBP][5][5], ed5[NBP][5], ed3[NBP][5];
  double eint11[NBP][NBP][5][5];
  double eint21[NBP][NBP][5][5][5];
  double eint22[NBP][NBP][5][5][5][5];
  double ehairpin[31], ebulge[31], einterior[31], eninio[MAXLOOP + 1];
  double eMLbase, eMLclosing, eMLintern, eTermAU;
  double etri[64], etetra[128], ehexa[64];
};

static const int PAIR[5][5] = {
    /*       @  A  C  G  U */
    /*@*/ {0, 0, 0, 0, 0},
    /*A*/ {0, 0, 0, 0, 5},
    /*C*/ {0, 0, 0, 1, 0},
    /*G*/ {0, 0, 2, 0, 3},
    /*U*/ {0, 6, 0, 4, 0}};
static const int RTYPE[8] = {0, 2, 1, 4, 3, 6, 5, 7};

static int enc(char c) {
  switch (c) {
    case 'A': case 'a': return 1;
    case 'C': case 'c': return 2;
    case 'G': case 'g': return 3;
    case 'U': case 'u': case 'T': case 't': return 4;
    default: return 0;
  }
}

/* ViennaRNA params.c: SMOOTH() used for the dangle-type Boltzmann factors when pf_smooth=1 */
static double smooth(double x) {
  const double SC = 10.0;
  if (x / SC < -1.2283697) return 0.0;
  if (x / SC > 0.8660254) return x;
  double s = sin(x / SC - 0.34242663) + 1.0;
  return SC * 0.38490018 * s * s;
}

orc_params *orc_params_create(const int32_t *b, int n_int32) {
  if (n_int32 < 3 || b[0] != 0x504E5244 || b[1] != 1 || b[2] != n_int32) return NULL;
  orc_params *P = (orc_params *)calloc(1, sizeof(orc_params));
  const int32_t *p = b + 3;
#define TAKE(dst, cnt) do { memcpy((dst), p, (size_t)(cnt) * 4); p += (cnt); } while (0)
  TAKE(P->stack, 64);
  TAKE(P->mmH, 200); TAKE(P->mmI, 200); TAKE(P->mm1nI, 200); TAKE(P->mm23I, 200);
  TAKE(P->mmM_raw, 200); TAKE(P->mmExt_raw, 200);
  TAKE(P->d5_raw, 40); TAKE(P->d3_raw, 40);
  TAKE(P->int11, 1600); TAKE(P->int21, 8000); TAKE(P->int22, 40000);
  TAKE(P->hairpin, 31); TAKE(P->bulge, 31); TAKE(P->interior, 31);
  P->ninio = *p++; P->max_ninio = *p++; P->MLbase = *p++; P->MLclosing = *p++;
  P->MLintern = *p++; P->DuplexInit = *p++; P->TerminalAU = *p++;
  memcpy(&P->lxc, p, 8); p += 2;
  P->n_tri = *p++; P->n_tetra = *p++; P->n_hexa = *p++;
  for (int i = 0; i < P->n_tri; i++) { memset(P->tri[i], 0, 8); memcpy(P->tri[i], p, 8); P->tri_e[i] = p[2]; p += 3; }
  for (int i = 0; i < P->n_tetra; i++) { memset(P->tetra[i], 0, 8); memcpy(P->tetra[i], p, 8); P->tetra_e[i] = p[2]; p += 3; }
  for (int i = 0; i < P->n_hexa; i++) { memset(P->hexa[i], 0, 12); memcpy(P->hexa[i], p, 8); P->hexa_e[i] = p[2]; p += 3; }
#undef TAKE
  /* App. A.2: dangles != 0 -> multi/exterior mismatches and dangles are clamped to <= 0 */
  for (int t = 0; t < NBP; t++)
    for (int a = 0; a < 5; a++) {
      P->d5[t][a] = MIN2(0, P->d5_raw[t][a]);
      P->d3[t][a] = MIN2(0, P->d3_raw[t][a]);
      for (int c = 0; c < 5; c++) {
        P->mmM[t][a][c] = MIN2(0, P->mmM_raw[t][a][c]);
        P->mmExt[t][a][c] = MIN2(0, P->mmExt_raw[t][a][c]);
      }
    }
  /* App. A.5 */
  double kT = (37.0 + K0) * GASCONST; /* cal/mol */
  P->kT = kT;
  P->pf_scale = exp(1.07 * 185.0 / kT);
  if (P->pf_scale < 1.0) P->pf_scale = 1.0;
#define BF(e) exp(-(double)(e) * 10.0 / kT)
  for (int t = 0; t < NBP; t++) {
    for (int u = 0; u < NBP; u++) P->estack[t][u] = BF(P->stack[t][u]);
    for (int a = 0; a < 5; a++) {
      P->ed5[t][a] = exp(smooth(-(double)P->d5_raw[t][a]) * 10.0 / kT);
      P->ed3[t][a] = exp(smooth(-(double)P->d3_raw[t][a]) * 10.0 / kT);
      for (int c = 0; c < 5; c++) {
        P->emmH[t][a][c] = BF(P->mmH[t][a][c]);
        P->emmI[t][a][c] = BF(P->mmI[t][a][c]);
        P->emm1nI[t][a][c] = BF(P->mm1nI[t][a][c]);
        P->emm23I[t][a][c] = BF(P->mm23I[t][a][c]);
        P->emmM[t][a][c] = exp(smooth(-(double)P->mmM_raw[t][a][c]) * 10.0 / kT);
        P->emmExt[t][a][c] = exp(smooth(-(double)P->mmExt_raw[t][a][c]) * 10.0 / kT);
      }
    }
  }
  {
    const int *s11 = &P->int11[0][0][0][0]; double *d11 = &P->eint11[0][0][0][0];
    for (int i = 0; i < 1600; i++) d11[i] = BF(s11[i]);
    const int *s21 = &P->int21[0][0][0][0][0]; double *d21 = &P->eint21[0][0][0][0][0];
    for (int i = 0; i < 8000; i++) d21[i] = BF(s21[i]);
    const int *s22 = &P->int22[0][0][0][0][0][0]; double *d22 = &P->eint22[0][0][0][0][0][0];
    for (int i = 0; i < 40000; i++) d22[i] = BF(s22[i]);
  }
  for (int i = 0; i <= 30; i++) {
    P->ehairpin[i] = BF(P->hairpin[i]);
    P->ebulge[i] = BF(P->bulge[i]);
    P->einterior[i] = BF(P->interior[i]);
    P->eninio[i] = BF(MIN2(P->max_ninio, i * P->ninio));
  }
  P->eMLbase = BF(P->MLbase); P->eMLclosing = BF(P->MLclosing); P->eMLintern = BF(P->MLintern);
  P->eTermAU = BF(P->TerminalAU);
  for (int i = 0; i < P->n_tri; i++) P->etri[i] = BF(P->tri_e[i]);
  for (int i = 0; i < P->n_tetra; i++) P->etetra[i] = BF(P->tetra_e[i]);
  for (int i = 0; i < P->n_hexa; i++) P->ehexa[i] = BF(P->hexa_e[i]);
#undef BF
  return P;
}

void orc_params_destroy(orc_params *P) { free(P); }

/* ---------------------------------------------------------------- loop energies, App. A.3 */

static int special_index(const char (*tab)[8], int cnt, const char *s, int len) {
  for (int k = 0; k < cnt; k++)
    if (memcmp(tab[k], s, (size_t)len) == 0) return k;
  return -1;
}
static int special_index12(const char (*tab)[12], int cnt, const char *s, int len) {
  for (int k = 0; k < cnt; k++)
    if (memcmp(tab[k], s, (size_t)len) == 0) return k;
  return -1;
}

/* ViennaRNA E_Hairpin; str points at the closing 5' base (ASCII, upper-case ACGU) */
static int E_Hairpin(const orc_params *P, int u, int t, int si1, int sj1, const char *str) {
  int e = (u <= 30) ? P->hairpin[u] : P->hairpin[30] + (int)(P->lxc * log(u / 30.0));
  if (u < 3) return e;
  if (u == 4) {
    int k = special_index(P->tetra, P->n_tetra, str, 6);
    if (k >= 0) return P->tetra_e[k];
  } else if (u == 6) {
    int k = special_index12(P->hexa, P->n_hexa, str, 8);
    if (k >= 0) return P->hexa_e[k];
  } else if (u == 3) {
    int k = special_index(P->tri, P->n_tri, str, 5);
    if (k >= 0) return P->tri_e[k];
    return e + (t > 2 ? P->TerminalAU : 0);
  }
  return e + P->mmH[t][si1][sj1];
}

/* ViennaRNA E_IntLoop; t2 is rtype of the inner pair */
static int E_IntLoop(const orc_params *P, int n1, int n2, int t, int t2, int si1, int sj1, int sp1, int sq1) {
  int nl = MAX2(n1, n2), ns = MIN2(n1, n2), e;
  if (nl == 0) return P->stack[t][t2];
  if (ns == 0) {
    e = (nl <= MAXLOOP) ? P->bulge[nl] : P->bulge[30] + (int)(P->lxc * log(nl / 30.0));
    if (nl == 1) e += P->stack[t][t2];
    else {
      if (t > 2) e += P->TerminalAU;
      if (t2 > 2) e += P->TerminalAU;
    }
    return e;
  }
  if (ns == 1) {
    if (nl == 1) return P->int11[t][t2][si1][sj1];
    if (nl == 2) return (n1 == 1) ? P->int21[t][t2][si1][sq1][sj1] : P->int21[t2][t][sq1][si1][sp1];
    e = (nl + 1 <= MAXLOOP) ? P->interior[nl + 1] : P->interior[30] + (int)(P->lxc * log((nl + 1) / 30.0));
    e += MIN2(P->max_ninio, (nl - ns) * P->ninio);
    return e + P->mm1nI[t][si1][sj1] + P->mm1nI[t2][sq1][sp1];
  }
  if (ns == 2) {
    if (nl == 2) return P->int22[t][t2][si1][sp1][sq1][sj1];
    if (nl == 3) return P->interior[5] + P->ninio + P->mm23I[t][si1][sj1] + P->mm23I[t2][sq1][sp1];
  }
  {
    int u = nl + ns;
    e = (u <= MAXLOOP) ? P->interior[u] : P->interior[30] + (int)(P->lxc * log(u / 30.0));
    e += MIN2(P->max_ninio, (nl - ns) * P->ninio);
    return e + P->mmI[t][si1][sj1] + P->mmI[t2][sq1][sp1];
  }
}

/* s5 / s3 < 0 means "no neighbour" */
static int E_MLstem(const orc_params *P, int t, int s5, int s3) {
  int e = 0;
  if (s5 >= 0 && s3 >= 0) e = P->mmM[t][s5][s3];
  else if (s5 >= 0) e = P->d5[t][s5];
  else if (s3 >= 0) e = P->d3[t][s3];
  if (t > 2) e += P->TerminalAU;
  return e + P->MLintern;
}
static int E_ExtLoop(const orc_params *P, int t, int s5, int s3) {
  int e = 0;
  if (s5 >= 0 && s3 >= 0) e = P->mmExt[t][s5][s3];
  else if (s5 >= 0) e = P->d5[t][s5];
  else if (s3 >= 0) e = P->d3[t][s3];
  if (t > 2) e += P->TerminalAU;
  return e;
}

static double X_Hairpin(const orc_params *P, int u, int t, int si1, int sj1, const char *str) {
  double q = (u <= 30) ? P->ehairpin[u] : P->ehairpin[30] * exp(-(P->lxc * log(u / 30.0)) * 10.0 / P->kT);
  if (u < 3) return q;
  if (u == 4) {
    int k = special_index(P->tetra, P->n_tetra, str, 6);
    if (k >= 0) return P->etetra[k];
  } else if (u == 6) {
    int k = special_index12(P->hexa, P->n_hexa, str, 8);
    if (k >= 0) return P->ehexa[k];
  } else if (u == 3) {
    int k = special_index(P->tri, P->n_tri, str, 5);
    if (k >= 0) return P->etri[k];
    return (t > 2) ? q * P->eTermAU : q;
  }
  return q * P->emmH[t][si1][sj1];
}

static double X_IntLoop(const orc_params *P, int u1, int u2, int t, int t2, int si1, int sj1, int sp1, int sq1) {
  int ul = MAX2(u1, u2), us = MIN2(u1, u2);
  double z;
  if (ul == 0) return P->estack[t][t2];
  if (us == 0) {
    z = P->ebulge[ul];
    if (ul == 1) z *= P->estack[t][t2];
    else {
      if (t > 2) z *= P->eTermAU;
      if (t2 > 2) z *= P->eTermAU;
    }
    return z;
  }
  if (us == 1) {
    if (ul == 1) return P->eint11[t][t2][si1][sj1];
    if (ul == 2) return (u1 == 1) ? P->eint21[t][t2][si1][sq1][sj1] : P->eint21[t2][t][sq1][si1][sp1];
    z = P->einterior[ul + us] * P->emm1nI[t][si1][sj1] * P->emm1nI[t2][sq1][sp1];
    return z * P->eninio[ul - us];
  }
  if (us == 2) {
    if (ul == 2) return P->eint22[t][t2][si1][sp1][sq1][sj1];
    if (ul == 3) return P->einterior[5] * P->emm23I[t][si1][sj1] * P->emm23I[t2][sq1][sp1] * P->eninio[1];
  }
  z = P->einterior[ul + us] * P->emmI[t][si1][sj1] * P->emmI[t2][sq1][sp1];
  return z * P->eninio[ul - us];
}
static double X_MLstem(const orc_params *P, int t, int s5, int s3) {
  double e = 1.0;
  if (s5 >= 0 && s3 >= 0) e = P->emmM[t][s5][s3];
  else if (s5 >= 0) e = P->ed5[t][s5];
  else if (s3 >= 0) e = P->ed3[t][s3];
  if (t > 2) e *= P->eTermAU;
  return e * P->eMLintern;
}
static double X_ExtLoop(const orc_params *P, int t, int s5, int s3) {
  double e = 1.0;
  if (s5 >= 0 && s3 >= 0) e = P->emmExt[t][s5][s3];
  else if (s5 >= 0) e = P->ed5[t][s5];
  else if (s3 >= 0) e = P->ed3[t][s3];
  if (t > 2) e *= P->eTermAU;
  return e;
}

/* ---------------------------------------------------------------- eval_structure, App. A.6/A.7 */

typedef struct {
  const orc_params *P;
  const int *S;   /* 1..n, S[0]=S[n], S[n+1]=S[1] */
  const int *pt;  /* 1..n, 0 = unpaired */
  const char *seq; /* 0-based ASCII upper */
  int n, cut;     /* cut = last position of strand 1 (0 = single strand) */
} evalctx;

static int ptype_eval(const evalctx *E, int i, int j) {
  int t = PAIR[E->S[i]][E->S[j]];
  return t ? t : 7;
}
static int same_strand(const evalctx *E, int a, int b) {
  if (a < 1 || b < 1 || a > E->n || b > E->n) return 0;
  if (E->cut <= 0) return 1;
  return (a <= E->cut) == (b <= E->cut);
}

/* energy of the loop closed by (i,j) plus everything it encloses (ViennaRNA eval.c stack_energy) */
static int eval_loop(const evalctx *E, int i, int j) {
  const orc_params *P = E->P;
  const int *S = E->S, *pt = E->pt;
  int energy = 0;
  /* collect the stems directly inside (i,j) */
  int nstems = 0, unpaired = 0, p = i + 1;
  int first_p = 0, first_q = 0;
  int nick_here = 0;
  /* does the backbone of this loop contain the nick (cut | cut+1)? */
  if (E->cut > 0 && i <= E->cut && E->cut < j) {
    nick_here = 1;
    int k = i + 1;
    while (k < j) {
      if (pt[k] > k) {
        if (k <= E->cut && E->cut < pt[k]) { nick_here = 0; break; }
        k = pt[k] + 1;
      } else k++;
    }
  }
  while (p < j) {
    if (pt[p] > p) {
      if (!nstems) { first_p = p; first_q = pt[p]; }
      nstems++;
      p = pt[p] + 1;
    } else { unpaired++; p++; }
  }
  if (nick_here) {
    /* App. A.7: loop with the nick is scored like an exterior loop */
    int tt = ptype_eval(E, j, i);
    energy += E_ExtLoop(P, tt, same_strand(E, j - 1, j) ? S[j - 1] : -1, same_strand(E, i + 1, i) ? S[i + 1] : -1);
    p = i + 1;
    while (p < j) {
      if (pt[p] > p) {
        int q = pt[p];
        int t2 = ptype_eval(E, p, q);
        energy += E_ExtLoop(P, t2, same_strand(E, p - 1, p) ? S[p - 1] : -1, same_strand(E, q + 1, q) ? S[q + 1] : -1);
        energy += eval_loop(E, p, q);
        p = q + 1;
      } else p++;
    }
    return energy;
  }
  int t = ptype_eval(E, i, j);
  if (nstems == 0) return E_Hairpin(P, j - i - 1, t, S[i + 1], S[j - 1], E->seq + i - 1);
  if (nstems == 1) {
    int pp = first_p, q = first_q;
    int t2 = ptype_eval(E, q, pp);
    energy = E_IntLoop(P, pp - i - 1, j - q - 1, t, t2, S[i + 1], S[j - 1], S[pp - 1], S[q + 1]);
    return energy + eval_loop(E, pp, q);
  }
  /* multiloop (ViennaRNA energy_of_ml_pt, dangles=2) */
  energy = P->MLclosing + E_MLstem(P, ptype_eval(E, j, i), S[j - 1], S[i + 1]) + unpaired * P->MLbase;
  p = i + 1;
  while (p < j) {
    if (pt[p] > p) {
      int q = pt[p];
      energy += E_MLstem(P, ptype_eval(E, p, q), S[p - 1], S[q + 1]);
      energy += eval_loop(E, p, q);
      p = q + 1;
    } else p++;
  }
  return energy;
}

static int *encode_seq(const char *seq, int n) {
  int *S = (int *)calloc((size_t)n + 2, sizeof(int));
  for (int i = 1; i <= n; i++) S[i] = enc(seq[i - 1]);
  S[0] = S[n];
  S[n + 1] = S[1];
  return S;
}

int orc_eval_structure_cut(const orc_params *P, const char *seq, const char *db, int n, int cut) {
  int *S = encode_seq(seq, n);
  int *pt = (int *)calloc((size_t)n + 2, sizeof(int));
  int *stk = (int *)malloc(sizeof(int) * (size_t)(n + 1));
  char *up = (char *)malloc((size_t)n + 1);
  int sp = 0, energy = 0, bad = 0;
  for (int i = 0; i < n; i++) {
    char ch = seq[i];
    up[i] = (ch >= 'a' && ch <= 'z') ? (char)(ch - 32) : ch;
    if (up[i] == 'T') up[i] = 'U';
  }
  up[n] = 0;
  for (int i = 1; i <= n; i++) {
    if (db[i - 1] == '(') stk[sp++] = i;
    else if (db[i - 1] == ')') {
      if (!sp) { bad = 1; break; }
      int o = stk[--sp];
      pt[o] = i; pt[i] = o;
    }
  }
  if (sp) bad = 1;
  if (!bad) {
    evalctx E = {P, S, pt, up, n, cut};
    int inter = 0;
    for (int i = 1; i <= n;) {
      if (pt[i] > i) {
        int j = pt[i];
        int t = ptype_eval(&E, i, j);
        int s5 = (i > 1 && same_strand(&E, i - 1, i)) ? S[i - 1] : -1;
        int s3 = (j < n && same_strand(&E, j + 1, j)) ? S[j + 1] : -1;
        energy += E_ExtLoop(P, t, s5, s3);
        energy += eval_loop(&E, i, j);
        i = j + 1;
      } else i++;
    }
    if (cut > 0)
      for (int i = 1; i <= cut; i++)
        if (pt[i] > cut) inter = 1;
    if (inter) energy += P->DuplexInit;
  } else energy = INF;
  free(S); free(pt); free(stk); free(up);
  return energy;
}

int orc_eval_structure(const orc_params *P, const char *seq, const char *db, int n) {
  return orc_eval_structure_cut(P, seq, db, n, 0);
}

/* ---------------------------------------------------------------- MFE fill + traceback, App. A.4 */

typedef struct {
  int n;
  int *S;
  char *up;            /* upper-case ASCII */
  unsigned char *pty;  /* (n+2)*(n+2) pair type or 0 */
  int *c, *fML, *fMLt, *f5;
} mfectx;

#define IDX(i, j) ((size_t)(i) * (size_t)(W) + (size_t)(j))

/* per-thread scratch buffers, reused across calls: a fold allocates ~2 MB of tables, and with one
 * fold per core the page faults of fresh mmap'ed memory would dominate the timed CPU baseline */
#define TL_SLOTS 24
static __thread void *tl_buf[TL_SLOTS];
static __thread size_t tl_cap[TL_SLOTS];
static void *tl_get(int slot, size_t bytes) {
  if (tl_cap[slot] < bytes) {
    free(tl_buf[slot]);
    tl_buf[slot] = malloc(bytes);
    tl_cap[slot] = bytes;
  }
  return tl_buf[slot];
}

static void mfe_fill(const orc_params *P, mfectx *M, const char *seq, int n, const unsigned char *nopair) {
  const int W = n + 2;
  M->n = n;
  M->S = encode_seq(seq, n);
  M->up = (char *)malloc((size_t)n + 1);
  for (int i = 0; i < n; i++) {
    char ch = seq[i];
    ch = (ch >= 'a' && ch <= 'z') ? (char)(ch - 32) : ch;
    M->up[i] = (ch == 'T') ? 'U' : ch;
  }
  M->up[n] = 0;
  size_t W2 = (size_t)W * (size_t)W;
  M->pty = (unsigned char *)tl_get(0, W2);
  memset(M->pty, 0, W2);
  M->c = (int *)tl_get(1, W2 * sizeof(int));
  M->fML = (int *)tl_get(2, W2 * sizeof(int));
  M->fMLt = (int *)tl_get(3, W2 * sizeof(int));
  M->f5 = (int *)calloc((size_t)n + 2, sizeof(int));
  int *S = M->S, *c = M->c, *fML = M->fML, *fMLt = M->fMLt;
  int *DML = (int *)tl_get(4, W2 * sizeof(int)); /* decomp[i][j] = min_u fML[i,u]+fML[u+1,j] */
  int *ci = (int *)tl_get(5, W2 * sizeof(int));  /* c + mismatchI of the pair seen as the inner pair of a loop */
  int Lgen[MAXLOOP + 1][MAXLOOP + 2];
  for (int a = 0; a <= MAXLOOP; a++)
    for (int b = 0; b <= MAXLOOP + 1; b++) {
      int u = a + b, nl = MAX2(a, b), ns = MIN2(a, b);
      Lgen[a][b] = (u <= MAXLOOP) ? P->interior[u] + MIN2(P->max_ninio, (nl - ns) * P->ninio) : INF;
    }
  for (size_t k = 0; k < W2; k++) { c[k] = INF; fML[k] = INF; fMLt[k] = INF; DML[k] = INF; ci[k] = INF; }
  for (int i = 1; i <= n; i++)
    for (int j = i + TURN + 1; j <= n; j++) {
      if (nopair && (nopair[i - 1] || nopair[j - 1])) continue;
      M->pty[IDX(i, j)] = (unsigned char)PAIR[S[i]][S[j]];
    }
  /* ViennaRNA mfe.c fill_arrays: i descending, j ascending */
  for (int i = n - TURN - 1; i >= 1; i--) {
    for (int j = i + TURN + 1; j <= n; j++) {
      int t = M->pty[IDX(i, j)];
      int e = INF;
      if (t) {
        e = E_Hairpin(P, j - i - 1, t, S[i + 1], S[j - 1], M->up + i - 1);
        /* multiloop closed by (i,j): decomp of [i+1, j-1] */
        int d = DML[IDX(i + 1, j - 1)];
        if (d < INF) {
          d += P->MLclosing + E_MLstem(P, RTYPE[t], S[j - 1], S[i + 1]);
          e = MIN2(e, d);
        }
        /* interior loops (vrna_E_int_loop).  Same candidates and energies as E_IntLoop; the generic
         * shapes (both sides >= 2, not 2x2 / 2x3) read c + mismatchI(inner) precombined in ci[] and a
         * 31x31 size table, so that loop vectorises -- this oracle is also the timed CPU baseline */
        int pmax = MIN2(j - 2 - TURN, i + MAXLOOP + 1);
        int egen = INF;
        for (int p = i + 1; p <= pmax; p++) {
          int u1 = p - i - 1;
          int minq = j - i + p - MAXLOOP - 2;
          if (minq < p + 1 + TURN) minq = p + 1 + TURN;
          const unsigned char *prow = M->pty + IDX(p, 0);
          const int *crow = c + IDX(p, 0);
          int u2max = j - 1 - minq;                 /* u2 = j - q - 1 in [0, u2max] */
          int g0 = u1 < 2 ? u2max + 1 : (u1 == 2 ? 4 : (u1 == 3 ? 3 : 2));   /* first generic u2 */
          int sp_hi = MIN2(g0 - 1, u2max);
          for (int u2 = 0; u2 <= sp_hi; u2++) {
            int q = j - 1 - u2;
            int t2 = prow[q];
            if (!t2) continue;
            int en = crow[q];
            if (en >= INF) continue;
            en += E_IntLoop(P, u1, u2, t, RTYPE[t2], S[i + 1], S[j - 1], S[p - 1], S[q + 1]);
            e = MIN2(e, en);
          }
          if (g0 <= u2max) {
            const int *cir = ci + IDX(p, 0) + (j - 1);
            const int *Lr = Lgen[u1];
            int m = INF;
            for (int u2 = g0; u2 <= u2max; u2++) {
              int v = cir[-u2] + Lr[u2];
              m = MIN2(m, v);
            }
            egen = MIN2(egen, m);
          }
        }
        if (egen < INF / 2) e = MIN2(e, egen + P->mmI[t][S[i + 1]][S[j - 1]]);
      }
      c[IDX(i, j)] = e;
      ci[IDX(i, j)] = (t && e < INF) ? e + P->mmI[RTYPE[t]][S[j + 1]][S[i - 1]] : INF;
      /* fML (vrna_E_ml_stems_fast, dangles=2) */
      int f = INF;
      if (fML[IDX(i + 1, j)] < INF) f = fML[IDX(i + 1, j)] + P->MLbase;
      if (fML[IDX(i, j - 1)] < INF) f = MIN2(f, fML[IDX(i, j - 1)] + P->MLbase);
      if (e < INF) f = MIN2(f, e + E_MLstem(P, t, S[i - 1], S[j + 1]));
      int dec = INF;
      {
        const int *row = fML + IDX(i, 0);
        const int *col = fMLt + IDX(j, 0);
        for (int u = i + 1 + TURN; u <= j - 2 - TURN; u++) {
          int v = row[u] + col[u + 1];
          dec = MIN2(dec, v);
        }
      }
      if (dec >= INF / 2) dec = INF; /* INF + finite must stay "no decomposition" */
      DML[IDX(i, j)] = dec;
      f = MIN2(f, dec);
      fML[IDX(i, j)] = f;
      fMLt[IDX(j, i)] = f;
    }
  }
  /* exterior loop */
  int *f5 = M->f5;
  for (int j = 0; j <= MIN2(TURN + 1, n); j++) f5[j] = 0;
  for (int j = TURN + 2; j <= n; j++) {
    int f = f5[j - 1];
    for (int i = j - TURN - 1; i >= 1; i--) {
      int t = M->pty[IDX(i, j)];
      if (!t || c[IDX(i, j)] >= INF) continue;
      int en = f5[i - 1] + c[IDX(i, j)] + E_ExtLoop(P, t, i > 1 ? S[i - 1] : -1, j < n ? S[j + 1] : -1);
      f = MIN2(f, en);
    }
    f5[j] = f;
  }
}

static void mfe_free(mfectx *M) {
  free(M->S); free(M->up); free(M->f5);   /* the big tables live in the per-thread arena */
}

/* ViennaRNA mfe.c backtrack(): sector stack; ml: 0 = exterior f5[1..j], 1 = fML[i..j], 2 = pair (i,j) */
static void mfe_traceback(const orc_params *P, const mfectx *M, char *ss) {
  const int n = M->n, W = n + 2;
  const int *S = M->S, *c = M->c, *fML = M->fML, *f5 = M->f5;
  typedef struct { int i, j, ml; } sect;
  sect *st = (sect *)malloc(sizeof(sect) * (size_t)(4 * n + 8));
  int sp = 0;
  memset(ss, '.', (size_t)n);
  ss[n] = 0;
  st[sp++] = (sect){1, n, 0};
  while (sp > 0) {
    sect s = st[--sp];
    int i = s.i, j = s.j;
    if (s.ml == 0) {
      /* vrna_BT_ext_loop_f5 */
      if (j < TURN + 2) continue;
      int jj = j;
      while (jj > 0 && f5[jj] == f5[jj - 1]) jj--;
      if (jj < TURN + 2) continue;
      int fij = f5[jj], found = 0;
      for (int u = jj - TURN - 1; u >= 1; u--) {
        int t = M->pty[IDX(u, jj)];
        if (!t || c[IDX(u, jj)] >= INF) continue;
        int en = c[IDX(u, jj)] + E_ExtLoop(P, t, u > 1 ? S[u - 1] : -1, jj < n ? S[jj + 1] : -1);
        if (fij == en + f5[u - 1]) {
          st[sp++] = (sect){1, u - 1, 0};
          st[sp++] = (sect){u, jj, 2};
          found = 1;
          break;
        }
      }
      if (!found) { fprintf(stderr, "oracle: f5 traceback failed at %d\n", jj); break; }
      continue;
    }
    if (s.ml == 1) {
      /* vrna_BT_mb_loop_split */
      while (j > i && fML[IDX(i, j)] == fML[IDX(i, j - 1)] + P->MLbase) j--;
      while (i < j && fML[IDX(i, j)] == fML[IDX(i + 1, j)] + P->MLbase) i++;
      if (j < i + TURN + 1) { fprintf(stderr, "oracle: fML traceback underflow\n"); break; }
      int fij = fML[IDX(i, j)];
      int t = M->pty[IDX(i, j)];
      if (t && c[IDX(i, j)] < INF && fij == c[IDX(i, j)] + E_MLstem(P, t, S[i - 1], S[j + 1])) {
        st[sp++] = (sect){i, j, 2};
        continue;
      }
      int found = 0;
      for (int u = i + 1 + TURN; u <= j - 2 - TURN; u++)
        if (fij == fML[IDX(i, u)] + fML[IDX(u + 1, j)]) {
          st[sp++] = (sect){i, u, 1};
          st[sp++] = (sect){u + 1, j, 1};
          found = 1;
          break;
        }
      if (!found) { fprintf(stderr, "oracle: fML traceback failed at %d,%d\n", i, j); break; }
      continue;
    }
    /* pair (i,j): hairpin, interior (p ascending, q descending), multiloop */
    for (;;) {
      ss[i - 1] = '(';
      ss[j - 1] = ')';
      int t = M->pty[IDX(i, j)];
      int cij = c[IDX(i, j)];
      if (cij == E_Hairpin(P, j - i - 1, t, S[i + 1], S[j - 1], M->up + i - 1)) break;
      int found = 0;
      int pmax = MIN2(j - 2 - TURN, i + MAXLOOP + 1);
      for (int p = i + 1; p <= pmax && !found; p++) {
        int minq = j - i + p - MAXLOOP - 2;
        if (minq < p + 1 + TURN) minq = p + 1 + TURN;
        for (int q = j - 1; q >= minq; q--) {
          int t2 = M->pty[IDX(p, q)];
          if (!t2 || c[IDX(p, q)] >= INF) continue;
          int en = E_IntLoop(P, p - i - 1, j - q - 1, t, RTYPE[t2], S[i + 1], S[j - 1], S[p - 1], S[q + 1]);
          if (cij == en + c[IDX(p, q)]) {
            i = p; j = q; found = 1;
            break;
          }
        }
      }
      if (found) continue;
      /* vrna_BT_mb_loop */
      int e = cij - P->MLclosing - E_MLstem(P, RTYPE[t], S[j - 1], S[i + 1]);
      for (int u = i + 2 + TURN; u < j - 2 - TURN; u++)
        if (e == fML[IDX(i + 1, u)] + fML[IDX(u + 1, j - 1)]) {
          st[sp++] = (sect){i + 1, u, 1};
          st[sp++] = (sect){u + 1, j - 1, 1};
          found = 1;
          break;
        }
      if (!found) fprintf(stderr, "oracle: pair traceback failed at %d,%d\n", i, j);
      break;
    }
  }
  free(st);
}

int orc_mfe(const orc_params *P, const char *seq, int n, const unsigned char *nopair, char *ss) {
  mfectx M;
  mfe_fill(P, &M, seq, n, nopair);
  int e = M.f5[n];
  if (ss) mfe_traceback(P, &M, ss);
  mfe_free(&M);
  return e;
}

int orc_mfe_tables(const orc_params *P, const char *seq, int n, const unsigned char *nopair,
                   int32_t *c, int32_t *fML, int32_t *f5) {
  mfectx M;
  mfe_fill(P, &M, seq, n, nopair);
  size_t W2 = (size_t)(n + 2) * (size_t)(n + 2);
  if (c) memcpy(c, M.c, W2 * 4);
  if (fML) memcpy(fML, M.fML, W2 * 4);
  if (f5) memcpy(f5, M.f5, ((size_t)n + 1) * 4);
  int e = M.f5[n];
  mfe_free(&M);
  return e;
}

/* reference utils/sequence_utils.py:1166-1228 */
void orc_pk_struct(const orc_params *P, const char *seq, int n, const char *ss_nopk, char *ss_pk) {
  static const char OPEN[3] = {'[', '<', '{'}, CLOSE[3] = {']', '>', '}'};
  unsigned char *mask = (unsigned char *)calloc((size_t)n, 1);
  char *ss = (char *)malloc((size_t)n + 1);
  memcpy(ss_pk, ss_nopk, (size_t)n);
  ss_pk[n] = 0;
  for (int round = 0; round < 3; round++) {
    for (int k = 0; k < n; k++)
      if (ss_pk[k] != '.') mask[k] = 1; /* every bracket so far -> 'x' */
    orc_mfe(P, seq, n, mask, ss);
    int any = 0;
    for (int k = 0; k < n; k++) {
      if (ss[k] == '(') { ss_pk[k] = OPEN[round]; any = 1; }
      else if (ss[k] == ')') ss_pk[k] = CLOSE[round];
    }
    if (!any) break;
  }
  free(mask); free(ss);
}

/* ---------------------------------------------------------------- partition function, App. A.5 */

typedef struct {
  int n;
  int *S;
  char *up;
  unsigned char *pty;
  double *qb, *qm, *qm1, *q5, *scale, *eMLb;
} pfctx;

static void pf_fill(const orc_params *P, pfctx *F, const char *seq, int n) {
  const int W = n + 2;
  size_t W2 = (size_t)W * (size_t)W;
  F->n = n;
  F->S = encode_seq(seq, n);
  F->up = (char *)malloc((size_t)n + 1);
  for (int i = 0; i < n; i++) {
    char ch = seq[i];
    ch = (ch >= 'a' && ch <= 'z') ? (char)(ch - 32) : ch;
    F->up[i] = (ch == 'T') ? 'U' : ch;
  }
  F->up[n] = 0;
  F->pty = (unsigned char *)tl_get(8, W2);
  F->qb = (double *)tl_get(9, W2 * sizeof(double));
  F->qm = (double *)tl_get(10, W2 * sizeof(double));
  F->qm1 = (double *)tl_get(11, W2 * sizeof(double));
  memset(F->pty, 0, W2);
  memset(F->qb, 0, W2 * sizeof(double));
  memset(F->qm, 0, W2 * sizeof(double));
  memset(F->qm1, 0, W2 * sizeof(double));
  F->q5 = (double *)calloc((size_t)n + 2, sizeof(double));
  F->scale = (double *)malloc(sizeof(double) * (size_t)(n + 3));
  F->eMLb = (double *)malloc(sizeof(double) * (size_t)(n + 3));
  int *S = F->S;
  double *qb = F->qb, *qm = F->qm, *qm1 = F->qm1, *scale = F->scale, *eMLb = F->eMLb;
  double *qm1t = (double *)tl_get(12, W2 * sizeof(double)); /* qm1t[j][k] = qm1[k][j] */
  double *qbi = (double *)tl_get(13, W2 * sizeof(double));
  memset(qm1t, 0, W2 * sizeof(double));
  memset(qbi, 0, W2 * sizeof(double));  /* qb * expMismatchI of the pair seen as an inner pair */
  double Wgen[MAXLOOP + 1][MAXLOOP + 2];
  scale[0] = 1.0; eMLb[0] = 1.0;
  for (int k = 1; k <= n + 2; k++) {
    scale[k] = scale[k - 1] / P->pf_scale;
    eMLb[k] = eMLb[k - 1] * P->eMLbase / P->pf_scale; /* expMLbase^k * scale[k] */
  }
  for (int a = 0; a <= MAXLOOP; a++)
    for (int b2 = 0; b2 <= MAXLOOP + 1; b2++) {
      int u = a + b2, nl = MAX2(a, b2), ns = MIN2(a, b2);
      Wgen[a][b2] = (u <= MAXLOOP && u + 2 <= n + 2) ? P->einterior[u] * P->eninio[nl - ns] * scale[u + 2] : 0.0;
    }
  for (int i = 1; i <= n; i++)
    for (int j = i + TURN + 1; j <= n; j++) F->pty[IDX(i, j)] = (unsigned char)PAIR[S[i]][S[j]];
  /* ViennaRNA part_func.c fill_arrays: j ascending, i descending */
  for (int j = TURN + 2; j <= n; j++) {
    for (int i = j - TURN - 1; i >= 1; i--) {
      int t = F->pty[IDX(i, j)];
      double b = 0.0;
      if (t) {
        int u = j - i - 1;
        b = X_Hairpin(P, u, t, S[i + 1], S[j - 1], F->up + i - 1) * scale[u + 2];
        int pmax = MIN2(j - 2 - TURN, i + MAXLOOP + 1);
        double bgen = 0.0;
        for (int p = i + 1; p <= pmax; p++) {
          int u1 = p - i - 1;
          int minq = j - i + p - MAXLOOP - 2;
          if (minq < p + 1 + TURN) minq = p + 1 + TURN;
          int u2max = j - 1 - minq;
          int g0 = u1 < 2 ? u2max + 1 : (u1 == 2 ? 4 : (u1 == 3 ? 3 : 2));
          int sp_hi = MIN2(g0 - 1, u2max);
          for (int u2 = 0; u2 <= sp_hi; u2++) {
            int q = j - 1 - u2;
            int t2 = F->pty[IDX(p, q)];
            if (!t2) continue;
            b += qb[IDX(p, q)] * X_IntLoop(P, u1, u2, t, RTYPE[t2], S[i + 1], S[j - 1], S[p - 1], S[q + 1]) * scale[u1 + u2 + 2];
          }
          if (g0 <= u2max) {
            const double *qir = qbi + IDX(p, 0) + (j - 1);
            const double *Wr = Wgen[u1];
            double m = 0.0;
            for (int u2 = g0; u2 <= u2max; u2++) m += qir[-u2] * Wr[u2];
            bgen += m;
          }
        }
        b += bgen * P->emmI[t][S[i + 1]][S[j - 1]];
        /* multiloop: sum_k qm[i+1,k-1] * qm1[k,j-1] */
        double tmp = 0.0;
        const double *qmrow = qm + IDX(i + 1, 0);
        const double *q1col = qm1t + IDX(j - 1, 0);
        for (int k = i + 2 + TURN + 1; k <= j - 1 - TURN - 1; k++) tmp += qmrow[k - 1] * q1col[k];
        b += tmp * P->eMLclosing * X_MLstem(P, RTYPE[t], S[j - 1], S[i + 1]) * scale[2];
      }
      qb[IDX(i, j)] = b;
      qbi[IDX(i, j)] = t ? b * P->emmI[RTYPE[t]][S[j + 1]][S[i - 1]] : 0.0;
      /* qm1[i,j] = qm1[i,j-1]*expMLbase[1] + qb*expMLstem */
      double m1 = qm1[IDX(i, j - 1)] * eMLb[1];
      if (t) m1 += b * X_MLstem(P, t, i > 1 ? S[i - 1] : -1, j < n ? S[j + 1] : -1);
      qm1[IDX(i, j)] = m1;
      qm1t[IDX(j, i)] = m1;
      /* qm[i,j] = qm1[i,j] + sum_{k=i+1..j} (qm[i,k-1] + expMLbase[k-i]) * qm1[k,j] */
      double m = m1;
      {
        const double *qmrow = qm + IDX(i, 0);
        const double *q1col = qm1t + IDX(j, 0);
        for (int k = i + 1; k <= j - TURN - 1; k++) m += (qmrow[k - 1] + eMLb[k - i]) * q1col[k];
      }
      qm[IDX(i, j)] = m;
    }
  }
  double *q5 = F->q5;
  q5[0] = 1.0;
  for (int j = 1; j <= n; j++) {
    double q = q5[j - 1] * scale[1];
    for (int i = j - TURN - 1; i >= 1; i--) {
      int t = F->pty[IDX(i, j)];
      if (!t) continue;
      q += q5[i - 1] * qb[IDX(i, j)] * X_ExtLoop(P, t, i > 1 ? S[i - 1] : -1, j < n ? S[j + 1] : -1);
    }
    q5[j] = q;
  }
}

static void pf_free(pfctx *F) {
  free(F->S); free(F->up); free(F->q5);   /* the big tables live in the per-thread arena */
  free(F->scale); free(F->eMLb);
}

double orc_pf(const orc_params *P, const char *seq, int n) {
  pfctx F;
  pf_fill(P, &F, seq, n);
  double Z = F.q5[n];
  double e = (-log(Z) - n * log(P->pf_scale)) * P->kT / 1000.0;
  pf_free(&F);
  return e;
}

/* Outside recursion -> base-pair probabilities -> ensemble defect (reference
 * energy_scores.py:362-374; ViennaRNA vrna_ensemble_defect).  NOT golden-pinned (no vector in the
 * reference); Z, every P(i,j) and the defect are checked against the explicit sum over ALL structures weighted by
 * orc_boltzmann_weight (tests/test_oracle_golden.py::test_pf_bpp_defect_against_enumeration). */
double orc_ensemble_defect(const orc_params *P, const char *seq, int n, const char *target, double *bpp) {
  pfctx F;
  pf_fill(P, &F, seq, n);
  const int W = n + 2;
  size_t W2 = (size_t)W * (size_t)W;
  int *S = F.S;
  double *qb = F.qb, *qm = F.qm, *qm1 = F.qm1, *scale = F.scale, *eMLb = F.eMLb;
  double *Ob = (double *)calloc(W2, sizeof(double));
  double *Om = (double *)calloc(W2, sizeof(double));
  double *Om1 = (double *)calloc(W2, sizeof(double));
  double *q3 = (double *)calloc((size_t)n + 3, sizeof(double));
  q3[n + 1] = 1.0;
  for (int i = n; i >= 1; i--) {
    double q = q3[i + 1] * scale[1];
    for (int j = i + TURN + 1; j <= n; j++) {
      int t = F.pty[IDX(i, j)];
      if (!t) continue;
      q += qb[IDX(i, j)] * X_ExtLoop(P, t, i > 1 ? S[i - 1] : -1, j < n ? S[j + 1] : -1) * q3[j + 1];
    }
    q3[i] = q;
  }
  for (int d = n - 1; d >= TURN + 1; d--) {
    for (int i = 1; i + d <= n; i++) {
      int j = i + d;
      /* O_qm[i,j] */
      double om = Om[IDX(i, j)];
      /* O_qm1[i,j] += O_qm[i,j] */
      double om1 = Om1[IDX(i, j)] + om;
      Om1[IDX(i, j)] = om1;
      /* push O_qm[i,j] to the split terms: qm[i,j] ∋ (qm[i,u-1] + b^(u-i)) * qm1[u,j] */
      if (om != 0.0)
        for (int u = i + 1; u <= j - TURN - 1; u++) {
          double m1 = qm1[IDX(u, j)];
          if (m1 == 0.0) continue;
          Om[IDX(i, u - 1)] += om * m1;
          Om1[IDX(u, j)] += om * (qm[IDX(i, u - 1)] + eMLb[u - i]);
        }
      /* qm1[i,j] = qm1[i,j-1]*b + qb[i,j]*stem */
      if (om1 != 0.0) Om1[IDX(i, j - 1)] += om1 * eMLb[1];
      int t = F.pty[IDX(i, j)];
      if (!t) continue;
      double ob = Ob[IDX(i, j)];
      ob += om1 * X_MLstem(P, t, i > 1 ? S[i - 1] : -1, j < n ? S[j + 1] : -1);
      ob += F.q5[i - 1] * q3[j + 1] * X_ExtLoop(P, t, i > 1 ? S[i - 1] : -1, j < n ? S[j + 1] : -1);
      Ob[IDX(i, j)] = ob;
      if (ob == 0.0) continue;
      /* (i,j) closes an interior loop with inner pair (p,q) */
      int pmax = MIN2(j - 2 - TURN, i + MAXLOOP + 1);
      for (int p = i + 1; p <= pmax; p++) {
        int u1 = p - i - 1;
        int minq = j - i + p - MAXLOOP - 2;
        if (minq < p + 1 + TURN) minq = p + 1 + TURN;
        for (int q = j - 1; q >= minq; q--) {
          int t2 = F.pty[IDX(p, q)];
          if (!t2) continue;
          int u2 = j - q - 1;
          Ob[IDX(p, q)] += ob * X_IntLoop(P, u1, u2, t, RTYPE[t2], S[i + 1], S[j - 1], S[p - 1], S[q + 1]) * scale[u1 + u2 + 2];
        }
      }
      /* (i,j) closes a multiloop: qb ∋ close * qm[i+1,k-1] * qm1[k,j-1] */
      double cl = ob * P->eMLclosing * X_MLstem(P, RTYPE[t], S[j - 1], S[i + 1]) * scale[2];
      for (int k = i + 2 + TURN + 1; k <= j - 1 - TURN - 1; k++) {
        double a = qm[IDX(i + 1, k - 1)], m1 = qm1[IDX(k, j - 1)];
        if (a == 0.0 || m1 == 0.0) continue;
        Om[IDX(i + 1, k - 1)] += cl * m1;
        Om1[IDX(k, j - 1)] += cl * a;
      }
    }
  }
  double Z = F.q5[n];
  /* ensemble defect */
  int *pt = (int *)calloc((size_t)n + 2, sizeof(int));
  int *stk = (int *)malloc(sizeof(int) * (size_t)(n + 1));
  int sp = 0;
  for (int i = 1; i <= n; i++) {
    if (target[i - 1] == '(') stk[sp++] = i;
    else if (target[i - 1] == ')' && sp) { int o = stk[--sp]; pt[o] = i; pt[i] = o; }
  }
  double *pi = (double *)calloc((size_t)n + 2, sizeof(double));
  if (bpp) memset(bpp, 0, sizeof(double) * (size_t)(n + 1) * (size_t)(n + 1));
  double ed = 0.0;
  for (int i = 1; i <= n; i++)
    for (int j = i + TURN + 1; j <= n; j++) {
      if (!F.pty[IDX(i, j)]) continue;
      double p = Ob[IDX(i, j)] * qb[IDX(i, j)] / Z;
      if (bpp) bpp[(size_t)i * (size_t)(n + 1) + (size_t)j] = p;
      pi[i] += p; pi[j] += p;
    }
  for (int i = 1; i <= n; i++) {
    if (pt[i] == 0) ed += pi[i];
    else {
      int a = MIN2(i, pt[i]), b2 = MAX2(i, pt[i]);
      double p = F.pty[IDX(a, b2)] ? Ob[IDX(a, b2)] * qb[IDX(a, b2)] / Z : 0.0;
      ed += 1.0 - p;
    }
  }
  ed /= n;
  free(Ob); free(Om); free(Om1); free(q3); free(pt); free(stk); free(pi);
  pf_free(&F);
  return ed;
}

/* Boltzmann weight of ONE structure under the partition function's loop model (the X_* factors with pf_smooth,
 * untruncated log extrapolation; no pf_scale): exp(-E_pf(seq, db) / kT).  A structure walk, independent of pf_fill and
 * of the outside recursion, so that Z, the pair probabilities and the ensemble defect can be checked against an explicit
 * sum over all structures (tests/test_oracle_golden.py).  Only '(' ')' pair; pairs must be canonical (0 otherwise). */
static double weight_loop(const orc_params *P, const int *S, const int *pt, const char *up, int i, int j) {
  int t = PAIR[S[i]][S[j]];
  if (!t) return 0.0;
  int nstems = 0, unpaired = 0, fp = 0, fq = 0;
  for (int p = i + 1; p < j;) {
    if (pt[p] > p) { if (!nstems) { fp = p; fq = pt[p]; } nstems++; p = pt[p] + 1; }
    else { unpaired++; p++; }
  }
  if (nstems == 0) return X_Hairpin(P, j - i - 1, t, S[i + 1], S[j - 1], up + i - 1);
  if (nstems == 1) {
    int t2 = PAIR[S[fp]][S[fq]];
    if (!t2) return 0.0;
    return X_IntLoop(P, fp - i - 1, j - fq - 1, t, RTYPE[t2], S[i + 1], S[j - 1], S[fp - 1], S[fq + 1]) *
           weight_loop(P, S, pt, up, fp, fq);
  }
  double w = P->eMLclosing * X_MLstem(P, RTYPE[t], S[j - 1], S[i + 1]);
  for (int k = 0; k < unpaired; k++) w *= P->eMLbase;
  for (int p = i + 1; p < j;) {
    if (pt[p] > p) {
      int q = pt[p], t2 = PAIR[S[p]][S[q]];
      if (!t2) return 0.0;
      w *= X_MLstem(P, t2, S[p - 1], S[q + 1]) * weight_loop(P, S, pt, up, p, q);
      p = q + 1;
    } else p++;
  }
  return w;
}

double orc_boltzmann_weight(const orc_params *P, const char *seq, const char *db, int n) {
  int *S = encode_seq(seq, n);
  int *pt = (int *)calloc((size_t)n + 2, sizeof(int));
  int *stk = (int *)malloc(sizeof(int) * (size_t)(n + 1));
  char *up = (char *)malloc((size_t)n + 1);
  int sp = 0, bad = 0;
  for (int i = 0; i < n; i++) {
    char ch = seq[i];
    ch = (ch >= 'a' && ch <= 'z') ? (char)(ch - 32) : ch;
    up[i] = (ch == 'T') ? 'U' : ch;
  }
  up[n] = 0;
  for (int i = 1; i <= n && !bad; i++) {
    if (db[i - 1] == '(') stk[sp++] = i;
    else if (db[i - 1] == ')') {
      if (!sp) bad = 1;
      else { int o = stk[--sp]; pt[o] = i; pt[i] = o; }
    }
  }
  double w = (bad || sp) ? 0.0 : 1.0;
  for (int i = 1; i <= n && w != 0.0;) {
    if (pt[i] > i) {
      int j = pt[i], t = PAIR[S[i]][S[j]];
      w *= t ? X_ExtLoop(P, t, i > 1 ? S[i - 1] : -1, j < n ? S[j + 1] : -1) * weight_loop(P, S, pt, up, i, j) : 0.0;
      i = j + 1;
    } else i++;
  }
  free(S); free(pt); free(stk); free(up);
  return w;
}

/* ---------------------------------------------------------------- SimScore, reference utils/sim_score.py */

static int bracket_family(char ch, int *is_open) {
  static const char OP[] = "([<{ABCDE", CL[] = ")]>}abcde";
  for (int k = 0; k < 9; k++) {
    if (ch == OP[k]) { *is_open = 1; return k; }
    if (ch == CL[k]) { *is_open = 0; return k; }
  }
  return -1;
}

/* pairing_positions (sim_score.py:28-59): opens processed from the LAST one backwards, each
 * matched with the first still-free close of its family to its right. partner[i] = j, -1 for
 * '.'/'-', -2 = position absent from the dict (unmatched bracket / other char) */
static void pairing_positions(const char *s, int n, int *partner) {
  int *opens = (int *)malloc(sizeof(int) * (size_t)(n + 1));
  int *closes = (int *)malloc(sizeof(int) * (size_t)(n + 1));
  int *ofam = (int *)malloc(sizeof(int) * (size_t)(n + 1));
  int *cfam = (int *)malloc(sizeof(int) * (size_t)(n + 1));
  int no = 0, nc = 0;
  for (int i = 0; i < n; i++) {
    partner[i] = -2;
    int io, f = bracket_family(s[i], &io);
    if (f >= 0) {
      if (io) { opens[no] = i; ofam[no++] = f; }
      else { closes[nc] = i; cfam[nc++] = f; }
    }
    if (s[i] == '.' || s[i] == '-') partner[i] = -1;
  }
  /* the reference indexes l_closes[j] for j in range(len(l_opens)): it assumes no <= nc */
  for (int i = no - 1; i >= 0; i--) {
    int lim = MIN2(no, nc);
    for (int j = 0; j < lim; j++) {
      if (ofam[i] >= 0 && cfam[j] == ofam[i] && closes[j] > opens[i]) {
        partner[opens[i]] = closes[j];
        partner[closes[j]] = opens[i];
        ofam[i] = -1; cfam[j] = -3; /* "del l_opens[i][-1]; del l_closes[j][-1]" */
      }
    }
  }
  free(opens); free(closes); free(ofam); free(cfam);
}

static double py_round3(double x) {
  /* Python round(x, 3): correctly-rounded decimal repr; reproduce via printf's exact decimal conversion */
  char buf[64];
  snprintf(buf, sizeof buf, "%.3f", x);
  return strtod(buf, NULL);
}

void orc_simscore(const char *ref, const char *query, int n, double out[3], int conf[4]) {
  int *pr = (int *)malloc(sizeof(int) * (size_t)(n + 1));
  int *pq = (int *)malloc(sizeof(int) * (size_t)(n + 1));
  pairing_positions(ref, n, pr);
  pairing_positions(query, n, pq);
  long tp = 0, fp = 0, fn = 0, tn = 0;
  for (int i = 0; i < n; i++) {
    if (pr[i] == pq[i] && pr[i] != -1) tp++;
    if (pr[i] == pq[i] && pr[i] == -1) tn++;
    if (pr[i] != pq[i]) { if (pr[i] == -1) fp++; else fn++; }
  }
  double num, den;
  if (tp == 0 && fp == 0 && fn == 0 && tn != 0) { num = 1; den = 1; }
  else {
    num = (double)(tp * tn) - (double)(fp * fn);
    den = sqrt((double)((tp + fp) * (tp + fn) * (tn + fn) * (tn + fp)));
  }
  out[0] = py_round3(num / (den + 0.00001));
  out[1] = py_round3((double)tp / ((double)(tp + fn) + 0.001));
  out[2] = py_round3((double)tp / ((double)(tp + fp) + 0.001));
  if (conf) { conf[0] = (int)tp; conf[1] = (int)fp; conf[2] = (int)fn; conf[3] = (int)tn; }
  free(pr); free(pq);
}

/* ---------------------------------------------------------------- batch (cpu_baseline leg) */

void orc_score_batch(const orc_params *P, int R, int L, const char *seqs, int n_targets,
                     const char *targets, unsigned flags, int threads, double *Epf,
                     int32_t *Emfe, char *mfe_ss, int32_t *Ed) {
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#else
  (void)threads;
#endif
  /* flags & 8 (set by bench.py's cpu_baseline leg only): bind OpenMP thread t to the t-th CPU of the caller's affinity mask
   * for the duration of this call -- without it the threads of a share-limited lease pile onto a few cores and the all-cores
   * figure of the CPU baseline is not a statement about the cores (OMP_PROC_BIND cannot be used: set in the environment it
   * would also bind the caller's main thread, i.e. the Python process driving the GPU).  The masks are put back before the
   * call returns: the worker threads belong to the process-wide libgomp pool and other OpenMP users would inherit a pin. */
#if defined(_OPENMP) && defined(__linux__)
  cpu_set_t base;
  const int pin = (flags & 8u) && threads > 1 && sched_getaffinity(0, sizeof base, &base) == 0;   /* thread 0 = the caller, never bound */
  if (pin) {
    int ncpu = CPU_COUNT(&base);
#pragma omp parallel
    {
      int t = omp_get_thread_num() % (ncpu > 0 ? ncpu : 1), seen = 0;
      for (int c = 0; c < CPU_SETSIZE; c++)
        if (CPU_ISSET(c, &base) && seen++ == t) {
          cpu_set_t one; CPU_ZERO(&one); CPU_SET(c, &one);
          if (omp_get_thread_num() != 0) sched_setaffinity(0, sizeof one, &one);
          break;
        }
    }
  }
#endif
#pragma omp parallel for schedule(dynamic, 1)
  for (int r = 0; r < R; r++) {
    const char *seq = seqs + (size_t)r * (size_t)L;
    char *ss = (char *)malloc((size_t)L + 1);
    char *ss2 = (char *)malloc((size_t)L + 1);
    /* reference order: pf() first, then mfe() (energy_scores.py:150-151) */
    if ((flags & 1u) && Epf) Epf[r] = orc_pf(P, seq, L);
    if (flags & 2u) {
      int e = orc_mfe(P, seq, L, NULL, ss);
      if (Emfe) Emfe[r] = e;
      if (flags & 4u) { orc_pk_struct(P, seq, L, ss, ss2); memcpy(ss, ss2, (size_t)L); }
      if (mfe_ss) memcpy(mfe_ss + (size_t)r * (size_t)L, ss, (size_t)L);
    }
    if (Ed)
      for (int k = 0; k < n_targets; k++)
        Ed[(size_t)r * (size_t)n_targets + (size_t)k] = orc_eval_structure(P, seq, targets + (size_t)k * (size_t)L, L);
    free(ss); free(ss2);
  }
#if defined(_OPENMP) && defined(__linux__)
  if (pin) {
#pragma omp parallel
    { if (omp_get_thread_num() != 0) sched_setaffinity(0, sizeof base, &base); }
  }
#endif
}

/* ---------------------------------------------------------------- two strands: co-fold MFE and partition function
 *
 * fc.mfe_dimer() / fc.pf_dimer() of reference utils/energy_scores.py:154-158.  seq holds both strands
 * WITHOUT the '&'; cut = length of the first strand.  Restated from the published co-folding scheme (Bernhart et al. 2006:
 * concatenate the strands; a loop whose backbone contains the nick is an exterior loop; DuplexInit once for connected
 * structures; homodimer symmetry correction in the partition function) and PINNED on the two-strand trajectories the
 * reference committed (538 hetero-dimer + 170 homodimer rows: mfe_dimer strings and pf_dimer free energies).
 *
 * Differences from the one-strand recursions:
 *   - hairpins, multiloops and the unpaired stretches of interior loops must not contain the nick;
 *   - a pair (i,j) that joins the strands may close the "loop" that contains the nick: E_ExtLoop of the pair seen from
 *     inside + best exterior decomposition of [i+1..cut] + of [cut+1..j-1] (arrays fcA / fcB; qA3 / qB5 in the PF);
 *   - dangling neighbours count only inside a strand;
 *   - MFE = min(f5[n] + DuplexInit, fcA[1] + fcB[n]);  Q = (q5[n] - QA QB) expDuplexInit [/ 2 if both strands are equal]
 *     + QA QB.
 */
#define SAMESTR(a, b) (!((a) <= cut && (b) > cut))

typedef struct {
  int n, cut, *S, *c, *fML, *f5, *fcA, *fcB;
  unsigned char *pty;
  char *up;
} coctx;

static void co_fill(const orc_params *P, coctx *M, const char *seq, int n, int cut) {
  const int W = n + 2;
  size_t W2 = (size_t)W * (size_t)W;
  M->n = n; M->cut = cut;
  M->S = encode_seq(seq, n);
  M->up = (char *)malloc((size_t)n + 1);
  for (int i = 0; i < n; i++) {
    char ch = seq[i];
    ch = (ch >= 'a' && ch <= 'z') ? (char)(ch - 32) : ch;
    M->up[i] = (ch == 'T') ? 'U' : ch;
  }
  M->up[n] = 0;
  M->pty = (unsigned char *)calloc(W2, 1);
  M->c = (int *)malloc(W2 * sizeof(int));
  M->fML = (int *)malloc(W2 * sizeof(int));
  M->f5 = (int *)calloc((size_t)n + 2, sizeof(int));
  M->fcA = (int *)calloc((size_t)n + 3, sizeof(int));
  M->fcB = (int *)calloc((size_t)n + 3, sizeof(int));
  int *S = M->S, *c = M->c, *fML = M->fML, *fcA = M->fcA, *fcB = M->fcB;
  for (size_t k = 0; k < W2; k++) { c[k] = INF; fML[k] = INF; }
  for (int i = 1; i <= n; i++)
    for (int j = i + 1; j <= n; j++)
      if (j - i > TURN || !SAMESTR(i, j)) M->pty[IDX(i, j)] = (unsigned char)PAIR[S[i]][S[j]];
  for (int i = n; i >= 1; i--) {
    if (i == cut) {
      /* every pair inside the second strand is known: best exterior decomposition of [cut+1..j] */
      fcB[cut] = 0;
      for (int j = cut + 1; j <= n; j++) {
        int f = fcB[j - 1];
        for (int k = cut + 1; k < j; k++) {
          int t = M->pty[IDX(k, j)];
          if (!t || c[IDX(k, j)] >= INF) continue;
          int en = fcB[k - 1] + c[IDX(k, j)] + E_ExtLoop(P, t, k > cut + 1 ? S[k - 1] : -1, j < n ? S[j + 1] : -1);
          f = MIN2(f, en);
        }
        fcB[j] = f;
      }
    }
    for (int j = i + 1; j <= n; j++) {
      int t = M->pty[IDX(i, j)];
      int same = SAMESTR(i, j);
      int e = INF;
      if (t) {
        if (same) e = E_Hairpin(P, j - i - 1, t, S[i + 1], S[j - 1], M->up + i - 1);
        else {
          /* the nick lies on the backbone of the loop closed by (i,j) */
          int en = E_ExtLoop(P, RTYPE[t], SAMESTR(j - 1, j) ? S[j - 1] : -1, SAMESTR(i, i + 1) ? S[i + 1] : -1);
          e = MIN2(e, en + fcA[i + 1] + fcB[j - 1]);
        }
        /* multiloop closed by (i,j): its backbone must not contain the nick (a helix inside may enclose it) */
        if (SAMESTR(i, i + 1) && SAMESTR(j - 1, j)) {
          int dec = INF;
          for (int u = i + 2; u <= j - 2; u++) {
            if (!SAMESTR(u, u + 1)) continue;
            int a = fML[IDX(i + 1, u)], b = fML[IDX(u + 1, j - 1)];
            if (a < INF && b < INF) dec = MIN2(dec, a + b);
          }
          if (dec < INF) e = MIN2(e, dec + P->MLclosing + E_MLstem(P, RTYPE[t], S[j - 1], S[i + 1]));
        }
        /* interior loops: both unpaired stretches inside one strand */
        for (int p = i + 1; p <= MIN2(j - 2, i + MAXLOOP + 1); p++) {
          if (!SAMESTR(i, p)) break;
          int u1 = p - i - 1;
          for (int q = j - 1; q > p; q--) {
            int u2 = j - q - 1;
            if (u1 + u2 > MAXLOOP) break;
            if (!SAMESTR(q, j)) break;
            int t2 = M->pty[IDX(p, q)];
            if (!t2 || c[IDX(p, q)] >= INF) continue;
            int en = c[IDX(p, q)] + E_IntLoop(P, u1, u2, t, RTYPE[t2], S[i + 1], S[j - 1], S[p - 1], S[q + 1]);
            e = MIN2(e, en);
          }
        }
      }
      c[IDX(i, j)] = e;
      {
        int f = INF;
        if (SAMESTR(i, i + 1) && fML[IDX(i + 1, j)] < INF) f = fML[IDX(i + 1, j)] + P->MLbase;
        if (SAMESTR(j - 1, j) && fML[IDX(i, j - 1)] < INF) f = MIN2(f, fML[IDX(i, j - 1)] + P->MLbase);
        if (e < INF) f = MIN2(f, e + E_MLstem(P, t, (i > 1 && SAMESTR(i - 1, i)) ? S[i - 1] : -1, (j < n && SAMESTR(j, j + 1)) ? S[j + 1] : -1));
        for (int u = i + 1; u <= j - 2; u++) {
          if (!SAMESTR(u, u + 1)) continue;
          int a = fML[IDX(i, u)], b = fML[IDX(u + 1, j)];
          if (a < INF && b < INF) f = MIN2(f, a + b);
        }
        fML[IDX(i, j)] = f;
      }
    }
    if (i <= cut) {
      /* best exterior decomposition of [i..cut] */
      int f = fcA[i + 1];          /* fcA[cut+1] = 0 */
      for (int k = i + 1; k <= cut; k++) {
        int t = M->pty[IDX(i, k)];
        if (!t || c[IDX(i, k)] >= INF) continue;
        int en = c[IDX(i, k)] + E_ExtLoop(P, t, i > 1 ? S[i - 1] : -1, k < cut ? S[k + 1] : -1) + fcA[k + 1];
        f = MIN2(f, en);
      }
      fcA[i] = f;
    }
  }
  int *f5 = M->f5;
  f5[0] = 0;
  for (int j = 1; j <= n; j++) {
    int f = f5[j - 1];
    for (int i = j - 1; i >= 1; i--) {
      int t = M->pty[IDX(i, j)];
      if (!t || c[IDX(i, j)] >= INF) continue;
      int en = f5[i - 1] + c[IDX(i, j)] +
               E_ExtLoop(P, t, (i > 1 && SAMESTR(i - 1, i)) ? S[i - 1] : -1, (j < n && SAMESTR(j, j + 1)) ? S[j + 1] : -1);
      f = MIN2(f, en);
    }
    f5[j] = f;
  }
}

static void co_free(coctx *M) {
  free(M->S); free(M->up); free(M->pty); free(M->c); free(M->fML); free(M->f5); free(M->fcA); free(M->fcB);
}

int co_bt_order = 1;   /* candidate order of the pair traceback (see co_traceback); settled on the goldens */

/* sectors: 0 = f5[1..j], 1 = fML[i..j], 2 = pair (i,j), 3 = fcA[i..cut], 4 = fcB[cut+1..j] */
static void co_traceback(const orc_params *P, const coctx *M, char *ss, int dimer) {
  const int n = M->n, W = n + 2, cut = M->cut;
  const int *S = M->S, *c = M->c, *fML = M->fML, *f5 = M->f5, *fcA = M->fcA, *fcB = M->fcB;
  typedef struct { int i, j, ml; } sect;
  sect *st = (sect *)malloc(sizeof(sect) * (size_t)(4 * n + 8));
  int sp = 0;
  memset(ss, '.', (size_t)n);
  ss[n] = 0;
  if (dimer) st[sp++] = (sect){1, n, 0};
  else { st[sp++] = (sect){1, cut, 3}; st[sp++] = (sect){cut + 1, n, 4}; }
  while (sp > 0) {
    sect s = st[--sp];
    int i = s.i, j = s.j;
    if (s.ml == 0) {
      int jj = j;
      while (jj > 0 && f5[jj] == f5[jj - 1]) jj--;
      if (jj < 2) continue;
      int found = 0;
      for (int u = jj - 1; u >= 1; u--) {
        int t = M->pty[IDX(u, jj)];
        if (!t || c[IDX(u, jj)] >= INF) continue;
        int en = c[IDX(u, jj)] + E_ExtLoop(P, t, (u > 1 && SAMESTR(u - 1, u)) ? S[u - 1] : -1,
                                            (jj < n && SAMESTR(jj, jj + 1)) ? S[jj + 1] : -1);
        if (f5[jj] == en + f5[u - 1]) { st[sp++] = (sect){1, u - 1, 0}; st[sp++] = (sect){u, jj, 2}; found = 1; break; }
      }
      if (!found) { fprintf(stderr, "oracle cofold: f5 traceback failed at %d\n", jj); break; }
      continue;
    }
    if (s.ml == 3) {                     /* fcA[i]: [i..cut] */
      while (i <= cut && fcA[i] == fcA[i + 1]) i++;
      if (i > cut) continue;
      int found = 0;
      for (int k = i + 1; k <= cut; k++) {
        int t = M->pty[IDX(i, k)];
        if (!t || c[IDX(i, k)] >= INF) continue;
        int en = c[IDX(i, k)] + E_ExtLoop(P, t, i > 1 ? S[i - 1] : -1, k < cut ? S[k + 1] : -1);
        if (fcA[i] == en + fcA[k + 1]) { st[sp++] = (sect){k + 1, cut, 3}; st[sp++] = (sect){i, k, 2}; found = 1; break; }
      }
      if (!found) { fprintf(stderr, "oracle cofold: fcA traceback failed at %d\n", i); break; }
      continue;
    }
    if (s.ml == 4) {                     /* fcB[j]: [cut+1..j] */
      while (j > cut && fcB[j] == fcB[j - 1]) j--;
      if (j <= cut) continue;
      int found = 0;
      for (int k = j - 1; k > cut; k--) {
        int t = M->pty[IDX(k, j)];
        if (!t || c[IDX(k, j)] >= INF) continue;
        int en = c[IDX(k, j)] + E_ExtLoop(P, t, k > cut + 1 ? S[k - 1] : -1, j < n ? S[j + 1] : -1);
        if (fcB[j] == en + fcB[k - 1]) { st[sp++] = (sect){cut + 1, k - 1, 4}; st[sp++] = (sect){k, j, 2}; found = 1; break; }
      }
      if (!found) { fprintf(stderr, "oracle cofold: fcB traceback failed at %d\n", j); break; }
      continue;
    }
    if (s.ml == 1) {
      while (j > i && SAMESTR(j - 1, j) && fML[IDX(i, j - 1)] < INF && fML[IDX(i, j)] == fML[IDX(i, j - 1)] + P->MLbase) j--;
      while (i < j && SAMESTR(i, i + 1) && fML[IDX(i + 1, j)] < INF && fML[IDX(i, j)] == fML[IDX(i + 1, j)] + P->MLbase) i++;
      int fij = fML[IDX(i, j)];
      int t = M->pty[IDX(i, j)];
      if (t && c[IDX(i, j)] < INF &&
          fij == c[IDX(i, j)] + E_MLstem(P, t, (i > 1 && SAMESTR(i - 1, i)) ? S[i - 1] : -1, (j < n && SAMESTR(j, j + 1)) ? S[j + 1] : -1)) {
        st[sp++] = (sect){i, j, 2};
        continue;
      }
      int found = 0;
      for (int u = i + 1; u <= j - 2; u++)
        if (SAMESTR(u, u + 1) && fML[IDX(i, u)] < INF && fML[IDX(u + 1, j)] < INF && fij == fML[IDX(i, u)] + fML[IDX(u + 1, j)]) {
          st[sp++] = (sect){i, u, 1}; st[sp++] = (sect){u + 1, j, 1}; found = 1; break;
        }
      if (!found) { fprintf(stderr, "oracle cofold: fML traceback failed at %d,%d\n", i, j); break; }
      continue;
    }
    for (;;) {
      ss[i - 1] = '(';
      ss[j - 1] = ')';
      int t = M->pty[IDX(i, j)];
      int cij = c[IDX(i, j)];
      if (SAMESTR(i, j) && cij == E_Hairpin(P, j - i - 1, t, S[i + 1], S[j - 1], M->up + i - 1)) break;
      int found = 0;
      for (int pass = 0; pass < 3 && !found; pass++) {
        const int what = co_bt_order == 0 ? (pass == 0 ? 'N' : pass == 1 ? 'I' : 'M')
                       : co_bt_order == 1 ? (pass == 0 ? 'I' : pass == 1 ? 'N' : 'M')
                                          : (pass == 0 ? 'I' : pass == 1 ? 'M' : 'N');
        if (what == 'N') {
          if (SAMESTR(i, j)) continue;
          int en = E_ExtLoop(P, RTYPE[t], SAMESTR(j - 1, j) ? S[j - 1] : -1, SAMESTR(i, i + 1) ? S[i + 1] : -1);
          if (cij == en + fcA[i + 1] + fcB[j - 1]) {
            st[sp++] = (sect){i + 1, cut, 3};
            st[sp++] = (sect){cut + 1, j - 1, 4};
            found = 2;
          }
        } else if (what == 'I') {
          for (int p = i + 1; p <= MIN2(j - 2, i + MAXLOOP + 1) && !found; p++) {
            if (!SAMESTR(i, p)) break;
            for (int q = j - 1; q > p; q--) {
              if (p - i - 1 + j - q - 1 > MAXLOOP) break;
              if (!SAMESTR(q, j)) break;
              int t2 = M->pty[IDX(p, q)];
              if (!t2 || c[IDX(p, q)] >= INF) continue;
              int en = E_IntLoop(P, p - i - 1, j - q - 1, t, RTYPE[t2], S[i + 1], S[j - 1], S[p - 1], S[q + 1]);
              if (cij == en + c[IDX(p, q)]) { i = p; j = q; found = 1; break; }
            }
          }
        } else {
          if (!(SAMESTR(i, i + 1) && SAMESTR(j - 1, j))) continue;
          int e = cij - P->MLclosing - E_MLstem(P, RTYPE[t], S[j - 1], S[i + 1]);
          for (int u = i + 2; u <= j - 2; u++)
            if (SAMESTR(u, u + 1) && fML[IDX(i + 1, u)] < INF && fML[IDX(u + 1, j - 1)] < INF &&
                e == fML[IDX(i + 1, u)] + fML[IDX(u + 1, j - 1)]) {
              st[sp++] = (sect){i + 1, u, 1}; st[sp++] = (sect){u + 1, j - 1, 1}; found = 2; break;
            }
        }
      }
      if (found == 1) continue;
      if (!found) fprintf(stderr, "oracle cofold: pair traceback failed at %d,%d\n", i, j);
      break;
    }
  }
  free(st);
}

int orc_cofold_mfe(const orc_params *P, const char *seq, int n, int cut, char *ss) {
  coctx M;
  co_fill(P, &M, seq, n, cut);
  const int e_dimer = M.f5[n] + P->DuplexInit, e_mono = M.fcA[1] + M.fcB[n];
  const int dimer = e_dimer < e_mono;
  if (ss) co_traceback(P, &M, ss, dimer);
  co_free(&M);
  return dimer ? e_dimer : e_mono;
}

/* out[0] = FA, out[1] = FB, out[2] = FcAB (true hybrids, DuplexInit and symmetry applied), out[3] = FAB */
void orc_cofold_pf(const orc_params *P, const char *seq, int n, int cut, double out[4]) {
  const int W = n + 2;
  size_t W2 = (size_t)W * (size_t)W;
  int *S = encode_seq(seq, n);
  char *up = (char *)malloc((size_t)n + 1);
  for (int i = 0; i < n; i++) {
    char ch = seq[i];
    ch = (ch >= 'a' && ch <= 'z') ? (char)(ch - 32) : ch;
    up[i] = (ch == 'T') ? 'U' : ch;
  }
  up[n] = 0;
  double *qb = (double *)calloc(W2, sizeof(double)), *qm = (double *)calloc(W2, sizeof(double)),
         *qm1 = (double *)calloc(W2, sizeof(double));
  double *q5 = (double *)calloc((size_t)n + 2, sizeof(double)), *qA3 = (double *)calloc((size_t)n + 3, sizeof(double)),
         *qB5 = (double *)calloc((size_t)n + 3, sizeof(double));
  double *scale = (double *)malloc(sizeof(double) * (size_t)(n + 3)), *eMLb = (double *)malloc(sizeof(double) * (size_t)(n + 3));
  scale[0] = 1.0; eMLb[0] = 1.0;
  for (int k = 1; k <= n + 2; k++) { scale[k] = scale[k - 1] / P->pf_scale; eMLb[k] = eMLb[k - 1] * P->eMLbase / P->pf_scale; }
#define PT(i, j) (((j) - (i) > TURN || !SAMESTR(i, j)) ? PAIR[S[i]][S[j]] : 0)
  for (int i = n; i >= 1; i--) {
    if (i == cut) {
      qB5[cut] = 1.0;
      for (int j = cut + 1; j <= n; j++) {
        double q = qB5[j - 1] * scale[1];
        for (int k = cut + 1; k < j; k++) {
          int t = PT(k, j);
          if (!t) continue;
          q += qB5[k - 1] * qb[IDX(k, j)] * X_ExtLoop(P, t, k > cut + 1 ? S[k - 1] : -1, j < n ? S[j + 1] : -1);
        }
        qB5[j] = q;
      }
    }
    for (int j = i + 1; j <= n; j++) {
      int t = PT(i, j);
      int same = SAMESTR(i, j);
      double b = 0.0;
      if (t) {
        if (same) {
          int u = j - i - 1;
          b = X_Hairpin(P, u, t, S[i + 1], S[j - 1], up + i - 1) * scale[u + 2];
        } else {
          b += qA3[i + 1] * qB5[j - 1] * scale[2] *
               X_ExtLoop(P, RTYPE[t], SAMESTR(j - 1, j) ? S[j - 1] : -1, SAMESTR(i, i + 1) ? S[i + 1] : -1);
        }
        if (SAMESTR(i, i + 1) && SAMESTR(j - 1, j)) {
          /* multiloop: the backbone stays inside a strand, a helix inside may enclose the nick */
          double tmp = 0.0;
          for (int k = i + 3; k <= j - 2; k++)
            if (SAMESTR(k - 1, k)) tmp += qm[IDX(i + 1, k - 1)] * qm1[IDX(k, j - 1)];
          b += tmp * P->eMLclosing * X_MLstem(P, RTYPE[t], S[j - 1], S[i + 1]) * scale[2];
        }
        for (int p = i + 1; p <= MIN2(j - 2, i + MAXLOOP + 1); p++) {
          if (!SAMESTR(i, p)) break;
          int u1 = p - i - 1;
          for (int q = j - 1; q > p; q--) {
            int u2 = j - q - 1;
            if (u1 + u2 > MAXLOOP) break;
            if (!SAMESTR(q, j)) break;
            int t2 = PT(p, q);
            if (!t2) continue;
            b += qb[IDX(p, q)] * X_IntLoop(P, u1, u2, t, RTYPE[t2], S[i + 1], S[j - 1], S[p - 1], S[q + 1]) * scale[u1 + u2 + 2];
          }
        }
      }
      qb[IDX(i, j)] = b;
      {
        double m1 = SAMESTR(j - 1, j) ? qm1[IDX(i, j - 1)] * eMLb[1] : 0.0;
        if (t) m1 += b * X_MLstem(P, t, (i > 1 && SAMESTR(i - 1, i)) ? S[i - 1] : -1, (j < n && SAMESTR(j, j + 1)) ? S[j + 1] : -1);
        qm1[IDX(i, j)] = m1;
        double m = m1;
        for (int k = i + 1; k <= j - 1; k++) {
          double left = SAMESTR(k - 1, k) ? qm[IDX(i, k - 1)] : 0.0;     /* qm[i,k-1] then a stem at k: k-1, k adjacent */
          if (SAMESTR(i, k)) left += eMLb[k - i];                        /* i..k-1 unpaired, stem at k */
          m += left * qm1[IDX(k, j)];
        }
        qm[IDX(i, j)] = m;
      }
    }
    if (i <= cut) {
      double q = (i + 1 <= cut ? qA3[i + 1] : 1.0) * scale[1];
      for (int k = i + 1; k <= cut; k++) {
        int t = PT(i, k);
        if (!t) continue;
        q += qb[IDX(i, k)] * X_ExtLoop(P, t, i > 1 ? S[i - 1] : -1, k < cut ? S[k + 1] : -1) * (k + 1 <= cut ? qA3[k + 1] : 1.0);
      }
      qA3[i] = q;
    }
    if (i == cut + 1) qA3[cut + 1] = 1.0;
  }
  q5[0] = 1.0;
  for (int j = 1; j <= n; j++) {
    double q = q5[j - 1] * scale[1];
    for (int i = j - 1; i >= 1; i--) {
      int t = PT(i, j);
      if (!t) continue;
      q += q5[i - 1] * qb[IDX(i, j)] *
           X_ExtLoop(P, t, (i > 1 && SAMESTR(i - 1, i)) ? S[i - 1] : -1, (j < n && SAMESTR(j, j + 1)) ? S[j + 1] : -1);
    }
    q5[j] = q;
  }
#undef PT
  const double kT = P->kT / 1000.0, lsc = log(P->pf_scale);
  const double QA = qA3[1], QB = qB5[n], Q0 = q5[n];
  double QAB = (Q0 - QA * QB) * exp(-(double)P->DuplexInit * 10.0 / P->kT);
  if (n == 2 * cut && strncmp(up, up + cut, (size_t)cut) == 0) QAB /= 2.0;   /* rotational symmetry of a homodimer */
  const double Qtot = QA * QB + QAB;
  out[0] = -kT * (log(QA) + cut * lsc);
  out[1] = -kT * (log(QB) + (n - cut) * lsc);
  out[2] = QAB > 1e-17 ? -kT * (log(QAB) + n * lsc) : 999.0;
  out[3] = -kT * (log(Qtot) + n * lsc);
  free(S); free(up); free(qb); free(qm); free(qm1); free(q5); free(qA3); free(qB5); free(scale); free(eMLb);
}

/* ---------------------------------------------------------------- second-best structure
 *
 * get_first_suboptimal_structure_and_energy(seq, fc, 1)[1] of reference utils/energy_scores.py:453-488: ViennaRNA's
 * subopt (Wuchty et al. 1999, uniq_ML = 1: every structure once) is run with a growing energy band (1, 2, ... 49 kcal/mol)
 * until it returns at least two structures; they are sorted by energy and the second one is taken.  What the caller
 * uses is that structure's ENERGY = the lowest energy over all structures other than one minimum-energy structure
 * (equal to the MFE if the ground state is degenerate); 0.0 if no second structure lies within 49 kcal/mol.
 *
 * Restated as a two-best dynamic programme over an UNAMBIGUOUS decomposition (each structure has exactly one
 * derivation, so the two smallest values of a table entry belong to two different structures):
 *   F[j]    = { F[j-1] ; F[i-1] + C[i,j] + ext(i,j) }
 *   C[i,j]  = { hairpin ; C[p,q] + interior ; M2[i+1,j-1] + closing }
 *   M[i,j]  (>= 1 stem) = { M[i,j-1] + b ; (k-i) b + C[k,j] + stem ; M[i,k-1] + C[k,j] + stem }
 *   M2[i,j] (>= 2 stems) = { M2[i,j-1] + b ; M[i,k-1] + C[k,j] + stem }
 * NOT pinned by the reference (no golden for -nd on); checked against exhaustive enumeration on short sequences. */
typedef struct { int a, b; } top2;
static inline top2 t2_new(void) { top2 t = {INF, INF}; return t; }
static inline void t2_add(top2 *t, int v) {
  if (v >= INF / 2) return;
  if (v < t->a) { t->b = t->a; t->a = v; }
  else if (v < t->b) t->b = v;
}
static inline void t2_add_sum(top2 *t, top2 x, int e) {       /* x + e */
  if (x.a < INF / 2) t2_add(t, x.a + e);
  if (x.b < INF / 2) t2_add(t, x.b + e);
}
static inline void t2_add_sum2(top2 *t, top2 x, top2 y, int e) { /* x + y + e: the three best combinations */
  if (x.a >= INF / 2 || y.a >= INF / 2) return;
  t2_add(t, x.a + y.a + e);
  if (y.b < INF / 2) t2_add(t, x.a + y.b + e);
  if (x.b < INF / 2) t2_add(t, x.b + y.a + e);
}

/* returns 1 and the two lowest energies (dcal/mol) in e[0] <= e[1]; e[1] = INF when there is only one structure */
int orc_two_best(const orc_params *P, const char *seq, int n, int e[2]) {
  const int W = n + 2;
  size_t W2 = (size_t)W * (size_t)W;
  int *S = encode_seq(seq, n);
  char *up = (char *)malloc((size_t)n + 1);
  for (int i = 0; i < n; i++) {
    char ch = seq[i];
    ch = (ch >= 'a' && ch <= 'z') ? (char)(ch - 32) : ch;
    up[i] = (ch == 'T') ? 'U' : ch;
  }
  up[n] = 0;
  top2 *C = (top2 *)malloc(W2 * sizeof(top2)), *M = (top2 *)malloc(W2 * sizeof(top2)), *M2 = (top2 *)malloc(W2 * sizeof(top2));
  top2 *F = (top2 *)malloc(sizeof(top2) * (size_t)(n + 2));
  for (size_t k = 0; k < W2; k++) { C[k] = t2_new(); M[k] = t2_new(); M2[k] = t2_new(); }
  for (int d = TURN + 1; d < n; d++)
    for (int i = 1; i + d <= n; i++) {
      int j = i + d;
      int t = PAIR[S[i]][S[j]];
      top2 c = t2_new();
      if (t) {
        t2_add(&c, E_Hairpin(P, j - i - 1, t, S[i + 1], S[j - 1], up + i - 1));
        for (int p = i + 1; p <= MIN2(j - 2 - TURN, i + MAXLOOP + 1); p++)
          for (int q = j - 1; q >= p + TURN + 1; q--) {
            if (p - i - 1 + j - q - 1 > MAXLOOP) break;
            int t2 = PAIR[S[p]][S[q]];
            if (!t2) continue;
            t2_add_sum(&c, C[IDX(p, q)], E_IntLoop(P, p - i - 1, j - q - 1, t, RTYPE[t2], S[i + 1], S[j - 1], S[p - 1], S[q + 1]));
          }
        t2_add_sum(&c, M2[IDX(i + 1, j - 1)], P->MLclosing + E_MLstem(P, RTYPE[t], S[j - 1], S[i + 1]));
      }
      C[IDX(i, j)] = c;
      top2 m = t2_new(), m2 = t2_new();
      t2_add_sum(&m, M[IDX(i, j - 1)], P->MLbase);
      t2_add_sum(&m2, M2[IDX(i, j - 1)], P->MLbase);
      for (int k = i; k <= j - TURN - 1; k++) {
        int tk = PAIR[S[k]][S[j]];
        if (!tk) continue;
        top2 ck = C[IDX(k, j)];
        if (ck.a >= INF / 2) continue;
        int st = E_MLstem(P, tk, S[k - 1], S[j + 1]);
        t2_add_sum(&m, ck, (k - i) * P->MLbase + st);
        if (k - 1 >= i) {
          t2_add_sum2(&m, M[IDX(i, k - 1)], ck, st);
          t2_add_sum2(&m2, M[IDX(i, k - 1)], ck, st);
        }
      }
      M[IDX(i, j)] = m;
      M2[IDX(i, j)] = m2;
    }
  F[0] = t2_new(); F[0].a = 0;
  for (int j = 1; j <= n; j++) {
    top2 f = t2_new();
    t2_add_sum(&f, F[j - 1], 0);
    for (int i = j - TURN - 1; i >= 1; i--) {
      int t = PAIR[S[i]][S[j]];
      if (!t || C[IDX(i, j)].a >= INF / 2) continue;
      t2_add_sum2(&f, F[i - 1], C[IDX(i, j)], E_ExtLoop(P, t, i > 1 ? S[i - 1] : -1, j < n ? S[j + 1] : -1));
    }
    F[j] = f;
  }
  e[0] = F[n].a; e[1] = F[n].b;
  free(S); free(up); free(C); free(M); free(M2); free(F);
  return 1;
}

/* the reference's number: energy (dcal/mol) of subopt_list[1]; 0 when no second structure lies within 4900 dcal/mol */
int orc_subopt_energy(const orc_params *P, const char *seq, int n) {
  int e[2];
  orc_two_best(P, seq, n, e);
  if (e[1] >= INF / 2 || e[1] - e[0] > 4900) return 0;
  return e[1];
}
