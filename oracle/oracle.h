/*
 * oracle.h -- CPU restatement of the thermodynamic scoring path of fryzjergda/DesiRNA.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (desirna_amd/, include/) may include,
 * link or call this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * What it restates: the ViennaRNA 2.6.4 calls the reference makes on every scored sequence
 * (reference utils/energy_scores.py:128-159 get_mfe_e_ss, :75 eval_structure, :350-374
 * get_MFE / get_ensemble_defect; utils/sequence_utils.py:1166-1228 get_pk_struct;
 * utils/sim_score.py:28-147 SimScore).  ViennaRNA itself (PyPI viennarna==2.6.4, pinned in the
 * reference's DesiRNA-env.yml:101) is not in the reference tree and is not installable here, so
 * the recursions are restated from its published algorithm (Zuker MFE, McCaskill partition
 * function, Turner nearest-neighbour loop model) and PINNED against the
 * numbers the reference committed: example_files/outputs/ (trajectory CSVs: Epf, E(target), MFE
 * structures, pk-annotated structures, 1-MCC/recall/precision) and
 * eterna_benchmark/Eterna100V1_benchmark_results (MFE(sequence) == structure).  See
 * tests/golden/ and tests/test_oracle_golden.py.  Outside recursion / ensemble defect has no
 * golden vector in the reference ("parity unpinned" for that one quantity); it is verified against
 * an explicit Boltzmann-weighted sum over all structures of short sequences instead.
 *
 * Model in force (reference energy_scores.py:27-28 sets only compute_bpp=0): 37 degC, dangles=2,
 * noLP=0, noGU=0, special hairpins on, TURN=3, MAXLOOP=30, linear, no G-quadruplexes,
 * pf_smooth=1, energies are int in dcal/mol (0.01 kcal/mol), INF = 10000000.
 */
#ifndef DRNA_ORACLE_H
#define DRNA_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_params orc_params;

/* blob: layout documented in desirna_amd/params.py */
orc_params *orc_params_create(const int32_t *blob, int n_int32);
void orc_params_destroy(orc_params *P);

/* RNA.fold_compound(seq).eval_structure(db): only '(' ')' are pairs; returns dcal/mol */
int orc_eval_structure(const orc_params *P, const char *seq, const char *db, int n);

/* two strands: seq/db WITHOUT the '&'; cut = length of first strand (0 = single strand) */
int orc_eval_structure_cut(const orc_params *P, const char *seq, const char *db, int n, int cut);

/* fc.mfe(): fill + traceback.  nopair (may be NULL): nopair[k]!=0 -> position k (0-based) must stay
 * unpaired (hard constraint 'x').  ss receives n chars + NUL.  Returns MFE in dcal/mol. */
int orc_mfe(const orc_params *P, const char *seq, int n, const unsigned char *nopair, char *ss);

/* fc.pf()[1] with compute_bpp=0: ensemble free energy in kcal/mol */
double orc_pf(const orc_params *P, const char *seq, int n);

/* get_pk_struct(): greedy pseudoknot annotation by up to three constrained re-folds */
void orc_pk_struct(const orc_params *P, const char *seq, int n, const char *ss_nopk, char *ss_pk);

/* fc.ensemble_defect(target) after mfe/rescale/pf with bpp on; bpp (may be NULL) gets the
 * (n+1)*(n+1) base-pair probability matrix, 1-based, upper triangle */
double orc_ensemble_defect(const orc_params *P, const char *seq, int n, const char *target, double *bpp);

/* exp(-E/kT) of ONE structure under the partition function's loop model (pf_smooth dangles, no pf_scale); a structure
 * walk independent of the DP, used by the tests to check Z, P(i,j) and the defect against explicit enumeration */
double orc_boltzmann_weight(const orc_params *P, const char *seq, const char *db, int n);

/* SimScore(ref, query): out[0]=mcc, out[1]=recall, out[2]=precision, each round(x,3) (NOT 1-x);
 * conf (may be NULL) gets tp, fp, fn, tn */
void orc_simscore(const char *ref, const char *query, int n, double out[3], int conf[4]);

/* One replica-fold per sequence (pf + mfe [+pk] + eval targets), OpenMP over sequences.
 * seqs: R*L chars; targets: n_targets*L chars (target first, then alt structures).
 * flags: bit0 PF, bit1 MFE, bit2 PK.  Outputs may be NULL. */
void orc_score_batch(const orc_params *P, int R, int L, const char *seqs, int n_targets,
                     const char *targets, unsigned flags, int threads, double *Epf,
                     int32_t *Emfe, char *mfe_ss, int32_t *Ed);

/* two strands (seq WITHOUT '&', cut = length of the first strand): fc.mfe_dimer() -> structure (n chars + NUL, no '&')
 * and energy (dcal/mol); fc.pf_dimer() -> out[0..3] = FA, FB, FcAB, FAB (kcal/mol; the reference uses [-1] = FAB) */
int orc_cofold_mfe(const orc_params *P, const char *seq, int n, int cut, char *ss);
void orc_cofold_pf(const orc_params *P, const char *seq, int n, int cut, double out[4]);

/* the two lowest structure energies (dcal/mol; e[1] = 10000000 if there is one structure only) and the number the reference
 * takes from ViennaRNA's subopt for -nd on: energy of the second entry of the sorted list (0 beyond 49 kcal/mol) */
int orc_two_best(const orc_params *P, const char *seq, int n, int e[2]);
int orc_subopt_energy(const orc_params *P, const char *seq, int n);

/* table dumps for kernel-level parity tests: c / fML as (n+2)*(n+2) row-major int32 */
int orc_mfe_tables(const orc_params *P, const char *seq, int n, const unsigned char *nopair,
                   int32_t *c, int32_t *fML, int32_t *f5);

#ifdef __cplusplus
}
#endif
#endif
