/*
 * desirna_amd.h -- C ABI of the MI355X (gfx950) replica-scoring engine.
 *
 * This is the drop-in boundary for the one hot path of fryzjergda/DesiRNA: scoring every mutated
 * sequence of every replica.  The reference has no FFI of its own; its de-facto operator boundary
 * is the set of ViennaRNA SWIG calls made per sequence by utils/energy_scores.py.
 * Each entry point below names the reference call sites it replaces (paths relative to the
 * reference tree).  Plain C types only; every buffer is owned by the caller; a handle is bound to
 * one GPU and must be used from one host thread at a time.
 *
 * Units: integer energies are dcal/mol (0.01 kcal/mol) exactly as ViennaRNA keeps them internally
 * (the Python shim divides by 100 to reproduce ViennaRNA's float returns); Epf is kcal/mol.
 * Sequences are upper- or lower-case A C G U (T is read as U); structures are dot-bracket strings
 * in which only '(' and ')' denote pairs for energy evaluation.
 */
#ifndef DESIRNA_AMD_H
#define DESIRNA_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct drna_engine drna_engine;

/* Layout version of this interface.  Bumped whenever the meaning or size of a caller-owned buffer changes under an unchanged
 * symbol name (2 -> 3: rng_state of drna_propose_batch[_alt] / drna_metropolis_batch / drna_mc_run became R x DRNA_RNG_WORDS
 * uint32 MT19937 streams instead of R uint64).  A binding compares drna_abi_version() with the DRNA_ABI_VERSION it was written
 * against at load time, so a stale caller fails there instead of overrunning a buffer. */
#define DRNA_ABI_VERSION 3
int drna_abi_version(void);

/* return codes */
enum {
  DRNA_OK = 0,
  DRNA_ERR_ARG = -1,        /* null pointer, R/L out of range, bad flags                    */
  DRNA_ERR_PARAMS = -2,     /* malformed parameter blob                                     */
  DRNA_ERR_DEVICE = -3,     /* HIP error (no device, out of memory, launch failure)         */
  DRNA_ERR_SEQUENCE = -4,   /* a sequence holds a character other than A C G U T            */
  DRNA_ERR_STRUCTURE = -5,  /* unbalanced '(' ')' in a target structure                     */
  DRNA_ERR_PF_RANGE = -6,   /* partition function left the fp64 range despite scaling       */
  DRNA_ERR_INTERNAL = -7    /* traceback could not reproduce a table value (engine bug)     */
};

/* what drna_score_batch computes */
enum {
  DRNA_NEED_PF = 1u,          /* Epf            : fc.pf()            energy_scores.py:150           */
  DRNA_NEED_MFE = 2u,         /* mfe_ss + Emfe  : fc.mfe()           energy_scores.py:151, :354     */
  DRNA_NEED_PK = 4u,          /* pk annotation  : get_pk_struct()    sequence_utils.py:1166-1228    */
  DRNA_NEED_EVAL = 8u         /* Ed             : fc.eval_structure  energy_scores.py:75, :99       */
};

/*
 * Create an engine on HIP device `device`.
 * Replaces RNA.params_load(path) (DesiRNA.py:455-456) and the per-call table allocation of
 * RNA.fold_compound(seq, md) (energy_scores.py:147): parameters are expanded once into device
 * tables, workspaces are sized once for max_R sequences of at most max_L nucleotides.
 * `params` is the int32 blob produced by desirna_amd/params.py (layout documented there).
 */
int drna_create(const int32_t *params, int n_int32, int device, int max_R, int max_L, drna_engine **out);

void drna_destroy(drna_engine *e);

/* message for the last non-OK return on this engine (or for a failed drna_create when e == NULL) */
const char *drna_last_error(const drna_engine *e);

/*
 * Install the target structure(s) every sequence is evaluated against: targets[0] is the design
 * target (input file >sec_struct, '&' removed), targets[1..] the >alt_sec_struct lines.
 * Replaces the argument of fc.eval_structure(...) at energy_scores.py:75 and :99.
 * `targets` holds n_targets strings of exactly L characters, back to back (no terminators).
 */
int drna_set_targets(drna_engine *e, int n_targets, int L, const char *targets);

/*
 * Score R sequences of length L (host buffers).  Replaces, for all R replicas at once, the
 * ViennaRNA calls of score_sequence(): get_mfe_e_ss (energy_scores.py:128-159), eval_structure
 * (:75,:99), RNA.fold(seq)[1] (:354) and get_pk_struct (sequence_utils.py:1166-1228).
 *   seqs    R*L chars                                   in
 *   flags   OR of DRNA_NEED_*                           in
 *   Epf     R doubles, kcal/mol                         out (NEED_PF)      may be NULL otherwise
 *   Emfe    R int32, dcal/mol                           out (NEED_MFE)
 *   mfe_ss  R*L chars, no terminators                   out (NEED_MFE; pk-annotated with NEED_PK)
 *   Ed      R*n_targets int32, dcal/mol                 out (NEED_EVAL; needs drna_set_targets)
 */
int drna_score_batch(drna_engine *e, int R, int L, const char *seqs, uint32_t flags,
                     double *Epf, int32_t *Emfe, char *mfe_ss, int32_t *Ed);

/*
 * Same with every buffer already resident in device memory (e.g. torch tensors on the engine's
 * device).  Work is enqueued on the engine's streams and the call returns after they have drained.
 * status (device, R int32, may be NULL) receives per-sequence status words.
 */
int drna_score_batch_device(drna_engine *e, int R, int L, const char *d_seqs, uint32_t flags,
                            double *d_Epf, int32_t *d_Emfe, char *d_mfe_ss, int32_t *d_Ed);

/*
 * Timing hook for bench.py: average device time in milliseconds of the most recent
 * drna_score_batch*_ call, per kernel family, measured with HIP events on the engine's own
 * streams.  out[0] = MFE fill+traceback kernel, out[1] = PF kernel, out[2] = eval kernel,
 * out[3] = whole call (first launch to last completion).
 */
int drna_last_timing(const drna_engine *e, float out[4]);

/*
 * Ragged batches: sequences of DIFFERENT lengths scored in one call -- BASELINE config 4 (the 100 Eterna100-V1 puzzles,
 * 12 ... 400 nt, R replicas each) is one such batch instead of 100 calls.  The reference has no counterpart (it scores one
 * sequence per call); per sequence the results are those of drna_score_batch.
 *
 * drna_set_targets_ragged: n_targets structures of lengths lens[t], concatenated without terminators.
 * drna_score_ragged: R sequences of lengths lens[r], concatenated; target_of[r] names the structure E(target) is evaluated
 * on (same length; needed with DRNA_NEED_EVAL only).  Outputs: Epf R doubles, Emfe R int32, mfe_ss concatenated like the
 * sequences, Ed R int32 (ONE structure per sequence).  Sum of lengths <= max_R * max_L.
 */
int drna_set_targets_ragged(drna_engine *e, int n_targets, const int32_t *lens, const char *targets);
int drna_score_ragged(drna_engine *e, int R, const int32_t *lens, const char *seqs, const int32_t *target_of,
                      uint32_t flags, double *Epf, int32_t *Emfe, char *mfe_ss, int32_t *Ed);

/*
 * Two interacting strands (reference oligo_state homodimer / heterodimer, utils/energy_scores.py:154-158): R sequence
 * pairs of total length L, both strands concatenated WITHOUT the '&', the first strand `cut` nucleotides long.
 * Replaces fc.mfe_dimer() (structure + energy; the caller re-inserts the '&' at `cut`), fc.pf_dimer() (F4 = FA, FB, FcAB,
 * FAB in kcal/mol per pair; the reference's Epf is FAB = pf_dimer()[-1]; homodimer symmetry correction included) and the
 * two-strand fc.eval_structure (Ed against the structures of drna_set_targets, '&' removed, same cut).
 *   F4 R*4 doubles (NEED_PF), Emfe R int32 + mfe_ss R*L chars (NEED_MFE), Ed R*n_targets int32 (NEED_EVAL)
 */
int drna_cofold_batch(drna_engine *e, int R, int L, int cut, const char *seqs, uint32_t flags, double *F4,
                      int32_t *Emfe, char *mfe_ss, int32_t *Ed);

/*
 * Energy of the second-best structure of R sequences (negative design, -nd on).  Replaces
 * get_first_suboptimal_structure_and_energy(seq, fc, 1)[1] (utils/energy_scores.py:105-107, :453-488): ViennaRNA's subopt
 * enumeration with a growing energy band until it holds two structures, sorted by energy, second entry.
 *   E2   R int32, dcal/mol; 0 when no second structure lies within 4900 dcal/mol of the MFE (the reference's fallback)
 *   E12  R*2 int32, may be NULL: the two lowest structure energies (second = 10000000 if there is one structure only)
 */
int drna_subopt_energy_batch(drna_engine *e, int R, int L, const char *seqs, int32_t *E2, int32_t *E12);

/*
 * The K lowest-energy structures of R sequences, energies and dot-bracket strings.  Replaces
 * get_first_suboptimal_structure_and_energy(seq, fc, k)[0] for k = 1 .. #alternative structures, the call behind
 * get_alt_mcc() (utils/sequence_utils.py:766-793, utils/energy_scores.py:453-488): entry k of ViennaRNA's energy-sorted
 * subopt list (uniq_ML = 1) is rank k here (rank 0 = a ground state).  Structures of equal energy come in this engine's
 * own fixed order (ViennaRNA's order among ties is pinned nowhere in the reference).  R is not limited by max_R (the
 * batch is worked off in chunks).
 *   K    1 .. 8
 *   E    R*K int32, dcal/mol, ascending per sequence; 10000000 where the sequence has fewer than rank+1 structures
 *   ss   R*K*L chars (no terminator); all dots where E = 10000000
 */
int drna_subopt_structs_batch(drna_engine *e, int R, int L, const char *seqs, int K, int32_t *E, char *ss);

/*
 * Ensemble defect of R sequences against targets[0] (needs drna_set_targets with the same L): inside fill,
 * outside recursion, base-pair probabilities, then (1/L) * [ sum_{i unpaired in target} sum_j P(i,j)
 * + sum_{i paired with m in target} (1 - P(i,m)) ], '(' ')' pairs only.
 * Replaces ScoreSeq.get_ensemble_defect (energy_scores.py:362-374: new fold compound with bpp on, fc.mfe(),
 * fc.exp_params_rescale(mfe), fc.pf(), fc.ensemble_defect(target)); the rescale only moves pf_scale, which
 * cancels in every probability.
 *   edef  R doubles in [0,1]                                     out
 *   bpp   R*(L+1)*(L+1) doubles, P(i,j) at [r][i][j], 1 <= i < j <= L  out, may be NULL
 */
int drna_ensemble_defect_batch(drna_engine *e, int R, int L, const char *seqs, double *edef, double *bpp);

/* Same with device-resident buffers; d_bpp (may be NULL) must be zero-filled by the caller. */
int drna_ensemble_defect_batch_device(drna_engine *e, int R, int L, const char *d_seqs, double *d_edef,
                                      double *d_bpp);

/* device ms summed over the drna_score_batch[_device] calls since the last reset: out[0..3] as drna_last_timing, out[4] = number
 * of calls; out may be NULL (reset only).  Lets a caller time a loop of calls without a query per call. */
int drna_timing_sums(drna_engine *e, double out[5], int reset);

/* device ms of the last drna_ensemble_defect_batch*: out[0] = inside kernel, out[1] = outside kernel */
int drna_last_edef_timing(const drna_engine *e, float out[2]);

/*
 * Engine options.  "dual" (default 1): in batches small enough to leave half of the chip idle (4 R <= compute units, L <= 200)
 * the MFE fold of every sequence of at least 170 nt is done by TWO workgroups (fold_mfe_dual.hpp).  Results do not depend
 * on it (integer minima: bit-identical).  0 = always one workgroup per sequence, 2 = two workgroups whenever the batch allows,
 * whatever the length.  DRNA_DUAL in the environment sets the default.
 * "strips" (default 1): sequences of 201 .. 2046 nt are folded by strips of columns, one workgroup per strip of <= 120 columns
 * (fold_pf_strip.hpp, fold_mfe_strip.hpp), in drna_score_batch* and drna_score_ragged.  MFE energies and structures do not
 * depend on it (bit-identical), Epf agrees to 1e-13 kcal/mol (another summation order).  0 = the general one-workgroup kernels,
 * 2 = two strips also for 64 < n <= 200 (diagnostics).  DRNA_STRIPS in the environment sets the default.
 * "pf_helper" (default 1): batches small enough to leave CUs idle (2 R + the MFE fold's workgroups <= CUs, 120 <= n <= 200) fold the
 * partition function with a helper workgroup per sequence that takes the far multiloop split points; Epf is bit-identical either way.
 * "mfe_fark_min_strips" (default 4, i.e. n > 360; one more beside a partition function when there are no pseudoknot rounds):
 * from this many strips on the MFE strips fold their multiloop splits in blocked form (16 x 16 tiles, the far split points as
 * (min,+) tile products); results are identical either way.
 * "mfe_split" (default 2): with pseudoknot rounds the strip path takes a fill launch and a traceback launch per round; a batch of
 * >= 32 sequences then goes in two halves on two streams so that one half's traceback runs under the other's fill; 1 = one part.
 */
int drna_set_option(drna_engine *e, const char *name, int value);

/* reads an option back ("dual", "strips"), or the counter "sync_fallbacks": calls in which a fold by several workgroups lost a
 * partner (a bounded wait expired -- HIP promises no dispatch order) and which were therefore redone, transparently, with one
 * workgroup per fold.  Option "strip_fault" = 1 injects such a loss into every strip launch (tests).  "last_workgroups": fold
 * workgroups (partition function + MFE kernels, resident side by side) of the last drna_score_batch call.  "solo_calls_left":
 * after three such calls in a row (a GPU shared with another process) the engine folds with one workgroup per fold for the
 * next 1000 calls, then probes again; setting "dual", "strips" or "pf_helper" ends that at once. */
int drna_get_option(const drna_engine *e, const char *name, int *value);

/* diagnostics (engine created with DRNA_STRIP_DEBUG=1 in the environment): start / end wall clocks (100 MHz ticks) of the MFE
 * strip workgroups of the last launch, out[slot][18][2]; returns the number of sequence slots copied (0 without the buffers) */
int drna_debug_strip_clocks(drna_engine *e, long long *out, int nslots);

/* engine facts: out[0]=device, out[1]=max_R, out[2]=max_L, out[3]=threads per workgroup,
 * out[4]=compute units, out[5]=bytes of device workspace */
int drna_info(const drna_engine *e, int64_t out[6]);

/*
 * Host-side pieces of the Monte-Carlo inner loop, batched over replicas (plain CPU code, no engine needed).
 *
 * drna_simscore_batch: SimScore(ref, query) of utils/sim_score.py:62-147 for R query structures against one
 * reference; outputs are the reference's rounded mcc / recall / precision (NOT 1 - x).  '&' must already be
 * replaced by "Ee" as utils/energy_scores.py:79 does.
 */
int drna_simscore_batch(int R, int L, const char *ref, const char *queries, double *mcc, double *recall,
                        double *precision);

/*
 * Per-replica random streams of the host helpers: the reference's worker calls random.seed(replica_index) at the start of
 * every exchange step (utils/replica_exchange_monte_carlo.py:227-228 with seeds = [0 .. R-1], :250) and then draws from
 * Python's global Mersenne Twister.  drna_rng_seed(seeds) = random.seed(int) for each of R streams (MT19937 init_by_array
 * on the integer's 32-bit digits); drna_rng_random = one random.random() per stream (53-bit, two outputs).
 *   rng_state   R * DRNA_RNG_WORDS uint32 (624 state words + position), owned by the caller
 */
#define DRNA_RNG_WORDS 625
int drna_rng_seed(int R, const uint64_t *seeds, uint32_t *rng_state);
int drna_rng_random(int R, uint32_t *rng_state, double *out);

/*
 * drna_propose_batch: one proposal per replica, the move set of mutate_sequence / get_mutation_position /
 * expand_cases (utils/sequence_utils.py:926-1136) for single-chain targets without alternative structures.
 *   target        L chars, every bracket family is a design pair
 *   allowed_mask  L bytes, bit0 A, bit1 C, bit2 G, bit3 U: letters_allowed of get_nt_list (:454-525)
 *   seqs, mfe_ss  R*L chars: current sequence and current MFE structure of each replica
 *   shelf_index   R ints: index of the replica's temperature shelf; the targeted-mutation probability is
 *                 round(linspace(tm_max, tm_min, n_shelves)[index], 2) (:963-967)
 *   rng_state     R * DRNA_RNG_WORDS uint32: one MT19937 stream per replica (drna_rng_seed), advanced in place; the
 *                 draws are CPython's (random(), choice(), choices()), in the order of the reference's code
 *   out_seqs      R*L chars
 */
int drna_propose_batch(int R, int L, const char *target, const unsigned char *allowed_mask, const char *seqs,
                       const char *mfe_ss, const int32_t *shelf_index, int n_shelves, double tm_max, double tm_min,
                       int targeted, uint32_t *rng_state, char *out_seqs);

/*
 * drna_propose_batch_alt: the same move set for targets WITH alternative structures (>alt_sec_struct): positions that sit
 * in a "snake" (a connected component of the pair graph of target + alternative structures, utils/sequence_utils.py:143-388)
 * move the whole component to another of its Watson-Crick colourings (:1081-1095); alternative pairs outside snakes are
 * ordinary design pairs.
 *   partner        L int32: partner of every design pair (target pairs + ordinary alternative pairs), -1 = none
 *   snake_of       L int32: snake index of a position or -1 (may be NULL when n_snakes == 0)
 *   snake_off      n_snakes+1 int32: snake k owns snake_nodes[snake_off[k] .. snake_off[k+1])  (ascending positions)
 *   snake_nstates  n_snakes int32 (1..4); snake_states: for snake k, 4 slots of len_k letters at 4*snake_off[k]
 * False negatives / positives of the targeted-mutation rule are taken against `target` only (input_file.target_pairs_tupl).
 */
int drna_propose_batch_alt(int R, int L, const char *target, const int32_t *partner, const unsigned char *allowed_mask,
                           const int32_t *snake_of, int n_snakes, const int32_t *snake_off, const int32_t *snake_nodes,
                           const int32_t *snake_nstates, const char *snake_states, const char *seqs, const char *mfe_ss,
                           const int32_t *shelf_index, int n_shelves, double tm_max, double tm_min, int targeted,
                           uint32_t *rng_state, char *out_seqs);

/*
 * drna_metropolis_batch: mc_delta of utils/replica_exchange_monte_carlo.py:26-57 for R replicas: accept iff
 * score_m <= score_o, else with probability exp(-Lconst / T * (score_m - score_o)) (one draw, only then).
 */
int drna_metropolis_batch(int R, const double *score_o, const double *score_m, const double *temps, double Lconst,
                          uint32_t *rng_state, unsigned char *accept, unsigned char *better);

/*
 * drna_mc_run: n_iter Monte-Carlo iterations of ALL replicas without returning to the caller -- the body of
 * single_replica_design (utils/replica_exchange_monte_carlo.py:176-210) for R replicas in lock-step: proposal
 * (drna_propose_batch[_alt] rules; partner / snake arrays may be NULL / 0 for plain targets), scoring of the R proposals on the
 * GPU (drna_score_batch with flags | PF | MFE | EVAL against the structures of drna_set_targets), SimScore, the -sf sum
 * (term_id: 0 Ed-Epf, 1 1-MCC, 2 sln_Epf, 3 Ed-MFE, 4 1-precision, 5 1-recall, 6 Edef; weights term_w; + mean E(alt) - Epf when
 * alternative structures are installed), Metropolis acceptance, state update.
 *   in/out per replica: seqs, mfe_ss (R*L chars), score, mcc1 (= 1 - MCC), Epf, Ed (kcal/mol), rng_state
 *   counters[3] += accepted, accepted-better, rejected;  best[4] = {1-MCC, score, Epf, Ed} and best_seq / best_ss (L chars)
 *   are replaced whenever an accepted state is better (lower 1-MCC, then lower score)
 */
int drna_mc_run(drna_engine *e, int R, int L, int n_iter, const char *target, const int32_t *partner,
                const unsigned char *allowed_mask, const int32_t *snake_of, int n_snakes, const int32_t *snake_off,
                const int32_t *snake_nodes, const int32_t *snake_nstates, const char *snake_states,
                const int32_t *shelf_index, int n_shelves, double tm_max, double tm_min, int targeted, const double *temps,
                double Lconst, int n_terms, const int32_t *term_id, const double *term_w, uint32_t flags,
                uint32_t *rng_state, char *seqs, char *mfe_ss, double *score, double *mcc1, double *Epf, double *Ed,
                int64_t *counters, char *best_seq, char *best_ss, double *best);

#ifdef __cplusplus
}
#endif
#endif
