// engine.hip -- host side of the gfx950 replica-scoring engine and its C ABI (include/desirna_amd.h).
//
// One engine = one GPU.  Per call the three kernel families run concurrently on their own HIP
// streams (MFE fill+traceback, partition function, structure evaluation: they are independent per
// sequence, reference utils/energy_scores.py:150-151,75), bracketed by HIP events so bench.py can
// read per-kernel device times without a profiler.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <algorithm>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/desirna_amd.h"
#include "eval_structure.hpp"
#include "fold_mfe.hpp"
#include "fold_cofold.hpp"
#include "fold_mfe_lds.hpp"
#include "fold_mfe_dual.hpp"
#include "fold_fused.hpp"
#include "fold_mfe_strip.hpp"
#include "fold_outside.hpp"
#include "fold_pf.hpp"
#include "fold_pf_lds.hpp"
#include "fold_pf_strip.hpp"
#include "fold_subopt.hpp"
#include "host_driver.hpp"
#include "tables.hpp"

using namespace drna;

static std::string g_create_error;

#ifndef DRNA_MFE_FARK_MIN_STRIPS
#define DRNA_MFE_FARK_MIN_STRIPS 4
#endif
constexpr int MFE_FARK_MIN_STRIPS = DRNA_MFE_FARK_MIN_STRIPS;   // (measured: tools/time_strip_variants.py)
struct drna_engine {
  int device = 0, max_R = 0, max_L = 0, nt = 1024, cus = 0;
  bool lds_path = true;   // LDS-resident kernels when n fits (DRNA_PATH=global forces the general path)
  HostTables H;
  MfeTables* d_mfeT = nullptr;
  PfTables* d_pfT = nullptr;
  Plan* d_plan = nullptr;
  int *d_hp_len = nullptr, *d_bulge_len = nullptr, *d_int_len = nullptr;
  double *d_hp_w = nullptr, *d_scale = nullptr, *d_eMLb = nullptr;
  int32_t* d_ws_mfe = nullptr;
  double* d_ws_pf = nullptr;
  size_t ws_bytes = 0;
  int ws_slots = 0;               // sequences the fold workspaces hold at once: max_R, or fewer under the workspace budget (DRNA_WS_GB,
                                  // default 8): larger batches are folded in chunks of ws_slots, back to back on the same streams
  // staging for the host-buffer entry point
  char* d_seqs = nullptr;
  double* d_Epf = nullptr;
  int32_t* d_Emfe = nullptr;
  char* d_ss = nullptr;
  int32_t* d_Ed = nullptr;
  size_t ed_cap = 0;
  // per-sequence status words, host-mapped so no copy is needed after the streams drain
  int32_t *h_status = nullptr, *d_status = nullptr;   // [0,R) mfe, [max_R, max_R+R) pf
  short* d_pt = nullptr;
  int n_targets = 0, L_targets = 0;
  hipStream_t s_mfe = nullptr, s_pf = nullptr, s_eval = nullptr;
  hipEvent_t ev_mfe2 = nullptr;                // the second half of a batch whose MFE fold takes several launches runs on s_eval (no stream of
                                               // its own: the runtime maps streams onto a few hardware queues, and streams that share one serialize)
  hipEvent_t ev_start = nullptr, ev_end = nullptr, ev_m0 = nullptr, ev_m1 = nullptr, ev_p0 = nullptr,
             ev_p1 = nullptr, ev_e0 = nullptr, ev_e1 = nullptr;
  float timing[4] = {0, 0, 0, 0};
  double timing_sum[5] = {0, 0, 0, 0, 0};     // mfe, pf, eval, total ms summed over drna_score_batch_device calls; [4] = calls
  // ensemble defect (outside recursion): workspace allocated on first use
  double* d_ws_out = nullptr;
  double* d_edef = nullptr;
  hipEvent_t ev_o0 = nullptr, ev_o1 = nullptr, ev_o2 = nullptr;
  hipEvent_t ev_gate = nullptr;                // the partition function's launch waits for it (see pf_gate_round)
  float timing_edef[2] = {0, 0};
  // ragged batches: per-sequence descriptors (len, off, target, two index lists) and the structures' pair tables
  int* d_rg = nullptr;
  short* d_rpt = nullptr;
  int* d_rpt_off = nullptr;
  std::vector<int> rt_len;
  double* d_F4 = nullptr;   // co-fold free energies (FA, FB, FcAB, FAB per pair)
  // K-best structures: workspace for kb_chunk sequences, allocated on first use
  int32_t* d_ws_kb = nullptr;
  int32_t* d_kbE = nullptr;
  char* d_kbss = nullptr;
  int kb_chunk = 0;
  // host-mapped staging of the host-buffer entry point: the kernels read the sequences from and write their results to
  // pinned host memory directly, so a batch costs no hipMemcpy round trips (12.8 KB in, 13.6 KB out at R=64, L=200)
  char *hm_seqs = nullptr, *hm_ss = nullptr, *dm_seqs = nullptr, *dm_ss = nullptr;
  double *hm_Epf = nullptr, *dm_Epf = nullptr;
  int32_t *hm_Emfe = nullptr, *dm_Emfe = nullptr, *hm_Ed = nullptr, *dm_Ed = nullptr;
  size_t hm_Ed_cap = 0;
  bool zero_copy = true;
  // two-workgroup kernels (small batches: 4 R <= CUs, n <= 200): exchange rows and flags, allocated on first use
  bool dual = true;               // DRNA_DUAL=0 turns them off
  bool dual_force = false;        // option "dual" = 2: also beside a partition function (tests, diagnostics)
  int dual_cap = 0;               // sequences the exchange buffers hold
  int dual_epoch = 0;             // grows by one per launch; flags and epoch go back to zero at DUAL_EPOCH_RESET (fold_common.hpp, DualLink)
  int* d_dflags = nullptr;        // [2 kernels][dual_cap][64]
  int32_t *d_xs = nullptr, *d_xa_mfe = nullptr, *d_xb_mfe = nullptr;
  // strip kernels (fold_pf_strip.hpp: 200 < n <= 2046, several workgroups per sequence): one flag line per (sequence, strip)
  int strips = 1;                 // 0 off (general kernel), 1 for n > 200, 2 also for 64 < n <= 200 (two strips; diagnostics)
  int strip_epoch = 0;            // grows by one per launch; flags and epoch go back to zero at STRIP_EPOCH_RESET
  int flag_resets = 0;            // times the hand-over flags were zeroed because an epoch neared the compare range
  int strip_fault = 0;            // option "strip_fault": inject a lost strip (tests)
  bool cur_with_pf = false;       // the call being enqueued also folds the partition function
  int mfe_fark_min_strips = MFE_FARK_MIN_STRIPS;   // option "mfe_fark_min_strips": MFE strips fold in blocked form from this many strips on
  int mfe_split = 2;              // option "mfe_split": parts of a batch (on two streams) for the pseudoknot rounds of the strip path; 1 = off
  bool helper_fault = false;      // tests: the helper workgroups of the partition function leave at once (a lost partner)
  bool pf_helper = true;          // small batches: a helper workgroup per sequence computes the far multiloop split points of the
                                  // partition function (fold_pf_lds.hpp, pf_kfar_helper); option "pf_helper"
  int* d_pflags = nullptr;        // its hand-over flags: per sequence two 128-byte lines
  int pflags_cap = 0, pfh_epoch = 0;
  bool fused = true;              // small batches: both folds in ONE launch of 4 R workgroups (fold_fused.hpp); option "fused", DRNA_FUSED=0 turns it off.
                                  // On since the end of round 4: 0.413 against 0.420 ms of device time at R = 64 x L = 200 (the two launches' kernels
                                  // start and end a few us apart; the host pays the same 14 us either way)
  int fused_blocks_per_cu = -1;   // occupancy query of the fused kernel (-1: not asked yet)
  int pair_blocks_per_cu = -1;    // ... of the two-workgroup MFE kernel and the partition function with helpers (the smaller of the two answers)
  long long *h_clk = nullptr, *d_clk = nullptr;   // host-mapped: start / end wall clock of every block of the fused launch
  int clk_cap = 0;
  bool last_fused = false;        // the last drna_score_batch_device call went through the fused launch
  int last_wgs = 0;               // fold workgroups of the last drna_score_batch call (partition function + MFE kernels, resident side by side)
  int sync_fallbacks = 0;         // calls that lost a multi-workgroup fold (ST_SYNC) and were redone with one workgroup per fold
  // A GPU shared with another process (or a runtime that stops dispatching in block order) loses partners call after call, and
  // every lost call costs its wait budget before it is redone.  Three fallbacks in a row switch the multi-workgroup paths off
  // for the next SOLO_CALLS calls (option "solo_calls_left"); then one call probes again.  Any set_option of the paths resets it.
  int fallback_streak = 0, solo_left = 0;
  bool in_fallback = false;
  int mc_threads_used = 0;        // worker threads (the caller included) of the last drna_mc_run
  int mc_threads = 0;             // option "mc_threads" / DRNA_MC_THREADS: worker threads of drna_mc_run's host work (0 = min(8, usable CPUs / 2))
  int* d_sflags = nullptr;        // [2: partition function, MFE][max_R][STRIP_MAXS][32]
  int32_t* d_srec = nullptr;      // MFE strips: exchange records and list counts, srec_stride int32 per sequence
  long long srec_stride = 0;
  long long* d_sclk = nullptr;    // DRNA_STRIP_DEBUG=1: start / end clocks of the MFE strip workgroups of the last launch
  int* d_sdbg = nullptr;          // DRNA_STRIP_DEBUG=1: [2][max_R][8] words written by a strip whose wait failed
  std::string err;
};

#define HIP_TRY(call)                                                                      \
  do {                                                                                     \
    hipError_t _e = (call);                                                                \
    if (_e != hipSuccess) {                                                                \
      e->err = std::string(#call) + ": " + hipGetErrorString(_e);                          \
      /* kernels of this call may already be enqueued on the engine's other streams: the next call assumes idle   */ \
      /* streams (it rewrites h_status and the workspaces), so drain the device before handing the error back     */ \
      (void)hipDeviceSynchronize();                                                        \
      return DRNA_ERR_DEVICE;                                                              \
    }                                                                                      \
  } while (0)

template <typename T>
static hipError_t upload(T** dst, const T* src, size_t count) {
  hipError_t r = hipMalloc((void**)dst, count * sizeof(T));
  if (r != hipSuccess) return r;
  return hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice);
}

static size_t mfe_ws_stride(int ld) { return (size_t)5 * ld * ld; }                       // int32
// hand-over flags hold (epoch << 12 | diagonal) for the strips and ((epoch * 8 + round) << 10 | diagonal) for the two-workgroup
// kernel, compared wrap-safe: valid while live values are less than 2^31 apart, i.e. 2^19 (2^18) epochs.  Reset at a quarter of that.
constexpr int STRIP_EPOCH_RESET = 1 << 17, DUAL_EPOCH_RESET = 1 << 16;
#ifndef DRNA_PF_HELPER_NMIN
#define DRNA_PF_HELPER_NMIN 95
#endif
constexpr int PF_HELPER_NMIN = DRNA_PF_HELPER_NMIN;       // shorter sequences have too few far split points to repay a second workgroup
                                                          // (tools/pf_helper_lengths.py, R = 64: 90 nt 0.159 against 0.161 ms, 100 nt 0.176 against 0.182, 120 nt 0.215 against 0.228)

// strips of a sequence of length n (0 = not a strip case): widest strip STRIP_WMAX columns; the exchange records of the
// S - 1 strip boundaries must fit tables 0 and 1 of the sequence's workspace
static int strips_for(const drna_engine* e, int n, int ld) {
  if (!e->strips || !e->lds_path || e->nt != 1024 || n > STRIP_NMAX) return 0;
  int S = 0;
  if (n > PF_FAST_NMAX) S = strip_count(n, STRIP_WMAX);
  else if (e->strips == 2 && n > 64) S = 2;
  if (S < 2 || S > STRIP_MAXS) return 0;
  if ((long long)(S - 1) * STRIP_REC > 2ll * ld) return 0;
  return S;
}

// flags (and, for the MFE fold, the record buffer) of the strip kernels: allocated (flags zeroed, once) on first use
static int strip_flags(drna_engine* e, bool mfe) {
  if (!e->d_sflags) {
    const size_t b = (size_t)2 * e->max_R * STRIP_MAXS * 32 * sizeof(int);
    HIP_TRY(hipMalloc((void**)&e->d_sflags, b));
    HIP_TRY(hipMemset(e->d_sflags, 0, b));
    HIP_TRY(hipDeviceSynchronize());      // the memset runs on the null stream, the kernels on non-blocking streams of their own
    e->strip_epoch = 0;
    if (getenv("DRNA_STRIP_DEBUG")) {
      HIP_TRY(hipMalloc((void**)&e->d_sclk, (size_t)e->max_R * STRIP_MAXS * 2 * sizeof(long long)));
      HIP_TRY(hipMalloc((void**)&e->d_sdbg, (size_t)2 * e->max_R * 8 * sizeof(int)));
      HIP_TRY(hipMemset(e->d_sdbg, 0, (size_t)2 * e->max_R * 8 * sizeof(int)));
      HIP_TRY(hipDeviceSynchronize());
    }
  }
  // Flag compares are wrap-safe over HALF the 32-bit range only (2^19 epochs of 4096 values): a slot that was never written, or
  // not written for 2^19 launches (the MFE half after a long partition-function-only phase, a larger batch than seen before,
  // more strips than before), would then read as already published.  Every stream is idle here, so long before that point the
  // flags go back to zero and the epochs start over.
  if (e->strip_epoch >= STRIP_EPOCH_RESET) {
    HIP_TRY(hipMemset(e->d_sflags, 0, (size_t)2 * e->max_R * STRIP_MAXS * 32 * sizeof(int)));
    HIP_TRY(hipDeviceSynchronize());
    e->strip_epoch = 0;
    e->flag_resets++;
  }
  if (mfe && !e->d_srec) {
    const int smax = std::min(STRIP_MAXS, strip_count(std::min(e->max_L, STRIP_NMAX), STRIP_WMAX) + 1);
    e->srec_stride = (long long)std::max(smax, 2) * (e->max_L + 2) * MSTRIP_REC;
    HIP_TRY(hipMalloc((void**)&e->d_srec, (size_t)e->srec_stride * e->max_R * sizeof(int32_t)));
  }
  return DRNA_OK;
}
static int next_strip_epoch(drna_engine* e) {
  e->strip_epoch = (int)((unsigned)e->strip_epoch + 1u);          // reset by strip_flags() long before the compares' half range
  return (int)((unsigned)e->strip_epoch << 12);
}

// nseq sequences (slots first_slot ...; idx = their sequence numbers or null) by S strips each
constexpr int SOLO_CALLS = 1000;      // calls with one workgroup per fold after three lost calls in a row (see fallback_streak)
static void launch_pf_strips(drna_engine* e, const PfArgs& a, int nseq, int S, int first_slot, const int* idx, hipStream_t st) {
  StripLink lk;
  lk.flags = e->d_sflags + (size_t)first_slot * STRIP_MAXS * 32;
  lk.base = next_strip_epoch(e);
  lk.nseq = nseq; lk.S = S; lk.idx = idx; lk.pad = strip_pad(S); lk.fault = e->strip_fault;
  lk.dbg = e->d_sdbg ? e->d_sdbg + (size_t)first_slot * 8 : nullptr;
  const int groups = (nseq + 7) / 8;
  hipLaunchKernelGGL(pf_strip_kernel<1024>, dim3(groups * 8 * (S + strip_pad(S))), dim3(1024), 0, st, a, lk);
}

// MFE fold of nseq sequences by S strips each: per pseudoknot round one launch of the fill and one of the traceback
// one pseudoknot round of nseq sequences (slots first_slot ...; idx = their sequence numbers, or null: sequences r0 ...)
static void launch_mfe_strips_round(drna_engine* e, const MfeArgs& a, int nseq, int S, int first_slot, const int* idx, int r0,
                                    hipStream_t st, int round, hipEvent_t after_fill = nullptr) {
  StripRec xr;
  xr.rec = e->d_srec; xr.stride = e->srec_stride;
  const int groups = (nseq + 7) / 8;
  StripLink lk;
  lk.flags = e->d_sflags + ((size_t)e->max_R + first_slot) * STRIP_MAXS * 32;
  lk.base = next_strip_epoch(e);
  lk.nseq = nseq; lk.S = S; lk.idx = idx; lk.r0 = r0; lk.pad = strip_pad(S); lk.fault = e->strip_fault;
  // blocked multiloop splits for the long folds (fold_mfe_strip.hpp, MKT_L).  Alone they win from four strips on (400 nt x 256:
  // 4.86 -> 4.48 ms); beside the partition function's strips only from five on (400 nt: both folds 9.31 -> 9.94 ms, 600 nt x 128:
  // 11.3 -> 9.9 ms) -- unless the MFE fold is a chain of pseudoknot rounds, which then dominates the call (config 5: 10.4 -> 9.6 ms)
  lk.fark = S >= e->mfe_fark_min_strips + ((e->cur_with_pf && a.pk_rounds == 0) ? 1 : 0);
  lk.dbg = e->d_sdbg ? e->d_sdbg + ((size_t)e->max_R + first_slot) * 8 : nullptr;
  lk.clk = e->d_sclk ? e->d_sclk + (size_t)first_slot * STRIP_MAXS * 2 : nullptr;
  if (lk.fark) hipLaunchKernelGGL((mfe_strip_kernel<1024, true>), dim3(groups * 8 * (S + strip_pad(S))), dim3(1024), 0, st, a, lk, xr, round);
  else hipLaunchKernelGGL((mfe_strip_kernel<1024, false>), dim3(groups * 8 * (S + strip_pad(S))), dim3(1024), 0, st, a, lk, xr, round);
  if (after_fill) (void)hipEventRecord(after_fill, st);
  hipLaunchKernelGGL(mfe_strip_trace_kernel, dim3(nseq), dim3(TRACE_WAVES * WAVE), 0, st, a, idx, nseq, round, r0);
}
// MFE fold of nseq sequences by S strips each: per pseudoknot round one launch of the fill and one of the traceback
static void launch_mfe_strips(drna_engine* e, const MfeArgs& a, int nseq, int S, int first_slot, const int* idx, hipStream_t st) {
  if (e->d_sdbg) fprintf(stderr, "mfe strips: rec %p stride %lld max_R %d nseq %d S %d ld %d L %d ws %p\n", (void*)e->d_srec, e->srec_stride, e->max_R, nseq, S, a.ld, a.L, (void*)a.ws);
  for (int round = 0; round <= a.pk_rounds; round++) launch_mfe_strips_round(e, a, nseq, S, first_slot, idx, 0, st, round);
}
static size_t pf_ws_stride(int ld) { return (size_t)7 * ld * ld + ((size_t)ld * ld + 7) / 8; }  // doubles

static int create_impl(drna_engine* e, const int32_t* params, int n_int32, int device, int max_R, int max_L) {
  if (!params || max_R < 1 || max_L < 1 || max_L > MAXN - 2) {
    e->err = "drna_create: bad argument (1 <= max_L <= 2046, max_R >= 1)";
    return DRNA_ERR_ARG;
  }
  std::string msg = build_tables(params, n_int32, e->H);
  if (!msg.empty()) { e->err = msg; return DRNA_ERR_PARAMS; }
  size_tables(e->H, max_L + 2);
  e->device = device; e->max_R = max_R; e->max_L = max_L;
  if (const char* s = getenv("DRNA_NT")) {
    int v = atoi(s);
    if (v == 256 || v == 512 || v == 1024) e->nt = v;
  }
  if (const char* s = getenv("DRNA_PATH")) e->lds_path = std::string(s) != "global";
  HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  e->cus = prop.multiProcessorCount;
  if (const char* dv = getenv("DRNA_DUAL")) { e->dual = atoi(dv) != 0; e->dual_force = atoi(dv) == 2; }
  if (const char* hv = getenv("DRNA_PF_HELPER")) e->pf_helper = atoi(hv) != 0;
  if (const char* fv = getenv("DRNA_FUSED")) e->fused = atoi(fv) != 0;
  if (const char* mv = getenv("DRNA_MFE_SPLIT")) { const int v = atoi(mv); e->mfe_split = v < 1 ? 1 : v > 8 ? 8 : v; }
  if (const char* fv = getenv("DRNA_MFE_FARK_MIN_STRIPS")) { const int v = atoi(fv); e->mfe_fark_min_strips = v < 1 ? 1 : v; }
  if (const char* tv = getenv("DRNA_MC_THREADS")) { const int v = atoi(tv); e->mc_threads = v < 0 ? 0 : v > 64 ? 64 : v; }
  if (const char* sv = getenv("DRNA_STRIPS")) { const int v = atoi(sv); e->strips = v < 0 ? 0 : v > 2 ? 2 : v; }
  HIP_TRY(upload(&e->d_mfeT, &e->H.mfe, 1));
  HIP_TRY(upload(&e->d_pfT, &e->H.pf, 1));
  HIP_TRY(upload(&e->d_plan, &e->H.plan, 1));
  HIP_TRY(upload(&e->d_hp_len, e->H.hp_len.data(), e->H.hp_len.size()));
  HIP_TRY(upload(&e->d_bulge_len, e->H.bulge_len.data(), e->H.bulge_len.size()));
  HIP_TRY(upload(&e->d_int_len, e->H.int_len.data(), e->H.int_len.size()));
  HIP_TRY(upload(&e->d_hp_w, e->H.hp_w.data(), e->H.hp_w.size()));
  HIP_TRY(upload(&e->d_scale, e->H.scale.data(), e->H.scale.size()));
  HIP_TRY(upload(&e->d_eMLb, e->H.eMLb.data(), e->H.eMLb.size()));
  const int ld = max_L + 2;
  {
    // O(ld^2) tables per sequence: 7 fp64 + 5 int32 tables, 12.3 MB at 400 nt.  A batch of 3,200 sequences used to take 39.8 GB;
    // the workspaces now hold at most DRNA_WS_GB (default 8) and a larger batch goes through them in chunks
    double gb = 8.0;
    if (const char* g = getenv("DRNA_WS_GB")) gb = std::max(0.05, atof(g));
    const double per = (double)mfe_ws_stride(ld) * sizeof(int32_t) + (double)pf_ws_stride(ld) * sizeof(double);
    const long long fit = (long long)(gb * 1e9 / per);
    e->ws_slots = (int)std::max(1ll, std::min((long long)max_R, fit));
  }
  size_t bm = mfe_ws_stride(ld) * sizeof(int32_t) * e->ws_slots, bp = pf_ws_stride(ld) * sizeof(double) * e->ws_slots;
  HIP_TRY(hipMalloc((void**)&e->d_ws_mfe, bm));
  HIP_TRY(hipMalloc((void**)&e->d_ws_pf, bp));
  e->ws_bytes = bm + bp;
  HIP_TRY(hipMalloc((void**)&e->d_seqs, (size_t)max_R * max_L));
  HIP_TRY(hipMalloc((void**)&e->d_Epf, (size_t)max_R * sizeof(double)));
  HIP_TRY(hipMalloc((void**)&e->d_Emfe, (size_t)max_R * sizeof(int32_t)));
  HIP_TRY(hipMalloc((void**)&e->d_ss, (size_t)max_R * max_L));
  HIP_TRY(hipHostMalloc((void**)&e->h_status, (size_t)2 * max_R * sizeof(int32_t), hipHostMallocMapped));
  HIP_TRY(hipHostGetDevicePointer((void**)&e->d_status, e->h_status, 0));
  memset(e->h_status, 0, (size_t)2 * max_R * sizeof(int32_t));
  if (const char* z = getenv("DRNA_STAGING")) e->zero_copy = std::string(z) != "device";
  HIP_TRY(hipHostMalloc((void**)&e->hm_seqs, (size_t)max_R * max_L, hipHostMallocMapped));
  HIP_TRY(hipHostMalloc((void**)&e->hm_ss, (size_t)max_R * max_L, hipHostMallocMapped));
  HIP_TRY(hipHostMalloc((void**)&e->hm_Epf, (size_t)max_R * sizeof(double), hipHostMallocMapped));
  HIP_TRY(hipHostMalloc((void**)&e->hm_Emfe, (size_t)max_R * sizeof(int32_t), hipHostMallocMapped));
  HIP_TRY(hipHostGetDevicePointer((void**)&e->dm_seqs, e->hm_seqs, 0));
  HIP_TRY(hipHostGetDevicePointer((void**)&e->dm_ss, e->hm_ss, 0));
  HIP_TRY(hipHostGetDevicePointer((void**)&e->dm_Epf, e->hm_Epf, 0));
  HIP_TRY(hipHostGetDevicePointer((void**)&e->dm_Emfe, e->hm_Emfe, 0));
  HIP_TRY(hipStreamCreateWithFlags(&e->s_mfe, hipStreamNonBlocking));
  HIP_TRY(hipStreamCreateWithFlags(&e->s_pf, hipStreamNonBlocking));
  HIP_TRY(hipStreamCreateWithFlags(&e->s_eval, hipStreamNonBlocking));
  HIP_TRY(hipEventCreateWithFlags(&e->ev_mfe2, hipEventDisableTiming));
  hipEvent_t* evs[] = {&e->ev_start, &e->ev_end, &e->ev_m0, &e->ev_m1, &e->ev_p0, &e->ev_p1, &e->ev_e0, &e->ev_e1,
                       &e->ev_o0, &e->ev_o1, &e->ev_o2, &e->ev_gate};
  for (hipEvent_t* ev : evs) HIP_TRY(hipEventCreate(ev));
  return DRNA_OK;
}

extern "C" int drna_create(const int32_t* params, int n_int32, int device, int max_R, int max_L, drna_engine** out) {
  if (!out) return DRNA_ERR_ARG;
  *out = nullptr;
  drna_engine* e = new drna_engine();
  int rc = create_impl(e, params, n_int32, device, max_R, max_L);
  if (rc != DRNA_OK) {
    g_create_error = e->err;
    drna_destroy(e);
    return rc;
  }
  *out = e;
  return DRNA_OK;
}

extern "C" void drna_destroy(drna_engine* e) {
  if (!e) return;
  void* bufs[] = {e->d_mfeT, e->d_pfT, e->d_plan, e->d_hp_len, e->d_bulge_len, e->d_int_len, e->d_hp_w, e->d_scale,
                  e->d_eMLb, e->d_ws_mfe, e->d_ws_pf, e->d_seqs, e->d_Epf, e->d_Emfe, e->d_ss, e->d_Ed, e->d_pt,
                  e->d_ws_out, e->d_edef, e->d_rg, e->d_rpt, e->d_rpt_off, e->d_F4, e->d_ws_kb, e->d_kbE, e->d_kbss,
                  e->d_dflags, e->d_xs, e->d_xa_mfe, e->d_xb_mfe, e->d_sflags, e->d_srec, e->d_sdbg, e->d_sclk, e->d_pflags};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  if (e->h_status) (void)hipHostFree(e->h_status);
  void* hm[] = {e->hm_seqs, e->hm_ss, e->hm_Epf, e->hm_Emfe, e->hm_Ed, e->h_clk};
  for (void* b : hm)
    if (b) (void)hipHostFree(b);
  hipStream_t ss[] = {e->s_mfe, e->s_pf, e->s_eval};
  if (e->ev_mfe2) (void)hipEventDestroy(e->ev_mfe2);
  for (hipStream_t s : ss)
    if (s) (void)hipStreamDestroy(s);
  hipEvent_t evs[] = {e->ev_start, e->ev_end, e->ev_m0, e->ev_m1, e->ev_p0, e->ev_p1, e->ev_e0, e->ev_e1,
                      e->ev_o0, e->ev_o1, e->ev_o2, e->ev_gate};
  for (hipEvent_t ev : evs)
    if (ev) (void)hipEventDestroy(ev);
  delete e;
}

extern "C" int drna_abi_version(void) { return DRNA_ABI_VERSION; }

extern "C" int drna_set_option(drna_engine* e, const char* name, int value) {
  if (!e || !name) return DRNA_ERR_ARG;
  if (!strcmp(name, "dual") || !strcmp(name, "strips") || !strcmp(name, "pf_helper") || !strcmp(name, "fused")) { e->fallback_streak = 0; e->solo_left = 0; }
  if (!strcmp(name, "fused")) { e->fused = value != 0; return DRNA_OK; }
  if (!strcmp(name, "dual")) { e->dual = value != 0; e->dual_force = value == 2; return DRNA_OK; }
  if (!strcmp(name, "strips")) { e->strips = value < 0 ? 0 : value > 2 ? 2 : value; return DRNA_OK; }
  if (!strcmp(name, "pf_helper")) { e->pf_helper = value != 0; return DRNA_OK; }
  if (!strcmp(name, "helper_fault")) { e->helper_fault = value != 0; return DRNA_OK; }
  if (!strcmp(name, "strip_fault")) { e->strip_fault = value != 0; return DRNA_OK; }
  if (!strcmp(name, "mfe_fark_min_strips")) { e->mfe_fark_min_strips = value < 1 ? 1 : value; return DRNA_OK; }
  if (!strcmp(name, "mc_threads")) { e->mc_threads = value < 0 ? 0 : value > 64 ? 64 : value; return DRNA_OK; }
  if (!strcmp(name, "mfe_split")) { e->mfe_split = value < 1 ? 1 : value > 8 ? 8 : value; return DRNA_OK; }
  if (!strcmp(name, "debug_epoch")) { e->strip_epoch = value; e->dual_epoch = value; e->pfh_epoch = value; return DRNA_OK; }     // tests: jump near the reset point
  e->err = std::string("drna_set_option: unknown option ") + name;
  return DRNA_ERR_ARG;
}

// diagnostics (DRNA_STRIP_DEBUG=1): start / end wall clocks (100 MHz ticks) of the MFE strip workgroups of the last launch,
// out[slot][strip][2]; returns the number of slots copied (0 without the debug buffers)
extern "C" int drna_debug_strip_clocks(drna_engine* e, long long* out, int nslots) {
  if (!e || !e->d_sclk || !out) return 0;
  const int m = std::min(nslots, e->max_R);
  if (hipMemcpy(out, e->d_sclk, (size_t)m * STRIP_MAXS * 2 * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess) return 0;
  return m;
}

extern "C" int drna_get_option(const drna_engine* e, const char* name, int* value) {
  if (!e || !name || !value) return DRNA_ERR_ARG;
  if (!strcmp(name, "dual")) { *value = e->dual ? (e->dual_force ? 2 : 1) : 0; return DRNA_OK; }
  if (!strcmp(name, "strips")) { *value = e->strips; return DRNA_OK; }
  if (!strcmp(name, "pf_helper")) { *value = e->pf_helper ? 1 : 0; return DRNA_OK; }
  if (!strcmp(name, "fused")) { *value = e->fused ? 1 : 0; return DRNA_OK; }
  if (!strcmp(name, "last_fused")) { *value = e->last_fused ? 1 : 0; return DRNA_OK; }
  if (!strcmp(name, "fused_blocks_per_cu")) { *value = e->fused_blocks_per_cu; return DRNA_OK; }
  if (!strcmp(name, "pair_blocks_per_cu")) { *value = e->pair_blocks_per_cu; return DRNA_OK; }
  if (!strcmp(name, "sync_fallbacks")) { *value = e->sync_fallbacks; return DRNA_OK; }
  if (!strcmp(name, "solo_calls_left")) { *value = e->solo_left; return DRNA_OK; }
  if (!strcmp(name, "last_workgroups")) { *value = e->last_wgs; return DRNA_OK; }
  if (!strcmp(name, "mc_threads")) { *value = e->mc_threads; return DRNA_OK; }
  if (!strcmp(name, "mc_threads_used")) { *value = e->mc_threads_used; return DRNA_OK; }
  if (!strcmp(name, "workspace_slots")) { *value = e->ws_slots; return DRNA_OK; }
  if (!strcmp(name, "flag_resets")) { *value = e->flag_resets; return DRNA_OK; }
  if (!strcmp(name, "debug_epoch")) { *value = std::max(e->pfh_epoch, std::max(e->strip_epoch, e->dual_epoch)); return DRNA_OK; }
  return DRNA_ERR_ARG;
}

extern "C" const char* drna_last_error(const drna_engine* e) { return e ? e->err.c_str() : g_create_error.c_str(); }

extern "C" int drna_set_targets(drna_engine* e, int n_targets, int L, const char* targets) {
  if (!e) return DRNA_ERR_ARG;
  if (n_targets < 0 || L < 1 || L > e->max_L || (n_targets > 0 && !targets)) {
    e->err = "drna_set_targets: bad argument";
    return DRNA_ERR_ARG;
  }
  std::vector<short> pt((size_t)n_targets * (L + 2), 0);
  std::vector<int> stk;
  for (int k = 0; k < n_targets; k++) {
    stk.clear();
    const char* s = targets + (size_t)k * L;
    short* p = pt.data() + (size_t)k * (L + 2);
    for (int i = 1; i <= L; i++) {
      if (s[i - 1] == '(') stk.push_back(i);
      else if (s[i - 1] == ')') {
        if (stk.empty()) { e->err = "drna_set_targets: unbalanced ')'"; return DRNA_ERR_STRUCTURE; }
        int o = stk.back(); stk.pop_back();
        p[o] = (short)i; p[i] = (short)o;
      }
    }
    if (!stk.empty()) { e->err = "drna_set_targets: unbalanced '('"; return DRNA_ERR_STRUCTURE; }
  }
  HIP_TRY(hipSetDevice(e->device));
  if (e->d_pt) { (void)hipFree(e->d_pt); e->d_pt = nullptr; }
  if (e->d_Ed) { (void)hipFree(e->d_Ed); e->d_Ed = nullptr; }
  e->n_targets = n_targets; e->L_targets = L;
  if (n_targets) {
    HIP_TRY(upload(&e->d_pt, pt.data(), pt.size()));
    HIP_TRY(hipMalloc((void**)&e->d_Ed, (size_t)e->max_R * n_targets * sizeof(int32_t)));
  }
  return DRNA_OK;
}

template <int NT>
static void launch_mfe(const MfeArgs& a, int R, hipStream_t s) {
  hipLaunchKernelGGL(mfe_kernel<NT>, dim3(R), dim3(NT), 0, s, a);
}
template <int NT>
static void launch_pf(const PfArgs& a, int R, hipStream_t s) {
  hipLaunchKernelGGL(pf_kernel<NT>, dim3(R), dim3(NT), 0, s, a);
}

// Folds by several workgroups that wait for each other (fold_mfe_dual.hpp, pf_kfar_helper) need ALL their workgroups resident at
// once.  Every such launch is therefore sized against the occupancy query, which is the check hipLaunchCooperativeKernel makes
// (that call itself costs ~17 us more per launch, MI355X_MICROARCH.md 'coop-launch', and gives the same residency as a plain
// launch): workgroups of all concurrent launches <= blocks per CU x CUs.  What remains outside the engine's control is another
// process on the same GPU; the bounded waits and the one-workgroup fallback cover that (ST_SYNC).
static int pair_blocks_per_cu(drna_engine* e) {
  if (e->pair_blocks_per_cu < 0) {
    int a = 0, b = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, mfe_dual_kernel<1024>, 1024, 0) != hipSuccess) a = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, pf_lds_kernel<1024, false>, 1024, 0) != hipSuccess) b = 0;      // (the instance of the helper launches)
    e->pair_blocks_per_cu = std::min(a, b);
  }
  return e->pair_blocks_per_cu;
}

// The fused launch's workgroups wait for each other, so the whole grid must be resident at once: the grid is checked against
// the occupancy query (what hipLaunchCooperativeKernel would check, without its ~17 us per launch)
static bool fused_grid_fits(drna_engine* e, int R) {
  if (e->fused_blocks_per_cu < 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, score_fused_kernel<1024>, 1024, 0) != hipSuccess) nb = 0;
    e->fused_blocks_per_cu = nb;
  }
  return (long long)fused_grid(R) <= (long long)e->fused_blocks_per_cu * e->cus;
}

extern "C" int drna_score_batch_device(drna_engine* e, int R, int L, const char* d_seqs, uint32_t flags, double* d_Epf,
                                       int32_t* d_Emfe, char* d_mfe_ss, int32_t* d_Ed) {
  if (!e) return DRNA_ERR_ARG;
  const bool want_pf = flags & DRNA_NEED_PF, want_mfe = flags & (DRNA_NEED_MFE | DRNA_NEED_PK),
             want_pk = flags & DRNA_NEED_PK, want_ev = flags & DRNA_NEED_EVAL;
  if (R < 1 || R > e->max_R || L < 1 || L > e->max_L || !d_seqs || (want_pf && !d_Epf) ||
      (want_mfe && (!d_Emfe || !d_mfe_ss)) || (want_ev && !d_Ed)) {
    e->err = "drna_score_batch: bad argument (R, L within the engine's limits; output pointers for every requested flag)";
    return DRNA_ERR_ARG;
  }
  if (want_ev && (e->n_targets < 1 || e->L_targets != L)) {
    e->err = "drna_score_batch: DRNA_NEED_EVAL needs drna_set_targets() with the same L";
    return DRNA_ERR_ARG;
  }
  e->cur_with_pf = want_pf;
  if (!e->in_fallback && e->solo_left > 0) {             // after repeated lost partners: one workgroup per fold for a while
    const int s_strips = e->strips;
    const bool s_dual = e->dual, s_help = e->pf_helper;
    e->in_fallback = true; e->strips = 0; e->dual = false; e->pf_helper = false;
    const int rc = drna_score_batch_device(e, R, L, d_seqs, flags, d_Epf, d_Emfe, d_mfe_ss, d_Ed);
    e->strips = s_strips; e->dual = s_dual; e->pf_helper = s_help; e->in_fallback = false;
    e->solo_left--;
    return rc;
  }
  HIP_TRY(hipSetDevice(e->device));
  if (R > e->ws_slots) {
    // more sequences than the workspaces hold (DRNA_WS_GB): one sub-batch of ws_slots after the other
    const int nt = std::max(1, e->n_targets);
    for (int r0 = 0; r0 < R; r0 += e->ws_slots) {
      const int m = std::min(e->ws_slots, R - r0);
      const int rc = drna_score_batch_device(e, m, L, d_seqs + (size_t)r0 * L, flags, d_Epf ? d_Epf + r0 : nullptr,
                                             d_Emfe ? d_Emfe + r0 : nullptr, d_mfe_ss ? d_mfe_ss + (size_t)r0 * L : nullptr,
                                             d_Ed ? d_Ed + (size_t)r0 * nt : nullptr);
      if (rc != DRNA_OK) return rc;
    }
    return DRNA_OK;
  }
  const int ld = L + 2;
  for (int k = 0; k < 2 * e->max_R; k++) e->h_status[k] = ST_OK;
  // small batches leave most CUs idle with one workgroup per fold (R = 64: 128 workgroups on 256 CUs): the MFE fold then
  // takes a main and a helper workgroup per sequence (fold_mfe_dual.hpp; the same split of the partition function did not
  // pay).  Needs 4 R <= CUs; larger batches keep the one-workgroup kernels, which saturate the chip by themselves
  // (R = 64 x L = 200: the MFE fold takes 0.50 instead of 0.59 ms, 1.65 instead of 1.81 ms with the pseudoknot re-folds; the
  // partition function running beside it loses 1 % to the busier chip's lower clock)
  // Shorter sequences do not repay the hand-shake: break-even (tools/dual_lengths.py, R = 64, with a sequence's two workgroups on
  // one XCD) at n = 120 without pseudoknot rounds (n = 130: 0.275 against 0.292 ms, n = 160: 0.336 against 0.383) and at n = 165
  // with them (their re-folds of masked sequences have little for the helper to do).
  const long long resident = (long long)e->cus * ((e->dual || e->pf_helper) && e->nt == 1024 ? pair_blocks_per_cu(e) : 1);   // workgroups the chip holds at once
  const bool use_dual = e->dual && e->lds_path && e->nt == 1024 && L <= MFE_FAST_NMAX && L > 2 * TURN + 2 && 4ll * R <= resident &&
                        (L >= (want_pk ? 170 : 125) || e->dual_force);
  if (use_dual) {
    if (e->dual_cap < R) {
      void* old[] = {e->d_dflags, e->d_xs, e->d_xa_mfe, e->d_xb_mfe};
      for (void* b : old) if (b) (void)hipFree(b);
      e->d_dflags = nullptr; e->d_xs = nullptr; e->d_xa_mfe = nullptr; e->d_xb_mfe = nullptr;
      e->dual_cap = 0;
      const size_t rows = (size_t)2 * (MFE_FAST_NMAX + 2) * XP;
      HIP_TRY(hipMalloc((void**)&e->d_dflags, (size_t)2 * R * 64 * sizeof(int)));
      HIP_TRY(hipMemset(e->d_dflags, 0, (size_t)2 * R * 64 * sizeof(int)));
      HIP_TRY(hipDeviceSynchronize());    // the memset runs on the null stream, the kernels on non-blocking streams of their own
      HIP_TRY(hipMalloc((void**)&e->d_xs, (size_t)R * 256 * sizeof(int32_t)));
      HIP_TRY(hipMalloc((void**)&e->d_xa_mfe, (size_t)R * rows * sizeof(int32_t)));
      HIP_TRY(hipMalloc((void**)&e->d_xb_mfe, (size_t)R * rows * sizeof(int32_t)));
      e->dual_cap = R;
      e->dual_epoch = 0;
    }
    if (e->dual_epoch >= DUAL_EPOCH_RESET) {                       // same reasoning as in strip_flags(): 8192 flag values per epoch
      HIP_TRY(hipMemset(e->d_dflags, 0, (size_t)2 * e->dual_cap * 64 * sizeof(int)));
      HIP_TRY(hipDeviceSynchronize());
      e->dual_epoch = 0;
      e->flag_resets++;
    }
    e->dual_epoch = (int)((unsigned)e->dual_epoch + 1u);
  }
  // longer sequences (and, as an option, short ones in small batches): the partition function by strips of columns, one
  // workgroup each (fold_pf_strip.hpp)
  const int pf_strips = want_pf ? strips_for(e, L, ld) : 0;
  const int mfe_strips = want_mfe ? strips_for(e, L, ld) : 0;
  e->last_wgs = 0;
  // partition function of a small batch: a helper workgroup per sequence on a CU that would idle takes the far multiloop split
  // points (the main workgroup's vector-memory path is what they saturate); needs room for 2 R workgroups beside the MFE fold's
  const int mfe_wgs = want_mfe ? (use_dual ? 2 * R : mfe_strips ? R * mfe_strips : R) : 0;
  const bool pf_help = want_pf && e->pf_helper && !pf_strips && e->lds_path && e->nt == 1024 && L <= PF_FAST_NMAX && L >= PF_HELPER_NMIN &&
                       2ll * R + mfe_wgs <= resident;
  if (pf_help) {
    if (e->pflags_cap < R) {
      if (e->d_pflags) (void)hipFree(e->d_pflags);
      e->d_pflags = nullptr; e->pflags_cap = 0;
      HIP_TRY(hipMalloc((void**)&e->d_pflags, (size_t)R * 64 * sizeof(int)));
      HIP_TRY(hipMemset(e->d_pflags, 0, (size_t)R * 64 * sizeof(int)));
      HIP_TRY(hipDeviceSynchronize());
      e->pflags_cap = R; e->pfh_epoch = 0;
    }
    if (e->pfh_epoch >= STRIP_EPOCH_RESET) {                       // see strip_flags(): 4096 flag values per epoch
      HIP_TRY(hipMemset(e->d_pflags, 0, (size_t)e->pflags_cap * 64 * sizeof(int)));
      HIP_TRY(hipDeviceSynchronize());
      e->pfh_epoch = 0;
      e->flag_resets++;
    }
    e->pfh_epoch++;
  }
  if (pf_strips || mfe_strips) { const int rc = strip_flags(e, mfe_strips != 0); if (rc != DRNA_OK) return rc; }
  // every stream of the engine is idle here (each call drains them before it returns), so nothing has to be fenced at the
  // start; the two folds run side by side on disjoint CUs and a launch costs ~10 us, so the one that took longer in the
  // previous call is enqueued first
  // with a helper workgroup per sequence, E(targets) is evaluated by the helpers inside the partition-function launch
  const bool ev_in_pf = want_ev && pf_help;
  auto make_eval_args = [&]() {
    EvalArgs a{};
    a.T = e->d_mfeT; a.hp_len = e->d_hp_len; a.bulge_len = e->d_bulge_len; a.int_len = e->d_int_len;
    a.seqs = d_seqs; a.pt = e->d_pt; a.L = L; a.n_targets = e->n_targets; a.Ed = d_Ed;
    return a;
  };
  // An MFE fold with pseudoknot rounds on the strip path is a chain of launches whose later links are sparse (only sequences that
  // found a pair fold again), the partition function is one dense launch.  On a chip the first fill already fills (R x strips >=
  // CUs) the partition function is therefore started when the last-but-one round has been queued: it runs beside the sparse rounds
  // instead of halving the CUs of the dense first ones (400 nt x 128: 10.6 -> 9.0 ms, x 256: 19.3 -> 16.5 ms; on a chip with idle CUs
  // it would only delay the partition function: 400 nt x 32 5.4 -> 5.7 ms).  DRNA_PF_GATE=-1 switches it off, k >= 0 forces round k.
  static const int pf_gate_env = getenv("DRNA_PF_GATE") ? atoi(getenv("DRNA_PF_GATE")) : -2;
  int pf_gate_round = -1;
  static const int pf_gate_part = getenv("DRNA_PF_GATE_PART") ? atoi(getenv("DRNA_PF_GATE_PART")) : 0;
  bool pf_gated = false;
  auto make_pf_args = [&]() {
    PfArgs a;
    a.T = e->d_pfT; a.plan = e->d_plan; a.hp_w = e->d_hp_w; a.scale = e->d_scale; a.eMLb = e->d_eMLb;
    a.seqs = d_seqs; a.L = L; a.ld = ld;
    a.ws = e->d_ws_pf; a.ws_stride = (long long)pf_ws_stride(ld);
    a.Epf = d_Epf; a.status = e->d_status + e->max_R;
    if (pf_help) { a.helper = e->helper_fault ? 2 : 1; a.hflags = e->d_pflags; a.hbase = (int)((unsigned)e->pfh_epoch << 12); }
    return a;
  };
  auto enqueue_pf = [&]() -> int {
    PfArgs a = make_pf_args();
    if (pf_gated) HIP_TRY(hipStreamWaitEvent(e->s_pf, e->ev_gate, 0));
    HIP_TRY(hipEventRecord(e->ev_p0, e->s_pf));
    e->last_wgs += pf_strips ? R * pf_strips : pf_help ? 2 * R : R;
    if (pf_strips) launch_pf_strips(e, a, R, pf_strips, 0, nullptr, e->s_pf);
    else if (pf_help)
      hipLaunchKernelGGL((pf_lds_kernel<1024, false>), dim3(pair_grid(R)), dim3(1024), 0, e->s_pf, a, ev_in_pf ? make_eval_args() : EvalArgs{}, R);
    else if (e->lds_path && e->nt == 1024 && L <= PF_FAST_NMAX)
      hipLaunchKernelGGL(pf_lds_kernel<1024>, dim3(R), dim3(1024), 0, e->s_pf, a, EvalArgs{}, R);
    else if (e->nt == 256) launch_pf<256>(a, R, e->s_pf);
    else if (e->nt == 512) launch_pf<512>(a, R, e->s_pf);
    else launch_pf<1024>(a, R, e->s_pf);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(e->ev_p1, e->s_pf));
    return DRNA_OK;
  };
  auto make_mfe_args = [&]() {
    MfeArgs a;
    a.T = e->d_mfeT; a.plan = e->d_plan; a.hp_len = e->d_hp_len; a.seqs = d_seqs; a.L = L; a.ld = ld;
    a.pk_rounds = want_pk ? 3 : 0;
    a.ws = e->d_ws_mfe; a.ws_stride = (long long)mfe_ws_stride(ld);
    a.Emfe = d_Emfe; a.ss = d_mfe_ss; a.status = e->d_status;
    return a;
  };
  auto make_dual_link = [&]() {
    DualLink lk;
    lk.flagA = e->d_dflags; lk.flagB = e->d_dflags;
    lk.xs = e->d_xs; lk.xa = e->d_xa_mfe; lk.xb = e->d_xb_mfe; lk.epoch = e->dual_epoch;
    return lk;
  };
  auto enqueue_mfe = [&]() -> int {
    MfeArgs a = make_mfe_args();
    HIP_TRY(hipEventRecord(e->ev_m0, e->s_mfe));
    e->last_wgs += use_dual ? 2 * R : mfe_strips ? R * mfe_strips : R;
    if (use_dual) {
      DualLink lk = make_dual_link();
      hipLaunchKernelGGL(mfe_dual_kernel<1024>, dim3(pair_grid(R)), dim3(1024), 0, e->s_mfe, a, lk, R);
    } else if (mfe_strips && a.pk_rounds > 0 && R >= 16 * e->mfe_split && e->mfe_split > 1) {
      // every round is a fill launch and a traceback launch (one wave per sequence, ~0.2 ms with the chip idle): the batch goes
      // in parts on two streams, so that one part's traceback runs under another part's fill
      const int np = e->mfe_split, per = ((R + np - 1) / np + 7) / 8 * 8;
      pf_gate_round = pf_gate_env >= -1 ? pf_gate_env : (pf_strips && R * mfe_strips >= e->cus && a.pk_rounds >= 2) ? a.pk_rounds - 1 : -1;
      for (int round = 0; round <= a.pk_rounds; round++) {
        for (int part = 0, r0 = 0; r0 < R; part++, r0 += per) {
          const bool gate_here = round == pf_gate_round && want_pf && part == pf_gate_part;
          launch_mfe_strips_round(e, a, std::min(per, R - r0), mfe_strips, r0, nullptr, r0, (part & 1) ? e->s_eval : e->s_mfe, round,
                                  gate_here ? e->ev_gate : nullptr);
          if (gate_here) pf_gated = true;
        }
      }
      HIP_TRY(hipEventRecord(e->ev_mfe2, e->s_eval));
      HIP_TRY(hipStreamWaitEvent(e->s_mfe, e->ev_mfe2, 0));
    } else if (mfe_strips) launch_mfe_strips(e, a, R, mfe_strips, 0, nullptr, e->s_mfe);
    else if (e->lds_path && e->nt == 1024 && L <= MFE_FAST_NMAX)
      hipLaunchKernelGGL(mfe_lds_kernel<1024>, dim3(R), dim3(1024), 0, e->s_mfe, a);
    else if (e->nt == 256) launch_mfe<256>(a, R, e->s_mfe);
    else if (e->nt == 512) launch_mfe<512>(a, R, e->s_mfe);
    else launch_mfe<1024>(a, R, e->s_mfe);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(e->ev_m1, e->s_mfe));
    return DRNA_OK;
  };
  // ---- small batches (the headline shape): both folds in ONE launch, 4 R workgroups resident side by side (fold_fused.hpp)
  e->last_fused = false;
  const bool use_fused = e->fused && use_dual && pf_help && want_mfe && want_pf && (!want_ev || ev_in_pf) && fused_grid_fits(e, R);
  if (use_fused) {
    const int grid = fused_grid(R);
    if (e->clk_cap < grid) {
      if (e->h_clk) (void)hipHostFree(e->h_clk);
      e->h_clk = nullptr; e->clk_cap = 0;
      HIP_TRY(hipHostMalloc((void**)&e->h_clk, (size_t)2 * grid * sizeof(long long), hipHostMallocMapped));
      HIP_TRY(hipHostGetDevicePointer((void**)&e->d_clk, e->h_clk, 0));
      e->clk_cap = grid;
    }
    HIP_TRY(hipEventRecord(e->ev_m0, e->s_mfe));
    hipLaunchKernelGGL(score_fused_kernel<1024>, dim3(grid), dim3(1024), 0, e->s_mfe, make_mfe_args(), make_dual_link(), make_pf_args(),
                       want_ev ? make_eval_args() : EvalArgs{}, R, e->d_clk);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(e->ev_m1, e->s_mfe));
    e->last_wgs = 4 * R;
    e->last_fused = true;
    HIP_TRY(hipStreamSynchronize(e->s_mfe));
    float tot = 0.f;
    HIP_TRY(hipEventElapsedTime(&tot, e->ev_m0, e->ev_m1));
    // per-fold times from the blocks' own clocks (100 MHz): first start of any block to the last end of the fold's blocks
    long long t0 = 0, end_mfe = 0, end_pf = 0;
    bool first = true;
    for (int b = 0; b < grid; b++) {
      int r, role;
      fused_block_role(b, r, role, grid);
      if (r >= R) continue;
      const long long s0 = e->h_clk[2 * b], s1 = e->h_clk[2 * b + 1];
      if (first || s0 < t0) t0 = s0;
      first = false;
      if (role <= ROLE_MFE_HELPER) { if (s1 > end_mfe) end_mfe = s1; } else if (s1 > end_pf) end_pf = s1;
    }
    e->timing[0] = (float)((end_mfe - t0) * 1e-5); e->timing[1] = (float)((end_pf - t0) * 1e-5);
    e->timing[2] = 0.f; e->timing[3] = tot;
    for (int k = 0; k < 4; k++) e->timing_sum[k] += e->timing[k];
    e->timing_sum[4] += 1.0;
  }
  const bool mfe_first = want_mfe && (!want_pf || e->timing[0] > e->timing[1]);
  // the evaluation kernel (~20 us) rides in FRONT of the shorter fold on that fold's stream: one stream less to drain at the end
  hipStream_t s_ev = e->s_eval;
  if (want_ev && want_mfe && want_pf) s_ev = mfe_first ? e->s_pf : e->s_mfe;
  auto enqueue_eval = [&]() -> int {
    if (ev_in_pf) return DRNA_OK;
    EvalArgs a = make_eval_args();
    HIP_TRY(hipEventRecord(e->ev_e0, s_ev));
    hipLaunchKernelGGL(eval_kernel, dim3(R * e->n_targets), dim3(WAVE), 0, s_ev, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(e->ev_e1, s_ev));
    return DRNA_OK;
  };
  static const bool host_profile = getenv("DRNA_HOST_PROFILE") != nullptr;      // diagnostics: where a call's host time goes (stderr, every 64 calls)
  static double hp_acc[4] = {0, 0, 0, 0};
  static int hp_n = 0;
  auto hp_now = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; };
  const double hp0 = host_profile ? hp_now() : 0.0;
  double hp1 = 0.0, hp2 = 0.0;
  if (!use_fused) {
  if (mfe_first) { const int rc = enqueue_mfe(); if (rc != DRNA_OK) return rc; }
  if (want_ev && s_ev == e->s_pf) { const int rc = enqueue_eval(); if (rc != DRNA_OK) return rc; }
  if (want_pf) { const int rc = enqueue_pf(); if (rc != DRNA_OK) return rc; }
  if (want_ev && s_ev == e->s_mfe) { const int rc = enqueue_eval(); if (rc != DRNA_OK) return rc; }
  if (want_mfe && !mfe_first) { const int rc = enqueue_mfe(); if (rc != DRNA_OK) return rc; }
  if (want_ev && s_ev == e->s_eval) { const int rc = enqueue_eval(); if (rc != DRNA_OK) return rc; }
  // join on the host: the streams are drained one after the other (a device-side join -- stream-wait-event packets
  // plus an end marker -- costs ~15 us after the last kernel); "total" = first start event to the latest end event
  auto drain = [&](hipStream_t st) -> hipError_t { return hipStreamSynchronize(st); };
  if (host_profile) hp1 = hp_now();
  if (want_ev && !ev_in_pf && s_ev == e->s_eval) HIP_TRY(drain(e->s_eval));
  if (mfe_first && want_pf) HIP_TRY(drain(e->s_pf));
  if (want_mfe) HIP_TRY(drain(e->s_mfe));
  if (!mfe_first && want_pf) HIP_TRY(drain(e->s_pf));
  if (host_profile) hp2 = hp_now();
  e->timing[0] = e->timing[1] = e->timing[2] = 0.f;
  if (want_mfe) HIP_TRY(hipEventElapsedTime(&e->timing[0], e->ev_m0, e->ev_m1));
  if (want_pf) HIP_TRY(hipEventElapsedTime(&e->timing[1], e->ev_p0, e->ev_p1));
  if (want_ev && !ev_in_pf) HIP_TRY(hipEventElapsedTime(&e->timing[2], e->ev_e0, e->ev_e1));
  {
    hipEvent_t first = mfe_first ? e->ev_m0 : want_pf ? e->ev_p0 : e->ev_e0;
    float t = 0.f, tot = 0.f;
    if (want_mfe) { HIP_TRY(hipEventElapsedTime(&t, first, e->ev_m1)); tot = t > tot ? t : tot; }
    if (want_pf) { HIP_TRY(hipEventElapsedTime(&t, first, e->ev_p1)); tot = t > tot ? t : tot; }
    if (want_ev && !ev_in_pf) { HIP_TRY(hipEventElapsedTime(&t, first, e->ev_e1)); tot = t > tot ? t : tot; }
    e->timing[3] = tot;
    for (int k = 0; k < 4; k++) e->timing_sum[k] += e->timing[k];
    e->timing_sum[4] += 1.0;
  }
  if (host_profile) {
    const double hp3 = hp_now();
    hp_acc[0] += hp1 - hp0; hp_acc[1] += hp2 - hp1; hp_acc[2] += hp3 - hp2; hp_acc[3] += e->timing[3] * 1e3;
    if (++hp_n == 64) {
      fprintf(stderr, "drna_score_batch_device: host us per call: enqueue %.1f, drain %.1f (device %.1f), event queries %.1f\n",
              hp_acc[0] / 64, hp_acc[1] / 64, hp_acc[3] / 64, hp_acc[2] / 64);
      hp_n = 0; hp_acc[0] = hp_acc[1] = hp_acc[2] = hp_acc[3] = 0;
    }
  }
  }   // !use_fused
  for (int r = 0; r < R; r++) {
    const int sm = want_mfe ? e->h_status[r] : ST_OK, sp = want_pf ? e->h_status[e->max_R + r] : ST_OK;
    const int st = sm != ST_OK ? sm : sp;
    if (st == ST_OK) continue;
    char buf[160];
    if (st == ST_BAD_CHAR) {
      snprintf(buf, sizeof buf, "sequence %d holds a character other than A C G U T", r);
      e->err = buf;
      return DRNA_ERR_SEQUENCE;
    }
    if (st == ST_PF_RANGE) {
      snprintf(buf, sizeof buf, "sequence %d: partition function left the fp64 range (pf_scale too small/large)", r);
      e->err = buf;
      return DRNA_ERR_PF_RANGE;
    }
    if (st == ST_SYNC && !e->in_fallback) {
      // HIP promises no dispatch order: a multi-workgroup fold whose bounded wait expired is not an error of the batch -- the
      // whole call is redone with one workgroup per fold (general / LDS-resident kernels), which need nobody
      const int s_strips = e->strips;
      const bool s_dual = e->dual, s_help = e->pf_helper;
      e->in_fallback = true; e->strips = 0; e->dual = false; e->pf_helper = false;
      for (int k = 0; k < 4; k++) e->timing_sum[k] -= e->timing[k];        // the lost attempt (up to the wait budget) is not a kernel time
      e->timing_sum[4] -= 1.0;
      const int rc = drna_score_batch_device(e, R, L, d_seqs, flags, d_Epf, d_Emfe, d_mfe_ss, d_Ed);
      e->strips = s_strips; e->dual = s_dual; e->pf_helper = s_help; e->in_fallback = false;
      e->sync_fallbacks++;
      if (++e->fallback_streak >= 3) { e->fallback_streak = 0; e->solo_left = SOLO_CALLS; }
      return rc;
    }
    if (st == ST_SYNC) {
      snprintf(buf, sizeof buf, "sequence %d: the workgroups of the fold lost each other (a wait expired)", r);
      if (e->d_sdbg) {
        std::vector<int> dbg((size_t)2 * e->max_R * 8);
        (void)hipMemcpy(dbg.data(), e->d_sdbg, dbg.size() * sizeof(int), hipMemcpyDeviceToHost);
        for (size_t k = 0; k < dbg.size(); k += 8)
          if (dbg[k]) fprintf(stderr, "strip debug slot %zu: strip %d step %d saw flag %d (base %d: %d) block %d n %d\n", k / 8, dbg[k] - 1, dbg[k + 1],
                              dbg[k + 2], dbg[k + 3], dbg[k + 2] - dbg[k + 3], dbg[k + 4], dbg[k + 5]);
      }
    }
    else snprintf(buf, sizeof buf, "sequence %d: traceback could not reproduce a table value", r);
    e->err = buf;
    return DRNA_ERR_INTERNAL;
  }
  if (!e->in_fallback) e->fallback_streak = 0;          // nobody lost anybody
  return DRNA_OK;
}

extern "C" int drna_score_batch(drna_engine* e, int R, int L, const char* seqs, uint32_t flags, double* Epf,
                                int32_t* Emfe, char* mfe_ss, int32_t* Ed) {
  if (!e) return DRNA_ERR_ARG;
  const bool want_pf = flags & DRNA_NEED_PF, want_mfe = flags & (DRNA_NEED_MFE | DRNA_NEED_PK),
             want_ev = flags & DRNA_NEED_EVAL;
  if (R < 1 || R > e->max_R || L < 1 || L > e->max_L || !seqs || (want_pf && !Epf) || (want_mfe && (!Emfe || !mfe_ss)) ||
      (want_ev && !Ed)) {
    e->err = "drna_score_batch: bad argument (R, L within the engine's limits; output pointers for every requested flag)";
    return DRNA_ERR_ARG;
  }
  HIP_TRY(hipSetDevice(e->device));
  if (e->zero_copy) {
    const size_t ned = (size_t)R * (e->n_targets > 0 ? e->n_targets : 1);
    if (want_ev && e->hm_Ed_cap < ned) {
      if (e->hm_Ed) (void)hipHostFree(e->hm_Ed);
      e->hm_Ed = nullptr; e->hm_Ed_cap = 0;
      const size_t cap = (size_t)e->max_R * (e->n_targets > 0 ? e->n_targets : 1);
      HIP_TRY(hipHostMalloc((void**)&e->hm_Ed, cap * sizeof(int32_t), hipHostMallocMapped));
      HIP_TRY(hipHostGetDevicePointer((void**)&e->dm_Ed, e->hm_Ed, 0));
      e->hm_Ed_cap = cap;
    }
    std::memcpy(e->hm_seqs, seqs, (size_t)R * L);
    int rc = drna_score_batch_device(e, R, L, e->dm_seqs, flags, e->dm_Epf, e->dm_Emfe, e->dm_ss, e->dm_Ed);
    if (rc != DRNA_OK) return rc;
    if (want_pf) std::memcpy(Epf, e->hm_Epf, (size_t)R * sizeof(double));
    if (want_mfe) {
      std::memcpy(Emfe, e->hm_Emfe, (size_t)R * sizeof(int32_t));
      std::memcpy(mfe_ss, e->hm_ss, (size_t)R * L);
    }
    if (want_ev) std::memcpy(Ed, e->hm_Ed, ned * sizeof(int32_t));
    return DRNA_OK;
  }
  HIP_TRY(hipMemcpy(e->d_seqs, seqs, (size_t)R * L, hipMemcpyHostToDevice));
  int rc = drna_score_batch_device(e, R, L, e->d_seqs, flags, e->d_Epf, e->d_Emfe, e->d_ss, e->d_Ed);
  if (rc != DRNA_OK) return rc;
  if (want_pf) HIP_TRY(hipMemcpy(Epf, e->d_Epf, (size_t)R * sizeof(double), hipMemcpyDeviceToHost));
  if (want_mfe) {
    HIP_TRY(hipMemcpy(Emfe, e->d_Emfe, (size_t)R * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(mfe_ss, e->d_ss, (size_t)R * L, hipMemcpyDeviceToHost));
  }
  if (want_ev) HIP_TRY(hipMemcpy(Ed, e->d_Ed, (size_t)R * e->n_targets * sizeof(int32_t), hipMemcpyDeviceToHost));
  return DRNA_OK;
}

extern "C" int drna_last_timing(const drna_engine* e, float out[4]) {
  if (!e || !out) return DRNA_ERR_ARG;
  for (int k = 0; k < 4; k++) out[k] = e->timing[k];
  return DRNA_OK;
}

extern "C" int drna_timing_sums(drna_engine* e, double out[5], int reset) {
  if (!e) return DRNA_ERR_ARG;
  if (out) for (int k = 0; k < 5; k++) out[k] = e->timing_sum[k];
  if (reset) for (int k = 0; k < 5; k++) e->timing_sum[k] = 0.0;
  return DRNA_OK;
}

extern "C" int drna_info(const drna_engine* e, int64_t out[6]) {
  if (!e || !out) return DRNA_ERR_ARG;
  out[0] = e->device; out[1] = e->max_R; out[2] = e->max_L; out[3] = e->nt; out[4] = e->cus; out[5] = (int64_t)e->ws_bytes;
  return DRNA_OK;
}

// ---------------------------------------------------------------- ensemble defect (inside + outside recursion)

template <int NT>
static void launch_outside(const OutArgs& a, int R, hipStream_t s) {
  hipLaunchKernelGGL(outside_kernel<NT>, dim3(R), dim3(NT), 0, s, a);
}

extern "C" int drna_ensemble_defect_batch_device(drna_engine* e, int R, int L, const char* d_seqs, double* d_edef,
                                                 double* d_bpp) {
  if (!e) return DRNA_ERR_ARG;
  if (R < 1 || R > e->max_R || L < 1 || L > e->max_L || !d_seqs || !d_edef) {
    e->err = "drna_ensemble_defect_batch: bad argument (R, L within the engine's limits; seqs and edef required)";
    return DRNA_ERR_ARG;
  }
  if (e->n_targets < 1 || e->L_targets != L) {
    e->err = "drna_ensemble_defect_batch: needs drna_set_targets() with the same L (targets[0] is the reference structure)";
    return DRNA_ERR_ARG;
  }
  if (R > e->ws_slots) {
    // more sequences than the workspaces hold (DRNA_WS_GB): one sub-batch of ws_slots after the other, as drna_score_batch_device does
    for (int r0 = 0; r0 < R; r0 += e->ws_slots) {
      const int m = std::min(e->ws_slots, R - r0);
      const int rc = drna_ensemble_defect_batch_device(e, m, L, d_seqs + (size_t)r0 * L, d_edef + r0,
                                                       d_bpp ? d_bpp + (size_t)r0 * (L + 1) * (L + 1) : nullptr);
      if (rc != DRNA_OK) return rc;
    }
    return DRNA_OK;
  }
  HIP_TRY(hipSetDevice(e->device));
  const int ldmax = e->max_L + 2, ld = L + 2;
  if (!e->d_ws_out) {
    HIP_TRY(hipMalloc((void**)&e->d_ws_out, (size_t)outside_ws_stride(ldmax) * sizeof(double) * e->ws_slots));
    e->ws_bytes += (size_t)outside_ws_stride(ldmax) * sizeof(double) * e->ws_slots;
  }
  for (int k = 0; k < R; k++) e->h_status[e->max_R + k] = ST_OK;
  // the general inside kernel: it leaves qb / qm / qm1 in the workspace (the LDS kernel keeps only rings)
  PfArgs a;
  a.T = e->d_pfT; a.plan = e->d_plan; a.hp_w = e->d_hp_w; a.scale = e->d_scale; a.eMLb = e->d_eMLb;
  a.seqs = d_seqs; a.L = L; a.ld = ld;
  a.ws = e->d_ws_pf; a.ws_stride = (long long)pf_ws_stride(ld);
  a.Epf = e->d_Epf; a.status = e->d_status + e->max_R;
  a.q5_stride = outside_ws_stride(ld);
  a.q5out = e->d_ws_out + (size_t)4 * ld * ld;
  OutArgs o;
  o.T = e->d_pfT; o.plan = e->d_plan; o.scale = e->d_scale; o.eMLb = e->d_eMLb;
  o.seqs = d_seqs; o.L = L; o.ld = ld;
  o.ws = e->d_ws_pf; o.ws_stride = a.ws_stride;
  o.wo = e->d_ws_out; o.wo_stride = outside_ws_stride(ld);
  o.pt = e->d_pt; o.edef = d_edef; o.bpp = d_bpp; o.pf_status = e->d_status + e->max_R;
  HIP_TRY(hipEventRecord(e->ev_o0, e->s_pf));
  if (e->nt == 256) launch_pf<256>(a, R, e->s_pf);
  else if (e->nt == 512) launch_pf<512>(a, R, e->s_pf);
  else launch_pf<1024>(a, R, e->s_pf);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(e->ev_o1, e->s_pf));
  if (e->nt == 256) launch_outside<256>(o, R, e->s_pf);
  else if (e->nt == 512) launch_outside<512>(o, R, e->s_pf);
  else launch_outside<1024>(o, R, e->s_pf);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(e->ev_o2, e->s_pf));
  HIP_TRY(hipStreamSynchronize(e->s_pf));
  HIP_TRY(hipEventElapsedTime(&e->timing_edef[0], e->ev_o0, e->ev_o1));
  HIP_TRY(hipEventElapsedTime(&e->timing_edef[1], e->ev_o1, e->ev_o2));
  for (int r = 0; r < R; r++) {
    const int st = e->h_status[e->max_R + r];
    if (st == ST_OK) continue;
    char buf[160];
    if (st == ST_BAD_CHAR) {
      snprintf(buf, sizeof buf, "sequence %d holds a character other than A C G U T", r);
      e->err = buf;
      return DRNA_ERR_SEQUENCE;
    }
    snprintf(buf, sizeof buf, "sequence %d: partition function left the fp64 range (pf_scale too small/large)", r);
    e->err = buf;
    return DRNA_ERR_PF_RANGE;
  }
  return DRNA_OK;
}

extern "C" int drna_ensemble_defect_batch(drna_engine* e, int R, int L, const char* seqs, double* edef, double* bpp) {
  if (!e) return DRNA_ERR_ARG;
  if (R < 1 || R > e->max_R || L < 1 || L > e->max_L || !seqs || !edef) {
    e->err = "drna_ensemble_defect_batch: bad argument (R, L within the engine's limits; seqs and edef required)";
    return DRNA_ERR_ARG;
  }
  HIP_TRY(hipSetDevice(e->device));
  if (!e->d_edef) HIP_TRY(hipMalloc((void**)&e->d_edef, (size_t)e->max_R * sizeof(double)));
  HIP_TRY(hipMemcpy(e->d_seqs, seqs, (size_t)R * L, hipMemcpyHostToDevice));
  double* d_bpp = nullptr;
  const size_t nb = (size_t)R * (L + 1) * (L + 1) * sizeof(double);
  if (bpp) {
    HIP_TRY(hipMalloc((void**)&d_bpp, nb));
    hipError_t z = hipMemset(d_bpp, 0, nb);
    if (z != hipSuccess) { (void)hipFree(d_bpp); e->err = "hipMemset(bpp)"; return DRNA_ERR_DEVICE; }
  }
  int rc = drna_ensemble_defect_batch_device(e, R, L, e->d_seqs, e->d_edef, d_bpp);
  if (rc == DRNA_OK) {
    hipError_t c1 = hipMemcpy(edef, e->d_edef, (size_t)R * sizeof(double), hipMemcpyDeviceToHost);
    hipError_t c2 = bpp ? hipMemcpy(bpp, d_bpp, nb, hipMemcpyDeviceToHost) : hipSuccess;
    if (c1 != hipSuccess || c2 != hipSuccess) { e->err = "hipMemcpy(edef/bpp)"; rc = DRNA_ERR_DEVICE; }
  }
  if (d_bpp) (void)hipFree(d_bpp);
  return rc;
}

extern "C" int drna_last_edef_timing(const drna_engine* e, float out[2]) {
  if (!e || !out) return DRNA_ERR_ARG;
  out[0] = e->timing_edef[0]; out[1] = e->timing_edef[1];
  return DRNA_OK;
}

// ---------------------------------------------------------------- ragged batches (sequences of different lengths)

extern "C" int drna_set_targets_ragged(drna_engine* e, int n_targets, const int32_t* lens, const char* targets) {
  if (!e) return DRNA_ERR_ARG;
  if (n_targets < 1 || !lens || !targets) { e->err = "drna_set_targets_ragged: bad argument"; return DRNA_ERR_ARG; }
  std::vector<int> off(n_targets);
  size_t total = 0, chars = 0;
  for (int t = 0; t < n_targets; t++) {
    if (lens[t] < 1 || lens[t] > e->max_L) { e->err = "drna_set_targets_ragged: structure length outside [1, max_L]"; return DRNA_ERR_ARG; }
    off[t] = (int)total;
    total += (size_t)lens[t] + 2;
  }
  std::vector<short> pt(total, 0);
  std::vector<int> stk;
  for (int t = 0; t < n_targets; t++) {
    stk.clear();
    const char* s = targets + chars;
    short* p = pt.data() + off[t];
    for (int i = 1; i <= lens[t]; i++) {
      if (s[i - 1] == '(') stk.push_back(i);
      else if (s[i - 1] == ')') {
        if (stk.empty()) { e->err = "drna_set_targets_ragged: unbalanced ')'"; return DRNA_ERR_STRUCTURE; }
        const int o = stk.back(); stk.pop_back();
        p[o] = (short)i; p[i] = (short)o;
      }
    }
    if (!stk.empty()) { e->err = "drna_set_targets_ragged: unbalanced '('"; return DRNA_ERR_STRUCTURE; }
    chars += (size_t)lens[t];
  }
  HIP_TRY(hipSetDevice(e->device));
  if (e->d_rpt) { (void)hipFree(e->d_rpt); e->d_rpt = nullptr; }
  if (e->d_rpt_off) { (void)hipFree(e->d_rpt_off); e->d_rpt_off = nullptr; }
  HIP_TRY(upload(&e->d_rpt, pt.data(), pt.size()));
  HIP_TRY(upload(&e->d_rpt_off, off.data(), off.size()));
  e->rt_len.assign(lens, lens + n_targets);
  return DRNA_OK;
}

extern "C" int drna_score_ragged(drna_engine* e, int R, const int32_t* lens, const char* seqs, const int32_t* target_of,
                                 uint32_t flags, double* Epf, int32_t* Emfe, char* mfe_ss, int32_t* Ed) {
  if (!e) return DRNA_ERR_ARG;
  const bool want_pf = flags & DRNA_NEED_PF, want_mfe = flags & (DRNA_NEED_MFE | DRNA_NEED_PK),
             want_pk = flags & DRNA_NEED_PK, want_ev = flags & DRNA_NEED_EVAL;
  if (R < 1 || R > e->max_R || !lens || !seqs || (want_pf && !Epf) || (want_mfe && (!Emfe || !mfe_ss)) ||
      (want_ev && (!Ed || !target_of))) {
    e->err = "drna_score_ragged: bad argument (R within the engine's limit; output pointers for every requested flag)";
    return DRNA_ERR_ARG;
  }
  e->cur_with_pf = want_pf;
  if (!e->in_fallback && e->solo_left > 0) {             // (see drna_score_batch_device)
    const int s_strips = e->strips;
    e->in_fallback = true; e->strips = 0;
    const int rc = drna_score_ragged(e, R, lens, seqs, target_of, flags, Epf, Emfe, mfe_ss, Ed);
    e->strips = s_strips; e->in_fallback = false;
    e->solo_left--;
    return rc;
  }
  // The batch is folded in SORTED order, longest first: position q on the device is sequence order[q] of the caller.  Then
  // (a) the three classes -- strips (n > 200), general kernels, LDS-resident kernels (n <= 200) -- are ranges of q, and (b) a
  // batch larger than the workspaces (ws_slots sequences, DRNA_WS_GB) goes through them in CHUNKS of consecutive q: chunk c
  // uses slot q - c0, i.e. a workspace pointer moved back by c0 slots, its launches queue behind the previous chunk's on the
  // same streams (the MFE stream owns the MFE workspace, the partition-function stream the other), and nothing waits in between.
  // descriptors: len | off | target_of | index lists (q values) of the LDS-resident kernels | - | strip kernels | general kernels
  std::vector<int> order(R);
  std::iota(order.begin(), order.end(), 0);
  size_t total = 0;
  for (int r = 0; r < R; r++) {
    if (lens[r] < 1 || lens[r] > e->max_L) { e->err = "drna_score_ragged: sequence length outside [1, max_L]"; return DRNA_ERR_ARG; }
    total += (size_t)lens[r];
    if (want_ev) {
      const int t = target_of[r];
      if (t < 0 || t >= (int)e->rt_len.size() || e->rt_len[t] != lens[r]) {
        e->err = "drna_score_ragged: target_of[r] must name a structure of drna_set_targets_ragged() with the sequence's length";
        return DRNA_ERR_ARG;
      }
    }
  }
  if (total > (size_t)e->max_R * e->max_L) { e->err = "drna_score_ragged: more nucleotides than max_R * max_L"; return DRNA_ERR_ARG; }
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lens[a] > lens[b]; });   // longest first
  std::vector<size_t> src_off(R);
  { size_t o = 0; for (int r = 0; r < R; r++) { src_off[r] = o; o += (size_t)lens[r]; } }
  std::vector<int> h((size_t)7 * R);
  std::vector<char> sorted_seqs(total);
  int* off = h.data() + R;
  {
    size_t o = 0;
    for (int q = 0; q < R; q++) {
      const int r = order[q];
      h[q] = lens[r];
      off[q] = (int)o;
      memcpy(sorted_seqs.data() + o, seqs + src_off[r], (size_t)lens[r]);
      o += (size_t)lens[r];
      if (want_ev) h[(size_t)2 * R + q] = target_of[r];
    }
  }
  const bool fast_ok = e->lds_path && e->nt == 1024;
  int nA = 0, nC = 0, nD = 0;
  int *idxA = h.data() + (size_t)3 * R, *idxC = h.data() + (size_t)5 * R, *idxD = h.data() + (size_t)6 * R;
  const int ld = e->max_L + 2;
  for (int q = 0; q < R; q++) {
    const int len = h[q];
    if (fast_ok && len <= MFE_FAST_NMAX && len <= PF_FAST_NMAX) idxA[nA++] = q;
    else if (strips_for(e, len, ld) && len > PF_FAST_NMAX) idxC[nC++] = q;
    else idxD[nD++] = q;
  }
  HIP_TRY(hipSetDevice(e->device));
  if (!e->d_rg) HIP_TRY(hipMalloc((void**)&e->d_rg, (size_t)7 * e->max_R * sizeof(int)));
  HIP_TRY(hipMemcpy(e->d_rg, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->d_seqs, sorted_seqs.data(), total, hipMemcpyHostToDevice));
  if (want_ev && R > 0) {
    if (!e->d_Ed) HIP_TRY(hipMalloc((void**)&e->d_Ed, (size_t)e->max_R * std::max(1, e->n_targets) * sizeof(int32_t)));
  }
  for (int k = 0; k < 2 * e->max_R; k++) e->h_status[k] = ST_OK;
  if ((want_pf || want_mfe) && nC) { const int rc = strip_flags(e, want_mfe); if (rc != DRNA_OK) return rc; }
  Ragged rg;
  rg.len = e->d_rg; rg.off = e->d_rg + R;
  // the part of an index list (ascending q) that falls into the chunk [c0, c1)
  auto part = [&](const int* idx, int cnt, int c0, int c1, int& first, int& num) {
    first = (int)(std::lower_bound(idx, idx + cnt, c0) - idx);
    num = (int)(std::lower_bound(idx, idx + cnt, c1) - idx) - first;
  };
  const int slots = e->ws_slots;
  HIP_TRY(hipEventRecord(e->ev_start, e->s_mfe));
  HIP_TRY(hipStreamWaitEvent(e->s_pf, e->ev_start, 0));
  HIP_TRY(hipStreamWaitEvent(e->s_eval, e->ev_start, 0));
  if (want_mfe) {
    MfeArgs a;
    a.T = e->d_mfeT; a.plan = e->d_plan; a.hp_len = e->d_hp_len; a.seqs = e->d_seqs; a.L = 0; a.ld = ld;
    a.pk_rounds = want_pk ? 3 : 0;
    a.ws_stride = (long long)mfe_ws_stride(ld);
    a.Emfe = e->d_Emfe; a.ss = e->d_ss; a.status = e->d_status;
    a.rg = rg;
    HIP_TRY(hipEventRecord(e->ev_m0, e->s_mfe));
    for (int c0 = 0; c0 < R; c0 += slots) {
      const int c1 = std::min(R, c0 + slots);
      a.ws = e->d_ws_mfe - (long long)c0 * a.ws_stride;           // slot of sequence q: q - c0
      int f, m;
      part(idxC, nC, c0, c1, f, m);
      a.rg.idx = nullptr;
      for (int k = f; k < f + m;) {                 // the long sequences first: strip kernels, one launch per number of strips
        const int S = strips_for(e, h[idxC[k]], ld);
        int k2 = k;
        while (k2 < f + m && strips_for(e, h[idxC[k2]], ld) == S) k2++;
        launch_mfe_strips(e, a, k2 - k, S, k, e->d_rg + (size_t)5 * R + k, e->s_mfe);
        k = k2;
      }
      part(idxD, nD, c0, c1, f, m);
      if (m) {
        a.rg.idx = e->d_rg + (size_t)6 * R + f;
        if (e->nt == 256) launch_mfe<256>(a, m, e->s_mfe);
        else if (e->nt == 512) launch_mfe<512>(a, m, e->s_mfe);
        else launch_mfe<1024>(a, m, e->s_mfe);
      }
      part(idxA, nA, c0, c1, f, m);
      if (m) {
        a.rg.idx = e->d_rg + (size_t)3 * R + f;
        hipLaunchKernelGGL(mfe_lds_kernel<1024>, dim3(m), dim3(1024), 0, e->s_mfe, a);
      }
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(e->ev_m1, e->s_mfe));
  }
  if (want_pf) {
    PfArgs a;
    a.T = e->d_pfT; a.plan = e->d_plan; a.hp_w = e->d_hp_w; a.scale = e->d_scale; a.eMLb = e->d_eMLb;
    a.seqs = e->d_seqs; a.L = 0; a.ld = ld;
    a.ws_stride = (long long)pf_ws_stride(ld);
    a.Epf = e->d_Epf; a.status = e->d_status + e->max_R;
    a.rg = rg;
    HIP_TRY(hipEventRecord(e->ev_p0, e->s_pf));
    for (int c0 = 0; c0 < R; c0 += slots) {
      const int c1 = std::min(R, c0 + slots);
      a.ws = e->d_ws_pf - (long long)c0 * a.ws_stride;
      int f, m;
      part(idxC, nC, c0, c1, f, m);
      a.rg.idx = nullptr;
      for (int k = f; k < f + m;) {                 // the strip kernel: one launch per number of strips, most strips first
        const int S = strips_for(e, h[idxC[k]], ld);
        int k2 = k;
        while (k2 < f + m && strips_for(e, h[idxC[k2]], ld) == S) k2++;
        launch_pf_strips(e, a, k2 - k, S, k, e->d_rg + (size_t)5 * R + k, e->s_pf);
        k = k2;
      }
      part(idxD, nD, c0, c1, f, m);
      if (m) {
        a.rg.idx = e->d_rg + (size_t)6 * R + f;
        if (e->nt == 256) launch_pf<256>(a, m, e->s_pf);
        else if (e->nt == 512) launch_pf<512>(a, m, e->s_pf);
        else launch_pf<1024>(a, m, e->s_pf);
      }
      part(idxA, nA, c0, c1, f, m);
      if (m) {
        a.rg.idx = e->d_rg + (size_t)3 * R + f;
        hipLaunchKernelGGL(pf_lds_kernel<1024>, dim3(m), dim3(1024), 0, e->s_pf, a, EvalArgs{}, m);
      }
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(e->ev_p1, e->s_pf));
    HIP_TRY(hipStreamWaitEvent(e->s_mfe, e->ev_p1, 0));
  }
  if (want_ev) {
    EvalArgs a;
    a.T = e->d_mfeT; a.hp_len = e->d_hp_len; a.bulge_len = e->d_bulge_len; a.int_len = e->d_int_len;
    a.seqs = e->d_seqs; a.pt = e->d_rpt; a.L = 0; a.n_targets = 1; a.Ed = e->d_Ed;
    a.rg = rg; a.target_of = e->d_rg + (size_t)2 * R; a.pt_off = e->d_rpt_off;
    HIP_TRY(hipEventRecord(e->ev_e0, e->s_eval));
    hipLaunchKernelGGL(eval_kernel, dim3(R), dim3(WAVE), 0, e->s_eval, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(e->ev_e1, e->s_eval));
    HIP_TRY(hipStreamWaitEvent(e->s_mfe, e->ev_e1, 0));
  }
  HIP_TRY(hipEventRecord(e->ev_end, e->s_mfe));
  HIP_TRY(hipStreamSynchronize(e->s_mfe));
  e->timing[0] = e->timing[1] = e->timing[2] = 0.f;
  if (want_mfe) HIP_TRY(hipEventElapsedTime(&e->timing[0], e->ev_m0, e->ev_m1));
  if (want_pf) HIP_TRY(hipEventElapsedTime(&e->timing[1], e->ev_p0, e->ev_p1));
  if (want_ev) HIP_TRY(hipEventElapsedTime(&e->timing[2], e->ev_e0, e->ev_e1));
  HIP_TRY(hipEventElapsedTime(&e->timing[3], e->ev_start, e->ev_end));
  for (int q = 0; q < R; q++) {
    const int sm = want_mfe ? e->h_status[q] : ST_OK, sp = want_pf ? e->h_status[e->max_R + q] : ST_OK;
    const int st = sm != ST_OK ? sm : sp;
    if (st == ST_OK) continue;
    if (st == ST_SYNC && !e->in_fallback) {               // (see drna_score_batch_device)
      const int s_strips = e->strips;
      e->in_fallback = true; e->strips = 0;
      const int rc = drna_score_ragged(e, R, lens, seqs, target_of, flags, Epf, Emfe, mfe_ss, Ed);
      e->strips = s_strips; e->in_fallback = false;
      e->sync_fallbacks++;
      if (++e->fallback_streak >= 3) { e->fallback_streak = 0; e->solo_left = SOLO_CALLS; }
      return rc;
    }
    char buf[160];
    snprintf(buf, sizeof buf, st == ST_BAD_CHAR ? "sequence %d holds a character other than A C G U T"
                              : st == ST_PF_RANGE ? "sequence %d: partition function left the fp64 range"
                              : st == ST_SYNC ? "sequence %d: the two workgroups of the fold lost each other (wait expired)"
                                                  : "sequence %d: traceback could not reproduce a table value", order[q]);
    e->err = buf;
    return st == ST_BAD_CHAR ? DRNA_ERR_SEQUENCE : st == ST_PF_RANGE ? DRNA_ERR_PF_RANGE : DRNA_ERR_INTERNAL;
  }
  if (!e->in_fallback) e->fallback_streak = 0;
  // results back in the caller's order
  if (want_pf) {
    std::vector<double> t(R);
    HIP_TRY(hipMemcpy(t.data(), e->d_Epf, (size_t)R * sizeof(double), hipMemcpyDeviceToHost));
    for (int q = 0; q < R; q++) Epf[order[q]] = t[q];
  }
  if (want_mfe) {
    std::vector<int32_t> t(R);
    HIP_TRY(hipMemcpy(t.data(), e->d_Emfe, (size_t)R * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int q = 0; q < R; q++) Emfe[order[q]] = t[q];
    HIP_TRY(hipMemcpy(sorted_seqs.data(), e->d_ss, total, hipMemcpyDeviceToHost));
    for (int q = 0; q < R; q++) memcpy(mfe_ss + src_off[order[q]], sorted_seqs.data() + off[q], (size_t)h[q]);
  }
  if (want_ev) {
    std::vector<int32_t> t(R);
    HIP_TRY(hipMemcpy(t.data(), e->d_Ed, (size_t)R * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int q = 0; q < R; q++) Ed[order[q]] = t[q];
  }
  return DRNA_OK;
}

// ---------------------------------------------------------------- second-best structure energy (-nd on)

template <int NT>
static void launch_subopt(const SubArgs& a, int R, hipStream_t s) {
  hipLaunchKernelGGL(subopt_kernel<NT>, dim3(R), dim3(NT), 0, s, a);
}

extern "C" int drna_subopt_energy_batch(drna_engine* e, int R, int L, const char* seqs, int32_t* E2, int32_t* E12) {
  if (!e) return DRNA_ERR_ARG;
  if (R < 1 || R > e->max_R || L < 1 || L > e->max_L || !seqs || !E2) {
    e->err = "drna_subopt_energy_batch: bad argument (R, L within the engine's limits; seqs and E2 required)";
    return DRNA_ERR_ARG;
  }
  if (R > e->ws_slots) { e->err = "drna_subopt_energy_batch: batch larger than the workspace (raise DRNA_WS_GB or split the batch)"; return DRNA_ERR_ARG; }
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipMemcpy(e->d_seqs, seqs, (size_t)R * L, hipMemcpyHostToDevice));
  const int ld = L + 2;
  for (int k = 0; k < e->max_R; k++) e->h_status[k] = ST_OK;
  SubArgs a;
  a.T = e->d_mfeT; a.plan = e->d_plan; a.hp_len = e->d_hp_len; a.seqs = e->d_seqs; a.L = L; a.ld = ld;
  a.ws = reinterpret_cast<int32_t*>(e->d_ws_pf); a.ws_stride = 2 * (long long)pf_ws_stride(ld);   // int32 units of the PF workspace
  a.E2 = e->d_Emfe; a.E12 = reinterpret_cast<int32_t*>(e->d_Epf); a.status = e->d_status;
  HIP_TRY(hipEventRecord(e->ev_m0, e->s_mfe));
  if (e->nt == 256) launch_subopt<256>(a, R, e->s_mfe);
  else if (e->nt == 512) launch_subopt<512>(a, R, e->s_mfe);
  else launch_subopt<1024>(a, R, e->s_mfe);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(e->ev_m1, e->s_mfe));
  HIP_TRY(hipStreamSynchronize(e->s_mfe));
  HIP_TRY(hipEventElapsedTime(&e->timing[0], e->ev_m0, e->ev_m1));
  e->timing[1] = e->timing[2] = 0.f; e->timing[3] = e->timing[0];
  for (int r = 0; r < R; r++)
    if (e->h_status[r] == ST_BAD_CHAR) {
      char buf[96];
      snprintf(buf, sizeof buf, "sequence %d holds a character other than A C G U T", r);
      e->err = buf;
      return DRNA_ERR_SEQUENCE;
    }
  HIP_TRY(hipMemcpy(E2, e->d_Emfe, (size_t)R * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (E12) HIP_TRY(hipMemcpy(E12, e->d_Epf, (size_t)2 * R * sizeof(int32_t), hipMemcpyDeviceToHost));
  return DRNA_OK;
}

template <int NT, int K>
static void launch_kbest(const KbArgs& a, int R, hipStream_t s) {
  hipLaunchKernelGGL((kbest_kernel<NT, K>), dim3(R), dim3(NT), 0, s, a);
}

extern "C" int drna_subopt_structs_batch(drna_engine* e, int R, int L, const char* seqs, int K, int32_t* E, char* ss) {
  if (!e) return DRNA_ERR_ARG;
  if (R < 1 || L < 1 || L > e->max_L || K < 1 || K > 8 || !seqs || !E || !ss) {
    e->err = "drna_subopt_structs_batch: bad argument (L within the engine's limit, 1 <= K <= 8; seqs, E and ss required)";
    return DRNA_ERR_ARG;
  }
  HIP_TRY(hipSetDevice(e->device));
  const int KT = K <= 4 ? 4 : 8;                      // kernel instantiations
  const int ldmax = e->max_L + 2;
  const size_t stride_max = (size_t)3 * 8 * ldmax * ldmax;
  if (!e->d_ws_kb) {
    e->kb_chunk = e->max_R < 16 ? e->max_R : 16;
    HIP_TRY(hipMalloc((void**)&e->d_ws_kb, stride_max * sizeof(int32_t) * e->kb_chunk));
    HIP_TRY(hipMalloc((void**)&e->d_kbE, (size_t)8 * e->kb_chunk * sizeof(int32_t)));
    HIP_TRY(hipMalloc((void**)&e->d_kbss, (size_t)8 * e->kb_chunk * e->max_L));
    e->ws_bytes += stride_max * sizeof(int32_t) * e->kb_chunk;
  }
  const int ld = L + 2;
  std::vector<int32_t> hE((size_t)KT * e->kb_chunk);
  std::vector<char> hs((size_t)KT * e->kb_chunk * L);
  float ms_total = 0.f;
  for (int r0 = 0; r0 < R; r0 += e->kb_chunk) {
    const int rc = R - r0 < e->kb_chunk ? R - r0 : e->kb_chunk;
    HIP_TRY(hipMemcpy(e->d_seqs, seqs + (size_t)r0 * L, (size_t)rc * L, hipMemcpyHostToDevice));
    for (int k = 0; k < rc; k++) e->h_status[k] = ST_OK;
    KbArgs a;
    a.T = e->d_mfeT; a.plan = e->d_plan; a.hp_len = e->d_hp_len; a.seqs = e->d_seqs; a.L = L; a.ld = ld;
    a.ws = e->d_ws_kb; a.ws_stride = (long long)3 * KT * ld * ld;
    a.E = e->d_kbE; a.ss = e->d_kbss; a.status = e->d_status;
    HIP_TRY(hipEventRecord(e->ev_m0, e->s_mfe));
    if (KT == 4) {
      if (e->nt == 256) launch_kbest<256, 4>(a, rc, e->s_mfe);
      else if (e->nt == 512) launch_kbest<512, 4>(a, rc, e->s_mfe);
      else launch_kbest<1024, 4>(a, rc, e->s_mfe);
    } else {
      if (e->nt == 256) launch_kbest<256, 8>(a, rc, e->s_mfe);
      else if (e->nt == 512) launch_kbest<512, 8>(a, rc, e->s_mfe);
      else launch_kbest<1024, 8>(a, rc, e->s_mfe);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(e->ev_m1, e->s_mfe));
    HIP_TRY(hipStreamSynchronize(e->s_mfe));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e->ev_m0, e->ev_m1));
    ms_total += ms;
    for (int r = 0; r < rc; r++) {
      if (e->h_status[r] == ST_BAD_CHAR) {
        char buf[96];
        snprintf(buf, sizeof buf, "sequence %d holds a character other than A C G U T", r0 + r);
        e->err = buf;
        return DRNA_ERR_SEQUENCE;
      }
      if (e->h_status[r] != ST_OK) {
        char buf[96];
        snprintf(buf, sizeof buf, "sequence %d: traceback of a ranked structure failed (internal error)", r0 + r);
        e->err = buf;
        return DRNA_ERR_INTERNAL;
      }
    }
    HIP_TRY(hipMemcpy(hE.data(), e->d_kbE, (size_t)KT * rc * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(hs.data(), e->d_kbss, (size_t)KT * rc * L, hipMemcpyDeviceToHost));
    for (int r = 0; r < rc; r++)
      for (int k = 0; k < K; k++) {
        E[(size_t)(r0 + r) * K + k] = hE[(size_t)r * KT + k];
        memcpy(ss + ((size_t)(r0 + r) * K + k) * L, hs.data() + ((size_t)r * KT + k) * L, (size_t)L);
      }
  }
  e->timing[0] = ms_total; e->timing[1] = e->timing[2] = 0.f; e->timing[3] = ms_total;
  return DRNA_OK;
}

// ---------------------------------------------------------------- two strands (co-fold)

template <int NT>
static void launch_cofold(const CoArgs& a, int R, bool mfe, bool pf, hipStream_t sm, hipStream_t sp) {
  if (pf) hipLaunchKernelGGL(cofold_pf_kernel<NT>, dim3(R), dim3(NT), 0, sp, a);
  if (mfe) hipLaunchKernelGGL(cofold_mfe_kernel<NT>, dim3(R), dim3(NT), 0, sm, a);
}

extern "C" int drna_cofold_batch(drna_engine* e, int R, int L, int cut, const char* seqs, uint32_t flags, double* F4,
                                 int32_t* Emfe, char* mfe_ss, int32_t* Ed) {
  if (!e) return DRNA_ERR_ARG;
  const bool want_pf = flags & DRNA_NEED_PF, want_mfe = flags & DRNA_NEED_MFE, want_ev = flags & DRNA_NEED_EVAL;
  if (R < 1 || R > e->max_R || L < 2 || L > e->max_L || cut < 1 || cut >= L || !seqs || (want_pf && !F4) ||
      (want_mfe && (!Emfe || !mfe_ss)) || (want_ev && !Ed) || (flags & DRNA_NEED_PK)) {
    e->err = "drna_cofold_batch: bad argument (1 <= cut < L <= max_L, R <= max_R; output pointers for every requested flag; no NEED_PK)";
    return DRNA_ERR_ARG;
  }
  if (want_ev && (e->n_targets < 1 || e->L_targets != L)) {
    e->err = "drna_cofold_batch: DRNA_NEED_EVAL needs drna_set_targets() with the same L ('&' removed)";
    return DRNA_ERR_ARG;
  }
  if (R > e->ws_slots) { e->err = "drna_cofold_batch: batch larger than the workspace (raise DRNA_WS_GB or split the batch)"; return DRNA_ERR_ARG; }
  HIP_TRY(hipSetDevice(e->device));
  if (!e->d_F4) HIP_TRY(hipMalloc((void**)&e->d_F4, (size_t)4 * e->max_R * sizeof(double)));
  HIP_TRY(hipMemcpy(e->d_seqs, seqs, (size_t)R * L, hipMemcpyHostToDevice));
  const int ld = L + 2;
  for (int k = 0; k < 2 * e->max_R; k++) e->h_status[k] = ST_OK;
  CoArgs a;
  a.T = e->d_mfeT; a.F = e->d_pfT; a.plan = e->d_plan; a.hp_len = e->d_hp_len; a.hp_w = e->d_hp_w;
  a.scale = e->d_scale; a.eMLb = e->d_eMLb; a.seqs = e->d_seqs; a.L = L; a.cut = cut; a.ld = ld;
  a.DuplexInit = e->H.DuplexInit;
  a.eDuplexInit = std::exp(-(double)e->H.DuplexInit * 10.0 / e->H.pf.kT);
  a.wsm = e->d_ws_mfe; a.wsm_stride = (long long)mfe_ws_stride(ld);
  a.wsp = e->d_ws_pf; a.wsp_stride = (long long)pf_ws_stride(ld);
  a.Emfe = e->d_Emfe; a.ss = e->d_ss; a.F4 = e->d_F4; a.status = e->d_status; a.status_pf = e->d_status + e->max_R;
  HIP_TRY(hipEventRecord(e->ev_p0, e->s_pf));
  HIP_TRY(hipEventRecord(e->ev_m0, e->s_mfe));
  if (e->nt == 256) launch_cofold<256>(a, R, want_mfe, want_pf, e->s_mfe, e->s_pf);
  else if (e->nt == 512) launch_cofold<512>(a, R, want_mfe, want_pf, e->s_mfe, e->s_pf);
  else launch_cofold<1024>(a, R, want_mfe, want_pf, e->s_mfe, e->s_pf);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(e->ev_p1, e->s_pf));
  HIP_TRY(hipEventRecord(e->ev_m1, e->s_mfe));
  if (want_ev) {
    EvalArgs v;
    v.T = e->d_mfeT; v.hp_len = e->d_hp_len; v.bulge_len = e->d_bulge_len; v.int_len = e->d_int_len;
    v.seqs = e->d_seqs; v.pt = e->d_pt; v.L = L; v.n_targets = e->n_targets; v.Ed = e->d_Ed;
    v.cut = cut; v.DuplexInit = e->H.DuplexInit;
    hipLaunchKernelGGL(eval_kernel, dim3(R * e->n_targets), dim3(WAVE), 0, e->s_eval, v);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipStreamSynchronize(e->s_pf));
  HIP_TRY(hipStreamSynchronize(e->s_mfe));
  HIP_TRY(hipStreamSynchronize(e->s_eval));
  HIP_TRY(hipEventElapsedTime(&e->timing[0], e->ev_m0, e->ev_m1));
  HIP_TRY(hipEventElapsedTime(&e->timing[1], e->ev_p0, e->ev_p1));
  e->timing[2] = 0.f; e->timing[3] = e->timing[0] > e->timing[1] ? e->timing[0] : e->timing[1];
  for (int r = 0; r < R; r++) {
    const int sm = want_mfe ? e->h_status[r] : ST_OK, sp = want_pf ? e->h_status[e->max_R + r] : ST_OK;
    const int st = sm != ST_OK ? sm : sp;
    if (st == ST_OK) continue;
    char buf[160];
    snprintf(buf, sizeof buf, st == ST_BAD_CHAR ? "sequence %d holds a character other than A C G U T"
                              : st == ST_PF_RANGE ? "sequence %d: partition function left the fp64 range"
                              : st == ST_SYNC ? "sequence %d: the two workgroups of the fold lost each other (wait expired)"
                                                  : "sequence %d: traceback could not reproduce a table value", r);
    e->err = buf;
    return st == ST_BAD_CHAR ? DRNA_ERR_SEQUENCE : st == ST_PF_RANGE ? DRNA_ERR_PF_RANGE : DRNA_ERR_INTERNAL;
  }
  if (want_pf) HIP_TRY(hipMemcpy(F4, e->d_F4, (size_t)4 * R * sizeof(double), hipMemcpyDeviceToHost));
  if (want_mfe) {
    HIP_TRY(hipMemcpy(Emfe, e->d_Emfe, (size_t)R * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(mfe_ss, e->d_ss, (size_t)R * L, hipMemcpyDeviceToHost));
  }
  if (want_ev) HIP_TRY(hipMemcpy(Ed, e->d_Ed, (size_t)R * e->n_targets * sizeof(int32_t), hipMemcpyDeviceToHost));
  return DRNA_OK;
}

// ---------------------------------------------------------------- host-side batched MC helpers (no device work)

extern "C" int drna_simscore_batch(int R, int L, const char* ref, const char* queries, double* mcc, double* recall,
                                   double* precision) {
  using namespace drna_host;
  if (R < 0 || L < 1 || L > 2048 || !ref || (R > 0 && (!queries || !mcc || !recall || !precision))) return DRNA_ERR_ARG;
  std::vector<int> pr(L), pq(L);
  if (!pair_table(ref, L, pr.data())) return DRNA_ERR_STRUCTURE;
  for (int r = 0; r < R; r++) {
    if (!pair_table(queries + (size_t)r * L, L, pq.data())) return DRNA_ERR_STRUCTURE;
    const SimMetrics m = sim_metrics(pr.data(), pq.data(), L);
    mcc[r] = m.mcc; recall[r] = m.recall; precision[r] = m.precision;
  }
  return DRNA_OK;
}

extern "C" int drna_rng_seed(int R, const uint64_t* seeds, uint32_t* rng_state) {
  if (R < 0 || (R > 0 && (!seeds || !rng_state))) return DRNA_ERR_ARG;
  for (int r = 0; r < R; r++) drna_host::mt_seed_int(drna_host::Mt{rng_state + (size_t)r * drna_host::RNG_WORDS}, seeds[r]);
  return DRNA_OK;
}

extern "C" int drna_rng_random(int R, uint32_t* rng_state, double* out) {
  if (R < 0 || (R > 0 && (!rng_state || !out))) return DRNA_ERR_ARG;
  for (int r = 0; r < R; r++) out[r] = drna_host::rnd01(drna_host::Mt{rng_state + (size_t)r * drna_host::RNG_WORDS});
  return DRNA_OK;
}

// The design problem as the proposer sees it, built once per call (drna_propose_batch*) or per exchange step (drna_mc_run):
// pair tables of the target and of all design pairs, the mutable positions, the snakes of alternative-structure designs
struct ProposeCtx {
  int L = 0, n_shelves = 1, targeted = 0;
  double tm_max = 0, tm_min = 0;
  const unsigned char* allowed_mask = nullptr;
  const int32_t *snake_of = nullptr, *snake_off = nullptr, *snake_nodes = nullptr, *snake_nstates = nullptr;
  const char* snake_states = nullptr;
  std::vector<int> pt, pd, mutable_pos;
};
static int propose_ctx_init(ProposeCtx& c, int L, const char* target, const int32_t* partner, const unsigned char* allowed_mask,
                            const int32_t* snake_of, const int32_t* snake_off, const int32_t* snake_nodes,
                            const int32_t* snake_nstates, const char* snake_states, int n_shelves, double tm_max, double tm_min,
                            int targeted) {
  using namespace drna_host;
  c.L = L; c.n_shelves = n_shelves; c.targeted = targeted; c.tm_max = tm_max; c.tm_min = tm_min; c.allowed_mask = allowed_mask;
  c.snake_of = snake_of; c.snake_off = snake_off; c.snake_nodes = snake_nodes; c.snake_nstates = snake_nstates; c.snake_states = snake_states;
  c.pt.assign(L, -1); c.pd.assign(L, -1); c.mutable_pos.clear();
  if (!pair_table(target, L, c.pt.data())) return DRNA_ERR_STRUCTURE;
  for (int i = 0; i < L; i++) {
    c.pd[i] = partner ? partner[i] : c.pt[i];
    if (c.pd[i] >= L || (c.pd[i] >= 0 && (partner ? partner[c.pd[i]] : c.pt[c.pd[i]]) != i)) return DRNA_ERR_ARG;
  }
  for (int i = 0; i < L; i++)
    if (__builtin_popcount(allowed_mask[i] & 15u) != 1) c.mutable_pos.push_back(i);
  if (c.mutable_pos.empty()) return DRNA_ERR_ARG;
  return DRNA_OK;
}
// round(numpy.linspace(tm_max, tm_min, n_shelves)[shelf], 2): linspace is start + k * step with the last point set to the stop
// value; round() is the correctly rounded decimal, like printf
static double shelf_probability(const ProposeCtx& c, int shelf) {
  double p = c.tm_max;
  if (c.n_shelves > 1) {
    const double step = (c.tm_min - c.tm_max) / (double)(c.n_shelves - 1);
    p = shelf == c.n_shelves - 1 ? c.tm_min : (double)shelf * step + c.tm_max;
  }
  char buf[32]; snprintf(buf, sizeof buf, "%.2f", p);
  return strtod(buf, nullptr);
}
// targeted moves: the positions a proposal may pick from = ends of false-negative / false-positive pairs of the current MFE
// structure (pair table pq) against the target, widened by +-3 (position 0 never enters); returns their number (0: none).
// mark is scratch of L entries.  The pool depends on the replica's CURRENT structure only, so drna_mc_run keeps it until a
// proposal is accepted
static int targeted_pool(const ProposeCtx& c, const int* pq, char* mark, int* pool) {
  const int L = c.L;
  std::memset(mark, 0, (size_t)L);
  bool any = false;
  for (int i = 0; i < L; i++)
    if (c.pt[i] != pq[i] && (c.pt[i] >= 0 || pq[i] >= 0) && __builtin_popcount(c.allowed_mask[i] & 15u) != 1) {
      any = true;                                       // end of a false-negative or false-positive pair
      for (int k = -3; k <= 3; k++) { const int x = i + k; if (x > 0 && x <= L - 1) mark[x] = 1; }
    }
  int np = 0;
  if (any)
    for (int i = 0; i < L; i++) if (mark[i]) pool[np++] = i;
  return any ? np : -1;                                  // -1: no mispaired position (no draw is made then)
}
// one proposal of one replica: s = its sequence, pool / np = targeted_pool of its current MFE structure (np = -1 without targeted
// moves), p_shelf = shelf_probability of its temperature shelf
static int propose_one(const ProposeCtx& c, const char* s, const int* pool, int np, double p_shelf, drna_host::Mt st, char* o) {
  using namespace drna_host;
  static const char LET[4] = {'A', 'C', 'G', 'U'};
  static const unsigned CANPAIR[4] = {8u, 4u, 2u | 8u, 1u | 4u};   // A-U, C-G, G-C/U, U-A/G
  const int L = c.L;
  const unsigned char* allowed_mask = c.allowed_mask;
  auto letter_index = [](char ch) { return ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : 3; };
  std::memcpy(o, s, (size_t)L);
  int pos = -1;
  if (np >= 0 && rnd_choices2(st, p_shelf) == 0 && np > 0)          // choices([expanded, mutable], weights=[p, 1 - p])
    pos = pool[rnd_below(st, np)];
  if (pos < 0) pos = c.mutable_pos[rnd_below(st, (int)c.mutable_pos.size())];
  const unsigned am = allowed_mask[pos] & 15u;
  const int cur = letter_index(s[pos]);
  const int j = c.pd[pos];
  if (c.snake_of && c.snake_of[pos] >= 0) {
    // alternative structures: the whole connected component moves to another of its Watson-Crick colourings
    // (reference utils/sequence_utils.py:1081-1095)
    const int k = c.snake_of[pos], n0 = c.snake_off[k], len = c.snake_off[k + 1] - n0, ns = c.snake_nstates[k];
    const char* states = c.snake_states + (size_t)4 * n0;
    int x = 0;
    while (x < len && c.snake_nodes[n0 + x] != pos) x++;
    if (x == len) return DRNA_ERR_ARG;
    int curst = -1;
    for (int q = 0; q < ns; q++) if (states[(size_t)q * len + x] == s[pos]) { curst = q; break; }
    const int nopt = ns - (curst >= 0 ? 1 : 0);
    if (nopt > 0) {
      int pick = rnd_below(st, nopt);
      if (curst >= 0 && pick >= curst) pick++;
      for (int y = 0; y < len; y++) o[c.snake_nodes[n0 + y]] = states[(size_t)pick * len + y];
    }
  } else if (j < 0) {
    unsigned opts = __builtin_popcount(am) > 1 ? (am & ~(1u << cur)) : 0u;
    if (opts) {
      int k = rnd_below(st, __builtin_popcount(opts));
      for (int b = 0; b < 4; b++) if (opts & (1u << b)) { if (!k--) { o[pos] = LET[b]; break; } }
    }
  } else {
    unsigned o1 = __builtin_popcount(am) != 1 ? (am & ~(1u << cur)) : am;
    if (!o1) o1 = am;
    int k = rnd_below(st, __builtin_popcount(o1));
    int n1 = 0;
    for (int b = 0; b < 4; b++) if (o1 & (1u << b)) { if (!k--) { n1 = b; break; } }
    const unsigned o2 = (allowed_mask[j] & 15u) & CANPAIR[n1];
    if (o2) {
      int k2 = rnd_below(st, __builtin_popcount(o2));
      for (int b = 0; b < 4; b++) if (o2 & (1u << b)) { if (!k2--) { o[pos] = LET[n1]; o[j] = LET[b]; break; } }
    }
  }
  return DRNA_OK;
}

// one proposal per replica; partner = partner of every design pair (target + ordinary alternative pairs), snakes optional
static int propose_impl(int R, int L, const char* target, const int32_t* partner, const unsigned char* allowed_mask,
                        const int32_t* snake_of, const int32_t* snake_off, const int32_t* snake_nodes,
                        const int32_t* snake_nstates, const char* snake_states, const char* seqs, const char* mfe_ss,
                        const int32_t* shelf_index, int n_shelves, double tm_max, double tm_min, int targeted,
                        uint32_t* rng_state, char* out_seqs) {
  using namespace drna_host;
  ProposeCtx c;
  int rc = propose_ctx_init(c, L, target, partner, allowed_mask, snake_of, snake_off, snake_nodes, snake_nstates, snake_states,
                            n_shelves, tm_max, tm_min, targeted);
  if (rc != DRNA_OK) return rc;
  std::vector<int> pq(L), pool(L);
  std::vector<char> mark(L);
  for (int r = 0; r < R; r++) {
    int np = -1;
    if (targeted) {
      if (!pair_table(mfe_ss + (size_t)r * L, L, pq.data())) return DRNA_ERR_STRUCTURE;
      np = targeted_pool(c, pq.data(), mark.data(), pool.data());
    }
    rc = propose_one(c, seqs + (size_t)r * L, pool.data(), np, targeted ? shelf_probability(c, shelf_index[r]) : 0.0,
                     Mt{rng_state + (size_t)r * RNG_WORDS}, out_seqs + (size_t)r * L);
    if (rc != DRNA_OK) return rc;
  }
  return DRNA_OK;
}

extern "C" int drna_propose_batch(int R, int L, const char* target, const unsigned char* allowed_mask, const char* seqs,
                                  const char* mfe_ss, const int32_t* shelf_index, int n_shelves, double tm_max, double tm_min,
                                  int targeted, uint32_t* rng_state, char* out_seqs) {
  if (R < 0 || L < 1 || L > 2048 || !target || !allowed_mask || (R > 0 && (!seqs || !mfe_ss || !shelf_index || !rng_state || !out_seqs)))
    return DRNA_ERR_ARG;
  return propose_impl(R, L, target, nullptr, allowed_mask, nullptr, nullptr, nullptr, nullptr, nullptr, seqs, mfe_ss,
                      shelf_index, n_shelves, tm_max, tm_min, targeted, rng_state, out_seqs);
}

extern "C" int drna_propose_batch_alt(int R, int L, const char* target, const int32_t* partner,
                                      const unsigned char* allowed_mask, const int32_t* snake_of, int n_snakes,
                                      const int32_t* snake_off, const int32_t* snake_nodes, const int32_t* snake_nstates,
                                      const char* snake_states, const char* seqs, const char* mfe_ss,
                                      const int32_t* shelf_index, int n_shelves, double tm_max, double tm_min, int targeted,
                                      uint32_t* rng_state, char* out_seqs) {
  if (R < 0 || L < 1 || L > 2048 || !target || !partner || !allowed_mask || n_snakes < 0 ||
      (n_snakes > 0 && (!snake_of || !snake_off || !snake_nodes || !snake_nstates || !snake_states)) ||
      (R > 0 && (!seqs || !mfe_ss || !shelf_index || !rng_state || !out_seqs)))
    return DRNA_ERR_ARG;
  for (int k = 0; k < n_snakes; k++) {
    if (snake_off[k] < 0 || snake_off[k + 1] <= snake_off[k] || snake_nstates[k] < 1 || snake_nstates[k] > 4) return DRNA_ERR_ARG;
    for (int x = snake_off[k]; x < snake_off[k + 1]; x++)
      if (snake_nodes[x] < 0 || snake_nodes[x] >= L || snake_of[snake_nodes[x]] != k) return DRNA_ERR_ARG;
  }
  if (n_snakes > 0)
    for (int i = 0; i < L; i++)
      if (snake_of[i] >= n_snakes) return DRNA_ERR_ARG;
  return propose_impl(R, L, target, partner, allowed_mask, n_snakes ? snake_of : nullptr, snake_off, snake_nodes,
                      snake_nstates, snake_states, seqs, mfe_ss, shelf_index, n_shelves, tm_max, tm_min, targeted, rng_state,
                      out_seqs);
}

extern "C" int drna_metropolis_batch(int R, const double* score_o, const double* score_m, const double* temps, double Lconst,
                                     uint32_t* rng_state, unsigned char* accept, unsigned char* better) {
  if (R < 0 || (R > 0 && (!score_o || !score_m || !temps || !rng_state || !accept || !better))) return DRNA_ERR_ARG;
  for (int r = 0; r < R; r++) {
    if (score_m[r] <= score_o[r]) { accept[r] = 1; better[r] = 1; continue; }
    better[r] = 0;
    const double p = std::exp((-Lconst / temps[r]) * (score_m[r] - score_o[r]));
    accept[r] = p > drna_host::rnd01(drna_host::Mt{rng_state + (size_t)r * drna_host::RNG_WORDS}) ? 1 : 0;   // one draw, only when the mutant is worse
  }
  return DRNA_OK;
}

// ---------------------------------------------------------------- the whole Monte-Carlo inner loop of one exchange step

extern "C" int drna_mc_run(drna_engine* e, int R, int L, int n_iter, const char* target, const int32_t* partner,
                           const unsigned char* allowed_mask, const int32_t* snake_of, int n_snakes, const int32_t* snake_off,
                           const int32_t* snake_nodes, const int32_t* snake_nstates, const char* snake_states,
                           const int32_t* shelf_index, int n_shelves, double tm_max, double tm_min, int targeted,
                           const double* temps, double Lconst, int n_terms, const int32_t* term_id, const double* term_w,
                           uint32_t flags, uint32_t* rng_state, char* seqs, char* mfe_ss, double* score, double* mcc1,
                           double* Epf, double* Ed, int64_t* counters, char* best_seq, char* best_ss, double* best) {
  using namespace drna_host;
  if (!e) return DRNA_ERR_ARG;
  if (R < 1 || R > e->max_R || L < 1 || L > e->max_L || n_iter < 0 || !target || !allowed_mask || !shelf_index || !temps ||
      n_terms < 1 || !term_id || !term_w || !rng_state || !seqs || !mfe_ss || !score || !mcc1 || !Epf || !Ed || !counters ||
      !best_seq || !best_ss || !best || e->n_targets < 1 || e->L_targets != L) {
    e->err = "drna_mc_run: bad argument (targets installed with drna_set_targets for this L; every state array given)";
    return DRNA_ERR_ARG;
  }
  const int nt = e->n_targets;
  std::vector<char> prop((size_t)R * L), pss((size_t)R * L);
  std::vector<double> pEpf(R), pscore(R), pmcc(R), pEdef, p_shelf(R, 0.0);
  bool want_edef = false;                  // term 6: ensemble defect against targets[0] (utils/energy_scores.py:362-374,397-398)
  for (int k = 0; k < n_terms; k++) {
    want_edef |= term_id[k] == 6;
    if (term_id[k] < 0 || term_id[k] > 6) { e->err = "drna_mc_run: unknown scoring term"; return DRNA_ERR_ARG; }
  }
  if (want_edef) pEdef.resize(R);
  std::vector<int32_t> pEmfe(R), pEd((size_t)R * nt);
  std::vector<unsigned char> acc(R), better(R);
  ProposeCtx ctx;
  {
    const int rc = propose_ctx_init(ctx, L, target, partner, allowed_mask, n_snakes > 0 ? snake_of : nullptr, snake_off, snake_nodes,
                                    snake_nstates, snake_states, n_shelves, tm_max, tm_min, targeted);
    if (rc != DRNA_OK) { e->err = rc == DRNA_ERR_STRUCTURE ? "drna_mc_run: unbalanced target structure" : "drna_mc_run: proposal failed"; return rc; }
  }
  const int* pr = ctx.pt.data();
  // Per replica, of its CURRENT structure: the pair table, its SimScore against the target and the targeted-move pool (what the
  // proposal compares with the target).  Parsed once here and replaced when a proposal is accepted; a proposal whose MFE
  // structure equals the current one (most single mutations of a converged replica) reuses all three
  std::vector<int> cur_pq((size_t)R * L), prop_pq((size_t)R * L), cur_pool((size_t)R * L), cur_np(R, -1);
  std::vector<SimMetrics> cur_m(R);
  // host work per replica may be dealt to worker threads (McPool, host_driver.hpp; option "mc_threads", default 1: at ~0.4 us per
  // replica the hand-off to spinning workers costs what it saves, measured on the GPU box); replicas own their random streams and state
  const int T = std::max(1, std::min(e->mc_threads > 0 ? e->mc_threads : 1, (R + 3) / 4));
  e->mc_threads_used = T;
  McPool workers(T);
  std::vector<std::vector<char>> mark(T, std::vector<char>(L));
  std::atomic<int> fail{DRNA_OK};
  auto range = [&](int w, int& r0, int& r1) { r0 = (int)((long long)R * w / T); r1 = (int)((long long)R * (w + 1) / T); };
  auto refresh_pool = [&](int w, int r) {
    if (targeted) cur_np[r] = targeted_pool(ctx, cur_pq.data() + (size_t)r * L, mark[w].data(), cur_pool.data() + (size_t)r * L);
  };
  auto propose_range = [&](int r0, int r1) {
    for (int r = r0; r < r1; r++) {
      const int rc = propose_one(ctx, seqs + (size_t)r * L, cur_pool.data() + (size_t)r * L, cur_np[r], p_shelf[r],
                                 Mt{rng_state + (size_t)r * RNG_WORDS}, prop.data() + (size_t)r * L);
      if (rc != DRNA_OK) fail.store(rc);
    }
  };
  workers.run([&](int w) {
    int r0, r1; range(w, r0, r1);
    for (int r = r0; r < r1; r++) {
      if (targeted) p_shelf[r] = shelf_probability(ctx, shelf_index[r]);
      if (!pair_table(mfe_ss + (size_t)r * L, L, cur_pq.data() + (size_t)r * L)) { fail.store(DRNA_ERR_STRUCTURE); continue; }
      cur_m[r] = sim_metrics(pr, cur_pq.data() + (size_t)r * L, L);
      refresh_pool(w, r);
    }
    if (n_iter > 0 && fail.load() == DRNA_OK) propose_range(r0, r1);
  });
  if (fail.load() != DRNA_OK) { e->err = "drna_mc_run: proposal failed (unbalanced structure in the state, or a bad design problem)"; return fail.load(); }
  static const bool mc_profile = getenv("DRNA_MC_PROFILE") != nullptr;     // diagnostics: where an iteration's host time goes (stderr)
  double prof[3] = {0, 0, 0};
  auto now_us = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; };
  for (int it = 0; it < n_iter; it++) {
    const double tp0 = mc_profile ? now_us() : 0.0;
    int rc = drna_score_batch(e, R, L, prop.data(), flags | DRNA_NEED_PF | DRNA_NEED_MFE | DRNA_NEED_EVAL, pEpf.data(), pEmfe.data(),
                              pss.data(), pEd.data());
    if (rc != DRNA_OK) return rc;
    const double tp1 = mc_profile ? now_us() : 0.0;
    if (want_edef) {                        // inside + outside recursion of every proposal (fold_outside.hpp)
      rc = drna_ensemble_defect_batch(e, R, L, prop.data(), pEdef.data(), nullptr);
      if (rc != DRNA_OK) return rc;
    }
    const bool more = it + 1 < n_iter;
    workers.run([&](int w) {
      int r0, r1; range(w, r0, r1);
      for (int r = r0; r < r1; r++) {
        // SimScore of the proposal's structure against the target (utils/sim_score.py:62-147)
        int* ppq = prop_pq.data() + (size_t)r * L;
        const bool same_ss = std::memcmp(pss.data() + (size_t)r * L, mfe_ss + (size_t)r * L, (size_t)L) == 0;
        SimMetrics m = cur_m[r];
        if (!same_ss) {
          if (!pair_table(pss.data() + (size_t)r * L, L, ppq)) { fail.store(DRNA_ERR_STRUCTURE); continue; }
          m = sim_metrics(pr, ppq, L);
        }
        const double ed = pEd[(size_t)r * nt] / 100.0;
        // -sf terms (utils/energy_scores.py:376-398): 0 Ed-Epf, 1 1-MCC, 2 sln_Epf, 3 Ed-MFE, 4 1-precision, 5 1-recall, 6 Edef
        double tot = 0.0;
        for (int k = 0; k < n_terms; k++) {
          double v = 0.0;
          switch (term_id[k]) {
            case 0: v = ed - pEpf[r]; break;
            case 1: v = (1 - m.mcc) * 10; break;
            case 2: v = (pEpf[r] + 0.3759 * L + 5.7534) / 10; break;
            case 3: v = ed - pEmfe[r] / 100.0; break;
            case 4: v = (1 - m.precision) * 10; break;
            case 5: v = (1 - m.recall) * 10; break;
            case 6: v = pEdef[r]; break;
          }
          tot += v * term_w[k];
        }
        if (nt > 1) {                                           // alternative structures (:98-102)
          double sum = 0.0;
          for (int t = 1; t < nt; t++) sum += pEd[(size_t)r * nt + t] / 100.0;
          tot += sum / (nt - 1) - pEpf[r];
        }
        pscore[r] = tot; pmcc[r] = 1 - m.mcc;
        // Metropolis (utils/replica_exchange_monte_carlo.py:26-57): one draw from the replica's stream, only when the mutant is worse
        (void)drna_metropolis_batch(1, score + r, pscore.data() + r, temps + r, Lconst, rng_state + (size_t)r * RNG_WORDS, acc.data() + r,
                                    better.data() + r);
        if (acc[r]) {
          std::memcpy(seqs + (size_t)r * L, prop.data() + (size_t)r * L, (size_t)L);
          if (!same_ss) {
            std::memcpy(mfe_ss + (size_t)r * L, pss.data() + (size_t)r * L, (size_t)L);
            std::memcpy(cur_pq.data() + (size_t)r * L, ppq, (size_t)L * sizeof(int));
            cur_m[r] = m;
            refresh_pool(w, r);
          }
          score[r] = pscore[r]; mcc1[r] = pmcc[r]; Epf[r] = pEpf[r]; Ed[r] = ed;
        }
      }
      if (more && fail.load() == DRNA_OK) propose_range(r0, r1);          // the next iteration's proposals (same streams, after the Metropolis draw)
    });
    if (fail.load() != DRNA_OK) { e->err = "drna_mc_run: unbalanced MFE structure from the engine, or a failed proposal"; return fail.load(); }
    const double tp2 = mc_profile ? now_us() : 0.0;
    // counters and the best state, replica by replica in replica order (first strictly better wins)
    for (int r = 0; r < R; r++) {
      if (acc[r]) {
        counters[0]++;
        if (better[r]) counters[1]++;
        if (mcc1[r] < best[0] || (mcc1[r] == best[0] && score[r] < best[1])) {
          best[0] = mcc1[r]; best[1] = score[r]; best[2] = Epf[r]; best[3] = Ed[r];
          std::memcpy(best_seq, seqs + (size_t)r * L, (size_t)L);
          std::memcpy(best_ss, mfe_ss + (size_t)r * L, (size_t)L);
        }
      } else counters[2]++;
    }
    if (mc_profile) { prof[0] += tp1 - tp0 - e->timing[3] * 1e3; prof[1] += tp2 - tp1; prof[2] += now_us() - tp2; }
  }
  if (mc_profile && n_iter > 0)
    fprintf(stderr, "drna_mc_run: per iteration, host us: score call beyond device time %.1f, per-replica work (%d threads) %.1f, bookkeeping %.1f\n",
            prof[0] / n_iter, T, prof[1] / n_iter, prof[2] / n_iter);
  return DRNA_OK;
}

#if defined(DRNA_STAMPS) || defined(MSTRIP_STAMPS) || defined(DRNA_TL)
// diagnostic build only: copy `count` int32 of the MFE workspace starting at int32 offset `off`
extern "C" int drna_debug_read_mfe_ws(drna_engine* e, long long off, int count, int32_t* out) {
  if (!e || !out) return DRNA_ERR_ARG;
  HIP_TRY(hipMemcpy(out, e->d_ws_mfe + off, (size_t)count * 4, hipMemcpyDeviceToHost));
  return DRNA_OK;
}
extern "C" int drna_debug_read_pf_ws(drna_engine* e, long long off, int count, double* out) {
  if (!e || !out) return DRNA_ERR_ARG;
  HIP_TRY(hipMemcpy(out, e->d_ws_pf + off, (size_t)count * 8, hipMemcpyDeviceToHost));
  return DRNA_OK;
}
#endif
