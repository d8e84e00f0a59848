// emu_kernels.cpp -- TEST-ONLY: compiles the unmodified kernel headers against hip_emu.h and exposes
// them through a tiny C interface for tests/test_kernels_emulated.py (CPU, no GPU needed).
#include "hip_emu.h"

thread_local emu_dim3 threadIdx;
thread_local emu_dim3 blockIdx;
thread_local emu_group* emu_g = nullptr;

#include "../../desirna_amd/csrc/eval_structure.hpp"
#include "../../desirna_amd/csrc/fold_mfe.hpp"
#include "../../desirna_amd/csrc/fold_mfe_lds.hpp"
#include "../../desirna_amd/csrc/fold_mfe_dual.hpp"
#include "../../desirna_amd/csrc/fold_mfe_strip.hpp"
#include "../../desirna_amd/csrc/fold_pf.hpp"
#include "../../desirna_amd/csrc/fold_pf_lds.hpp"
#include "../../desirna_amd/csrc/fold_pf_strip.hpp"
#include "../../desirna_amd/csrc/fold_outside.hpp"
#include "../../desirna_amd/csrc/fold_cofold.hpp"
#include "../../desirna_amd/csrc/fold_subopt.hpp"

using namespace drna;

namespace {
struct Ctx {
  HostTables H;
  bool ok = false;
};
Ctx* make_ctx(const int32_t* blob, int n, int max_L) {
  Ctx* c = new Ctx();
  c->ok = build_tables(blob, n, c->H).empty();
  if (c->ok) size_tables(c->H, max_L + 2);
  return c;
}
}  // namespace

// two-workgroup MFE kernel: the main and the helper role of every sequence run side by side (2 x nt OS threads), exchanging
// rows through ordinary memory with the same flags / epochs as on the GPU; `calls` repeats the batch to exercise the epochs
template <int NT>
static void run_mfe_dual(const MfeArgs& a, int R, int calls) {
  const size_t rows = (size_t)2 * (MFE_FAST_NMAX + 2) * XP;
  std::vector<int> flags((size_t)R * 64, 0);
  std::vector<int32_t> xs((size_t)R * 256, 0), xa(rows * R, 0), xb(rows * R, 0);
  auto* smA = new MfeFastSmem<NT>();
  auto* smB = new MfeHelperSmem<NT>();
  for (int call = 1; call <= calls; call++)
    for (int r = 0; r < R; r++) {
      DualLink lk;
      lk.flagA = flags.data() + r * 64; lk.flagB = flags.data() + r * 64 + 32;
      lk.xs = xs.data() + (size_t)r * 256; lk.xa = xa.data() + rows * r; lk.xb = xb.data() + rows * r; lk.epoch = call;
      std::vector<std::function<void()>> fns;
      fns.push_back([&, r, lk]() { mfe_lds_body<NT, true>(*smA, a, r, lk); });
      fns.push_back([&, r, lk]() { mfe_helper<NT>(*smB, a, r, lk); });
      emu_launch_many(2 * r, NT, fns);
    }
  delete smA; delete smB;
}

// strip kernel of the partition function: the S strips of a sequence run side by side (S x nt OS threads) and hand their
// records over through ordinary memory with the same flags / epochs as on the GPU; `calls` repeats the batch (epochs)
template <int NT>
static void run_pf_strip(const PfArgs& a, int R, int S, int calls) {
  std::vector<int> flags((size_t)R * STRIP_MAXS * 32, 0);
  std::vector<PfStripSmem<NT>*> sms;
  for (int s = 0; s < S; s++) sms.push_back(new PfStripSmem<NT>());
  for (int call = 1; call <= calls; call++)
    for (int r = 0; r < R; r++) {
      StripLink lk;
      lk.flags = flags.data(); lk.base = call << 12; lk.nseq = R; lk.S = S;
      std::vector<std::function<void()>> fns;
      for (int s = 0; s < S; s++) fns.push_back([&, r, s, lk]() { pf_strip_body<NT>(*sms[s], a, lk, r, s); });
      emu_launch_many(r * S, NT, fns);
    }
  for (auto* p : sms) delete p;
}

// strip kernel of the MFE fold: per pseudoknot round the S strips of a sequence side by side, then the traceback "launch"
template <int NT>
static void run_mfe_strip(const MfeArgs& a, int R, int S, int calls, int fark) {
  std::vector<int> flags((size_t)R * STRIP_MAXS * 32, 0);
  StripRec xr;
  xr.stride = (long long)S * a.ld * MSTRIP_REC;
  std::vector<int32_t> rec((size_t)xr.stride * R, 0);
  xr.rec = rec.data();
  std::vector<MfeStripSmem<NT>*> sms;
  for (int s = 0; s < S; s++) sms.push_back(new MfeStripSmem<NT>());
  auto* smt = new MfeTraceSmem();
  int epoch = 0;
  for (int call = 1; call <= calls; call++)
    for (int round = 0; round <= a.pk_rounds; round++) {
      epoch++;
      for (int r = 0; r < R; r++) {
        StripLink lk;
        lk.flags = flags.data(); lk.base = epoch << 12; lk.nseq = R; lk.S = S; lk.fark = fark;
        std::vector<std::function<void()>> fns;
        for (int s = 0; s < S; s++)
          fns.push_back([&, r, s, lk, round]() {
            if (lk.fark) mfe_strip_body<NT, true>(*sms[s], a, lk, xr, r, s, round);
            else mfe_strip_body<NT, false>(*sms[s], a, lk, xr, r, s, round);
          });
        emu_launch_many(r * S, NT, fns);
      }
      for (int r = 0; r < R; r++) emu_launch(r, TRACE_WAVES * WAVE, [&, r, round]() { mfe_strip_trace_body(*smt, a, nullptr, r, round, 0); });
    }
  for (auto* p : sms) delete p;
  delete smt;
}

extern "C" {

int emu_mfe_strip(const int32_t* blob, int n_int32, int R, int L, const char* seqs, int pk_rounds, int nt, int S, int calls,
                  int32_t* Emfe, char* ss, int32_t* status, int fark) {
  Ctx* c = make_ctx(blob, n_int32, L);
  if (!c->ok) { delete c; return -1; }
  const int ld = L + 2;
  std::vector<int32_t> ws((size_t)5 * ld * ld * R, 0);
  MfeArgs a;
  a.T = &c->H.mfe; a.plan = &c->H.plan; a.hp_len = c->H.hp_len.data();
  a.seqs = seqs; a.L = L; a.ld = ld; a.pk_rounds = pk_rounds;
  a.ws = ws.data(); a.ws_stride = (long long)5 * ld * ld;
  a.Emfe = Emfe; a.ss = ss; a.status = status;
  if (nt == 256) run_mfe_strip<256>(a, R, S, calls, fark);
  else run_mfe_strip<1024>(a, R, S, calls, fark);
  delete c;
  return 0;
}

int emu_pf_strip(const int32_t* blob, int n_int32, int R, int L, const char* seqs, int nt, int S, int calls, double* Epf,
                 int32_t* status) {
  Ctx* c = make_ctx(blob, n_int32, L);
  if (!c->ok) { delete c; return -1; }
  const int ld = L + 2;
  const size_t stride = (size_t)7 * ld * ld + ((size_t)ld * ld + 7) / 8;
  std::vector<double> ws(stride * R, 0.0);
  PfArgs a;
  a.T = &c->H.pf; a.plan = &c->H.plan; a.hp_w = c->H.hp_w.data(); a.scale = c->H.scale.data();
  a.eMLb = c->H.eMLb.data(); a.seqs = seqs; a.L = L; a.ld = ld;
  a.ws = ws.data(); a.ws_stride = (long long)stride;
  a.Epf = Epf; a.status = status;
  if (nt == 256) run_pf_strip<256>(a, R, S, calls);
  else run_pf_strip<1024>(a, R, S, calls);
  delete c;
  return 0;
}

// returns 0 on success.  tables (optional, may be null): Wc / FML dumps of the LAST sequence, ld*ld int32 each
int emu_mfe(const int32_t* blob, int n_int32, int R, int L, const char* seqs, int pk_rounds, int nt, int32_t* Emfe,
            char* ss, int32_t* status, int32_t* dumpWc, int32_t* dumpFML) {
  Ctx* c = make_ctx(blob, n_int32, L);
  if (!c->ok) { delete c; return -1; }
  const int ld = L + 2;
  std::vector<int32_t> ws((size_t)5 * ld * ld, 0);
  for (int r = 0; r < R; r++) {
    MfeArgs a;
    a.T = &c->H.mfe; a.plan = &c->H.plan; a.hp_len = c->H.hp_len.data();
    a.seqs = seqs; a.L = L; a.ld = ld; a.pk_rounds = pk_rounds;
    a.ws = ws.data() - (size_t)r * 5 * ld * ld; a.ws_stride = (long long)5 * ld * ld;   // same buffer for every r
    a.Emfe = Emfe; a.ss = ss; a.status = status;
    auto fn = [&]() {
      if (nt == 64) mfe_kernel<64>(a);
      else if (nt == 128) mfe_kernel<128>(a);
      else if (nt == 256) mfe_kernel<256>(a);
      else if (nt == -256) mfe_lds_kernel<256>(a);     // LDS-resident path (needs n <= 64 at 4 waves)
      else mfe_lds_kernel<1024>(a);                    // nt == -1024
    };
    emu_launch(r, nt < 0 ? -nt : nt, fn);
  }
  if (dumpWc) std::memcpy(dumpWc, ws.data(), (size_t)ld * ld * 4);
  if (dumpFML) std::memcpy(dumpFML, ws.data() + (size_t)2 * ld * ld, (size_t)ld * ld * 4);
  delete c;
  return 0;
}

int emu_mfe_dual(const int32_t* blob, int n_int32, int R, int L, const char* seqs, int pk_rounds, int nt, int calls, int32_t* Emfe,
                 char* ss, int32_t* status) {
  Ctx* c = make_ctx(blob, n_int32, L);
  if (!c->ok) { delete c; return -1; }
  const int ld = L + 2;
  std::vector<int32_t> ws((size_t)5 * ld * ld * R, 0);
  MfeArgs a;
  a.T = &c->H.mfe; a.plan = &c->H.plan; a.hp_len = c->H.hp_len.data();
  a.seqs = seqs; a.L = L; a.ld = ld; a.pk_rounds = pk_rounds;
  a.ws = ws.data(); a.ws_stride = (long long)5 * ld * ld;
  a.Emfe = Emfe; a.ss = ss; a.status = status;
  if (nt == 256) run_mfe_dual<256>(a, R, calls);
  else run_mfe_dual<1024>(a, R, calls);
  delete c;
  return 0;
}

int emu_pf(const int32_t* blob, int n_int32, int R, int L, const char* seqs, int nt, double* Epf, int32_t* status) {
  Ctx* c = make_ctx(blob, n_int32, L);
  if (!c->ok) { delete c; return -1; }
  const int ld = L + 2;
  const size_t stride = (size_t)7 * ld * ld + ((size_t)ld * ld + 7) / 8;
  std::vector<double> ws(stride, 0.0);
  for (int r = 0; r < R; r++) {
    PfArgs a;
    a.T = &c->H.pf; a.plan = &c->H.plan; a.hp_w = c->H.hp_w.data(); a.scale = c->H.scale.data();
    a.eMLb = c->H.eMLb.data(); a.seqs = seqs; a.L = L; a.ld = ld;
    a.ws = ws.data() - (size_t)r * stride; a.ws_stride = (long long)stride;
    a.Epf = Epf; a.status = status;
    // nt = -257 / -1025: the LDS-resident kernel with a helper workgroup per sequence (far multiloop split points), side by side
    const bool helper = nt == -257 || nt == -1025;
    std::vector<int> hflags(64, 0);
    if (helper) { a.helper = 1; a.hflags = hflags.data() - (size_t)r * 64; a.hbase = 3 << 12; }
    auto fn = [&]() {
      if (nt == 64) pf_kernel<64>(a);
      else if (nt == 128) pf_kernel<128>(a);
      else if (nt == 256) pf_kernel<256>(a);
      else if (nt == -256 || nt == -257) pf_lds_kernel<256>(a, EvalArgs{}, 0);
      else pf_lds_kernel<1024>(a, EvalArgs{}, 0);
    };
    if (helper) {           // two workgroups side by side, each with an LDS image of its own
      auto* s256 = new PfFastSmem<256>[2];
      auto* s1024 = new PfFastSmem<1024>[2];
      std::vector<std::function<void()>> fns;
      for (int b = 0; b < 2; b++)
        fns.push_back([&, b]() { if (nt == -257) pf_lds_body<256>(s256[b], a, EvalArgs{}); else pf_lds_body<1024>(s1024[b], a, EvalArgs{}); });
      emu_launch_many(2 * r, nt == -257 ? 256 : 1024, fns);
      delete[] s256; delete[] s1024;
    } else emu_launch(r, nt < 0 ? -nt : nt, fn);
  }
  delete c;
  return 0;
}

int emu_eval(const int32_t* blob, int n_int32, int R, int L, const char* seqs, int n_targets, const short* pt,
             int32_t* Ed) {
  Ctx* c = make_ctx(blob, n_int32, L);
  if (!c->ok) { delete c; return -1; }
  EvalArgs a;
  a.T = &c->H.mfe; a.hp_len = c->H.hp_len.data(); a.bulge_len = c->H.bulge_len.data(); a.int_len = c->H.int_len.data();
  a.seqs = seqs; a.pt = pt; a.L = L; a.n_targets = n_targets; a.Ed = Ed;
  for (int b = 0; b < R * n_targets; b++) emu_launch(b, 64, [&]() { eval_kernel(a); });
  delete c;
  return 0;
}

// general pf_kernel followed by outside_kernel; pt = pair table of the target (L+2 shorts); bpp optional R*(L+1)*(L+1)
int emu_edef(const int32_t* blob, int n_int32, int R, int L, const char* seqs, const short* pt, int nt, double* edef,
             double* bpp, int32_t* status) {
  Ctx* c = make_ctx(blob, n_int32, L);
  if (!c->ok) { delete c; return -1; }
  const int ld = L + 2;
  const size_t stride = (size_t)7 * ld * ld + ((size_t)ld * ld + 7) / 8;
  const size_t ostride = (size_t)outside_ws_stride(ld);
  std::vector<double> ws(stride, 0.0), wo(ostride, 0.0), Epf(R, 0.0);
  for (int r = 0; r < R; r++) {
    PfArgs a;
    a.T = &c->H.pf; a.plan = &c->H.plan; a.hp_w = c->H.hp_w.data(); a.scale = c->H.scale.data();
    a.eMLb = c->H.eMLb.data(); a.seqs = seqs; a.L = L; a.ld = ld;
    a.ws = ws.data() - (size_t)r * stride; a.ws_stride = (long long)stride;
    a.Epf = Epf.data(); a.status = status;
    a.q5out = wo.data() + (size_t)4 * ld * ld - (size_t)r * ostride; a.q5_stride = (long long)ostride;
    OutArgs o;
    o.T = &c->H.pf; o.plan = &c->H.plan; o.scale = c->H.scale.data(); o.eMLb = c->H.eMLb.data();
    o.seqs = seqs; o.L = L; o.ld = ld; o.ws = a.ws; o.ws_stride = a.ws_stride;
    o.wo = wo.data() - (size_t)r * ostride; o.wo_stride = (long long)ostride;
    o.pt = pt; o.edef = edef; o.bpp = bpp; o.pf_status = status;
    auto f1 = [&]() {
      if (nt == 64) pf_kernel<64>(a);
      else if (nt == 128) pf_kernel<128>(a);
      else pf_kernel<256>(a);
    };
    auto f2 = [&]() {
      if (nt == 64) outside_kernel<64>(o);
      else if (nt == 128) outside_kernel<128>(o);
      else outside_kernel<256>(o);
    };
    emu_launch(r, nt, f1);
    emu_launch(r, nt, f2);
  }
  delete c;
  return 0;
}

// ragged batch: sequences of different lengths (concatenated) through the MFE and PF kernels of one "launch";
// lds != 0 selects the LDS-resident kernels (256 threads: n <= 64).  ld = max_L + 2 for every sequence.
int emu_ragged(const int32_t* blob, int n_int32, int R, int max_L, const int32_t* lens, const int32_t* offs, const char* seqs,
               int lds, int32_t* Emfe, char* ss, double* Epf, int32_t* status) {
  Ctx* c = make_ctx(blob, n_int32, max_L);
  if (!c->ok) { delete c; return -1; }
  const int ld = max_L + 2;
  std::vector<int32_t> wsm((size_t)5 * ld * ld, 0);
  const size_t pstride = (size_t)7 * ld * ld + ((size_t)ld * ld + 7) / 8;
  std::vector<double> wsp(pstride, 0.0);
  for (int r = 0; r < R; r++) {
    MfeArgs a;
    a.T = &c->H.mfe; a.plan = &c->H.plan; a.hp_len = c->H.hp_len.data();
    a.seqs = seqs; a.L = 0; a.ld = ld; a.pk_rounds = 0;
    a.ws = wsm.data() - (size_t)r * 5 * ld * ld; a.ws_stride = (long long)5 * ld * ld;
    a.Emfe = Emfe; a.ss = ss; a.status = status;
    a.rg.len = lens; a.rg.off = offs;
    PfArgs b;
    b.T = &c->H.pf; b.plan = &c->H.plan; b.hp_w = c->H.hp_w.data(); b.scale = c->H.scale.data();
    b.eMLb = c->H.eMLb.data(); b.seqs = seqs; b.L = 0; b.ld = ld;
    b.ws = wsp.data() - (size_t)r * pstride; b.ws_stride = (long long)pstride;
    b.Epf = Epf; b.status = status + R;
    b.rg.len = lens; b.rg.off = offs;
    if (lds) {
      emu_launch(r, 256, [&]() { mfe_lds_kernel<256>(a); });
      emu_launch(r, 256, [&]() { pf_lds_kernel<256>(b, EvalArgs{}, 0); });
    } else {
      emu_launch(r, 128, [&]() { mfe_kernel<128>(a); });
      emu_launch(r, 128, [&]() { pf_kernel<128>(b); });
    }
  }
  delete c;
  return 0;
}

// two strands: co-fold MFE + PF kernels (and the eval kernel with the nick) for R pairs of total length L
int emu_cofold(const int32_t* blob, int n_int32, int R, int L, int cut, const char* seqs, int nt, int32_t* Emfe, char* ss,
               double* F4, int32_t* status, const short* pt, int32_t* Ed) {
  Ctx* c = make_ctx(blob, n_int32, L);
  if (!c->ok) { delete c; return -1; }
  const int ld = L + 2;
  std::vector<int32_t> wsm((size_t)5 * ld * ld, 0);
  const size_t pstride = (size_t)7 * ld * ld + ((size_t)ld * ld + 7) / 8;
  std::vector<double> wsp(pstride, 0.0);
  for (int r = 0; r < R; r++) {
    CoArgs a;
    a.T = &c->H.mfe; a.F = &c->H.pf; a.plan = &c->H.plan; a.hp_len = c->H.hp_len.data(); a.hp_w = c->H.hp_w.data();
    a.scale = c->H.scale.data(); a.eMLb = c->H.eMLb.data(); a.seqs = seqs; a.L = L; a.cut = cut; a.ld = ld;
    a.DuplexInit = c->H.DuplexInit; a.eDuplexInit = std::exp(-(double)c->H.DuplexInit * 10.0 / c->H.pf.kT);
    a.wsm = wsm.data() - (size_t)r * 5 * ld * ld; a.wsm_stride = (long long)5 * ld * ld;
    a.wsp = wsp.data() - (size_t)r * pstride; a.wsp_stride = (long long)pstride;
    a.Emfe = Emfe; a.ss = ss; a.F4 = F4; a.status = status; a.status_pf = status + R;
    if (nt == 64) { emu_launch(r, 64, [&]() { cofold_mfe_kernel<64>(a); }); emu_launch(r, 64, [&]() { cofold_pf_kernel<64>(a); }); }
    else { emu_launch(r, 128, [&]() { cofold_mfe_kernel<128>(a); }); emu_launch(r, 128, [&]() { cofold_pf_kernel<128>(a); }); }
  }
  if (pt && Ed) {
    EvalArgs v;
    v.T = &c->H.mfe; v.hp_len = c->H.hp_len.data(); v.bulge_len = c->H.bulge_len.data(); v.int_len = c->H.int_len.data();
    v.seqs = seqs; v.pt = pt; v.L = L; v.n_targets = 1; v.Ed = Ed; v.cut = cut; v.DuplexInit = c->H.DuplexInit;
    for (int b = 0; b < R; b++) emu_launch(b, 64, [&]() { eval_kernel(v); });
  }
  delete c;
  return 0;
}

// two-best energies (second-best structure) of R sequences
int emu_subopt(const int32_t* blob, int n_int32, int R, int L, const char* seqs, int nt, int32_t* E2, int32_t* E12, int32_t* status) {
  Ctx* c = make_ctx(blob, n_int32, L);
  if (!c->ok) { delete c; return -1; }
  const int ld = L + 2;
  std::vector<int32_t> ws((size_t)6 * ld * ld, 0);
  for (int r = 0; r < R; r++) {
    SubArgs a;
    a.T = &c->H.mfe; a.plan = &c->H.plan; a.hp_len = c->H.hp_len.data(); a.seqs = seqs; a.L = L; a.ld = ld;
    a.ws = ws.data() - (size_t)r * 6 * ld * ld; a.ws_stride = (long long)6 * ld * ld;
    a.E2 = E2; a.E12 = E12; a.status = status;
    if (nt == 64) emu_launch(r, 64, [&]() { subopt_kernel<64>(a); });
    else emu_launch(r, 128, [&]() { subopt_kernel<128>(a); });
  }
  delete c;
  return 0;
}
// K lowest-energy structures (energies + strings) of R sequences
int emu_kbest(const int32_t* blob, int n_int32, int R, int L, const char* seqs, int nt, int K, int32_t* E, char* ss, int32_t* status) {
  Ctx* c = make_ctx(blob, n_int32, L);
  if (!c->ok) { delete c; return -1; }
  const int ld = L + 2;
  const size_t stride = (size_t)3 * K * ld * ld;
  std::vector<int32_t> ws(stride, 0);
  for (int r = 0; r < R; r++) {
    KbArgs a;
    a.T = &c->H.mfe; a.plan = &c->H.plan; a.hp_len = c->H.hp_len.data(); a.seqs = seqs; a.L = L; a.ld = ld;
    a.ws = ws.data() - (size_t)r * stride; a.ws_stride = (long long)stride;
    a.E = E; a.ss = ss; a.status = status;
    if (K == 4) {
      if (nt == 64) emu_launch(r, 64, [&]() { kbest_kernel<64, 4>(a); });
      else emu_launch(r, 128, [&]() { kbest_kernel<128, 4>(a); });
    } else if (K == 8) {
      if (nt == 64) emu_launch(r, 64, [&]() { kbest_kernel<64, 8>(a); });
      else emu_launch(r, 128, [&]() { kbest_kernel<128, 8>(a); });
    } else { delete c; return -2; }
  }
  delete c;
  return 0;
}
}

