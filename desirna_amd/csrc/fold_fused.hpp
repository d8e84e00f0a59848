// fold_fused.hpp -- ONE launch for both folds of a small batch (n <= 200, 4 R <= CUs): the MFE fold by two workgroups
// (fold_mfe_dual.hpp), the partition function by a main and a helper workgroup (fold_pf_lds.hpp, pf_kfar_helper) with
// E(targets) evaluated by the helper.  Replaces the reference's three calls per sequence -- fc.pf(), fc.mfe(),
// fc.eval_structure() (utils/energy_scores.py:150,151,75) -- by one kernel of 4 R workgroups, one per CU.
//
// Why one launch: the four workgroups of a sequence spin on each other (DualLink, the helper flags), so all of them must be
// resident at once.  Two launches on two streams left that to the dispatcher; one launch of a grid the host has checked against
// the occupancy query (engine.hip, fused_grid_fits) is resident as a whole, and the host pays one launch and one stream drain
// instead of two.  Results are those of the separate kernels, bit for bit (the same device functions run).
//
// Block -> (sequence, role).  Blocks are dealt round-robin over the 8 XCDs (block b on XCD b % 8: speed only, never relied on
// for correctness), each with its own L2.  The grid is four runs of rp = R rounded up to a multiple of 8 blocks, one run per role
// -- MFE main, PF main, MFE helper, PF helper -- and block k rp + r is sequence r: the four workgroups of a sequence sit on the
// SAME XCD (rp is a multiple of 8), so what they hand each other (rows, list rows, flags: agent-scope stores and loads) meets in
// that XCD's L2, and every XCD holds the same number of workgroups of each role.  Measured against round 4's first mapping (the
// four roles of a sequence on four XCDs, main roles on the even ones): the launch 0.4105 -> 0.4048 ms; the order of the runs does
// not matter (helpers first: 0.4049), interleaving the two main roles' runs loses half of it (0.4078).
#pragma once
#include "fold_mfe_dual.hpp"
#include "fold_pf_lds.hpp"

namespace drna {

enum : int { ROLE_MFE_MAIN = 0, ROLE_MFE_HELPER = 1, ROLE_PF_MAIN = 2, ROLE_PF_HELPER = 3 };
__host__ __device__ inline int fused_grid(int R) { return 4 * ((R + 7) / 8 * 8); }
__host__ __device__ inline void fused_block_role(int b, int& r, int& role, int grid) {
  const int rp = grid >> 2, k = b / rp;
  r = b - k * rp;
  role = k == 0 ? ROLE_MFE_MAIN : k == 1 ? ROLE_PF_MAIN : k == 2 ? ROLE_MFE_HELPER : ROLE_PF_HELPER;
}

// clk (optional, host-mapped): per block the 100 MHz wall clock at its start and end, from which the host reads the time of
// each fold without a profiler (the HIP events around the launch only see the whole kernel)
template <int NT>
__global__ __launch_bounds__(NT) void score_fused_kernel(MfeArgs MA, DualLink lk, PfArgs PA, EvalArgs EV, int R, long long* clk) {
  constexpr size_t B0 = sizeof(MfeFastSmem<NT>) > sizeof(MfeHelperSmem<NT>) ? sizeof(MfeFastSmem<NT>) : sizeof(MfeHelperSmem<NT>);
  constexpr size_t BYTES = B0 > sizeof(PfFastSmem<NT>) ? B0 : sizeof(PfFastSmem<NT>);
  __shared__ __attribute__((aligned(16))) unsigned char raw[BYTES];
  int r, role;
  fused_block_role(blockIdx.x, r, role, gridDim.x);
  if (r >= R) return;
  if (clk && threadIdx.x == 0) clk[2 * blockIdx.x] = wall_clock_100mhz();
  if (role == ROLE_MFE_MAIN || role == ROLE_MFE_HELPER) {
    lk.flagA += r * 64; lk.flagB += r * 64 + 32;
    lk.xs += (long long)r * 256;
    lk.xa = reinterpret_cast<int32_t*>(lk.xa) + (long long)r * 2 * (MFE_FAST_NMAX + 2) * XP;
    lk.xb = reinterpret_cast<int32_t*>(lk.xb) + (long long)r * 2 * (MFE_FAST_NMAX + 2) * XP;
    if (role == ROLE_MFE_HELPER) mfe_helper<NT>(*reinterpret_cast<MfeHelperSmem<NT>*>(raw), MA, r, lk);
    else mfe_lds_body<NT, true>(*reinterpret_cast<MfeFastSmem<NT>*>(raw), MA, r, lk);
  } else {
    pf_lds_body<NT, false>(*reinterpret_cast<PfFastSmem<NT>*>(raw), PA, EV, r, role == ROLE_PF_HELPER ? 1 : 0);      // (the fused launch always has helper workgroups: the instance without tile code)
  }
  if (clk && threadIdx.x == 0) clk[2 * blockIdx.x + 1] = wall_clock_100mhz();
}

}  // namespace drna
