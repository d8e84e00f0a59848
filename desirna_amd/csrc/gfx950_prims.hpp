// gfx950_prims.hpp -- the handful of CDNA4 primitives the fold kernels use that have no portable spelling: agent-scope (sc1)
// loads / stores, counted vmcnt waits, the wall clock behind the bounded spins, ds_bpermute, a three-operand fp64 FMA and the
// fp64 matrix instruction.  Everything else in the kernels is plain HIP C++ plus compiler builtins.  The CPU test emulation of
// the kernels (tests/emu) compiles the same kernel headers against its own stand-ins for THIS file (tests/emu/hip_emu_prims.h,
// selected through DRNA_PRIMS_HEADER in fold_common.hpp): no other product header knows about the emulator.
#pragma once
#include <hip/hip_runtime.h>

namespace drna {

template <typename T> __device__ __forceinline__ T ld_agent(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ void st_agent(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void spin_pause() { __builtin_amdgcn_s_sleep(2); }
__device__ __forceinline__ long long wall_clock_100mhz() { return (long long)wall_clock64(); }

// wait until at most N of the wave's vector-memory operations are outstanding (they retire in issue order)
template <int N>
__device__ __forceinline__ void stores_in_flight() {
  static_assert(N >= 0 && N <= 63, "stores_in_flight: the counter has six bits");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// A wait for another workgroup is bounded by TIME: HIP promises no dispatch order, so a partner may never have been scheduled,
// and the engine then redoes the call with one workgroup per fold -- which must not cost seconds.  10 ms of the 100 MHz wall
// clock is far beyond any legitimate wait (a whole 2046-nt fold by strips takes 35 ms and its strips wait for each other one
// diagonal at a time) and keeps a lost call in the tens of milliseconds.
constexpr long long SPIN_BUDGET_TICKS = 1000000;          // 10 ms at 100 MHz
struct SpinClock {
  long long t0 = wall_clock_100mhz();
  int n = 0;
  __device__ __forceinline__ bool expired() { return (++n & 15) == 0 && wall_clock_100mhz() - t0 > SPIN_BUDGET_TICKS; }
};

// all lanes of the wave have executed their LDS operations up to here (wave-private staging through LDS)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

// value of lane src_lane (per-lane index): ds_bpermute
__device__ __forceinline__ int lane_fetch_i32(int v, int src_lane) { return __builtin_amdgcn_ds_bpermute(src_lane * 4, v); }

// d = a * b + c with a destination of its own (the compiler prefers v_fmac + a copy)
__device__ __forceinline__ double fma3_f64(double a, double b, double c) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// v_mfma_f64_16x16x4_f64: A[l & 15][l >> 4], B[l >> 4][l & 15], D[(l >> 4) + 4 r][l & 15] in register r (CDNA4 guide)
typedef double f64x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f64x4 mfma_f64_16x16x4(double a, double b, f64x4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
// v_mfma_f64_4x4x4_4b_f64: four independent 4x4x4 products (block = (l >> 2) & 3): A[i][k] in lane i + 4 block + 16 k,
// B[k][j] in lane j + 4 block + 16 k, D[i][j] in lane j + 4 block + 16 i (found by experiment: tools/probe/mfma4x4_probe.hip)
__device__ __forceinline__ double mfma_f64_4x4x4_4b(double a, double b, double c) {
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

}  // namespace drna
