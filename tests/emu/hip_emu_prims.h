// hip_emu_prims.h -- CPU stand-ins for desirna_amd/csrc/gfx950_prims.hpp (selected by -DDRNA_PRIMS_HEADER, see fold_common.hpp):
// the kernels' hardware primitives restated with OS threads.  Test infrastructure only.
#pragma once
#include <cmath>
#include <sched.h>

namespace drna {

template <typename T> __device__ __forceinline__ T ld_agent(const T* p) { T v; __atomic_load(p, &v, __ATOMIC_ACQUIRE); return v; }
template <typename T> __device__ __forceinline__ void st_agent(T* p, T v) { __atomic_store(p, &v, __ATOMIC_RELEASE); }
__device__ __forceinline__ void drain_vmem() {}
__device__ __forceinline__ void spin_pause() { sched_yield(); }
__device__ __forceinline__ long long wall_clock_100mhz() { return 0; }
template <int N> __device__ __forceinline__ void stores_in_flight() {}

// the emulation counts polls instead of reading a clock
constexpr int SPIN_LIMIT = 1 << 22;
struct SpinClock {
  int n = 0;
  __device__ __forceinline__ bool expired() { return ++n > SPIN_LIMIT; }
};

__device__ __forceinline__ void wave_lds_sync() { pthread_barrier_wait(&emu_g->waves[threadIdx.x >> 6].bar); }
__device__ __forceinline__ int lane_fetch_i32(int v, int src_lane) { return emu_exchange(v, src_lane); }
__device__ __forceinline__ double fma3_f64(double a, double b, double c) { return a * b + c; }

struct f64x4 { double v[4]; double& operator[](int k) { return v[k]; } double operator[](int k) const { return v[k]; } };
// v_mfma_f64_16x16x4_f64 as the CDNA4 guide gives it: A[l & 15][l >> 4], B[l >> 4][l & 15], D[(l >> 4) + 4 r][l & 15] in register r
__device__ __forceinline__ f64x4 mfma_f64_16x16x4(double a, double b, f64x4 c) {
  const int lane = threadIdx.x & 63, col = lane & 15;
  for (int k = 0; k < 4; k++) {
    const double bk = emu_exchange(b, col + 16 * k);
    for (int r = 0; r < 4; r++) {
      const double ak = emu_exchange(a, (lane >> 4) + 4 * r + 16 * k);
      c.v[r] = std::fma(ak, bk, c.v[r]);
    }
  }
  return c;
}
// v_mfma_f64_4x4x4_4b_f64 (lane layout: gfx950_prims.hpp)
__device__ __forceinline__ double mfma_f64_4x4x4_4b(double a, double b, double c) {
  const int lane = threadIdx.x & 63, j = lane & 3, blk4 = lane & 12, i = lane >> 4;
  for (int k = 0; k < 4; k++) {
    const double ak = emu_exchange(a, i + blk4 + 16 * k), bk = emu_exchange(b, j + blk4 + 16 * k);
    c = std::fma(ak, bk, c);
  }
  return c;
}

}  // namespace drna
