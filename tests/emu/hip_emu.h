// hip_emu.h -- TEST-ONLY stand-in for the handful of HIP device constructs the fold kernels use, so
// that the *unmodified* kernel sources (desirna_amd/csrc/*.hpp) can be compiled with g++ and their
// logic checked against the oracle in a container that has no GPU.  One workgroup is emulated at a
// time: every GPU thread is an OS thread, __syncthreads() is a real barrier, wave collectives
// (__shfl_xor, __ballot) rendezvous the 64 lanes of a wave.  Nothing here is part of the product.
#pragma once
// the kernels' hardware primitives (desirna_amd/csrc/gfx950_prims.hpp) are replaced by the stand-ins of hip_emu_prims.h: the one
// hook the product headers offer (fold_common.hpp); g++ is given -I tests/emu
#define DRNA_PRIMS_HEADER "hip_emu_prims.h"
#include <sched.h>
#include <pthread.h>
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

struct emu_dim3 { int x = 0, y = 0, z = 0; };
extern thread_local emu_dim3 threadIdx;
extern thread_local emu_dim3 blockIdx;     // per OS thread: two workgroups can be emulated side by side (emu_launch2)

struct emu_wave {
  pthread_barrier_t bar;
  unsigned long long slot[64];
};
struct emu_group {
  pthread_barrier_t bar;
  std::vector<emu_wave> waves;
};
extern thread_local emu_group* emu_g;

static inline void __syncthreads() { pthread_barrier_wait(&emu_g->bar); }

template <typename T>
static inline T emu_exchange(T v, int src_lane) {
  static_assert(sizeof(T) <= 8, "emu_exchange: 8 bytes at most");
  emu_wave& w = emu_g->waves[threadIdx.x >> 6];
  unsigned long long raw = 0;
  std::memcpy(&raw, &v, sizeof(T));
  w.slot[threadIdx.x & 63] = raw;
  pthread_barrier_wait(&w.bar);
  unsigned long long got = w.slot[src_lane & 63];
  pthread_barrier_wait(&w.bar);
  T out;
  std::memcpy(&out, &got, sizeof(T));
  return out;
}
template <typename T>
static inline T __shfl_xor(T v, int mask) { return emu_exchange(v, (threadIdx.x & 63) ^ mask); }
template <typename T>
static inline T __shfl(T v, int lane) { return emu_exchange(v, lane); }

static inline unsigned long long __ballot(bool p) {
  emu_wave& w = emu_g->waves[threadIdx.x >> 6];
  w.slot[threadIdx.x & 63] = p ? 1ull : 0ull;
  pthread_barrier_wait(&w.bar);
  unsigned long long m = 0;
  for (int k = 0; k < 64; k++) m |= (w.slot[k] & 1ull) << k;
  pthread_barrier_wait(&w.bar);
  return m;
}
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __builtin_amdgcn_readfirstlane(int x) { return emu_exchange(x, 0); }   // all lanes of the wave are active wherever the kernels use it
static inline int __builtin_amdgcn_readlane(int v, int lane) { return emu_exchange(v, lane); }
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  (void)bank_mask;
  const int L = threadIdx.x & 63, r = L >> 4;
  if (ctrl == 0xE4) return src;        // identity quad_perm (as_vector()): no lane talks to another, no rendezvous needed
  int from = -1;
  if (ctrl >= 0x111 && ctrl <= 0x11F) { const int nsh = ctrl - 0x110; if ((L & 15) >= nsh) from = L - nsh; }
  else if (ctrl >= 0x150 && ctrl <= 0x15F) from = r * 16 + (ctrl - 0x150);     // row_newbcast: lane n of the own row
  else if (ctrl == 0x142) { if (r > 0) from = r * 16 - 1; }
  else if (ctrl == 0x143) { if (r >= 2) from = 31; }
  const int got = emu_exchange(src, from < 0 ? L : from);       // every lane takes part in the rendezvous
  if (!((row_mask >> r) & 1)) return old;
  if (from < 0) return bound_ctrl ? 0 : old;
  return got;
}
static inline int __double2loint(double v) { unsigned long long b; std::memcpy(&b, &v, 8); return (int)(b & 0xffffffffull); }
static inline int __double2hiint(double v) { unsigned long long b; std::memcpy(&b, &v, 8); return (int)(b >> 32); }
static inline double __hiloint2double(int hi, int lo) {
  unsigned long long b = ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo; double v; std::memcpy(&v, &b, 8); return v;
}
// buffer descriptor + raw buffer loads (byte offsets)
struct emu_rsrc { const char* p; };
struct emu_u32x2 { unsigned v[2]; unsigned operator[](int k) const { return v[k]; } };
static inline emu_rsrc __builtin_amdgcn_make_buffer_rsrc(void* p, short, int, int) { return emu_rsrc{(const char*)p}; }
static inline emu_u32x2 __builtin_amdgcn_raw_buffer_load_b64(emu_rsrc r, int voff, int soff, int) {
  emu_u32x2 o; std::memcpy(o.v, r.p + voff + soff, 8); return o;
}
static inline unsigned __builtin_amdgcn_raw_buffer_load_b32(emu_rsrc r, int voff, int soff, int) {
  unsigned o; std::memcpy(&o, r.p + voff + soff, 4); return o;
}
struct emu_u32x4 { unsigned v[4]; unsigned operator[](int k) const { return v[k]; } };
static inline emu_u32x4 __builtin_amdgcn_raw_buffer_load_b128(emu_rsrc r, int voff, int soff, int) {
  emu_u32x4 o; std::memcpy(o.v, r.p + voff + soff, 16); return o;
}
static inline void __builtin_amdgcn_s_barrier() { pthread_barrier_wait(&emu_g->bar); }
#define __builtin_amdgcn_fence(...) ((void)0)
static inline int atomicMin(int* p, int v) {
  int old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (v < old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
using std::min;
using std::max;

// run workgroups of nt threads each, all at the same time (one function per workgroup; block index = first + position)
static inline void emu_launch_many(int first_block, int nt, const std::vector<std::function<void()>>& fns) {
  std::vector<emu_group> gs(fns.size());
  for (auto& g : gs) {
    g.waves.resize(nt / 64);
    pthread_barrier_init(&g.bar, nullptr, nt);
    for (auto& w : g.waves) pthread_barrier_init(&w.bar, nullptr, 64);
  }
  std::vector<std::thread> th;
  th.reserve((size_t)nt * fns.size());
  for (size_t b = 0; b < fns.size(); b++)
    for (int t = 0; t < nt; t++)
      th.emplace_back([t, b, first_block, &gs, &fns]() {
        threadIdx.x = t;
        blockIdx.x = first_block + (int)b;
        emu_g = &gs[b];
        fns[b]();
      });
  for (auto& t : th) t.join();
  for (auto& g : gs) {
    pthread_barrier_destroy(&g.bar);
    for (auto& w : g.waves) pthread_barrier_destroy(&w.bar);
  }
}
// run one workgroup of nt threads executing fn()
static inline void emu_launch(int block, int nt, const std::function<void()>& fn) {
  emu_launch_many(block, nt, std::vector<std::function<void()>>{fn});
}
