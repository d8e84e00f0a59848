// pingpong.hip -- micro-benchmark for the two-workgroup role split: how fast can two workgroups on different CUs hand rows
// to each other through L2/fabric?  Pair p = blocks 2p (A) and 2p+1 (B).  Protocol per the CDNA4 guide: payload stored with
// sc1 (write-through) stores, every storing wave drains vmcnt, workgroup barrier, one lane stores the flag (sc1); the consumer
// polls the flag with sc1 loads from one lane, workgroup barrier, then loads the payload with sc1 loads.
// Modes: 0 = strict ping-pong (A publishes row k, waits for B's row k, ...): round trip; LAG > 0 = A publishes row k and
// needs B's row k - LAG (pipelined, what the fold kernels would do).  Rows of 200 doubles, 128-byte aligned pitch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
constexpr int PITCH = 208, NROW = 256;
__device__ __forceinline__ int ld_flag(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_flag(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_d(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_d(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ bool wait_flag(const int* f, int target, int* lds_ok) {
  // one lane polls, the workgroup learns the outcome through LDS + barrier
  if (threadIdx.x == 0) {
    int ok = 0;
    for (int spin = 0; spin < (1 << 20); spin++) {
      if (ld_flag(f) >= target) { ok = 1; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    *lds_ok = ok;
  }
  __syncthreads();
  return *lds_ok != 0;
}

__global__ __launch_bounds__(1024) void pingpong(double* rowsA, double* rowsB, int* flags, int nsteps, int lag, int work, long long* clk, int* errs) {
  __shared__ int ok;
  __shared__ double scratch[1024];
  const int pair = blockIdx.x >> 1, role = blockIdx.x & 1, tid = threadIdx.x;
  double* mine = (role ? rowsB : rowsA) + (size_t)pair * NROW * PITCH;
  const double* theirs = (role ? rowsA : rowsB) + (size_t)pair * NROW * PITCH;
  int* fmine = flags + (pair * 2 + role) * 32;
  const int* ftheirs = flags + (pair * 2 + (role ^ 1)) * 32;
  long long t0 = clock64();
  double acc = 0.0;
  int bad = 0;
  for (int k = 1; k <= nsteps; k++) {
    const int row = k % NROW;
    if (role == 0) {
      // A: publish row k, then consume B's row k - lag
      if (tid < 200) st_d(mine + row * PITCH + tid, (double)(k * 1000 + tid));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) st_flag(fmine, k);
      const int need = k - lag;
      if (need >= 1) {
        if (!wait_flag(ftheirs, need, &ok)) { bad = 1; break; }
        if (tid < 200) {
          const double v = ld_d(theirs + (need % NROW) * PITCH + tid);
          if (v != (double)(need * 1000 + tid) * 2.0) bad++;
          acc += v;
        }
      }
    } else {
      // B: wait for A's row k, "compute", publish row k
      if (!wait_flag(ftheirs, k, &ok)) { bad = 1; break; }
      double v = 0.0;
      if (tid < 200) v = ld_d(theirs + row * PITCH + tid);
      if (tid < 200 && v != (double)(k * 1000 + tid)) bad++;
      if (tid < 200) st_d(mine + row * PITCH + tid, v * 2.0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) st_flag(fmine, k);
    }
    // busy work standing in for the fold's own step (LDS traffic + ALU)
    for (int w = 0; w < work; w++) { scratch[tid] = acc + w; __syncthreads(); acc += scratch[(tid * 7 + w) & 1023]; }
  }
  __syncthreads();
  if (tid == 0) { clk[blockIdx.x] = clock64() - t0; }
  if (bad) atomicAdd(errs, bad);
  if (acc == 12345.678) clk[0] = 0;
}

int main(int argc, char** argv) {
  const int pairs = argc > 1 ? atoi(argv[1]) : 128, nsteps = argc > 2 ? atoi(argv[2]) : 2000;
  double *a, *b; int* f; long long* clk; int* errs;
  hipMalloc(&a, (size_t)pairs * NROW * PITCH * 8); hipMalloc(&b, (size_t)pairs * NROW * PITCH * 8);
  hipMalloc(&f, pairs * 2 * 32 * 4); hipMalloc(&clk, pairs * 2 * 8); hipMalloc(&errs, 4);
  for (int work : {0, 4, 16})
    for (int lag : {0, 1, 2, 4, 8}) {
      hipMemset(f, 0, pairs * 2 * 32 * 4); hipMemset(errs, 0, 4);
      hipMemset(a, 0, (size_t)pairs * NROW * PITCH * 8); hipMemset(b, 0, (size_t)pairs * NROW * PITCH * 8);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      pingpong<<<pairs * 2, 1024>>>(a, b, f, nsteps, lag, work, clk, errs);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      int herr = 0; hipMemcpy(&herr, errs, 4, hipMemcpyDeviceToHost);
      printf("pairs %d work %2d lag %d: %.3f ms for %d steps = %.3f us/step, errors %d\n", pairs, work, lag, ms, nsteps, ms * 1e3 / nsteps, herr);
      fflush(stdout);
    }
  return 0;
}
