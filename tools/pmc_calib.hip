// pmc_calib.hip -- calibration kernels for the SQ counters used in the roofline (tools/gpu_round2.sh): known numbers
// of conflict-free ds_read_b64 / ds_read_b32 and of dependent-free v_add_u32 per wave, so that the units of
// SQ_LDS_IDX_ACTIVE, SQ_ACTIVE_INST_LDS, SQ_ACTIVE_INST_VALU and SQ_BUSY_CYCLES can be read off a profile.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int N = 8192;
__global__ __launch_bounds__(256) void calib_lds_b64(double* out) {
  __shared__ double buf[512];
  buf[threadIdx.x] = threadIdx.x; buf[threadIdx.x + 256] = 1.0;
  __syncthreads();
  double acc = 0.0;
  const int lane = threadIdx.x & 63;
  for (int k = 0; k < N; k += 8) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      double v;
      asm volatile("ds_read_b64 %0, %1 offset:%2\n" : "=v"(v) : "v"(lane * 8), "n"(u * 512 % 2048));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      acc += v;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void calib_lds_b32(float* out) {
  __shared__ float buf[1024];
  for (int k = threadIdx.x; k < 1024; k += 256) buf[k] = k;
  __syncthreads();
  float acc = 0.f;
  const int lane = threadIdx.x & 63;
  for (int k = 0; k < N; k += 8) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      float v;
      asm volatile("ds_read_b32 %0, %1 offset:%2\n" : "=v"(v) : "v"(lane * 4), "n"(u * 256));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      acc += v;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void calib_valu(unsigned* out) {
  unsigned a = threadIdx.x, b = blockIdx.x;
  for (int k = 0; k < N; k += 8) {
#pragma unroll
    for (int u = 0; u < 8; u++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
  }
  out[blockIdx.x * 256 + threadIdx.x] = a;
}
int main() {
  void* d; hipMalloc(&d, 256 * 256 * 8);
  for (int r = 0; r < 3; r++) {
    calib_lds_b64<<<256, 256>>>((double*)d);
    calib_lds_b32<<<256, 256>>>((float*)d);
    calib_valu<<<256, 256>>>((unsigned*)d);
  }
  hipDeviceSynchronize();
  printf("calib: 256 blocks x 4 waves x %d instructions per kernel\n", N);
  return 0;
}
