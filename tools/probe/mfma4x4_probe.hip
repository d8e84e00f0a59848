// Lane layout and issue rate of v_mfma_f64_4x4x4_4b_f64 on gfx950, found by experiment (one-hot A, numbered B).
// Build: hipcc --offload-arch=gfx950 -O2 -o mfma4x4_probe mfma4x4_probe.hip ; prints a table tools/probe/README reads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(double* out) {
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; la++) {
    const double a = lane == la ? 1.0 : 0.0, b = 1.0 + lane;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[la * 64 + lane] = d;
  }
}

__global__ void rate(long long* clk, double* sink, int n) {
  const int lane = threadIdx.x & 63;
  double a = 1.0 + lane * 1e-9, b = 1.0 - lane * 1e-9;
  double d0 = 0, d1 = 0, d2 = 0, d3 = 0;
  long long t0 = clock64();
  for (int k = 0; k < n; k++) d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d0, 0, 0, 0);
  long long t1 = clock64();
  for (int k = 0; k < n; k++) {
    d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d0, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d1, 0, 0, 0);
    d2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d2, 0, 0, 0);
    d3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d3, 0, 0, 0);
  }
  long long t2 = clock64();
  if (threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = t2 - t1; }
  sink[threadIdx.x] = d0 + d1 + d2 + d3;
}

int main() {
  double* out; long long* clk; double* sink;
  hipMalloc(&out, 64 * 64 * 8); hipMalloc(&clk, 64); hipMalloc(&sink, 1024 * 8);
  probe<<<1, 64>>>(out);
  std::vector<double> h(64 * 64);
  hipMemcpy(h.data(), out, 64 * 64 * 8, hipMemcpyDeviceToHost);
  for (int la = 0; la < 64; la++) {
    printf("A lane %2d ->", la);
    for (int o = 0; o < 64; o++) if (h[la * 64 + o] != 0.0) printf("  D lane %2d = B lane %2d", o, (int)h[la * 64 + o] - 1);
    printf("\n");
  }
  const int n = 4096;
  rate<<<1, 64>>>(clk, sink, n);
  long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
  printf("dependent chain: %.1f clk64 ticks per mfma; four independent chains: %.1f per mfma (one wave)\n", (double)c[0] / n, (double)c[1] / (4.0 * n));
  rate<<<1, 256>>>(clk, sink, n);
  hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
  printf("four waves (one per SIMD): dependent %.1f, independent %.1f ticks per mfma and wave\n", (double)c[0] / n, (double)c[1] / (4.0 * n));
  rate<<<1, 1024>>>(clk, sink, n);
  hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
  printf("sixteen waves: dependent %.1f, independent %.1f ticks per mfma and wave\n", (double)c[0] / n, (double)c[1] / (4.0 * n));
  return 0;
}
