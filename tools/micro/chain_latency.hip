// Microbenchmark: latency of dependent fp64 chains in ONE wave of ONE workgroup (what the single-workgroup dense
// kernels are made of).  hipcc --offload-arch=gfx950 -O3 -o chain_latency chain_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ double bcastd(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

__global__ void k_chain(double* out, long long* ticks, int reps) {
  double a = out[threadIdx.x], b = 1.0000001, c = 1e-9;
  long long t0 = wall_clock64();
  for (int i = 0; i < reps; ++i) {
#pragma unroll
    for (int k = 0; k < 32; ++k) a = fma(a, b, c);            // dependent FMA chain
  }
  long long t1 = wall_clock64();
  double x = a;
  for (int i = 0; i < reps; ++i) {
#pragma unroll
    for (int k = 0; k < 32; ++k) x -= b * bcastd(x, k);       // FMA + readlane broadcast chain (triangular solve step)
  }
  long long t2 = wall_clock64();
  double r = x;
  for (int i = 0; i < reps; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {                             // rcp + 2 Newton steps chain (pivot reciprocal)
      double q = __builtin_amdgcn_rcp(r);
      q = fma(fma(-r, q, 1.0), q, q);
      q = fma(fma(-r, q, 1.0), q, q);
      r = q + 1.5;
    }
  }
  long long t3 = wall_clock64();
  out[threadIdx.x] = a + x + r;
  if (threadIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = t2 - t1; ticks[2] = t3 - t2; }
}

int main() {
  double* d; long long* t;
  CK(hipMalloc(&d, 64 * 8)); CK(hipMalloc(&t, 3 * 8));
  CK(hipMemset(d, 0, 64 * 8));
  const int reps = 2000;
  for (int it = 0; it < 3; ++it) {
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, d, t, reps);
    CK(hipDeviceSynchronize());
    long long h[3];
    CK(hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost));
    printf("dependent fma: %.1f ns/step; fma + 2 readlane: %.1f ns/step; rcp + 2 Newton (+add): %.1f ns/step\n",
           10.0 * h[0] / (reps * 32.0), 10.0 * h[1] / (reps * 32.0), 10.0 * h[2] / (reps * 8.0));
  }
  return 0;
}
