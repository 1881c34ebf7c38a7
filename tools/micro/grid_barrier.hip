// Microbenchmark: what a dependency between two phases costs on one MI355X --
//   (a) a kernel boundary on a stream (chain of dependent launches of W workgroups that each do a token amount of work),
//   (b) a grid-wide barrier inside ONE persistent kernel of W co-resident workgroups (arrive counter + generation flag in
//       global memory, device-scope release / acquire: data written before the barrier by any workgroup is read after it
//       by any other, across XCDs -- what a fused level kernel of the solver would need between levels).
// hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip && ./grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_phase(double* a, int stride) {           // one phase: every workgroup reads its left neighbour's slot and writes its own
  const int w = blockIdx.x, nw = gridDim.x;
  if (threadIdx.x == 0) a[(size_t)w * stride] = a[(size_t)((w + nw - 1) % nw) * stride] * 0.5 + 1.0;
}

__global__ void k_persistent(double* a, int stride, int phases, unsigned* arrive, volatile unsigned* gen, unsigned long long max_spin, int* timed_out) {
  const int w = blockIdx.x, nw = gridDim.x;
  for (int p = 0; p < phases; ++p) {
    if (threadIdx.x == 0) a[(size_t)w * stride] = a[(size_t)((w + nw - 1) % nw) * stride] * 0.5 + 1.0;
    // grid barrier
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();                                       // release: this workgroup's writes visible device-wide
      const unsigned g = *gen;
      if (atomicAdd(arrive, 1u) == (unsigned)nw - 1) {
        *arrive = 0;
        __threadfence();
        atomicAdd((unsigned*)gen, 1u);
      } else {
        unsigned long long spins = 0;
        while (*gen == g) { if (++spins > max_spin) { *timed_out = 1; break; } __builtin_amdgcn_s_sleep(1); }   // (exit condition every wave reaches)
      }
      __threadfence();                                       // acquire
    }
    __syncthreads();
  }
}

int main() {
  const int phases = 200, reps = 20, stride = 64;
  double* a; unsigned *arrive, *gen; int* to;
  CK(hipMalloc(&a, 4096 * stride * 8)); CK(hipMemset(a, 0, 4096 * stride * 8));
  CK(hipMalloc(&arrive, 4)); CK(hipMalloc(&gen, 4)); CK(hipMalloc(&to, 4));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int nw : {64, 256, 512, 1024}) {                       // (1024 workgroups of 64 threads are co-resident on 256 CUs)
    for (int i = 0; i < phases; ++i) hipLaunchKernelGGL(k_phase, dim3(nw), dim3(64), 0, st, a, stride);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r)
      for (int i = 0; i < phases; ++i) hipLaunchKernelGGL(k_phase, dim3(nw), dim3(64), 0, st, a, stride);
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms_launch; CK(hipEventElapsedTime(&ms_launch, e0, e1));
    CK(hipMemset(arrive, 0, 4)); CK(hipMemset(gen, 0, 4)); CK(hipMemset(to, 0, 4));
    hipLaunchKernelGGL(k_persistent, dim3(nw), dim3(64), 0, st, a, stride, phases, arrive, gen, 4000000ull, to);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_persistent, dim3(nw), dim3(64), 0, st, a, stride, phases, arrive, gen, 4000000ull, to);
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms_bar; CK(hipEventElapsedTime(&ms_bar, e0, e1));
    int h_to = 0; CK(hipMemcpy(&h_to, to, 4, hipMemcpyDeviceToHost));
    printf("%5d workgroups: kernel boundary %.2f us per phase, grid barrier %.2f us per phase%s\n", nw,
           1e3 * ms_launch / reps / phases, 1e3 * ms_bar / reps / phases, h_to ? "  (SPIN LIMIT HIT)" : "");
  }
  return 0;
}
