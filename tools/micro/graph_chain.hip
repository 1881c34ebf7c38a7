// Microbenchmark: a chain of N dependent small kernels, launched on a stream vs replayed as a hipGraph.
// hipcc --offload-arch=gfx950 -O3 -o graph_chain graph_chain.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_step(double* a, const double* b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = a[i] * 0.999 + b[i];
}

int main(int argc, char** argv) {
  const int chain = argc > 1 ? atoi(argv[1]) : 100;
  const int reps = 50;
  for (size_t n : {(size_t)1 << 12, (size_t)1 << 16, (size_t)1 << 20, (size_t)1 << 23}) {
    double *a, *b;
    CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8));
    CK(hipMemset(a, 0, n * 8)); CK(hipMemset(b, 0, n * 8));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)((n + 255) / 256);
    auto run_stream = [&]() { for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(k_step, dim3(grid), dim3(256), 0, st, a, b, n); };
    run_stream(); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) run_stream();
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms_stream; CK(hipEventElapsedTime(&ms_stream, e0, e1));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    run_stream();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms_graph; CK(hipEventElapsedTime(&ms_graph, e0, e1));
    // host cost of enqueueing
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) run_stream();
    auto t1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(st));
    auto t2 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
    auto t3 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(st));
    printf("n=%8zu chain=%d: stream %.2f us/kernel, graph %.2f us/kernel; host enqueue stream %.2f us/kernel, graph %.2f us/kernel\n", n, chain,
           1e3 * ms_stream / reps / chain, 1e3 * ms_graph / reps / chain,
           std::chrono::duration<double, std::micro>(t1 - t0).count() / reps / chain,
           std::chrono::duration<double, std::micro>(t3 - t2).count() / reps / chain);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    CK(hipFree(a)); CK(hipFree(b)); CK(hipStreamDestroy(st));
  }
  return 0;
}
