// Launch-floor probe: a chain of K small dependent kernels issued (a) one by one on a stream, (b) as one captured
// hipGraph.  usage: graph_probe [K] [n]   (n = elements each kernel touches)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void step_kernel(double* a, long n, double s)
{
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = a[i] * s + 1.0;
}
int main(int argc, char** argv)
{
  const int K = argc > 1 ? atoi(argv[1]) : 40;
  const long n = argc > 2 ? atol(argv[2]) : 4096;
  double* a;
  CK(hipMalloc(&a, sizeof(double) * n));
  CK(hipMemset(a, 0, sizeof(double) * n));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  auto chain = [&]() { for (int k = 0; k < K; ++k) hipLaunchKernelGGL(step_kernel, grid, block, 0, st, a, n, 0.5); };
  for (int w = 0; w < 5; ++w) chain();
  CK(hipStreamSynchronize(st));
  const int reps = 200;
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r) chain();
  CK(hipStreamSynchronize(st));
  auto t1 = std::chrono::steady_clock::now();
  const double us_stream = std::chrono::duration<double, std::micro>(t1 - t0).count() / reps;
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
  chain();
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int w = 0; w < 5; ++w) CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  t1 = std::chrono::steady_clock::now();
  const double us_graph = std::chrono::duration<double, std::micro>(t1 - t0).count() / reps;
  // each replay followed by a sync (a step ends with one read-back)
  t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r) { CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st)); }
  t1 = std::chrono::steady_clock::now();
  const double us_graph_sync = std::chrono::duration<double, std::micro>(t1 - t0).count() / reps;
  t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r) { chain(); CK(hipStreamSynchronize(st)); }
  t1 = std::chrono::steady_clock::now();
  const double us_stream_sync = std::chrono::duration<double, std::micro>(t1 - t0).count() / reps;
  printf("K=%d n=%ld: stream %.1f us (%.2f per kernel), graph %.1f us (%.2f per kernel); with a sync per chain: stream %.1f, graph %.1f us\n",
         K, n, us_stream, us_stream / K, us_graph, us_graph / K, us_stream_sync, us_graph_sync);
  return 0;
}
