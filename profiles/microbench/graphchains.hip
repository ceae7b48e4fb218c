// How many dependent kernel chains does the chip run side by side, and do hipGraph branches fill each other's kernel
// boundaries?  Every kernel = 64 work-groups x 512 threads with 150 KB of LDS (one work-group per CU, like k_sep2 at
// 128-frame tiles) spinning for SPIN_US; a chain = N such kernels, each depending on the previous one.
//   A: S streams, one graph of ONE chain each         B: one stream, one graph of C independent chains
//   C: S streams, one graph of C chains each
// Build: hipcc --offload-arch=gfx950 -O3 -o graphchains graphchains.hip     (GPU_MAX_HW_QUEUES is read from the environment)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(512) k_spin(int ticks, int* sink) {
  extern __shared__ int sm[];
  const long long t0 = __builtin_amdgcn_s_memrealtime();
  while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
  if (ticks < 0) sink[threadIdx.x] = sm[threadIdx.x];
}

static unsigned g_seed = 1;
static int g_jitter = 0;                                    // 1: every kernel spins 0.6 .. 1.4 x the nominal time (same mean)
static hipGraphExec_t make_graph(int chains, int n, int ticks0, int* sink) {
  hipGraph_t g;
  CK(hipGraphCreate(&g, 0));
  int ticks = ticks0;
  void* args[2] = {&ticks, &sink};
  hipKernelNodeParams p = {};
  p.func = (void*)k_spin;
  p.gridDim = dim3(64);
  p.blockDim = dim3(512);
  p.sharedMemBytes = 150 * 1024;
  p.kernelParams = args;
  for (int c = 0; c < chains; ++c) {
    hipGraphNode_t prev = nullptr;
    for (int i = 0; i < n; ++i) {
      hipGraphNode_t node;
      g_seed = g_seed * 1664525u + 1013904223u;
      ticks = g_jitter ? (int)(ticks0 * (0.6 + 0.8 * ((g_seed >> 8) & 0xffff) / 65535.0)) : ticks0;
      CK(hipGraphAddKernelNode(&node, g, prev ? &prev : nullptr, prev ? 1 : 0, &p));
      prev = node;
    }
  }
  hipGraphExec_t ex;
  CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  return ex;
}

static double run(int streams, int chains, int n, int ticks, int* sink) {
  std::vector<hipStream_t> st(streams);
  std::vector<hipGraphExec_t> ex(streams);
  for (int s = 0; s < streams; ++s) {
    CK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking));
    ex[s] = make_graph(chains, n, ticks, sink);
  }
  for (int s = 0; s < streams; ++s) CK(hipGraphLaunch(ex[s], st[s]));   // warm-up (upload)
  CK(hipDeviceSynchronize());
  const auto t0 = std::chrono::steady_clock::now();
  for (int s = 0; s < streams; ++s) CK(hipGraphLaunch(ex[s], st[s]));
  CK(hipDeviceSynchronize());
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  for (int s = 0; s < streams; ++s) { CK(hipGraphExecDestroy(ex[s])); CK(hipStreamDestroy(st[s])); }
  return us;
}

int main(int argc, char** argv) {
  const int n = 60, spin_us = argc > 1 ? atoi(argv[1]) : 12, ticks = 100 * spin_us;
  CK(hipFuncSetAttribute((const void*)k_spin, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  int* sink;
  CK(hipMalloc(&sink, 4096));
  printf("kernel = 64 work-groups spinning %d us, chain = %d kernels; GPU_MAX_HW_QUEUES=%s\n", spin_us, n,
         getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(default)");
  g_jitter = argc > 2 ? atoi(argv[2]) : 0;
  printf("jitter %d\n", g_jitter);
  const int cfg[][2] = {{1, 1}, {2, 1}, {4, 1}, {5, 1}, {6, 1}, {8, 1}, {1, 2}, {1, 4}, {1, 8}, {2, 2}, {4, 2}, {2, 4}, {4, 3}};
  for (auto& c : cfg) {
    const double us = run(c[0], c[1], n, ticks, sink);
    const int nch = c[0] * c[1];
    printf("%d stream(s) x graph of %d chain(s): %8.1f us  -> %6.2f us per kernel and chain, %5.2f chains' worth of the chip's 4 x 64 CUs busy\n",
           c[0], c[1], us, us / n, nch * n * (double)spin_us / us);
  }
  return 0;
}
