// How many long kernels of G work-groups (one per CU: 150 KB of LDS) run side by side when launched on S streams?
// (persistent-launch experiment, DESIGN.md 5.6: 8 launches of 32 work-groups DO share the chip with GPU_MAX_HW_QUEUES=8, 4 with the
//  default).  Build: hipcc --offload-arch=gfx950 -O3 -o concur concur.hip   (output: concur_r03.txt)
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
int main(int argc, char** argv) {
  const int spin_us = argc > 1 ? atoi(argv[1]) : 1000;
  CK(hipFuncSetAttribute((const void*)k_spin, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  int* sink; CK(hipMalloc(&sink, 4096));
  printf("spin %d us per kernel; GPU_MAX_HW_QUEUES=%s\n", spin_us, getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(default)");
  const int cfg[][3] = {{32, 1, 1}, {32, 2, 1}, {32, 4, 1}, {32, 6, 1}, {32, 8, 1}, {32, 12, 1}, {32, 16, 1}, {64, 4, 1}, {64, 8, 1}, {256, 1, 1}, {256, 2, 1},
                        {32, 8, 3}, {32, 8, 5}};
  for (auto& c : cfg) {
    const int G = c[0], S = c[1], R = c[2];
    std::vector<hipStream_t> st(S);
    for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (auto& s : st) hipLaunchKernelGGL(k_spin, dim3(G), dim3(512), 150 * 1024, s, 100, sink);
    CK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < R; ++r)
      for (auto& s : st) hipLaunchKernelGGL(k_spin, dim3(G), dim3(512), 150 * 1024, s, 100 * spin_us, sink);
    CK(hipDeviceSynchronize());
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("%3d work-groups x %2d streams x %d kernels each: %8.1f us = %.2f x one kernel; %5.1f CUs busy on average\n", G, S, R, us, us / spin_us,
           (double)G * S * R * spin_us / us);
    for (auto& s : st) CK(hipStreamDestroy(s));
  }
  return 0;
}
