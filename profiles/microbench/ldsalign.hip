// LDS access cost by width and ALIGNMENT on gfx950 (asm volatile reads: nothing is hoisted).  One work-group of 8 waves,
// each issues ITERS x 8 reads and waits; reported: cycles of the slowest wave / (8 waves x reads per wave) = LDS cycles
// per wave-instruction.  Build: hipcc --offload-arch=gfx950 -O3 -o ldsalign ldsalign.hip   (output: ldsalign_r03.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v3i __attribute__((ext_vector_type(3)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <int W, int MIS, int STRIDE>
__global__ void __launch_bounds__(512) k(long long* prof, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char s[65536];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 65536; i += 512) s[i] = (unsigned char)i;
  __syncthreads();
  unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)s + wave * 4096 + lane * STRIDE + MIS;
  v4i a = {0,0,0,0}, b = a, c = a, d = a;
  v3i a3 = {0,0,0}, b3 = a3, c3 = a3, d3 = a3;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (W == 16) asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n"
                              "ds_read_b128 %0, %4 offset:32\n ds_read_b128 %1, %4 offset:1056\n ds_read_b128 %2, %4 offset:2080\n ds_read_b128 %3, %4 offset:3104\n s_waitcnt lgkmcnt(0)"
                              : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(base));
    if (W == 8) asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:1024\n ds_read_b64 %2, %4 offset:2048\n ds_read_b64 %3, %4 offset:3072\n"
                              "ds_read_b64 %0, %4 offset:32\n ds_read_b64 %1, %4 offset:1056\n ds_read_b64 %2, %4 offset:2080\n ds_read_b64 %3, %4 offset:3104\n s_waitcnt lgkmcnt(0)"
                              : "=&v"(*(v2i*)&a), "=&v"(*(v2i*)&b), "=&v"(*(v2i*)&c), "=&v"(*(v2i*)&d) : "v"(base));
    if (W == 4) asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:1024\n ds_read_b32 %2, %4 offset:2048\n ds_read_b32 %3, %4 offset:3072\n"
                              "ds_read_b32 %0, %4 offset:32\n ds_read_b32 %1, %4 offset:1056\n ds_read_b32 %2, %4 offset:2080\n ds_read_b32 %3, %4 offset:3104\n s_waitcnt lgkmcnt(0)"
                              : "=&v"(a[0]), "=&v"(b[0]), "=&v"(c[0]), "=&v"(d[0]) : "v"(base));
    if (W == 12) asm volatile("ds_read_b96 %0, %4\n ds_read_b96 %1, %4 offset:1024\n ds_read_b96 %2, %4 offset:2048\n ds_read_b96 %3, %4 offset:3072\n"
                              "ds_read_b96 %0, %4 offset:32\n ds_read_b96 %1, %4 offset:1056\n ds_read_b96 %2, %4 offset:2080\n ds_read_b96 %3, %4 offset:3104\n s_waitcnt lgkmcnt(0)"
                              : "=&v"(a3), "=&v"(b3), "=&v"(c3), "=&v"(d3) : "v"(base));
    if (W == 116) asm volatile("ds_write_b128 %4, %0\n ds_write_b128 %4, %1 offset:1024\n ds_write_b128 %4, %2 offset:2048\n ds_write_b128 %4, %3 offset:3072\n"
                              "ds_write_b128 %4, %0 offset:32\n ds_write_b128 %4, %1 offset:1056\n ds_write_b128 %4, %2 offset:2080\n ds_write_b128 %4, %3 offset:3104\n s_waitcnt lgkmcnt(0)"
                              : : "v"(a), "v"(b), "v"(c), "v"(d), "v"(base) : "memory");
    if (W == 104) asm volatile("ds_write_b32 %4, %0\n ds_write_b32 %4, %1 offset:1024\n ds_write_b32 %4, %2 offset:2048\n ds_write_b32 %4, %3 offset:3072\n"
                              "ds_write_b32 %4, %0 offset:32\n ds_write_b32 %4, %1 offset:1056\n ds_write_b32 %4, %2 offset:2080\n ds_write_b32 %4, %3 offset:3104\n s_waitcnt lgkmcnt(0)"
                              : : "v"(a[0]), "v"(b[0]), "v"(c[0]), "v"(d[0]), "v"(base) : "memory");
    if (W == 99) asm volatile("ds_read_b64_tr_b8 %0, %4\n ds_read_b64_tr_b8 %1, %4 offset:1024\n ds_read_b64_tr_b8 %2, %4 offset:2048\n ds_read_b64_tr_b8 %3, %4 offset:3072\n"
                              "ds_read_b64_tr_b8 %0, %4 offset:32\n ds_read_b64_tr_b8 %1, %4 offset:1056\n ds_read_b64_tr_b8 %2, %4 offset:2080\n ds_read_b64_tr_b8 %3, %4 offset:3104\n s_waitcnt lgkmcnt(0)"
                              : "=&v"(*(v2i*)&a), "=&v"(*(v2i*)&b), "=&v"(*(v2i*)&c), "=&v"(*(v2i*)&d) : "v"(base));
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) prof[wave] = t1 - t0;
  if (a[0] + b[0] + c[0] + d[0] + a3[0] + b3[1] + c3[2] + d3[0] == 0x7fffffff) prof[8] = 1;
}
template <int W, int MIS, int STRIDE>
void run(const char* name) {
  long long* dp; CK(hipMalloc(&dp, 128));
  const int iters = 512;
  hipLaunchKernelGGL((k<W, MIS, STRIDE>), dim3(1), dim3(512), 0, 0, dp, iters);
  hipLaunchKernelGGL((k<W, MIS, STRIDE>), dim3(1), dim3(512), 0, 0, dp, iters);
  CK(hipDeviceSynchronize());
  long long p[8]; CK(hipMemcpy(p, dp, 64, hipMemcpyDeviceToHost));
  long long mx = 0; for (int w = 0; w < 8; ++w) mx = mx > p[w] ? mx : p[w];
  printf("%-50s %7.2f cycles per wave-instruction\n", name, (double)mx / (8.0 * iters * 8));
  CK(hipFree(dp));
}
int main() {
  run<16, 0, 16>("ds_read_b128 lane*16");
  run<16, 4, 16>("ds_read_b128 lane*16 + 4");
  run<16, 8, 16>("ds_read_b128 lane*16 + 8");
  run<16, 1, 16>("ds_read_b128 lane*16 + 1");
  run<12, 0, 16>("ds_read_b96 lane*16");
  run<12, 4, 16>("ds_read_b96 lane*16 + 4");
  run<12, 1, 16>("ds_read_b96 lane*16 + 1");
  run<8, 0, 8>("ds_read_b64 lane*8");
  run<8, 4, 8>("ds_read_b64 lane*8 + 4");
  run<8, 1, 8>("ds_read_b64 lane*8 + 1");
  run<8, 0, 16>("ds_read_b64 lane*16");
  run<8, 4, 16>("ds_read_b64 lane*16 + 4");
  run<8, 1, 16>("ds_read_b64 lane*16 + 1");
  run<4, 0, 4>("ds_read_b32 lane*4");
  run<4, 1, 4>("ds_read_b32 lane*4 + 1");
  run<4, 2, 4>("ds_read_b32 lane*4 + 2");
  run<4, 0, 16>("ds_read_b32 lane*16");
  run<4, 1, 16>("ds_read_b32 lane*16 + 1");
  run<4, 3, 16>("ds_read_b32 lane*16 + 3");
  run<116, 0, 16>("ds_write_b128 lane*16");
  run<116, 4, 16>("ds_write_b128 lane*16 + 4");
  run<116, 5, 16>("ds_write_b128 lane*16 + 5");
  run<104, 0, 4>("ds_write_b32 lane*4");
  run<104, 1, 4>("ds_write_b32 lane*4 + 1");
  run<99, 0, 8>("ds_read_b64_tr_b8 lane*8");
  run<99, 0, 16>("ds_read_b64_tr_b8 lane*16");
  return 0;
}
