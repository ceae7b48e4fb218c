// Do the two waves of a SIMD overlap matrix-pipe and VALU work?  One work-group of 512 threads on one CU: waves 0-3 take
// role A, waves 4-7 role B (wave i runs on SIMD i & 3, so wave i and i + 4 are partners).  Every wave reports its own
// s_memtime ticks for LOOPS trips of 8 instructions.  Build: hipcc --offload-arch=gfx950 -O3 -o coissue coissue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define LOOPS 512

enum { IDLE = 0, MFMA4, MFMA32, F64RQ, I32, F32, LDSRD, PHASED, PHASED_AL };

template <int ROLE>
__device__ __forceinline__ int work(int seed) {
  int r = 0;
  if constexpr (ROLE == MFMA4) {
    v4i a0 = {seed, 1, 2, 3}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    int x = seed * 3, w = seed * 5;
    for (int it = 0; it < LOOPS; ++it) {
      a0 = __builtin_amdgcn_mfma_i32_4x4x4i8(w, x, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_i32_4x4x4i8(w, x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_i32_4x4x4i8(w, x, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_i32_4x4x4i8(w, x, a3, 0, 0, 0);
      a4 = __builtin_amdgcn_mfma_i32_4x4x4i8(w, x, a4, 0, 0, 0); a5 = __builtin_amdgcn_mfma_i32_4x4x4i8(w, x, a5, 0, 0, 0);
      a6 = __builtin_amdgcn_mfma_i32_4x4x4i8(w, x, a6, 0, 0, 0); a7 = __builtin_amdgcn_mfma_i32_4x4x4i8(w, x, a7, 0, 0, 0);
    }
    r = a0[0] + a1[1] + a2[2] + a3[3] + a4[0] + a5[1] + a6[2] + a7[3];
  } else if constexpr (ROLE == MFMA32) {
    v16i c0 = {}, c1 = {};
    v4i a = {seed, 1, 2, 3}, b = {3, 2, 1, seed};
    for (int it = 0; it < LOOPS; ++it) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
      }
    }
    r = c0[0] + c1[5];
  } else if constexpr (ROLE == F64RQ) {                     // the requantisation: cvt + fma + med3, 8 per trip = 24 VALU
    int z0 = seed, z1 = seed + 1, z2 = seed + 2, z3 = seed + 3, z4 = seed + 4, z5 = seed + 5, z6 = seed + 6, z7 = seed + 7;
    const double M = 1e-3 + seed * 1e-9, MG = 6755399441055744.0;
    for (int it = 0; it < LOOPS; ++it) {
#define RQ(z) { double t = __builtin_fma((double)z, M, MG); int q = __double2loint(t); asm volatile("v_med3_i32 %0, %1, %2, %3" : "=v"(z) : "v"(q), "v"(-127), "v"(seed | 0x7fff)); }
      RQ(z0) RQ(z1) RQ(z2) RQ(z3) RQ(z4) RQ(z5) RQ(z6) RQ(z7)
    }
    r = z0 + z1 + z2 + z3 + z4 + z5 + z6 + z7;
  } else if constexpr (ROLE == I32) {                       // 24 integer VALU per trip
    int z0 = seed, z1 = seed + 1, z2 = seed + 2, z3 = seed + 3, z4 = seed + 4, z5 = seed + 5, z6 = seed + 6, z7 = seed + 7;
    for (int it = 0; it < LOOPS; ++it) {
#define I3(z) asm volatile("v_add_u32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_med3_i32 %0, %0, %1, %2" : "+v"(z) : "v"(seed), "v"(it));
      I3(z0) I3(z1) I3(z2) I3(z3) I3(z4) I3(z5) I3(z6) I3(z7)
    }
    r = z0 + z1 + z2 + z3 + z4 + z5 + z6 + z7;
  } else if constexpr (ROLE == F32) {                       // 24 float32 VALU per trip
    float z0 = seed, z1 = seed + 1, z2 = seed + 2, z3 = seed + 3, z4 = seed + 4, z5 = seed + 5, z6 = seed + 6, z7 = seed + 7;
    const float m = 1.0001f;
    for (int it = 0; it < LOOPS; ++it) {
#define F3(z) asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_rndne_f32 %0, %0" : "+v"(z) : "v"(m));
      F3(z0) F3(z1) F3(z2) F3(z3) F3(z4) F3(z5) F3(z6) F3(z7)
    }
    r = (int)(z0 + z1 + z2 + z3 + z4 + z5 + z6 + z7);
  } else if constexpr (ROLE == LDSRD) {                     // 8 ds_read_b128 per trip
    extern __shared__ v4i sm[];
    v4i s = {};
    for (int it = 0; it < LOOPS; ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        v4i v = sm[(threadIdx.x + 64 * k + it) & 1023];
        s += v;
      }
    }
    r = s[0] + s[1] + s[2] + s[3];
  }
  else if constexpr (ROLE == PHASED || ROLE == PHASED_AL) {  // a depthwise group: 168 MFMAs (8 chains x 21 steps), then ~250 VALU
    v4i a0 = {seed, 1, 2, 3}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    int x = seed * 3, w = seed * 5, w2 = seed * 7;
    int z0 = seed, z1 = seed + 1, z2 = seed + 2, z3 = seed + 3, z4 = seed + 4, z5 = seed + 5, z6 = seed + 6, z7 = seed + 7;
    const double M = 1e-3 + seed * 1e-9, MG = 6755399441055744.0;
    for (int it = 0; it < LOOPS / 16; ++it) {
#pragma unroll
      for (int st = 0; st < 21; ++st) {
        if constexpr (ROLE == PHASED_AL) w = __builtin_amdgcn_alignbyte(w2, w, st & 3);
        a0 = __builtin_amdgcn_mfma_i32_4x4x4i8(w, x, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_i32_4x4x4i8(w, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_i32_4x4x4i8(w, x, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_i32_4x4x4i8(w, x, a3, 0, 0, 0);
        a4 = __builtin_amdgcn_mfma_i32_4x4x4i8(w, x, a4, 0, 0, 0); a5 = __builtin_amdgcn_mfma_i32_4x4x4i8(w, x, a5, 0, 0, 0);
        a6 = __builtin_amdgcn_mfma_i32_4x4x4i8(w, x, a6, 0, 0, 0); a7 = __builtin_amdgcn_mfma_i32_4x4x4i8(w, x, a7, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 10; ++k) {                         // 10 x 24 VALU
        RQ(z0) RQ(z1) RQ(z2) RQ(z3) RQ(z4) RQ(z5) RQ(z6) RQ(z7)
      }
      __builtin_amdgcn_sched_barrier(0);
      x += z0;
    }
    r = a0[0] + a1[1] + a2[2] + a3[3] + a4[0] + a5[1] + a6[2] + a7[3] + z0 + z1 + z2 + z3 + z4 + z5 + z6 + z7;
  }
  return r;
}

template <int RA, int RB>
__global__ void __launch_bounds__(512, 2) k_pair(long long* out, int* sink, int seed) {
  extern __shared__ v4i sm[];
  for (int i = threadIdx.x; i < 1024; i += 512) sm[i] = (v4i){i, seed, 2, 3};
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  const long long t0 = __builtin_amdgcn_s_memtime();
  int r = wave < 4 ? work<RA>(seed + threadIdx.x) : work<RB>(seed + threadIdx.x);
  const long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
  sink[threadIdx.x] = r;
}

template <int RA, int RB>
static void run(const char* name, long long* dout, int* dsink) {
  long long h[8];
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k_pair<RA, RB>), dim3(1), dim3(512), 16384, 0, dout, dsink, 7);
    hipDeviceSynchronize();
  }
  hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-28s A (wave 0) %8lld ticks  B (wave 4) %8lld ticks   per trip A %.1f B %.1f\n", name, h[0], h[4], h[0] / (double)LOOPS, h[4] / (double)LOOPS);
}

int main() {
  long long* dout; int* dsink;
  hipMalloc(&dout, 64); hipMalloc(&dsink, 4096);
  run<MFMA4, IDLE>("mfma4x4x4 | idle", dout, dsink);
  run<IDLE, F64RQ>("idle | f64 requant", dout, dsink);
  run<IDLE, I32>("idle | int32 valu", dout, dsink);
  run<IDLE, F32>("idle | f32 valu", dout, dsink);
  run<IDLE, LDSRD>("idle | lds b128", dout, dsink);
  run<MFMA4, MFMA4>("mfma4x4x4 | mfma4x4x4", dout, dsink);
  run<F64RQ, F64RQ>("f64 requant | f64 requant", dout, dsink);
  run<MFMA4, F64RQ>("mfma4x4x4 | f64 requant", dout, dsink);
  run<MFMA4, I32>("mfma4x4x4 | int32 valu", dout, dsink);
  run<MFMA4, F32>("mfma4x4x4 | f32 valu", dout, dsink);
  run<MFMA4, LDSRD>("mfma4x4x4 | lds b128", dout, dsink);
  run<PHASED, IDLE>("phased (32 groups) | idle", dout, dsink);
  run<PHASED, PHASED>("phased | phased", dout, dsink);
  run<PHASED_AL, IDLE>("phased+alignbyte | idle", dout, dsink);
  run<PHASED_AL, PHASED_AL>("phased+align | phased+align", dout, dsink);
  run<MFMA32, IDLE>("mfma32x32x32 | idle", dout, dsink);
  run<MFMA32, F64RQ>("mfma32x32x32 | f64 requant", dout, dsink);
  run<MFMA32, I32>("mfma32x32x32 | int32 valu", dout, dsink);
  run<MFMA32, MFMA32>("mfma32x32x32 | mfma32x32x32", dout, dsink);
  return 0;
}
