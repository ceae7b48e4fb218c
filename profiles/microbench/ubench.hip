// Issue-rate / bandwidth micro-benchmarks behind DESIGN.md §5 (gfx950).  Build: hipcc --offload-arch=gfx950 -O3 -o ubench ubench.hip
// Every figure is s_memtime ticks (= shader cycles) per instruction per wave, with W waves per SIMD resident
// (one work-group of 256*W threads on one CU) - or, for the bandwidth rows, bytes per cycle per CU with every CU busy.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define REP8(x) x x x x x x x x
#define LOOPS 512

// ---- VALU issue-rate kernels: 8 independent chains of one instruction per loop trip --------------------------------
__global__ void k_cvt_f64_i32(long long* out, int* sink, int seed) {
  int s = seed + threadIdx.x; double a0, a1, a2, a3, a4, a5, a6, a7;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < LOOPS; ++it) {
    asm volatile("v_cvt_f64_i32 %0, %8\n v_cvt_f64_i32 %1, %8\n v_cvt_f64_i32 %2, %8\n v_cvt_f64_i32 %3, %8\n v_cvt_f64_i32 %4, %8\n v_cvt_f64_i32 %5, %8\n v_cvt_f64_i32 %6, %8\n v_cvt_f64_i32 %7, %8" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(s));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t0; out[2 * (threadIdx.x >> 6) + 1] = t1; }
  sink[threadIdx.x] = (int)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
__global__ void k_fma_f64(long long* out, int* sink, int seed) {
  double m = seed * 1e-3 + threadIdx.x; double b = 1.5; double a0 = m + 0, a1 = m + 1, a2 = m + 2, a3 = m + 3, a4 = m + 4, a5 = m + 5, a6 = m + 6, a7 = m + 7;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < LOOPS; ++it) {
    asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(b));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t0; out[2 * (threadIdx.x >> 6) + 1] = t1; }
  sink[threadIdx.x] = (int)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
__global__ void k_max_f64(long long* out, int* sink, int seed) {
  double m = seed * 1e-3 + threadIdx.x; double a0 = m + 0, a1 = m + 1, a2 = m + 2, a3 = m + 3, a4 = m + 4, a5 = m + 5, a6 = m + 6, a7 = m + 7;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < LOOPS; ++it) {
    asm volatile("v_max_f64 %0, %0, %8\n v_max_f64 %1, %1, %8\n v_max_f64 %2, %2, %8\n v_max_f64 %3, %3, %8\n v_max_f64 %4, %4, %8\n v_max_f64 %5, %5, %8\n v_max_f64 %6, %6, %8\n v_max_f64 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t0; out[2 * (threadIdx.x >> 6) + 1] = t1; }
  sink[threadIdx.x] = (int)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
__global__ void k_add_f64(long long* out, int* sink, int seed) {
  double m = seed * 1e-3 + threadIdx.x; double a0 = m + 0, a1 = m + 1, a2 = m + 2, a3 = m + 3, a4 = m + 4, a5 = m + 5, a6 = m + 6, a7 = m + 7;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < LOOPS; ++it) {
    asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t0; out[2 * (threadIdx.x >> 6) + 1] = t1; }
  sink[threadIdx.x] = (int)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
__global__ void k_mul_f32(long long* out, int* sink, int seed) {
  float m = seed * 1e-3f + threadIdx.x; float a0 = m + 0, a1 = m + 1, a2 = m + 2, a3 = m + 3, a4 = m + 4, a5 = m + 5, a6 = m + 6, a7 = m + 7;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < LOOPS; ++it) {
    asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t0; out[2 * (threadIdx.x >> 6) + 1] = t1; }
  sink[threadIdx.x] = (int)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
__global__ void k_rndne_f32(long long* out, int* sink, int seed) {
  float m = seed * 1e-3f + threadIdx.x; float a0 = m + 0, a1 = m + 1, a2 = m + 2, a3 = m + 3, a4 = m + 4, a5 = m + 5, a6 = m + 6, a7 = m + 7;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < LOOPS; ++it) {
    asm volatile("v_rndne_f32 %0, %0\n v_rndne_f32 %1, %1\n v_rndne_f32 %2, %2\n v_rndne_f32 %3, %3\n v_rndne_f32 %4, %4\n v_rndne_f32 %5, %5\n v_rndne_f32 %6, %6\n v_rndne_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t0; out[2 * (threadIdx.x >> 6) + 1] = t1; }
  sink[threadIdx.x] = (int)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
__global__ void k_cvt_f32_i32(long long* out, int* sink, int seed) {
  int m = seed + threadIdx.x; int a0 = m + 0, a1 = m + 1, a2 = m + 2, a3 = m + 3, a4 = m + 4, a5 = m + 5, a6 = m + 6, a7 = m + 7;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < LOOPS; ++it) {
    asm volatile("v_cvt_f32_i32 %0, %0\n v_cvt_f32_i32 %1, %1\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3\n v_cvt_f32_i32 %4, %4\n v_cvt_f32_i32 %5, %5\n v_cvt_f32_i32 %6, %6\n v_cvt_f32_i32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t0; out[2 * (threadIdx.x >> 6) + 1] = t1; }
  sink[threadIdx.x] = (int)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
__global__ void k_med3_i32(long long* out, int* sink, int seed) {
  int m = seed + threadIdx.x; int n = seed * 3; int a0 = m + 0, a1 = m + 1, a2 = m + 2, a3 = m + 3, a4 = m + 4, a5 = m + 5, a6 = m + 6, a7 = m + 7;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < LOOPS; ++it) {
    asm volatile("v_med3_i32 %0, %0, %8, %9\n v_med3_i32 %1, %1, %8, %9\n v_med3_i32 %2, %2, %8, %9\n v_med3_i32 %3, %3, %8, %9\n v_med3_i32 %4, %4, %8, %9\n v_med3_i32 %5, %5, %8, %9\n v_med3_i32 %6, %6, %8, %9\n v_med3_i32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(n));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t0; out[2 * (threadIdx.x >> 6) + 1] = t1; }
  sink[threadIdx.x] = (int)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
__global__ void k_perm_b32(long long* out, int* sink, int seed) {
  int m = seed + threadIdx.x; int n = seed * 3; int a0 = m + 0, a1 = m + 1, a2 = m + 2, a3 = m + 3, a4 = m + 4, a5 = m + 5, a6 = m + 6, a7 = m + 7;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < LOOPS; ++it) {
    asm volatile("v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(n));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t0; out[2 * (threadIdx.x >> 6) + 1] = t1; }
  sink[threadIdx.x] = (int)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
__global__ void k_alignbyte(long long* out, int* sink, int seed) {
  int m = seed + threadIdx.x; int n = seed * 3; int a0 = m + 0, a1 = m + 1, a2 = m + 2, a3 = m + 3, a4 = m + 4, a5 = m + 5, a6 = m + 6, a7 = m + 7;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < LOOPS; ++it) {
    asm volatile("v_alignbyte_b32 %0, %0, %8, %9\n v_alignbyte_b32 %1, %1, %8, %9\n v_alignbyte_b32 %2, %2, %8, %9\n v_alignbyte_b32 %3, %3, %8, %9\n v_alignbyte_b32 %4, %4, %8, %9\n v_alignbyte_b32 %5, %5, %8, %9\n v_alignbyte_b32 %6, %6, %8, %9\n v_alignbyte_b32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(n));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t0; out[2 * (threadIdx.x >> 6) + 1] = t1; }
  sink[threadIdx.x] = (int)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
__global__ void k_mad_i32_i24(long long* out, int* sink, int seed) {
  int m = seed + threadIdx.x; int n = seed * 3; int a0 = m + 0, a1 = m + 1, a2 = m + 2, a3 = m + 3, a4 = m + 4, a5 = m + 5, a6 = m + 6, a7 = m + 7;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < LOOPS; ++it) {
    asm volatile("v_mad_i32_i24 %0, %0, %8, %9\n v_mad_i32_i24 %1, %1, %8, %9\n v_mad_i32_i24 %2, %2, %8, %9\n v_mad_i32_i24 %3, %3, %8, %9\n v_mad_i32_i24 %4, %4, %8, %9\n v_mad_i32_i24 %5, %5, %8, %9\n v_mad_i32_i24 %6, %6, %8, %9\n v_mad_i32_i24 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(n));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t0; out[2 * (threadIdx.x >> 6) + 1] = t1; }
  sink[threadIdx.x] = (int)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}

// ---- requant sequences over 16 accumulators per lane -----------------------------------------------------------
// f64 form: lo32(fma(f64(z), M, 1.5*2^52)) then integer clamp
__global__ void k_rq_f64(long long* out, int* sink, int seed) {
  int z[16];
  for (int i = 0; i < 16; ++i) z[i] = seed * 977 + threadIdx.x * 13 + i * 1001;
  const double M = 1.0 / (3.0 + seed);
  int acc = 0;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < LOOPS; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      double t = __builtin_fma((double)z[i], M, 6755399441055744.0);
      int q = __double2loint(t);
      q = min(max(q, -128), 127);
      acc += q;
      z[i] += q + it;
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t0; out[2 * (threadIdx.x >> 6) + 1] = t1; }
  sink[threadIdx.x] = acc;
}
// f32 fast path with the ambiguity vote (round 1's production form)
__global__ void k_rq_f32(long long* out, int* sink, int seed) {
  int z[16];
  for (int i = 0; i < 16; ++i) z[i] = seed * 977 + threadIdx.x * 13 + i * 1001;
  const double M = 1.0 / (3.0 + seed);
  const float Mf = (float)M, tau = 129.f * 1.5e-7f;
  int acc = 0;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < LOOPS; ++it) {
    float r[16];
    bool amb = false;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float p = (float)z[i] * Mf;
      r[i] = rintf(p);
      amb |= (0.5f - fabsf(p - r[i]) <= tau);
    }
    if (__any(amb)) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        double t = __builtin_fma((double)z[i], M, 6755399441055744.0);
        t = fmin(fmax(t, 6755399441055744.0 - 128.0), 6755399441055744.0 + 127.0);
        int q = __double2loint(t);
        acc += q;
        z[i] += q + it;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        int q = (int)__builtin_amdgcn_fmed3f(r[i], -128.f, 127.f);
        acc += q;
        z[i] += q + it;
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t0; out[2 * (threadIdx.x >> 6) + 1] = t1; }
  sink[threadIdx.x] = acc;
}

// ---- MFMA issue rates --------------------------------------------------------------------------------------------
__global__ void k_mfma444(long long* out, int* sink, int seed) {
  v4i c[8];
  for (int i = 0; i < 8; ++i) c[i] = (v4i){0, 0, 0, 0};
  int a = seed + threadIdx.x, b = seed * 7 + threadIdx.x;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < LOOPS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_i32_4x4x4i8(a, b, c[i], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t0; out[2 * (threadIdx.x >> 6) + 1] = t1; }
  int s = 0;
  for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  sink[threadIdx.x] = s;
}
template <int NC>
__global__ void k_mfma444_chains(long long* out, int* sink, int seed) {
  v4i c[NC];
  for (int i = 0; i < NC; ++i) c[i] = (v4i){0, 0, 0, 0};
  int a = seed + threadIdx.x, b = seed * 7 + threadIdx.x;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < LOOPS; ++it) {
#pragma unroll
    for (int r = 0; r < 8 / NC; ++r)
#pragma unroll
      for (int i = 0; i < NC; ++i) c[i] = __builtin_amdgcn_mfma_i32_4x4x4i8(a, b, c[i], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t0; out[2 * (threadIdx.x >> 6) + 1] = t1; }
  int s = 0;
  for (int i = 0; i < NC; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  sink[threadIdx.x] = s;
}
__global__ void k_mfma32(long long* out, int* sink, int seed) {
  v16i c[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) c[i][r] = 0;
  v4i a = {seed, seed + 1, (int)threadIdx.x, 3}, b = {seed * 3, 5, (int)threadIdx.x * 7, 1};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < LOOPS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c[i & 3], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t0; out[2 * (threadIdx.x >> 6) + 1] = t1; }
  int s = 0;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += c[i][r];
  sink[threadIdx.x] = s;
}
__global__ void k_mfma16(long long* out, int* sink, int seed) {
  v4i c[8];
  for (int i = 0; i < 8; ++i) c[i] = (v4i){0, 0, 0, 0};
  v4i a = {seed, seed + 1, (int)threadIdx.x, 3}, b = {seed * 3, 5, (int)threadIdx.x * 7, 1};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < LOOPS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[i], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t0; out[2 * (threadIdx.x >> 6) + 1] = t1; }
  int s = 0;
  for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  sink[threadIdx.x] = s;
}

// ---- do VALU instructions overlap a wave's own (and its SIMD partner's) MFMAs?  per loop trip: 8 x 32x32x32 MFMA
// (4 chains) and / or 16 fp64 requants (MODE 1: MFMA only, 2: requant only, 3: both, one MFMA then two requants;
// 4: 6 x v_mul_f32 per slot, no MFMA)
template <int MODE>
__global__ void k_coissue(long long* out, int* sink, int seed) {
  v16i c[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) c[i][r] = 0;
  v4i a = {seed, seed + 1, (int)threadIdx.x, 3}, b = {seed * 3, 5, (int)threadIdx.x * 7, 1};
  int z[16];
  float f[6];
  for (int i = 0; i < 16; ++i) z[i] = seed * 977 + threadIdx.x * 13 + i * 1001;
  for (int i = 0; i < 6; ++i) f[i] = 1.0f + 0.001f * (threadIdx.x + i);
  const double M = 1.0 / (3.0 + seed);
  int acc = 0;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < LOOPS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE & 1) c[i & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c[i & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE == 2 || MODE == 3) {
#pragma unroll
        for (int j = 2 * i; j < 2 * i + 2; ++j) {
          double t = __builtin_fma((double)z[j], M, 6755399441055744.0);
          int q = __double2loint(t);
          q = min(max(q, -128), 127);
          acc += q;
          z[j] += q + it;
        }
      }
      if (MODE == 4) {
#pragma unroll
        for (int j = 0; j < 6; ++j) f[j] = f[j] * 1.0001f;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t0; out[2 * (threadIdx.x >> 6) + 1] = t1; }
  int s = acc;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += c[i][r];
  for (int i = 0; i < 6; ++i) s += (int)f[i];
  sink[threadIdx.x] = s;
}

// ---- LDS store forms: 16 bytes per lane as 16 x ds_write_b8, 4 x b32, 1 x b128 ----------------------------------
__global__ void k_lds_b8(long long* out, int* sink, int seed) {
  __shared__ volatile unsigned char s[64 * 1024];
  const int base = (threadIdx.x * 4) & 0x3fff;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < LOOPS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) s[base + i * 528 + ((it & 3) << 14)] = (unsigned char)(seed + i + it);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t0; out[2 * (threadIdx.x >> 6) + 1] = t1; }
  sink[threadIdx.x] = s[threadIdx.x];
}
__global__ void k_lds_b32(long long* out, int* sink, int seed) {
  __shared__ volatile unsigned s32[16 * 1024];
  const int base = threadIdx.x & 0xfff;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < LOOPS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) s32[base + i * 132 + ((it & 3) << 12)] = seed + i + it;
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t0; out[2 * (threadIdx.x >> 6) + 1] = t1; }
  sink[threadIdx.x] = s32[threadIdx.x];
}

// ---- ds_read_b64_tr_b8 probe: which LDS bytes land in which lane/byte ------------------------------------------
__global__ void k_tr8(int* out, int stride) {
  __shared__ __attribute__((aligned(16))) unsigned char s[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) s[i] = 0;
  __syncthreads();
  // tag every byte of a [rows][stride] image with (row, col): row in the high nibble.. use 16-bit ids via two probes
  for (int i = threadIdx.x; i < 4096; i += 64) s[i] = (unsigned char)(i & 0xff);
  __syncthreads();
  const int l = threadIdx.x;
  // lane l supplies the address of 8 contiguous bytes: row = l >> 1 (32 rows), 8-byte half = l & 1
  const unsigned addr = (unsigned)(size_t)s + (l >> 1) * stride + 8 * (l & 1);
  v2i r;
  asm volatile("ds_read_b64_tr_b8 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
  out[2 * l] = r[0];
  out[2 * l + 1] = r[1];
}

// ---- L2 -> VGPR bandwidth per CU: every work-group streams the same `bytes` region (16 B per lane) ---------------
__global__ void __launch_bounds__(512) k_l2bw(const v4i* __restrict__ w, int n16_per_thread, long long* out, int* sink) {
  v4i acc = {0, 0, 0, 0};
  const v4i* p = w + threadIdx.x;
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 8
  for (int i = 0; i < n16_per_thread; ++i) {
    v4i v = p[(size_t)i * 512];
    acc[0] ^= v[0]; acc[1] ^= v[1]; acc[2] ^= v[2]; acc[3] ^= v[3];
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (acc[0] == 0x12345678) sink[0] = acc[1] + acc[2] + acc[3];
}

// every work-group streams ITS OWN `bytes` region (16 B per lane), as the depthwise window loads do
__global__ void __launch_bounds__(512) k_l2bw_own(const v4i* __restrict__ w, int n16_per_thread, long long* out, int* sink) {
  v4i acc = {0, 0, 0, 0};
  const v4i* p = w + (size_t)blockIdx.x * n16_per_thread * 512 + threadIdx.x;
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 4
  for (int i = 0; i < n16_per_thread; ++i) {
    v4i v = p[(size_t)i * 512];
    acc[0] ^= v[0]; acc[1] ^= v[1]; acc[2] ^= v[2]; acc[3] ^= v[3];
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (acc[0] == 0x12345678) sink[0] = acc[1] + acc[2] + acc[3];
}

template <class K>
static double run1(K kern, int waves_per_simd, long long* d_out, int* d_sink) {
  long long h[64];
  hipLaunchKernelGGL(kern, dim3(1), dim3(256 * waves_per_simd), 0, 0, d_out, d_sink, 3);
  hipLaunchKernelGGL(kern, dim3(1), dim3(256 * waves_per_simd), 0, 0, d_out, d_sink, 3);
  hipMemcpy(h, d_out, 16 * 4 * waves_per_simd, hipMemcpyDeviceToHost);
  long long lo = h[0], hi = h[1];
  for (int w = 0; w < 4 * waves_per_simd; ++w) { lo = h[2 * w] < lo ? h[2 * w] : lo; hi = h[2 * w + 1] > hi ? h[2 * w + 1] : hi; }
  return (double)(hi - lo);                                  // first start -> last end over all waves of the work-group
}

int main() {
  long long* d_out;
  int* d_sink;
  hipMalloc(&d_out, 8 * 4096);
  hipMalloc(&d_sink, 4 * 4096);
#define ROW(name, kern, per_loop)                                                                        \
  do {                                                                                                   \
    printf("%-28s", name);                                                                               \
    for (int w = 1; w <= 4; w *= 2) printf("  W=%d: %7.2f", w, run1(kern, w, d_out, d_sink) / (LOOPS * (per_loop))); \
    printf("   ticks per instruction per wave (first start -> last end of the W waves sharing a SIMD)\n");                                                        \
  } while (0)
  ROW("v_cvt_f64_i32", k_cvt_f64_i32, 8);
  ROW("v_fma_f64", k_fma_f64, 8);
  ROW("v_max_f64", k_max_f64, 8);
  ROW("v_add_f64", k_add_f64, 8);
  ROW("v_mul_f32", k_mul_f32, 8);
  ROW("v_rndne_f32", k_rndne_f32, 8);
  ROW("v_cvt_f32_i32", k_cvt_f32_i32, 8);
  ROW("v_med3_i32", k_med3_i32, 8);
  ROW("v_perm_b32", k_perm_b32, 8);
  ROW("v_alignbyte_b32", k_alignbyte, 8);
  ROW("v_mad_i32_i24", k_mad_i32_i24, 8);
  ROW("requant f64 (per value)", k_rq_f64, 16);
  ROW("requant f32+vote (per value)", k_rq_f32, 16);
  ROW("v_mfma_i32_4x4x4_16b_i8", k_mfma444, 8);
  ROW("  4x4x4, 1 dependent chain", k_mfma444_chains<1>, 8);
  ROW("  4x4x4, 2 chains", k_mfma444_chains<2>, 8);
  ROW("  4x4x4, 4 chains", k_mfma444_chains<4>, 8);
  ROW("v_mfma_i32_32x32x32_i8", k_mfma32, 8);
  ROW("v_mfma_i32_16x16x64_i8", k_mfma16, 8);
  ROW("loop trip: 8 MFMA 32x32x32", k_coissue<1>, 1);
  ROW("loop trip: 16 fp64 requants", k_coissue<2>, 1);
  ROW("loop trip: both interleaved", k_coissue<3>, 1);
  ROW("loop trip: 48 v_mul_f32", k_coissue<4>, 1);
  ROW("ds_write_b8 (scatter)", k_lds_b8, 8);
  ROW("ds_write_b32", k_lds_b32, 8);

  // transposed 8-bit LDS read: print the source byte offsets each lane receives
  {
    int* d;
    hipMalloc(&d, 512);
    int h[128];
    for (int stride = 16; stride <= 64; stride *= 2) {
      hipLaunchKernelGGL(k_tr8, dim3(1), dim3(64), 0, 0, d, stride);
      hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
      printf("ds_read_b64_tr_b8, lane l supplies &img[(l>>1)*%d + 8*(l&1)]; received byte offsets (mod 256):\n", stride);
      for (int l = 0; l < 64; ++l) {
        if (l < 20 || l == 31 || l == 32 || l == 33 || l == 63) {
          printf("  lane %2d:", l);
          for (int j = 0; j < 8; ++j) printf(" %3d", (h[2 * l + (j >> 2)] >> (8 * (j & 3))) & 0xff);
          printf("\n");
        }
      }
    }
  }
  // L2 -> register bandwidth
  {
    const size_t bytes = 256 * 1024;
    v4i* w;
    hipMalloc(&w, bytes);
    hipMemset(w, 1, bytes);
    std::vector<long long> h(2048);
    for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu) {
      const int grid = 256 * wg_per_cu, n = (int)(bytes / (512 * 16));
      for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_l2bw, dim3(grid), dim3(512), 0, 0, w, n, d_out, d_sink);
      hipMemcpy(h.data(), d_out, 8 * grid, hipMemcpyDeviceToHost);
      double s = 0, mx = 0;
      for (int i = 0; i < grid; ++i) { s += h[i]; mx = h[i] > mx ? h[i] : mx; }
      printf("L2->VGPR, %d x 512-thread WGs each streaming the same 256 KiB (16 B/lane): mean %.0f ticks, max %.0f  -> %.1f B/clk per WG (mean)\n",
             grid, s / grid, mx, bytes / (s / grid));
    }
  }
  {
    const size_t per_wg = 64 * 1024;
    v4i* w;
    hipMalloc(&w, per_wg * 256);
    hipMemset(w, 1, per_wg * 256);
    std::vector<long long> h(256);
    for (int rep = 0; rep < 4; ++rep) hipLaunchKernelGGL(k_l2bw_own, dim3(256), dim3(512), 0, 0, w, (int)(per_wg / (512 * 16)), d_out, d_sink);
    hipMemcpy(h.data(), d_out, 8 * 256, hipMemcpyDeviceToHost);
    double s = 0, mx = 0;
    for (int i = 0; i < 256; ++i) { s += h[i]; mx = h[i] > mx ? h[i] : mx; }
    printf("L2->VGPR, 256 x 512-thread WGs each streaming ITS OWN 64 KiB (16 MiB in all, 4th back-to-back launch): mean %.0f ticks, max %.0f -> %.1f B/clk per WG\n",
           s / 256, mx, per_wg / (s / 256));
  }
  return 0;
}
