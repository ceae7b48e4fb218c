// Device-side helpers shared by the kernel translation units (requantisation arithmetic, packing).
#pragma once
#include "qasr_internal.h"

namespace qasr {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define MAGIC_RNE 6755399441055744.0 /* 1.5 * 2^52: fma(z, M, MAGIC) rounds z*M half-to-even into the low word */

// clamp(rint(z * M), lo, hi): fixedpoint_mul.forward (quant_utils.py:196-198,213).  z*M is exact in fp64
// for |z| < 2^22 (m < 2^31), otherwise it is the same single fp64 rounding the reference performs.
// The clamp is applied in the double domain (MAGIC+lo, MAGIC+hi) so huge products cannot wrap the low word.
__device__ __forceinline__ int requant_clamp(int z, double M, int lo, int hi) {
  double t = __builtin_fma((double)z, M, MAGIC_RNE);
  t = fmin(fmax(t, MAGIC_RNE + (double)lo), MAGIC_RNE + (double)hi);
  return __double2loint(t);
}
// Same result, cheaper on average: fp64 VALU ops issue at ~1/4 rate on gfx950, so the product is first formed in
// float32 and the fp64 path above only runs when that cannot decide the rounding.
//   p = fl32(z) * fl32(M):  |p - z*M| <= |z*M| * (2^-23 + 2^-48)  (|z| < 2^24 exact in f32; two roundings of 2^-24 each;
//   a batch holding |z| >= 2^24, where the conversion itself rounds, takes the fp64 path).
//   Inside the target range (|p| <= R = max(|lo|,|hi|) + 1) the error is < R * 1.2e-7, hence r = rint(p) is the exact
//   answer unless p lies within tau = R * 1.5e-7 of a half-integer (p - r is exact in f32); outside the range both the
//   exact and the approximate product round beyond lo / hi and the clamp (applied to r, in float) gives the same result.
//   Batches holding an ambiguous value (~4 % of the 1024-value wave batches for 8-bit ranges) take the fp64 path.
// The wave votes once for a whole batch of values, so the fallback costs one uniform branch.
__device__ __forceinline__ float requant_tau(float lo, float hi) { return (fmaxf(fabsf(lo), fabsf(hi)) + 1.0f) * 1.5e-7f; }
__device__ __forceinline__ bool requant_ambiguous(float p, float r, float tau) { return 0.5f - fabsf(p - r) <= tau; }

// requantise N accumulators of one lane (same multiplier): fast float32 path with exact fp64 fallback
template <int N>
__device__ __forceinline__ void requant_batch(int (&q)[N], const int (&z)[N], double M, int lo, int hi) {
  const float Mf = (float)M, flo = (float)lo, fhi = (float)hi, tau = requant_tau(flo, fhi);
  bool amb = false;
  float r[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const float p = __fmul_rn((float)z[i], Mf);
    r[i] = rintf(p);
    amb |= requant_ambiguous(p, r[i], tau) | (__builtin_abs(z[i]) >= (1 << 24));   // (float)z inexact: a third rounding
  }
  if (__builtin_expect(__any(amb), 0)) {
#pragma unroll
    for (int i = 0; i < N; ++i) q[i] = requant_clamp(z[i], M, lo, hi);
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) q[i] = (int)__builtin_amdgcn_fmed3f(r[i], flo, fhi);
  }
}

// same with one multiplier per group of 4 consecutive values (4 channels x 4 values in the whole-utterance kernel)
template <int N>
__device__ __forceinline__ void requant_batch4(int (&q)[N], const int (&z)[N], const double (&M)[N / 4], int lo, int hi) {
  const float flo = (float)lo, fhi = (float)hi, tau = requant_tau(flo, fhi);
  bool amb = false;
  float r[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const float p = __fmul_rn((float)z[i], (float)M[i / 4]);
    r[i] = rintf(p);
    amb |= requant_ambiguous(p, r[i], tau) | (__builtin_abs(z[i]) >= (1 << 24));
  }
  if (__builtin_expect(__any(amb), 0)) {
#pragma unroll
    for (int i = 0; i < N; ++i) q[i] = requant_clamp(z[i], M[i / 4], lo, hi);
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) q[i] = (int)__builtin_amdgcn_fmed3f(r[i], flo, fhi);
  }
}

// rint(z*M) as a double (RESADD sums two of these before clamping, quant_utils.py:211)
__device__ __forceinline__ double requant_d(int z, double M) { return rint((double)z * M); }

// z_int = round(x / pre_act_scaling_factor) of fixedpoint_mul (quant_utils.py:187) taken through the
// float32 view y = fl32(fl32(acc) * s_b); equals acc whenever |acc| < 2^22 (then the caller skips this).
// a * b rounded to float32 as an operation of its own: HIP's __fmul_rn is a plain operator that the backend contracts
// with a following add/sub into v_fma_f32 (pragmas notwithstanding), so the product is made opaque
__device__ __forceinline__ float mul_f32_unfused(float a, float b) {
  float r;
  asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ int z_roundtrip(int acc, float sb, bool relu) {
  float y = mul_f32_unfused((float)acc, sb);
  if (relu) y = fmaxf(y, 0.0f);
  return (int)rintf(__fdiv_rn(y, sb));
}

__device__ __forceinline__ unsigned pack4(int a, int b, int c, int d) {
  return (unsigned)(a & 0xff) | ((unsigned)(b & 0xff) << 8) | ((unsigned)(c & 0xff) << 16) | ((unsigned)d << 24);
}

// ------------------------------------------------------------------------------------------------ epilogue helpers
// Integer result of one conv accumulator for the non-RESADD case: ReLU'd z.
__device__ __forceinline__ int epi_z(int acc, const EpiP& e, float sb) {
  bool relu = e.flags & QASR_F_RELU;
  if (e.flags & QASR_F_EXACT_Z) return z_roundtrip(acc, sb, relu);
  return relu ? max(acc, 0) : acc;
}
__device__ __forceinline__ int out_value(int z, const OutP& o, double Mc) {
  if (o.mode == 2) return z;
  return requant_clamp(z, o.mode == 1 ? Mc : o.m, o.lo, o.hi);
}


// 1x1 conv weights are stored in MFMA B-fragment order (qasr/pack.py:fragment_order): the 16 bytes lane `lane`
// needs for output-channel tile `co >> 5` and K step `ks` sit at ((tile * nks + ks) * 64 + lane) * 16.
__device__ __forceinline__ const v4i* w_frag(const int8_t* __restrict__ w, int cin_pad, int co_row, int ks) {
  const int lane = threadIdx.x & 63;
  return (const v4i*)(w + ((size_t)((co_row >> 5) * (cin_pad >> 5) + ks) * 64 + lane) * 16);
}

// time index of accumulator register r for lane half h in the C/D layout of the 32x32 MFMA
__device__ __forceinline__ int mfma32_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

}  // namespace qasr
