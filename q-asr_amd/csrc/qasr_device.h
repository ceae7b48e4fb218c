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
// Batches of values sharing a multiplier (k_sep / k_requant epilogues).  Round 1 formed the product in float32
// first and took the fp64 path only for a batch holding a value within tau of a rounding tie (54 cycles per value with
// the wave vote); the round-2 micro-benchmarks (profiles/microbench/ubench.hip) put fp64 fma at the integer ALU's issue
// rate on gfx950, which makes the exact form the cheap one (~20 cycles per value) - so that is all there is now.
template <int N>
__device__ __forceinline__ void requant_batch(int (&q)[N], const int (&z)[N], double M, int lo, int hi) {
#pragma unroll
  for (int i = 0; i < N; ++i) q[i] = requant_clamp(z[i], M, lo, hi);
}

// same with one multiplier per group of 4 consecutive values (4 channels x 4 values in the whole-utterance kernel)
template <int N>
__device__ __forceinline__ void requant_batch4(int (&q)[N], const int (&z)[N], const double (&M)[N / 4], int lo, int hi) {
#pragma unroll
  for (int i = 0; i < N; ++i) q[i] = requant_clamp(z[i], M[i / 4], lo, hi);
}

// rint(z * M) for |z * M| < 2^31 (the packer flags ops that cannot promise this QASR_F_WIDE_RQ: k_sep2 leaves them
// to k_sep, which then clamps in the double domain): one fp64 fma rounds
// half-to-even into the low mantissa word (quant_utils.py:196-198: round(f64(z) * f64(m) / 2^e), M = m * 2^-e)
__device__ __forceinline__ int rq_rint(int z, double M) { return __double2loint(__builtin_fma((double)z, M, MAGIC_RNE)); }
// clamp(x, lo, hi) for lo <= hi as ONE v_med3_i32 (the compiler keeps min/max apart: it cannot know lo <= hi)
__device__ __forceinline__ int med3i(int x, int lo, int hi) {
  int r;
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "v"(hi));
  return r;
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
