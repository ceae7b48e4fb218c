// Percentile calibration on device: the two torch.quantile calls of QuantAct's percentile mode
// (nemo/quantization/utils/quant_modules.py:121-125: x_min = quantile(x, 1 - p/100), x_max = quantile(x, p/100)) as one
// exact radix select instead of two full sorts.
//   * keys: float bits mapped to ascending unsigned order; four 8-bit passes, each a histogram over the elements that
//     still match the selected prefix (pass 0 shares one histogram between all order statistics);
//   * up to four order statistics are selected together: floor/ceil rank of the lower and of the upper quantile;
//   * result = torch's linear interpolation: rank = fl32(q * (n-1)) and w = fl32(rank - floor(rank)) are separate tensor
//     operations in ATen (two roundings); lerp(a, b, w) = w < 0.5 ? fma(w, b - a, a) : fma(-(1 - w), b - a, b) is ONE
//     fused operation in ATen's CPU kernel (LerpKernel.cpp: vec::fmadd; checked against torch 2.10 on an AVX-512 host:
//     300 of 300 inputs where fused and unfused differ follow the fused form).
// HBM-bound: pass 0 reads every element once; later passes read them again but touch the histogram only for the
// (few) prefix matches.  NaNs are not supported (activations are finite).
#include <algorithm>

#include "qasr_device.h"

namespace qasr {

#define QS_NSEL 4
#define QS_NT 256

struct QSelState {                 // lives in the caller's workspace
  unsigned hist[QS_NSEL][256];
  unsigned prefix[QS_NSEL];        // key bits selected so far (high bits)
  unsigned long long rank[QS_NSEL];   // rank still to resolve inside the prefix
  float weight[2];                 // interpolation weights of the two quantiles
};

__device__ __forceinline__ unsigned qs_key(float v) {
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float qs_unkey(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ void __launch_bounds__(64) k_qs_init(QSelState* st, unsigned long long n, float q_lo, float q_hi) {
  const int t = threadIdx.x;
  for (int i = t; i < QS_NSEL * 256; i += 64) (&st->hist[0][0])[i] = 0;
  if (t < 2) {
    const float q = t ? q_hi : q_lo;
    const float rank = mul_f32_unfused(q, (float)(n - 1)); // torch: ranks = q * (size - 1) in the input dtype
    const float below = floorf(rank), above = ceilf(rank);
    st->rank[2 * t] = (unsigned long long)below;
    st->rank[2 * t + 1] = (unsigned long long)above;
    st->weight[t] = __fsub_rn(rank, below);
    st->prefix[2 * t] = st->prefix[2 * t + 1] = 0;
  }
}

// PASS 0: one shared histogram of the top byte (every selection has the empty prefix).  PASS p > 0: per selection,
// histogram of byte (3 - p) over the elements whose top p bytes equal the selection's prefix.
template <int PASS>
__global__ void __launch_bounds__(QS_NT) k_qs_hist(const float* __restrict__ x, unsigned long long n, QSelState* st) {
  __shared__ unsigned h[PASS == 0 ? 1 : QS_NSEL][256];
  const int tid = threadIdx.x;
  for (int i = tid; i < (PASS == 0 ? 1 : QS_NSEL) * 256; i += QS_NT) (&h[0][0])[i] = 0;
  unsigned pre[QS_NSEL];
#pragma unroll
  for (int s = 0; s < QS_NSEL; ++s) pre[s] = st->prefix[s];
  __syncthreads();
  constexpr int shift = 24 - 8 * PASS;
  auto visit = [&](float v) {
    const unsigned k = qs_key(v);
    if (PASS == 0) {
      atomicAdd(&h[0][k >> 24], 1u);
    } else {
      const unsigned top = k >> ((shift + 8) & 31);
#pragma unroll
      for (int s = 0; s < QS_NSEL; ++s)
        if (top == (pre[s] >> ((shift + 8) & 31))) atomicAdd(&h[s][(k >> shift) & 255u], 1u);
    }
  };
  const unsigned long long n4 = n / 4, stride = (unsigned long long)gridDim.x * QS_NT;
  const float4* x4 = (const float4*)x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * QS_NT + tid; i < n4; i += stride) {
    const float4 v = x4[i];
    visit(v.x); visit(v.y); visit(v.z); visit(v.w);
  }
  if (blockIdx.x == 0 && tid < (int)(n - 4 * n4)) visit(x[4 * n4 + tid]);
  __syncthreads();
  for (int i = tid; i < (PASS == 0 ? 1 : QS_NSEL) * 256; i += QS_NT) {
    const unsigned c = (&h[0][0])[i];
    if (c) {
      if (PASS == 0) {
#pragma unroll
        for (int s = 0; s < QS_NSEL; ++s) atomicAdd(&st->hist[s][i], c);
      } else {
        atomicAdd(&(&st->hist[0][0])[i], c);
      }
    }
  }
}

// one wave per selection: find the bin holding the wanted rank, extend the prefix, clear the histogram
template <int PASS>
__global__ void __launch_bounds__(64 * QS_NSEL) k_qs_scan(QSelState* st) {
  const int s = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int shift = 24 - 8 * PASS;
  unsigned c[4];
  unsigned long long sum = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    c[i] = st->hist[s][4 * lane + i];
    sum += c[i];
  }
  unsigned long long incl = sum;                            // inclusive prefix sum over lanes
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long up = __shfl_up(incl, o);
    if (lane >= o) incl += up;
  }
  const unsigned long long excl = incl - sum, want = st->rank[s];
  if (want >= excl && want < incl) {                        // exactly one lane
    unsigned long long before = excl;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (want < before + c[i]) {
        st->prefix[s] |= (unsigned)(4 * lane + i) << shift;
        st->rank[s] = want - before;
        break;
      }
      before += c[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) st->hist[s][4 * lane + i] = 0;
}

__global__ void k_qs_finish(const QSelState* st, float* out2) {
  const int t = threadIdx.x;
  if (t < 2) {
    const float a = qs_unkey(st->prefix[2 * t]), b = qs_unkey(st->prefix[2 * t + 1]), w = st->weight[t];
    const float diff = __fsub_rn(b, a);
    out2[t] = (w < 0.5f) ? __builtin_fmaf(w, diff, a) : __builtin_fmaf(-__fsub_rn(1.0f, w), diff, b);
  }
}

}  // namespace qasr

extern "C" {

size_t qasr_quantile_workspace_bytes(void) { return sizeof(qasr::QSelState); }

int qasr_quantile2(void* stream, const float* x, size_t n, float q_lo, float q_hi, float* out2, void* workspace,
                   size_t workspace_bytes) {
  using namespace qasr;
  if (!x || !out2 || !workspace || n == 0 || workspace_bytes < sizeof(QSelState) || !(q_lo >= 0.f && q_lo <= 1.f) ||
      !(q_hi >= 0.f && q_hi <= 1.f) || ((uintptr_t)x & 15) || ((uintptr_t)workspace & 7))
    return QASR_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  QSelState* st = (QSelState*)workspace;
  const unsigned long long nn = n;
  const int blocks = (int)std::min<unsigned long long>(2048, (nn / 4 + QS_NT - 1) / QS_NT + 1);
  hipLaunchKernelGGL(k_qs_init, dim3(1), dim3(64), 0, s, st, nn, q_lo, q_hi);
  hipLaunchKernelGGL(k_qs_hist<0>, dim3(blocks), dim3(QS_NT), 0, s, x, nn, st);
  hipLaunchKernelGGL(k_qs_scan<0>, dim3(1), dim3(64 * QS_NSEL), 0, s, st);
  hipLaunchKernelGGL(k_qs_hist<1>, dim3(blocks), dim3(QS_NT), 0, s, x, nn, st);
  hipLaunchKernelGGL(k_qs_scan<1>, dim3(1), dim3(64 * QS_NSEL), 0, s, st);
  hipLaunchKernelGGL(k_qs_hist<2>, dim3(blocks), dim3(QS_NT), 0, s, x, nn, st);
  hipLaunchKernelGGL(k_qs_scan<2>, dim3(1), dim3(64 * QS_NSEL), 0, s, st);
  hipLaunchKernelGGL(k_qs_hist<3>, dim3(blocks), dim3(QS_NT), 0, s, x, nn, st);
  hipLaunchKernelGGL(k_qs_scan<3>, dim3(1), dim3(64 * QS_NSEL), 0, s, st);
  hipLaunchKernelGGL(k_qs_finish, dim3(1), dim3(64), 0, s, st, out2);
  return hipGetLastError() == hipSuccess ? QASR_OK : QASR_ERR_HIP;
}
}
