// Dynamic-quantisation device path: QuantAct.forward with dynamic=True
// (nemo/quantization/utils/quant_modules.py:149-194) re-derives every activation range from the batch it is looking
// at, so the requantisation multipliers, the bias integers of the following conv and its output scales are data
// dependent.  The reference does this with a .min()/.max() pair, host-side numpy frexp (quant_utils.py:121-147, a
// device -> host -> device round trip per QuantAct) and fp64 tensor arithmetic; here every step is a kernel on the
// caller's stream and nothing leaves the device:
//   k_dyn_range        min / max of the float32 tensor QuantAct sees — the int32 accumulators (or int8 codes) of its
//                      producer(s) times their float32 scales, (+ identity,) ReLU, MaskedConv1d's length mask
//   k_dyn_x_act        the same tensor written out for the percentile ranges (torch.quantile twice, quant_modules.py:158-167
//                      -> qasr_quantile2's radix select) when qm.set_percentile is in force
//   k_dyn_act_params   act_scaling_factor (quant_utils.py:28-54) and, per channel, fixedpoint_mul's multiplier
//                      m 2^-e from batch_frexp of f64(pre_sf) / f64(act_sf) (quant_utils.py:190-196,121-147)
//   k_dyn_requant      fixedpoint_mul.forward (quant_utils.py:163-216) for one or two operands -> int8 / uint8 codes
//   k_dyn_quant_in     the first layer's SymmetricQuantFunction on float features (quant_modules.py:180-184)
//   k_dyn_conv_params  the next conv's output scales s_w[c] * act_sf and bias integers (quant_modules.py:293-299)
// The conv accumulators themselves come from the production kernels (qasr_dw_conv_acc / qasr_pw_conv_acc).
#include "qasr_device.h"

#include <algorithm>
#include <cstdio>

namespace qasr {

// float -> unsigned whose order is the float order (for atomicMin / atomicMax)
__device__ __forceinline__ unsigned f2ord(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

struct DynView {            // a float32 tensor given as integers x per-channel (or per-tensor) float32 scale
  const void* v;            // int32 [B][C][Tp] accumulators, or int8 [B][C][Tp] codes
  const float* s;           // [C] (per_channel) or [1]
  int is8, per_channel;
  const int32_t* r_lo;      // division residue of the conv that produced v, in units of 2^-24: r_lo + 128 r_hi
  const int32_t* r_hi;      // (nullptr: none)
};

__device__ __forceinline__ float view_at(const DynView& a, size_t i, int c) {
  const int z = a.is8 ? (int)((const int8_t*)a.v)[i] : ((const int32_t*)a.v)[i];
  float zf = (float)z;
  // conv_int = F.conv1d(x_int, ...) in double on x_int = fl32(x / pre_sf): the integers plus sum(w residue)
  // (quant_modules.py:301-305); .type(torch.float) rounds the sum once
  if (a.r_lo) zf = (float)((double)z + ldexp((double)((long long)a.r_lo[i] + 128ll * a.r_hi[i]), -24));
  return mul_f32_unfused(zf, a.s[a.per_channel ? c : 0]);           // conv_int.float() * scale (quant_modules.py:305-308)
}

struct DynRangeP {
  DynView a, b;             // b.v == nullptr: one operand; else x_act = identity(b) + x(a) (quant_modules.py:108)
  const float* xf;          // first layer: the float features themselves [B][C][Tx] (a, b unused)
  int Tx;
  const int32_t* lens;      // MaskedConv1d mask (jasper.py:177-181): t >= lens[b] reads as 0; nullptr: none (res_act)
  int relu, B, C, T, Tp;
  unsigned* out;            // [0] = ordered min, [1] = ordered max
};

__global__ void k_dyn_range_init(unsigned* out) {
  out[0] = 0xffffffffu;
  out[1] = 0u;
}

// x_act at (row = (b, c), t): what QuantAct.forward sees (quant_modules.py:108), as float32
__device__ __forceinline__ float dyn_x_act(const DynRangeP& p, int row, int c, int t, int len) {
  if (t >= len) return 0.f;
  if (p.xf) return p.xf[(size_t)row * p.Tx + t];
  const size_t i = (size_t)row * p.Tp + t;
  float v = view_at(p.a, i, c);
  if (p.b.v) v = view_at(p.b, i, c) + v;
  return p.relu ? fmaxf(v, 0.f) : v;
}

__global__ void __launch_bounds__(256) k_dyn_range(DynRangeP p) {
  const int row = blockIdx.x;                                // (b, c)
  const int b = row / p.C, c = row - b * p.C;
  const int len = p.lens ? min(p.lens[b], p.T) : p.T;
  float lo = INFINITY, hi = -INFINITY;
  for (int t = threadIdx.x; t < p.T; t += 256) {
    const float v = dyn_x_act(p, row, c, t, len);
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, o));
    hi = fmaxf(hi, __shfl_xor(hi, o));
  }
  if ((threadIdx.x & 63) == 0 && lo <= hi) {
    atomicMin(&p.out[0], f2ord(lo));
    atomicMax(&p.out[1], f2ord(hi));
  }
}

// percentile ranges (quant_modules.py:158-167): x_act itself, [B][C][T] without the row padding, for the radix select
__global__ void __launch_bounds__(256) k_dyn_x_act(DynRangeP p, float* __restrict__ out) {
  const int row = blockIdx.x;
  const int b = row / p.C, c = row - b * p.C;
  const int len = p.lens ? min(p.lens[b], p.T) : p.T;
  for (int t = threadIdx.x; t < p.T; t += 256) out[(size_t)row * p.T + t] = dyn_x_act(p, row, c, t, len);
}

// the two quantiles (float32, written over minmax by k_qs_finish) -> the ordered form of k_dyn_range
__global__ void k_dyn_range_encode(unsigned* mm) {
  const float lo = __uint_as_float(mm[0]), hi = __uint_as_float(mm[1]);
  mm[0] = f2ord(lo);
  mm[1] = f2ord(hi);
}

// symmetric_linear_quantization_params (quant_utils.py:44-54), float32
__device__ __forceinline__ float dyn_scale(const unsigned* mm, int bits) {
  const float n = (float)((1 << (bits - 1)) - 1);
  const float m = fmaxf(fmaxf(fabsf(ord2f(mm[0])), fabsf(ord2f(mm[1]))), 1e-8f);
  return __fdiv_rn(m, n);
}
// batch_frexp (quant_utils.py:121-147): r = mant 2^ex, m = round_half_up(mant 2^31), e = 31 - ex; returned as the
// float64 m 2^-e the kernels multiply with (exact: m <= 2^31)
__device__ __forceinline__ double dyn_multiplier(float pre, float act) {
  const double r = (double)pre / (double)act;
  int ex;
  const double mant = frexp(r, &ex);
  const double m = floor(mant * 2147483648.0 + 0.5);
  return ldexp(m, ex - 31);
}

struct DynActP {
  const unsigned* mm;
  int bits, C;
  const float* sa;          // scales of operand a: [C] or [1]
  int a_per_channel;
  const float* sb;          // operand b (nullptr: none)
  int b_per_channel;
  float* s_out;             // [1] act_scaling_factor
  double* Ma;               // [C]
  double* Mb;               // [C] (with sb)
};

__global__ void __launch_bounds__(256) k_dyn_act_params(DynActP p) {
  const float s = dyn_scale(p.mm, p.bits);
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c == 0) p.s_out[0] = s;
  if (c >= p.C) return;
  if (p.sa) p.Ma[c] = dyn_multiplier(p.sa[p.a_per_channel ? c : 0], s);
  if (p.sb) p.Mb[c] = dyn_multiplier(p.sb[p.b_per_channel ? c : 0], s);
}

struct DynRequantP {
  DynView a, b;
  const double* Ma;
  const double* Mb;
  const int32_t* lens;
  int relu, B, C, T, Tp, lo, hi;
  int8_t* out;              // [B][C][Tp]; columns >= min(lens[b], T) are written as 0
};

// z_int = round(pre_act / pre_act_scaling_factor) (quant_utils.py:187) through the float32 view, then
// round(f64(z) m / 2^e) per operand, sum, clamp (quant_utils.py:196-214)
__device__ __forceinline__ double dyn_operand(const DynView& a, size_t i, int c, bool relu, double M) {
  const float s = a.s[a.per_channel ? c : 0];
  float y = view_at(a, i, c);
  if (relu) y = fmaxf(y, 0.f);
  const double z = (double)rintf(__fdiv_rn(y, s));
  return rint(z * M);
}

__global__ void __launch_bounds__(256) k_dyn_requant(DynRequantP p) {
  const int row = blockIdx.x;
  const int b = row / p.C, c = row - b * p.C;
  const int len = p.lens ? min(p.lens[b], p.T) : p.T;
  const double Ma = p.Ma[c], Mb = p.b.v ? p.Mb[c] : 0.0;
  for (int t = threadIdx.x; t < p.Tp; t += 256) {
    const size_t i = (size_t)row * p.Tp + t;
    int q = 0;
    if (t < len) {
      double r = dyn_operand(p.a, i, c, p.relu, Ma);
      if (p.b.v) r = dyn_operand(p.b, i, c, false, Mb) + r;
      q = (int)fmin(fmax(r, (double)p.lo), (double)p.hi);
    }
    p.out[i] = (int8_t)q;
  }
}

struct DynQuantInP {
  const float* x;           // [B][C][Tx]
  const unsigned* mm;
  const int32_t* lens;
  int bits, B, C, T, Tx, Tp;
  float* s_out;
  int8_t* out;              // [B][C][Tp]
};

// first layer: clamp(round(fl32(1/s) x), -n, n-1) (quant_utils.py:12-26,57-79); the fixedpoint_mul that follows
// multiplies by s/s = 1 and clamps to [-n-1, n] (quant_modules.py:180-190): the identity on these codes
__global__ void __launch_bounds__(256) k_dyn_quant_in(DynQuantInP p) {
  const int row = blockIdx.x;
  const int b = row / p.C;
  const float s = dyn_scale(p.mm, p.bits);
  if (row == 0 && threadIdx.x == 0) p.s_out[0] = s;
  const float inv = __fdiv_rn(1.0f, s);
  const float n = (float)((1 << (p.bits - 1)) - 1);
  const int len = min(p.lens[b], p.T);
  for (int t = threadIdx.x; t < p.Tp; t += 256) {
    int q = 0;
    if (t < len) q = (int)fminf(fmaxf(rintf(mul_f32_unfused(inv, p.x[(size_t)row * p.Tx + t])), -n), n - 1.0f);
    p.out[(size_t)row * p.Tp + t] = (int8_t)q;
  }
}

// x_int = (x / pre_act_scaling_factor).type(torch.double) (quant_modules.py:301) with x = fl32(q s) (QuantAct's output,
// :192): the float32 quotient is q or a float32 neighbour of q, so conv_int carries sum(w residue) beside the integers.
// residue(q) = fl32(fl32(q s) / s) - q is a multiple of 2^-24 with |residue| <= 2^-16 for |q| <= 255: written as
// lo + 128 hi (|lo| <= 64, |hi| <= 2) in units of 2^-24, two int8 tensors the integer conv kernels run on.
struct DynResidueP {
  const unsigned char* codes;
  const float* s_x;         // [1]
  int x_unsigned;
  size_t n;                 // bytes, multiple of 16
  int8_t* lo;
  int8_t* hi;
};

__global__ void __launch_bounds__(256) k_dyn_residue_codes(DynResidueP p) {
  __shared__ short lut[256];
  {
    const int idx = threadIdx.x;
    const int q = p.x_unsigned ? idx : (int)(int8_t)idx;
    const float s = p.s_x[0];
    const float back = __fdiv_rn(mul_f32_unfused((float)q, s), s);
    const int v = (int)rint(ldexp((double)back - (double)q, 24));
    const int hi = (v + (v >= 0 ? 64 : -64)) / 128;              // nearest multiple of 128 (|v| <= 256)
    const int lo = v - 128 * hi;
    lut[idx] = (short)(((hi & 0xff) << 8) | (lo & 0xff));
  }
  __syncthreads();
  for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; 16 * g < p.n; g += (size_t)gridDim.x * 256) {
    const uint4 c = *(const uint4*)(p.codes + 16 * g);
    const unsigned in[4] = {c.x, c.y, c.z, c.w};
    unsigned lo[4], hi[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      lo[d] = hi[d] = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned e = (unsigned short)lut[(in[d] >> (8 * k)) & 0xff];
        lo[d] |= (e & 0xff) << (8 * k);
        hi[d] |= (e >> 8) << (8 * k);
      }
    }
    *(uint4*)(p.lo + 16 * g) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    *(uint4*)(p.hi + 16 * g) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
  }
}

struct DynConvP {
  const float* s_x;         // [1] act_scaling_factor of the conv's input
  const float* s_w;         // [C] per-channel weight scales
  const float* bprime;      // [C] (BN-folded) float bias, nullptr: none
  const int32_t* wsum128;   // [C] 128 sum(W[c]) for inputs stored as u8 (kernels feed x - 128), nullptr: signed input
  int C, C_pad;
  float* sf_out;            // [C_pad] correct_scaling_factor = s_w s_x (quant_modules.py:294,307)
  int32_t* bias;            // [C_pad]
};

__global__ void __launch_bounds__(256) k_dyn_conv_params(DynConvP p) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= p.C_pad) return;
  if (c >= p.C) {
    p.sf_out[c] = 1.0f;
    p.bias[c] = 0;
    return;
  }
  const float sb = mul_f32_unfused(p.s_w[c], p.s_x[0]);
  p.sf_out[c] = sb;
  long long bi = 0;
  if (p.bprime) {           // SymmetricQuantFunction at 32 bits, evaluated in float32 (quant_modules.py:293-299)
    const float q = rintf(mul_f32_unfused(__fdiv_rn(1.0f, sb), p.bprime[c]));
    bi = (long long)fminf(fmaxf(q, -2147483648.0f), 2147483520.0f);
  }
  if (p.wsum128) bi += p.wsum128[c];
  p.bias[c] = (int32_t)bi;
}

}  // namespace qasr

using namespace qasr;

static DynView make_view(const qasr_dyn_view* q) {
  DynView a;
  a.v = q->data;
  a.s = q->scale;
  a.is8 = q->is_int8;
  a.per_channel = q->per_channel;
  a.r_lo = q->residue_lo;
  a.r_hi = q->residue_hi;
  return a;
}
static bool view_ok(const qasr_dyn_view* q) {
  return q && q->data && q->scale && (!q->residue_lo == !q->residue_hi) && !(q->is_int8 && q->residue_lo);
}

extern "C" {

int qasr_dyn_range(void* stream, const qasr_dyn_view* a, const qasr_dyn_view* b, const float* xf, int Tx, const int32_t* lens,
                   int relu, int B, int C, int T, int Tp, uint32_t* minmax) {
  if (!minmax || B < 1 || C < 1 || T < 1 || (!xf && (!view_ok(a) || T > Tp)) || (xf && T > Tx) || (b && !view_ok(b)))
    return QASR_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  DynRangeP p{};
  if (!xf) p.a = make_view(a);
  if (b) p.b = make_view(b);
  p.xf = xf;
  p.Tx = Tx;
  p.lens = lens;
  p.relu = relu;
  p.B = B, p.C = C, p.T = T, p.Tp = Tp;
  p.out = minmax;
  hipLaunchKernelGGL(k_dyn_range_init, dim3(1), dim3(1), 0, s, minmax);
  hipLaunchKernelGGL(k_dyn_range, dim3(B * C), dim3(256), 0, s, p);
  return hipGetLastError() == hipSuccess ? QASR_OK : QASR_ERR_HIP;
}

int qasr_dyn_range_percentile(void* stream, const qasr_dyn_view* a, const qasr_dyn_view* b, const float* xf, int Tx,
                              const int32_t* lens, int relu, int B, int C, int T, int Tp, float q_lo, float q_hi, float* x_act,
                              void* workspace, size_t workspace_bytes, uint32_t* minmax) {
  if (!minmax || !x_act || B < 1 || C < 1 || T < 1 || (!xf && (!view_ok(a) || T > Tp)) || (xf && T > Tx) || (b && !view_ok(b)))
    return QASR_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  DynRangeP p{};
  if (!xf) p.a = make_view(a);
  if (b) p.b = make_view(b);
  p.xf = xf;
  p.Tx = Tx;
  p.lens = lens;
  p.relu = relu;
  p.B = B, p.C = C, p.T = T, p.Tp = Tp;
  hipLaunchKernelGGL(k_dyn_x_act, dim3(B * C), dim3(256), 0, s, p, x_act);
  const int rc = qasr_quantile2(stream, x_act, (size_t)B * C * T, q_lo, q_hi, (float*)minmax, workspace, workspace_bytes);
  if (rc != QASR_OK) return rc;
  hipLaunchKernelGGL(k_dyn_range_encode, dim3(1), dim3(1), 0, s, minmax);
  return hipGetLastError() == hipSuccess ? QASR_OK : QASR_ERR_HIP;
}

int qasr_dyn_act_params(void* stream, const uint32_t* minmax, int bits, int C, const float* sa, int a_per_channel,
                        const float* sb, int b_per_channel, float* s_out, double* Ma, double* Mb) {
  if (!minmax || bits < 2 || bits > 16 || C < 1 || !s_out || (sa && !Ma) || (sb && !Mb)) return QASR_ERR_ARG;
  DynActP p{};
  p.mm = minmax;
  p.bits = bits;
  p.C = C;
  p.sa = sa, p.a_per_channel = a_per_channel;
  p.sb = sb, p.b_per_channel = b_per_channel;
  p.s_out = s_out;
  p.Ma = Ma, p.Mb = Mb;
  hipLaunchKernelGGL(k_dyn_act_params, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, p);
  return hipGetLastError() == hipSuccess ? QASR_OK : QASR_ERR_HIP;
}

int qasr_dyn_requant(void* stream, const qasr_dyn_view* a, const double* Ma, const qasr_dyn_view* b, const double* Mb,
                     const int32_t* lens, int relu, int B, int C, int T, int Tp, int lo, int hi, int8_t* out) {
  if (!view_ok(a) || !Ma || !out || (b && (!view_ok(b) || !Mb)) || B < 1 || C < 1 || T > Tp ||
      lo > hi || lo < -256 || hi > 255 || hi - lo > 255)
    return QASR_ERR_ARG;
  DynRequantP p{};
  p.a = make_view(a);
  if (b) p.b = make_view(b);
  p.Ma = Ma, p.Mb = Mb;
  p.lens = lens;
  p.relu = relu;
  p.B = B, p.C = C, p.T = T, p.Tp = Tp, p.lo = lo, p.hi = hi;
  p.out = out;
  hipLaunchKernelGGL(k_dyn_requant, dim3(B * C), dim3(256), 0, (hipStream_t)stream, p);
  return hipGetLastError() == hipSuccess ? QASR_OK : QASR_ERR_HIP;
}

int qasr_dyn_quant_in(void* stream, const float* x, int Tx, const uint32_t* minmax, const int32_t* lens, int bits, int B,
                      int C, int T, int Tp, float* s_out, int8_t* out) {
  if (!x || !minmax || !lens || !s_out || !out || bits < 2 || bits > 8 || T > Tx || T > Tp) return QASR_ERR_ARG;
  DynQuantInP p{};
  p.x = x, p.mm = minmax, p.lens = lens;
  p.bits = bits, p.B = B, p.C = C, p.T = T, p.Tx = Tx, p.Tp = Tp;
  p.s_out = s_out, p.out = out;
  hipLaunchKernelGGL(k_dyn_quant_in, dim3(B * C), dim3(256), 0, (hipStream_t)stream, p);
  return hipGetLastError() == hipSuccess ? QASR_OK : QASR_ERR_HIP;
}

int qasr_dyn_residue_codes(void* stream, const int8_t* codes, int x_unsigned, const float* s_x, size_t n, int8_t* lo,
                           int8_t* hi) {
  if (!codes || !s_x || !lo || !hi || n == 0 || (n & 15) || (((uintptr_t)codes | (uintptr_t)lo | (uintptr_t)hi) & 15))
    return QASR_ERR_ARG;
  DynResidueP p{};
  p.codes = (const unsigned char*)codes, p.s_x = s_x, p.x_unsigned = x_unsigned, p.n = n, p.lo = lo, p.hi = hi;
  const int blocks = (int)std::min<size_t>(2048, (n / 16 + 255) / 256);
  hipLaunchKernelGGL(k_dyn_residue_codes, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
  return hipGetLastError() == hipSuccess ? QASR_OK : QASR_ERR_HIP;
}

int qasr_dyn_conv_params(void* stream, const float* s_x, const float* s_w, const float* bprime, const int32_t* wsum128,
                         int C, int C_pad, float* sf_out, int32_t* bias) {
  if (!s_x || !s_w || !sf_out || !bias || C < 1 || C_pad < C) return QASR_ERR_ARG;
  DynConvP p{};
  p.s_x = s_x, p.s_w = s_w, p.bprime = bprime, p.wsum128 = wsum128;
  p.C = C, p.C_pad = C_pad;
  p.sf_out = sf_out, p.bias = bias;
  hipLaunchKernelGGL(k_dyn_conv_params, dim3((C_pad + 255) / 256), dim3(256), 0, (hipStream_t)stream, p);
  return hipGetLastError() == hipSuccess ? QASR_OK : QASR_ERR_HIP;
}

}  // extern "C"
