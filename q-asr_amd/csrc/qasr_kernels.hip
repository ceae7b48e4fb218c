// Hand-written gfx950 (CDNA4) kernels of the integer-only ASR encoder.
//
// Activations live in HBM as int8 [B][C][Tp] (time contiguous, Tp = T rounded up to 64).
//   k_pw      1x1 convs (pointwise / residual / block-17 / decoder) as int8 MFMA GEMMs
//             (v_mfma_i32_32x32x32_i8), int32 accumulate, fused fixed-point requant epilogue
//   k_dw_*    depthwise convs as wavefront stencils along time (v_dot4c_i32_i8 + v_alignbyte)
//   k_quant_in / k_requant / k_logsoftmax / k_lens   small element-wise passes
// Arithmetic contract: SURVEY.md Appendix A (reference: nemo/quantization/utils/quant_utils.py:163-216,
// quant_modules.py:272-309).  Integer results are bit-exact; see DESIGN.md for the proofs used.
#include "qasr_device.h"

namespace qasr {

// ------------------------------------------------------------------------------------------------ lens
__global__ void k_lens(const int32_t* lens_in, int32_t* lens_all, const qasr_domain_desc* doms, int n_domains, int B) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  lens_all[b] = lens_in[b];
  for (int d = 1; d < n_domains; ++d) {
    qasr_domain_desc dd = doms[d];
    int l = lens_all[dd.parent * B + b];
    // MaskedConv1d.get_seq_len (jasper.py:170-173); floor division like torch's `//` on non-negative values
    int num = l + 2 * (int)dd.padding - (int)dd.dilation * ((int)dd.kernel - 1) - 1;
    int q = num >= 0 ? num / (int)dd.stride : -((-num + (int)dd.stride - 1) / (int)dd.stride);
    lens_all[d * B + b] = q + 1;
  }
}
void launch_lens(hipStream_t s, const int32_t* lens_in, int32_t* lens_all, const qasr_domain_desc* doms,
                 int n_domains, int B) {
  hipLaunchKernelGGL(k_lens, dim3((B + 63) / 64), dim3(64), 0, s, lens_in, lens_all, doms, n_domains, B);
}

// ------------------------------------------------------------------------------------------------ quant_in
// First-layer QuantAct (quant_modules.py:180-184): q = clamp(rint(fl32(1/s) * x), -n, n-1); x masked at t >= len.
__global__ void k_quant_in(QuantInP p) {
  int tq = blockIdx.x * blockDim.x + threadIdx.x;      // dword index along time
  int c = blockIdx.y, b = blockIdx.z;
  if (tq * 4 >= p.Tp) return;
  int len = p.lens[b];
  const float* row = p.x + ((size_t)b * p.C + c) * p.T;
  int v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int t = tq * 4 + i;
    float x = (t < p.T && t < len) ? row[t] : 0.0f;
    float r = rintf(__fmul_rn(p.inv_scale, x));
    v[i] = (int)fminf(fmaxf(r, (float)p.lo), (float)p.hi);
  }
  ((unsigned*)(p.out + ((size_t)b * p.C + c) * p.Tp))[tq] = pack4(v[0], v[1], v[2], v[3]);
}
void launch_quant_in(hipStream_t s, const QuantInP& p) {
  dim3 g((p.Tp / 4 + 63) / 64, p.C, p.B);
  hipLaunchKernelGGL(k_quant_in, g, dim3(64), 0, s, p);
}

// ------------------------------------------------------------------------------------------------ depthwise
// Generic depthwise conv (any kernel / stride / dilation): one lane per output, taps read through L1.
__global__ void __launch_bounds__(256) k_dw_generic(DwP p) {
  const EpiP& e = p.e;
  int row = blockIdx.x;                       // b * C + c
  int b = row / p.C, c = row - b * p.C;
  int t = blockIdx.y * 256 + threadIdx.x;
  if (t >= e.Tp) return;
  const int8_t* xr = p.x + (size_t)row * p.Tp_in;
  const int8_t* wr = p.w + (size_t)c * p.kpad;
  int acc = p.bias[c];
  int base = t * p.stride - p.padding;
  for (int k = 0; k < p.K; ++k) {
    int ti = base + k * p.dilation;
    int v = (ti >= 0 && ti < p.T_in) ? (int)xr[ti] : 0;        // zero padding
    if (p.x_unsigned) v = (int)(int8_t)((v & 0xff) ^ 0x80);     // u8 fed as x-128; bias carries +128*sum(w)
    acc += (int)wr[k] * v;
  }
  if (e.acc_dbg && t < e.T) e.acc_dbg[(size_t)row * e.Tp + t] = acc;
  bool live = t < e.T && (!(e.flags & QASR_F_MASK_OUT) || t < e.lens[b]);
  int z = epi_z(acc, e, e.sb[c]);
  for (int j = 0; j < e.n_outs; ++j) {
    const OutP& o = e.outs[j];
    if (o.mode == 3) { ((int*)o.ptr)[(size_t)row * e.Tp + t] = live ? z : 0; continue; }
    int v = live ? out_value(z, o, o.mode == 1 ? o.mtab[c] : 0.0) : 0;
    ((int8_t*)o.ptr)[(size_t)row * e.Tp + t] = (int8_t)v;
  }
}

// Fast depthwise stencil, stride 1, dilation 1, compile-time kernel size K (odd).
// One wavefront owns 256 consecutive outputs of one (b, c) row: the input window is staged in LDS as dwords,
// every lane produces 4 consecutive outputs with v_dot4c_i32_i8 over byte-aligned windows (v_alignbyte_b32);
// the K taps are wave-uniform (scalar loads).
template <int K>
__global__ void __launch_bounds__(256) k_dw_fast(DwP p) {
  constexpr int PAD = K / 2;
  constexpr int HALO = (PAD + 3) / 4 * 4;
  constexpr int DELTA = HALO - PAD;
  constexpr int KP4 = (K + 3) / 4;
  constexpr int NW = 64 + 2 * (HALO / 4) + 1;     // staged dwords per wave
  constexpr int NR = KP4 + 2;                      // dwords each lane reads
  __shared__ unsigned win[4][NW + 3];
  const EpiP& e = p.e;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row_raw = blockIdx.x * 4 + wave;       // b * C + c
  const bool row_ok = row_raw < e.B * p.C;
  const int row = row_ok ? row_raw : 0;
  const int b = row / p.C, c = row - b * p.C;
  const int t0 = blockIdx.y * 256;
  const unsigned* xr = (const unsigned*)(p.x + (size_t)row * p.Tp_in);
  const unsigned flip = p.x_unsigned ? 0x80808080u : 0u;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int j = lane + 64 * i;
    if (j < NW) {
      int t = t0 - HALO + 4 * j;                   // multiple of 4
      unsigned v = (t >= 0 && t < p.Tp_in) ? xr[t >> 2] : 0u;
      win[wave][j] = v ^ flip;
    }
  }
  const int* wr = (const int*)(p.w + (size_t)c * (KP4 * 4));
  int wk[KP4];
#pragma unroll
  for (int j = 0; j < KP4; ++j) wk[j] = wr[j];
  __syncthreads();                                 // window staged (each wave reads only its own slice)
  unsigned xw[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) xw[i] = win[wave][lane + i];
  int acc[4];
  const int bias = p.bias[c];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int sh = (s + DELTA) & 3, q0 = (s + DELTA) >> 2;
    int a = bias;
#pragma unroll
    for (int j = 0; j < KP4; ++j) {
      unsigned d = sh ? __builtin_amdgcn_alignbyte(xw[q0 + j + 1], xw[q0 + j], sh) : xw[q0 + j];
      a = __builtin_amdgcn_sdot4((int)d, wk[j], a, false);
    }
    acc[s] = a;
  }
  const int tq = t0 + 4 * lane;
  if (!row_ok || tq >= e.Tp) return;
  if (e.acc_dbg) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (tq + s < e.T) e.acc_dbg[(size_t)row * e.Tp + tq + s] = acc[s];
  }
  const int lim = (e.flags & QASR_F_MASK_OUT) ? min(e.lens[b], e.T) : e.T;
  const float sb = e.sb[c];
  int z[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) z[s] = epi_z(acc[s], e, sb);
  for (int j = 0; j < e.n_outs; ++j) {
    const OutP& o = e.outs[j];
    if (o.mode == 3) {
      int* op = (int*)o.ptr + (size_t)row * e.Tp + tq;
#pragma unroll
      for (int s = 0; s < 4; ++s) op[s] = (tq + s < lim) ? z[s] : 0;
      continue;
    }
    const double Mc = o.mode == 1 ? o.mtab[c] : 0.0;
    int v[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) v[s] = (tq + s < lim) ? out_value(z[s], o, Mc) : 0;
    *(unsigned*)((int8_t*)o.ptr + (size_t)row * e.Tp + tq) = pack4(v[0], v[1], v[2], v[3]);
  }
}

void launch_dw(hipStream_t s, const DwP& p) {
  const int rows = p.e.B * p.C;
  const bool fast = p.stride == 1 && p.dilation == 1 && p.padding == p.K / 2 && (p.K & 1) && p.e.T == p.T_in;
  dim3 gf((rows + 3) / 4, (p.e.Tp + 255) / 256);
  if (fast && p.K == 33) { hipLaunchKernelGGL(k_dw_fast<33>, gf, dim3(256), 0, s, p); return; }
  if (fast && p.K == 39) { hipLaunchKernelGGL(k_dw_fast<39>, gf, dim3(256), 0, s, p); return; }
  if (fast && p.K == 51) { hipLaunchKernelGGL(k_dw_fast<51>, gf, dim3(256), 0, s, p); return; }
  if (fast && p.K == 63) { hipLaunchKernelGGL(k_dw_fast<63>, gf, dim3(256), 0, s, p); return; }
  if (fast && p.K == 75) { hipLaunchKernelGGL(k_dw_fast<75>, gf, dim3(256), 0, s, p); return; }
  if (fast && p.K == 11) { hipLaunchKernelGGL(k_dw_fast<11>, gf, dim3(256), 0, s, p); return; }
  if (fast && p.K == 13) { hipLaunchKernelGGL(k_dw_fast<13>, gf, dim3(256), 0, s, p); return; }
  dim3 g(rows, (p.e.Tp + 255) / 256);
  hipLaunchKernelGGL(k_dw_generic, g, dim3(256), 0, s, p);
}

// ------------------------------------------------------------------------------------------------ pointwise GEMM
// out[b][co][t] = epi( bias[co] + sum_ci W[co][ci] * X[b][ci][t] )   (+ residual 1x1 convs + res_act)
// Work-group tile: one utterance b, 64 time positions, 128 output channels; 4 wavefronts, wave w owns
// channels [32w, 32w+32) x 64 positions = two v_mfma_i32_32x32x32_i8 accumulators.
//   A operand (rows = time):  X tile transposed on the fly into LDS as Xs[t][ci] (v_perm 4x4 byte transposes)
//   B operand (cols = co):    W rows read straight from L2 into VGPRs (each element is used by one wave only)
// The accumulator layout then has the output channel on the lane (per-channel requant parameters are per-lane
// scalars) and 4 consecutive time steps in 4 consecutive registers (packs to one dword along time).
#define PW_TT 64
#define PW_MT 128
#define PW_KC 256
#define PW_XP (PW_KC + 16)   // LDS row pitch of Xs in bytes (16-B aligned; 68 dwords -> conflict-free b128 reads)
#define PW_OP 80             // LDS row pitch of the output staging tile (64 t + 16)

__device__ __forceinline__ void pw_gemm(v16i& acc0, v16i& acc1, const int8_t* __restrict__ x, const int8_t* __restrict__ w,
                                        int cin, int cin_pad, int Tp, bool x_unsigned, int b, int t0, int co_row,
                                        unsigned char* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cq = lane & 15, t16 = lane >> 4, h = lane >> 5, r31 = lane & 31;
  const unsigned flip = x_unsigned ? 0x80808080u : 0u;
  for (int kc = 0; kc < cin_pad; kc += PW_KC) {
    const int kw = min(PW_KC, cin_pad - kc);            // multiple of 64
    // (1) weight fragments for this K chunk: W[co_row][kc + 32*ks + 16*h .. +15]
    v4i wf[PW_KC / 32];
    const v4i* wp = w_frag(w, cin_pad, co_row, kc >> 5);
#pragma unroll
    for (int ks = 0; ks < PW_KC / 32; ++ks)
      if (32 * ks < kw) wf[ks] = wp[64 * ks];
    // (2) X chunk -> LDS, transposed: this wave stages input channels [kc + 64*wave, +64)
    if (64 * wave < kw) {
      const int ci0 = kc + 64 * wave + 4 * cq;
      v4i r[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ci = ci0 + j;
        r[j] = (ci < cin) ? *(const v4i*)(x + ((size_t)b * cin + ci) * Tp + t0 + 16 * t16) : (v4i){0, 0, 0, 0};
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        unsigned lo01 = __builtin_amdgcn_perm(r[1][q], r[0][q], 0x05010400u);
        unsigned hi01 = __builtin_amdgcn_perm(r[1][q], r[0][q], 0x07030602u);
        unsigned lo23 = __builtin_amdgcn_perm(r[3][q], r[2][q], 0x05010400u);
        unsigned hi23 = __builtin_amdgcn_perm(r[3][q], r[2][q], 0x07030602u);
        unsigned c0 = __builtin_amdgcn_perm(lo23, lo01, 0x05040100u) ^ flip;
        unsigned c1 = __builtin_amdgcn_perm(lo23, lo01, 0x07060302u) ^ flip;
        unsigned c2 = __builtin_amdgcn_perm(hi23, hi01, 0x05040100u) ^ flip;
        unsigned c3 = __builtin_amdgcn_perm(hi23, hi01, 0x07060302u) ^ flip;
        unsigned char* dst = smem + (16 * t16 + 4 * q) * PW_XP + 64 * wave + 4 * cq;
        *(unsigned*)(dst) = c0;
        *(unsigned*)(dst + PW_XP) = c1;
        *(unsigned*)(dst + 2 * PW_XP) = c2;
        *(unsigned*)(dst + 3 * PW_XP) = c3;
      }
    }
    __syncthreads();
    // (3) MFMA over the chunk
#pragma unroll
    for (int ks = 0; ks < PW_KC / 32; ++ks) {
      if (32 * ks < kw) {
        v4i a0 = *(const v4i*)(smem + r31 * PW_XP + 32 * ks + 16 * h);
        v4i a1 = *(const v4i*)(smem + (32 + r31) * PW_XP + 32 * ks + 16 * h);
        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, wf[ks], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, wf[ks], acc1, 0, 0, 0);
      }
    }
    __syncthreads();
  }
}

// time index of accumulator register r of tile ti for lane half h (C/D layout of the 32x32 MFMA)
__device__ __forceinline__ int pw_t(int ti, int r, int h) { return 32 * ti + (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ void pw_dump_acc(int32_t* dbg, const v16i& a0, const v16i& a1, int b, int co, int cout, int t0,
                                            int h, int T, int Tp) {
  if (!dbg || co >= cout) return;
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int t = t0 + pw_t(ti, r, h);
      if (t < T) dbg[((size_t)b * cout + co) * Tp + t] = ti ? a1[r] : a0[r];
    }
}

// ------------------------------------------------------------------------------------------------ dense conv (Jasper)
// Correctness-first implicit GEMM: same tile/MFMA structure as k_pw, the K loop runs over (tap, ci) and the
// X tile for tap k is the input shifted by k*dilation - padding (stride handled by the gather).  Staged with
// byte gathers (generic); performance work on this path is scheduled after the QuartzNet path.
__device__ __forceinline__ void dense_gemm(v16i& acc0, v16i& acc1, const DenseP& p, int b, int t0, int co_row,
                                           unsigned char* smem) {
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, r31 = lane & 31;
  const int flip = p.x_unsigned ? 0x80 : 0;
  for (int k = 0; k < p.K; ++k) {
    for (int kc = 0; kc < p.cin_pad; kc += PW_KC) {
      const int kw = min(PW_KC, p.cin_pad - kc);
      v4i wf[PW_KC / 32];
      const int8_t* wrow = p.w + ((size_t)co_row * p.K + k) * p.cin_pad + kc + 16 * h;
#pragma unroll
      for (int ks = 0; ks < PW_KC / 32; ++ks)
        if (32 * ks < kw) wf[ks] = *(const v4i*)(wrow + 32 * ks);
      // gather Xs[t][ci] = x[b][kc+ci][(t0+t)*stride - padding + k*dilation]  (zero outside [0, T_in))
      for (int idx = tid; idx < PW_TT * kw; idx += 256) {
        const int tl = idx & (PW_TT - 1), ci = idx >> 6;
        const int ti = (t0 + tl) * p.stride - p.padding + k * p.dilation;
        int v = 0;
        if (kc + ci < p.cin && ti >= 0 && ti < p.T_in) v = p.x[((size_t)b * p.cin + kc + ci) * p.Tp_in + ti] & 0xff;
        smem[tl * PW_XP + ci] = (unsigned char)(v ^ flip);
      }
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < PW_KC / 32; ++ks) {
        if (32 * ks < kw) {
          v4i a0 = *(const v4i*)(smem + r31 * PW_XP + 32 * ks + 16 * h);
          v4i a1 = *(const v4i*)(smem + (32 + r31) * PW_XP + 32 * ks + 16 * h);
          acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, wf[ks], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, wf[ks], acc1, 0, 0, 0);
        }
      }
      __syncthreads();
    }
  }
}

__global__ void __launch_bounds__(256) k_dense(DenseP p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[PW_TT * PW_XP];
  const EpiP& e = p.e;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const int t0 = blockIdx.x * PW_TT, b = blockIdx.z;
  const int co_l = 32 * wave + (lane & 31);
  const int co = blockIdx.y * PW_MT + co_l;
  v16i acc0, acc1;
  {
    const int bv = p.bias[co];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = bv; acc1[r] = bv; }
  }
  dense_gemm(acc0, acc1, p, b, t0, co, smem);
  pw_dump_acc(e.acc_dbg, acc0, acc1, b, co, e.cout, t0, h, e.T, e.Tp);
  const int lim = (e.flags & QASR_F_MASK_OUT) ? min(e.lens[b], e.T) : e.T;
  int z0[16], z1[16];
  if (e.flags & QASR_F_RESADD) {
    const bool exact = e.flags & QASR_F_EXACT_Z;
    const double Mm = e.m_main[co];
    const float sbm = e.sb[co];
    double d0[16], d1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      d0[r] = requant_d(exact ? z_roundtrip(acc0[r], sbm, false) : acc0[r], Mm);
      d1[r] = requant_d(exact ? z_roundtrip(acc1[r], sbm, false) : acc1[r], Mm);
    }
    for (int pi = 0; pi < p.n_panes; ++pi) {
      const PaneP& pn = p.panes[pi];
      const int bv = pn.bias[co];
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = bv; acc1[r] = bv; }
      pw_gemm(acc0, acc1, pn.x, pn.w, pn.cin, pn.cin_pad, e.Tp, pn.x_unsigned, b, t0, co, smem);
      pw_dump_acc(pn.acc_dbg, acc0, acc1, b, co, e.cout, t0, h, e.T, e.Tp);
      const double Mp = pn.m[co];
      const float sbp = pn.sb[co];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        double s0 = d0[r] + requant_d(exact ? z_roundtrip(acc0[r], sbp, false) : acc0[r], Mp);
        double s1 = d1[r] + requant_d(exact ? z_roundtrip(acc1[r], sbp, false) : acc1[r], Mp);
        d0[r] = fmin(fmax(s0, (double)e.qlo), (double)e.qhi);
        d1[r] = fmin(fmax(s1, (double)e.qlo), (double)e.qhi);
      }
    }
    const bool relu = e.flags & QASR_F_RELU;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int q0 = (int)d0[r], q1 = (int)d1[r];
      z0[r] = relu ? max(q0, 0) : q0;
      z1[r] = relu ? max(q1, 0) : q1;
    }
  } else {
    const float sb = e.sb[co];
#pragma unroll
    for (int r = 0; r < 16; ++r) { z0[r] = epi_z(acc0[r], e, sb); z1[r] = epi_z(acc1[r], e, sb); }
  }
  for (int j = 0; j < e.n_outs; ++j) {
    const OutP& o = e.outs[j];
    if (o.mode == 3) {
      if (co < e.cout) {
        int* op = (int*)o.ptr + ((size_t)b * e.cout + co) * e.Tp + t0;
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            int tl = pw_t(ti, r, h);
            op[tl] = (t0 + tl < lim) ? (ti ? z1[r] : z0[r]) : 0;
          }
      }
      continue;
    }
    const double Mc = (o.mode == 1) ? o.mtab[co] : 0.0;
    __syncthreads();
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        int v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * g + i;
          const int tl = pw_t(ti, r, h);
          const int z = ti ? z1[r] : z0[r];
          v[i] = (t0 + tl < lim) ? out_value(z, o, Mc) : 0;
        }
        *(unsigned*)(smem + co_l * PW_OP + 32 * ti + 8 * g + 4 * h) = pack4(v[0], v[1], v[2], v[3]);
      }
    __syncthreads();
    {
      const int row = tid >> 1, half = tid & 1;
      const int cor = blockIdx.y * PW_MT + row;
      if (cor < e.cout) {
        int8_t* dst = (int8_t*)o.ptr + ((size_t)b * e.cout + cor) * e.Tp + t0 + 32 * half;
        const v4i* src = (const v4i*)(smem + row * PW_OP + 32 * half);
        ((v4i*)dst)[0] = src[0];
        ((v4i*)dst)[1] = src[1];
      }
    }
  }
}

void launch_dense(hipStream_t s, const DenseP& p) {
  const int cout_pad = (p.e.cout + PW_MT - 1) / PW_MT * PW_MT;
  dim3 g(p.e.Tp / PW_TT, cout_pad / PW_MT, p.e.B);
  hipLaunchKernelGGL(k_dense, g, dim3(256), 0, s, p);
}

// ------------------------------------------------------------------------------------------------ requant (stand-alone)
// One lane = 16 consecutive frames of one (utterance, channel) row: wide loads, the production requant_batch
// (float32 fast path + fp64 fallback), one 16-byte store per consumer (up to QASR_RQ_MAX consumers per launch: Jasper's
// dense-residual values feed up to 11 QuantActs, and the stored value is read once for all of them).
__global__ void __launch_bounds__(256) k_requant(RequantP p) {
  const int tq = blockIdx.x * blockDim.x + threadIdx.x;      // 16-frame group along time
  const int row = blockIdx.y;                                 // b * C + c
  if (tq * 16 >= p.Tp) return;
  const int b = row / p.C, c = row - b * p.C;
  const size_t idx = (size_t)row * p.Tp + 16 * tq;
  int acc[16];
  if (p.in_is_i32) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const v4i v = *(const v4i*)((const int*)p.in + idx + 4 * g);
      acc[4 * g] = v[0]; acc[4 * g + 1] = v[1]; acc[4 * g + 2] = v[2]; acc[4 * g + 3] = v[3];
    }
  } else {
    const v4i v = *(const v4i*)((const int8_t*)p.in + idx);
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (int)(int8_t)((unsigned)v[i >> 2] >> (8 * (i & 3)));
  }
  const bool relu = p.flags & QASR_F_RELU;
  int z[16];
  if (p.flags & QASR_F_EXACT_Z) {
    const float sb = p.sb[c];
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = z_roundtrip(acc[i], sb, relu);
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = relu ? max(acc[i], 0) : acc[i];
  }
  const int lim = (p.flags & QASR_F_MASK_OUT) ? min(p.T, p.lens[b]) : p.T;
  for (int j = 0; j < p.n_outs; ++j) {                        // the stored value is read once for all its consumers
    const OutP& o = p.outs[j];
    int q[16];
    if (o.mode == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) q[i] = z[i];
    } else {
      requant_batch<16>(q, z, o.mode == 1 ? o.mtab[c] : o.m, o.lo, o.hi);
    }
    v4i pk;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      int v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = (16 * tq + 4 * g + i < lim) ? q[4 * g + i] : 0;
      pk[g] = (int)pack4(v[0], v[1], v[2], v[3]);
    }
    *(v4i*)((int8_t*)o.ptr + idx) = pk;
  }
}
void launch_requant(hipStream_t s, const RequantP& p) {
  dim3 g((p.Tp / 16 + 63) / 64, p.B * p.C);
  hipLaunchKernelGGL(k_requant, g, dim3(64), 0, s, p);
}

// ------------------------------------------------------------------------------------------------ log-softmax + argmax
// torch.nn.functional.log_softmax(dim=-1) + argmax (conv_asr.py:275, ctc_models.py:405); one lane per (b, t) row.
__global__ void k_logsoftmax(const float* logits, float* logp, int32_t* tokens, int rows, int ncls) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float* x = logits + (size_t)r * ncls;
  float m = x[0];
  int am = 0;
  for (int c = 1; c < ncls; ++c) {
    float v = x[c];
    if (v > m) { m = v; am = c; }            // first maximum wins, like torch.argmax
  }
  float s = 0.f;
  for (int c = 0; c < ncls; ++c) s += expf(x[c] - m);
  float ls = logf(s);
  if (logp)
    for (int c = 0; c < ncls; ++c) logp[(size_t)r * ncls + c] = (x[c] - m) - ls;
  if (tokens) tokens[r] = am;
}
void launch_logsoftmax(hipStream_t s, const float* logits, float* logp, int32_t* tokens, int rows, int ncls) {
  hipLaunchKernelGGL(k_logsoftmax, dim3((rows + 127) / 128), dim3(128), 0, s, logits, logp, tokens, rows, ncls);
}

}  // namespace qasr
