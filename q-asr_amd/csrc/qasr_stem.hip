// Fused stem: block 0 of a time-channel-separable encoder (QuartzNet: examples/asr/conf/quartznet_15x5.yaml:55-66 -
// depthwise k33 stride 2 on the 64 mel features, 1x1 64 -> 256, BN, ReLU) as ONE launch instead of four:
//   lengths of every time domain      MaskedConv1d.get_seq_len, jasper.py:170-173            (was k_lens)
//   [normalize_batch                  features.py:53-67, from k_mel's per-tile sums           (was k_norm; forward_audio)]
//   first-layer QuantAct              quant_modules.py:180-184: clamp(round(fl32(1/s) x))     (was k_quant_in)
//   strided depthwise QuantConv1d     quant_modules.py:301-305 + the 1x1 conv's QuantAct      (was k_dw_generic)
//   1x1 QuantConv1d + consumers' QuantAct                                                     (was k_sep<0, 1, 1>)
// The three intermediate tensors never exist.  Work-group = one utterance x 32 output frames: the float32 feature
// window (64 channels x 95 frames) is quantised straight into LDS, every thread produces 4 consecutive depthwise
// outputs of one channel with v_dot4_i32_i8 over byte-aligned windows (stride 2: v_alignbyte by 0 / 2), the
// requantised result is the [channel][frame] A image of the 1x1 GEMM (ds_read_b64_tr_b8 fragments, 32x32x32 MFMA,
// K = 64 padded to 128 with zero rows), and the epilogue is k_sep2's: requantise per consumer in the MFMA C layout,
// two permlane32 swaps, one 16-byte store per lane.
#include "qasr_sep2_impl.h"

namespace qasr {

#define STEM_NT 512
#define STEM_TT 32
#define STEM_KP4 9            /* taps padded to 36 */
#define STEM_CMAX 64
#define STEM_NIN (2 * (STEM_TT - 1) + 4 * STEM_KP4)   /* input frames a work-group touches, 98 (of which 95 carry taps) */
#define STEM_XP 112           /* LDS row pitch of the quantised window (bytes, multiple of 16) */

struct StemP {
  // first layer
  const float* x;           // [B][C][Tx] float32 features
  const int32_t* lens_in;   // [B]
  float inv_scale;
  int qlo, qhi, C, Tx;
  // depthwise, stride 2
  const int8_t* wdw;        // [C][36]
  const int32_t* bias_dw;   // [C_pad]
  const double* m_dw;       // [C_pad] towards the 1x1 conv's QuantAct
  const float* sb_dw;       // [C_pad] (EXACT_Z)
  unsigned flags_dw;
  int dw_lo, dw_hi, K, padding, T_mid;
  int32_t* dw_acc_dbg;      // optional i32 [B][C][Tp]
  // 1x1 conv
  const int8_t* w;          // fragment order, cin_pad = 128
  const int32_t* bias;
  int cin_pad;
  EpiP e;                   // outs (mode 1), flags, sb, acc_dbg, T, Tp, cout, B
  // lengths of all time domains (k_lens)
  const qasr_domain_desc* doms;
  int n_domains;
  int32_t* lens_all;        // [n_domains][B]
  // normalize_batch folded in (features.py:53-67; forward_audio): x holds k_mel's un-normalised log-mel and stats its
  // per-tile sums [B][n_stat_tiles][C][2] (sum, squared deviations from the tile mean) over 16-frame tiles; nullptr: x is
  // normalised already
  const double* stats;
  int n_stat_tiles, n_frames;
};

__device__ __forceinline__ int stem_out_len(int l, int kernel, int stride, int dilation, int padding) {
  const int num = l + 2 * padding - dilation * (kernel - 1) - 1;
  const int q = num >= 0 ? num / stride : -((-num + stride - 1) / stride);
  return q + 1;
}

__global__ void __launch_bounds__(STEM_NT) k_stem(StemP p) {
  __shared__ __attribute__((aligned(16))) unsigned char xq[STEM_CMAX * STEM_XP];   // quantised window [C][XP]
  __shared__ __attribute__((aligned(16))) unsigned char xd[128 * 32];              // [channel][32 frames] A image
  const EpiP& e = p.e;
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x, t0 = blockIdx.y * STEM_TT;
  // ---- lengths of every domain, once (the other kernels of the forward read them; this one derives its own)
  if (blockIdx.x == 0 && blockIdx.y == 0) {
    for (int bb = tid; bb < e.B; bb += STEM_NT) {
      p.lens_all[bb] = p.lens_in[bb];
      for (int d = 1; d < p.n_domains; ++d) {
        const qasr_domain_desc dd = p.doms[d];
        p.lens_all[d * e.B + bb] = stem_out_len(p.lens_all[dd.parent * e.B + bb], (int)dd.kernel, (int)dd.stride, (int)dd.dilation,
                                                (int)dd.padding);
      }
    }
  }
  const int len_in = p.lens_in[b];
  const int len_mid = stem_out_len(len_in, p.K, 2, 1, p.padding);
  // ---- first-layer QuantAct of the window: input frame of byte i = 2 t0 - padding + i
  const int tin0 = 2 * t0 - p.padding;
  const int lim_in = min(len_in, p.stats ? p.n_frames : p.Tx);
  __shared__ float s_mean[STEM_CMAX], s_sd[STEM_CMAX];
  {
    // 64 channels x 28 dwords = 1792 items over 512 threads: all of a thread's loads leave before the first use
    // (unconditional, from clamped frame indices; a load under a branch is waited for on the spot)
    constexpr int NI = (STEM_CMAX * (STEM_XP / 4) + STEM_NT - 1) / STEM_NT;
    const int nitem = p.C * (STEM_XP / 4);
    float xv[NI][4];
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      const int g = min(tid + u * STEM_NT, nitem - 1);
      const int c = g / (STEM_XP / 4), i4 = g - c * (STEM_XP / 4);
      const float* row = p.x + ((size_t)b * p.C + c) * p.Tx;
#pragma unroll
      for (int k = 0; k < 4; ++k) xv[u][k] = row[min(max(tin0 + 4 * i4 + k, 0), p.Tx - 1)];
    }
    // (the window's loads are in flight while the statistics are combined)
    if (p.stats) {
      // mean / unbiased std of every mel bin over the utterance's valid frames from the tiles' float64 partial sums:
      // sum_t (x - m)^2 = sum_tiles [M2_i + n_i (mean_i - m)^2] with m the float32 mean the reference subtracts
      // (8 threads per bin, fixed order: the result does not depend on which work-group computes it)
      const int c = tid >> 3, j = tid & 7;
      const int n = lim_in;
      const double* st = p.stats + ((size_t)b * p.n_stat_tiles * p.C + min(c, p.C - 1)) * 2;
      constexpr int NC = 4;                                    // tiles per thread kept in registers (32 tiles = 5 s of audio)
      double2 sv[NC];
#pragma unroll
      for (int q = 0; q < NC; ++q) {
        const int i = j + 8 * q;
        sv[q] = *(const double2*)(st + (size_t)min(i, p.n_stat_tiles - 1) * p.C * 2);
        if (i >= p.n_stat_tiles) sv[q] = make_double2(0.0, 0.0);
      }
      double sum = 0.0;
#pragma unroll
      for (int q = 0; q < NC; ++q) sum += sv[q].x;
      for (int i = j + 8 * NC; i < p.n_stat_tiles; i += 8) sum += st[(size_t)i * p.C * 2];
      sum += __shfl_xor(sum, 1);
      sum += __shfl_xor(sum, 2);
      sum += __shfl_xor(sum, 4);
      const float mean = (float)(sum / (double)n);
      double m2 = 0.0;
      auto tile_term = [&](int i, double s_i, double m2_i) {
        const int ni = max(0, min(n - QASR_MEL_TILE * i, QASR_MEL_TILE));
        if (ni > 0) {
          const double d = s_i / (double)ni - (double)mean;
          m2 += m2_i + (double)ni * d * d;
        }
      };
#pragma unroll
      for (int q = 0; q < NC; ++q) tile_term(j + 8 * q, sv[q].x, sv[q].y);
      for (int i = j + 8 * NC; i < p.n_stat_tiles; i += 8) tile_term(i, st[(size_t)i * p.C * 2], st[(size_t)i * p.C * 2 + 1]);
      m2 += __shfl_xor(m2, 1);
      m2 += __shfl_xor(m2, 2);
      m2 += __shfl_xor(m2, 4);
      if (j == 0 && c < p.C) {
        s_mean[c] = mean;
        s_sd[c] = (float)sqrt(m2 / (double)(n - 1)) + 1e-5f;     // torch.std (unbiased) + CONSTANT (features.py:63-65)
      }
      __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      const int g = tid + u * STEM_NT;
      if (g < nitem) {
        const int c = g / (STEM_XP / 4), i4 = g - c * (STEM_XP / 4);
        int v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int ti = tin0 + 4 * i4 + k;
          float xk = (ti >= 0 && ti < lim_in) ? xv[u][k] : 0.0f;            // zero padding and MaskedConv1d's mask
          if (p.stats && ti >= 0 && ti < lim_in) xk = __fdiv_rn(__fsub_rn(xk, s_mean[c]), s_sd[c]);
          v[k] = (int)fminf(fmaxf(rintf(__fmul_rn(p.inv_scale, xk)), (float)p.qlo), (float)p.qhi);
        }
        *(unsigned*)(xq + c * STEM_XP + 4 * i4) = pack4(v[0], v[1], v[2], v[3]);
      }
    }
  }
  for (int g = tid; g < 128 * 32 / 16; g += STEM_NT) *(v4i*)(xd + 16 * g) = (v4i){0, 0, 0, 0};   // K rows >= C stay zero
  __syncthreads();
  // ---- depthwise, stride 2: thread = channel c, output frames f0 .. f0 + 3 (window bytes 2 f0 .. 2 f0 + 41)
  if (tid < 8 * p.C) {
    const int c = tid >> 3, f0 = 4 * (tid & 7);
    const unsigned* xr = (const unsigned*)(xq + c * STEM_XP + 2 * f0);      // 8-byte aligned
    unsigned xw[STEM_KP4 + 2];
#pragma unroll
    for (int i = 0; i < STEM_KP4 + 2; ++i) xw[i] = xr[i];
    const int* wr = (const int*)(p.wdw + (size_t)c * (4 * STEM_KP4));
    int wk[STEM_KP4];
#pragma unroll
    for (int j = 0; j < STEM_KP4; ++j) wk[j] = wr[j];
    const int bias = p.bias_dw[c];
    const double M = p.m_dw[c];
    const float sb = (p.flags_dw & QASR_F_EXACT_Z) ? p.sb_dw[c] : 1.0f;
    const int lim_mid = (p.flags_dw & QASR_F_MASK_OUT) ? min(len_mid, p.T_mid) : p.T_mid;
    int q[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int sh = (2 * s) & 3, q0 = (2 * s) >> 2;
      int a = bias;
#pragma unroll
      for (int j = 0; j < STEM_KP4; ++j) {
        const unsigned d = sh ? __builtin_amdgcn_alignbyte(xw[q0 + j + 1], xw[q0 + j], sh) : xw[q0 + j];
        a = __builtin_amdgcn_sdot4((int)d, wk[j], a, false);
      }
      const int t = t0 + f0 + s;
      if (p.dw_acc_dbg && t < p.T_mid) p.dw_acc_dbg[((size_t)b * p.C + c) * e.Tp + t] = a;
      int z = a;
      if (p.flags_dw & QASR_F_EXACT_Z) z = z_roundtrip(a, sb, p.flags_dw & QASR_F_RELU);
      else if (p.flags_dw & QASR_F_RELU) z = max(a, 0);
      q[s] = t < lim_mid ? requant_clamp(z, M, p.dw_lo, p.dw_hi) : 0;
    }
    *(unsigned*)(xd + c * 32 + f0) = pack4(q[0], q[1], q[2], q[3]);
  }
  __syncthreads();
  // ---- 1x1 GEMM: wave w = output channels [256 ps + 32 w, +32) x 32 frames, K = cin_pad
  const lds_u8* const xd_lane = (const lds_u8*)xd + sep2_a_lane_off(lane);
  const int nks = p.cin_pad >> 5;
  const int len_out = len_mid;                                              // the 1x1 conv keeps the length
  const int lim = (e.flags & QASR_F_MASK_OUT) ? min(len_out, e.T) : e.T;
  const bool f_relu = e.flags & QASR_F_RELU;
  for (int ps = 0; ps < (e.cout + 255) / 256; ++ps) {
    const int co = 256 * ps + 32 * wave + (lane & 31);
    const bool co_ok = co < e.cout;
    const int cor = co_ok ? co : 0;
    v16i acc;
    const int bias = p.bias[cor];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bias;
    for (int ks = 0; ks < nks; ++ks) {
      const v4i a = sep2_a_frag(xd_lane, ks);
      const v4i wf = *w_frag(p.w, p.cin_pad, cor, ks);
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, wf, acc, 0, 0, 0);
    }
    const float sb = (e.flags & QASR_F_EXACT_Z) ? e.sb[cor] : 1.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = t0 + mfma32_row(r, h);
      if (e.acc_dbg && co_ok && t < e.T) e.acc_dbg[((size_t)b * e.cout + co) * e.Tp + t] = acc[r];
      int z = acc[r];
      if (e.flags & QASR_F_EXACT_Z) z = z_roundtrip(z, sb, f_relu);
      else if (f_relu) z = max(z, 0);
      acc[r] = t < lim ? z : 0;                                             // 0 requantises to 0 for every consumer
    }
    for (int j = 0; j < e.n_outs; ++j) {
      const OutP& o = e.outs[j];
      const double Mo = o.mtab[cor];
      int q[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) q[r] = requant_clamp(acc[r], Mo, o.lo, o.hi);
      // MFMA C layout: register 4 g + i of lane (c, h) = frame 8 g + 4 h + i; after the two half-wave swaps the lower
      // half-wave holds frames 0..15 and the upper one frames 16..31 of its channel
      unsigned P[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) P[g] = pack4b(q[4 * g], q[4 * g + 1], q[4 * g + 2], q[4 * g + 3]);
      const auto s02 = __builtin_amdgcn_permlane32_swap(P[0], P[2], false, false);
      const auto s13 = __builtin_amdgcn_permlane32_swap(P[1], P[3], false, false);
      const v4i pk = {(int)s02[0], (int)s02[1], (int)s13[0], (int)s13[1]};
      if (co_ok) *(v4i*)((int8_t*)o.ptr + ((size_t)b * e.cout + co) * e.Tp + t0 + 16 * h) = pk;
    }
  }
}

bool stem_supported(const QuantInP& qi, const DwP& dw, const SepP& pw) {
  const EpiP& e = pw.e;
  if (dw.stride != 2 || dw.dilation != 1 || dw.K > 4 * STEM_KP4 || dw.K - 1 + 2 * (STEM_TT - 1) + 4 > STEM_XP || dw.C > STEM_CMAX ||
      dw.kpad != 4 * STEM_KP4 || dw.x_unsigned || dw.e.n_outs != 1 || dw.e.outs[0].mode != 1 || pw.K != 0 || pw.n_panes != 0 ||
      pw.cin_pad != 128 || pw.pw_unsigned || pw.cin != dw.C || (e.flags & (QASR_F_LOGITS | QASR_F_RESADD)) || e.n_outs < 1 ||
      e.Tp % STEM_TT || qi.C != dw.C || e.B < 1 || e.B > 65535)
    return false;
  for (int j = 0; j < e.n_outs; ++j)
    if (e.outs[j].mode != 1) return false;
  return true;
}

int launch_stem(hipStream_t s, const QuantInP& qi, const DwP& dw, const SepP& pw, const qasr_domain_desc* doms, int n_domains,
                const int32_t* lens_in, int32_t* lens_all, const double* stats, int n_stat_tiles, int n_frames) {
  const EpiP& e = pw.e;
  if (!stem_supported(qi, dw, pw) || !qi.x || !lens_in || !lens_all || !doms) return QASR_ERR_UNSUPPORTED;
  StemP p{};
  p.x = qi.x, p.lens_in = lens_in, p.inv_scale = qi.inv_scale, p.qlo = qi.lo, p.qhi = qi.hi, p.C = qi.C, p.Tx = qi.T;
  p.wdw = dw.w, p.bias_dw = dw.bias, p.m_dw = dw.e.outs[0].mtab, p.sb_dw = dw.e.sb, p.flags_dw = dw.e.flags;
  p.dw_lo = dw.e.outs[0].lo, p.dw_hi = dw.e.outs[0].hi, p.K = dw.K, p.padding = dw.padding, p.T_mid = dw.e.T;
  p.dw_acc_dbg = dw.e.acc_dbg;
  p.w = pw.w, p.bias = pw.bias, p.cin_pad = pw.cin_pad;
  p.e = e;
  p.doms = doms, p.n_domains = n_domains, p.lens_all = lens_all;
  p.stats = stats, p.n_stat_tiles = n_stat_tiles, p.n_frames = n_frames;
  if (stats && (n_stat_tiles < 1 || n_frames < 1 || n_frames > qi.T)) return QASR_ERR_ARG;
  hipLaunchKernelGGL(k_stem, dim3(e.B, e.Tp / STEM_TT), dim3(STEM_NT), 0, s, p);
  return QASR_OK;
}

}  // namespace qasr
