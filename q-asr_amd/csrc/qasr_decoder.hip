// Fused CTC decoder: ConvASRDecoder.forward (nemo/collections/asr/modules/conv_asr.py:270-275) after its QuantAct -
// the 1x1 QuantConv1d with bias (1024 -> 29, quant_modules.py:301-309), log_softmax over the classes and the greedy
// argmax of EncDecCTCModel.forward (ctc_models.py:405) - as ONE launch.  The generic path ran the conv on the separable
// layer kernel with the 29 output channels padded to 128 (4x the matrix work and weight bytes), wrote float32 logits
// [B][T][29] with 4-byte scattered stores (40 MB of HBM writes per step for 0.93 MB of logits) and read them back in a
// second launch (k_logsoftmax).
//
// Work-group = one utterance x 32 frames, 8 waves.  The int8 tile [cin][32 frames] is copied to LDS in the
// [channel][frame] image k_sep2 uses, so the MFMA A fragments come from the same transposing reads
// (ds_read_b64_tr_b8); the K loop (cin / 32 steps) is split over the 8 waves, each accumulating a 32 x 32 partial product
// with v_mfma_i32_32x32x32_i8; the partials are added through LDS in wave order (integers: exact), scaled to float32
// logits exactly as the generic path does (fl32(fl32(acc) * s_b[c])), and one lane per frame runs the same sequential
// max / sum-exp / log as k_logsoftmax, so log-probabilities are bit-identical to the two-launch path.
#include "qasr_sep2_impl.h"

namespace qasr {

struct DecP {
  const int8_t* x;          // [B][cin][Tp] codes of the decoder's QuantAct
  const int8_t* w;          // fragment-ordered [cout_pad128][cin_pad]
  const int32_t* bias;      // [cout_pad128] (+ 128 sum(W) for u8 codes)
  const float* sb;          // [cout_pad128] conv output scales
  const int32_t* lens;      // [B] encoded lengths (copied to lens_out)
  int32_t* lens_out;        // optional
  int32_t* acc_dbg;         // optional i32 [B][cout][Tp]
  float* logits;            // optional f32 [B][T][cout]
  float* logp;              // optional f32 [B][T][ncls]
  int32_t* tokens;          // optional i32 [B][T]
  int cin, cin_pad, x_unsigned, B, T, Tp, ncls;
};

#define DEC_NT 512
#define DEC_TT 32

__global__ void __launch_bounds__(DEC_NT) k_dec(DecP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x, t0 = blockIdx.y * DEC_TT;
  lds_u8* const Xd = (lds_u8*)smem;                                       // [cin_pad][32]
  int* const red = (int*)(smem + (size_t)p.cin_pad * 32);                 // [8][32 frames][33]
  float* const lg = (float*)(red + 8 * 32 * 33);                          // [32 frames][33]
  if (blockIdx.y == 0 && tid == 0 && p.lens_out) p.lens_out[b] = p.lens[b];
  // ---- the tile's codes: two 16-byte granules (16 frames each) per channel row; rows >= cin are zero
  const unsigned flip = p.x_unsigned ? 0x80808080u : 0u;                  // u8 codes are fed as x - 128 (bias carries 128 sum(W))
  for (int base = tid; base < 2 * p.cin_pad; base += 4 * DEC_NT) {        // four granules per thread in flight together
    v4i r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int g = min(base + u * DEC_NT, 2 * p.cin_pad - 1), c = min(g >> 1, p.cin - 1), half = g & 1;
      r[u] = *(const v4i*)(p.x + ((size_t)b * p.cin + c) * p.Tp + t0 + 16 * half);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int g = base + u * DEC_NT;
      if (g < 2 * p.cin_pad) {
        const int c = g >> 1, half = g & 1;
        v4i v = r[u];
        v[0] ^= flip, v[1] ^= flip, v[2] ^= flip, v[3] ^= flip;
        if (c >= p.cin) v = (v4i){0, 0, 0, 0};
        *(lds_v4i*)(Xd + c * 32 + 16 * half) = v;
      }
    }
  }
  __syncthreads();
  // ---- this wave's share of the K loop
  const int nks = p.cin_pad >> 5, per = nks >> 3;                         // launch_decoder: cin_pad % 256 == 0
  const lds_u8* const xd_lane = Xd + sep2_a_lane_off(lane);
  v16i acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0;
  for (int i = 0; i < per; ++i) {
    const int ks = wave * per + i;
    const v4i a = sep2_a_frag(xd_lane, ks);
    const v4i wf = *w_frag(p.w, p.cin_pad, 0, ks);
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, wf, acc, 0, 0, 0);
  }
  // C layout: lane & 31 = output channel, register r of half h = frame mfma32_row(r, h)
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 32 + mfma32_row(r, h)) * 33 + (lane & 31)] = acc[r];
  __syncthreads();
  // ---- add the partials (wave order), bias, scale: 32 x 32 values over 512 threads
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int f = 16 * pass + (tid >> 5), c = tid & 31;
    int a = p.bias[c];
#pragma unroll
    for (int w = 0; w < 8; ++w) a += red[(w * 32 + f) * 33 + c];
    const int t = t0 + f;
    if (c < p.ncls && t < p.T) {
      if (p.acc_dbg) p.acc_dbg[((size_t)b * p.ncls + c) * p.Tp + t] = a;
      const float z = mul_f32_unfused((float)a, p.sb[c]);                // conv_int.float() * scale (quant_modules.py:305-308)
      lg[f * 33 + c] = z;
      if (p.logits) p.logits[((size_t)b * p.T + t) * p.ncls + c] = z;
    }
  }
  __syncthreads();
  // ---- log_softmax + argmax, one lane per frame, the operation order of k_logsoftmax
  if (tid < DEC_TT && t0 + tid < p.T) {
    const float* x = lg + tid * 33;
    float m = x[0];
    int am = 0;
    for (int c = 1; c < p.ncls; ++c) {
      const float v = x[c];
      if (v > m) { m = v; am = c; }                                        // first maximum wins, like torch.argmax
    }
    float s = 0.f;
    for (int c = 0; c < p.ncls; ++c) s += expf(x[c] - m);
    const float ls = logf(s);
    const size_t row = (size_t)b * p.T + t0 + tid;
    if (p.logp)
      for (int c = 0; c < p.ncls; ++c) p.logp[row * p.ncls + c] = (x[c] - m) - ls;
    if (p.tokens) p.tokens[row] = am;
  }
}

// The decoder op of the plan (a K == 0 SepP with QASR_F_LOGITS) as one fused launch; false: shape outside the kernel
bool decoder_fusable(const SepP& p) {
  return p.K == 0 && p.dense_k <= 1 && p.n_panes == 0 && (p.e.flags & QASR_F_LOGITS) && p.e.cout <= 32 && p.cin_pad % 256 == 0 &&
         p.cin_pad <= 2048 && p.e.Tp % DEC_TT == 0;
}

int launch_decoder(hipStream_t s, const SepP& q, float* logp, int32_t* tokens, int32_t* lens_out, bool keep_logits) {
  if (!decoder_fusable(q) || !q.x || !q.w || !q.bias || !q.e.sb || !q.e.lens || q.e.B < 1 || q.e.T > q.e.Tp) return QASR_ERR_ARG;
  DecP p{};
  p.x = q.x, p.w = q.w, p.bias = q.bias, p.sb = q.e.sb, p.lens = q.e.lens, p.lens_out = lens_out;
  p.acc_dbg = q.e.acc_dbg;
  p.logits = keep_logits ? q.e.logits : nullptr;
  p.logp = logp, p.tokens = tokens;
  p.cin = q.cin, p.cin_pad = q.cin_pad, p.x_unsigned = q.pw_unsigned;
  p.B = q.e.B, p.T = q.e.T, p.Tp = q.e.Tp, p.ncls = q.e.cout;
  const size_t smem = (size_t)p.cin_pad * 32 + sizeof(int) * 8 * 32 * 33 + sizeof(float) * 32 * 33;
  static int attr_dev = -1;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (attr_dev != dev) {
    (void)hipFuncSetAttribute((const void*)k_dec, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_dev = dev;
  }
  hipLaunchKernelGGL(k_dec, dim3(p.B, p.Tp / DEC_TT), dim3(DEC_NT), smem, s, p);
  return QASR_OK;
}

}  // namespace qasr
