// Internal parameter blocks shared by the engine (qasr_engine.hip) and the kernels (qasr_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/qasr.h"

namespace qasr {

// One consumer of an op's integer result (device view of qasr_out).
struct OutP {
  void* ptr;            // i8 [B][C][Tp]  (mode 3: i32 [B][C][Tp])
  const double* mtab;   // mode 1: per-channel multipliers
  double m;             // mode 0: scalar multiplier
  int lo, hi, mode;
  int pad_;
};

struct PaneP {          // one residual 1x1 conv of a RESADD op
  const int8_t* x;      // [B][cin][Tp]
  const int8_t* w;      // [cout_pad][cin_pad]
  const int32_t* bias;  // [cout_pad]
  const double* m;      // [cout_pad]
  const float* sb;      // [cout_pad]
  int32_t* acc_dbg;     // optional [B][cout][Tp]
  int cin, cin_pad, x_unsigned, pad_;
};

struct EpiP {
  OutP outs[QASR_MAX_OUTS];
  int n_outs;
  unsigned flags;
  const float* sb;        // conv output scale (EXACT_Z, LOGITS)
  const double* m_main;   // RESADD
  const int32_t* lens;    // [B] valid frames in the OUTPUT domain
  int32_t* acc_dbg;       // optional i32 [B][cout][Tp]
  float* logits;          // LOGITS: f32 [B][T][cout]
  int qlo, qhi;
  int T, Tp;              // output frames / row pitch
  int cout, B;
};

struct DwP {              // depthwise conv
  const int8_t* x;        // [B][C][Tp_in]
  const int8_t* w;        // [C][kpad]
  const int32_t* bias;    // [C_pad] (128*sum(w) for u8 inputs, else 0)
  int C, K, kpad, stride, dilation, padding;
  int T_in, Tp_in, x_unsigned, pad_;
  EpiP e;
};

struct DenseP {           // dense k>1 conv as implicit GEMM (+ residual panes)
  const int8_t* x;        // [B][cin][Tp_in]
  const int8_t* w;        // [cout_pad][K][cin_pad]
  const int32_t* bias;
  int cin, cin_pad, K, stride, dilation, padding;
  int T_in, Tp_in, x_unsigned, n_panes;
  PaneP panes[QASR_MAX_PANES];
  EpiP e;
};

struct SepP {             // fused separable layer: depthwise stencil -> QuantAct -> 1x1 GEMM (+ panes) -> epilogue
  // depthwise stage (absent when K == 0: the 1x1 conv reads `x` directly)
  const int8_t* x;        // [B][cin][Tp]   input of the depthwise conv (or of the 1x1 conv when K == 0)
  const int8_t* wdw;      // [cin][kpad4]
  const int8_t* wdw2;     // [cin][kpad4 + 32]: the same taps behind 8 zero bytes and followed by zeros (MFMA depthwise)
  const int32_t* bias_dw; // [cin_pad]  128*sum(w) for u8 inputs
  const double* m_dw;     // [cin_pad]  requant of the dw accumulator towards the 1x1 conv's QuantAct
  int32_t* dw_acc_dbg;    // optional i32 [B][cin][Tp]
  int dw_lo, dw_hi;       // clamp of that QuantAct
  int K, x_unsigned;      // taps (stride 1, 'same' padding)
  int dilation, tile;     // dilation 1, or 2 (window staged de-interleaved by parity); tile: frames per work-group, 32 or 64
  // pointwise stage
  const int8_t* w;        // [cout_pad][cin_pad]
  const int32_t* bias;    // [cout_pad]
  int cin, cin_pad, pw_unsigned, n_panes;
  int dense_k, gen;       // dense_k > 1: dense conv with that many taps (K == 0 kernels; `w` tap-major, `dilation` = tap spacing);
                          // gen: 2 = route to k_sep2 where it has the shape (engine default), else k_sep
  long long* prof;        // diagnostics: s_memtime stamps of work-group (0,0,0), wave 0 (qasr_debug_prof)
  int prof_mode;          // 1 (qasr_debug_timeline): every work-group writes {start, end (s_memrealtime, 100 MHz), HW_ID | XCC_ID << 32, shader cycles}
  int prof_cap;           // ... for work-groups below this count (the caller's buffer)
  int etile;              // the engine's tile_frames option (32: built for one step in flight; 64 / 128: several), before per-op adjustments
  PaneP panes[QASR_MAX_PANES];
  EpiP e;
};

// qasr_dense2.hip: Jasper's plain dense convs (no residual panes) on 128-channel x 256-frame work-groups
bool dense2_takes(const SepP& p);
int launch_dense2(hipStream_t s, const SepP& p);
void dense2_label(const SepP& p, char* buf, size_t cap);

extern long long* g_prof;
extern int g_prof_mode;
extern int g_prof_cap;      // timeline mode: work-groups the buffer holds (4 int64 each)

struct QuantInP {
  const float* x;         // [B][C][T]
  int8_t* out;            // [B][C][Tp]
  const int32_t* lens;
  float inv_scale;
  int lo, hi, C, T, Tp, B;
};

#define QASR_RQ_MAX 8      /* consumers one k_requant launch serves (Jasper's dense residual: up to 11 per stored value) */
struct RequantP {         // stand-alone requant of a stored value towards n_outs consumers (read once)
  const void* in;         // i32 or s8 [B][C][Tp]
  int in_is_i32, n_outs;
  OutP outs[QASR_RQ_MAX];
  const float* sb;
  const int32_t* lens;
  unsigned flags;
  int C, T, Tp, B;
};

void launch_quant_in(hipStream_t s, const QuantInP& p);
void launch_dw(hipStream_t s, const DwP& p);
void launch_dense(hipStream_t s, const DenseP& p);
bool sep_supported(int K, int dilation);
int launch_sep(hipStream_t s, const SepP& p);      // QASR_OK, or QASR_ERR_UNSUPPORTED / QASR_ERR_ARG without launching
void sep_kernel_label(const SepP& p, char* buf, size_t cap);
int sep_tile_for(const SepP& p);
void launch_requant(hipStream_t s, const RequantP& p);
void launch_logsoftmax(hipStream_t s, const float* logits, float* logp, int32_t* tokens, int rows, int ncls);
// qasr_stem.hip: lengths + first-layer QuantAct + strided depthwise conv + 1x1 conv of block 0 as one launch
bool stem_supported(const QuantInP& qi, const DwP& dw, const SepP& pw);
// stats != nullptr: x holds un-normalised log-mel and normalize_batch runs inside (per-tile sums from frontend_mel_stats)
int launch_stem(hipStream_t s, const QuantInP& qi, const DwP& dw, const SepP& pw, const qasr_domain_desc* doms, int n_domains,
                const int32_t* lens_in, int32_t* lens_all, const double* stats = nullptr, int n_stat_tiles = 0, int n_frames = 0);
// qasr_frontend.hip: k_mel alone (un-normalised log-mel, feat_lens, per-tile statistics [B][tiles][n_mels][2] f64)
#define QASR_MEL_TILE 16
size_t frontend_stats_bytes(int B, int S, int n_mels);
int frontend_mel_stats(hipStream_t s, const float* audio, const int32_t* audio_lens, int B, int S, const float* fb,
                       const float* window, int n_mels, float preemph, int pad_to, float* feats, int32_t* feat_lens,
                       const void* workspace, size_t workspace_bytes, double* stats, int* n_tiles, int* n_frames);
// qasr_decoder.hip: the decoder's 1x1 conv + log_softmax + argmax (+ the encoded lengths) as one launch
bool decoder_fusable(const SepP& p);
int launch_decoder(hipStream_t s, const SepP& p, float* logp, int32_t* tokens, int32_t* lens_out, bool keep_logits);
void launch_lens(hipStream_t s, const int32_t* lens_in, int32_t* lens_all, const qasr_domain_desc* doms,
                 int n_domains, int B);

}  // namespace qasr
