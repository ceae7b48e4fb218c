/*
 * qasr.h — C ABI of the MI355X integer-only ASR engine (libqasr_hip.so).
 *
 * The reference (kssteven418/Q-ASR) has no FFI: its boundary for this path is the Python
 * nn.Module API (QuantAct / QuantConv1d / ConvASREncoder / ConvASRDecoder / EncDecCTCModel).
 * This header is the native surface that sits directly under that API; each
 * entry point names the reference interface it replaces.  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions: plain pointers and sizes only; the caller owns every device buffer it passes;
 * the engine owns its packed blob copy and scratch arena; every call returns an int status
 * (QASR_OK == 0) and never throws; one engine per device, not thread-safe; work is enqueued on
 * the caller's HIP stream (passed as void*) with no host synchronisation unless stated.
 */
#ifndef QASR_H
#define QASR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QASR_OK 0
#define QASR_ERR_ARG 1        /* bad argument / shape */
#define QASR_ERR_BLOB 2       /* malformed packed model */
#define QASR_ERR_HIP 3        /* HIP runtime error (see qasr_last_error) */
#define QASR_ERR_UNSUPPORTED 4

#define QASR_BLOB_MAGIC 0x52534151u /* "QASR" */
#define QASR_BLOB_VERSION 6u

/* ---- packed model ("blob") layout, produced by qasr/pack.py --------------------------------
 * header | tensor table | op table | data (int8 weights, int32 biases, f64 requant multipliers,
 * f32 scales).  All offsets are bytes from the start of the blob; data items are 16-B aligned. */
typedef struct qasr_blob_header {
  uint32_t magic, version;
  uint32_t n_tensors, n_ops;
  uint32_t feat_in;          /* mel bins entering the encoder */
  uint32_t n_classes;        /* decoder outputs incl. blank */
  uint32_t weight_bit, act_bit;
  uint32_t n_domains, reserved;
  uint64_t tensors_off, ops_off, domains_off, data_off, total_bytes;
} qasr_blob_header;

/* A time domain: every tensor of a domain has the same frame count T and per-utterance lengths.
 * Domain 0 is the encoder input; a strided conv opens a new one with
 * len' = (len + 2*padding - dilation*(kernel-1) - 1)/stride + 1   (MaskedConv1d.get_seq_len, jasper.py:170-173). */
typedef struct qasr_domain_desc {
  int32_t parent;              /* -1 for domain 0 */
  uint32_t kernel, stride, dilation, padding;
  uint32_t reserved[3];
} qasr_domain_desc;

/* dtype of an activation tensor stored as [B][C][Tp] (time contiguous, Tp = T rounded up to 64) */
enum { QASR_DT_S8 = 0, QASR_DT_U8 = 1, QASR_DT_F32 = 2, QASR_DT_I32 = 3 };

typedef struct qasr_tensor_desc {
  uint32_t channels;
  uint32_t dtype;
  uint32_t domain;
  uint32_t reserved;
  int32_t producer;          /* op index that writes it, -1 for the network input */
  int32_t last_use;          /* last op index that reads it */
} qasr_tensor_desc;

enum {
  QASR_OP_QUANT_IN = 0,      /* QuantAct first layer: f32 features -> s8 (quant_modules.py:180-184) */
  QASR_OP_DW = 1,            /* depthwise QuantConv1d + next QuantAct requant (jasper.py:569-583) */
  QASR_OP_PW = 2,            /* 1x1 QuantConv1d (+ residual 1x1 convs + res_act) + next QuantAct requant */
  QASR_OP_DENSE = 3,         /* dense k>1 QuantConv1d (Jasper) (+ residual panes) */
  QASR_OP_LOGSOFTMAX = 4,    /* log_softmax + argmax over classes (conv_asr.py:275, ctc_models.py:405) */
  QASR_OP_REQUANT = 5        /* stand-alone QuantAct requant of a stored s8 / i32 value (extra consumers) */
};

enum {
  QASR_F_RELU = 1u << 0,        /* ReLU on the conv output before requantisation (jasper.py:661,687) */
  QASR_F_MASK_OUT = 1u << 1,    /* write 0 for t >= len[b]: the consumer's MaskedConv1d mask (jasper.py:177-181) */
  QASR_F_EXACT_Z = 1u << 2,     /* |acc| may reach 2^22: take z through the float32 round trip (quant_utils.py:187) */
  QASR_F_LOGITS = 1u << 3,      /* decoder: emit f32 logits = fl32(fl32(acc)*s_b) as [B][T][C] */
  QASR_F_RESADD = 1u << 4,      /* res_act: q = clamp(rq(main) + rq(pane) ...) sequentially over panes */
  QASR_F_TAPMAJOR = 1u << 5,    /* DENSE op (stride 1, 'same' padding): weights stored as one MFMA-fragment-ordered
                                   [cout_pad][cin_pad] matrix per tap; runs on the k_sep tile kernel */
  QASR_F_WIDE_RQ = 1u << 6,     /* some |acc * m| of this op may reach 2^30 (a QuantAct calibrated on near-silence): the
                                   requantisation must clamp in the double domain (k_sep), not on the low word (k_sep2) */
  QASR_F_W6PACK = 1u << 7       /* weights of this op and of its panes are stored sub-byte (weight_bit <= 6): 4 two's-
                                   complement 6-bit codes per 3 bytes (b0 = c0 | c1 << 6, b1 = c1 >> 2 | c2 << 4,
                                   b2 = c2 >> 4 | c3 << 2), element order unchanged; DW ops then carry no zero-margined
                                   tap array (m_off = 0).  qasr_engine_create expands them to int8 on the device */
};

#define QASR_MAX_PANES 12
#define QASR_MAX_OUTS 3

typedef struct qasr_pane {     /* one residual 1x1 conv feeding res_act (jasper.py:664-682) */
  int32_t in;                  /* tensor id (u8, already requantised for this conv's QuantAct) */
  uint32_t cin;
  uint64_t w_off;              /* s8, MFMA fragment order like the main 1x1 weights */
  uint64_t bias_off;           /* i32 [cout], includes the +128*sum(W) correction for u8 inputs */
  uint64_t m_off;              /* f64 [cout]: s_b[c] / S  as m*2^-e (batch_frexp) */
  uint64_t sb_off;             /* f32 [cout]: conv output scale (only read with QASR_F_EXACT_Z) */
} qasr_pane;

typedef struct qasr_out {      /* one consumer of the op's integer result */
  int32_t tensor;              /* tensor id, -1 = unused */
  int32_t lo, hi;              /* clamp range of the consumer's QuantAct */
  uint32_t mode;               /* 0: scalar multiplier `m`; 1: per-channel table at m_off (f64[cout]);
                                  2: identity copy of the integer result; 3: raw int32 accumulator */
  uint64_t m_off;
  double m;
} qasr_out;

typedef struct qasr_op_desc {
  uint32_t kind, flags;
  int32_t in;                  /* main input tensor */
  uint32_t cin, cout, kernel, stride, dilation, padding;
  uint32_t n_panes;
  uint64_t w_off;              /* s8 weights: DW [c][kpad4]; PW cout_pad x cin_pad in MFMA fragment order (see
                                  qasr_pw_conv_acc); DENSE [cout_pad][k][cin_pad] */
  uint64_t bias_off;           /* i32 [cout] (0 = none) */
  uint64_t m_off;              /* RESADD: f64 [cout] multiplier of the main accumulator towards S;
                                  DW ops: s8 [cout][kpad4 + 32], the taps again behind 8 zero bytes and followed by zeros
                                  (per-lane pre-shifted tap streams of the MFMA depthwise stage) */
  uint64_t sb_off;             /* f32 [cout] conv output scale s_w[c]*s_x */
  int32_t qlo, qhi;            /* RESADD clamp of res_act */
  float in_inv_scale;          /* QUANT_IN: fl32(1/s) ; int32 range in qlo/qhi */
  uint32_t reserved;
  qasr_out outs[QASR_MAX_OUTS];
  qasr_pane panes[QASR_MAX_PANES];
} qasr_op_desc;

/* Host-only validation of a packed model (no GPU is touched): QASR_OK, or QASR_ERR_BLOB with a message in `err` (may be
 * NULL) when the header, a table entry or any data offset + extent an op refers to falls outside the blob, an index is
 * out of range, an op's output mode does not match its tensor's element size, or operand time domains disagree.  The
 * reference's loader is unchecked (nemo/core/classes/modelPT.py:379-400); here a blob may have travelled over RCCL
 * (qasr/dist.py) and every field ends up in a kernel argument.  qasr_engine_create[_ex] runs it first. */
int qasr_blob_check(const void* blob, size_t blob_bytes, char* err, size_t err_cap);

/* ---- engine -------------------------------------------------------------------------------- */
typedef struct qasr_engine qasr_engine;

/* Build an engine from a packed model.  Replaces model construction + `qm.evaluate(model)` state:
 * EncDecCTCModel.__init__ / encoder.bn_folding / calibrated QuantAct ranges
 * (nemo/collections/asr/models/ctc_models.py:91-147, examples/asr/quantization/inference.py:105-136).
 * `debug` bit 0 keeps every intermediate tensor and int32 accumulator alive for qasr_engine_read_*;
 * bit 1 only records one HIP event per op on the launch stream (qasr_engine_last_op_ms), no other change;
 * bit 2 (round 1's whole-utterance kernels k_utt) is retired: refused with QASR_ERR_UNSUPPORTED;
 * bit 3 makes k_sep use 64-frame tiles (half the weight / halo traffic per frame and half as many work-groups per
 * launch: faster when several steps are in flight on separate streams, slower for a single step);
 * bit 4 replays the forward as a hipGraph: the second forward with the same shape and the same five buffer pointers is
 * captured, later ones are one hipGraphLaunch (a new pointer set or shape starts over; ignored with bits 0/1).
 * This entry keeps qasr_engine_opts.fuse_norm = 0: qasr_engine_forward_audio leaves the normalised log-mel in `feats`, as in
 * rounds 1 / 2 (the fused normalisation is opt-in through qasr_engine_create_ex, whose default has it on). */
int qasr_engine_create(const void* blob, size_t blob_bytes, int device, int debug, qasr_engine** out);
void qasr_engine_destroy(qasr_engine* e);

/* The same with every launch-plan choice spelled out (what the `debug` bits and, before round 3, environment variables
 * selected).  Tri-state fields: -1 = the engine's default, 0 = off, 1 = on.  `struct_size` = sizeof(qasr_engine_opts) of
 * the caller's header (a shorter, older struct is accepted: missing fields take their defaults; a longer one is refused).
 * Two engines in one process are configured independently of each other through this call.
 * Environment variables remain ONLY as A/B overrides for profiling runs of an unmodified caller, read once per create
 * call AFTER the options: QASR_TILE128=0|1 (128-frame tiles when tile_frames >= 64), QASR_RES_TILE128, QASR_DENSE_TILE128,
 * QASR_SEP_GEN=1|2, QASR_NO_FUSE, QASR_NO_FUSE_STEM, QASR_NO_FUSE_DEC, QASR_WIDE_TILES, QASR_NO_FUSE_NORM; QASR_SEP2_TUNE is a kernel-internal experiment knob (csrc/qasr_sep2_impl.h). */
typedef struct qasr_engine_opts {
  uint32_t struct_size;
  uint32_t debug;              /* bit 0: keep every tensor + int32 accumulators (parity hooks); bit 1: one HIP event per op */
  int32_t tile_frames;         /* frames per work-group of the separable-layer kernels: 0 (= 32), 32, 64 or 128.  32: one
                                  launch of a 32-utterance batch fills the chip (one step in flight); 128: cheapest frame,
                                  B * Tp / 128 work-groups per launch (several steps in flight) */
  int32_t sep_gen;             /* 0 / 2: k_sep2 where it has the shape; 1: round 1's k_sep everywhere */
  int32_t fuse_dw;             /* depthwise conv fused into the following 1x1 conv's launch */
  int32_t fuse_stem;           /* block 0 (lengths, first-layer QuantAct, strided depthwise, 1x1) as one launch */
  int32_t fuse_decoder;        /* decoder conv + log-softmax + argmax + encoded lengths as one launch */
  int32_t graph;               /* replay the forward as one hipGraph launch (second call with the same buffers captures) */
  int32_t retired_whole_utterance; /* round 1's k_utt kernels, removed in round 4: must be <= 0 */
  int32_t res_tile128;         /* block-end (residual) layers on 128-frame tiles too when tile_frames == 128 */
  int32_t dense_tile128;       /* Jasper's plain dense convs on 128-frame tiles when tile_frames >= 64 */
  int32_t retired_legacy_pw;   /* round 1's k_pw, removed in round 4: must be <= 0 */
  int32_t retired_persistent;  /* round 3's persistent per-utterance launch (built, bit-exact, measured slower: DESIGN.md closed
                                  routes), removed in round 4: must be <= 0.  The three retired fields keep the struct layout */
  int32_t fuse_norm;           /* qasr_engine_forward_audio with the fused block 0: normalize_batch folded into k_stem from
                                  per-tile sums k_mel writes (no k_norm launch; `feats` then holds the UN-normalised log-mel) */
  int32_t reserved[2];
} qasr_engine_opts;
/* fills `o` with struct_size and the defaults (-1 / 0) */
void qasr_engine_default_opts(qasr_engine_opts* o);
int qasr_engine_create_ex(const void* blob, size_t blob_bytes, int device, const qasr_engine_opts* opts, qasr_engine** out);

/* Encoder + decoder for one batch, replacing ConvASREncoder.forward + ConvASRDecoder.forward + argmax
 * (nemo/collections/asr/modules/conv_asr.py:194-206,270-275; ctc_models.py:403-405).
 * feats   device f32 [B][feat_in][T]   (the preprocessor's output, time contiguous)
 * lens    device i32 [B]               valid frames per utterance
 * logp    device f32 [B][T'][n_classes] log-probabilities (may be NULL)
 * tokens  device i32 [B][T']           greedy argmax (may be NULL)
 * lens_out device i32 [B]              encoded lengths (may be NULL)
 * T' = qasr_engine_out_frames(e, T). */
int qasr_engine_forward(qasr_engine* e, void* stream, const float* feats, const int32_t* lens, int B, int T,
                        float* logp, int32_t* tokens, int32_t* lens_out);
/* The same with the mel front-end in front (qasr_frontend_mel_planned into the caller's `feats` [B][n_mels][T_pad] /
 * `feat_lens` [B] buffers, T_pad = qasr_frontend_frames(S, pad_to)): AudioToMelSpectrogramPreprocessor + encoder +
 * decoder of EncDecCTCModel.forward (ctc_models.py:383-406) as ONE call - and, with graph replay on, one hipGraph launch
 * per batch.  `frontend_plan`: a workspace filled by qasr_frontend_plan for this filterbank.
 * `feats` is working storage of the call: with qasr_engine_opts.fuse_norm (default on where block 0 runs as k_stem) it holds
 * the log-mel BEFORE normalize_batch (features.py:53-67 runs inside k_stem); fuse_norm = 0 leaves the normalised features
 * qasr_frontend_mel returns. */
int qasr_engine_forward_audio(qasr_engine* e, void* stream, const float* audio, const int32_t* audio_lens, int B, int S,
                              const float* fb, const float* window, int n_mels, float preemph, int pad_to,
                              const void* frontend_plan, size_t plan_bytes, float* feats, int32_t* feat_lens, float* logp,
                              int32_t* tokens, int32_t* lens_out);
int qasr_engine_out_frames(const qasr_engine* e, int T);
int qasr_engine_num_ops(const qasr_engine* e);
/* kernel launches of the last forward with the current plan: encoder + decoder (80 for QuartzNet15x5 with the default
 * options) plus, after qasr_engine_forward_audio, the front-end's (k_mel; + k_norm when fuse_norm is
 * off: 81 / 82); -1 before the first forward */
int qasr_engine_num_launches(const qasr_engine* e);

/* Parity hooks (debug engines only; synchronise the stream).  acc: int32 [B][cout][T_out] = the
 * value rint(conv_int) of QuantConv1d.int_conv (quant_modules.py:304) for op `op` (pane < 0: main conv). */
int qasr_engine_read_acc(qasr_engine* e, int op, int pane, int32_t* host_out, size_t n_elems);
/* read_tensor also serves a production (non-debug) engine for a tensor whose arena slot no later tensor reused - e.g.
 * the decoder's input, the final encoder codes - and refuses the others.  Every engine refuses a tensor its launch plan
 * never stores (a depthwise output inside the fused layer's launch, k_stem's intermediates, the float logits inside k_dec). */
int qasr_engine_read_tensor(qasr_engine* e, int tensor, void* host_out, size_t n_bytes, int* T_out, int* Tp_out);
/* average device time (ms) per op kind over the last forward, measured with HIP events (debug engines) */
int qasr_engine_last_op_ms(qasr_engine* e, float* ms_per_op, int n_ops);
/* Roofline hook: after a forward, replay each op `reps` times back to back between one pair of HIP events on
 * `stream` and return the average duration per launch in ms (synchronises). */
int qasr_engine_time_ops(qasr_engine* e, void* stream, int reps, float* ms_per_launch, int n_ops);
/* Name of the kernel instantiation op `op` is routed to, with its template arguments as rocprofv3 prints them
 * ("k_sep<75, 1, 1, false>", "k_dw", ...); ops executed inside the next op's launch report "(fused ...)". */
int qasr_engine_op_label(qasr_engine* e, int op, char* buf, size_t cap);
/* Re-enqueue a single op of the last forward's plan (diagnostics / profiling of one layer). */
int qasr_engine_run_op(qasr_engine* e, void* stream, int op);

/* ---- calibration helper -------------------------------------------------------- */

/* The two torch.quantile calls of QuantAct's percentile calibration
 * (nemo/quantization/utils/quant_modules.py:121-125: x_min = quantile(x_act, 1 - p/100), x_max = quantile(x_act, p/100))
 * over a flattened float32 device tensor of n elements (16-byte aligned, finite values): exact order statistics by a
 * 4-pass radix select + torch's float32 linear interpolation.  out2[0] = quantile(q_lo), out2[1] = quantile(q_hi);
 * bit-identical to torch.quantile on the CPU.  Enqueued on `stream`; `workspace` (device, qasr_quantile_workspace_bytes()
 * bytes) must stay untouched until the stream has finished. */
size_t qasr_quantile_workspace_bytes(void);
int qasr_quantile2(void* stream, const float* x, size_t n, float q_lo, float q_hi, float* out2, void* workspace,
                   size_t workspace_bytes);

/* ---- stand-alone operators (same kernels the engine launches; device pointers) -------------- */

/* AudioToMelSpectrogramPreprocessor.forward / FilterbankFeatures.forward
 * (nemo/collections/asr/parts/features.py:334-397): pre-emphasis, STFT(512, hop 160, hann 320),
 * power, mel filterbank, log, per-feature normalisation, mask, pad to a multiple of `pad_to`.
 * audio f32 [B][S]; audio_lens i32 [B] (samples); fb f32 [n_mels][257]; window f32 [320];
 * feats f32 [B][n_mels][T_pad]; feat_lens i32 [B].  T_pad = qasr_frontend_frames(S, pad_to).
 * workspace: device memory of qasr_frontend_workspace_bytes(B, S, n_mels) bytes, 16-byte aligned: tables that depend
 * on the filterbank only (non-zero run of every mel filter, the runs' weights packed back to back, FFT twiddles).
 * qasr_frontend_mel fills them and runs (self-contained, 5 launches).  A caller whose filterbank is fixed — every
 * model is — calls qasr_frontend_plan once and qasr_frontend_mel_planned per batch (2 launches: k_mel, k_norm); the
 * workspace is read-only from then on and may be shared by any number of streams. */
int qasr_frontend_mel(void* stream, const float* audio, const int32_t* audio_lens, int B, int S,
                      const float* fb, const float* window, int n_mels, float preemph, int pad_to,
                      float* feats, int32_t* feat_lens, void* workspace, size_t workspace_bytes);
int qasr_frontend_plan(void* stream, const float* fb, int n_mels, void* workspace, size_t workspace_bytes);
int qasr_frontend_mel_planned(void* stream, const float* audio, const int32_t* audio_lens, int B, int S,
                              const float* fb, const float* window, int n_mels, float preemph, int pad_to,
                              float* feats, int32_t* feat_lens, const void* workspace, size_t workspace_bytes);
int qasr_frontend_frames(int S, int pad_to);
size_t qasr_frontend_workspace_bytes(int B, int S, int n_mels);

/* QuantConv1d.int_conv accumulator only (quant_modules.py:301-305) for a 1x1 conv:
 * x i8 [B][cin][Tp] (x_unsigned: bytes are u8, and bias must carry +128*sum(W)); bias i32 [cout_pad128] or NULL;
 * w s8, cout_pad128 x cin_pad128 values in MFMA fragment order: byte ((tile*(cin_pad/32) + ks)*64 + lane)*16 + j holds
 * W[32*tile + (lane & 31)][32*ks + 16*(lane >> 5) + j]; acc i32 [B][cout][Tp].  T valid columns. */
int qasr_pw_conv_acc(void* stream, const int8_t* x, int x_unsigned, const int8_t* w, const int32_t* bias,
                     int B, int cin, int cin_pad, int cout, int T, int Tp, int32_t* acc);
/* depthwise int_conv accumulator: w s8 [c][kpad]; acc i32 [B][c][Tp_out]; bias i32 [c] or NULL (x_unsigned: bytes
 * are u8 and the kernel feeds x - 128, so bias carries 128 * sum(w[c]) to give the accumulator of the u8 codes) */
int qasr_dw_conv_acc(void* stream, const int8_t* x, int x_unsigned, const int8_t* w, const int32_t* bias, int B, int c,
                     int kernel, int kpad, int stride, int dilation, int padding, int T, int Tp, int T_out, int Tp_out,
                     int32_t* acc);
/* dense (groups = 1) int_conv accumulator for any kernel / stride / dilation (Jasper's convs; quant_modules.py:301-305):
 * w s8 [cout_pad128][kernel][cin_pad128]; bias i32 [cout_pad128] or NULL (x_unsigned: + 128 * sum(W[co]) over all taps);
 * acc i32 [B][cout][Tp_out] */
int qasr_dense_conv_acc(void* stream, const int8_t* x, int x_unsigned, const int8_t* w, const int32_t* bias, int B, int cin,
                        int cin_pad, int cout, int kernel, int stride, int dilation, int padding, int T, int Tp, int T_out,
                        int Tp_out, int32_t* acc);
/* fixedpoint_mul.forward for one operand (quant_utils.py:187-198,213): q = clamp(rint(acc*m[c]), lo, hi)
 * with m[c] = mantissa*2^-e as f64; exact_z selects the float32 round trip through sb[c]. */
int qasr_requant(void* stream, const int32_t* acc, const double* m, const float* sb, int exact_z, int relu,
                 int B, int c, int Tp, int lo, int hi, int8_t* out);

/* ---- dynamic-quantisation device path (QuantAct.forward with dynamic=True, quant_modules.py:149-194) --------------
 * In dynamic mode every QuantAct takes its range from the batch in front of it, so multipliers, bias integers and
 * output scales are data dependent.  These entries keep the whole derivation on the device (the reference goes through
 * host numpy in batch_frexp, quant_utils.py:121-147); qasr/dynamic.py strings them together with the conv accumulator
 * kernels above into ConvASREncoder / ConvASRDecoder forward passes.  A float tensor is handed over as the integers of
 * its producer times their float32 scales: */
typedef struct qasr_dyn_view {
  const void* data;            /* int32 [B][C][Tp] accumulators, or int8 [B][C][Tp] codes (is_int8) */
  const float* scale;          /* f32 [C] (per_channel) or [1] */
  int32_t is_int8, per_channel;
  /* Optional (both or neither; accumulators only): the reference's conv_int is F.conv1d in double over
   * x_int = fl32(x / pre_sf) (quant_modules.py:301-305), and that float32 quotient is the integer or a float32 neighbour
   * of it - conv_int = integers + sum(w residue), rounded once by .type(torch.float).  residue_lo + 128 residue_hi is that
   * sum in units of 2^-24 (the int_conv accumulators of the two tensors qasr_dyn_residue_codes writes, no bias).  It never
   * changes a code, but it is in the float tensor a dynamic QuantAct ranges over. */
  const int32_t* residue_lo;
  const int32_t* residue_hi;
} qasr_dyn_view;
/* x_act.min() / .max() (quant_modules.py:152-153): x_act = relu?(a) (+ b as the identity), zero where t >= lens[b]
 * (MaskedConv1d's mask, jasper.py:177-181; lens NULL: no mask), over t < T.  xf != NULL: the tensor is the float input
 * itself, [B][C][Tx] (first layer).  minmax: two order-preserving encodings of the float32 min / max. */
int qasr_dyn_range(void* stream, const qasr_dyn_view* a, const qasr_dyn_view* b, const float* xf, int Tx,
                   const int32_t* lens, int relu, int B, int C, int T, int Tp, uint32_t* minmax);
/* The percentile form of the same range (qm.set_percentile in force, quant_modules.py:158-167): minmax =
 * (torch.quantile(x_act, q_lo), torch.quantile(x_act, q_hi)) over ALL B C T elements, masked zeros included, in the
 * encoding of qasr_dyn_range.  x_act: device scratch of B C T floats (the tensor is written out once, then
 * qasr_quantile2 selects on it); workspace as for qasr_quantile2. */
int qasr_dyn_range_percentile(void* stream, const qasr_dyn_view* a, const qasr_dyn_view* b, const float* xf, int Tx,
                              const int32_t* lens, int relu, int B, int C, int T, int Tp, float q_lo, float q_hi,
                              float* x_act, void* workspace, size_t workspace_bytes, uint32_t* minmax);
/* act_scaling_factor = max(|min|, |max|, 1e-8) / (2^(bits-1) - 1) (quant_utils.py:44-54) -> s_out[0]; per channel the
 * fixedpoint_mul multiplier m 2^-e of f64(pre_sf[c]) / f64(act_sf) with (m, e) = batch_frexp (quant_utils.py:121-147,
 * 190-196) as float64 -> Ma[c] (operand a, scales sa) and Mb[c] (operand b, scales sb; NULL: none). */
int qasr_dyn_act_params(void* stream, const uint32_t* minmax, int bits, int C, const float* sa, int a_per_channel,
                        const float* sb, int b_per_channel, float* s_out, double* Ma, double* Mb);
/* fixedpoint_mul.forward (quant_utils.py:163-216): z = round(relu?(view) / pre_sf) through the float32 view, out =
 * clamp(round(z_a Ma) (+ round(z_b Mb)), lo, hi) as int8 / uint8 bytes [B][C][Tp]; columns >= min(lens[b], T) are 0. */
int qasr_dyn_requant(void* stream, const qasr_dyn_view* a, const double* Ma, const qasr_dyn_view* b, const double* Mb,
                     const int32_t* lens, int relu, int B, int C, int T, int Tp, int lo, int hi, int8_t* out);
/* first layer (quant_modules.py:180-184): s_out[0] = act_scaling_factor, out = clamp(round(fl32(1/s) x), -n, n-1) */
int qasr_dyn_quant_in(void* stream, const float* x, int Tx, const uint32_t* minmax, const int32_t* lens, int bits, int B,
                      int C, int T, int Tp, float* s_out, int8_t* out);
/* residue(q) = fl32(fl32(q s_x[0]) / s_x[0]) - q of every code byte (see qasr_dyn_view), in units of 2^-24, as
 * lo + 128 hi with |lo| <= 64, |hi| <= 2: two s8 tensors of n bytes (n a multiple of 16, 16-byte aligned pointers). */
int qasr_dyn_residue_codes(void* stream, const int8_t* codes, int x_unsigned, const float* s_x, size_t n, int8_t* lo,
                           int8_t* hi);
/* the conv that consumes those codes (quant_modules.py:293-299,307): sf_out[c] = s_w[c] s_x, bias[c] =
 * clamp(round(fl32(1/sf_out[c]) bprime[c])) evaluated in float32 (+ wsum128[c] = 128 sum(W[c]) when the codes are
 * stored as u8); rows C..C_pad-1: scale 1, bias 0. */
int qasr_dyn_conv_params(void* stream, const float* s_x, const float* s_w, const float* bprime, const int32_t* wsum128,
                         int C, int C_pad, float* sf_out, int32_t* bias);

/* One fused time-channel-separable layer exactly as the engine launches it (k_sep2 / k_sep): depthwise QuantConv1d
 * (K taps, stride 1, 'same' padding) -> QuantAct requant -> 1x1 QuantConv1d [-> residual 1x1 QuantConv1d + res_act] ->
 * ReLU -> the consumers' QuantAct requant (jasper.py:569-600,664-687; quant_modules.py:186-190,301-309).  Every pointer
 * is a device pointer in the blob's layouts (see qasr_op_desc): the parity tests drive the production kernels through
 * this entry with operands of their own making (accumulators beyond 2^22, rounding ties, ragged lengths).
 * K == 0: no depthwise stage (`x` feeds the 1x1 conv).  gen: 2 = k_sep2 where it has the shape, 1 = k_sep.
 * tile: 32, 64 or 128 (k_sep2's plain layers; others fall back to 64) frames per work-group.  `label` (optional)
 * receives the kernel instantiation that ran. */
typedef struct qasr_sep_layer_args {
  int32_t B, T, Tp, cin, cout, K, dilation, tile, gen;
  uint32_t flags;              /* QASR_F_RELU | MASK_OUT | EXACT_Z | RESADD */
  const int8_t* x;             /* [B][cin][Tp] depthwise input (1x1 input when K == 0) */
  int32_t x_unsigned, dw_lo, dw_hi, n_outs;
  const int8_t* wdw;           /* s8 [cin][kpad4] */
  const int8_t* wdw2;          /* s8 [cin][kpad4 + 32]: taps behind 8 zero bytes */
  const int32_t* bias_dw;      /* i32 [cin_pad128] (128 * sum(w) for u8 inputs) */
  const double* m_dw;          /* f64 [cin_pad128] */
  const int8_t* w;             /* s8 cout_pad128 x cin_pad128, MFMA fragment order */
  const int32_t* bias;         /* i32 [cout_pad128] */
  const float* sb;             /* f32 [cout_pad128] (EXACT_Z) */
  const double* m_main;        /* f64 [cout_pad128] (RESADD) */
  const int32_t* lens;         /* i32 [B] valid frames */
  const int8_t* rx;            /* RESADD: residual conv input [B][rcin][Tp] */
  const int8_t* rw;            /*         its weights, fragment order */
  const int32_t* rbias;
  const double* rm;
  const float* rsb;
  int32_t rcin, r_unsigned, qlo, qhi;
  struct {
    void* ptr;                 /* i8 [B][cout][Tp] */
    const double* mtab;        /* mode 1 */
    double m;                  /* mode 0 */
    int32_t lo, hi, mode, pad_;
  } outs[QASR_MAX_OUTS];
  int32_t* dw_acc;             /* optional hooks: i32 [B][cin][Tp], [B][cout][Tp], [B][cout][Tp] */
  int32_t* acc;
  int32_t* racc;
} qasr_sep_layer_args;
int qasr_sep_layer(void* stream, const qasr_sep_layer_args* a, char* label, size_t label_cap);

/* Diagnostics: when set to a device buffer of 32 int64, work-group (1,0,0) of every k_sep launch writes s_memtime
 * stamps at its phase boundaries (slot 31 = number of stamps); NULL (default) disables. */
int qasr_debug_prof(void* dev_buf);
/* per-work-group timeline of the k_sep2 launches that follow: dev_buf[4 wg .. 4 wg + 3] = {start, end (100 MHz
 * s_memrealtime), HW_ID | XCC_ID << 32, shader cycles (s_memtime) between the two}, wg = blockIdx.y * gridDim.x +
 * blockIdx.x < capacity_work_groups (dev_buf holds 4 * capacity_work_groups int64; work-groups beyond it write nothing);
 * NULL switches it off.  A forward that would CAPTURE a hipGraph while either diagnostic buffer is set is refused. */
int qasr_debug_timeline(void* dev_buf, size_t capacity_work_groups);

const char* qasr_last_error(void);
const char* qasr_version(void);

#ifdef __cplusplus
}
#endif
#endif /* QASR_H */
