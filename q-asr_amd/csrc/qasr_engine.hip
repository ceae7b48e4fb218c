// Host runtime of the integer-only ASR engine: blob parsing, static buffer arena, launch plan, C ABI.
// One engine per device; all work is enqueued on the caller's stream; no host synchronisation on the
// forward path (parity hooks excepted).  See include/qasr.h for the contract.
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "qasr_internal.h"

#include <unordered_map>

using namespace qasr;

// ---- sub-byte weight storage (QASR_F_W6PACK): expansion to int8 at load time --------------------------------------
// In-register unpack would cost ~28 VALU instructions per 16-byte MFMA fragment (120 cycles against the MFMA's 32), so
// the packed form is what travels (file, RCCL broadcast) and the kernels keep reading int8 fragments.
__global__ void __launch_bounds__(256) k_unpack6(const uint8_t* __restrict__ in, int8_t* __restrict__ out, size_t n_quads) {
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < n_quads; q += (size_t)gridDim.x * blockDim.x) {
    const unsigned w = in[3 * q] | ((unsigned)in[3 * q + 1] << 8) | ((unsigned)in[3 * q + 2] << 16);
    unsigned o = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = (int)((w >> (6 * i)) & 0x3f);
      o |= (unsigned)(((c ^ 0x20) - 0x20) & 0xff) << (8 * i);      // sign-extend the 6-bit field
    }
    ((unsigned*)out)[q] = o;
  }
}
// zero-margined tap rows of the MFMA depthwise stage: [c][kp + 32], taps behind 8 zero bytes (pack.py writes them
// into byte-per-code blobs; sub-byte blobs derive them here)
__global__ void __launch_bounds__(256) k_tap_rows(const int8_t* __restrict__ w, int8_t* __restrict__ out, int C, int K, int kp) {
  const int ks = kp + 32;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)C * ks; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i / ks), j = (int)(i - (size_t)c * ks) - 8;
    out[i] = (j >= 0 && j < K) ? w[(size_t)c * kp + j] : (int8_t)0;
  }
}

static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIPCHK(x)                                                                          \
  do {                                                                                     \
    hipError_t _e = (x);                                                                   \
    if (_e != hipSuccess) return fail(QASR_ERR_HIP, "%s: %s", #x, hipGetErrorString(_e)); \
  } while (0)

static inline int rup(int x, int m) { return (x + m - 1) / m * m; }
static inline size_t dt_size(uint32_t dt) { return (dt == QASR_DT_F32 || dt == QASR_DT_I32) ? 4 : 1; }

struct TensorRT {
  qasr_tensor_desc d;
  int T = 0, Tp = 0;
  size_t bytes = 0;
  void* ptr = nullptr;
  int slot = -1;
};

struct qasr_engine {
  int device = 0;
  bool debug = false;                  // keep every tensor + dump int32 accumulators (parity hooks)
  bool timing = false;                 // per-op HIP events (qasr_engine_last_op_ms)
  std::vector<uint8_t> blob;           // host copy
  uint8_t* dblob = nullptr;            // device copy
  int8_t* dexp = nullptr;              // QASR_F_W6PACK: int8 expansion of every sub-byte weight array (+ derived tap rows)
  std::unordered_map<uint64_t, const int8_t*> wexp, wexp2;   // blob data offset -> expanded array / tap rows
  qasr_blob_header h{};
  const qasr_tensor_desc* tdesc = nullptr;
  const qasr_op_desc* ops = nullptr;
  const qasr_domain_desc* doms = nullptr;
  // shape plan
  int B = 0, T0 = 0;
  std::vector<int> domT;
  std::vector<TensorRT> tens;
  std::vector<void*> slots;            // owned device buffers
  std::vector<size_t> slot_bytes;
  int32_t* lens_all = nullptr;
  int32_t* time_tokens = nullptr;      // scratch token buffer for qasr_engine_time_ops
  std::vector<std::vector<int32_t*>> acc_dbg;   // [op][1 + pane]
  std::vector<hipEvent_t> ev;          // debug timing: n_ops + 1 events
  bool timed = false;
  bool fuse = true;                    // fuse depthwise -> pointwise pairs into k_sep (QASR_NO_FUSE=1 disables)
  std::vector<int> fused_dw;           // per op: index of the DW op fused into this PW op, or -1
  std::vector<char> skip;              // per op: launched as part of the following op
  bool fuse_stem = true;               // block 0 (lengths, first-layer quantisation, strided depthwise, 1x1) in one launch (QASR_NO_FUSE_STEM=1: four)
  bool stem = false;                   // ... and the plan has that shape: ops 0..2 run as k_stem
  bool fuse_norm = true;               // forward_audio: normalize_batch inside k_stem from k_mel's per-tile sums (qasr_engine_opts.fuse_norm)
  double* norm_stats = nullptr;        // [B][tiles][n_mels][2]
  size_t norm_stats_bytes = 0;
  int norm_tiles = 0, norm_frames = 0; // of the forward being enqueued (0: the stem reads normalised features)
  int fe_launches = 0;                 // front-end kernels of the last forward (forward_audio: k_mel [+ k_norm])
  const int32_t* cur_lens = nullptr;   // the caller's lengths of the current / last forward (k_stem derives every domain's from them)
  bool fuse_dec = true;                // decoder conv + log-softmax + argmax in one launch (QASR_NO_FUSE_DEC=1: two launches)
  std::vector<char> rq_skip;           // per op: REQUANT op served by the launch of an earlier REQUANT op of the same stored value
  std::vector<char> dec_skip;          // per op: LOGSOFTMAX op that ran inside the preceding decoder launch
  bool tile128 = true;                 // tile_frames == 128 (QASR_TILE128=0: k_sep2's plain layers stay on 64-frame tiles, A/B runs)
  bool res_tile128 = true;             // block-end layers on 128-frame tiles too (qasr_engine_opts.res_tile128)
  bool dense_tile128 = true;           // QASR_DENSE_TILE128=0 keeps Jasper's dense convs on 64-frame tiles (A/B runs)
  bool wide_tiles = false;             // k_sep with 64-frame tiles (throughput mode: bit 3 of `debug`, or QASR_WIDE_TILES=1)
  int sep_gen = 2;                     // 2: k_sep2 where it has the shape; 1 (QASR_SEP_GEN=1): k_sep everywhere (A/B runs)
  // hipGraph replay (bit 4 of `debug`): the whole forward of one (shape, buffer set) is captured once and re-launched
  // with one call; key = the caller's pointers, which a serving loop keeps stable
  bool forwarded = false;              // a forward has been enqueued with the current plan
  bool use_graph = false;
  hipGraphExec_t gexec = nullptr;
  const void* gkey[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  // everything else the captured front-end nodes bake in: S, n_mels, pad_to, preemph (bits), fb, window (a caller that
  // reuses its buffers with another S of equal T_pad, or swaps the filterbank, must not replay the old graph)
  uint64_t gfe[6] = {0, 0, 0, 0, 0, 0};
  int gcalls = 0;                      // forwards seen with the current key (1st: direct launches, 2nd: capture)
};

template <class T>
static const T* dev_at(const qasr_engine* e, uint64_t off) {
  return off ? reinterpret_cast<const T*>(e->dblob + e->h.data_off + off) : nullptr;
}
// int8 weight array at blob offset `off`: the blob bytes, or their load-time expansion for sub-byte blobs
static const int8_t* dev_w(const qasr_engine* e, uint64_t off) {
  if (!off) return nullptr;
  auto it = e->wexp.find(off);
  return it != e->wexp.end() ? it->second : dev_at<int8_t>(e, off);
}

static int conv_out_len(int len, const qasr_domain_desc& d) {
  int num = len + 2 * (int)d.padding - (int)d.dilation * ((int)d.kernel - 1) - 1;
  int q = num >= 0 ? num / (int)d.stride : -((-num + (int)d.stride - 1) / (int)d.stride);
  return q + 1;
}

static void free_plan(qasr_engine* e) {
  for (void* p : e->slots) (void)hipFree(p);
  e->slots.clear();
  e->slot_bytes.clear();
  if (e->lens_all) (void)hipFree(e->lens_all);
  e->lens_all = nullptr;
  if (e->time_tokens) (void)hipFree(e->time_tokens);
  e->time_tokens = nullptr;
  if (e->norm_stats) (void)hipFree(e->norm_stats);
  e->norm_stats = nullptr;
  e->norm_stats_bytes = 0;
  e->norm_tiles = e->norm_frames = 0;
  for (auto& v : e->acc_dbg)
    for (auto p : v)
      if (p) (void)hipFree(p);
  e->acc_dbg.clear();
  e->tens.clear();
  e->B = e->T0 = 0;
  e->forwarded = false;
}

static int build_plan(qasr_engine* e, int B, int T0) {
  free_plan(e);
  const auto& h = e->h;
  e->domT.assign(h.n_domains, 0);
  e->domT[0] = T0;
  for (uint32_t d = 1; d < h.n_domains; ++d) e->domT[d] = conv_out_len(e->domT[e->doms[d].parent], e->doms[d]);
  for (uint32_t d = 0; d < h.n_domains; ++d)
    if (e->domT[d] <= 0) return fail(QASR_ERR_ARG, "input of %d frames is too short for domain %u", T0, d);
  e->tens.resize(h.n_tensors);
  for (uint32_t i = 0; i < h.n_tensors; ++i) {
    TensorRT& t = e->tens[i];
    t.d = e->tdesc[i];
    t.T = e->domT[t.d.domain];
    t.Tp = rup(t.T, 64);
    // f32 logits are [B][T][C]; everything else [B][C][Tp]
    t.bytes = (t.d.dtype == QASR_DT_F32) ? (size_t)B * t.T * t.d.channels * 4 : (size_t)B * t.d.channels * t.Tp * dt_size(t.d.dtype);
    t.bytes = (t.bytes + 255) / 256 * 256;
  }
  // fusion plan first: it moves the point where a depthwise input is read to the following launch
  e->fused_dw.assign(h.n_ops, -1);
  e->skip.assign(h.n_ops, 0);
  e->dec_skip.assign(h.n_ops, 0);
  e->rq_skip.assign(h.n_ops, 0);
  if (e->fuse)
    for (uint32_t oi = 0; oi + 1 < h.n_ops; ++oi) {
      const qasr_op_desc& d = e->ops[oi];
      const qasr_op_desc& q = e->ops[oi + 1];
      if (d.kind != QASR_OP_DW || q.kind != QASR_OP_PW) continue;
      const uint32_t same_pad = d.dilation > 1 ? (d.dilation * d.kernel) / 2 - 1 : d.kernel / 2;
      if (d.stride != 1 || d.padding != same_pad || !(d.kernel & 1) || !sep_supported((int)d.kernel, (int)d.dilation)) continue;
      if (d.outs[1].tensor >= 0 || d.outs[0].mode != 1 || q.in != d.outs[0].tensor) continue;
      if (e->tdesc[d.outs[0].tensor].last_use != (int)oi + 1 || (d.flags & QASR_F_EXACT_Z)) continue;
      e->fused_dw[oi + 1] = (int)oi;
      e->skip[oi] = 1;
    }
  // greedy arena: a slot is reused once its tensor's last reader has been enqueued (stream order makes that safe)
  std::vector<int> free_slots;
  auto acquire = [&](TensorRT& t) -> int {
    int best = -1;
    if (!e->debug)
      for (size_t k = 0; k < free_slots.size(); ++k) {
        int s = free_slots[k];
        if (e->slot_bytes[s] >= t.bytes && (best < 0 || e->slot_bytes[s] < e->slot_bytes[free_slots[best]])) best = (int)k;
      }
    int s;
    if (best >= 0) {
      s = free_slots[best];
      free_slots.erase(free_slots.begin() + best);
    } else {
      void* p = nullptr;
      if (hipMalloc(&p, t.bytes) != hipSuccess) return -1;
      e->slots.push_back(p);
      e->slot_bytes.push_back(t.bytes);
      s = (int)e->slots.size() - 1;
    }
    t.slot = s;
    t.ptr = e->slots[s];
    return s;
  };
  for (uint32_t oi = 0; oi < h.n_ops; ++oi) {
    for (uint32_t i = 1; i < h.n_tensors; ++i)      // tensor 0 is the caller's feature buffer
      if (e->tens[i].d.producer == (int)oi && acquire(e->tens[i]) < 0) return fail(QASR_ERR_HIP, "hipMalloc failed (arena)");
    for (uint32_t i = 1; i < h.n_tensors; ++i) {
      TensorRT& t = e->tens[i];
      // a depthwise op fused into the next op's launch reads its input THERE: the input must outlive that launch,
      // or the fused kernel's output could be planned into the very buffer its halo reads come from
      int last = std::max(t.d.last_use, t.d.producer);
      if (last >= 0 && last + 1 < (int)h.n_ops && e->skip[last]) last += 1;
      bool dead_after = t.d.producer <= (int)oi && t.slot >= 0 && last == (int)oi;
      if (dead_after && !e->debug) free_slots.push_back(t.slot);
    }
  }
  HIPCHK(hipMalloc((void**)&e->lens_all, sizeof(int32_t) * h.n_domains * B));
  HIPCHK(hipMalloc((void**)&e->time_tokens, sizeof(int32_t) * (size_t)B * (T0 + 64)));
  if (e->debug) {
    e->acc_dbg.resize(h.n_ops);
    for (uint32_t oi = 0; oi < h.n_ops; ++oi) {
      const qasr_op_desc& op = e->ops[oi];
      if (op.kind != QASR_OP_DW && op.kind != QASR_OP_PW && op.kind != QASR_OP_DENSE) continue;
      const TensorRT& o = e->tens[op.outs[0].tensor];
      size_t n = (size_t)B * op.cout * rup(o.T, 64);
      for (uint32_t k = 0; k < 1 + op.n_panes; ++k) {
        int32_t* p = nullptr;
        HIPCHK(hipMalloc((void**)&p, n * 4));
        HIPCHK(hipMemset(p, 0, n * 4));
        e->acc_dbg[oi].push_back(p);
      }
    }
  }
  if (e->timing && e->ev.empty()) {
    e->ev.resize(h.n_ops + 1);
    for (auto& v : e->ev) HIPCHK(hipEventCreate(&v));
  }
  e->B = B;
  e->T0 = T0;
  return QASR_OK;
}

static void fill_out(const qasr_engine* e, const qasr_out& o, OutP& d) {
  d.ptr = e->tens[o.tensor].ptr;
  d.mtab = dev_at<double>(e, o.m_off);
  d.m = o.m;
  d.lo = o.lo;
  d.hi = o.hi;
  d.mode = (int)o.mode;
  d.pad_ = 0;
}

static int fill_epi(const qasr_engine* e, int oi, const qasr_op_desc& op, EpiP& ep) {
  memset(&ep, 0, sizeof ep);
  int n = 0;
  for (int j = 0; j < QASR_MAX_OUTS; ++j)
    if (op.outs[j].tensor >= 0) fill_out(e, op.outs[j], ep.outs[n++]);
  ep.n_outs = n;
  ep.flags = op.flags;
  ep.sb = dev_at<float>(e, op.sb_off);
  ep.m_main = dev_at<double>(e, op.m_off);
  const TensorRT& o0 = e->tens[op.outs[0].tensor];
  ep.lens = e->lens_all + (size_t)o0.d.domain * e->B;
  ep.acc_dbg = (e->debug && !e->acc_dbg[oi].empty()) ? e->acc_dbg[oi][0] : nullptr;
  ep.qlo = op.qlo;
  ep.qhi = op.qhi;
  ep.T = o0.T;
  ep.Tp = rup(o0.T, 64);
  ep.cout = (int)op.cout;
  ep.B = e->B;
  if (op.flags & QASR_F_LOGITS) {
    ep.logits = (float*)o0.ptr;
    ep.n_outs = 0;
  }
  return QASR_OK;
}

static void fill_panes(const qasr_engine* e, int oi, const qasr_op_desc& op, PaneP* panes) {
  for (uint32_t k = 0; k < op.n_panes; ++k) {
    const qasr_pane& s = op.panes[k];
    PaneP& d = panes[k];
    const TensorRT& t = e->tens[s.in];
    d.x = (const int8_t*)t.ptr;
    d.w = dev_w(e, s.w_off);
    d.bias = dev_at<int32_t>(e, s.bias_off);
    d.m = dev_at<double>(e, s.m_off);
    d.sb = dev_at<float>(e, s.sb_off);
    d.acc_dbg = e->debug ? e->acc_dbg[oi][1 + k] : nullptr;
    d.cin = (int)s.cin;
    d.cin_pad = rup((int)s.cin, 128);
    d.x_unsigned = t.d.dtype == QASR_DT_U8;
    d.pad_ = 0;
  }
}

extern "C" {

const char* qasr_last_error(void) { return g_err.c_str(); }
const char* qasr_version(void) { return "qasr-hip 0.1 (gfx950)"; }

int qasr_debug_prof(void* dev_buf) {
  qasr::g_prof = (long long*)dev_buf;
  qasr::g_prof_mode = 0;
  qasr::g_prof_cap = 0;
  return QASR_OK;
}
int qasr_debug_timeline(void* dev_buf, size_t capacity_work_groups) {
  if (dev_buf && (capacity_work_groups < 1 || capacity_work_groups > (1u << 22))) return fail(QASR_ERR_ARG, "debug_timeline: capacity");
  qasr::g_prof = (long long*)dev_buf;
  qasr::g_prof_mode = dev_buf ? 1 : 0;
  qasr::g_prof_cap = dev_buf ? (int)capacity_work_groups : 0;
  return QASR_OK;
}

void qasr_engine_default_opts(qasr_engine_opts* o) {
  if (!o) return;
  memset(o, 0, sizeof *o);
  o->struct_size = (uint32_t)sizeof *o;
  o->fuse_dw = o->fuse_stem = o->fuse_decoder = o->res_tile128 = o->dense_tile128 = o->fuse_norm = -1;
}

// the `debug` bits of round 1 / 2 callers, as options
int qasr_engine_create(const void* blob, size_t n, int device, int debug, qasr_engine** out) {
  qasr_engine_opts o;
  qasr_engine_default_opts(&o);
  o.debug = (uint32_t)debug & 3u;
  if (debug & 4) return fail(QASR_ERR_UNSUPPORTED, "qasr_engine_create: bit 2 (k_utt) was retired in round 4");
  o.tile_frames = (debug & 8) ? 128 : 32;
  o.graph = (debug & 16) != 0;
  o.fuse_norm = 0;       // callers of this entry read `feats` of qasr_engine_forward_audio as the NORMALISED log-mel (rounds 1 / 2)
  return qasr_engine_create_ex(blob, n, device, &o, out);
}

int qasr_engine_create_ex(const void* blob, size_t n, int device, const qasr_engine_opts* opts, qasr_engine** out) {
  if (!blob || !out || n < sizeof(qasr_blob_header)) return fail(QASR_ERR_ARG, "null / short blob");
  qasr_engine_opts o;
  qasr_engine_default_opts(&o);
  if (opts) {
    if (opts->struct_size < 8 || opts->struct_size > sizeof o) return fail(QASR_ERR_ARG, "qasr_engine_opts.struct_size %u (this library: %zu)", opts->struct_size, sizeof o);
    memcpy(&o, opts, opts->struct_size);                     // an older, shorter struct keeps the defaults of the newer fields
    o.struct_size = (uint32_t)sizeof o;
  }
  if (o.tile_frames != 0 && o.tile_frames != 32 && o.tile_frames != 64 && o.tile_frames != 128)
    return fail(QASR_ERR_ARG, "qasr_engine_opts.tile_frames %d (0, 32, 64 or 128)", o.tile_frames);
  if (o.sep_gen < 0 || o.sep_gen > 2) return fail(QASR_ERR_ARG, "qasr_engine_opts.sep_gen %d (0, 1 or 2)", o.sep_gen);
  if (o.retired_whole_utterance > 0 || o.retired_legacy_pw > 0 || o.retired_persistent > 0)
    return fail(QASR_ERR_UNSUPPORTED, "qasr_engine_opts: whole_utterance / legacy_pw / persistent were retired in round 4 (include/qasr.h)");
  const int debug = (int)o.debug;
  {                                                          // every offset / index / shape of the blob, before any HIP call
    char why[256];
    if (qasr_blob_check(blob, n, why, sizeof why) != QASR_OK) return fail(QASR_ERR_BLOB, "%s", why);
  }
  qasr_blob_header h;
  memcpy(&h, blob, sizeof h);
  HIPCHK(hipSetDevice(device));
  qasr_engine* e = new qasr_engine();
  e->device = device;
  e->debug = (debug & 1) != 0;
  e->timing = (debug & 3) != 0;
  auto tri = [](int32_t v, bool dflt) { return v < 0 ? dflt : v != 0; };
  e->fuse = tri(o.fuse_dw, true);
  e->fuse_stem = tri(o.fuse_stem, true);
  e->fuse_dec = tri(o.fuse_decoder, true);
  e->fuse_norm = tri(o.fuse_norm, true);
  e->wide_tiles = o.tile_frames >= 64;
  e->tile128 = o.tile_frames == 128;
  e->res_tile128 = tri(o.res_tile128, true);
  e->dense_tile128 = tri(o.dense_tile128, true);
  e->sep_gen = o.sep_gen == 1 ? 1 : 2;
  e->use_graph = o.graph > 0;
  // environment: A/B overrides for profiling runs of an unmodified caller (include/qasr.h lists them), read per create call
  if (getenv("QASR_NO_FUSE")) e->fuse = false;
  if (getenv("QASR_WIDE_TILES")) { e->wide_tiles = true; e->tile128 = true; }
  if (const char* g = getenv("QASR_TILE128")) e->tile128 = atoi(g) != 0;
  if (const char* g = getenv("QASR_RES_TILE128")) e->res_tile128 = atoi(g) != 0;
  if (const char* g = getenv("QASR_NO_FUSE_DEC")) e->fuse_dec = atoi(g) == 0;
  if (const char* g = getenv("QASR_NO_FUSE_STEM")) e->fuse_stem = atoi(g) == 0;
  if (const char* g = getenv("QASR_NO_FUSE_NORM")) e->fuse_norm = atoi(g) == 0;
  if (const char* g = getenv("QASR_DENSE_TILE128")) e->dense_tile128 = atoi(g) != 0;
  if (const char* g = getenv("QASR_SEP_GEN")) e->sep_gen = atoi(g) == 1 ? 1 : 2;
  e->blob.assign((const uint8_t*)blob, (const uint8_t*)blob + n);
  e->h = h;
  e->tdesc = (const qasr_tensor_desc*)(e->blob.data() + h.tensors_off);
  e->ops = (const qasr_op_desc*)(e->blob.data() + h.ops_off);
  e->doms = (const qasr_domain_desc*)(e->blob.data() + h.domains_off);
  if (hipMalloc((void**)&e->dblob, n + 256) != hipSuccess ||      // slack: 16-byte granule copies may overrun an array's tail
      hipMemcpy(e->dblob, blob, n, hipMemcpyHostToDevice) != hipSuccess) {
    delete e;
    return fail(QASR_ERR_HIP, "blob upload failed");
  }
  // sub-byte blobs: expand every packed weight array (and derive the depthwise tap rows) once, on the device
  {
    struct Job { uint64_t off; size_t packed, expanded; int C, K, kp; };
    std::vector<Job> jobs;
    size_t total = 0;
    auto add = [&](uint64_t off, size_t expanded, int C = 0, int K = 0, int kp = 0) {
      if (!off || e->wexp.count(off)) return;
      e->wexp[off] = nullptr;
      jobs.push_back({off, expanded / 4 * 3, expanded, C, K, kp});
      total += (expanded + 255) / 256 * 256;
      if (C) total += ((size_t)C * (kp + 32) + 64 + 255) / 256 * 256;
    };
    for (uint32_t i = 0; i < h.n_ops; ++i) {
      const qasr_op_desc& op = e->ops[i];
      if (!(op.flags & QASR_F_W6PACK)) continue;
      const size_t cp = rup((int)op.cout, 128), cinp = rup((int)op.cin, 128);
      if (op.kind == QASR_OP_DW) add(op.w_off, (size_t)op.cout * rup((int)op.kernel, 4), (int)op.cout, (int)op.kernel, rup((int)op.kernel, 4));
      else add(op.w_off, cp * cinp * (op.kind == QASR_OP_DENSE ? op.kernel : 1));
      for (uint32_t k = 0; k < op.n_panes; ++k) add(op.panes[k].w_off, cp * (size_t)rup((int)op.panes[k].cin, 128));
    }
    if (!jobs.empty()) {
      if (hipMalloc((void**)&e->dexp, total) != hipSuccess) {
        qasr_engine_destroy(e);
        return fail(QASR_ERR_HIP, "weight expansion buffer (%zu bytes)", total);
      }
      size_t pos = 0;
      for (const Job& j : jobs) {
        if (h.data_off + j.off + j.packed > n) {
          qasr_engine_destroy(e);
          return fail(QASR_ERR_BLOB, "packed weight array out of range");
        }
        int8_t* dst = e->dexp + pos;
        pos += (j.expanded + 255) / 256 * 256;
        hipLaunchKernelGGL(k_unpack6, dim3(1024), dim3(256), 0, 0, e->dblob + h.data_off + j.off, dst, j.expanded / 4);
        e->wexp[j.off] = dst;
        if (j.C) {
          int8_t* rows = e->dexp + pos;
          pos += ((size_t)j.C * (j.kp + 32) + 64 + 255) / 256 * 256;
          (void)hipMemsetAsync(rows, 0, (size_t)j.C * (j.kp + 32) + 64, 0);
          hipLaunchKernelGGL(k_tap_rows, dim3(256), dim3(256), 0, 0, dst, rows, j.C, j.K, j.kp);
          e->wexp2[j.off] = rows;
        }
      }
      if (hipDeviceSynchronize() != hipSuccess) {
        qasr_engine_destroy(e);
        return fail(QASR_ERR_HIP, "weight expansion failed");
      }
    }
  }
  *out = e;
  return QASR_OK;
}

void qasr_engine_destroy(qasr_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  if (e->gexec) (void)hipGraphExecDestroy(e->gexec);
  free_plan(e);
  for (auto v : e->ev) (void)hipEventDestroy(v);
  if (e->dblob) (void)hipFree(e->dblob);
  if (e->dexp) (void)hipFree(e->dexp);
  delete e;
}

int qasr_engine_num_ops(const qasr_engine* e) { return e ? (int)e->h.n_ops : -1; }

int qasr_engine_num_launches(const qasr_engine* e) {
  if (!e || !e->B || !e->forwarded) return -1;
  int n = (e->stem ? 0 : 1) + e->fe_launches;               // (k_lens, which the stem absorbs; front-end of forward_audio)
  for (uint32_t oi = 0; oi < e->h.n_ops; ++oi) {
    if (e->skip[oi] || e->dec_skip[oi] || e->rq_skip[oi] || (e->stem && oi >= 1 && oi <= 2)) continue;
    n += 1;
  }
  return n;
}

int qasr_engine_out_frames(const qasr_engine* e, int T) {
  if (!e) return -1;
  std::vector<int> dT(e->h.n_domains);
  dT[0] = T;
  for (uint32_t d = 1; d < e->h.n_domains; ++d) dT[d] = conv_out_len(dT[e->doms[d].parent], e->doms[d]);
  const qasr_op_desc& last = e->ops[e->h.n_ops - 1];
  return dT[e->tdesc[last.in].domain];
}

// parameter block of the fused separable-layer kernels (k_sep2 / k_sep / k_dense2) for PW op `oi`
static void build_sep(qasr_engine* e, uint32_t oi, SepP& p) {
  const qasr_op_desc& op = e->ops[oi];
  const TensorRT& tin = e->tens[op.in];
  p.w = dev_w(e, op.w_off);
  p.bias = dev_at<int32_t>(e, op.bias_off);
  p.cin = (int)op.cin;
  p.cin_pad = rup(p.cin, 128);
  p.n_panes = (int)op.n_panes;
  p.tile = e->wide_tiles ? (e->tile128 ? 128 : 64) : 32;   // 128: k_sep2's separable layers only (sep2_tile), everything else 64
  p.etile = p.tile;
  if (p.tile == 128 && !e->res_tile128 && (op.flags & QASR_F_RESADD)) p.tile = 64;
  p.gen = e->sep_gen;
  fill_panes(e, oi, op, p.panes);
  fill_epi(e, oi, op, p.e);
  const int di = e->fused_dw[oi];
  if (di >= 0) {
    const qasr_op_desc& d = e->ops[di];
    const TensorRT& din = e->tens[d.in];
    p.x = (const int8_t*)din.ptr;
    p.wdw = dev_w(e, d.w_off);
    {                                                       // zero-margined tap rows: in the blob (m_off), or derived on load
      auto it = e->wexp2.find(d.w_off);
      p.wdw2 = it != e->wexp2.end() ? it->second : dev_at<int8_t>(e, d.m_off);
    }
    p.bias_dw = dev_at<int32_t>(e, d.bias_off);
    p.m_dw = dev_at<double>(e, d.outs[0].m_off);
    p.dw_acc_dbg = (e->debug && !e->acc_dbg[di].empty()) ? e->acc_dbg[di][0] : nullptr;
    p.dw_lo = d.outs[0].lo;
    p.dw_hi = d.outs[0].hi;
    p.K = (int)d.kernel;
    p.dilation = (int)d.dilation;
    p.x_unsigned = din.d.dtype == QASR_DT_U8;
    p.pw_unsigned = 0;
    if (d.flags & QASR_F_WIDE_RQ) p.gen = 1;                // k_sep2 clamps on the low word of the rounded product
  } else {
    p.x = (const int8_t*)tin.ptr;
    p.K = 0;
    p.dilation = 1;
    p.pw_unsigned = tin.d.dtype == QASR_DT_U8;
    if (op.kind == QASR_OP_DENSE) {
      p.dense_k = (int)op.kernel;
      p.dilation = (int)op.dilation;
      // the window of all input channels must fit the LDS next to the residual operand and the staging tiles
      const int halo = ((p.dense_k - 1) * p.dilation / 2 + 3) & ~3;
      size_t xr = 0;
      for (int k = 0; k < p.n_panes; ++k) xr = std::max(xr, (size_t)64 * (p.panes[k].cin_pad + 16));
      if (p.tile > 64) p.tile = 64;
      if ((size_t)(64 + 2 * halo) * (p.cin_pad + 16) + xr + 37 * 1024 > 160 * 1024) p.tile = 32;
      // plain dense convs in throughput mode: 128-frame tiles (every weight fragment feeds four frame tiles; a launch
      // then has B * Tp / 128 work-groups and two launches of different steps share the chip) where the window fits
      const TensorRT& o0 = e->tens[op.outs[0].tensor];
      if (e->wide_tiles && e->dense_tile128 && p.n_panes == 0 && rup(o0.T, 64) % 128 == 0 &&
          (size_t)(128 + 2 * halo) * (p.cin_pad + 16) + 37 * 1024 <= 160 * 1024)
        p.tile = 128;
    }
  }
}

static void build_quant_in(const qasr_engine* e, uint32_t oi, QuantInP& p) {
  const qasr_op_desc& op = e->ops[oi];
  const TensorRT& tin = e->tens[op.in];
  const TensorRT& to = e->tens[op.outs[0].tensor];
  p.x = (const float*)tin.ptr;
  p.out = (int8_t*)to.ptr;
  p.lens = e->lens_all + (size_t)to.d.domain * e->B;
  p.inv_scale = op.in_inv_scale;
  p.lo = op.qlo;
  p.hi = op.qhi;
  p.C = (int)op.cin;
  p.T = tin.T;
  p.Tp = to.Tp;
  p.B = e->B;
}
static void build_dw(qasr_engine* e, uint32_t oi, DwP& p) {
  const qasr_op_desc& op = e->ops[oi];
  const TensorRT& tin = e->tens[op.in];
  p.x = (const int8_t*)tin.ptr;
  p.w = dev_w(e, op.w_off);
  p.bias = dev_at<int32_t>(e, op.bias_off);
  p.C = (int)op.cin;
  p.K = (int)op.kernel;
  p.kpad = rup(p.K, 4);
  p.stride = (int)op.stride;
  p.dilation = (int)op.dilation;
  p.padding = (int)op.padding;
  p.T_in = tin.T;
  p.Tp_in = tin.Tp;
  p.x_unsigned = tin.d.dtype == QASR_DT_U8;
  fill_epi(e, oi, op, p.e);
}
// ops 0..2 = [first-layer QuantAct, strided depthwise conv, 1x1 conv], each feeding only the next: the stem k_stem runs
static bool stem_shape(qasr_engine* e) {
  if (e->h.n_ops < 3) return false;
  const qasr_op_desc &a = e->ops[0], &b = e->ops[1], &c = e->ops[2];
  if (a.kind != QASR_OP_QUANT_IN || b.kind != QASR_OP_DW || c.kind != QASR_OP_PW || e->skip[1] || e->fused_dw[2] >= 0) return false;
  if (b.in != a.outs[0].tensor || c.in != b.outs[0].tensor || a.outs[1].tensor >= 0 || b.outs[1].tensor >= 0) return false;
  for (uint32_t oi = 3; oi < e->h.n_ops; ++oi) {             // nobody else reads the two intermediate tensors
    const qasr_op_desc& q = e->ops[oi];
    if (q.in == a.outs[0].tensor || q.in == b.outs[0].tensor) return false;
    for (uint32_t k = 0; k < q.n_panes; ++k)
      if (q.panes[k].in == a.outs[0].tensor || q.panes[k].in == b.outs[0].tensor) return false;
  }
  QuantInP qi{};
  DwP dw{};
  SepP pw{};
  build_quant_in(e, 0, qi);
  build_dw(e, 1, dw);
  build_sep(e, 2, pw);
  return stem_supported(qi, dw, pw);
}
static int stem_launch(qasr_engine* e, hipStream_t s) {
  QuantInP qi{};
  DwP dw{};
  SepP pw{};
  build_quant_in(e, 0, qi);
  build_dw(e, 1, dw);
  build_sep(e, 2, pw);
  int rc = launch_stem(s, qi, dw, pw, (const qasr_domain_desc*)(e->dblob + e->h.domains_off), (int)e->h.n_domains, e->cur_lens,
                       e->lens_all, e->norm_tiles ? e->norm_stats : nullptr, e->norm_tiles, e->norm_frames);
  return rc ? fail(rc, "k_stem launch") : QASR_OK;
}

static int launch_op(qasr_engine* e, hipStream_t s, uint32_t oi, float* logp, int32_t* tokens, int32_t* lens_out) {
  const qasr_op_desc& op = e->ops[oi];
  const int B = e->B;
  if (e->skip[oi]) return QASR_OK;        // runs inside the next op's k_sep launch
  if (e->stem && oi <= 2) return oi == 0 ? stem_launch(e, s) : QASR_OK;   // block 0 as one launch
  const TensorRT& tin = e->tens[op.in];
  switch (op.kind) {
    case QASR_OP_QUANT_IN: {
      QuantInP p{};
      build_quant_in(e, oi, p);
      launch_quant_in(s, p);
      break;
    }
    case QASR_OP_DW: {
      DwP p{};
      build_dw(e, oi, p);
      launch_dw(s, p);
      break;
    }
    case QASR_OP_PW: {
      SepP p{};
      build_sep(e, oi, p);
      if (e->fuse_dec && (op.flags & QASR_F_LOGITS) && oi + 1 < e->h.n_ops && e->ops[oi + 1].kind == QASR_OP_LOGSOFTMAX &&
          e->ops[oi + 1].in == op.outs[0].tensor && decoder_fusable(p)) {
        int rc = launch_decoder(s, p, logp, tokens, lens_out, e->debug);
        if (rc) return fail(rc, "op %u: decoder launch", oi);
        e->dec_skip[oi + 1] = 1;
        break;
      }
      int rc = launch_sep(s, p);
      if (rc) return fail(rc, "op %u: no k_sep instantiation for K=%d dilation=%d (or bad launch shape)", oi, p.K, p.dilation);
      break;
    }
    case QASR_OP_DENSE: {
      if (op.flags & QASR_F_TAPMAJOR) {                     // stride-1 'same' dense conv: taps shifted 1x1 GEMMs on the tile kernel
        SepP p{};
        build_sep(e, oi, p);
        int rc = launch_sep(s, p);
        if (rc) return fail(rc, "op %u: dense conv has no k_sep launch shape", oi);
        break;
      }
      DenseP p{};
      p.x = (const int8_t*)tin.ptr;
      p.w = dev_w(e, op.w_off);
      p.bias = dev_at<int32_t>(e, op.bias_off);
      p.cin = (int)op.cin;
      p.cin_pad = rup(p.cin, 128);
      p.K = (int)op.kernel;
      p.stride = (int)op.stride;
      p.dilation = (int)op.dilation;
      p.padding = (int)op.padding;
      p.T_in = tin.T;
      p.Tp_in = tin.Tp;
      p.x_unsigned = tin.d.dtype == QASR_DT_U8;
      p.n_panes = (int)op.n_panes;
      fill_panes(e, oi, op, p.panes);
      fill_epi(e, oi, op, p.e);
      launch_dense(s, p);
      break;
    }
    case QASR_OP_REQUANT: {
      if (e->rq_skip[oi]) break;                             // served by an earlier launch on the same stored value
      const TensorRT& to = e->tens[op.outs[0].tensor];
      RequantP p{};
      p.in = tin.ptr;
      p.in_is_i32 = tin.d.dtype == QASR_DT_I32;
      fill_out(e, op.outs[0], p.outs[0]);
      p.n_outs = 1;
      // the packer emits the REQUANT ops of one stored value back to back: one launch serves up to QASR_RQ_MAX of them
      for (uint32_t oj = oi + 1; oj < e->h.n_ops && p.n_outs < QASR_RQ_MAX; ++oj) {
        const qasr_op_desc& q = e->ops[oj];
        if (q.kind != QASR_OP_REQUANT || q.in != op.in || q.flags != op.flags || q.sb_off != op.sb_off || q.cin != op.cin ||
            e->tens[q.outs[0].tensor].d.domain != to.d.domain)
          break;
        fill_out(e, q.outs[0], p.outs[p.n_outs++]);
        e->rq_skip[oj] = 1;
      }
      p.sb = dev_at<float>(e, op.sb_off);
      p.lens = e->lens_all + (size_t)to.d.domain * B;
      p.flags = op.flags & QASR_F_MASK_OUT;   // the stored value is already ReLU'd / round-tripped z
      p.C = (int)op.cin;
      p.T = to.T;
      p.Tp = to.Tp;
      p.B = B;
      launch_requant(s, p);
      break;
    }
    case QASR_OP_LOGSOFTMAX: {
      if (e->dec_skip[oi]) break;                            // ran inside the decoder's launch (k_dec), lengths included
      launch_logsoftmax(s, (const float*)tin.ptr, logp, tokens, B * tin.T, (int)op.cin);
      if (lens_out)
        HIPCHK(hipMemcpyAsync(lens_out, e->lens_all + (size_t)tin.d.domain * B, sizeof(int32_t) * B,
                              hipMemcpyDeviceToDevice, s));
      break;
    }
    default:
      return fail(QASR_ERR_UNSUPPORTED, "op kind %u", op.kind);
  }
  return QASR_OK;
}

// front-end of a forward_audio call (nullptr: the caller's features are the input)
struct FrontArgs {
  const float* audio;
  const int32_t* audio_lens;
  int S;
  const float* fb;
  const float* window;
  int n_mels;
  float preemph;
  int pad_to;
  const void* plan;
  size_t plan_bytes;
};

static int forward_impl(qasr_engine* e, hipStream_t s, const FrontArgs* fe, float* feats, int32_t* lens, int B, int T,
                        float* logp, int32_t* tokens, int32_t* lens_out) {
  if (B != e->B || T != e->T0) {
    int rc = build_plan(e, B, T);
    if (rc) return rc;
    if (e->gexec) (void)hipGraphExecDestroy(e->gexec);
    e->gexec = nullptr;
    e->gkey[0] = nullptr;
  }
  const auto& h = e->h;
  e->tens[0].ptr = (void*)feats;
  e->cur_lens = lens;
  e->stem = e->fuse_stem && stem_shape(e);
  // normalize_batch inside k_stem: k_mel leaves per-tile sums, no k_norm launch
  const bool norm_in_stem = fe && e->stem && e->fuse_norm && fe->n_mels == (int)h.feat_in;
  e->norm_tiles = e->norm_frames = 0;                        // (a forward on features hands k_stem normalised input)
  e->fe_launches = fe ? (norm_in_stem ? 1 : 2) : 0;
  if (norm_in_stem) {
    e->norm_frames = 1 + fe->S / 160;
    e->norm_tiles = (e->norm_frames + QASR_MEL_TILE - 1) / QASR_MEL_TILE;
    const size_t need = frontend_stats_bytes(B, fe->S, fe->n_mels);
    if (need > e->norm_stats_bytes) {                        // (a captured graph holds the old pointer)
      if (e->gexec) (void)hipGraphExecDestroy(e->gexec);
      e->gexec = nullptr;
      e->gkey[0] = nullptr;
      if (e->norm_stats) HIPCHK(hipFree(e->norm_stats));
      e->norm_stats = nullptr;
      e->norm_stats_bytes = 0;
      HIPCHK(hipMalloc(&e->norm_stats, need));
      e->norm_stats_bytes = need;
    }
  }
  auto enqueue = [&]() -> int {
    if (fe && norm_in_stem) {
      int nt = 0, nf = 0;
      int rc = frontend_mel_stats(s, fe->audio, fe->audio_lens, B, fe->S, fe->fb, fe->window, fe->n_mels, fe->preemph, fe->pad_to,
                                  feats, lens, fe->plan, fe->plan_bytes, e->norm_stats, &nt, &nf);
      if (rc || nt != e->norm_tiles || nf != e->norm_frames) return fail(rc ? rc : QASR_ERR_ARG, "forward_audio: front-end (k_mel with statistics)");
    } else if (fe) {                                         // mel front-end into the caller's feature / length buffers
      int rc = qasr_frontend_mel_planned(s, fe->audio, fe->audio_lens, B, fe->S, fe->fb, fe->window, fe->n_mels, fe->preemph,
                                         fe->pad_to, feats, lens, fe->plan, fe->plan_bytes);
      if (rc) return fail(rc, "forward_audio: front-end");
    }
    const qasr_domain_desc* ddoms = (const qasr_domain_desc*)(e->dblob + h.domains_off);
    if (!e->stem) launch_lens(s, lens, e->lens_all, ddoms, (int)h.n_domains, B);   // (k_stem derives them itself)
    for (uint32_t oi = 0; oi < h.n_ops; ++oi) {
      if (e->timing) HIPCHK(hipEventRecord(e->ev[oi], s));
      int rc = launch_op(e, s, oi, logp, tokens, lens_out);
      if (rc) return rc;
    }
    return QASR_OK;
  };
  e->forwarded = true;
  if (e->use_graph && s != nullptr && !e->timing && !e->debug) {   // the legacy default stream cannot be captured
    const void* key[8] = {feats, lens, logp, tokens, lens_out, fe ? fe->audio : nullptr, fe ? fe->audio_lens : nullptr,
                          fe ? fe->plan : nullptr};
    uint64_t fkey[6] = {0, 0, 0, 0, 0, 0};
    if (fe) {
      uint32_t pre_bits;
      memcpy(&pre_bits, &fe->preemph, 4);
      fkey[0] = (uint64_t)fe->S; fkey[1] = (uint64_t)fe->n_mels; fkey[2] = (uint64_t)fe->pad_to; fkey[3] = pre_bits;
      fkey[4] = (uint64_t)(uintptr_t)fe->fb; fkey[5] = (uint64_t)(uintptr_t)fe->window;
    }
    bool same = true;
    for (int i = 0; i < 8; ++i) same = same && key[i] == e->gkey[i];
    for (int i = 0; i < 6; ++i) same = same && fkey[i] == e->gfe[i];
    if (!same) {                                             // new buffer set / front-end arguments: drop the old graph, start over
      if (e->gexec) (void)hipGraphExecDestroy(e->gexec);
      e->gexec = nullptr;
      e->gcalls = 0;
      for (int i = 0; i < 8; ++i) e->gkey[i] = key[i];
      for (int i = 0; i < 6; ++i) e->gfe[i] = fkey[i];
    }
    if (e->gexec) {
      HIPCHK(hipGraphLaunch(e->gexec, s));
      return QASR_OK;
    }
    if (e->gcalls++ >= 1) {                                  // second call with this key: capture (the first one ran every
      hipGraph_t g = nullptr;                                // kernel's one-time attribute setup outside a capture)
      if (g_prof) return fail(QASR_ERR_ARG, "graph capture while qasr_debug_prof / qasr_debug_timeline is set: the buffer would be baked into the graph");
      HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      int rc = enqueue();
      hipError_t ce = hipStreamEndCapture(s, &g);
      if (rc) {                                              // an enqueue error inside the capture: nothing is kept
        if (g) (void)hipGraphDestroy(g);
        e->gcalls = 0;
        return rc;
      }
      if (ce != hipSuccess || !g) return fail(QASR_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(ce));
      hipError_t ie = hipGraphInstantiate(&e->gexec, g, nullptr, nullptr, 0);
      (void)hipGraphDestroy(g);
      if (ie != hipSuccess) return fail(QASR_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(ie));
      HIPCHK(hipGraphLaunch(e->gexec, s));
      return QASR_OK;
    }
  }
  {
    int rc = enqueue();
    if (rc) return rc;
  }
  if (e->timing) {
    HIPCHK(hipEventRecord(e->ev[h.n_ops], s));
    e->timed = true;
  }
  HIPCHK(hipGetLastError());
  return QASR_OK;
}

int qasr_engine_forward(qasr_engine* e, void* stream, const float* feats, const int32_t* lens, int B, int T,
                        float* logp, int32_t* tokens, int32_t* lens_out) {
  if (!e || !feats || !lens || B <= 0 || T <= 0) return fail(QASR_ERR_ARG, "bad forward arguments");
  return forward_impl(e, (hipStream_t)stream, nullptr, (float*)feats, (int32_t*)lens, B, T, logp, tokens, lens_out);
}

int qasr_engine_forward_audio(qasr_engine* e, void* stream, const float* audio, const int32_t* audio_lens, int B, int S,
                              const float* fb, const float* window, int n_mels, float preemph, int pad_to,
                              const void* frontend_plan, size_t plan_bytes, float* feats, int32_t* feat_lens, float* logp,
                              int32_t* tokens, int32_t* lens_out) {
  if (!e || !audio || !audio_lens || !fb || !window || !frontend_plan || !feats || !feat_lens || B <= 0 || S <= 0)
    return fail(QASR_ERR_ARG, "bad forward_audio arguments");
  FrontArgs fe{audio, audio_lens, S, fb, window, n_mels, preemph, pad_to, frontend_plan, plan_bytes};
  return forward_impl(e, (hipStream_t)stream, &fe, feats, feat_lens, B, qasr_frontend_frames(S, pad_to), logp, tokens, lens_out);
}

// Replays every op `reps` times back to back between ONE pair of HIP events on `stream` and returns the
// average duration per launch (ms).  The buffers hold the activations of the last forward, so operands are
// real data; each op reads its inputs and rewrites its own outputs, which makes the replay idempotent.
int qasr_engine_time_ops(qasr_engine* e, void* stream, int reps, float* ms_per_launch, int n_ops) {
  if (!e || !e->B || reps <= 0 || n_ops != (int)e->h.n_ops || !ms_per_launch) return fail(QASR_ERR_ARG, "time_ops: run a forward first");
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t a, b;
  HIPCHK(hipEventCreate(&a));
  HIPCHK(hipEventCreate(&b));
  for (uint32_t oi = 0; oi < e->h.n_ops; ++oi) {
    int rc = launch_op(e, s, oi, nullptr, e->time_tokens, nullptr);      // warm
    if (rc) return rc;
    HIPCHK(hipEventRecord(a, s));
    for (int r = 0; r < reps; ++r) launch_op(e, s, oi, nullptr, e->time_tokens, nullptr);
    HIPCHK(hipEventRecord(b, s));
    HIPCHK(hipEventSynchronize(b));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    ms_per_launch[oi] = ms / reps;
  }
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  return QASR_OK;
}

int qasr_engine_op_label(qasr_engine* e, int op, char* buf, size_t cap) {
  if (!e || !buf || cap < 8 || op < 0 || op >= (int)e->h.n_ops) return fail(QASR_ERR_ARG, "op_label: bad argument");
  const qasr_op_desc& d = e->ops[op];
  const char* name = "?";
  if (e->skip[op]) name = "(fused into the next op)";
  else if (e->dec_skip[op]) name = "(fused into the previous op)";
  else if (e->rq_skip[op]) name = "(served by the previous k_requant launch)";
  else if (e->stem && op <= 2) name = op == 0 ? "k_stem" : "(fused into the first op)";
  else switch (d.kind) {
    case QASR_OP_QUANT_IN: name = "k_quant_in"; break;
    case QASR_OP_DW: name = "k_dw"; break;
    case QASR_OP_DENSE:
      if (d.flags & QASR_F_TAPMAJOR) {
        SepP p{};
        build_sep(e, (uint32_t)op, p);
        sep_kernel_label(p, buf, cap);
        return QASR_OK;
      }
      name = "k_dense";
      break;
    case QASR_OP_LOGSOFTMAX: name = "k_logsoftmax"; break;
    case QASR_OP_REQUANT: name = "k_requant"; break;
    case QASR_OP_PW:
      {
        SepP p{};
        build_sep(e, (uint32_t)op, p);
        if (e->fuse_dec && (d.flags & QASR_F_LOGITS) && op + 1 < (int)e->h.n_ops && e->dec_skip[op + 1]) {
          name = "k_dec";
          break;
        }
        sep_kernel_label(p, buf, cap);
        return QASR_OK;
      }
    default: break;
  }
  snprintf(buf, cap, "%s", name);
  return QASR_OK;
}

int qasr_engine_run_op(qasr_engine* e, void* stream, int op) {
  if (!e || !e->B || op < 0 || op >= (int)e->h.n_ops) return fail(QASR_ERR_ARG, "run_op: run a forward first");
  return launch_op(e, (hipStream_t)stream, (uint32_t)op, nullptr, e->time_tokens, nullptr);
}

int qasr_engine_read_acc(qasr_engine* e, int op, int pane, int32_t* host_out, size_t n_elems) {
  if (!e || !e->debug || op < 0 || op >= (int)e->h.n_ops || e->acc_dbg.empty()) return fail(QASR_ERR_ARG, "read_acc: not a debug engine / bad op");
  const auto& v = e->acc_dbg[op];
  int k = pane < 0 ? 0 : 1 + pane;
  if (k >= (int)v.size()) return fail(QASR_ERR_ARG, "read_acc: op %d has no accumulator %d", op, k);
  const qasr_op_desc& d = e->ops[op];
  const TensorRT& o = e->tens[d.outs[0].tensor];
  size_t n = (size_t)e->B * d.cout * rup(o.T, 64);
  if (n_elems != n) return fail(QASR_ERR_ARG, "read_acc: expected %zu elements", n);
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(host_out, v[k], n * 4, hipMemcpyDeviceToHost));
  return QASR_OK;
}

int qasr_engine_read_tensor(qasr_engine* e, int tensor, void* host_out, size_t n_bytes, int* T_out, int* Tp_out) {
  if (!e || tensor <= 0 || tensor >= (int)e->tens.size()) return fail(QASR_ERR_ARG, "read_tensor: bad tensor / no forward yet");
  const TensorRT& t = e->tens[tensor];
  {                                                          // a tensor the launch plan never stores has no bytes to serve
    const int pr = t.d.producer;
    const bool in_launch = pr >= 0 && pr < (int)e->skip.size() && e->skip[pr];           // depthwise output inside the fused layer's launch
    const bool in_stem = e->stem && pr >= 0 && pr <= 1;                                    // k_stem's intermediates
    const bool in_dec = !e->debug && pr >= 0 && pr + 1 < (int)e->dec_skip.size() && e->dec_skip[pr + 1] &&
                        (e->ops[pr].flags & QASR_F_LOGITS);                                // float logits inside k_dec
    if (in_launch || in_stem || in_dec)
      return fail(QASR_ERR_ARG, "read_tensor: tensor %d is never materialised by this plan (its producer, op %d, runs fused inside another launch)", tensor, pr);
  }
  if (!e->debug) {                                           // production engines reuse arena slots: only a tensor nobody overwrote
    for (size_t i = 1; i < e->tens.size(); ++i)
      if ((int)i != tensor && e->tens[i].slot == t.slot && e->tens[i].d.producer > t.d.producer)
        return fail(QASR_ERR_ARG, "read_tensor: the arena slot of tensor %d was reused by tensor %zu (debug engines keep every tensor)", tensor, i);
  }
  if (T_out) *T_out = t.T;
  if (Tp_out) *Tp_out = t.Tp;
  size_t want = (t.d.dtype == QASR_DT_F32) ? (size_t)e->B * t.T * t.d.channels * 4 : (size_t)e->B * t.d.channels * t.Tp * dt_size(t.d.dtype);
  if (!host_out) return QASR_OK;
  if (n_bytes != want) return fail(QASR_ERR_ARG, "read_tensor: expected %zu bytes", want);
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(host_out, t.ptr, want, hipMemcpyDeviceToHost));
  return QASR_OK;
}

int qasr_engine_last_op_ms(qasr_engine* e, float* ms, int n_ops) {
  if (!e || !e->timing || !e->timed || n_ops != (int)e->h.n_ops) return fail(QASR_ERR_ARG, "last_op_ms: no timed forward");
  HIPCHK(hipEventSynchronize(e->ev[n_ops]));
  for (int i = 0; i < n_ops; ++i) HIPCHK(hipEventElapsedTime(&ms[i], e->ev[i], e->ev[i + 1]));
  return QASR_OK;
}

// ---------------------------------------------------------------------------------- stand-alone operators
static void* g_zero = nullptr;
static const size_t kZeroBytes = 1 << 20;
static int zero_buf(void** p) {
  if (!g_zero) {
    HIPCHK(hipMalloc(&g_zero, kZeroBytes));
    HIPCHK(hipMemset(g_zero, 0, kZeroBytes));
  }
  *p = g_zero;
  return QASR_OK;
}

int qasr_pw_conv_acc(void* stream, const int8_t* x, int x_unsigned, const int8_t* w, const int32_t* bias, int B,
                     int cin, int cin_pad, int cout, int T, int Tp, int32_t* acc) {
  if (!x || !w || !acc || cin_pad % 128 || Tp % 64 || T > Tp) return fail(QASR_ERR_ARG, "pw_conv_acc: bad arguments");
  void* z;
  int rc = zero_buf(&z);
  if (rc) return rc;
  if ((size_t)rup(cout, 128) * 8 > kZeroBytes) return fail(QASR_ERR_ARG, "cout too large");
  SepP p{};                            // the production 1x1 kernel (k_sep<0>) with no consumers: accumulators only
  p.x = x;
  p.w = w;
  p.bias = bias ? bias : (const int32_t*)z;
  p.cin = cin;
  p.cin_pad = cin_pad;
  p.pw_unsigned = x_unsigned;
  p.K = 0;
  p.dilation = 1;
  p.e.sb = (const float*)z;
  p.e.acc_dbg = acc;
  p.e.T = T;
  p.e.Tp = Tp;
  p.e.cout = cout;
  p.e.B = B;
  p.e.lens = (const int32_t*)z;
  int lrc = launch_sep((hipStream_t)stream, p);
  if (lrc) return fail(lrc, "pw_conv_acc: launch rejected");
  HIPCHK(hipGetLastError());
  return QASR_OK;
}

int qasr_dw_conv_acc(void* stream, const int8_t* x, int x_unsigned, const int8_t* w, const int32_t* bias, int B, int c,
                     int kernel, int kpad, int stride, int dilation, int padding, int T, int Tp, int T_out, int Tp_out,
                     int32_t* acc) {
  if (!x || !w || !acc || kpad % 4 || kpad < kernel || Tp % 64 || Tp_out % 64) return fail(QASR_ERR_ARG, "dw_conv_acc: bad arguments");
  void* z;
  int rc = zero_buf(&z);
  if (rc) return rc;
  DwP p{};
  p.x = x;
  p.w = w;
  p.bias = bias ? bias : (const int32_t*)z;   // u8 data: the caller's bias carries 128 * sum(w)
  p.C = c;
  p.K = kernel;
  p.kpad = kpad;
  p.stride = stride;
  p.dilation = dilation;
  p.padding = padding;
  p.T_in = T;
  p.Tp_in = Tp;
  p.x_unsigned = x_unsigned;
  p.e.sb = (const float*)z;
  p.e.acc_dbg = acc;
  p.e.T = T_out;
  p.e.Tp = Tp_out;
  p.e.cout = c;
  p.e.B = B;
  p.e.lens = (const int32_t*)z;
  launch_dw((hipStream_t)stream, p);
  HIPCHK(hipGetLastError());
  return QASR_OK;
}

int qasr_dense_conv_acc(void* stream, const int8_t* x, int x_unsigned, const int8_t* w, const int32_t* bias, int B, int cin,
                        int cin_pad, int cout, int kernel, int stride, int dilation, int padding, int T, int Tp, int T_out,
                        int Tp_out, int32_t* acc) {
  if (!x || !w || !acc || cin_pad % 128 || cin > cin_pad || Tp % 64 || Tp_out % 64 || T > Tp || T_out > Tp_out || kernel < 1 ||
      stride < 1 || dilation < 1 || B < 1 || cout < 1)
    return fail(QASR_ERR_ARG, "dense_conv_acc: bad arguments");
  void* z;
  int rc = zero_buf(&z);
  if (rc) return rc;
  if ((size_t)rup(cout, 128) * 8 > kZeroBytes) return fail(QASR_ERR_ARG, "cout too large");
  DenseP p{};                          // the generic dense conv kernel (k_dense) with no consumers: accumulators only
  p.x = x;
  p.w = w;
  p.bias = bias ? bias : (const int32_t*)z;
  p.cin = cin;
  p.cin_pad = cin_pad;
  p.K = kernel;
  p.stride = stride;
  p.dilation = dilation;
  p.padding = padding;
  p.T_in = T;
  p.Tp_in = Tp;
  p.x_unsigned = x_unsigned;
  p.e.sb = (const float*)z;
  p.e.acc_dbg = acc;
  p.e.T = T_out;
  p.e.Tp = Tp_out;
  p.e.cout = cout;
  p.e.B = B;
  p.e.lens = (const int32_t*)z;
  launch_dense((hipStream_t)stream, p);
  HIPCHK(hipGetLastError());
  return QASR_OK;
}

int qasr_sep_layer(void* stream, const qasr_sep_layer_args* a, char* label, size_t label_cap) {
  if (!a || !a->x || !a->w || !a->bias || !a->lens || a->B < 1 || a->cin < 1 || a->cout < 1 || a->Tp % 64 || a->T > a->Tp ||
      a->n_outs < 0 || a->n_outs > QASR_MAX_OUTS || (a->tile != 32 && a->tile != 64 && a->tile != 128))
    return fail(QASR_ERR_ARG, "sep_layer: bad arguments");
  SepP p{};
  p.x = a->x;
  p.wdw = a->wdw;
  p.wdw2 = a->wdw2;
  p.bias_dw = a->bias_dw;
  p.m_dw = a->m_dw;
  p.dw_acc_dbg = a->dw_acc;
  p.dw_lo = a->dw_lo;
  p.dw_hi = a->dw_hi;
  p.K = a->K;
  p.x_unsigned = a->K > 0 ? a->x_unsigned : 0;
  p.pw_unsigned = a->K > 0 ? 0 : a->x_unsigned;
  p.dilation = a->K > 0 ? a->dilation : 1;
  p.tile = a->tile;
  p.gen = a->gen;
  p.w = a->w;
  p.bias = a->bias;
  p.cin = a->cin;
  p.cin_pad = rup(a->cin, 128);
  if (a->flags & QASR_F_RESADD) {
    if (!a->rx || !a->rw || !a->rbias || !a->rm || !a->m_main || a->rcin < 1) return fail(QASR_ERR_ARG, "sep_layer: residual operands missing");
    p.n_panes = 1;
    PaneP& d = p.panes[0];
    d.x = a->rx;
    d.w = a->rw;
    d.bias = a->rbias;
    d.m = a->rm;
    d.sb = a->rsb;
    d.acc_dbg = a->racc;
    d.cin = a->rcin;
    d.cin_pad = rup(a->rcin, 128);
    d.x_unsigned = a->r_unsigned;
  }
  EpiP& e = p.e;
  e.n_outs = a->n_outs;
  for (int j = 0; j < a->n_outs; ++j) {
    e.outs[j].ptr = a->outs[j].ptr;
    e.outs[j].mtab = a->outs[j].mtab;
    e.outs[j].m = a->outs[j].m;
    e.outs[j].lo = a->outs[j].lo;
    e.outs[j].hi = a->outs[j].hi;
    e.outs[j].mode = a->outs[j].mode;
    if (!a->outs[j].ptr || (a->outs[j].mode == 1 && !a->outs[j].mtab)) return fail(QASR_ERR_ARG, "sep_layer: consumer %d incomplete", j);
  }
  e.flags = a->flags & (QASR_F_RELU | QASR_F_MASK_OUT | QASR_F_EXACT_Z | QASR_F_RESADD);
  if ((e.flags & QASR_F_EXACT_Z) && !a->sb) return fail(QASR_ERR_ARG, "sep_layer: EXACT_Z needs the conv output scales");
  e.sb = a->sb;
  e.m_main = a->m_main;
  e.lens = a->lens;
  e.acc_dbg = a->acc;
  e.qlo = a->qlo;
  e.qhi = a->qhi;
  e.T = a->T;
  e.Tp = a->Tp;
  e.cout = a->cout;
  e.B = a->B;
  if (label && label_cap >= 8) sep_kernel_label(p, label, label_cap);
  int rc = launch_sep((hipStream_t)stream, p);
  if (rc) return fail(rc, "sep_layer: no kernel instantiation for K=%d dilation=%d (or bad launch shape)", p.K, p.dilation);
  HIPCHK(hipGetLastError());
  return QASR_OK;
}

int qasr_requant(void* stream, const int32_t* acc, const double* m, const float* sb, int exact_z, int relu, int B, int c,
                 int Tp, int lo, int hi, int8_t* out) {
  if (!acc || !m || !out || (exact_z && !sb)) return fail(QASR_ERR_ARG, "requant: bad arguments");
  RequantP p{};
  p.in = acc;
  p.in_is_i32 = 1;
  p.n_outs = 1;
  p.outs[0].ptr = out;
  p.outs[0].mtab = m;
  p.outs[0].lo = lo;
  p.outs[0].hi = hi;
  p.outs[0].mode = 1;
  p.sb = sb;
  p.flags = (exact_z ? QASR_F_EXACT_Z : 0) | (relu ? QASR_F_RELU : 0);
  p.C = c;
  p.T = Tp;
  p.Tp = Tp;
  p.B = B;
  launch_requant((hipStream_t)stream, p);
  HIPCHK(hipGetLastError());
  return QASR_OK;
}

}  // extern "C"
