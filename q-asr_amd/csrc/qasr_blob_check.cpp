// Host-only validation of a packed model ("blob", include/qasr.h) before anything of it reaches the device.
//
// The reference loads a checkpoint without checks (nemo/core/classes/modelPT.py:379-400: tar -> yaml -> torch.load); here the
// blob arrives as plain bytes - from a file or, on ranks > 0, over the RCCL broadcast (qasr/dist.py) - and every offset,
// index and shape in it ends up in a kernel argument.  qasr_blob_check accepts a blob only if every array an op refers
// to lies inside the data section with the extent the kernels read, every tensor / domain / pane index is in range, the
// tensor an op writes has the element size the op's output mode stores, and the time domains of an op's operands agree
// (a mismatch would make a kernel walk past the end of an arena buffer).  No HIP, no allocation proportional to the
// blob: this file is also built alone with `g++ -fsanitize=address,undefined` by tests/test_blob_check.py.
//
// All reads go through memcpy: the tables are only required to be 8-byte aligned relative to the blob, and the caller's
// buffer may have any alignment.
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/qasr.h"

namespace {

struct Ctx {
  const unsigned char* p;
  size_t n;
  qasr_blob_header h;
  uint64_t data_bytes;            // total_bytes - data_off
  char* err;
  size_t cap;
};

int bad(const Ctx& c, const char* fmt, ...) {
  if (c.err && c.cap) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c.err, c.cap, fmt, ap);
    va_end(ap);
  }
  return QASR_ERR_BLOB;
}

template <class T>
T load(const Ctx& c, uint64_t off) {
  T v;
  memcpy(&v, c.p + off, sizeof v);
  return v;
}

inline uint64_t rup(uint64_t x, uint64_t m) { return (x + m - 1) / m * m; }

// data-section array at offset `off` (relative to data_off; 0 = absent) of `bytes` bytes
bool array_ok(const Ctx& c, uint64_t off, uint64_t bytes, bool required) {
  if (off == 0) return !required;
  if (off & 15) return false;                                   // the packer aligns every array to 16 bytes
  return off < c.data_bytes && bytes <= c.data_bytes - off;
}

bool table_ok(const Ctx& c, uint64_t off, uint64_t count, uint64_t rec, uint64_t end_limit) {
  if (off & 7) return false;
  if (off < sizeof(qasr_blob_header) || off > end_limit) return false;
  return count <= (end_limit - off) / rec;
}

const uint32_t MAX_CH = 1u << 16, MAX_TAPS = 1u << 10, MAX_SPAN = 1u << 12;

}  // namespace

extern "C" int qasr_blob_check(const void* blob, size_t n, char* err, size_t err_cap) {
  Ctx c{(const unsigned char*)blob, n, {}, 0, err, err_cap};
  if (err && err_cap) err[0] = 0;
  if (!blob || n < sizeof(qasr_blob_header)) return bad(c, "null / short blob (%zu bytes)", n);
  memcpy(&c.h, blob, sizeof c.h);
  const qasr_blob_header& h = c.h;
  if (h.magic != QASR_BLOB_MAGIC || h.version != QASR_BLOB_VERSION) return bad(c, "bad magic / version");
  if (h.total_bytes != n) return bad(c, "size mismatch: blob %zu vs header %llu", n, (unsigned long long)h.total_bytes);
  if (h.reserved != sizeof(qasr_op_desc)) return bad(c, "op record %u bytes vs %zu", h.reserved, sizeof(qasr_op_desc));
  if (h.data_off > n || (h.data_off & 15) || h.data_off < sizeof(qasr_blob_header)) return bad(c, "data_off out of range");
  if (!h.n_tensors || !h.n_ops || !h.n_domains || h.n_tensors > (1u << 20) || h.n_ops > (1u << 20) || h.n_domains > (1u << 10))
    return bad(c, "table counts out of range (%u tensors, %u ops, %u domains)", h.n_tensors, h.n_ops, h.n_domains);
  if (!table_ok(c, h.tensors_off, h.n_tensors, sizeof(qasr_tensor_desc), h.data_off) ||
      !table_ok(c, h.ops_off, h.n_ops, sizeof(qasr_op_desc), h.data_off) ||
      !table_ok(c, h.domains_off, h.n_domains, sizeof(qasr_domain_desc), h.data_off))
    return bad(c, "table offsets out of range");
  // the three tables may not overlap each other
  {
    const uint64_t b[3] = {h.tensors_off, h.ops_off, h.domains_off};
    const uint64_t e[3] = {h.tensors_off + (uint64_t)h.n_tensors * sizeof(qasr_tensor_desc),
                           h.ops_off + (uint64_t)h.n_ops * sizeof(qasr_op_desc),
                           h.domains_off + (uint64_t)h.n_domains * sizeof(qasr_domain_desc)};
    for (int i = 0; i < 3; ++i)
      for (int j = i + 1; j < 3; ++j)
        if (b[i] < e[j] && b[j] < e[i]) return bad(c, "tables overlap");
  }
  c.data_bytes = h.total_bytes - h.data_off;
  if (h.feat_in < 1 || h.feat_in > MAX_CH || h.n_classes < 2 || h.n_classes > MAX_CH) return bad(c, "feat_in / n_classes out of range");
  if (h.weight_bit < 2 || h.weight_bit > 8 || h.act_bit < 2 || h.act_bit > 8) return bad(c, "bit widths out of range");

  // ---- time domains
  for (uint32_t d = 0; d < h.n_domains; ++d) {
    const auto q = load<qasr_domain_desc>(c, h.domains_off + (uint64_t)d * sizeof(qasr_domain_desc));
    if (d == 0) {
      if (q.parent != -1) return bad(c, "domain 0 has a parent");
      continue;
    }
    if (q.parent < 0 || q.parent >= (int32_t)d) return bad(c, "domain %u: parent %d", d, q.parent);
    if (q.kernel < 1 || q.kernel > MAX_TAPS || q.stride < 1 || q.stride > 64 || q.dilation < 1 || q.dilation > 64 ||
        (uint64_t)q.kernel * q.dilation > MAX_SPAN || q.padding > MAX_SPAN)
      return bad(c, "domain %u: conv geometry out of range", d);
  }
  auto tensor = [&](uint32_t i) { return load<qasr_tensor_desc>(c, h.tensors_off + (uint64_t)i * sizeof(qasr_tensor_desc)); };
  auto domain = [&](uint32_t i) { return load<qasr_domain_desc>(c, h.domains_off + (uint64_t)i * sizeof(qasr_domain_desc)); };

  // ---- tensors
  for (uint32_t i = 0; i < h.n_tensors; ++i) {
    const qasr_tensor_desc t = tensor(i);
    if (t.channels < 1 || t.channels > MAX_CH || t.dtype > QASR_DT_I32 || t.domain >= h.n_domains)
      return bad(c, "tensor %u: channels / dtype / domain out of range", i);
    if (t.last_use < -1 || t.last_use >= (int32_t)h.n_ops) return bad(c, "tensor %u: last_use %d", i, t.last_use);
    if (i == 0) {                                               // the caller's feature buffer
      if (t.producer != -1 || t.dtype != QASR_DT_F32 || t.channels != h.feat_in || t.domain != 0)
        return bad(c, "tensor 0 is not the f32 [feat_in] network input");
    } else if (t.producer < 0 || t.producer >= (int32_t)h.n_ops) {
      return bad(c, "tensor %u: producer %d (every arena tensor needs one)", i, t.producer);
    }
  }

  // ---- ops
  bool seen_logits = false;
  for (uint32_t oi = 0; oi < h.n_ops; ++oi) {
    const qasr_op_desc op = load<qasr_op_desc>(c, h.ops_off + (uint64_t)oi * sizeof(qasr_op_desc));
    if (op.kind > QASR_OP_REQUANT) return bad(c, "op %u: kind %u", oi, op.kind);
    if (op.n_panes > QASR_MAX_PANES) return bad(c, "op %u: %u panes", oi, op.n_panes);
    if (op.in < 0 || op.in >= (int32_t)h.n_tensors) return bad(c, "op %u: input tensor %d", oi, op.in);
    const bool conv = op.kind == QASR_OP_DW || op.kind == QASR_OP_PW || op.kind == QASR_OP_DENSE;
    if (op.cin < 1 || op.cin > MAX_CH || op.cout < 1 || op.cout > MAX_CH || op.kernel < 1 || op.kernel > MAX_TAPS ||
        op.stride < 1 || op.stride > 64 || op.dilation < 1 || op.dilation > 64 || (uint64_t)op.kernel * op.dilation > MAX_SPAN ||
        op.padding > MAX_SPAN)
      return bad(c, "op %u: geometry out of range", oi);
    if (!conv && op.n_panes) return bad(c, "op %u: panes on a non-conv op", oi);
    if ((op.flags & QASR_F_RESADD) ? (!conv || op.kind == QASR_OP_DW || op.n_panes == 0) : op.n_panes != 0)
      return bad(c, "op %u: RESADD flag and pane count disagree", oi);
    const qasr_tensor_desc tin = tensor((uint32_t)op.in);
    if (tin.channels != op.cin) return bad(c, "op %u: cin %u vs input tensor's %u channels", oi, op.cin, tin.channels);
    if (op.in == 0 ? (op.kind != QASR_OP_QUANT_IN || oi != 0) : (tin.producer >= (int32_t)oi))
      return bad(c, "op %u: input tensor %d is not produced before it", oi, op.in);
    if (tin.last_use < (int32_t)oi) return bad(c, "op %u: input tensor %d is dead (last_use %d)", oi, op.in, tin.last_use);
    if (op.kind == QASR_OP_DW && op.cin != op.cout) return bad(c, "op %u: depthwise cin != cout", oi);
    const bool in_f32 = op.kind == QASR_OP_QUANT_IN || op.kind == QASR_OP_LOGSOFTMAX;
    if (in_f32 != (tin.dtype == QASR_DT_F32)) return bad(c, "op %u: input dtype %u", oi, tin.dtype);
    if (conv && tin.dtype == QASR_DT_I32) return bad(c, "op %u: conv over an int32 tensor", oi);
    if (op.kind == QASR_OP_LOGSOFTMAX && (op.cin != h.n_classes || op.cout != h.n_classes)) return bad(c, "op %u: log-softmax width", oi);

    // arrays of the op
    const uint64_t cp = rup(op.cout, 128), cinp = rup(op.cin, 128), kp = rup(op.kernel, 4);
    const bool w6 = (op.flags & QASR_F_W6PACK) != 0;
    if (w6 && !conv) return bad(c, "op %u: W6PACK on a non-conv op", oi);
    auto wbytes = [&](uint64_t expanded) { return w6 ? expanded / 4 * 3 : expanded; };
    if (conv) {
      const uint64_t wn = op.kind == QASR_OP_DW ? (uint64_t)op.cout * kp : cp * cinp * (op.kind == QASR_OP_DENSE ? op.kernel : 1);
      if (!array_ok(c, op.w_off, wbytes(wn), true)) return bad(c, "op %u: weight array out of range", oi);
      if (!array_ok(c, op.bias_off, cp * 4, false)) return bad(c, "op %u: bias array out of range", oi);
      if (!array_ok(c, op.sb_off, cp * 4, (op.flags & (QASR_F_EXACT_Z | QASR_F_LOGITS)) != 0)) return bad(c, "op %u: scale array out of range", oi);
      if (op.flags & QASR_F_RESADD) {
        if (!array_ok(c, op.m_off, cp * 8, true)) return bad(c, "op %u: res_act multiplier array out of range", oi);
        if (op.qlo > op.qhi) return bad(c, "op %u: res_act clamp", oi);
      } else if (op.kind == QASR_OP_DW) {
        if (!array_ok(c, op.m_off, (uint64_t)op.cout * (kp + 32), !w6)) return bad(c, "op %u: tap-row array out of range", oi);
      } else if (op.m_off) {
        return bad(c, "op %u: stray m_off", oi);
      }
    } else {
      if (op.w_off || op.bias_off || op.m_off) return bad(c, "op %u: stray weight offsets", oi);
      if (!array_ok(c, op.sb_off, cp * 4, (op.flags & QASR_F_EXACT_Z) != 0)) return bad(c, "op %u: scale array out of range", oi);
    }

    // outputs: element size the mode stores vs the tensor's, channel count, domain
    int out_domain = -1;
    int n_outs = 0;
    for (int j = 0; j < QASR_MAX_OUTS; ++j) {
      const qasr_out& o = op.outs[j];
      if (o.tensor == -1) continue;
      if (o.tensor < 1 || o.tensor >= (int32_t)h.n_tensors) return bad(c, "op %u out %d: tensor %d", oi, j, o.tensor);
      if (j != n_outs) return bad(c, "op %u: outputs are not packed from slot 0", oi);
      ++n_outs;
      const qasr_tensor_desc t = tensor((uint32_t)o.tensor);
      if (t.producer != (int32_t)oi) return bad(c, "op %u out %d: tensor %d names producer %d", oi, j, o.tensor, t.producer);
      if (t.channels != op.cout) return bad(c, "op %u out %d: %u channels vs cout %u", oi, j, t.channels, op.cout);
      if (o.mode > 3) return bad(c, "op %u out %d: mode %u", oi, j, o.mode);
      const bool logits = conv && (op.flags & QASR_F_LOGITS);
      const uint32_t want_bytes = o.mode == 3 ? 4 : 1;
      const uint32_t have_bytes = (t.dtype == QASR_DT_F32 || t.dtype == QASR_DT_I32) ? 4 : 1;
      if (want_bytes != have_bytes || (t.dtype == QASR_DT_F32) != logits)
        return bad(c, "op %u out %d: mode %u into a dtype-%u tensor", oi, j, o.mode, t.dtype);
      if (logits && (j != 0 || o.mode != 3 || op.cout != h.n_classes)) return bad(c, "op %u: logits output", oi);
      if (o.mode <= 1 && o.lo > o.hi) return bad(c, "op %u out %d: clamp %d > %d", oi, j, o.lo, o.hi);
      if (o.mode <= 1 && (o.lo < -256 || o.hi > 255)) return bad(c, "op %u out %d: clamp beyond a byte", oi, j);
      if (!array_ok(c, o.m_off, cp * 8, o.mode == 1)) return bad(c, "op %u out %d: multiplier table out of range", oi, j);
      if (o.mode == 0 && !(o.m == o.m)) return bad(c, "op %u out %d: NaN multiplier", oi, j);
      if (out_domain >= 0 && (int)t.domain != out_domain) return bad(c, "op %u: outputs in different domains", oi);
      out_domain = (int)t.domain;
    }
    if (op.kind == QASR_OP_LOGSOFTMAX) {
      if (n_outs) return bad(c, "op %u: log-softmax with stored outputs", oi);
      if (oi + 1 != h.n_ops) return bad(c, "op %u: log-softmax is not the last op", oi);
      continue;
    }
    if (!n_outs) return bad(c, "op %u: no output", oi);
    if (conv && (op.flags & QASR_F_LOGITS)) seen_logits = true;
    // time domains: same frames in and out, or the domain this very conv opens
    const bool same_len = !conv || (op.stride == 1 && 2 * (uint64_t)op.padding == (uint64_t)op.dilation * (op.kernel - 1));
    if (same_len) {
      if (out_domain != (int)tin.domain) return bad(c, "op %u: output domain %d vs input domain %u", oi, out_domain, tin.domain);
    } else {
      const qasr_domain_desc d = domain((uint32_t)out_domain);
      if (d.parent != (int32_t)tin.domain || d.kernel != op.kernel || d.stride != op.stride || d.dilation != op.dilation ||
          d.padding != op.padding)
        return bad(c, "op %u: output domain %d is not the one this conv opens", oi, out_domain);
    }
    // residual panes (1x1 convs over earlier tensors, all in the output's domain)
    for (uint32_t k = 0; k < op.n_panes; ++k) {
      const qasr_pane& s = op.panes[k];
      if (s.in < 1 || s.in >= (int32_t)h.n_tensors) return bad(c, "op %u pane %u: tensor %d", oi, k, s.in);
      const qasr_tensor_desc t = tensor((uint32_t)s.in);
      if (t.channels != s.cin || s.cin < 1) return bad(c, "op %u pane %u: cin %u vs %u channels", oi, k, s.cin, t.channels);
      if (t.dtype != QASR_DT_S8 && t.dtype != QASR_DT_U8) return bad(c, "op %u pane %u: input dtype %u", oi, k, t.dtype);
      if (t.producer >= (int32_t)oi || t.last_use < (int32_t)oi) return bad(c, "op %u pane %u: tensor %d not live here", oi, k, s.in);
      if ((int)t.domain != out_domain) return bad(c, "op %u pane %u: domain %u vs %d", oi, k, t.domain, out_domain);
      if (!array_ok(c, s.w_off, wbytes(cp * rup(s.cin, 128)), true) || !array_ok(c, s.bias_off, cp * 4, false) ||
          !array_ok(c, s.m_off, cp * 8, true) || !array_ok(c, s.sb_off, cp * 4, (op.flags & QASR_F_EXACT_Z) != 0))
        return bad(c, "op %u pane %u: array out of range", oi, k);
    }
  }
  if (!seen_logits) return bad(c, "no decoder (LOGITS) op");
  const qasr_op_desc last = load<qasr_op_desc>(c, h.ops_off + (uint64_t)(h.n_ops - 1) * sizeof(qasr_op_desc));
  if (last.kind != QASR_OP_LOGSOFTMAX) return bad(c, "the last op is not log-softmax");
  return QASR_OK;
}
