// k_sep: host-side routing of the fused separable-layer kernel (qasr_sep_impl.h holds the kernel; the
// instantiations live in qasr_sep_t{32,64}{,_dbg}.hip so that they compile in parallel).
#include "qasr_sep_impl.h"
#include "qasr_sep2_impl.h"

namespace qasr {

long long* g_prof = nullptr;
int g_prof_mode = 0;
int g_prof_cap = 0;

extern template int launch_sep_inst<32, false>(hipStream_t, const SepP&);
extern template int launch_sep_inst<32, true>(hipStream_t, const SepP&);
extern template int launch_sep_inst<64, false>(hipStream_t, const SepP&);
extern template int launch_sep_inst<64, true>(hipStream_t, const SepP&);
extern template int launch_sep2_inst<32, false>(hipStream_t, const SepP&);
extern template int launch_sep2_inst<32, true>(hipStream_t, const SepP&);
extern template int launch_sep2_inst<64, false>(hipStream_t, const SepP&);
extern template int launch_sep2_inst<64, true>(hipStream_t, const SepP&);
extern template int launch_sep2_inst<128, false>(hipStream_t, const SepP&);
extern template int launch_sep2_inst<128, true>(hipStream_t, const SepP&);

// k_sep2 takes the stride-1 separable layers it is built for (sep2_shape_ok) unless the engine was told to stay on k_sep
bool sep2_takes(const SepP& p) { return p.gen == 2 && sep2_shape_ok(p); }
// frames per work-group of a k_sep2 launch: 128 for the depthwise-separable layers when the engine asks for it
// (throughput mode: every weight fragment then feeds four frame tiles, B * Tp / 128 work-groups per launch), else the
// engine's 32 / 64.  The block-end (residual) layers too: with the decoder and the stem fused it is 0.417 vs 0.424 ms per step
// and steadier (four 64-work-group launches fit the chip side by side, four 128-work-group ones queue);
// qasr_engine_opts.res_tile128 = 0 keeps them on 64 (the engine then hands those ops over with tile = 64)
static int sep2_tile(const SepP& p) {
  if (p.tile == 128) return (p.K > 0 && p.e.Tp % 128 == 0) ? 128 : 64;
  if (p.K > 0 && p.dilation == 2) return 64;                 // the dilation-2 form has no 32-frame instantiation (16 samples per parity)
  return p.tile == 64 ? 64 : 32;
}

int launch_sep_dense128(hipStream_t s, const SepP& p);   // qasr_sep_t128.hip

bool sep_supported(int K, int dilation) {
  if (dilation == 2) return K == 87 || K == 15;
  return dilation == 1 && (K == 0 || K == 11 || K == 13 || K == 33 || K == 39 || K == 51 || K == 63 || K == 75);
}

// template arguments of the k_sep instantiation launch_sep picks for `p` (as rocprofv3 prints them)
void sep_kernel_label(const SepP& p, char* buf, size_t cap) {
  const bool dbg = p.e.acc_dbg || p.dw_acc_dbg;
  if (dense2_takes(p)) {
    dense2_label(p, buf, cap);
    return;
  }
  if (sep2_takes(p)) {
    snprintf(buf, cap, "k_sep2<%d, %d, %d, %d, %s, %d, %d>", p.K, p.cin_pad >> 7,
             (p.e.flags & QASR_F_RESADD) ? p.panes[0].cin_pad >> 7 : 0, (p.e.cout + 255) / 256, dbg ? "true" : "false", sep2_tile(p),
             p.K > 0 ? p.dilation : 1);
    return;
  }
  snprintf(buf, cap, "k_sep<%d, %d, %d, %s, %d>", p.K, p.K > 0 ? p.dilation : 1, sep_epilogue_class(p), dbg ? "true" : "false",
           sep_tile_for(p));
}

// The generic epilogue (logits, raw int32 outs, several residual panes) does not fit the register file with two frame
// tiles per wave (hundreds of spilled VGPRs): 1x1 ops of that class run with 32-frame tiles (QuartzNet decoder 32 -> 17 us).
// Dense k > 1 convs keep the wide tile: halving their weight traffic outweighs the spills (Jasper 9.7 vs 9.9 ms).
int sep_tile_for(const SepP& p) {
  if (p.tile == 128 && p.dense_k > 1 && sep_epilogue_class(p) == EP_PLAIN) return 128;   // Jasper's plain dense convs
  return (p.tile >= 64 && (p.dense_k > 1 || sep_epilogue_class(p) != EP_GENERIC)) ? 64 : 32;
}

int launch_sep(hipStream_t s, const SepP& p) {
  const bool dbg = p.e.acc_dbg || p.dw_acc_dbg;
  if (dense2_takes(p)) return launch_dense2(s, p);           // Jasper's plain dense convs (qasr_dense2.hip)
  if (sep2_takes(p)) {
    const int tt = sep2_tile(p);
    if (tt == 128) return dbg ? launch_sep2_inst<128, true>(s, p) : launch_sep2_inst<128, false>(s, p);
    if (tt == 64) return dbg ? launch_sep2_inst<64, true>(s, p) : launch_sep2_inst<64, false>(s, p);
    return dbg ? launch_sep2_inst<32, true>(s, p) : launch_sep2_inst<32, false>(s, p);
  }
  if (!sep_supported(p.K, p.K > 0 ? p.dilation : 1)) return QASR_ERR_UNSUPPORTED;
  if (sep_tile_for(p) == 128) return launch_sep_dense128(s, p);
  if (sep_tile_for(p) == 64) return dbg ? launch_sep_inst<64, true>(s, p) : launch_sep_inst<64, false>(s, p);
  return dbg ? launch_sep_inst<32, true>(s, p) : launch_sep_inst<32, false>(s, p);
}

}  // namespace qasr
