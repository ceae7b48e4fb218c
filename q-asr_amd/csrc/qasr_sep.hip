// k_sep: host-side routing of the fused separable-layer kernel (qasr_sep_impl.h holds the kernel; the
// instantiations live in qasr_sep_t{32,64}{,_dbg}.hip so that they compile in parallel).
#include "qasr_sep_impl.h"

namespace qasr {

long long* g_prof = nullptr;

extern template void launch_sep_inst<32, false>(hipStream_t, const SepP&);
extern template void launch_sep_inst<32, true>(hipStream_t, const SepP&);
extern template void launch_sep_inst<64, false>(hipStream_t, const SepP&);
extern template void launch_sep_inst<64, true>(hipStream_t, const SepP&);

bool sep_supported(int K, int dilation) {
  if (dilation == 2) return K == 87 || K == 15;
  return dilation == 1 && (K == 0 || K == 11 || K == 13 || K == 33 || K == 39 || K == 51 || K == 63 || K == 75);
}

// template arguments of the k_sep instantiation launch_sep picks for `p` (as rocprofv3 prints them)
void sep_kernel_label(const SepP& p, char* buf, size_t cap) {
  const bool dbg = p.e.acc_dbg || p.dw_acc_dbg;
  snprintf(buf, cap, "k_sep<%d, %d, %d, %s, %d>", p.K, p.K > 0 ? p.dilation : 1, sep_epilogue_class(p), dbg ? "true" : "false",
           p.tile == 64 ? 64 : 32);
}

void launch_sep(hipStream_t s, const SepP& p) {
  const bool dbg = p.e.acc_dbg || p.dw_acc_dbg;
  if (p.tile == 64) {
    if (dbg) launch_sep_inst<64, true>(s, p);
    else launch_sep_inst<64, false>(s, p);
  } else {
    if (dbg) launch_sep_inst<32, true>(s, p);
    else launch_sep_inst<32, false>(s, p);
  }
}

}  // namespace qasr
