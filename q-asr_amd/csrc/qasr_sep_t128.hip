// k_sep instantiations with 128-frame tiles: Jasper's dense k > 1 convs with the plain epilogue only (each weight
// fragment then feeds four frame tiles; two such launches of different steps share the chip) - see qasr_sep_impl.h
#include "qasr_sep_impl.h"

namespace qasr {
int launch_sep_dense128(hipStream_t s, const SepP& p) {
  const bool dbg = p.e.acc_dbg != nullptr;
  return dbg ? launch_sep_v<0, 1, EP_PLAIN, true, 128>(s, p) : launch_sep_v<0, 1, EP_PLAIN, false, 128>(s, p);
}
}  // namespace qasr
