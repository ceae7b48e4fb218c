// k_sep instantiations with 64-frame tiles, accumulator dumps for the parity hooks (see qasr_sep_impl.h)
#include "qasr_sep_impl.h"

namespace qasr {
template int launch_sep_inst<64, true>(hipStream_t, const SepP&);
}  // namespace qasr
