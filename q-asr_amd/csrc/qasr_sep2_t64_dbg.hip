// k_sep2 instantiations with 64-frame tiles, accumulator dumps (see qasr_sep2_impl.h)
#include "qasr_sep2_impl.h"

namespace qasr {
template int launch_sep2_inst<64, true>(hipStream_t, const SepP&);
}  // namespace qasr
