// k_sep2 instantiations with 64-frame tiles (see qasr_sep2_impl.h)
#include "qasr_sep2_impl.h"

namespace qasr {
template int launch_sep2_inst<64, false>(hipStream_t, const SepP&);
}  // namespace qasr
