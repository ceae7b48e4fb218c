// k_sep2 instantiations with 32-frame tiles (see qasr_sep2_impl.h)
#include "qasr_sep2_impl.h"

namespace qasr {
template int launch_sep2_inst<32, false>(hipStream_t, const SepP&);
}  // namespace qasr
