// k_sep instantiations with 32-frame tiles (see qasr_sep_impl.h)
#include "qasr_sep_impl.h"

namespace qasr {
template int launch_sep_inst<32, false>(hipStream_t, const SepP&);
}  // namespace qasr
