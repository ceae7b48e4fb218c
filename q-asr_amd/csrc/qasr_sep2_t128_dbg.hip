// k_sep2 instantiations with 128-frame tiles, accumulator dumps: plain layers in throughput mode (see qasr_sep2_impl.h)
#include "qasr_sep2_impl.h"

namespace qasr {
template int launch_sep2_inst<128, true>(hipStream_t, const SepP&);
}  // namespace qasr
