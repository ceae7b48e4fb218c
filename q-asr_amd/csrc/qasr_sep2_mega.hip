// k_sep2_mega: a run of consecutive separable layers as ONE persistent launch (qasr_engine_opts.persistent).
//
// Work-group = one utterance: it walks the layers of the run and, per layer, the utterance's time tiles of 128 frames
// with k_sep2's work-group body (qasr_sep2_impl.h: sep2_body).  Every tensor between two layers is read only by the
// work-group that wrote it (an utterance's rows of the [B][C][Tp] tensors: depthwise halos and residual operands stay
// inside the utterance), so there is NO exchange between work-groups and no grid-wide synchronisation: a layer boundary
// is one work-group barrier behind the stores (the CU's own L1 is coherent for its own work-group).  What this removes
// is the kernel boundary (~3.3 us per layer: dispatch -> first wave, last wave -> completion; 24 % of a QuartzNet
// forward's serial device time, profiles/r02_v9_wg_timeline_128.txt) and the idle tail of every launch; what it costs is
// parallelism inside one step: a B-utterance launch occupies B CUs, so the mode is for many steps in flight (8 launches
// of 32 work-groups run side by side with GPU_MAX_HW_QUEUES=8: profiles/microbench/concur.hip).
// Results are identical to the per-layer launches: same body, same operands, same order of integer operations.
#include "qasr_sep2_impl.h"

namespace qasr {

template <int TT>
__global__ void __launch_bounds__(SEP2_NT, SEP2_WPE) k_sep2_mega(const MegaOp* __restrict__ ops, int n_ops) {
  const int b = blockIdx.x;
  for (int oi = 0; oi < n_ops; ++oi) {
    const MegaOp& op = ops[oi];                              // wave-uniform address: scalar loads
    const int shape = __builtin_amdgcn_readfirstlane(op.shape);
    const int n_tiles = __builtin_amdgcn_readfirstlane(op.p.e.Tp) / TT;
    for (int tile = 0; tile < n_tiles; ++tile) {
      switch (shape) {
#define SEP2_MEGA_CASE(K_, NG_, NGP_, NP_)                                                  \
        case sep2_shape_index(K_, NG_, NGP_, NP_):                                            \
          if constexpr (K_ > 0) sep2_body<K_, NG_, NGP_, NP_, false, TT>(op.p, b, tile * TT, false, 0); \
          break;
        SEP2_INSTANCES(SEP2_MEGA_CASE)
#undef SEP2_MEGA_CASE
        default: break;
      }
      // the next tile (or layer) overwrites the LDS images every wave has just read, and reads - through other waves -
      // what this one stored: work-group barrier behind the stores (release / acquire at work-group scope)
      __syncthreads();
    }
  }
}

// shape id of an op the persistent kernel can run, -1 otherwise (k_sep2 shape with taps, 128-frame tiles)
int sep2_mega_shape(const SepP& p) {
  if (p.gen != 2 || !sep2_shape_ok(p) || p.K <= 0 || p.tile != 128 || p.e.Tp % 128 || p.e.acc_dbg || p.dw_acc_dbg) return -1;
  const int ng = p.cin_pad >> 7, ngp = (p.e.flags & QASR_F_RESADD) ? (p.panes[0].cin_pad >> 7) : 0, np = (p.e.cout + 255) / 256;
  int id = -1;
#define SEP2_MEGA_ID(K_, NG_, NGP_, NP_) \
  if (p.K == K_ && ng == NG_ && ngp == NGP_ && np == NP_) id = sep2_shape_index(K_, NG_, NGP_, NP_);
  SEP2_INSTANCES(SEP2_MEGA_ID)
#undef SEP2_MEGA_ID
  return id;
}

size_t sep2_mega_smem(const SepP& p) {
  size_t n = 0;
#define SEP2_MEGA_SMEM(K_, NG_, NGP_, NP_) \
  if constexpr (K_ > 0) { if (p.K == K_) n = sep2_smem_bytes<K_, 128>(p); }
  SEP2_INSTANCES(SEP2_MEGA_SMEM)
#undef SEP2_MEGA_SMEM
  return n;
}

int launch_sep2_mega(hipStream_t s, const MegaOp* dev_ops, int n_ops, int B, size_t smem) {
  if (!dev_ops || n_ops < 1 || B < 1 || smem > 160 * 1024) return QASR_ERR_ARG;
  static int attr_dev = -1;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (attr_dev != dev) {
    (void)hipFuncSetAttribute((const void*)k_sep2_mega<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_dev = dev;
  }
  hipLaunchKernelGGL((k_sep2_mega<128>), dim3(B), dim3(SEP2_NT), smem, s, dev_ops, n_ops);
  return QASR_OK;
}

}  // namespace qasr
