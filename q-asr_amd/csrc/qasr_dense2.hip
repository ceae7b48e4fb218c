// k_dense2: Jasper's dense (non-separable) stride-1 'same' convs without residual panes (jasper.py:601-630: MaskedConv1d
// -> BatchNorm (folded) -> ReLU), as taps-shifted 1x1 GEMMs on v_mfma_i32_32x32x32_i8 - the second-generation kernel for
// the ops round 1's k_sep<0, 1, EP_PLAIN> ran (QuantConv1d.int_conv, quant_modules.py:301-305; the consumers' QuantAct
// requantisation fixedpoint_mul, quant_utils.py:187-216, fused into the epilogue).
//
// Why a new decomposition.  k_sep's work-group is (utterance, 128 frames, EVERY output channel): each weight fragment a
// wave loads feeds 4 MFMAs, a work-group streams the whole layer's weights (14.7 MB for 768 -> 768, k = 25) - 32 B/clk of
// the CU's ~50 B/clk L2 -> register path at full matrix rate - and the generic kernel (256 VGPRs, 50 spilled) has no
// registers left to read A fragments ahead: 0.46 of the int8 MFMA peak with the chip filled.  Here
//   work-group = 256 threads = (utterance, 128 output channels, up to 256 frames); wave = 32 channels x up to 8 frame
//   tiles of 32: every weight fragment feeds 8 MFMAs (16 B/clk per CU), 128 accumulators per lane;
//   the window is staged per chunk of 128 input channels as Xs[frame][channel] (K contiguous: a tap shift is a row shift,
//   every A fragment an aligned ds_read_b128), 46 KB: two or three work-groups per CU, one staging while the other multiplies;
//   A fragments are read one K step ahead, weight fragments two (tap, chunk) items ahead.
// Grid (B, cout_pad / 128, ceil(Tp / (32 MT))), utterance fastest (an utterance's rows stay in one XCD's L2; consecutive
// work-groups share the weights of their channel block).
#include "qasr_sep2_impl.h"

namespace qasr {

#define DENSE2_NT 256
#define DENSE2_CK 128                   /* input channels per staged chunk */
#define DENSE2_XP (DENSE2_CK + 16)      /* LDS row pitch: 16-byte aligned rows, rows 36 banks apart */


// rows [tf, tf + rows) x channels [c0, c0 + 128) of x[B][cin][Tp] -> Xs[row][channel] (4 x 4 byte transposes; a task = 16
// frames of 4 channels: four 16-byte loads, sixteen dword stores).  Frames outside [0, Tp) and channels >= cin read as code
// 0 (conv zero padding; for u8 tensors the -128 flip makes that -128 everywhere, which the packer's +128 * sum(W) bias
// assumes).  Lane map: 4 consecutive lanes take 4 consecutive 16-frame granules of the same channels (64 contiguous bytes
// per row), the 16 lane quads of a wave 16 consecutive channel quads (64 contiguous LDS bytes per row: the 4 granules of a
// quad fall on the same banks - a 4-way conflict on ds_write_b32 costs 2x, 20 lanes on one bank would cost 10x).  Every load
// of a chunk is issued before the first transpose.
template <int MT>
__device__ __forceinline__ void dense2_stage(lds_u8* Xs, const int8_t* __restrict__ x, int cin, int Tp, int b, int c0, int tf, int rows,
                                             unsigned flip) {
  constexpr int MAXROWS = 32 * MT + 64, NTASK = (DENSE2_CK / 4) * ((MAXROWS / 16 + 3) / 4 * 4);
  constexpr int NIT = (NTASK + DENSE2_NT - 1) / DENSE2_NT;
  const int n16 = rows >> 4;
  v4i r[NIT][4];
  int cqs[NIT], tqs[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int task = threadIdx.x + DENSE2_NT * it;
    const int cq = (task >> 2) & 31, tq = 4 * (task >> 7) + (task & 3);
    cqs[it] = cq;
    tqs[it] = tq;
    const int t = tf + 16 * tq;
    const bool t_ok = tq < n16 && t >= 0 && t < Tp;           // a 16-frame granule lies entirely inside or outside
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ci = c0 + 4 * cq + j;
      r[it][j] = (ci < cin && t_ok) ? *(const v4i*)(x + ((size_t)b * cin + ci) * Tp + t) : (v4i){0, 0, 0, 0};
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    if (tqs[it] >= n16) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {                             // frames 4 q .. 4 q + 3 of the granule
      const unsigned r0 = r[it][0][q], r1 = r[it][1][q], r2 = r[it][2][q], r3 = r[it][3][q];
      const unsigned lo01 = __builtin_amdgcn_perm(r1, r0, 0x05010400u), hi01 = __builtin_amdgcn_perm(r1, r0, 0x07030602u);
      const unsigned lo23 = __builtin_amdgcn_perm(r3, r2, 0x05010400u), hi23 = __builtin_amdgcn_perm(r3, r2, 0x07030602u);
      lds_u8* dst = Xs + (16 * tqs[it] + 4 * q) * DENSE2_XP + 4 * cqs[it];
      *(lds_u32*)(dst) = __builtin_amdgcn_perm(lo23, lo01, 0x05040100u) ^ flip;
      *(lds_u32*)(dst + DENSE2_XP) = __builtin_amdgcn_perm(lo23, lo01, 0x07060302u) ^ flip;
      *(lds_u32*)(dst + 2 * DENSE2_XP) = __builtin_amdgcn_perm(hi23, hi01, 0x05040100u) ^ flip;
      *(lds_u32*)(dst + 3 * DENSE2_XP) = __builtin_amdgcn_perm(hi23, hi01, 0x07060302u) ^ flip;
    }
  }
}

// RES: block-end layers (jasper.py:664-682) - after the main conv, every residual pane (a 1x1 conv of an earlier block's
// output, dense residual: up to 10 of them) is multiplied in the same work-group and added with res_act's clamp after
// every add; 128 frames per work-group (MT = 4: main and pane accumulators side by side)
template <int MT, bool DBG, bool RES>
__global__ void __launch_bounds__(DENSE2_NT, 2) k_dense2(SepP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  lds_u8* const Xs = (lds_u8*)smem;
  const EpiP& e = p.e;
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, r31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x, t0 = blockIdx.z * (32 * MT);
  // diagnostics (qasr_debug_timeline): every work-group stamps start / end (100 MHz) and its shader cycles: the clock the
  // chip holds under this kernel = cycles / (end - start) x 100 MHz
  const int wg_id = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  const bool tline = p.prof && (p.prof_mode & 255) == 1 && tid == 0 && wg_id < p.prof_cap;
  long long tl_start = 0, tl_clk = 0;
  if (tline) {
    tl_start = (long long)__builtin_amdgcn_s_memrealtime();
    tl_clk = (long long)__builtin_amdgcn_s_memtime();
  }
  const int co_row = 128 * blockIdx.y + 32 * wave;           // this wave's 32 output channels (cout_pad is a multiple of 128)
  const int co = co_row + r31;                               // MFMA C layout: channel = lane & 31
  const int K = p.dense_k, dil = p.dilation, pad = dil * (K - 1) / 2;
  const int halo = (pad + 15) & ~15, rows = 32 * MT + 2 * halo;
  const int eT = e.T, eTp = e.Tp, ecout = e.cout, n_outs = e.n_outs, cin_pad = p.cin_pad;
  const int cout_pad = (ecout + 127) & ~127;
  const unsigned flags = e.flags;
  const bool f_relu = flags & QASR_F_RELU, f_exact = flags & QASR_F_EXACT_Z;
  const unsigned flip = p.pw_unsigned ? 0x80808080u : 0u;
  const int lim = (flags & QASR_F_MASK_OUT) ? min(e.lens[b], eT) : eT;
  const size_t tap_bytes = (size_t)cout_pad * cin_pad;
  const int n_chunks = cin_pad / DENSE2_CK, n_items = n_chunks * K;

  // per-lane parameters of its output channel
  const int bias = p.bias[co];
  const float sb = f_exact ? e.sb[co] : 1.0f;
  double Mo[QASR_MAX_OUTS];
#pragma unroll
  for (int j = 0; j < QASR_MAX_OUTS; ++j) Mo[j] = (!RES && j < n_outs) ? e.outs[j].mtab[co] : 0.0;

  v16i acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = bias;

  // weight fragments of item i = (chunk, tap): the 4 K steps of the chunk, fragment order (pack.py:fragment_order), one
  // fragment-ordered [cout_pad][cin_pad] matrix per tap (QASR_F_TAPMAJOR); two register sets, loaded one item (32 MT / 8
  // MFMAs = 1 - 2 k cycles) ahead
  v4i wa[4], wb[4];
  auto load_w = [&](v4i (&w4)[4], int i) {
    const int c = i / K, tap = i - c * K;
    const v4i* wp = w_frag(p.w + tap * tap_bytes, cin_pad, co_row, 4 * c);
#pragma unroll
    for (int k = 0; k < 4; ++k) w4[k] = wp[64 * k];
  };
  const lds_u8* const a_lane = Xs + r31 * DENSE2_XP + 16 * h;   // row (frame) r31 of a 32-frame tile, K half h
  // A fragment of tile mt, K step ks of tap `tap`: rows 32 mt + tap * dil + (halo - pad) .., channels 32 ks + 16 h .. of the
  // staged chunk.  Two register sets: the fragments of the NEXT K step - at the last step of a tap, of the next tap's first
  // step - are all requested before this step's MFMAs issue (left alone, the compiler sinks every ds_read next to the MFMA
  // that uses it and waits for it: one LDS latency per MFMA; without the hand-over between taps: one per 32 MFMAs, 15-20 %
  // of the kernel parked at s_waitcnt by the SQ_WAIT_ANY counter)
  v4i a[2][MT];
  auto read_a = [&](v4i (&dst)[MT], int tap, int ks) {
    const lds_u8* arow = a_lane + (halo - pad + tap * dil) * DENSE2_XP + 32 * ks;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) dst[mt] = *(const lds_v4i*)(arow + 32 * mt * DENSE2_XP);
  };
  auto run_item = [&](const v4i (&w4)[4], int i) {
    const int c = i / K, tap = i - c * K;
    if (tap == 0) {                                          // a new chunk of 128 input channels: stage its window
      if (c > 0) __syncthreads();                            // every wave has read the previous chunk
      dense2_stage<MT>(Xs, p.x, p.cin, eTp, b, DENSE2_CK * c, t0 - halo, rows, flip);
      __syncthreads();
      read_a(a[0], 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    const bool more = tap + 1 < K;                           // (uniform)
    sep2_for<0, 4>([&](auto ksc) {
      constexpr int ks = decltype(ksc)::value;
      // one read of the next step's fragments behind every MFMA (its issue slot lies in the MFMA's 32-cycle shadow; a burst
      // of 8 reads in front of the 8 MFMAs leaves the matrix pipe idle while it issues - one wave per SIMD and work-group)
      const lds_u8* nrow = a_lane + (halo - pad + (ks + 1 < 4 ? tap : tap + 1) * dil) * DENSE2_XP + 32 * ((ks + 1) & 3);
      sep2_for<0, MT>([&](auto mtc) {
        constexpr int mt = decltype(mtc)::value;
        acc[mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ks & 1][mt], w4[ks], acc[mt], 0, 0, 0);
        if (ks + 1 < 4 || more) a[(ks + 1) & 1][mt] = *(const lds_v4i*)(nrow + 32 * mt * DENSE2_XP);
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  };
  load_w(wa, 0);
  for (int i = 0; i < n_items; i += 2) {
    if (i + 1 < n_items) load_w(wb, i + 1);
    run_item(wa, i);
    if (i + 2 < n_items) load_w(wa, i + 2);
    if (i + 1 < n_items) run_item(wb, i + 1);
  }

  // ---- epilogue in the MFMA C layout (k_sep2's: requant, pack, two half-wave swaps, one 16-byte store per lane)
  if (DBG && e.acc_dbg && co < ecout) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int t = t0 + 32 * mt + mfma32_row(r, h);
        if (t < eT) e.acc_dbg[((size_t)b * ecout + co) * eTp + t] = acc[mt][r];
      }
  }
  auto any_wide = [&](const v16i& v) {
    int mx = v[0], mn = v[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) { mx = max(mx, v[r]); mn = min(mn, v[r]); }
    return __any(mx >= (1 << 21) || mn < -(1 << 21)) != 0;
  };
  // 16 codes of a lane (MFMA C layout) -> 16 consecutive frames per lane -> one masked 16-byte store
  auto store16 = [&](void* optr, int mt, const int (&q)[16]) {
    unsigned P[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) P[g] = pack4b(q[4 * g], q[4 * g + 1], q[4 * g + 2], q[4 * g + 3]);
    const auto s02 = __builtin_amdgcn_permlane32_swap(P[0], P[2], false, false);
    const auto s13 = __builtin_amdgcn_permlane32_swap(P[1], P[3], false, false);
    v4i pk = {(int)s02[0], (int)s02[1], (int)s13[0], (int)s13[1]};
    if (t0 + 32 * (mt + 1) > lim) pk = sep2_mask16(pk, lim - (t0 + 32 * mt + 16 * h));   // masked frames (uniform branch)
    if (co < ecout) *(v4i*)((int8_t*)optr + ((size_t)b * ecout + co) * eTp + t0 + 32 * mt + 16 * h) = pk;
  };
  if constexpr (RES) {
    // res_act (jasper.py:680-682; quant_utils.py:187-214): d = rint(z_main M_main); per pane d = clamp(d + rint(z_pane M_pane)),
    // the clamp after EVERY add; then ReLU and the consumers' QuantAct.  Every |z M| < 2^30 (the packer flags the others
    // QASR_F_WIDE_RQ and they stay on k_sep): the rounded products come out of the low mantissa word, the sum is an integer
    const int qlo = e.qlo, qhi = e.qhi;
    {
      const double Mm = e.m_main[co];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const bool wide = f_exact && any_wide(acc[mt]);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = rq_rint(wide ? z_roundtrip(acc[mt][r], sb, false) : acc[mt][r], Mm);
      }
    }
    // The panes are 1x1 convs: their operand needs no tap shifts, so a chunk of 128 input channels is staged as k_sep2
    // stages its residual operand - plain 16-byte copies of the [channel][frame] tensor into [tile][channel][32 frames] - and
    // the A fragments come out of ds_read_b64_tr_b8 (the [frame][channel] image of the main conv costs 16 v_perm and 16
    // conflicted ds_write_b32 per task and would serve only 4 MT MFMAs per wave here).  A chunk is only 4 MT MFMAs per wave
    // (512 matrix-pipe cycles) behind a global round trip of several thousand under load, and the pane tensors (up to ten
    // earlier blocks' outputs, 12.6 MB each at batch 64) come from the Infinity Cache: the (pane, chunk) items are therefore
    // software-pipelined - the weights and image granules of item i + 1 are requested before item i's MFMAs and land in
    // registers meanwhile, the image alternates between two LDS buffers, one barrier per item.
    {
      constexpr int NGR = DENSE2_CK * 2 * MT, NGI = NGR / DENSE2_NT;     // 16-byte granules of a chunk image, per thread
      constexpr int IMG = MT * DENSE2_CK * 32;                            // bytes of one image
      static_assert(NGR % DENSE2_NT == 0, "pane image granules");   // (launch_dense2_v sizes the LDS for two images)
      struct Item { v4i w[4]; v4i g[NGI]; };
      // requests of item (pi, c): this wave's weight fragments of the chunk's 4 K steps + this thread's image granules
      auto issue = [&](Item& it, int pi, int c) {
        const PaneP& pn = p.panes[pi];
        const v4i* wp = w_frag(pn.w, pn.cin_pad, co_row, 4 * c);
#pragma unroll
        for (int k = 0; k < 4; ++k) it.w[k] = wp[64 * k];
#pragma unroll
        for (int i = 0; i < NGI; ++i) {
          const int gi = tid + DENSE2_NT * i, cc = gi / (2 * MT), q = gi - cc * (2 * MT);
          const int ci = DENSE2_CK * c + cc, t = t0 + 16 * q;
          // (unconditional load from a clamped address + select: a load under a branch is waited for on the spot)
          const bool ok = ci < pn.cin && t < eTp;
          const v4i v = *(const v4i*)(pn.x + ((size_t)b * pn.cin + (ok ? ci : 0)) * eTp + (ok ? t : 0));
          it.g[i] = ok ? v : (v4i){0, 0, 0, 0};
        }
      };
      int pbias = 0;
      double Mp = 0.0;
      float sbp = 1.0f;
      v16i accp[MT];
      int pi = 0, c = 0, nc = p.n_panes > 0 ? p.panes[0].cin_pad / DENSE2_CK : 0, buf = 0;
      auto step = [&](Item& cur, Item& nxt) {                 // one (pane, chunk) item; returns false behind the last one
        const PaneP& pn = p.panes[pi];
        if (c == 0) {                                          // a new pane: its per-channel parameters, fresh accumulators
          pbias = pn.bias[co];
          Mp = pn.m[co];
          sbp = f_exact ? pn.sb[co] : 1.0f;
        }
        const unsigned rflip = pn.x_unsigned ? 0x80808080u : 0u;
        lds_u8* const img = Xs + buf * IMG;
#pragma unroll
        for (int i = 0; i < NGI; ++i) {
          const int gi = tid + DENSE2_NT * i, cc = gi / (2 * MT), q = gi - cc * (2 * MT);
          v4i v = cur.g[i];
          v[0] ^= rflip; v[1] ^= rflip; v[2] ^= rflip; v[3] ^= rflip;
          *(lds_v4i*)(img + (q >> 1) * (DENSE2_CK * 32) + cc * 32 + 16 * (q & 1)) = v;
        }
        // the next item's requests, one item (a barrier + 4 MT MFMAs) ahead of their use
        int npi = pi, ncc = c + 1;
        if (ncc == nc) { npi = pi + 1; ncc = 0; }
        const bool more = npi < p.n_panes;
        if (more) issue(nxt, npi, ncc);
        if (c == 0) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) accp[mt][r] = pbias;
        }
        __syncthreads();                                       // this image is complete; every wave is done with the other one
        const lds_u8* const pi_lane = img + sep2_a_lane_off(lane);
        v4i ap[2][MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) ap[0][mt] = sep2_a_frag(pi_lane + mt * (DENSE2_CK * 32), 0);
        __builtin_amdgcn_sched_barrier(0);
        sep2_for<0, 4>([&](auto ksc) {
          constexpr int ks = decltype(ksc)::value;
          sep2_for<0, MT>([&](auto mtc) {
            constexpr int mt = decltype(mtc)::value;
            accp[mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ap[ks & 1][mt], cur.w[ks], accp[mt], 0, 0, 0);
            if constexpr (ks + 1 < 4) ap[(ks + 1) & 1][mt] = sep2_a_frag(pi_lane + mt * (DENSE2_CK * 32), ks + 1);
            __builtin_amdgcn_sched_barrier(0);
          });
        });
        if (c + 1 == nc) {                                     // the pane's last chunk: d = clamp(d + rint(z_pane M_pane))
          if (DBG && pn.acc_dbg && co < ecout) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int t = t0 + 32 * mt + mfma32_row(r, h);
                if (t < eT) pn.acc_dbg[((size_t)b * ecout + co) * eTp + t] = accp[mt][r];
              }
          }
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const bool wide = f_exact && any_wide(accp[mt]);
#pragma unroll
            for (int r = 0; r < 16; ++r)
              acc[mt][r] = med3i(acc[mt][r] + rq_rint(wide ? z_roundtrip(accp[mt][r], sbp, false) : accp[mt][r], Mp), qlo, qhi);
          }
        }
        buf ^= 1;
        pi = npi;
        c = ncc;
        if (more && ncc == 0) nc = p.panes[npi].cin_pad / DENSE2_CK;
        return more;
      };
      Item ia, ib;
      if (p.n_panes > 0) {
        __syncthreads();                                       // every wave has read the main conv's last window
        issue(ia, 0, 0);
        while (step(ia, ib) && step(ib, ia)) {}
      }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (t0 + 32 * mt >= eTp) continue;
      int z[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) z[r] = f_relu ? max(acc[mt][r], 0) : acc[mt][r];
#pragma unroll 1
      for (int j = 0; j < n_outs; ++j) {
        const OutP& o = e.outs[j];
        if (o.mode == 2) {
          store16(o.ptr, mt, z);
        } else {                                             // mode 0: scalar multiplier towards the consumer's QuantAct
          const double Ms = o.m;
          const int olo = o.lo, ohi = o.hi;
          int q[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) q[r] = requant_clamp(z[r], Ms, olo, ohi);
          store16(o.ptr, mt, q);
        }
      }
    }
    if (tline) {
      long long* r = p.prof + 4 * (size_t)wg_id;
      r[0] = tl_start; r[1] = (long long)__builtin_amdgcn_s_memrealtime(); r[2] = 0;
      r[3] = (long long)__builtin_amdgcn_s_memtime() - tl_clk;
    }
    return;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    if (t0 + 32 * mt >= eTp) continue;                       // (last time tile of a long utterance; uniform)
    int z[16];
    // z == acc below 2^22; a wave holding a larger accumulator takes fixedpoint_mul's float32 round trip
    bool wide = false;
    if (f_exact) {
      int mx = acc[mt][0], mn = acc[mt][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) { mx = max(mx, acc[mt][r]); mn = min(mn, acc[mt][r]); }
      wide = __any(mx >= (1 << 21) || mn < -(1 << 21));
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = wide ? z_roundtrip(acc[mt][r], sb, f_relu) : (f_relu ? max(acc[mt][r], 0) : acc[mt][r]);
#pragma unroll
    for (int j = 0; j < QASR_MAX_OUTS; ++j) {
      if (j < n_outs) {
        const OutP& o = e.outs[j];
        const int olo = o.lo, ohi = o.hi;
        unsigned P[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
          P[g] = pack4b(requant_clamp(z[4 * g], Mo[j], olo, ohi), requant_clamp(z[4 * g + 1], Mo[j], olo, ohi),
                        requant_clamp(z[4 * g + 2], Mo[j], olo, ohi), requant_clamp(z[4 * g + 3], Mo[j], olo, ohi));
        const auto s02 = __builtin_amdgcn_permlane32_swap(P[0], P[2], false, false);
        const auto s13 = __builtin_amdgcn_permlane32_swap(P[1], P[3], false, false);
        v4i pk = {(int)s02[0], (int)s02[1], (int)s13[0], (int)s13[1]};
        if (t0 + 32 * (mt + 1) > lim) pk = sep2_mask16(pk, lim - (t0 + 32 * mt + 16 * h));   // masked frames (uniform branch)
        if (co < ecout) *(v4i*)((int8_t*)o.ptr + ((size_t)b * ecout + co) * eTp + t0 + 32 * mt + 16 * h) = pk;
      }
    }
  }
  if (tline) {
    long long* r = p.prof + 4 * (size_t)wg_id;
    r[0] = tl_start; r[1] = (long long)__builtin_amdgcn_s_memrealtime(); r[2] = 0;
    r[3] = (long long)__builtin_amdgcn_s_memtime() - tl_clk;
  }
}

// ops k_dense2 takes: tap-major dense convs ('same' padding, stride 1) with the plain epilogue, or with residual panes and
// res_act (Jasper's block ends)
static bool dense2_common(const SepP& p) {
  const EpiP& e = p.e;
  if (p.gen != 2 || p.dense_k <= 1 || p.K != 0 || p.cin_pad % DENSE2_CK || p.cin > p.cin_pad) return false;
  if (e.flags & QASR_F_LOGITS) return false;
  if (e.n_outs < 1 || e.n_outs > QASR_MAX_OUTS || e.Tp % 64 || e.T > e.Tp) return false;
  if (((p.dense_k - 1) * p.dilation) & 1) return false;
  const int pad = p.dilation * (p.dense_k - 1) / 2, halo = (pad + 15) & ~15;
  return halo <= 32;                                         // rows <= 32 MT + 64 (dense2_stage's task count); 46 KB of LDS at most
}
static bool dense2_res(const SepP& p) {
  const EpiP& e = p.e;
  if (!(e.flags & QASR_F_RESADD) || (e.flags & QASR_F_WIDE_RQ) || p.n_panes < 1 || p.n_panes > QASR_MAX_PANES || !e.m_main) return false;
  for (int k = 0; k < p.n_panes; ++k)
    if (!p.panes[k].x || !p.panes[k].w || !p.panes[k].bias || !p.panes[k].m || p.panes[k].cin_pad % DENSE2_CK ||
        p.panes[k].cin > p.panes[k].cin_pad || ((e.flags & QASR_F_EXACT_Z) && !p.panes[k].sb))
      return false;
  for (int j = 0; j < e.n_outs; ++j)
    if ((e.outs[j].mode != 0 && e.outs[j].mode != 2) || !e.outs[j].ptr) return false;
  return true;
}
static bool dense2_plain(const SepP& p) {
  const EpiP& e = p.e;
  if ((e.flags & QASR_F_RESADD) || p.n_panes != 0) return false;
  for (int j = 0; j < e.n_outs; ++j)
    if (e.outs[j].mode != 1 || !e.outs[j].mtab || !e.outs[j].ptr) return false;
  return true;
}
bool dense2_takes(const SepP& p) { return dense2_common(p) && (dense2_plain(p) || dense2_res(p)); }

// frame tiles per wave: the plain form takes up to 8 (Tp is a multiple of 64: 2, 4, 6 or 8), the block-end form 4 (2 for Tp = 64)
static int dense2_mt(const SepP& p) {
  if (p.e.flags & QASR_F_RESADD) return p.e.Tp >= 128 ? 4 : 2;
  // 8 tiles (256 frames) per work-group: a weight fragment feeds 8 MFMAs - the cheapest frame, for several steps in flight
  // (Jasper bs64: 4.54 vs 4.78 ms/step); an engine built for ONE step in flight (32-frame tiles elsewhere) takes 4: twice the
  // work-groups per launch fill the chip better than the weight reuse pays (5.49 vs 5.93 ms/step; profiles/r04_v4_jasper_ops*.txt)
  const int cap = (p.etile > 0 && p.etile <= 32) ? 4 : 8;
  return std::min(cap, p.e.Tp >= 256 ? 8 : p.e.Tp / 32);
}

void dense2_label(const SepP& p, char* buf, size_t cap) {
  const bool dbg = p.e.acc_dbg != nullptr;
  snprintf(buf, cap, "k_dense2<%d, %s, %s>", dense2_mt(p), dbg ? "true" : "false", (p.e.flags & QASR_F_RESADD) ? "true" : "false");
}

template <int MT, bool DBG, bool RES>
static int launch_dense2_v(hipStream_t s, const SepP& p) {
  const int pad = p.dilation * (p.dense_k - 1) / 2, halo = (pad + 15) & ~15;
  size_t smem = (size_t)(32 * MT + 2 * halo) * DENSE2_XP;
  if (RES) smem = std::max(smem, (size_t)2 * MT * DENSE2_CK * 32);      // two [tile][channel][32] pane images (double-buffered)
  if (!p.x || !p.w || !p.bias || !p.e.lens || p.e.B < 1 || p.e.cout < 1 || smem > 160 * 1024) return QASR_ERR_ARG;
  static int attr_dev = -1;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (attr_dev != dev) {
    (void)hipFuncSetAttribute((const void*)k_dense2<MT, DBG, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_dev = dev;
  }
  const int tt = 32 * MT;
  dim3 g(p.e.B, ((p.e.cout + 127) & ~127) / 128, (p.e.Tp + tt - 1) / tt);
  SepP q = p;
  q.prof = g_prof_mode == 1 ? g_prof : nullptr;
  q.prof_mode = g_prof_mode;
  q.prof_cap = g_prof_cap;
  hipLaunchKernelGGL((k_dense2<MT, DBG, RES>), g, dim3(DENSE2_NT), smem, s, q);
  return QASR_OK;
}

int launch_dense2(hipStream_t s, const SepP& p) {
  const bool dbg = p.e.acc_dbg != nullptr;
  if (p.e.flags & QASR_F_RESADD) {
    if (dense2_mt(p) == 4) return dbg ? launch_dense2_v<4, true, true>(s, p) : launch_dense2_v<4, false, true>(s, p);
    return dbg ? launch_dense2_v<2, true, true>(s, p) : launch_dense2_v<2, false, true>(s, p);
  }
  switch (dense2_mt(p)) {
    case 2: return dbg ? launch_dense2_v<2, true, false>(s, p) : launch_dense2_v<2, false, false>(s, p);
    case 4: return dbg ? launch_dense2_v<4, true, false>(s, p) : launch_dense2_v<4, false, false>(s, p);
    case 6: return dbg ? launch_dense2_v<6, true, false>(s, p) : launch_dense2_v<6, false, false>(s, p);
    case 8: return dbg ? launch_dense2_v<8, true, false>(s, p) : launch_dense2_v<8, false, false>(s, p);
    default: return QASR_ERR_UNSUPPORTED;
  }
}

}  // namespace qasr
