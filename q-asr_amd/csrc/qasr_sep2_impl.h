// k_sep2: second-generation fused separable-layer kernel (jasper.py:569-600: depthwise conv -> QuantAct -> 1x1 conv
// [-> residual 1x1 conv + res_act] -> ReLU -> the consumers' QuantAct), stride-1 / dilation-1 depthwise taps.
// Included by the per-instantiation translation units qasr_sep2_t{32,64}{,_dbg}.hip.
//
// Same decomposition as k_sep (work-group = 512 threads = one utterance x TT frames x every channel; grid (B, Tp/TT)
// with the utterance as the fastest index so that all tiles of an utterance share one XCD's L2), rebuilt around what
// the round-2 micro-benchmarks (profiles/microbench/) showed:
//   * integer / f64 VALU instructions all issue at ~4.3 cycles per wave-instruction and SIMD, so the f64 form
//     lo32(fma(f64(z), M, 1.5*2^52)) + v_med3_i32 (3 instructions, exact) replaces the 9-instruction float32 fast path
//     with its ambiguity vote;
//   * the matrix pipe is the floor (v_mfma_i32_4x4x4 8.3, 32x32x32 32.2 cycles): everything else has to hide
//     behind it, which needs >= 2 work-groups per CU -> <= 128 VGPRs, <= 80 KiB LDS;
//   * the depthwise stage read one LDS dword per 4x4x4 MFMA (LDS ~100 % busy): the 4 columns of a block now take
//     frames S = TT/4 apart, so a lane reads ONE contiguous, aligned run of its window row (ds_read_b64 / b128) and
//     every dword feeds the TT/16 accumulation chains;
//   * the requantised depthwise result is written [channel][frame] as packed dwords (4 consecutive frames of a lane)
//     and the GEMM reads its A fragments with ds_read_b64_tr_b8 (the transposing 8-bit LDS read) - no byte scatter;
//     the residual operand is a plain 16-byte copy of the [channel][frame] tensor for the same reason;
//   * the epilogue stays in the MFMA C layout: requant, pack, two v_permlane32_swap to give each lane 16 consecutive
//     frames, one 16-byte store per lane and consumer - no LDS staging tile, no wave barriers;
//   * 1x1 weights stream through two register buffers of 4 K steps (the next group travels during the current
//     group's MFMAs) instead of a 64-VGPR slab held across the depthwise stage.
#pragma once
#include <algorithm>
#include <cstdio>

#include "qasr_device.h"

namespace qasr {

typedef int v2i __attribute__((ext_vector_type(2)));

#define SEP2_NT 512
#ifndef SEP2_WPE
#define SEP2_WPE 4                      /* waves per SIMD the register budget is sized for (2 work-groups per CU) */
#endif
#define SEP2_CH 256                     /* channels per staged window chunk */

template <int K, int TT>
struct Sep2Geo {
  static constexpr int PAD = K / 2;
  static constexpr int HALO = (PAD + 15) / 16 * 16;      // staged halo (16-B granular)
  static constexpr int D = HALO - PAD;                   // byte offset of tap 0 of output frame 0 inside a window row
  static constexpr int NU = TT / 16;                     // accumulation chains per lane
  static constexpr int S = 4 * NU;                       // frame distance between the 4 columns of a 4x4x4 block
  static constexpr int MS = -(D & 3);                    // first tap offset: keeps the window dwords 4-byte aligned
  static constexpr int NS = (K + 3 - MS + 3) / 4;        // MFMA steps per chain
  static constexpr int A0 = D + MS;                      // byte offset of window dword 0 (multiple of 4)
  static constexpr int OFF = (A0 & (S - 1)) / 4;         // dwords skipped at the head of the S-aligned lane stream
  static constexpr int NE = OFF + NU + NS - 1;           // dwords of the lane stream
  static constexpr int NRD = (4 * NE + S - 1) / S;       // S-byte LDS reads per lane and group
  static constexpr int WLEN = TT + 2 * HALO;             // staged bytes per window row
  static constexpr int WP = WLEN;                        // LDS row pitch
  static constexpr int NPG = WLEN / 16;                  // 16-B granules per row
  static constexpr int NPT = (SEP2_CH * NPG + SEP2_NT - 1) / SEP2_NT;   // granules per thread and chunk
  static constexpr int KP4 = (K + 3) / 4;
  static constexpr int KS = 4 * KP4 + 32;                // row pitch of the zero-margined tap array (pack.py)
  static constexpr int NR = (NS + 1 + 3) / 4 * 4;        // tap dwords fetched per lane (whole 16-B loads)
  static_assert(3 * S + (A0 & ~(S - 1)) + NRD * S <= WP, "lane stream leaves the window row");
  static_assert(8 + MS - 3 >= 0 && 4 * (((8 + MS) >> 2) + NR) <= KS, "tap stream leaves the tap row");
};

// rint(z * M) for |z * M| < 2^31 (the packer routes ops that cannot promise this to k_sep): one fp64 fma rounds
// half-to-even into the low mantissa word (quant_utils.py:196-198: round(f64(z) * f64(m) / 2^e), M = m * 2^-e)
__device__ __forceinline__ int rq_rint(int z, double M) { return __double2loint(__builtin_fma((double)z, M, MAGIC_RNE)); }
// clamp(x, lo, hi) for lo <= hi as ONE v_med3_i32 (the compiler keeps min/max apart: it cannot know lo <= hi)
__device__ __forceinline__ int med3i(int x, int lo, int hi) {
  int r;
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ int rq_clamp(int z, double M, int lo, int hi) { return med3i(rq_rint(z, M), lo, hi); }
__device__ __forceinline__ unsigned pack4b(int a, int b, int c, int d) {
  const unsigned lo = __builtin_amdgcn_perm((unsigned)b, (unsigned)a, 0x0c0c0400u);    // [a0, b0, 0, 0]
  const unsigned hi = __builtin_amdgcn_perm((unsigned)d, (unsigned)c, 0x04000c0cu);    // [0, 0, c0, d0]
  return lo | hi;
}
// ---- 1x1 GEMM over [channel][32-frame] LDS images --------------------------------------------------------------
// A fragment of K step ks for lane (r = lane & 31, h = lane >> 5): channels 32 ks + 16 h + {0..15} of frame r.  Two
// transposing reads: the 16 lanes of a group supply the 8 rows x 16 bytes of a block (lane 2q+p: row q, bytes 8p..8p+7)
// and lane i of the group receives column i of the 8 rows.
__device__ __forceinline__ int sep2_a_lane_off(int lane) {
  return (16 * (lane >> 5) + ((lane & 15) >> 1)) * 32 + 16 * ((lane >> 4) & 1) + 8 * (lane & 1);
}
__device__ __forceinline__ v4i sep2_a_frag(const unsigned char* img_lane, int ks) {
  typedef v2i __attribute__((address_space(3))) * lds_v2i;
  const v2i lo = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i)(img_lane + ks * 1024));
  const v2i hi = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i)(img_lane + ks * 1024 + 256));
  return (v4i){lo[0], lo[1], hi[0], hi[1]};
}
__device__ __forceinline__ void sep2_load_wg(v4i (&wb)[4], const v4i* __restrict__ wp) {
#pragma unroll
  for (int i = 0; i < 4; ++i) wb[i] = wp[64 * i];          // consecutive K steps are 1 KiB apart (fragment order)
}
template <int MT>
__device__ __forceinline__ void sep2_mma_group(v16i (&acc)[MT], const v4i (&wb)[4], const unsigned char* img_lane, int mt_stride,
                                               int g) {
  // two K steps at a time: 8 MT registers of A fragments in flight instead of 16 MT
#pragma unroll
  for (int i2 = 0; i2 < 4; i2 += 2) {
    v4i a[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < 2; ++i) a[mt][i] = sep2_a_frag(img_lane + mt * mt_stride, 4 * g + i2 + i);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[mt][i], wb[i2 + i], acc[mt], 0, 0, 0);
  }
}
// n (even) groups of 4 K steps; wb0 holds group 0 on entry and the first group of `next` (the GEMM that follows; any
// valid fragment pointer when nothing follows) on exit.  The next group always travels while the current one is on the
// matrix cores.  One rolled pair loop, no guards: unrolled with per-group conditions the accumulators and both buffers
// were renamed at every join and fresh weight loads got spilled behind s_waitcnt vmcnt(0).
template <int MT>
__device__ __forceinline__ void sep2_gemm(v16i (&acc)[MT], v4i (&wb0)[4], v4i (&wb1)[4], const unsigned char* img_lane, int mt_stride,
                                          const v4i* __restrict__ wp, int n, const v4i* __restrict__ next) {
#pragma unroll 1
  for (int i = 0; i < n; i += 2) {
    sep2_load_wg(wb1, wp + 256 * (i + 1));
    sep2_mma_group<MT>(acc, wb0, img_lane + 4096 * i, mt_stride, 0);
    sep2_load_wg(wb0, i + 2 < n ? wp + 256 * (i + 2) : next);
    sep2_mma_group<MT>(acc, wb1, img_lane + 4096 * (i + 1), mt_stride, 0);
  }
}

enum { EP2_PLAIN = 1, EP2_RESADD1 = 2 };

template <int K, int EP, bool DBG, int TT>
__global__ void __launch_bounds__(SEP2_NT, SEP2_WPE) k_sep2(SepP p) {
  using G = Sep2Geo<K, TT>;
  constexpr int MT = TT / 32;
  constexpr int NU = G::NU, S = G::S, NS = G::NS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const EpiP& e = p.e;
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x, t0 = blockIdx.y * TT;
  const int cin = p.cin, cin_pad = p.cin_pad;
  unsigned char* const Xd = smem;                            // [MT][cin_pad][32]  A image of the 1x1 conv
  unsigned char* const Ws = smem + TT * cin_pad;             // [<= 256][WP] window chunk; later the residual A image

  const unsigned flags = e.flags;
  const int eT = e.T, eTp = e.Tp, ecout = e.cout, n_outs = e.n_outs;
  const int len_b = e.lens[b];
  const int lim = (flags & QASR_F_MASK_OUT) ? min(len_b, eT) : eT;
  const int dlim = min(len_b, eT);                           // the 1x1 conv's MaskedConv1d masks its input
  const bool f_relu = flags & QASR_F_RELU;
  const bool f_exact = flags & QASR_F_EXACT_Z;
  const int cout_pad = (ecout + 127) / 128 * 128;
  const int ng = cin_pad >> 7;                               // groups of 4 K steps
  const int dw_lo = p.dw_lo, dw_hi = p.dw_hi;
  const bool stamp = p.prof && blockIdx.x == 0 && blockIdx.y == 1 && tid == 0;
  int nst = 0;
#define STAMP2() do { if (stamp && nst < 31) p.prof[nst++] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
  STAMP2();

  // ------------------------------------------------------------------------------------------ window fetch / commit
  const unsigned flip = p.x_unsigned ? 0x80808080u : 0u;
  v4i pc[G::NPT];
  auto fetch = [&](int c0) {                                 // global -> registers (coalesced 16-B granules)
    const int nch = min(SEP2_CH, cin - c0);
#pragma unroll
    for (int i = 0; i < G::NPT; ++i) {
      const int pi = tid + SEP2_NT * i;
      const int row = pi / G::NPG, col = pi - row * G::NPG;
      const int t = t0 - G::HALO + 16 * col;                 // a granule lies entirely inside or outside [0, Tp)
      pc[i] = (v4i){0, 0, 0, 0};
      if (row < nch && t >= 0 && t < eTp) pc[i] = *(const v4i*)(p.x + ((size_t)b * cin + c0 + row) * eTp + t);
    }
  };
  auto commit = [&](int c0) {                                // registers -> LDS window
    const int nch = min(SEP2_CH, cin - c0);
#pragma unroll
    for (int i = 0; i < G::NPT; ++i) {
      const int pi = tid + SEP2_NT * i;
      const int row = pi / G::NPG, col = pi - row * G::NPG;
      if (row < nch) {
        v4i v = pc[i];
        v[0] ^= flip; v[1] ^= flip; v[2] ^= flip; v[3] ^= flip;
        *(v4i*)(Ws + row * G::WP + 16 * col) = v;
      }
    }
  };
  fetch(0);

  // first weight group of GEMM pass 0 (consumed after the depthwise stage, which hides its latency)
  const int co_l = 32 * wave + (lane & 31);                  // row inside a 256-channel pass
  v4i wb0[4], wb1[4];
  sep2_load_wg(wb0, w_frag(p.w, cin_pad, co_l < cout_pad ? co_l : 0, 0));
  __builtin_amdgcn_sched_barrier(0);

  // ------------------------------------------------------------------------------------------ depthwise stage
  // v_mfma_i32_4x4x4_16B_i8: 16 independent 4x4x4 products, block = channel.  A[i][k] = w[m0 + k - i] (lane i's own
  // pre-shifted tap stream), B[k][j] = win[S j + D + m0 + k] (lane j's window run), so the block accumulates
  // out[S j + 4 u + i] over taps m0 - i .. m0 - i + 3 for chain u when B is taken 4 u bytes further on; m0 advances
  // by 4 per step.  Lane l: channel l >> 2, row / column l & 3; register v of chain u = frame S (l & 3) + 4 u + v.
  const bool full_in = t0 + TT <= dlim;                      // no masked frame in this tile (uniform)
  v4i rr[(TT * 512 / 16 + SEP2_NT - 1) / SEP2_NT];           // residual operand granules in flight (EP2_RESADD1)
  constexpr int NRT = (TT * 512 / 16 + SEP2_NT - 1) / SEP2_NT;
  auto fetch_res = [&]() {
    const PaneP& pn = p.panes[0];
#pragma unroll
    for (int i = 0; i < NRT; ++i) {
      const int gi = tid + SEP2_NT * i;                      // granule: channel gi / (TT/16), 16 frames
      const int c = gi / (TT / 16), q = gi - c * (TT / 16);
      rr[i] = (v4i){0, 0, 0, 0};
      if (c < pn.cin) rr[i] = *(const v4i*)(pn.x + ((size_t)b * pn.cin + c) * eTp + t0 + 16 * q);
    }
  };
  for (int c0 = 0; c0 < cin; c0 += SEP2_CH) {
    const int nch = min(SEP2_CH, cin - c0);
    const int cb = lane >> 2, jl = lane & 3;
    constexpr int e0base = 8 + G::MS;                        // the lane's tap stream starts at byte e0base - jl of its row
    const int e0 = e0base - jl, tq = e0 >> 2, tsh = e0 & 3;
    if (c0) __syncthreads();                                 // previous chunk's window fully consumed
    commit(c0);
    __syncthreads();
    const bool last = c0 + SEP2_CH >= cin;
    if (!last) fetch(c0 + SEP2_CH);                          // next chunk's window travels during this chunk's math
    if (EP == EP2_RESADD1 && last) fetch_res();
    STAMP2();
#pragma unroll 1
    for (int g = 0; g < 2; ++g) {
      if (32 * wave + 16 * g >= nch) break;                  // wave-uniform: group entirely beyond the chunk
      const int row = 32 * wave + 16 * g + cb;
      const bool row_ok = row < nch;
      const int rowc = min(row, nch - 1);
      const int c = c0 + rowc;
      // taps: whole dwords by wide loads (rows are 4-byte aligned), funnel-shifted into place
      unsigned tw[NS];
      {
        v4i raw[G::NR / 4];
        const v4i* tp = (const v4i*)((const unsigned char*)p.wdw2 + (size_t)c * G::KS + 4 * tq);
#pragma unroll
        for (int i = 0; i < G::NR / 4; ++i) raw[i] = tp[i];
#pragma unroll
        for (int st = 0; st < NS; ++st)
          tw[st] = __builtin_amdgcn_alignbyte((unsigned)raw[(st + 1) >> 2][(st + 1) & 3], (unsigned)raw[st >> 2][st & 3], tsh);
      }
      const int biasg = p.bias_dw[c];
      const double Mg = p.m_dw[c];
      // the lane's window run: S-byte aligned reads, every dword feeds the NU chains
      unsigned xs[G::NRD * (S / 4)];
      {
        const unsigned char* wr = Ws + rowc * G::WP + S * jl + (G::A0 & ~(S - 1));
#pragma unroll
        for (int i = 0; i < G::NRD; ++i) {
          if constexpr (S == 16) {
            const v4i v = *(const v4i*)(wr + 16 * i);
            xs[4 * i] = v[0]; xs[4 * i + 1] = v[1]; xs[4 * i + 2] = v[2]; xs[4 * i + 3] = v[3];
          } else {
            const v2i v = *(const v2i*)(wr + 8 * i);
            xs[2 * i] = v[0]; xs[2 * i + 1] = v[1];
          }
        }
      }
      v4i acc[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) acc[u] = (v4i){biasg, biasg, biasg, biasg};
#pragma unroll
      for (int st = 0; st < NS; ++st)
#pragma unroll
        for (int u = 0; u < NU; ++u)
          acc[u] = __builtin_amdgcn_mfma_i32_4x4x4i8((int)tw[st], (int)xs[G::OFF + u + st], acc[u], 0, 0, 0);
      if (DBG && p.dw_acc_dbg && row_ok) {
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int f = S * jl + 4 * u + v;
            if (t0 + f < eT) p.dw_acc_dbg[((size_t)b * cin + c) * eTp + t0 + f] = acc[u][v];
          }
      }
      if (!full_in) {                                        // masked frames (t >= len): accumulator 0 requantises to 0
        int dl = dlim - t0 - S * jl;                         // (dw_lo <= 0 <= dw_hi); kept out of the loop-invariant code
        asm volatile("" : "+v"(dl));
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
          for (int v = 0; v < 4; ++v)
            if (4 * u + v >= dl) acc[u][v] = 0;
      }
      if (row_ok) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const int f = S * jl + 4 * u;                      // first of this dword's 4 frames
          const unsigned w = pack4b(rq_clamp(acc[u][0], Mg, dw_lo, dw_hi), rq_clamp(acc[u][1], Mg, dw_lo, dw_hi),
                                    rq_clamp(acc[u][2], Mg, dw_lo, dw_hi), rq_clamp(acc[u][3], Mg, dw_lo, dw_hi));
          *(unsigned*)(Xd + (f >> 5) * (cin_pad * 32) + c * 32 + (f & 31)) = w;
        }
      }
    }
    STAMP2();
  }
  __syncthreads();                                           // Xd complete, window dead
  if (EP == EP2_RESADD1) {                                   // residual A image [cin_r_pad][32] per 32-frame tile
    const PaneP& pn = p.panes[0];
    const unsigned rflip = pn.x_unsigned ? 0x80808080u : 0u;
#pragma unroll
    for (int i = 0; i < NRT; ++i) {
      const int gi = tid + SEP2_NT * i;
      const int c = gi / (TT / 16), q = gi - c * (TT / 16);
      if (c < pn.cin) {
        v4i v = rr[i];
        v[0] ^= rflip; v[1] ^= rflip; v[2] ^= rflip; v[3] ^= rflip;
        *(v4i*)(Ws + (q >> 1) * (pn.cin_pad * 32) + c * 32 + 16 * (q & 1)) = v;
      }
    }
    __syncthreads();
  }
  STAMP2();

  // ------------------------------------------------------------------------------------------ 1x1 GEMM passes of 256 channels
  const unsigned char* const xd_lane = Xd + sep2_a_lane_off(lane);
  const unsigned char* const xr_lane = Ws + sep2_a_lane_off(lane);
  const bool full_out = t0 + TT <= lim;
  // per-pane scalars once (the kernarg block is large; re-reading it inside the pass loop costs SGPRs and waits)
  const int8_t* const pw = EP == EP2_RESADD1 ? p.panes[0].w : p.w;
  const int pcin_pad = EP == EP2_RESADD1 ? p.panes[0].cin_pad : cin_pad;
  const int32_t* const pbias = EP == EP2_RESADD1 ? p.panes[0].bias : p.bias;
  const double* const pm = EP == EP2_RESADD1 ? p.panes[0].m : nullptr;
  const float* const psb = EP == EP2_RESADD1 ? p.panes[0].sb : nullptr;
  int32_t* const pdbg = EP == EP2_RESADD1 ? p.panes[0].acc_dbg : nullptr;
  const int qlo = f_relu ? max(e.qlo, 0) : e.qlo, qhi = e.qhi;
#pragma unroll 1
  for (int cbase = 0; cbase < cout_pad; cbase += 256) {
    const int co = cbase + co_l;
    const int cor = co < cout_pad ? co : 0;                  // waves beyond cout_pad (multiple of 128) redo tile 0, stores are off
    const bool co_ok = co < ecout;
    const int con = cbase + 256 + co_l < cout_pad ? cbase + 256 + co_l : 0;   // next pass (or a harmless re-read of tile 0)
    const v4i* const wmain = w_frag(p.w, cin_pad, cor, 0);
    const v4i* const wnext = w_frag(p.w, cin_pad, con, 0);
    const v4i* const wpane = w_frag(pw, pcin_pad, cor, 0);
    int rl = lim - t0 - 4 * h;                               // frames of this lane's registers below rl are valid;
    asm volatile("" : "+v"(rl));                             // opaque: keeps 16 MT masks out of the loop-invariant code
    v16i acc[MT];
    {
      const int bias = p.bias[cor];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = bias;
    }
    sep2_gemm<MT>(acc, wb0, wb1, xd_lane, cin_pad * 32, wmain, ng, EP == EP2_RESADD1 ? wpane : wnext);
    STAMP2();
    // accumulator hooks, then masked frames (t >= lim): an accumulator of 0 requantises to 0 for every consumer
    // (lo <= 0 <= hi) and through res_act
    auto finish = [&](v16i (&a)[MT], int32_t* dbg, float sb) {
      if (DBG && dbg && co_ok) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int t = t0 + 32 * mt + mfma32_row(r, h);
            if (t < eT) dbg[((size_t)b * ecout + co) * eTp + t] = a[mt][r];
          }
      }
      if (!full_out) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (32 * mt + (r & 3) + 8 * (r >> 2) >= rl) a[mt][r] = 0;
      }
      if (f_exact) {
        // z == acc is a theorem for |acc| < 2^22; a wave holding an accumulator beyond +-2^21 takes the
        // float32 round trip of fixedpoint_mul (quant_utils.py:187), which is the identity below the bound
        unsigned t = 0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) t |= (unsigned)(a[mt][r] + (1 << 21));
        if (__any((t >> 22) != 0)) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) a[mt][r] = z_roundtrip(a[mt][r], sb, EP == EP2_PLAIN && f_relu);
        }
      }
    };
    // one consumer's 16 requantised values of a 32-frame tile -> 16 consecutive frames per lane -> one 16-B store.
    // MFMA C layout: register 4 g + i of lane (c, h) = frame 8 g + 4 h + i; after the two half-wave swaps the lower
    // half-wave holds frames 0..15 and the upper one frames 16..31 of its channel in register order {0, 2, 1, 3}.
    auto store16 = [&](void* optr, int mt, const int (&q)[16]) {
      unsigned P[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) P[g] = pack4b(q[4 * g], q[4 * g + 1], q[4 * g + 2], q[4 * g + 3]);
      const auto s02 = __builtin_amdgcn_permlane32_swap(P[0], P[2], false, false);
      const auto s13 = __builtin_amdgcn_permlane32_swap(P[1], P[3], false, false);
      const v4i pk = {(int)s02[0], (int)s02[1], (int)s13[0], (int)s13[1]};
      if (co_ok) *(v4i*)((int8_t*)optr + ((size_t)b * ecout + co) * eTp + t0 + 32 * mt + 16 * h) = pk;
    };

    if (EP == EP2_RESADD1) {
      // res_act (jasper.py:680-682; quant_utils.py:187-214): q = clamp(rq(out) + rq(res)), then ReLU (folded into qlo)
      finish(acc, e.acc_dbg, f_exact ? e.sb[cor] : 1.0f);
      {
        const double Mm = e.m_main[cor];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][r] = rq_rint(acc[mt][r], Mm);
      }
      v16i accp[MT];
      {
        const int bv = pbias[cor];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) accp[mt][r] = bv;
      }
      sep2_gemm<MT>(accp, wb0, wb1, xr_lane, pcin_pad * 32, wpane, pcin_pad >> 7, wnext);
      STAMP2();
      finish(accp, pdbg, f_exact ? psb[cor] : 1.0f);
      const double Mp = pm[cor];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        int z[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = med3i(acc[mt][r] + rq_rint(accp[mt][r], Mp), qlo, qhi);
#pragma unroll 1
        for (int j = 0; j < n_outs; ++j) {
          const OutP& o = e.outs[j];
          if (o.mode == 2) {
            store16(o.ptr, mt, z);
          } else {                                           // mode 0: scalar multiplier towards the consumer's QuantAct
            const double Mo = o.m;
            const int olo = o.lo, ohi = o.hi;
            int q[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) q[r] = rq_clamp(z[r], Mo, olo, ohi);
            store16(o.ptr, mt, q);
          }
        }
      }
    } else {
      // ReLU is folded into the consumers' lower clamp bound by the packer (M > 0: rint(z M) <= 0 for z <= 0)
      finish(acc, e.acc_dbg, f_exact ? e.sb[cor] : 1.0f);
#pragma unroll 1
      for (int j = 0; j < n_outs; ++j) {
        const OutP& o = e.outs[j];
        const double Mo = o.mtab[cor];
        const int olo = o.lo, ohi = o.hi;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          int q[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) q[r] = rq_clamp(acc[mt][r], Mo, olo, ohi);
          store16(o.ptr, mt, q);
        }
      }
    }
    STAMP2();
  }
  if (stamp) p.prof[31] = nst;
#undef STAMP2
}

template <int K, int TT>
static inline size_t sep2_smem_bytes(const SepP& p) {
  using G = Sep2Geo<K, TT>;
  const size_t xd = (size_t)TT * p.cin_pad;
  size_t ws = (size_t)std::min(SEP2_CH, p.cin) * G::WP + 64;
  if (p.n_panes == 1) ws = std::max(ws, (size_t)TT * p.panes[0].cin_pad);
  return xd + ws;
}

// Shapes k_sep2 is built for; everything else stays on k_sep.
static inline bool sep2_shape_ok(const SepP& p) {
  const EpiP& e = p.e;
  if (p.K <= 0 || p.dilation != 1 || p.dense_k > 1) return false;
  if (!(p.K == 33 || p.K == 39 || p.K == 51 || p.K == 63 || p.K == 75 || p.K == 11 || p.K == 13)) return false;
  if ((p.cin_pad != 256 && p.cin_pad != 512) || e.cout > 512) return false;   // even counts of 128-deep K groups
  if (e.flags & (QASR_F_LOGITS | QASR_F_WIDE_RQ)) return false;
  if (e.n_outs < 1) return false;
  if (e.flags & QASR_F_RESADD) {
    if (p.n_panes != 1 || (p.panes[0].cin_pad != 256 && p.panes[0].cin_pad != 512)) return false;
    for (int j = 0; j < e.n_outs; ++j)
      if (e.outs[j].mode != 0 && e.outs[j].mode != 2) return false;
    return true;
  }
  if (p.n_panes != 0) return false;
  for (int j = 0; j < e.n_outs; ++j)
    if (e.outs[j].mode != 1) return false;
  return true;
}

template <int K, int EP, bool DBG, int TT>
static int launch_sep2_v(hipStream_t s, const SepP& p) {
  const size_t smem = sep2_smem_bytes<K, TT>(p);
  if (smem > 160 * 1024 || p.e.B < 1 || p.e.Tp % TT || !p.x || !p.w || !p.wdw2) return QASR_ERR_ARG;
  static int attr_dev = -1;                                  // the attribute is per device
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (attr_dev != dev) {
    (void)hipFuncSetAttribute((const void*)k_sep2<K, EP, DBG, TT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_dev = dev;
  }
  SepP q = p;
  q.prof = g_prof;
  hipLaunchKernelGGL((k_sep2<K, EP, DBG, TT>), dim3(p.e.B, p.e.Tp / TT, 1), dim3(SEP2_NT), smem, s, q);
  return QASR_OK;
}

template <int K, bool DBG, int TT>
static int launch_sep2_k(hipStream_t s, const SepP& p) {
  if (p.e.flags & QASR_F_RESADD) return launch_sep2_v<K, EP2_RESADD1, DBG, TT>(s, p);
  return launch_sep2_v<K, EP2_PLAIN, DBG, TT>(s, p);
}

// all kernel-size instantiations of one (tile, debug) pair; QASR_ERR_UNSUPPORTED for a tap count without one
template <int TT, bool DBG>
int launch_sep2_inst(hipStream_t s, const SepP& p) {
  switch (p.K) {
    case 11: return launch_sep2_k<11, DBG, TT>(s, p);
    case 13: return launch_sep2_k<13, DBG, TT>(s, p);
    case 33: return launch_sep2_k<33, DBG, TT>(s, p);
    case 39: return launch_sep2_k<39, DBG, TT>(s, p);
    case 51: return launch_sep2_k<51, DBG, TT>(s, p);
    case 63: return launch_sep2_k<63, DBG, TT>(s, p);
    case 75: return launch_sep2_k<75, DBG, TT>(s, p);
    default: return QASR_ERR_UNSUPPORTED;
  }
}

}  // namespace qasr
