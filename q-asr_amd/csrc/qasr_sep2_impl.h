// k_sep2: second-generation fused separable-layer kernel (jasper.py:569-600: depthwise conv -> QuantAct -> 1x1 conv
// [-> residual 1x1 conv + res_act] -> ReLU -> the consumers' QuantAct), stride-1 / dilation-1 depthwise taps.
// Included by the per-instantiation translation units qasr_sep2_t{32,64,128}{,_dbg}.hip.
//
// Same decomposition as k_sep (work-group = 512 threads = one utterance x TT frames x every channel; grid (B, Tp/TT)
// with the utterance as the fastest index so that all tiles of an utterance share one XCD's L2), rebuilt around what
// the round-2 micro-benchmarks (profiles/microbench/) and phase stamps showed:
//   * integer / f64 VALU instructions all issue at ~4.3 cycles per wave-instruction and SIMD, so the f64 form
//     lo32(fma(f64(z), M, 1.5*2^52)) + v_med3_i32 (3 instructions, exact) replaces the 9-instruction float32 fast path
//     with its ambiguity vote;
//   * a CU takes ~50 B/clk from L2 and one work-group asks for ~0.5 MB (256 KiB of 1x1 weights, the window, the taps):
//     the texture path is the busiest unit, so nothing is fetched twice - tap rows go through LDS once
//     (k_sep: every lane fetched its own pre-shifted copy, 4x the bytes) - and requests are issued in the
//     order the math needs them (window, taps, then the weight slab);
//   * the two waves of a SIMD share ONE instruction-issue port and the older wave wins it (profiles/microbench/
//     coissue.hip): their phases do not overlap, a SIMD's time is the sum of its waves' instructions, and VALU work
//     hides only behind the SAME wave's MFMAs.  Hence (a) the depthwise stage has no work-group barriers - every wave
//     stages the window / tap rows of its own 16 channels per group in a private LDS region -, (b) the requantisation
//     of depthwise group g-1 is issued one instruction behind every 4x4x4 MFMA of group g, (c) per-group constants
//     are computed once and length masks are applied to packed codes (fewer instructions);
//   * every global operand is requested >= 1 k cycles before its use: the whole weight slab of the next GEMM sits in
//     registers (re-requested group by group as soon as a group has been multiplied), per-channel parameters of the
//     next pass travel during the current one;
//   * the depthwise stage read one LDS dword per 4x4x4 MFMA (LDS ~100 % busy): the 4 columns of a block now take
//     frames S = TT/4 apart, so a lane reads ONE contiguous, aligned run of its window row (ds_read_b64 / b128) and
//     every dword feeds the TT/16 accumulation chains;
//   * the requantised depthwise result is written [channel][frame] as packed dwords (4 consecutive frames of a lane)
//     and the GEMM reads its A fragments with ds_read_b64_tr_b8 (the transposing 8-bit LDS read) - no byte scatter;
//     the residual operand is a plain 16-byte copy of the [channel][frame] tensor for the same reason;
//   * the epilogue stays in the MFMA C layout: requant, pack, two v_permlane32_swap to give each lane 16 consecutive
//     frames, one 16-byte store per lane and consumer - no LDS staging tile, no wave barriers;
//   * the K depth of both GEMMs is a template parameter (groups of 128 input channels): with run-time group guards the
//     compiler renamed accumulators and weight buffers at every join, spilled fresh weight loads behind
//     s_waitcnt vmcnt(0) and serialised the LDS reads.
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "qasr_device.h"

namespace qasr {

typedef int v2i __attribute__((ext_vector_type(2)));
// LDS pointers stay in address space 3 (32-bit): derived through generic pointers the offsets were computed with 64-bit
// multiply-adds whose unused upper halves dragged unrelated pending loads into every address (s_waitcnt vmcnt(0))
typedef unsigned char __attribute__((address_space(3))) lds_u8;
typedef unsigned __attribute__((address_space(3))) lds_u32;
typedef v2i __attribute__((address_space(3))) lds_v2i;
typedef v4i __attribute__((address_space(3))) lds_v4i;

#define SEP2_NT 512
#ifndef SEP2_WPE
#define SEP2_WPE 2                      /* waves per SIMD the register budget is sized for: 2 = one work-group per CU, 256 VGPRs */
#endif
#define SEP2_CH 128                     /* channels per depthwise group: 16 per wave */

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
  static constexpr int RG = S < 16 ? S : 16;             // LDS read granule of the lane stream (ds_read_b64 / b128)
  static constexpr int OFF = (A0 & (RG - 1)) / 4;        // dwords skipped at the head of the RG-aligned lane stream
  static constexpr int NE = OFF + NU + NS - 1;           // dwords of the lane stream
  static constexpr int NRD = (4 * NE + RG - 1) / RG;     // RG-byte LDS reads per lane and group
  static constexpr int WLEN = TT + 2 * HALO;             // staged bytes per window row
  // LDS row pitch: the lanes of one LDS access group read S-byte runs of 4 (b128) / 8 (b64) different rows; an odd
  // multiple of 4 S bytes puts those rows on disjoint banks (a power-of-two pitch made every read 4-way conflicted)
  // (128-frame tiles keep the dense pitch: the conflict-free one would not fit the LDS next to the 64 KiB A image)
#ifndef SEP2_CF128
#define SEP2_CF128 1                    /* conflict-free window pitch at 128 frames too: the wave-private rows fit */
#endif
  static constexpr int WP = (TT > 64 && !SEP2_CF128) ? WLEN : ((WLEN + 4 * S - 1) / (4 * S) | 1) * (4 * S);
  static constexpr int NPG = WLEN / 16;                  // 16-B granules per row
  static constexpr int NPT = (16 * NPG + 63) / 64;       // window granules per lane and group (a wave stages its own 16 rows)
  static constexpr int KP4 = (K + 3) / 4;
  static constexpr int KS = 4 * KP4 + 32;                // row pitch of the zero-margined tap array (pack.py)
  static constexpr int TAPB = 16 * KS;                   // tap bytes of a wave's 16 rows (KS granules of 16 B)
  static constexpr int NTT = (KS + 63) / 64;             // tap granules per lane and group
  static constexpr int WREG = 16 * WP + TAPB + 64;       // LDS bytes of one wave's private window + tap rows
  static_assert(3 * S + (A0 & ~(RG - 1)) + NRD * RG <= WLEN && WLEN <= WP, "lane stream leaves the window row");
  static_assert(8 + MS - 3 >= 0 && 8 + MS + 4 * (NS + 1) <= KS, "tap stream leaves the tap row");
  static_assert(WREG % 16 == 0, "wave regions are not 16-byte granular");
};

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
__device__ __forceinline__ v4i sep2_a_frag(const lds_u8* img_lane, int ks) {
  const v2i lo = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i*)(img_lane + ks * 1024));
  const v2i hi = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i*)(img_lane + ks * 1024 + 256));
  return (v4i){lo[0], lo[1], hi[0], hi[1]};
}
__device__ __forceinline__ void sep2_load_wg(v4i* wb, const v4i* __restrict__ wp) {
#pragma unroll
  for (int i = 0; i < 4; ++i) wb[i] = wp[64 * i];          // consecutive K steps are 1 KiB apart (fragment order)
}
// One 1x1 GEMM of N groups of 4 K steps on the register-resident weight slab wf (group g in wf[4g .. 4g+3]).  As soon
// as group g has been multiplied its registers are re-requested: from r0 + 256 g for g < N0, else from r1 + 256 g
// (nullptr: nothing) - the same group of the GEMM that runs next on these registers.  A fragments are double-buffered
// two K steps at a time so that the LDS latency of the next pair hides behind the current pair's MFMAs.
template <int MT, int N, int N0>
__device__ __forceinline__ void sep2_gemm(v16i (&acc)[MT], v4i (&wf)[16], const lds_u8* img_lane, int mt_stride,
                                          const v4i* __restrict__ r0, const v4i* __restrict__ r1) {
  constexpr int AB = MT > 2 ? 1 : 2;                         // K steps per A-fragment batch (register budget at MT = 4)
  constexpr int NQ = 4 * N / AB;
  v4i a[2][MT][AB];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int i = 0; i < AB; ++i) a[0][mt][i] = sep2_a_frag(img_lane + mt * mt_stride, i);
#pragma unroll
  for (int q = 0; q < NQ; ++q) {                             // batch of AB K steps; group g = q * AB / 4
    if (q + 1 < NQ) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < AB; ++i) a[(q + 1) & 1][mt][i] = sep2_a_frag(img_lane + mt * mt_stride, AB * (q + 1) + i);
    }
#pragma unroll
    for (int i = 0; i < AB; ++i)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[q & 1][mt][i], wf[AB * q + i], acc[mt], 0, 0, 0);
    if ((AB * (q + 1)) % 4 == 0) {
      const int g = (AB * (q + 1)) / 4 - 1;
      if (g < N0) {
        if (r0) sep2_load_wg(&wf[4 * g], r0 + 256 * g);
      } else {
        if (r1) sep2_load_wg(&wf[4 * g], r1 + 256 * g);
      }
    }
  }
}

// Dwords [D, NE) of a lane's RG-aligned window stream, in the widest aligned LDS reads that cover EXACTLY those dwords:
// a register of a wider read that nothing uses is handed out again at once, and the write to it waits for the read
// (an s_waitcnt lgkmcnt(0) in front of the requantisation the read should hide behind).
template <int D, int NE, int RG>
__device__ __forceinline__ void sep2_rd_stream(unsigned* xs, const lds_u8* wr) {
  if constexpr (D < NE) {
    constexpr int n = (RG == 16 && D % 4 == 0 && NE - D >= 4) ? 4 : (D % 2 == 0 && NE - D >= 2) ? 2 : 1;
    if constexpr (n == 4) {
      const v4i v = *(const lds_v4i*)(wr + 4 * D);
      xs[D] = v[0]; xs[D + 1] = v[1]; xs[D + 2] = v[2]; xs[D + 3] = v[3];
    } else if constexpr (n == 2) {
      const v2i v = *(const lds_v2i*)(wr + 4 * D);
      xs[D] = v[0]; xs[D + 1] = v[1];
    } else {
      xs[D] = *(const lds_u32*)(wr + 4 * D);
    }
    sep2_rd_stream<D + n, NE, RG>(xs, wr);
  }
}

// 16 consecutive output frames of a lane as 4 packed dwords: keep the first n (any int), zero the rest.  Masked frames
// (t >= len) are code 0 for every consumer - what a zeroed accumulator requantises to (lo <= 0 <= hi, also through
// res_act) - so the mask is applied to the packed codes of the ONE tile per utterance that straddles the length
// (a uniform branch) instead of to every accumulator of a partial work-group (64 + 64 VALU instructions per pass).
__device__ __forceinline__ v4i sep2_mask16(v4i pk, int n) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = min(max(n - 4 * i, 0), 4);
    pk[i] &= m >= 4 ? -1 : (int)((1u << (8 * m)) - 1u);
  }
  return pk;
}
// does any accumulator leave [-2^21, 2^21)?  (v_max3 / v_min3: one instruction per two values and bound)
template <int MT>
__device__ __forceinline__ bool sep2_any_wide(const v16i (&a)[MT]) {
  int mx = a[0][0], mn = a[0][0];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      mx = max(max(mx, a[mt][r]), a[mt][r + 1]);
      mn = min(min(mn, a[mt][r]), a[mt][r + 1]);
    }
  return __any(mx >= (1 << 21) || mn < -(1 << 21));
}

template <int I, int N, class F>
__device__ __forceinline__ void sep2_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sep2_for<I + 1, N>(f);
  }
}
// sep2_gemm with a slice of other work behind every MFMA: cb(integral_constant<slot>), slot < 4 N MT.  The matrix
// pipe runs a 32x32x32 MFMA for 32 cycles while the wave issues the slice's VALU instructions (the previous unit's
// requantisation); sched_barrier keeps the compiler from gathering the slices behind the last MFMA.
template <int MT, int N, int N0, int AB, class CB>
__device__ __forceinline__ void sep2_gemm_cb(v16i* __restrict__ acc, v4i (&wf)[16], const lds_u8* img_lane, int mt_stride,
                                             const v4i* __restrict__ r0, const v4i* __restrict__ r1, CB&& cb) {
  constexpr int NQ = 4 * N / AB;
  v4i a[2][MT][AB];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int i = 0; i < AB; ++i) a[0][mt][i] = sep2_a_frag(img_lane + mt * mt_stride, i);
  sep2_for<0, NQ>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    if constexpr (q + 1 < NQ) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < AB; ++i) a[(q + 1) & 1][mt][i] = sep2_a_frag(img_lane + mt * mt_stride, AB * (q + 1) + i);
    }
    // one slice behind EVERY MFMA (slot (q AB + i) MT + mt): a wave issues in order, so a longer slice behind a batch of
    // MFMAs runs past the last MFMA's 32 cycles and the matrix pipe idles until the next batch can issue
    sep2_for<0, AB * MT>([&](auto sc) {
      constexpr int sl = decltype(sc)::value, i = sl / MT, mt = sl % MT;
      acc[mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[q & 1][mt][i], wf[AB * q + i], acc[mt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      cb(std::integral_constant<int, q * AB * MT + sl>{});
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr ((AB * (q + 1)) % 4 == 0) {
      constexpr int g = (AB * (q + 1)) / 4 - 1;
      if constexpr (g < N0) {
        if (r0) sep2_load_wg(&wf[4 * g], r0 + 256 * g);
      } else {
        if (r1) sep2_load_wg(&wf[4 * g], r1 + 256 * g);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  });
}

// per-lane (= per output channel, MFMA C layout: channel = lane & 31) parameters of one 256-channel pass
struct Sep2PassP {
  int bias, pbias;
  double Mo[QASR_MAX_OUTS];                                  // PLAIN: the consumers' per-channel multipliers
  double Mm, Mp;                                             // RESADD: main / residual multipliers towards res_act
  float sbm, sbp;                                            // EXACT_Z: conv output scales
};

// K taps, NG groups of 128 input channels (cin_pad = 128 NG), NGP groups of the residual 1x1 conv (0: no res_act),
// NP passes of 256 output channels - all compile-time: the whole kernel is straight-line code, which is what lets the
// compiler count its s_waitcnt vmcnt(N) exactly instead of draining every prefetch at each join
// The work of ONE work-group: utterance b, frames [t0, t0 + TT), every channel.  k_sep2 below runs it once per work-group
// of a (B, Tp / TT) grid (round 3's persistent launch, removed since, walked layers and time tiles with it).  `p` may live
// in the kernel-argument segment or in global memory: its address is wave-uniform either way (scalar loads).
// DIL = 2 (QuartzNet's block 16, k = 87): out[t] = sum_m w[m] x[t - 86 + 2 m] only touches frames of t's parity, so every
// channel is staged as TWO rows - its even and its odd frames - and each row sees an ordinary dilation-1 conv over TT / 2
// samples with the same taps: a depthwise group is 8 real channels x 2 parities per wave (16 MFMA blocks as ever), the
// geometry is Sep2Geo<K, TT / 2>, and the requantised values go to the [channel][frame] image byte by byte (a lane's four
// values are frames of one parity, two apart).  Everything behind the depthwise stage is unchanged.
template <int K, int NG, int NGP, int NP, bool DBG, int TT, int DIL = 1>
__device__ __forceinline__ void sep2_body(const SepP& p, const int b, const int t0, const bool stamp_wg, const int wg_id) {
  using G = Sep2Geo<K, TT / DIL>;
  constexpr bool RES = NGP > 0;
  static_assert(DIL == 1 || (DIL == 2 && !RES && TT >= 64), "dilation 2: plain layers, 64- / 128-frame tiles");
  constexpr int MT = TT / 32;
  constexpr int NU = G::NU, S = G::S, NS = G::NS;
  constexpr int CIN_PAD = 128 * NG, PCIN_PAD = 128 * NGP;
  constexpr int RCH = 16 / DIL, GCH = SEP2_CH / DIL;         // real channels per wave / per work-group in one depthwise group
  constexpr int NCHUNK = (CIN_PAD + GCH - 1) / GCH;
  // staging granules of a wave and group: DIL 1: 16 rows x NPG granules of the row itself; DIL 2: RCH real channels x the
  // granules of the REAL window [t0 - 2 HALO, t0 + TT + 2 HALO) (each splits into 8 even + 8 odd bytes); tap rows: RCH x KS bytes
  constexpr int NPGR = DIL == 1 ? G::NPG : (TT + 2 * DIL * G::HALO) / 16;
  constexpr int NPT = (RCH * NPGR + 63) / 64, NTG = RCH * G::KS / 16, NTT = (NTG + 63) / 64;
  static_assert(DIL == 1 || (8 * NPGR == G::WLEN && (RCH * G::KS) % 16 == 0), "dilation 2: window / tap granules");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const EpiP& e = p.e;
  int tid_ = threadIdx.x;
  // (kept from the persistent-launch experiment: every lane-derived constant is recomputed inside the call - hoisted out of a layer loop they
  //  would stay live across all bodies and push the 245-251 VGPR ones into spilling)
  asm volatile("" : "+v"(tid_));
  const int tid = tid_, lane = tid & 63, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  lds_u8* const Xd = (lds_u8*)smem;                          // [MT][CIN_PAD][32]  A image of the 1x1 conv
  // depthwise operands are wave-private: wave w stages the window and tap rows of ITS 16 channels of a group in its own
  // region [16][WP] + [16][KS] - no work-group barrier before the A image is complete; afterwards the region holds the
  // residual A image
  lds_u8* const Un = Xd + TT * CIN_PAD;
  lds_u8* const Wsw = Un + wave * G::WREG;
  lds_u8* const Tlw = Wsw + 16 * G::WP;

  const unsigned flags = e.flags;
  const int eT = e.T, eTp = e.Tp, ecout = e.cout, n_outs = e.n_outs;
  const int len_b = e.lens[b];
  const int lim = (flags & QASR_F_MASK_OUT) ? min(len_b, eT) : eT;
  const int dlim = min(len_b, eT);                           // the 1x1 conv's MaskedConv1d masks its input
  const bool f_relu = flags & QASR_F_RELU;
  const bool f_exact = flags & QASR_F_EXACT_Z;
  int dw_lo = p.dw_lo, dw_hi = p.dw_hi;
  // fetched NOW: left to the compiler the scalar load sits in front of the first requantisation, and its
  // s_waitcnt lgkmcnt(0) (scalar loads return out of order) also waits for the next group's LDS reads issued just before
  asm volatile("" : "+v"(dw_lo), "+v"(dw_hi));               // (as VGPRs: v_med3_i32's operands; "+s" breaks the debug build)
  const bool stamp = p.prof && (p.prof_mode & 255) == 0 && stamp_wg && tid == 0;
  const bool tline = p.prof && (p.prof_mode & 255) == 1 && tid == 0;
  const int tune = p.prof_mode >> 8;                          // QASR_SEP2_TUNE (experiments)
  long long tl_start = 0, tl_clk = 0;
  if (tline) {
    tl_start = (long long)__builtin_amdgcn_s_memrealtime();
    tl_clk = (long long)__builtin_amdgcn_s_memtime();
  }
  int nst = 0;
#define STAMP2() do { if (stamp && nst < 31) p.prof[nst++] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
  STAMP2();

  // Shapes are exact (sep2_shape_ok): cin == CIN_PAD, cout == 256 NP, residual cin == PCIN_PAD - no row / channel guards.
  // ------------------------------------------------------------------------------------------ global requests
  // A CU's texture path takes one 64-lane x 16-B request per 16 cycles and a work-group needs ~350 of them (0.36 MB):
  // a wave that issues its share in one burst sits in that queue for thousands of cycles (phase stamps).  Only what
  // the first depthwise chunk needs is requested up front; everything else (next chunk's window and taps, the
  // residual operand, the weight slab of GEMM pass 0) is a list of single requests issued a few at a time between
  // the depthwise math (pf<CH>(lo, hi)).
  const unsigned flip = p.x_unsigned ? 0x80808080u : 0u;
  constexpr int NRT = RES ? (TT * PCIN_PAD / 16) / SEP2_NT : 0;
  v4i pc[NPT], pt[NTT], rr[NRT > 0 ? NRT : 1], wf[16];
  const int co_l = 32 * wave + (lane & 31);                  // row inside a 256-channel pass
  const v4i* const w0 = w_frag(p.w, CIN_PAD, co_l, 0);
  // window + tap rows of this wave's 16 channels of group g (channels 128 g + 16 wave ..): 16-B granules, lane-contiguous.
  // Everything per lane is the same for every group and computed once - byte offsets of its granules from the group's
  // (wave-uniform) row 0, zero-padding masks, LDS addresses: a group's staging is loads, one bit-op per dword, stores
  int woff[NPT], toff[NTT];
  unsigned wkeep[NPT];
  lds_u8* wlds[NPT];
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    const int pi = lane + 64 * i;
    const int row = pi / NPGR, col = pi - row * NPGR;
    const int t = t0 - DIL * G::HALO + 16 * col;             // a granule lies entirely inside or outside [0, Tp)
    // unconditional loads from clamped addresses (the keep mask zeroes what lies outside): a load under a branch with a
    // zero-initialised destination is waited for on the spot
    woff[i] = min(row, RCH - 1) * eTp + min(max(t, 0), eTp - 16);
    wkeep[i] = (t >= 0 && t < eTp) ? 0xffffffffu : 0u;       // conv zero padding beyond the tensor
    // DIL 2: the granule's even bytes go to row 2 r (8 bytes at 8 col), its odd bytes to row 2 r + 1
    wlds[i] = DIL == 1 ? Wsw + min(row, 15) * G::WP + 16 * col : Wsw + 2 * min(row, RCH - 1) * G::WP + 8 * col;
    asm volatile("" : "+v"(wkeep[i]));                       // a mask, not a predicate: (v & keep) ^ flip is ONE v_bitop3_b32
  }
#pragma unroll
  for (int i = 0; i < NTT; ++i) toff[i] = 16 * min(lane + 64 * i, NTG - 1);
  auto ld_grp = [&](int g) {
    const int cw = GCH * g + RCH * wave;                     // (wave-uniform: the bases below are scalar)
    const int8_t* const xg = p.x + ((size_t)b * CIN_PAD + cw) * eTp;
    const unsigned char* const tg = (const unsigned char*)p.wdw2 + (size_t)cw * G::KS;   // zero-margined tap rows [C][KS]
#pragma unroll
    for (int i = 0; i < NPT; ++i) pc[i] = *(const v4i*)(xg + woff[i]);
#pragma unroll
    for (int i = 0; i < NTT; ++i) pt[i] = *(const v4i*)(tg + toff[i]);
  };
  auto ld_res = [&](int i) {                                 // residual operand granule: channel gi / (TT/16), 16 frames
    if constexpr (RES) {
      const int gi = tid + SEP2_NT * i;
      const int c = gi / (TT / 16), q = gi - c * (TT / 16);
      rr[i] = *(const v4i*)(p.panes[0].x + ((size_t)b * PCIN_PAD + c) * eTp + t0 + 16 * q);
    }
  };
  // request list of depthwise group CH besides the next group's operands: [residual operand, last group] [this group's
  // share of the weight slab of GEMM pass 0]
  constexpr int SLC = NCHUNK >= 4 ? 3 : 1;                   // groups the slab's requests are spread over
  auto pf = [&](auto chc, int lo, int hi) {
    constexpr int CH = decltype(chc)::value;
    constexpr int n_res = CH + 1 == NCHUNK ? NRT : 0;
    constexpr int s0 = CH < SLC ? 4 * NG * CH / SLC : 0, n_slab = CH < SLC ? 4 * NG * (CH + 1) / SLC - s0 : 0;
#pragma unroll
    for (int i = lo; i < hi; ++i) {
      if (i < n_res) ld_res(i);
      else if (i < n_res + n_slab) wf[s0 + i - n_res] = w0[64 * (s0 + i - n_res)];
    }
  };
  auto commit = [&]() {                                      // registers -> this wave's LDS rows
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      if (64 * i + 63 < RCH * NPGR || lane + 64 * i < RCH * NPGR) {
        v4i v = pc[i];
        v[0] = (v[0] & wkeep[i]) ^ flip; v[1] = (v[1] & wkeep[i]) ^ flip; v[2] = (v[2] & wkeep[i]) ^ flip; v[3] = (v[3] & wkeep[i]) ^ flip;
        if constexpr (DIL == 1) {
          *(lds_v4i*)wlds[i] = v;
        } else {                                             // frames of one parity: bytes 0, 2, 4, 6 / 1, 3, 5, 7 of each dword pair
          const v2i ev = {(int)__builtin_amdgcn_perm((unsigned)v[1], (unsigned)v[0], 0x06040200u),
                          (int)__builtin_amdgcn_perm((unsigned)v[3], (unsigned)v[2], 0x06040200u)};
          const v2i od = {(int)__builtin_amdgcn_perm((unsigned)v[1], (unsigned)v[0], 0x07050301u),
                          (int)__builtin_amdgcn_perm((unsigned)v[3], (unsigned)v[2], 0x07050301u)};
          *(lds_v2i*)wlds[i] = ev;
          *(lds_v2i*)(wlds[i] + G::WP) = od;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NTT; ++i) {
      const int gi = lane + 64 * i;
      if (64 * i + 63 < NTG || gi < NTG) *(lds_v4i*)(Tlw + 16 * gi) = pt[i];
    }
  };
  // K == 0 (a bare 1x1 conv, e.g. block 17 of QuartzNet): no depthwise stage - the [channel][frame] tile of `x` IS the
  // A image, a plain 16-byte-granule copy (x_unsigned bytes take the -128 flip on the way)
  constexpr int NXT = K == 0 ? (TT * CIN_PAD / 16) / SEP2_NT : 1;
  v4i xg[NXT];
  if constexpr (K == 0) {
#pragma unroll
    for (int i = 0; i < NXT; ++i) {
      const int gi = tid + SEP2_NT * i;
      const int c = gi / (TT / 16), q = gi - c * (TT / 16);
      xg[i] = *(const v4i*)(p.x + ((size_t)b * CIN_PAD + c) * eTp + t0 + 16 * q);
    }
#pragma unroll
    for (int j = 0; j < 4 * NG; ++j) wf[j] = w0[64 * j];       // the whole first slab right behind the tile
  } else {
    ld_grp(0);
  }
  STAMP2();                                                  // requests of group 0 issued

  // per-group depthwise parameters of this lane's channels and the per-lane parameters of every GEMM pass: a handful
  // of registers, requested right behind chunk 0
  const int cb = lane >> 2, jl = lane & 3;
  int dbias[NCHUNK];
  double dM[NCHUNK];
  if constexpr (K > 0) {
#pragma unroll
    for (int gi = 0; gi < NCHUNK; ++gi) {
      const int c = GCH * gi + RCH * wave + cb / DIL;        // a wave owns 16 / DIL channels of every chunk
      dbias[gi] = p.bias_dw[c];
      dM[gi] = p.m_dw[c];
    }
  }
  Sep2PassP pps[NP];
  auto load_pps = [&]() {                                    // (issued behind the first group's operands: not on the way
#pragma unroll                                               //  to the first MFMA)
  for (int ps = 0; ps < NP; ++ps) {
    const int cor = 256 * ps + co_l;
    Sep2PassP& q = pps[ps];
    q.bias = p.bias[cor];
    if constexpr (RES) {
      q.pbias = p.panes[0].bias[cor];
      q.Mm = e.m_main[cor];
      q.Mp = p.panes[0].m[cor];
      q.sbp = f_exact ? p.panes[0].sb[cor] : 1.0f;
    } else {
#pragma unroll
      for (int j = 0; j < QASR_MAX_OUTS; ++j) q.Mo[j] = j < n_outs ? e.outs[j].mtab[cor] : 0.0;
    }
    q.sbm = f_exact ? e.sb[cor] : 1.0f;
  }
  };
  if constexpr (K == 0) load_pps();

  // ------------------------------------------------------------------------------------------ depthwise stage
  // v_mfma_i32_4x4x4_16B_i8: 16 independent 4x4x4 products, block = channel.  A[i][k] = w[m0 + k - i] (lane i's own
  // shifted tap stream), B[k][j] = win[S j + D + m0 + k] (lane j's window run), so the block accumulates
  // out[S j + 4 u + i] over taps m0 - i .. m0 - i + 3 for chain u when B is taken 4 u bytes further on; m0 advances
  // by 4 per step.  Lane l: channel l >> 2, row / column l & 3; register v of chain u = frame S (l & 3) + 4 u + v.
  const bool full_in = t0 + TT <= dlim;                      // no masked frame in this tile (uniform)
  constexpr int e0base = 8 + G::MS;                          // the lane's tap stream starts at byte e0base - jl of its row
  const int e0 = e0base - jl, tq = e0 >> 2, tsh = e0 & 3;
  unsigned fmask[NU];                                        // bytes of output dword u that lie below the utterance's length
  {
    const int dl = dlim - t0 - S * jl;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int n = min(max(dl - 4 * u, 0), 4);
      fmask[u] = n >= 4 ? 0xffffffffu : ((1u << (8 * n)) - 1u);
    }
  }
  struct DwIn {                                              // LDS operands of one group of 16 channels
    unsigned raw[NS + 1];                                    // the lane's tap dwords (whole words, funnel-shifted later)
    unsigned xs[G::NRD * (G::RG / 4)];                       // the lane's window run: every dword feeds the NU chains
  };
  // the lane's two LDS streams, each behind a base register of its own: every read is base + immediate offset
  const lds_u32* tr_lane = (const lds_u32*)(Tlw + (cb / DIL) * G::KS + 4 * tq);   // (DIL 2: both parities read their channel's taps)
  const lds_u8* wr_lane = Wsw + cb * G::WP + S * jl + (G::A0 & ~(G::RG - 1));
  asm volatile("" : "+v"(tr_lane), "+v"(wr_lane));
  auto dw_read = [&](DwIn& in) {                             // this wave's 16 channels: row = channel, lane (cb, jl)
    const lds_u32* tr = tr_lane;
#pragma unroll
    for (int i = 0; i <= NS; ++i) in.raw[i] = tr[i];
    const lds_u8* wr = wr_lane;
    sep2_rd_stream<G::OFF, G::NE, G::RG>(in.xs, wr);
  };
  // the MFMAs of one group; `pf3(k)` issues the k-th third of the group's request list
  auto dw_mfma = [&](const DwIn& in, v4i (&acc)[NU], int bias, auto&& pf3) {
    unsigned tw[NS];
#pragma unroll
    for (int st = 0; st < NS; ++st) tw[st] = __builtin_amdgcn_alignbyte(in.raw[st + 1], in.raw[st], tsh);
#pragma unroll
    for (int u = 0; u < NU; ++u) acc[u] = (v4i){bias, bias, bias, bias};
    pf3(0);
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      if (st == NS / 2) pf3(1);
#pragma unroll
      for (int u = 0; u < NU; ++u)
        acc[u] = __builtin_amdgcn_mfma_i32_4x4x4i8((int)tw[st], (int)in.xs[G::OFF + u + st], acc[u], 0, 0, 0);
    }
    pf3(2);
  };
  // masks, requantisation and the [channel][frame] image of one group
  auto dw_out = [&](v4i (&acc)[NU], int c0, double Mg) {
    if constexpr (DIL == 2) {
      // lane (cb, jl): channel cb >> 1, parity cb & 1, stream samples S jl + 4 u + v = frames 2 (..) + parity: one byte each
      // (masked frames are zeroed in one pass over the finished image of the ONE tile that straddles the length)
      const int c = c0 + RCH * wave + (cb >> 1), par = cb & 1;
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int f = 2 * (S * jl + 4 * u + v) + par;
          if (DBG && p.dw_acc_dbg && t0 + f < eT) p.dw_acc_dbg[((size_t)b * CIN_PAD + c) * eTp + t0 + f] = acc[u][v];
          *(lds_u8*)(Xd + (f >> 5) * (CIN_PAD * 32) + c * 32 + (f & 31)) = (unsigned char)rq_clamp(acc[u][v], Mg, dw_lo, dw_hi);
        }
      return;
    }
    const int c = c0 + 16 * wave + cb;
    if (DBG && p.dw_acc_dbg) {
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int f = S * jl + 4 * u + v;
          if (t0 + f < eT) p.dw_acc_dbg[((size_t)b * CIN_PAD + c) * eTp + t0 + f] = acc[u][v];
        }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int f = S * jl + 4 * u;                          // first of this dword's 4 frames
      const unsigned w = pack4b(rq_clamp(acc[u][0], Mg, dw_lo, dw_hi), rq_clamp(acc[u][1], Mg, dw_lo, dw_hi),
                                rq_clamp(acc[u][2], Mg, dw_lo, dw_hi), rq_clamp(acc[u][3], Mg, dw_lo, dw_hi));
      // masked frames (t >= len): code 0 (= what a zeroed accumulator requantises to, dw_lo <= 0 <= dw_hi) - one AND per
      // dword with the lane's precomputed byte masks, branch-free, so that this block can be interleaved with MFMAs
      *(lds_u32*)(Xd + (f >> 5) * (CIN_PAD * 32) + c * 32 + (f & 31)) = w & fmask[u];
    }
  };

  // MFMAs of one group with ONE instruction of the previous group's output work (dw_out, spelled out as single
  // instructions: cvt / fma / med3 per value, perm / perm / or / and / ds_write per dword) pinned behind each MFMA
  auto dw_mfma_ilv = [&](const DwIn& in, v4i (&acc)[NU], int bias, auto&& pf3, v4i (&prev)[NU], int cprev, double Mprev) {
    constexpr int NOPS = 17 * NU, OPM = (NOPS + NS * NU - 1) / (NS * NU);   // instructions behind each MFMA (1; 2 for k33 / k39)
    const int c = cprev + 16 * wave + cb;
#pragma unroll
    for (int u = 0; u < NU; ++u) acc[u] = (v4i){bias, bias, bias, bias};
    unsigned tws[2] = {__builtin_amdgcn_alignbyte(in.raw[1], in.raw[0], tsh), 0u}, P1 = 0, P2 = 0;
    double dreg = 0.0;
    int q[4] = {0, 0, 0, 0};
    __builtin_amdgcn_sched_barrier(0);
    sep2_for<0, NS * NU>([&](auto ic) {
      constexpr int i = decltype(ic)::value, st = i / NU, u = i % NU;
      if constexpr (u == 0) {
        if constexpr (st == 0) pf3(0);
        if constexpr (st == NS / 2) pf3(1);
      }
      // the next step's tap-stream shift half a step ahead, into the other register: computed where it is used it cost an
      // s_nop (VALU write -> MFMA read) per step
      if constexpr (u == NU / 2 && st + 1 < NS) tws[(st + 1) & 1] = __builtin_amdgcn_alignbyte(in.raw[st + 2], in.raw[st + 1], tsh);
      acc[u] = __builtin_amdgcn_mfma_i32_4x4x4i8((int)tws[st & 1], (int)in.xs[G::OFF + u + st], acc[u], 0, 0, 0);
      sep2_for<OPM * i, OPM * (i + 1)>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if constexpr (j < NOPS) {
        constexpr int dw = j / 17, r = j % 17;
        if constexpr (r < 12) {
          constexpr int val = r / 3, ph = r % 3;
          if constexpr (ph == 0) dreg = (double)prev[dw][val];
          if constexpr (ph == 1) dreg = __builtin_fma(dreg, Mprev, MAGIC_RNE);
          if constexpr (ph == 2) q[val] = med3i(__double2loint(dreg), dw_lo, dw_hi);
        } else if constexpr (r == 12) {
          P1 = __builtin_amdgcn_perm((unsigned)q[1], (unsigned)q[0], 0x0c0c0400u);
        } else if constexpr (r == 13) {
          P2 = __builtin_amdgcn_perm((unsigned)q[3], (unsigned)q[2], 0x04000c0cu);
        } else if constexpr (r == 14) {
          P1 |= P2;
        } else if constexpr (r == 15) {
          P1 &= fmask[dw];
        } else {
          constexpr int f0 = 4 * dw;                         // frame S jl + 4 dw: tile (S jl + 4 dw) >> 5, column & 31
          const int f = S * jl + f0;
          *(lds_u32*)(Xd + (f >> 5) * (CIN_PAD * 32) + c * 32 + (f & 31)) = P1;
        }
      }
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    pf3(2);
  };

  // Depthwise operands are wave-private, so nothing in this stage involves another wave: the two waves of a SIMD drift
  // apart and one's MFMAs run beside the other's requantisation / LDS traffic / waits (with work-group barriers between
  // 256-channel chunks every wave waited ~1.4 k cycles per barrier for its SIMD partner's MFMAs - phase stamps).
  // Software pipeline of a wave: MFMAs of group g | rows of group g+1 -> LDS, request group g+2, read the lane streams of
  // g+1 back (the LDS round trip runs under:) | requantisation of group g | MFMAs of group g+1 ...
  DwIn in;
  auto stage = [&](auto chc) {                               // rows of group CH: registers -> LDS -> lane streams
    constexpr int CH = decltype(chc)::value;
    commit();
    if constexpr (CH + 1 < NCHUNK) ld_grp(CH + 1);
    // the wave's own LDS writes -> its own reads: LDS executes a wave's instructions in order; the fences keep the
    // compiler from moving the reads above the writes of other lanes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    dw_read(in);
    __builtin_amdgcn_sched_barrier(0);                       // the reads are issued here; their first use is the next group
  };
  // The SIMD issues one VALU or MFMA instruction per ~4.6 cycles whichever wave it comes from, a 4x4x4 MFMA occupies the
  // matrix pipe for 8.3, and the older wave of a SIMD pair wins every issue slot it can use (profiles/microbench/
  // coissue.hip: two waves alternating 168 MFMAs / 240 VALU instructions take exactly the sum of their times) - VALU work
  // hides only in the shadow of the SAME wave's MFMAs.  So the requantisation of group g-1 (independent of group g's
  // MFMAs) is issued one instruction behind every MFMA of group g (dw_mfma_ilv; a sched_group_barrier pipeline over
  // dw_mfma + dw_out hoisted all 21 tap-stream shifts, spilled, and placed a VALU instruction only every ~20 MFMAs).
#ifndef SEP2_ILV
#define SEP2_ILV 1
#endif
#ifndef SEP2_TAIL1
#define SEP2_TAIL1 0                    /* 1: the last GEMM unit of a plain layer is one 32-frame tile (shorter exposed epilogue) */
#endif
  constexpr bool ILV = SEP2_ILV && (!RES || TT <= 64) && DIL == 1;   // (register budget of the block-end forms at 128 frames)
  v4i accs[ILV ? 2 : 1][NU];
  auto chunk = [&](auto chc) {
    constexpr int CH = decltype(chc)::value;
    constexpr int c0 = GCH * CH;
    constexpr int NPF = (CH + 1 == NCHUNK ? NRT : 0) + (CH < SLC ? 4 * NG * (CH + 1) / SLC - 4 * NG * CH / SLC : 0);
    constexpr int Q = (NPF + 2) / 3;                         // three issue points per group
    if constexpr (ILV) {
      if constexpr (CH == 0) dw_mfma(in, accs[0], dbias[0], [&](int k) { pf(chc, Q * k, Q * (k + 1)); });
      else {
        if (DBG && p.dw_acc_dbg) {                           // (debug builds: the plain order, accumulators dumped)
          dw_out(accs[(CH - 1) & 1], c0 - GCH, dM[CH - 1]);
          dw_mfma(in, accs[CH & 1], dbias[CH], [&](int k) { pf(chc, Q * k, Q * (k + 1)); });
        } else {
          dw_mfma_ilv(in, accs[CH & 1], dbias[CH], [&](int k) { pf(chc, Q * k, Q * (k + 1)); }, accs[(CH - 1) & 1],
                      c0 - GCH, dM[CH - 1]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      STAMP2();
      if constexpr (CH + 1 < NCHUNK) stage(std::integral_constant<int, CH + 1>{});
      else dw_out(accs[CH & 1], c0, dM[CH]);
      STAMP2();
    } else {
      dw_mfma(in, accs[0], dbias[CH], [&](int k) { pf(chc, Q * k, Q * (k + 1)); });
      STAMP2();
      if constexpr (CH + 1 < NCHUNK) stage(std::integral_constant<int, CH + 1>{});
      dw_out(accs[0], c0, dM[CH]);
      __builtin_amdgcn_sched_barrier(0);                     // (keeps the next group's tap-stream shifts - the first use of
      STAMP2();                                              //  the reads above - behind this group's requantisation)
    }
  };
  if constexpr (K > 0) {
    if (stamp) {                                             // diagnostics: when did group 0's window / taps land?
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      STAMP2();
    }
    if (tune & 1) {                                          // strict priority for one wave of every SIMD pair
      if (wave < 4) __builtin_amdgcn_s_setprio(3);
    }
    stage(std::integral_constant<int, 0>{});
    load_pps();
    if (tune & 2) {                                          // stagger the SIMD partners by (tune >> 4) x 64 cycles
      if (wave >= 4)
        for (int i = 0; i < ((tune >> 4) & 127); ++i) __builtin_amdgcn_s_sleep(1);
    }
    STAMP2();
    sep2_for<0, NCHUNK>([&](auto chc) { chunk(chc); });
  } else {
    static_assert(K > 0 || !RES, "the bare 1x1 form has no residual variant");
    const unsigned xflip = p.pw_unsigned ? 0x80808080u : 0u;
#pragma unroll
    for (int i = 0; i < NXT; ++i) {
      const int gi = tid + SEP2_NT * i;
      const int c = gi / (TT / 16), q = gi - c * (TT / 16);
      v4i v = xg[i];
      v[0] ^= xflip; v[1] ^= xflip; v[2] ^= xflip; v[3] ^= xflip;
      *(lds_v4i*)(Xd + (q >> 1) * (CIN_PAD * 32) + c * 32 + 16 * (q & 1)) = v;
    }
  }
  static_assert(NCHUNK <= 4 * DIL, "more than 512 input channels");
  if ((tune & 5) == 1) __builtin_amdgcn_s_setprio(0);        // (tune & 4: the priority stays through the GEMM passes)
  __syncthreads();                                           // Xd complete, windows dead
  if constexpr (DIL == 2) {
    if (t0 + TT > dlim) {                                    // (uniform) MaskedConv1d of the 1x1 conv: frames >= len are code 0
      for (int r = tid; r < MT * CIN_PAD; r += SEP2_NT) {
        const int mt = r / CIN_PAD, n = dlim - t0 - 32 * mt;  // valid frames of this 32-frame row
        if (n < 32) {
          lds_v4i* row = (lds_v4i*)(Xd + (size_t)r * 32);
          row[0] = sep2_mask16(row[0], n);
          row[1] = sep2_mask16(row[1], n - 16);
        }
      }
      __syncthreads();
    }
  }
  if constexpr (RES) {                                       // residual A image [PCIN_PAD][32] per 32-frame tile
    const unsigned rflip = p.panes[0].x_unsigned ? 0x80808080u : 0u;
#pragma unroll
    for (int i = 0; i < NRT; ++i) {
      const int gi = tid + SEP2_NT * i;
      const int c = gi / (TT / 16), q = gi - c * (TT / 16);
      v4i v = rr[i];
      v[0] ^= rflip; v[1] ^= rflip; v[2] ^= rflip; v[3] ^= rflip;
      *(lds_v4i*)(Un + (q >> 1) * (PCIN_PAD * 32) + c * 32 + 16 * (q & 1)) = v;
    }
    __syncthreads();
  }
  STAMP2();
  // ------------------------------------------------------------------------------------------ 1x1 GEMM passes of 256 channels
  const lds_u8* const xd_lane = Xd + sep2_a_lane_off(lane);
  const lds_u8* const xr_lane = Un + sep2_a_lane_off(lane);
  int32_t* const pdbg = RES ? p.panes[0].acc_dbg : nullptr;
  const int qlo = f_relu ? max(e.qlo, 0) : e.qlo, qhi = e.qhi;
  // Plain layers with one consumer: the requantisation of one unit (half the frame tiles of a pass; the whole pass at
  // 32 / 64-frame tiles with a single tile per half) runs in slices behind the MFMA batches of the next unit's GEMM -
  // the epilogue is VALU work (fp64 fma + med3 per value), the GEMM matrix-pipe work, and a work-group's phases are
  // otherwise serial.  Only the first GEMM unit and the last unit's epilogue stay exposed.
  constexpr bool PIPED = !RES && (NP * (MT >= 2 ? 2 : 1) > 1);   // (sep2_shape_ok: plain layers have exactly one consumer)
  if constexpr (PIPED) {
    {
      // Units: a pass is split into NH units of MH frame tiles each.  SEP2_TAIL1: the LAST unit of the last pass is split once
      // more into two single-tile units - the only requantise + pack + store nothing hides is then one 32-frame tile (16 values
      // per lane, one 16-byte store) instead of two, and half as many bytes are outstanding when the wave ends.
      constexpr int MH = MT >= 2 ? MT / 2 : 1, NH = MT / MH;
      constexpr bool SPLIT = SEP2_TAIL1 && MH == 2;
      constexpr int NUN0 = NP * NH, NUN = NUN0 + (SPLIT ? 1 : 0);
      const OutP& o0 = e.outs[0];
      const int olo = o0.lo, ohi = o0.hi;
      int8_t* const optr = (int8_t*)o0.ptr;
      v16i accs[2][MH];
      int q4[4];
      unsigned P[4];
      // accumulator hooks and EXACT_Z of a finished unit (tiles mt0 .. mt0 + mh - 1 of pass ps)
      auto unit_finish = [&](v16i* a, int ps, int mt0, auto mhc) __attribute__((always_inline)) {
        constexpr int mh = decltype(mhc)::value;
        const int co = 256 * ps + co_l;
        if (DBG && e.acc_dbg) {
#pragma unroll
          for (int mt = 0; mt < mh; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int t = t0 + 32 * (mt0 + mt) + mfma32_row(r, h);
              if (t < eT) e.acc_dbg[((size_t)b * ecout + co) * eTp + t] = a[mt][r];
            }
        }
        if (f_exact) {
          int mx = a[0][0], mn = a[0][0];
#pragma unroll
          for (int mt = 0; mt < mh; ++mt)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              mx = max(max(mx, a[mt][r]), a[mt][r + 1]);
              mn = min(min(mn, a[mt][r]), a[mt][r + 1]);
            }
          if (__any(mx >= (1 << 21) || mn < -(1 << 21))) {
#pragma unroll
            for (int mt = 0; mt < mh; ++mt)
#pragma unroll
              for (int r = 0; r < 16; ++r) a[mt][r] = z_roundtrip(a[mt][r], pps[ps].sbm, f_relu);
          }
        }
      };
      // values [v0, v1) of a finished unit: requantise; a completed tile leaves as one 16-byte store per lane
      auto unit_values = [&](v16i* a, int ps, int mt0, auto v0c, auto v1c) __attribute__((always_inline)) {
        constexpr int v0 = decltype(v0c)::value, v1 = decltype(v1c)::value;
        sep2_for<v0, v1>([&](auto vc) {
          constexpr int v = decltype(vc)::value, mt = v / 16, r = v % 16;
          q4[r & 3] = rq_clamp(a[mt][r], pps[ps].Mo[0], olo, ohi);
          if constexpr ((r & 3) == 3) P[r >> 2] = pack4b(q4[0], q4[1], q4[2], q4[3]);
          if constexpr (r == 15) {
            const auto s02 = __builtin_amdgcn_permlane32_swap(P[0], P[2], false, false);
            const auto s13 = __builtin_amdgcn_permlane32_swap(P[1], P[3], false, false);
            v4i pk = {(int)s02[0], (int)s02[1], (int)s13[0], (int)s13[1]};
            if (t0 + 32 * (mt0 + mt + 1) > lim) pk = sep2_mask16(pk, lim - (t0 + 32 * (mt0 + mt) + 16 * h));   // (uniform)
            *(v4i*)(optr + ((size_t)b * ecout + 256 * ps + co_l) * eTp + t0 + 32 * (mt0 + mt) + 16 * h) = pk;
          }
        });
      };
      // (pass, first tile, tiles) of unit u
      auto u_ps = [](int u) constexpr { return u < NUN0 ? u / NH : NP - 1; };
      auto u_mh = [](int u) constexpr { return (SPLIT && u >= NUN0 - 1) ? 1 : MH; };
      auto u_mt0 = [](int u) constexpr { return u < NUN0 ? (u % NH) * MH : (NH - 1) * MH + 1; };
      sep2_for<0, NUN>([&](auto uc) {
        constexpr int u = decltype(uc)::value, ps = u_ps(u), mh = u_mh(u), mt0 = u_mt0(u);
        constexpr int AB = mh >= 2 ? 1 : 2, NSL = 4 * NG * mh;   // (AB: register budget at two tiles per unit)
        v16i* const cur = accs[u & 1];
#pragma unroll
        for (int mt = 0; mt < mh; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) cur[mt][r] = pps[ps].bias;
        // the slab is re-requested with the next pass's rows of this wave while the pass's last unit multiplies
        const v4i* const wnext = (u < NUN0 && u % NH == NH - 1 && ps + 1 < NP) ? w_frag(p.w, CIN_PAD, 256 * (ps + 1) + co_l, 0) : nullptr;
        if constexpr (u > 0) unit_finish(accs[(u - 1) & 1], u_ps(u - 1), u_mt0(u - 1), std::integral_constant<int, u_mh(u - 1)>{});
        sep2_gemm_cb<mh, NG, 0, AB>(cur, wf, xd_lane + mt0 * (CIN_PAD * 32), CIN_PAD * 32, nullptr, wnext, [&](auto qc) {
          if constexpr (u > 0) {
            constexpr int NV = 16 * u_mh(u - 1), VPS = (NV + NSL - 1) / NSL;
            constexpr int q = decltype(qc)::value, v0 = q * VPS < NV ? q * VPS : NV, v1 = (q + 1) * VPS < NV ? (q + 1) * VPS : NV;   // q: MFMA slot
            unit_values(accs[(u - 1) & 1], u_ps(u - 1), u_mt0(u - 1), std::integral_constant<int, v0>{}, std::integral_constant<int, v1>{});
          }
        });
        STAMP2();
      });
      unit_finish(accs[(NUN - 1) & 1], NP - 1, u_mt0(NUN - 1), std::integral_constant<int, u_mh(NUN - 1)>{});
      unit_values(accs[(NUN - 1) & 1], NP - 1, u_mt0(NUN - 1), std::integral_constant<int, 0>{}, std::integral_constant<int, 16 * u_mh(NUN - 1)>{});
      STAMP2();
    }
  }
  if constexpr (!PIPED) {
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    const int cbase = 256 * ps;
    const int co = cbase + co_l, cor = co;
    constexpr bool co_ok = true;
    const bool more = ps + 1 < NP;
    const v4i* const wnext = more ? w_frag(p.w, CIN_PAD, co + 256, 0) : nullptr;   // this wave's rows in the next pass
    const v4i* const wpane = RES ? w_frag(p.panes[0].w, PCIN_PAD, cor, 0) : nullptr;
    const Sep2PassP& pp = pps[ps];
    v16i acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][r] = pp.bias;
    // once multiplied, a slab group is re-requested: with the residual conv's group (beyond its K depth: the next
    // pass's main group); after the residual GEMM with the next pass's main groups that are still missing
    if constexpr (RES) sep2_gemm<MT, NG, (NGP < NG ? NGP : NG)>(acc, wf, xd_lane, CIN_PAD * 32, wpane, wnext);
    else sep2_gemm<MT, NG, 0>(acc, wf, xd_lane, CIN_PAD * 32, nullptr, wnext);
    STAMP2();
    // accumulator hooks and EXACT_Z (masked frames t >= lim are zeroed as packed codes in store16)
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
      if (f_exact) {
        // z == acc is a theorem for |acc| < 2^22; a wave holding an accumulator beyond +-2^21 takes the
        // float32 round trip of fixedpoint_mul (quant_utils.py:187), which is the identity below the bound
        if (sep2_any_wide<MT>(a)) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) a[mt][r] = z_roundtrip(a[mt][r], sb, !RES && f_relu);
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
      v4i pk = {(int)s02[0], (int)s02[1], (int)s13[0], (int)s13[1]};
      if (t0 + 32 * (mt + 1) > lim) pk = sep2_mask16(pk, lim - (t0 + 32 * mt + 16 * h));   // masked frames (uniform branch)
      if (co_ok) *(v4i*)((int8_t*)optr + ((size_t)b * ecout + co) * eTp + t0 + 32 * mt + 16 * h) = pk;
    };

    if constexpr (RES) {
      // res_act (jasper.py:680-682; quant_utils.py:187-214): q = clamp(rq(out) + rq(res)), then ReLU (folded into qlo)
      finish(acc, e.acc_dbg, pp.sbm);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = rq_rint(acc[mt][r], pp.Mm);
      v16i accp[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) accp[mt][r] = pp.pbias;
      sep2_gemm<MT, NGP, (NGP < NG ? NGP : NG)>(accp, wf, xr_lane, PCIN_PAD * 32, wnext, nullptr);
      STAMP2();
      finish(accp, pdbg, pp.sbp);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        int z[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = med3i(acc[mt][r] + rq_rint(accp[mt][r], pp.Mp), qlo, qhi);
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
      finish(acc, e.acc_dbg, pp.sbm);
#pragma unroll
      for (int j = 0; j < QASR_MAX_OUTS; ++j) {
        if (j < n_outs) {
          const OutP& o = e.outs[j];
          const int olo = o.lo, ohi = o.hi;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            int q[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) q[r] = rq_clamp(acc[mt][r], pp.Mo[j], olo, ohi);
            store16(o.ptr, mt, q);
          }
        }
      }
    }
    STAMP2();
  }
  }
  if (stamp) p.prof[31] = nst;
  if (tline && wg_id < p.prof_cap) {
    long long* r = p.prof + 4 * (size_t)wg_id;
    r[0] = tl_start;
    r[1] = (long long)__builtin_amdgcn_s_memrealtime();
    r[3] = (long long)__builtin_amdgcn_s_memtime() - tl_clk;     // shader cycles of the work-group: / (r[1] - r[0]) = clock / 100 MHz
    r[2] = (long long)__builtin_amdgcn_s_getreg(63492) | ((long long)__builtin_amdgcn_s_getreg(6164) << 32);   // HW_ID, XCC_ID
  }
#undef STAMP2
}

template <int K, int NG, int NGP, int NP, bool DBG, int TT, int DIL = 1>
__global__ void __launch_bounds__(SEP2_NT, SEP2_WPE) k_sep2(SepP p) {
  sep2_body<K, NG, NGP, NP, DBG, TT, DIL>(p, blockIdx.x, blockIdx.y * TT, blockIdx.x == 0 && blockIdx.y == 1,
                                          blockIdx.y * gridDim.x + blockIdx.x);
}

template <int K, int TT, int DIL = 1>
static inline size_t sep2_smem_bytes(const SepP& p) {
  using G = Sep2Geo<K, TT / DIL>;
  const size_t xd = (size_t)TT * p.cin_pad;
  if (K == 0) return xd;
  size_t un = (size_t)(SEP2_NT / 64) * G::WREG;                                 // the waves' private window + tap rows
  if (p.n_panes == 1) un = std::max(un, (size_t)TT * p.panes[0].cin_pad);       // later: the residual A image
  return xd + un;
}

// (taps, cin groups, residual cin groups) k_sep2 is instantiated for: QuartzNet's separable layers (256 / 512 channels);
// X(K, NG, NGP, NP)  (NP = passes of 256 output channels)
#define SEP2_INSTANCES(X) \
  X(33, 2, 0, 1) X(39, 2, 0, 1) X(51, 2, 0, 2) X(51, 4, 0, 2) X(63, 4, 0, 2) X(75, 4, 0, 2) \
  X(33, 2, 2, 1) X(39, 2, 2, 1) X(51, 4, 2, 2) X(51, 4, 4, 2) X(63, 4, 4, 2) X(75, 4, 4, 2) \
  X(0, 4, 0, 4)

// Shapes k_sep2 is built for; everything else stays on k_sep.
static inline bool sep2_shape_ok(const SepP& p) {
  const EpiP& e = p.e;
  if (p.K < 0 || (p.dilation != 1 && p.dilation != 2) || p.dense_k > 1 || p.cin_pad & 127 || e.cout > 1024) return false;
  if (e.flags & (QASR_F_LOGITS | QASR_F_WIDE_RQ)) return false;
  if (e.n_outs < 1) return false;
  int ngp = 0;
  if (e.flags & QASR_F_RESADD) {
    if (p.n_panes != 1 || p.panes[0].cin_pad & 127) return false;
    ngp = p.panes[0].cin_pad >> 7;
    for (int j = 0; j < e.n_outs; ++j)
      if (e.outs[j].mode != 0 && e.outs[j].mode != 2) return false;
  } else {
    if (p.n_panes != 0 || e.n_outs != 1 || e.outs[0].mode != 1) return false;   // one consumer: the pipelined epilogue
  }
  if (p.cin != p.cin_pad || (e.cout & 255) || (ngp && p.panes[0].cin != p.panes[0].cin_pad)) return false;   // exact shapes
  const int ng = p.cin_pad >> 7, np = (e.cout + 255) / 256;
  if (p.K > 0 && p.dilation == 2) return p.K == 87 && ng == 4 && ngp == 0 && np == 2;   // QuartzNet's block 16 (64- / 128-frame tiles)
#define SEP2_MATCH(K_, NG_, NGP_, NP_) if (p.K == K_ && ng == NG_ && ngp == NGP_ && np == NP_) return true;
  SEP2_INSTANCES(SEP2_MATCH)
#undef SEP2_MATCH
  return false;
}

template <int K, int NG, int NGP, int NP, bool DBG, int TT, int DIL = 1>
static int launch_sep2_v(hipStream_t s, const SepP& p) {
  const size_t smem = sep2_smem_bytes<K, TT, DIL>(p);
  if (smem > 160 * 1024 || p.e.B < 1 || p.e.Tp % TT || !p.x || !p.w || (K > 0 && !p.wdw2)) return QASR_ERR_ARG;
  static int attr_dev = -1;                                  // the attribute is per device
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (attr_dev != dev) {
    (void)hipFuncSetAttribute((const void*)k_sep2<K, NG, NGP, NP, DBG, TT, DIL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_dev = dev;
  }
  SepP q = p;
  q.prof = g_prof;
  static const int tune = getenv("QASR_SEP2_TUNE") ? atoi(getenv("QASR_SEP2_TUNE")) : 0;
  q.prof_mode = g_prof_mode | (tune << 8);
  q.prof_cap = g_prof_cap;
  hipLaunchKernelGGL((k_sep2<K, NG, NGP, NP, DBG, TT, DIL>), dim3(p.e.B, p.e.Tp / TT, 1), dim3(SEP2_NT), smem, s, q);
  return QASR_OK;
}

// all instantiations of one (tile, debug) pair; QASR_ERR_UNSUPPORTED for a shape without one
template <int TT, bool DBG>
int launch_sep2_inst(hipStream_t s, const SepP& p) {
  const int ng = p.cin_pad >> 7, ngp = (p.e.flags & QASR_F_RESADD) ? (p.panes[0].cin_pad >> 7) : 0, np = (p.e.cout + 255) / 256;
  if (p.K > 0 && p.dilation == 2) {                          // block 16: the dilation-2 form (64- / 128-frame tiles)
    if constexpr (TT >= 64) {
      if (p.K == 87 && ng == 4 && ngp == 0 && np == 2) return launch_sep2_v<87, 4, 0, 2, DBG, TT, 2>(s, p);
    }
    return QASR_ERR_UNSUPPORTED;
  }
#define SEP2_LAUNCH(K_, NG_, NGP_, NP_)                                                  \
  if constexpr (TT <= 64 || K_ > 0) {   /* the bare 1x1 form spills at 128 frames */       \
    if (p.K == K_ && ng == NG_ && ngp == NGP_ && np == NP_) return launch_sep2_v<K_, NG_, NGP_, NP_, DBG, TT>(s, p); \
  }
  SEP2_INSTANCES(SEP2_LAUNCH)
#undef SEP2_LAUNCH
  return QASR_ERR_UNSUPPORTED;
}

}  // namespace qasr
