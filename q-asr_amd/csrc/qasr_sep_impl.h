// Implementation of the fused separable-layer kernel; included by the per-instantiation translation units
// qasr_sep_t{32,64}{,_dbg}.hip (one (tile, debug) pair each, so they compile in parallel).
#pragma once
// k_sep<K>: one launch per time-channel-separable layer of the encoder (jasper.py:569-600):
//   depthwise conv (k = K, stride 1, 'same') -> QuantAct requant -> 1x1 conv (int8 MFMA GEMM over ALL output
//   channels) [-> residual 1x1 convs + res_act] -> ReLU -> the consumers' QuantAct requant -> int8 stores.
// K == 0 drops the depthwise stage (block-17 1x1 conv, decoder).
//
// Work-group = 512 threads (8 waves) = one utterance x 32 output frames x every channel, so the depthwise result
// is produced exactly once and never leaves the CU:
//   1. the input window [cin][32 + halo] is staged into LDS with coalesced 16-B loads (256 channels at a time);
//   2. a lane owns 16 consecutive frames of one channel: v_alignbyte builds the 4 byte-shifted copies of its window
//      once, v_dot4c_i32_i8 accumulates the taps (weights per lane, L1 resident); the requantised s8 result is
//      written to LDS as Xs[frame][channel] - the K-contiguous layout the MFMA A operand wants;
//   3. wave w computes output channels [32w, 32w+32) (+256 per pass) x 32 frames with v_mfma_i32_32x32x32_i8,
//      the weight fragments coming straight from L2 (each is used by exactly one wave);
//   4. epilogue in registers (per-channel parameters are per-lane scalars), LDS-staged 16-B stores.

#include <algorithm>
#include <cstdio>

#include "qasr_device.h"

namespace qasr {


// TT (template parameter): output frames per work-group, 32 or 64.  64 halves the weight and halo traffic per frame
// but makes half as many work-groups (B * Tp / 64): the engine picks it for throughput runs that keep several steps
// in flight (two launches then share the CUs), 32 for the lowest single-step latency.
#ifndef SEP_NT
#define SEP_NT 512          // threads per work-group (8 waves); 256 was measured slower at every concurrency level
#endif
#define SEP_PASS (SEP_NT / 2)  // output channels per GEMM pass (32 per wave)
#define SEP_SP 36            // staging tile row pitch in words (32 frames + 4)
#define SEP_STG_BYTES (SEP_NT / 64 * 32 * SEP_SP * 4)   // all waves' staging tiles

// DIL == 2 (block 16, k = 87): out[t] = sum_k w[k] x[t - 86 + 2k] only touches frames of t's parity, so the window is
// staged de-interleaved (one LDS row per channel and parity) and each parity stream sees an ordinary stride-1 conv with
// PAD = K/2 stream samples; a task is then (channel, parity) = 16 outputs t0 + parity + 2*o.
template <int K, int DIL, int TT>
struct SepGeo {
  static constexpr int PAD = K / 2;                         // in stream samples
  static constexpr int HALO = (PAD + 15) / 16 * 16;        // staged halo, 16-B granular (stream samples)
  static constexpr int D = HALO - PAD;                      // byte offset of the first tap inside the staged window
  static constexpr int KP4 = (K + 3) / 4;
  static constexpr int KS = 4 * KP4 + 32;                   // row pitch of the zero-margined tap array (8 B in front)
  static constexpr int WLEN = TT / DIL + 2 * HALO;      // staged bytes per LDS row
  static constexpr int WP = WLEN + 16;                      // LDS row pitch of the window
  static constexpr int NX = KP4 + 5;                        // window dwords a lane reads for 16 outputs
  static constexpr int CHUNK = 256 / DIL;                   // channels staged at a time (256 LDS rows)
  static constexpr int NPG = DIL * WLEN / 16;               // 16-B global granules per channel
};

// ---- stage an input tile [cin][32] into LDS transposed as Xs[frame][channel] (4x4 byte transposes) ------------
// `rows` frames starting at frame tf (both multiples of 4); frames outside [0, Tp) read as 0 (conv zero padding)
__device__ __forceinline__ void sep_stage_rows(unsigned char* Xs, int XP, const int8_t* __restrict__ x, int cin, int cin_pad,
                                               int Tp, int b, int tf, int rows, bool x_unsigned) {
  const unsigned flip = x_unsigned ? 0x80808080u : 0u;
  const int rq = rows >> 2;
  for (int task = threadIdx.x; task < (cin_pad / 4) * rq; task += SEP_NT) {
    const int cq = task / rq, tq = task - cq * rq;          // 4 channels x 4 frames
    const int t = tf + 4 * tq;
    const bool t_ok = t >= 0 && t < Tp;
    unsigned r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ci = 4 * cq + j;
      r[j] = (ci < cin && t_ok) ? *(const unsigned*)(x + ((size_t)b * cin + ci) * Tp + t) : 0u;
    }
    const unsigned lo01 = __builtin_amdgcn_perm(r[1], r[0], 0x05010400u), hi01 = __builtin_amdgcn_perm(r[1], r[0], 0x07030602u);
    const unsigned lo23 = __builtin_amdgcn_perm(r[3], r[2], 0x05010400u), hi23 = __builtin_amdgcn_perm(r[3], r[2], 0x07030602u);
    unsigned char* dst = Xs + (4 * tq) * XP + 4 * cq;
    *(unsigned*)(dst) = __builtin_amdgcn_perm(lo23, lo01, 0x05040100u) ^ flip;
    *(unsigned*)(dst + XP) = __builtin_amdgcn_perm(lo23, lo01, 0x07060302u) ^ flip;
    *(unsigned*)(dst + 2 * XP) = __builtin_amdgcn_perm(hi23, hi01, 0x05040100u) ^ flip;
    *(unsigned*)(dst + 3 * XP) = __builtin_amdgcn_perm(hi23, hi01, 0x07060302u) ^ flip;
  }
}
template <int TT>
__device__ __forceinline__ void sep_stage_transposed(unsigned char* Xs, int XP, const int8_t* __restrict__ x, int cin,
                                                     int cin_pad, int Tp, int b, int t0, bool x_unsigned) {
  sep_stage_rows(Xs, XP, x, cin, cin_pad, Tp, b, t0, TT, x_unsigned);
}

// ---- weight fragments of one wave: W[co_row][kc + 32*ks + 16*h .. +15], ks < SEP_WK (one 512-deep K slab) ------
// Global-load latency under load is 2.5-4k cycles here while a 32x32x32 MFMA takes 32, so the whole slab is
// requested at once (64 VGPRs): one exposed latency per GEMM instead of one per chunk.
#define SEP_WK 16
// cin_pad is a multiple of 128 (pack.py CIN_ALIGN): K steps come in unconditional groups of 4, so the compiler
// emits 4 loads / 4 LDS reads / 4 MFMAs back to back instead of a branch + wait per step.
__device__ __forceinline__ void sep_load_w(v4i (&wf)[SEP_WK], const int8_t* __restrict__ w, int cin_pad, int co_row, int kc) {
  const v4i* wp = w_frag(w, cin_pad, co_row, kc >> 5);        // consecutive K steps are 1 KiB apart
#pragma unroll
  for (int g = 0; g < SEP_WK / 4; ++g)
    if (kc + 128 * g < cin_pad) {
#pragma unroll
      for (int i = 0; i < 4; ++i) wf[4 * g + i] = wp[64 * (4 * g + i)];
    }
}
template <int MT>
__device__ __forceinline__ void sep_mfma_chunk(v16i (&acc)[MT], const v4i (&wf)[SEP_WK], const unsigned char* Xs, int XP,
                                               int cin_pad, int kc) {
  const int lane = threadIdx.x & 63, h = lane >> 5, r31 = lane & 31;
  const unsigned char* arow = Xs + r31 * XP + kc + 16 * h;
#pragma unroll
  for (int g = 0; g < SEP_WK / 4; ++g)
    if (kc + 128 * g < cin_pad) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {                       // every weight fragment feeds MT frame tiles
        v4i a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *(const v4i*)(arow + 32 * mt * XP + 32 * (4 * g + i));
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], wf[4 * g + i], acc[mt], 0, 0, 0);
      }
    }
}
// 32*MT frames x 32 channels x cin_pad GEMM for one wave; `wf` holds slab 0 already (requested by the caller ahead
// of time); deeper K (cin > 512) streams further slabs.
template <int MT>
__device__ __forceinline__ void sep_gemm(v16i (&acc)[MT], v4i (&wf)[SEP_WK], const unsigned char* Xs, int XP,
                                         const int8_t* __restrict__ w, int cin_pad, int co_row) {
  for (int kc = 0; kc < cin_pad; kc += 32 * SEP_WK) {
    if (kc) sep_load_w(wf, w, cin_pad, co_row, kc);
    sep_mfma_chunk(acc, wf, Xs, XP, cin_pad, kc);
  }
}

// Dense k > 1 conv (Jasper, jasper.py:601-630) on the same tile: out[t] = sum_k W_k x[t + k*dil - pad] is a sum of taps 1x1
// GEMMs whose A rows are the staged window shifted by k*dil frames.  Weights are tap-major (pack.py: one fragment-
// ordered [cout_pad][cin_pad] matrix per tap).  Work items = (tap, 256-deep half slab); two register buffers so that
// the fragments of item i+1 travel while item i is on the matrix cores.
#define SEP_DK 8                                             /* K steps (of 32) per item */
__device__ __forceinline__ void sep_load_wd(v4i (&wf)[SEP_DK], const int8_t* __restrict__ w, int cin_pad, int co_row, int kc) {
  const v4i* wp = w_frag(w, cin_pad, co_row, kc >> 5);
#pragma unroll
  for (int g = 0; g < SEP_DK / 4; ++g)
    if (kc + 128 * g < cin_pad) {
#pragma unroll
      for (int i = 0; i < 4; ++i) wf[4 * g + i] = wp[64 * (4 * g + i)];
    }
}
template <int MT>
__device__ __forceinline__ void sep_mfma_d(v16i (&acc)[MT], const v4i (&wf)[SEP_DK], const unsigned char* Xrow, int XP, int cin_pad,
                                           int kc) {
  const int lane = threadIdx.x & 63, h = lane >> 5, r31 = lane & 31;
  const unsigned char* arow = Xrow + r31 * XP + kc + 16 * h;
  constexpr int AB = MT > 2 ? 2 : 4;                         // K steps whose A fragments are read ahead (register budget)
  v4i a[MT][AB];
#pragma unroll
  for (int g = 0; g < SEP_DK / AB; ++g)
    if (kc + 32 * AB * g < cin_pad) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < AB; ++i) a[mt][i] = *(const v4i*)(arow + 32 * mt * XP + 32 * (AB * g + i));
      // all reads of the group are in flight before the first MFMA: left alone, the compiler sinks every ds_read next
      // to its MFMA and waits for it, which makes the tap loop LDS-latency bound (Jasper: 10.4 -> 9.9 ms)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < AB; ++i)                            // every weight fragment feeds MT frame tiles
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[mt][i], wf[AB * g + i], acc[mt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
}
template <int MT>
__device__ __forceinline__ void sep_gemm_taps(v16i (&acc)[MT], const unsigned char* Xs0, int XP, const int8_t* __restrict__ w,
                                              int cin_pad, int cout_pad, int co_row, int taps, int dil) {
  const int nh = (cin_pad + 32 * SEP_DK - 1) / (32 * SEP_DK);       // half slabs per tap
  const int total = taps * nh;
  const size_t tap_bytes = (size_t)cout_pad * cin_pad;
  v4i wa[SEP_DK], wb[SEP_DK];
  auto load = [&](v4i (&wf)[SEP_DK], int i) {
    const int tap = i / nh, hs = i - tap * nh;
    sep_load_wd(wf, w + tap * tap_bytes, cin_pad, co_row, 32 * SEP_DK * hs);
  };
  auto mma = [&](const v4i (&wf)[SEP_DK], int i) {
    const int tap = i / nh, hs = i - tap * nh;
    sep_mfma_d<MT>(acc, wf, Xs0 + tap * dil * XP, XP, cin_pad, 32 * SEP_DK * hs);
  };
  load(wa, 0);
  for (int i = 0; i < total; i += 2) {
    if (i + 1 < total) load(wb, i + 1);
    mma(wa, i);
    if (i + 2 < total) load(wa, i + 2);
    if (i + 1 < total) mma(wb, i + 1);
  }
}

__device__ __forceinline__ void sep_dump(int32_t* dbg, const v16i& a, int b, int co, int cout, int t0, int h, int T, int Tp) {
  if (!dbg || co >= cout) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int t = t0 + mfma32_row(r, h);
    if (t < T) dbg[((size_t)b * cout + co) * Tp + t] = a[r];
  }
}

// per-lane (= per output channel, MFMA C layout: channel = lane & 31) parameters, fetched ahead of the GEMM
struct SepLaneP {
  int bias;
  float sb;
  double m_main;
};
template <int EP>
__device__ __forceinline__ SepLaneP sep_lane_params(const SepP& p, int cor) {
  const EpiP& e = p.e;
  SepLaneP q;
  q.bias = p.bias[cor];
  q.sb = (EP == 0 || (e.flags & QASR_F_EXACT_Z)) ? e.sb[cor] : 1.0f;
  q.m_main = (EP == 2 || (EP == 0 && (e.flags & QASR_F_RESADD))) ? e.m_main[cor] : 0.0;
  return q;
}

// wave-level LDS hand-over: DS operations of one wave execute in issue order, the fences only pin the compiler
__device__ __forceinline__ void sep_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// EP selects the epilogue the kernel is specialised for (dead paths cost SGPRs, scalar reloads of the kernarg
// block and instruction-cache space):
//   EP_GENERIC  everything (logits, raw int32 outs, EXACT_Z, any number of residual panes)  - Jasper, decoder
//   EP_PLAIN    ReLU? + per-channel requant to <= 3 consumers                                - mid-block layers
//   EP_RESADD1  one residual 1x1 conv + res_act + ReLU + scalar / identity requant           - QuartzNet block ends
// DBG adds the int32 accumulator dumps of the parity hooks.
enum { EP_GENERIC = 0, EP_PLAIN = 1, EP_RESADD1 = 2 };

template <int K, int DIL, int EP, bool DBG, int TT>
__global__ void __launch_bounds__(SEP_NT) k_sep(SepP p) {
  using G = SepGeo<(K > 0 ? K : 1), DIL, TT>;
  constexpr int SEP_MT = TT / 32;                            // 32-frame MFMA tiles per wave
  constexpr int SEP_TPC = TT / 16;                           // depthwise tasks (16 outputs) per channel
  constexpr bool GEN = EP == EP_GENERIC;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const EpiP& e = p.e;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  // XCD-aware mapping: work-groups are dealt round-robin over the 8 XCDs by linear id, so the utterance index is the
  // fastest grid dimension - all time tiles of utterance b (whose depthwise windows overlap) and the layer that
  // produced them run on XCD b % 8 and find their halos / inputs in that XCD's L2.
  const int b = blockIdx.x, t0 = blockIdx.y * TT;
  const int XP = p.cin_pad + 16;
  // dense k > 1 conv (K == 0 instantiations only): taps, 'same' padding (taps-1)*dil/2, window halo rounded up to 4 frames
  const bool dense = K == 0 && p.dense_k > 1;
  const int dpad = dense ? (p.dense_k - 1) * p.dilation / 2 : 0, dhalo = (dpad + 3) & ~3;
  unsigned char* Xs = smem;                                  // [TT (+ 2 dhalo)][XP]   A operand of the main GEMM
  unsigned char* Xr = Xs + (TT + 2 * dhalo) * XP;            // [TT][XPr]  A operand of the residual GEMMs
  int xr_bytes = 0;
  if (EP != EP_PLAIN)
    for (int k = 0; k < p.n_panes; ++k) xr_bytes = max(xr_bytes, TT * (p.panes[k].cin_pad + 16));
  unsigned char* Ws = Xr + xr_bytes;                         // [256][WP]  depthwise window, then the waves' staging tiles
  int* const stg = (int*)Ws + wave * (32 * SEP_SP);          // this wave's [32 channels][32 frames] int32 tile

  // hot scalars are copied out of the (large, spilled) kernarg block once: re-reading them inside the unrolled
  // epilogue loops costs an s_load + wait per use
  const unsigned flags = e.flags;
  const int eT = e.T, eTp = e.Tp, ecout = e.cout, qlo = e.qlo, qhi = e.qhi, n_outs = e.n_outs;
  const int len_b = e.lens[b];
  const int lim = (flags & QASR_F_MASK_OUT) ? min(len_b, eT) : eT;
  const int dlim = min(len_b, eT);                           // the 1x1 conv's MaskedConv1d masks its input
  const bool f_relu = flags & QASR_F_RELU;
  const bool f_resadd = GEN ? bool(flags & QASR_F_RESADD) : (EP == EP_RESADD1);
  const bool f_logits = GEN && (flags & QASR_F_LOGITS);
  const bool f_exact = (flags & QASR_F_EXACT_Z) && !f_logits;   // logits are fl32(acc) * s_b, no QuantAct follows
  const int n_panes = GEN ? p.n_panes : (EP == EP_RESADD1 ? 1 : 0);
  const int dw_lo = p.dw_lo, dw_hi = p.dw_hi;
  const bool stamp = p.prof && blockIdx.x == 0 && blockIdx.y == 1 && tid == 0;
  int nst = 0;
#define STAMP() do { if (stamp && nst < 31) p.prof[nst++] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
  STAMP();

  // ---- prefetch for GEMM pass 0: per-lane parameters and the first K chunk of this wave's weight rows.
  // They are consumed after the depthwise stage, whose VALU work hides their L2 / Infinity-Cache latency.
  const int cout_pad = (ecout + 127) / 128 * 128;
  const int co_l = 32 * wave + (lane & 31);                  // row inside a 256-channel pass
  const bool in0 = co_l < cout_pad;
  SepLaneP lp = sep_lane_params<EP>(p, in0 ? co_l : 0);
  v4i wf[SEP_WK];
  if (!dense) sep_load_w(wf, p.w, p.cin_pad, in0 ? co_l : 0, 0);        // consumed after the depthwise stage, which hides it
  __builtin_amdgcn_sched_barrier(0);                         // keep the requests up here (the scheduler sinks them to first use)

  if (K > 0) {
    // ------------------------------------------------------------------ depthwise stage
    const unsigned flip = p.x_unsigned ? 0x80808080u : 0u;
    constexpr int NP = G::NPG;                               // 16-B global granules per channel
    constexpr int CH = G::CHUNK;                             // channels per staged chunk
    constexpr int NPT = (CH * NP + SEP_NT - 1) / SEP_NT;     // granules per thread per chunk
    v4i pc[NPT];
    auto fetch = [&](int c0) {                               // global -> registers (coalesced 16-B granules)
      const int nch = min(CH, p.cin - c0);
#pragma unroll
      for (int i = 0; i < NPT; ++i) {
        const int pi = tid + SEP_NT * i;
        const int row = pi / NP, col = pi - row * NP;
        const int t = t0 - DIL * G::HALO + 16 * col;         // granule entirely inside or outside [0, Tp)
        pc[i] = (v4i){0, 0, 0, 0};
        if (row < nch && t >= 0 && t < eTp) pc[i] = *(const v4i*)(p.x + ((size_t)b * p.cin + c0 + row) * eTp + t);
      }
    };
    auto commit = [&](int c0) {                              // registers -> LDS window
      const int nch = min(CH, p.cin - c0);
#pragma unroll
      for (int i = 0; i < NPT; ++i) {
        const int pi = tid + SEP_NT * i;
        const int row = pi / NP, col = pi - row * NP;
        if (row < nch) {
          v4i v = pc[i];
          v[0] ^= flip; v[1] ^= flip; v[2] ^= flip; v[3] ^= flip;
          if (DIL == 1) {
            *(v4i*)(Ws + row * G::WP + 16 * col) = v;
          } else {                                           // de-interleave: even frames -> row 2c, odd -> row 2c+1
            const unsigned e0 = __builtin_amdgcn_perm(v[1], v[0], 0x06040200u), e1 = __builtin_amdgcn_perm(v[3], v[2], 0x06040200u);
            const unsigned o0 = __builtin_amdgcn_perm(v[1], v[0], 0x07050301u), o1 = __builtin_amdgcn_perm(v[3], v[2], 0x07050301u);
            *(uint2*)(Ws + (2 * row) * G::WP + 8 * col) = make_uint2(e0, e1);
            *(uint2*)(Ws + (2 * row + 1) * G::WP + 8 * col) = make_uint2(o0, o1);
          }
        }
      }
    };
    fetch(0);
    for (int c0 = 0; c0 < p.cin; c0 += CH) {
      const int nch = min(CH, p.cin - c0);
      if constexpr (DIL == 1) {
        // ---- depthwise taps on the matrix cores: v_mfma_i32_4x4x4_16B_i8 = 16 independent 4x4x4 products, block =
        // channel.  With A[i][k] = win[f0 + D + m0 + 4i + k] and B[k][j] = w[m0 + k - j] (0 outside the taps) the block
        // accumulates out[f0 + 4i + j] over the 4 taps m0-j .. m0-j+3; m0 advances by 4 per instruction.  Lane l of a
        // group: channel l >> 2, A row i = l & 3 (one aligned LDS dword), B column j = l & 3 (one dword of the lane's
        // own pre-shifted tap stream, read unaligned from the zero-margined tap array), D register v = row i.
        // Wave w owns rows [32w, 32w + 32) of the chunk: 2 groups of 16 channels x TT/16 frame tiles.
        constexpr int MS = -(G::D & 3);                      // first m0: keeps the A dwords 4-byte aligned
        constexpr int NS = (K + 3 - MS + 3) / 4;             // instructions per 16-frame tile
        constexpr int NZ = 4 * (TT / 16);                    // outputs per lane and group
        const int cb = lane >> 2, jl = lane & 3;
        // taps of both groups of this wave: requested before the barrier they do not depend on
        // the lane's tap stream starts at byte 8 + MS - jl of its channel's row: whole dwords are fetched with wide
        // loads (rows are 4-byte aligned) and funnel-shifted into place once per chunk
        constexpr int NR = (NS + 1 + 3) / 4 * 4;
        const int e0 = 8 + MS - jl, tq = e0 >> 2, tsh = e0 & 3;
        v4i raw[2][NR / 4];
        int biasg[2];
        double Mg[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int row = 32 * wave + 16 * g + cb;
          const int c = c0 + min(row, nch - 1);
          const v4i* tp = (const v4i*)((const unsigned char*)p.wdw2 + (size_t)c * G::KS + 4 * tq);
#pragma unroll
          for (int i = 0; i < NR / 4; ++i) raw[g][i] = tp[i];
          biasg[g] = p.bias_dw[c];
          Mg[g] = p.m_dw[c];
        }
        if (c0) __syncthreads();                             // previous chunk's window fully consumed
        commit(c0);
        __syncthreads();
        if (c0 + CH < p.cin) fetch(c0 + CH);                 // next chunk's window travels during this chunk's math
        STAMP();
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int row = 32 * wave + 16 * g + cb;
          if (32 * wave + 16 * g >= nch) break;              // wave-uniform: group entirely beyond the chunk
          const bool row_ok = row < nch;
          const int c = c0 + row;
          unsigned tw[NS];
#pragma unroll
          for (int st = 0; st < NS; ++st)
            tw[st] = __builtin_amdgcn_alignbyte((unsigned)raw[g][(st + 1) >> 2][(st + 1) & 3], (unsigned)raw[g][st >> 2][st & 3], tsh);
          int z[NZ];
          {
            // TT/16 frame tiles = independent accumulation chains, interleaved so that no MFMA waits for its predecessor
            v4i acc[TT / 16];
#pragma unroll
            for (int ft = 0; ft < TT / 16; ++ft) acc[ft] = (v4i){biasg[g], biasg[g], biasg[g], biasg[g]};
            const unsigned* ap = (const unsigned*)(Ws + min(row, nch - 1) * G::WP + (G::D + MS) + 4 * jl);
#pragma unroll
            for (int st = 0; st < NS; ++st) {
#pragma unroll
              for (int ft = 0; ft < TT / 16; ++ft)
                acc[ft] = __builtin_amdgcn_mfma_i32_4x4x4i8((int)ap[4 * ft + st], (int)tw[st], acc[ft], 0, 0, 0);
            }
#pragma unroll
            for (int ft = 0; ft < TT / 16; ++ft)
#pragma unroll
              for (int v = 0; v < 4; ++v) z[4 * ft + v] = acc[ft][v];
          }
          // z[4 ft + v] is local frame 16 ft + 4 v + jl of channel c
          if (DBG && p.dw_acc_dbg && row_ok) {
#pragma unroll
            for (int o = 0; o < NZ; ++o) {
              const int f = 16 * (o >> 2) + 4 * (o & 3) + jl;
              if (t0 + f < eT) p.dw_acc_dbg[((size_t)b * p.cin + c) * eTp + t0 + f] = z[o];
            }
          }
          int qv[NZ];
          requant_batch<NZ>(qv, z, Mg[g], dw_lo, dw_hi);
          if (row_ok) {
#pragma unroll
            for (int o = 0; o < NZ; ++o) {
              const int f = 16 * (o >> 2) + 4 * (o & 3) + jl;
              Xs[f * XP + c] = (unsigned char)((t0 + f < dlim) ? qv[o] : 0);
            }
          }
        }
        STAMP();
        continue;
      }
      // taps / parameters of this thread's first task of the chunk: issued before the barrier they do not depend on
      const int task0 = tid;
      const bool has0 = task0 < nch * SEP_TPC;
      const int c_first = c0 + (has0 ? (task0 / SEP_TPC) : 0);
      int wv[G::KP4];
      {
        const int* wk = (const int*)(p.wdw + (size_t)c_first * (G::KP4 * 4));
#pragma unroll
        for (int m = 0; m < G::KP4; ++m) wv[m] = wk[m];
      }
      int bias = p.bias_dw[c_first];
      double M = p.m_dw[c_first];
      if (c0) __syncthreads();                               // previous chunk's window fully consumed
      commit(c0);
      __syncthreads();
      if (c0 + CH < p.cin) fetch(c0 + CH);                   // next chunk's window travels during this chunk's math
      STAMP();
      for (int task = tid; task < nch * SEP_TPC; task += SEP_NT) {
        const int row = task / SEP_TPC, hq = task % SEP_TPC;   // DIL 1: 16 consecutive frames; DIL 2: (segment, parity)
        const int c = c0 + row;
        if (task != task0) {
          const int* wk = (const int*)(p.wdw + (size_t)c * (G::KP4 * 4));
#pragma unroll
          for (int m = 0; m < G::KP4; ++m) wv[m] = wk[m];
          bias = p.bias_dw[c];
          M = p.m_dw[c];
        }
        const unsigned* wrow = (DIL == 1) ? (const unsigned*)(Ws + row * G::WP) + 4 * hq + (G::D >> 2)
                                          : (const unsigned*)(Ws + (2 * row + (hq & 1)) * G::WP) + 4 * (hq >> 1) + (G::D >> 2);
        unsigned xw[G::NX];
#pragma unroll
        for (int i = 0; i < G::NX; ++i) xw[i] = wrow[i];
        int acc[16];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          constexpr int d = G::D & 3;
          const int sh = (s + d) & 3, off = (s + d) >> 2;
          unsigned A[G::KP4 + 3];
#pragma unroll
          for (int i = 0; i < G::KP4 + 3; ++i)
            A[i] = sh ? __builtin_amdgcn_alignbyte(xw[off + i + 1], xw[off + i], sh) : xw[off + i];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            int a = bias;
#pragma unroll
            for (int m = 0; m < G::KP4; ++m) a = __builtin_amdgcn_sdot4((int)A[j + m], wv[m], a, false);
            acc[4 * j + s] = a;
          }
        }
        // output o of the task is local frame fl(o)
        auto fl = [&](int o) { return DIL == 1 ? 16 * hq + o : (hq & 1) + 2 * (16 * (hq >> 1) + o); };
        if (DBG && p.dw_acc_dbg) {
#pragma unroll
          for (int o = 0; o < 16; ++o)
            if (t0 + fl(o) < eT) p.dw_acc_dbg[((size_t)b * p.cin + c) * eTp + t0 + fl(o)] = acc[o];
        }
        int qv[16];
        requant_batch<16>(qv, acc, M, dw_lo, dw_hi);
#pragma unroll
        for (int o = 0; o < 16; ++o) Xs[fl(o) * XP + c] = (unsigned char)((t0 + fl(o) < dlim) ? qv[o] : 0);
      }
      STAMP();
    }
    // zero the channel padding of Xs (cin..cin_pad) so padded K columns multiply weight 0 by a finite byte
    for (int i = tid; i < TT * (p.cin_pad - p.cin); i += SEP_NT) {
      const int r = i / (p.cin_pad - p.cin), cc = i - r * (p.cin_pad - p.cin);
      Xs[r * XP + p.cin + cc] = 0;
    }
  } else if (dense) {
    // dense conv: window of TT + 2*dhalo frames, every input channel; A rows of tap k start at row dhalo - pad + k*dil
    sep_stage_rows(Xs, XP, p.x, p.cin, p.cin_pad, e.Tp, b, t0 - dhalo, TT + 2 * dhalo, p.pw_unsigned);
  } else {
    sep_stage_transposed<TT>(Xs, XP, p.x, p.cin, p.cin_pad, e.Tp, b, t0, p.pw_unsigned);
  }
  // a single residual pane is staged here, once, next to the main operand (its loads overlap the barrier wait)
  if (f_resadd && n_panes == 1)
    sep_stage_transposed<TT>(Xr, p.panes[0].cin_pad + 16, p.panes[0].x, p.panes[0].cin, p.panes[0].cin_pad, e.Tp, b, t0,
                         p.panes[0].x_unsigned);
  __syncthreads();
  STAMP();

  // ---------------------------------------------------------------------- pointwise GEMM passes of 256 channels
  for (int cbase = 0; cbase < cout_pad; cbase += SEP_PASS) {
    const int co = cbase + co_l;
    const bool co_in = co < cout_pad;                        // wave-uniform (cout_pad is a multiple of 128)
    const int cor = co_in ? co : 0;
    const bool co_ok = co < ecout;
    const SepLaneP cur = lp;
    v16i acc[SEP_MT];
#pragma unroll
    for (int mt = 0; mt < SEP_MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][r] = cur.bias;
    if (co_in) {
      if (dense) sep_gemm_taps<SEP_MT>(acc, Xs + (dhalo - dpad) * XP, XP, p.w, p.cin_pad, cout_pad, cor, p.dense_k, p.dilation);
      else sep_gemm(acc, wf, Xs, XP, p.w, p.cin_pad, cor);
    }
    // prefetch what the NEXT GEMM of this wave needs while the epilogue below runs
    const bool more = cbase + SEP_PASS < cout_pad;
    const int con = more ? ((cbase + SEP_PASS + co_l < cout_pad) ? cbase + SEP_PASS + co_l : 0) : 0;
    SepLaneP nxt = cur;
    if (f_resadd && n_panes > 0) sep_load_w(wf, p.panes[0].w, p.panes[0].cin_pad, cor, 0);
    if (more) {
      nxt = sep_lane_params<EP>(p, con);
      if (!f_resadd && !dense) sep_load_w(wf, p.w, p.cin_pad, con, 0);
    }
    STAMP();
    if (DBG) {
#pragma unroll
      for (int mt = 0; mt < SEP_MT; ++mt) sep_dump(e.acc_dbg, acc[mt], b, co, ecout, t0 + 32 * mt, h, eT, eTp);
    }

    // ---- row-layout epilogue of one 32-frame tile: the int32 results cross from the MFMA C layout (lane = channel
    // lane & 31, frames {8g + 4h .. +3}) to lane l = (channel l >> 1, frames 16 (l & 1) .. +15) through this wave's
    // private LDS staging tile (wave-level synchronisation only), then one 16-byte store per lane and consumer.
    const int row = lane >> 1, half = lane & 1;
    const int co2 = cbase + 32 * wave + row;                   // this lane's channel in the row layout
    const bool co2_ok = co_in && co2 < ecout;
    const int cor2 = co2_ok ? co2 : 0;
    auto emit = [&](int mt, const int (&z)[16]) {
      {
        int* wr = stg + (lane & 31) * SEP_SP + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) *(v4i*)(wr + 8 * g) = (v4i){z[4 * g], z[4 * g + 1], z[4 * g + 2], z[4 * g + 3]};
      }
      sep_wave_sync();
      int zr[16];
      {
        const int* rd = stg + row * SEP_SP + 16 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const v4i v = *(const v4i*)(rd + 4 * g);
          zr[4 * g] = v[0]; zr[4 * g + 1] = v[1]; zr[4 * g + 2] = v[2]; zr[4 * g + 3] = v[3];
        }
      }
      sep_wave_sync();                                         // staging tile free for the next tile / pass
      const int tl0 = t0 + 32 * mt + 16 * half;                // first frame of this lane's 16
      if (f_logits) {                                          // decoder: logits[b][t][co] = fl32(fl32(acc) * s_b[co])
        if (co2_ok) {
          const float sb2 = e.sb[cor2];
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (tl0 + i < eT) e.logits[((size_t)b * eT + tl0 + i) * ecout + co2] = __fmul_rn((float)zr[i], sb2);
        }
        return;
      }
#pragma unroll
      for (int j = 0; j < QASR_MAX_OUTS; ++j) {
        if (j >= n_outs) break;
        const int omode = e.outs[j].mode, olo = e.outs[j].lo, ohi = e.outs[j].hi;
        void* const optr = e.outs[j].ptr;
        if (GEN && omode == 3) {                             // raw int32 (many-consumer values), rare path
          if (co2_ok) {
            int* op = (int*)optr + ((size_t)b * ecout + co2) * eTp + tl0;
#pragma unroll
            for (int i = 0; i < 16; ++i) op[i] = (tl0 + i < lim) ? zr[i] : 0;
          }
          continue;
        }
        int qo[16];
        if (EP != EP_PLAIN && omode == 2) {
#pragma unroll
          for (int i = 0; i < 16; ++i) qo[i] = zr[i];
        } else {
          const double Mc = (EP == EP_PLAIN || omode == 1) ? e.outs[j].mtab[cor2] : e.outs[j].m;
          requant_batch<16>(qo, zr, Mc, olo, ohi);
        }
        v4i pk;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          int v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = (tl0 + 4 * g + i < lim) ? qo[4 * g + i] : 0;
          pk[g] = (int)pack4(v[0], v[1], v[2], v[3]);
        }
        if (co2_ok) *(v4i*)((int8_t*)optr + ((size_t)b * ecout + co2) * eTp + tl0) = pk;
      }
    };
    auto any_big = [&](const v16i& a) {
      bool big = false;
#pragma unroll
      for (int r = 0; r < 16; ++r) big |= (a[r] >= (1 << 22)) | (a[r] <= -(1 << 22));
      return __any(big) != 0;
    };

    if (f_resadd && n_panes == 1) {
      // res_act (jasper.py:680-682; quant_utils.py:187-214): q = clamp(rq(out) + rq(res)), then ReLU
      const PaneP& pn = p.panes[0];
      const int XPr = pn.cin_pad + 16;
      const int bv = pn.bias[cor];
      const double Mp = pn.m[cor];
      const float sbp = pn.sb[cor];
      v16i accp[SEP_MT];
#pragma unroll
      for (int mt = 0; mt < SEP_MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) accp[mt][r] = bv;
      if (co_in) sep_gemm(accp, wf, Xr, XPr, pn.w, pn.cin_pad, cor);
      if (more && !dense) sep_load_w(wf, p.w, p.cin_pad, con, 0);
#pragma unroll
      for (int mt = 0; mt < SEP_MT; ++mt) {
        if (DBG) sep_dump(pn.acc_dbg, accp[mt], b, co, ecout, t0 + 32 * mt, h, eT, eTp);
        const bool exact = f_exact && any_big(acc[mt]), exactp = f_exact && any_big(accp[mt]);
        int z[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const double s = requant_d(exact ? z_roundtrip(acc[mt][r], cur.sb, false) : acc[mt][r], cur.m_main) +
                           requant_d(exactp ? z_roundtrip(accp[mt][r], sbp, false) : accp[mt][r], Mp);
          const int q = (int)fmin(fmax(s, (double)qlo), (double)qhi);
          z[r] = f_relu ? max(q, 0) : q;
        }
        emit(mt, z);
      }
    } else if (GEN && f_resadd && !(flags & QASR_F_WIDE_RQ)) {
      // dense residual (Jasper): pane after pane, clamp after every add; panes share the Xr staging buffer.  Every
      // |acc * M| of the op stays below 2^30 (the packer would have set QASR_F_WIDE_RQ): each operand's rounded product
      // comes out of the low mantissa word (rq_rint), the running sum is an integer, the clamp one v_med3_i32 - four
      // instructions per value and pane instead of six fp64 ones, and 32 registers less than the double-domain form
      // below (with 7-10 panes the epilogue, not the panes' GEMMs, is what these layers spend their time in)
      int d[SEP_MT][16];
#pragma unroll
      for (int mt = 0; mt < SEP_MT; ++mt) {
        const bool exact = f_exact && any_big(acc[mt]);
#pragma unroll
        for (int r = 0; r < 16; ++r) d[mt][r] = rq_rint(exact ? z_roundtrip(acc[mt][r], cur.sb, false) : acc[mt][r], cur.m_main);
      }
      for (int pi = 0; pi < n_panes; ++pi) {
        const PaneP& pn = p.panes[pi];
        const int XPr = pn.cin_pad + 16;
        const int bv = pn.bias[cor];
        const double Mp = pn.m[cor];
        const float sbp = pn.sb[cor];
        if (pi > 0) sep_load_w(wf, pn.w, pn.cin_pad, cor, 0);
        __syncthreads();
        sep_stage_transposed<TT>(Xr, XPr, pn.x, pn.cin, pn.cin_pad, e.Tp, b, t0, pn.x_unsigned);
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < SEP_MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][r] = bv;
        if (co_in) sep_gemm(acc, wf, Xr, XPr, pn.w, pn.cin_pad, cor);
#pragma unroll
        for (int mt = 0; mt < SEP_MT; ++mt) {
          if (DBG) sep_dump(pn.acc_dbg, acc[mt], b, co, ecout, t0 + 32 * mt, h, eT, eTp);
          const bool exactp = f_exact && any_big(acc[mt]);
#pragma unroll
          for (int r = 0; r < 16; ++r)
            d[mt][r] = med3i(d[mt][r] + rq_rint(exactp ? z_roundtrip(acc[mt][r], sbp, false) : acc[mt][r], Mp), qlo, qhi);
        }
      }
      if (more && !dense) sep_load_w(wf, p.w, p.cin_pad, con, 0);
#pragma unroll
      for (int mt = 0; mt < SEP_MT; ++mt) {
        int z[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = f_relu ? max(d[mt][r], 0) : d[mt][r];
        emit(mt, z);
      }
    } else if (GEN && f_resadd) {
      // dense residual (Jasper): pane after pane, clamp after every add; panes share the Xr staging buffer
      double d[SEP_MT][16];
#pragma unroll
      for (int mt = 0; mt < SEP_MT; ++mt) {
        const bool exact = f_exact && any_big(acc[mt]);
#pragma unroll
        for (int r = 0; r < 16; ++r) d[mt][r] = requant_d(exact ? z_roundtrip(acc[mt][r], cur.sb, false) : acc[mt][r], cur.m_main);
      }
      for (int pi = 0; pi < n_panes; ++pi) {
        const PaneP& pn = p.panes[pi];
        const int XPr = pn.cin_pad + 16;
        const int bv = pn.bias[cor];
        const double Mp = pn.m[cor];
        const float sbp = pn.sb[cor];
        if (pi > 0) sep_load_w(wf, pn.w, pn.cin_pad, cor, 0);
        __syncthreads();
        sep_stage_transposed<TT>(Xr, XPr, pn.x, pn.cin, pn.cin_pad, e.Tp, b, t0, pn.x_unsigned);
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < SEP_MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][r] = bv;
        if (co_in) sep_gemm(acc, wf, Xr, XPr, pn.w, pn.cin_pad, cor);
#pragma unroll
        for (int mt = 0; mt < SEP_MT; ++mt) {
          if (DBG) sep_dump(pn.acc_dbg, acc[mt], b, co, ecout, t0 + 32 * mt, h, eT, eTp);
          const bool exactp = f_exact && any_big(acc[mt]);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const double s = d[mt][r] + requant_d(exactp ? z_roundtrip(acc[mt][r], sbp, false) : acc[mt][r], Mp);
            d[mt][r] = fmin(fmax(s, (double)qlo), (double)qhi);
          }
        }
      }
      if (more && !dense) sep_load_w(wf, p.w, p.cin_pad, con, 0);
#pragma unroll
      for (int mt = 0; mt < SEP_MT; ++mt) {
        int z[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int q = (int)d[mt][r];
          z[r] = f_relu ? max(q, 0) : q;
        }
        emit(mt, z);
      }
    } else {
      // z == acc is a theorem for |acc| < 2^22; only a wave holding a larger accumulator takes the
      // float32 round trip of fixedpoint_mul (the packer flags layers whose bound allows that)
#pragma unroll
      for (int mt = 0; mt < SEP_MT; ++mt) {
        int z[16];
        if (f_exact && any_big(acc[mt])) {
#pragma unroll
          for (int r = 0; r < 16; ++r) z[r] = z_roundtrip(acc[mt][r], cur.sb, f_relu);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) z[r] = f_relu ? max(acc[mt][r], 0) : acc[mt][r];
        }
        emit(mt, z);
      }
    }
    STAMP();
    lp = nxt;
  }
  if (stamp) p.prof[31] = nst;
}

static inline size_t sep_smem_bytes(const SepP& p, int WP, int TT) {
  const int dpad = p.dense_k > 1 ? (p.dense_k - 1) * p.dilation / 2 : 0, dhalo = (dpad + 3) & ~3;
  size_t xs = (size_t)(TT + 2 * dhalo) * (p.cin_pad + 16);
  size_t xr = 0;
  for (int k = 0; k < p.n_panes; ++k) xr = std::max(xr, (size_t)TT * (p.panes[k].cin_pad + 16));
  size_t ws = std::max((size_t)256 * WP, (size_t)SEP_STG_BYTES);  // 256 window rows in either dilation mode
  return xs + xr + ws;
}

template <int K, int DIL, int EP, bool DBG, int TT>
static int launch_sep_v(hipStream_t s, const SepP& p) {
  using G = SepGeo<(K > 0 ? K : 1), DIL, TT>;
  const size_t smem = sep_smem_bytes(p, K > 0 ? G::WP : 0, TT);
  // launch shape checked on the host: a work-group's LDS image, the grid and every pointer the kernel dereferences
  // unconditionally (a kernel that faults can take the whole GPU down; see DESIGN.md §4, "bring-up abort")
  if (smem > 160 * 1024 || p.e.B < 1 || p.e.Tp < TT || p.e.Tp % TT || p.e.T > p.e.Tp || !p.x || !p.w || !p.bias || !p.e.lens ||
      p.cin < 1 || p.cin > p.cin_pad || p.cin_pad % 128 || p.e.cout < 1 || p.n_panes < 0 || p.n_panes > QASR_MAX_PANES ||
      (K > 0 && (!p.wdw || !p.wdw2 || !p.bias_dw || !p.m_dw)))
    return QASR_ERR_ARG;
  for (int k = 0; k < p.n_panes; ++k)
    if (!p.panes[k].x || !p.panes[k].w || !p.panes[k].bias || !p.panes[k].m || p.panes[k].cin_pad % 128) return QASR_ERR_ARG;
  static int attr_dev = -1;                                  // the attribute is per device, not per process
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (attr_dev != dev) {
    (void)hipFuncSetAttribute((const void*)k_sep<K, DIL, EP, DBG, TT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_dev = dev;
  }
  dim3 g(p.e.B, p.e.Tp / TT, 1);
  SepP q = p;
  q.prof = g_prof_mode == 0 ? g_prof : nullptr;
  hipLaunchKernelGGL((k_sep<K, DIL, EP, DBG, TT>), g, dim3(SEP_NT), smem, s, q);
  return QASR_OK;
}

// which specialised epilogue covers this op
static inline int sep_epilogue_class(const SepP& p) {
  const EpiP& e = p.e;
  if (e.flags & QASR_F_LOGITS) return EP_GENERIC;
  for (int j = 0; j < e.n_outs; ++j)
    if (e.outs[j].mode == 3) return EP_GENERIC;
  if (e.flags & QASR_F_RESADD) {
    if (p.n_panes != 1) return EP_GENERIC;
    for (int j = 0; j < e.n_outs; ++j)
      if (e.outs[j].mode == 1) return EP_GENERIC;
    return EP_RESADD1;
  }
  for (int j = 0; j < e.n_outs; ++j)
    if (e.outs[j].mode != 1) return EP_GENERIC;
  return EP_PLAIN;
}

template <int K, int DIL, bool DBG, int TT>
static int launch_sep_k(hipStream_t s, const SepP& p) {
  const int ep = sep_epilogue_class(p);
  if (ep == EP_PLAIN) return launch_sep_v<K, DIL, EP_PLAIN, DBG, TT>(s, p);
  if (ep == EP_RESADD1) return launch_sep_v<K, DIL, EP_RESADD1, DBG, TT>(s, p);
  return launch_sep_v<K, DIL, EP_GENERIC, DBG, TT>(s, p);
}

// all kernel-size instantiations of one (tile, debug) pair
// QASR_ERR_UNSUPPORTED for a tap count / dilation without an instantiation (nothing is launched), QASR_ERR_ARG for a
// launch shape the kernel must not see
template <int TT, bool DBG>
int launch_sep_inst(hipStream_t s, const SepP& p) {
  if (p.K > 0 && p.dilation == 2) {                         // depthwise dilation (for K == 0 `dilation` is the dense tap spacing)
    if (p.K == 87) return launch_sep_k<87, 2, DBG, TT>(s, p);
    if (p.K == 15) return launch_sep_k<15, 2, DBG, TT>(s, p);
    return QASR_ERR_UNSUPPORTED;
  }
  switch (p.K) {
    case 0: return launch_sep_k<0, 1, DBG, TT>(s, p);
    case 11: return launch_sep_k<11, 1, DBG, TT>(s, p);
    case 13: return launch_sep_k<13, 1, DBG, TT>(s, p);
    case 33: return launch_sep_k<33, 1, DBG, TT>(s, p);
    case 39: return launch_sep_k<39, 1, DBG, TT>(s, p);
    case 51: return launch_sep_k<51, 1, DBG, TT>(s, p);
    case 63: return launch_sep_k<63, 1, DBG, TT>(s, p);
    case 75: return launch_sep_k<75, 1, DBG, TT>(s, p);
    default: return QASR_ERR_UNSUPPORTED;
  }
}

}  // namespace qasr
