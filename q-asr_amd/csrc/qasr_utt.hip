// k_utt<K, EP>: one separable layer for ONE WHOLE UTTERANCE per work-group (T <= 256 frames), throughput mode.
//
// With B*T = 8000 positions per batch a 32-frame tile (k_sep) re-reads every weight 250 times and leaves the
// depthwise conv on the VALU.  Giving a work-group the whole time axis of one utterance changes both:
//   * depthwise conv = Toeplitz MFMA (v_mfma_i32_16x16x64_i8): per channel, A = 16 windows x 64 input frames read
//     straight from the [C][T] row in L2 (16 contiguous bytes per lane), B = the channel's taps laid out as a 64 x 16
//     Toeplitz slab built in registers (aligned loads + v_alignbyte with a per-lane shift); 1-2 MFMAs per channel
//     replace ~20 v_dot4 per output.  The requantised s8 result goes to LDS as XC[frame][channel];
//   * 1x1 conv: wave w owns 32 output channels x all 256 frames (8 accumulator tiles), so each weight fragment
//     fetched from L2 feeds 8 MFMAs; A fragments come from XC.
// Only 32 work-groups exist per launch (one per utterance): the chip is filled by keeping several independent steps
// in flight on separate streams (bench.py --streams), which is also what hides the per-layer launch gaps.
//
// Epilogues: EP_PLAIN (ReLU? + per-channel requant to <= 3 consumers), EP_RQ32 (residual 1x1 conv: emit
// rint(acc * M) as int32 for the following layer's res_act), EP_ADD32 (res_act: clamp(rq(main) + r32) -> ReLU ->
// scalar / identity requant).  Anything else (logits, strided / dilated depthwise, T > 256) stays on k_sep.
#include <algorithm>

#include "qasr_device.h"

namespace qasr {

typedef int v4i_ __attribute__((ext_vector_type(4)));

#define UT_NT 512
#define UT_TMAX 256
#define UT_SP 48              // per-wave output staging pitch (32 frames + 16)
#define UT_NTW 4              // 32-frame accumulator tiles a wave keeps live (128 frames)
#define UT_CG 8               // depthwise channels a wave prefetches together

enum { UEP_PLAIN = 0, UEP_RQ32 = 1, UEP_ADD32 = 2 };

template <int K>
struct UttGeo {
  static constexpr int PAD = K / 2;
  static constexpr int HALO = (PAD + 15) / 16 * 16;           // window m starts at frame 16 m - HALO (16-B aligned)
  static constexpr int D = HALO - PAD;                         // tap k of output n sits at window offset n + D + k
  static constexpr int SPAN = 16 + D + K - 1;                  // window bytes that matter
  static constexpr int NS = (SPAN + 63) / 64;                  // 64-deep MFMA slabs per channel
  static constexpr int KP4 = (K + 3) / 4;
  static constexpr int WROW = 4 * KP4;                         // bytes per channel in the packed taps array
};

// Toeplitz B fragment of slab s for this lane: bytes j = 0..15 hold tap k = 64 s + 16 g + j - n - D (0 outside [0, K)).
// `wz` points at the channel's taps inside a zero-padded LDS copy (ZP zero bytes in front, >= 80 behind).
template <int K>
__device__ __forceinline__ v4i taps_fragment(const unsigned char* wz, int s, int g, int n) {
  using G = UttGeo<K>;
  const int off = 64 * s + 16 * g - n - G::D;                  // byte offset of j = 0 relative to tap 0 (may be negative)
  const int a = off & ~3;                                      // aligned-down dword address, shift = off & 3
  const int sh = off & 3;
  const unsigned* q = (const unsigned*)(wz + a);
  const unsigned d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
  v4i f;
  f[0] = (int)__builtin_amdgcn_alignbyte(d1, d0, (unsigned)sh);
  f[1] = (int)__builtin_amdgcn_alignbyte(d2, d1, (unsigned)sh);
  f[2] = (int)__builtin_amdgcn_alignbyte(d3, d2, (unsigned)sh);
  f[3] = (int)__builtin_amdgcn_alignbyte(d4, d3, (unsigned)sh);
  return f;
}

template <int K, int EP, bool DBG>
__global__ void __launch_bounds__(UT_NT) k_utt(SepP p) {
  using G = UttGeo<(K > 0 ? K : 1)>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const EpiP& e = p.e;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, r31 = lane & 31;
  const int b = blockIdx.x;
  const int eT = e.T, eTp = e.Tp, ecout = e.cout, n_outs = e.n_outs;
  const unsigned flags = e.flags;
  const bool f_relu = flags & QASR_F_RELU, f_exact = flags & QASR_F_EXACT_Z;
  const int len_b = e.lens[b];
  const int lim = (flags & QASR_F_MASK_OUT) ? min(len_b, eT) : eT;
  const int dlim = min(len_b, eT);
  const int XP = p.cin_pad + 16;
  const int ntile = eTp >> 5;                                  // 32-frame accumulator tiles (<= 8)
  const bool stamp = p.prof && blockIdx.x == 1 && tid == 0;
  int nst = 0;
#define USTAMP() do { if (stamp && nst < 31) p.prof[nst++] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
  USTAMP();
  unsigned char* XC = smem;                                    // [Tp][XP]  A operand of the 1x1 GEMM
  unsigned char* ST = XC + ((eTp + 127) / 128 * 128) * XP;     // [8 waves][32][UT_SP] private output staging
  constexpr int ZP = 32;                                       // zero bytes in front of tap 0
  constexpr int WZP = (ZP + G::WROW + 96 + 15) / 16 * 16;      // row pitch of WZ
  unsigned char* WZ = ST + 8 * 32 * UT_SP + wave * UT_CG * WZP;   // [8 waves][UT_CG][WZP] zero-padded tap rows of the wave's channel group

  if (K > 0) {
    // ---- depthwise stage.  Wave w owns a contiguous slice of channels, processed UT_CG (8) at a time: the A
    //      fragments (global, ~2 us under load) and taps of the whole group are requested before any is consumed,
    //      and 4 consecutive channels are requantised together so their bytes go to XC as one dword per frame.
    //      Tap rows live zero-padded in LDS (the Toeplitz fragments index them with per-lane byte offsets): the zeros
    //      are written once, the KP4 tap dwords of each row are replaced per group.
    static_assert(WZP / 4 <= 64, "tap row must fit one wave store");
    static_assert(UT_CG % 4 == 0, "groups of 4 channels");
    const int m16 = lane & 15, g = lane >> 4;
    const unsigned flip = p.x_unsigned ? 0x80808080u : 0u;
    const int nwin = eTp >> 4;                                 // valid windows (16 frames each)
    const int cpw = (p.cin_pad / 8 + UT_CG - 1) / UT_CG * UT_CG;   // channels per wave, multiple of the group
    const int dw_lo = p.dw_lo, dw_hi = p.dw_hi;
#pragma unroll
    for (int j = 0; j < UT_CG; ++j)
      if (lane < WZP / 4) ((unsigned*)(WZ + j * WZP))[lane] = 0u;
    const int c_end = min((wave + 1) * cpw, p.cin_pad);
    for (int c0 = wave * cpw; c0 < c_end; c0 += UT_CG) {
      // phase 1: request the group's operands (A fragments, taps, bias, multipliers)
      v4i a[UT_CG][G::NS];
      unsigned tapd[UT_CG];
      int biasv[UT_CG];
      double Mv[UT_CG];
#pragma unroll
      for (int j = 0; j < UT_CG; ++j) {
        const int c = c0 + j;
        const bool cok = c < p.cin;
        const int8_t* xrow = p.x + ((size_t)b * p.cin + (cok ? c : 0)) * eTp;
#pragma unroll
        for (int s = 0; s < G::NS; ++s) {
          // A: window m16, bytes 64 s + 16 g .. +15 -> frames 16 m16 - HALO + 64 s + 16 g (16-B granule, in or out of the row)
          const int t = 16 * m16 - G::HALO + 64 * s + 16 * g;
          a[j][s] = (v4i){0, 0, 0, 0};
          if (cok && t >= 0 && t < eTp) a[j][s] = *(const v4i*)(xrow + t);
        }
        tapd[j] = (cok && lane < G::KP4) ? ((const unsigned*)p.wdw)[c * G::KP4 + lane] : 0u;
        biasv[j] = cok ? p.bias_dw[c] : 0;
        Mv[j] = cok ? p.m_dw[c] : 0.0;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // previous group's fragment reads are issued
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < UT_CG; ++j)
        if (lane < G::KP4) ((unsigned*)(WZ + j * WZP + ZP))[lane] = tapd[j];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // phase 2: all Toeplitz fragments of the group (independent LDS reads, one wait), then all MFMAs back to back
      v4i bt[UT_CG][G::NS];
#pragma unroll
      for (int j = 0; j < UT_CG; ++j)
#pragma unroll
        for (int s = 0; s < G::NS; ++s) bt[j][s] = taps_fragment<(K > 0 ? K : 1)>(WZ + j * WZP + ZP, s, g, m16);
      __builtin_amdgcn_sched_barrier(0);
      v4i accd[UT_CG];
#pragma unroll
      for (int j = 0; j < UT_CG; ++j) {
        accd[j] = (v4i){0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < G::NS; ++s) {
          v4i av = a[j][s];
          av[0] ^= flip; av[1] ^= flip; av[2] ^= flip; av[3] ^= flip;
          accd[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bt[j][s], accd[j], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // phase 3: requantise 4 consecutive channels at a time, one dword per frame into XC
      // D: column n = lane & 15 (frame inside the window), rows m = 4 g + r (window)
#pragma unroll
      for (int j4 = 0; j4 < UT_CG; j4 += 4) {
        int z[16];                                             // [channel j][window register r]
        double M4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int r = 0; r < 4; ++r) z[4 * j + r] = accd[j4 + j][r] + biasv[j4 + j];
          M4[j] = Mv[j4 + j];
        }
        if (DBG && p.dw_acc_dbg) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int t = 16 * (4 * g + r) + m16, c = c0 + j4 + j;
              if (c < p.cin && 4 * g + r < nwin && t < eT) p.dw_acc_dbg[((size_t)b * p.cin + c) * eTp + t] = z[4 * j + r];
            }
        }
        int q[16];
        requant_batch4<16>(q, z, M4, dw_lo, dw_hi);
        // channels beyond cin inside the padded group have zero taps / bias / M: they requantise to 0 = XC's padding
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int t = 16 * (4 * g + r) + m16;
          const unsigned v = (t < dlim) ? pack4(q[r], q[4 + r], q[8 + r], q[12 + r]) : 0u;
          if (4 * g + r < nwin) *(unsigned*)(XC + t * XP + c0 + j4) = v;
        }
      }
    }
  } else {
    // ---- no depthwise stage: transpose the [cin][Tp] input tile into XC (4x4 byte transposes)
    const unsigned flip = p.pw_unsigned ? 0x80808080u : 0u;
    const int tq_n = eTp >> 2;
    for (int task = tid; task < (p.cin_pad / 4) * tq_n; task += UT_NT) {
      const int cq = task / tq_n, tq = task - cq * tq_n;
      unsigned r[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ci = 4 * cq + j;
        r[j] = (ci < p.cin) ? *(const unsigned*)(p.x + ((size_t)b * p.cin + ci) * eTp + 4 * tq) : 0u;
      }
      const unsigned lo01 = __builtin_amdgcn_perm(r[1], r[0], 0x05010400u), hi01 = __builtin_amdgcn_perm(r[1], r[0], 0x07030602u);
      const unsigned lo23 = __builtin_amdgcn_perm(r[3], r[2], 0x05010400u), hi23 = __builtin_amdgcn_perm(r[3], r[2], 0x07030602u);
      unsigned char* dst = XC + (4 * tq) * XP + 4 * cq;
      *(unsigned*)(dst) = __builtin_amdgcn_perm(lo23, lo01, 0x05040100u) ^ flip;
      *(unsigned*)(dst + XP) = __builtin_amdgcn_perm(lo23, lo01, 0x07060302u) ^ flip;
      *(unsigned*)(dst + 2 * XP) = __builtin_amdgcn_perm(hi23, hi01, 0x05040100u) ^ flip;
      *(unsigned*)(dst + 3 * XP) = __builtin_amdgcn_perm(hi23, hi01, 0x07060302u) ^ flip;
    }
  }
  USTAMP();
  __syncthreads();
  USTAMP();

  // ---- 1x1 GEMM: wave w -> output channels [256 pass + 32 w, +32) x UT_NTW accumulator tiles (128 frames) at a
  //      time; the second half of the frames re-reads the wave's weight rows (L2-resident, hidden under the MFMAs)
  const int cout_pad = (ecout + 127) / 128 * 128;
  unsigned char* st = ST + wave * 32 * UT_SP;
  for (int cbase = 0; cbase < cout_pad; cbase += 256) {
    const int co = cbase + 32 * wave + r31;
    if (cbase + 32 * wave >= cout_pad) continue;               // wave-uniform: no block barrier below this point
    const bool co_ok = co < ecout;
    const int bias = p.bias[co];
    // per-lane (= per output channel) and per-consumer parameters, read once per pass
    const float sb = (f_exact || DBG) ? e.sb[co] : 1.0f;
    const double m_main = (EP != UEP_PLAIN) ? e.m_main[co] : 0.0;
    double m_out[QASR_MAX_OUTS];
    int omode[QASR_MAX_OUTS], olo_[QASR_MAX_OUTS], ohi_[QASR_MAX_OUTS];
    int8_t* optr[QASR_MAX_OUTS];
#pragma unroll
    for (int j = 0; j < QASR_MAX_OUTS; ++j) {
      const bool on = j < n_outs;
      omode[j] = on ? e.outs[j].mode : 0;
      olo_[j] = on ? e.outs[j].lo : 0;
      ohi_[j] = on ? e.outs[j].hi : 0;
      optr[j] = on ? (int8_t*)e.outs[j].ptr : nullptr;
      m_out[j] = !on ? 0.0 : (EP == UEP_PLAIN ? e.outs[j].mtab[co] : e.outs[j].m);
    }
    const int qlo = e.qlo, qhi = e.qhi;
    int* const rq_out = (EP == UEP_RQ32) ? (int*)e.outs[0].ptr : nullptr;
    const int* const r32 = p.r32;

    for (int nt0 = 0; nt0 < ntile; nt0 += UT_NTW) {
      v16i acc[UT_NTW];
#pragma unroll
      for (int nt = 0; nt < UT_NTW; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = bias;
      v4i wf[4], wn[4];
      {
        const v4i* wp = w_frag(p.w, p.cin_pad, co, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) wf[i] = wp[64 * i];
      }
      for (int kc = 0; kc < p.cin_pad; kc += 128) {
        if (kc + 128 < p.cin_pad) {                            // next 128-deep chunk travels during this one's MFMAs
          const v4i* wp = w_frag(p.w, p.cin_pad, co, (kc + 128) >> 5);
#pragma unroll
          for (int i = 0; i < 4; ++i) wn[i] = wp[64 * i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const unsigned char* arow = XC + (32 * nt0 + r31) * XP + kc + 32 * i + 16 * h;
          v4i a[UT_NTW];
#pragma unroll
          for (int nt = 0; nt < UT_NTW; ++nt) a[nt] = *(const v4i*)(arow + nt * 32 * XP);   // rows beyond Tp: unused tiles
#pragma unroll
          for (int nt = 0; nt < UT_NTW; ++nt) acc[nt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[nt], wf[i], acc[nt], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) wf[i] = wn[i];
      }
      USTAMP();
      // ---- epilogue, one 32-frame tile at a time, staged through this wave's private LDS tile
#pragma unroll
      for (int nt = 0; nt < UT_NTW; ++nt) {
        if (nt0 + nt >= ntile) break;
        const int tb = 32 * (nt0 + nt);
        if (DBG && e.acc_dbg && co_ok) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int t = tb + mfma32_row(r, h);
            if (t < eT) e.acc_dbg[((size_t)b * ecout + co) * eTp + t] = acc[nt][r];
          }
        }
        int z[16];
        bool big = false;
        if (f_exact) {
#pragma unroll
          for (int r = 0; r < 16; ++r) big |= (acc[nt][r] >= (1 << 22)) | (acc[nt][r] <= -(1 << 22));
        }
        const bool exact = f_exact && __any(big);
        const bool relu_z = (EP == UEP_PLAIN) && f_relu;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          z[r] = exact ? z_roundtrip(acc[nt][r], sb, relu_z) : (relu_z ? max(acc[nt][r], 0) : acc[nt][r]);

        if (EP == UEP_RQ32) {
          // residual 1x1 conv of a block: its res_act operand rint(z * s_b / S), kept exact in int32
          if (co_ok) {
            int* op = rq_out + ((size_t)b * ecout + co) * eTp + tb;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              v4i v;
#pragma unroll
              for (int i = 0; i < 4; ++i) v[i] = (int)requant_d(z[4 * g4 + i], m_main);
              *(v4i*)(op + 8 * g4 + 4 * h) = v;
            }
          }
          continue;
        }
        if (EP == UEP_ADD32) {
          // res_act (jasper.py:680-682): clamp(rq(main) + rq(res)), then the block's ReLU
          const int* rp = r32 + ((size_t)b * ecout + (co_ok ? co : 0)) * eTp + tb;
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const v4i rv = *(const v4i*)(rp + 8 * g4 + 4 * h);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const double s = requant_d(z[4 * g4 + i], m_main) + (double)rv[i];
              const int q = (int)fmin(fmax(s, (double)qlo), (double)qhi);
              z[4 * g4 + i] = f_relu ? max(q, 0) : q;
            }
          }
        }
        const bool full = tb + 32 <= lim;                      // wave-uniform: no frame of this tile is masked
#pragma unroll
        for (int j = 0; j < QASR_MAX_OUTS; ++j) {
          if (j >= n_outs) break;
          int qo[16];
          if (EP == UEP_ADD32 && omode[j] == 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) qo[r] = z[r];
          } else {
            requant_batch<16>(qo, z, m_out[j], olo_[j], ohi_[j]);
          }
          if (!full) {
#pragma unroll
            for (int r = 0; r < 16; ++r) qo[r] = (tb + mfma32_row(r, h) < lim) ? qo[r] : 0;
          }
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4)
            *(unsigned*)(st + r31 * UT_SP + 8 * g4 + 4 * h) = pack4(qo[4 * g4], qo[4 * g4 + 1], qo[4 * g4 + 2], qo[4 * g4 + 3]);
          // wave-private tile: DS ops of one wave execute in order, only the compiler must not reorder them
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          {
            const int row = lane >> 1, half = lane & 1;        // 32 rows x 2 halves of 16 B
            const int cow = cbase + 32 * wave + row;
            const v4i v = *(const v4i*)(st + row * UT_SP + 16 * half);
            if (cow < ecout) *(v4i*)(optr[j] + ((size_t)b * ecout + cow) * eTp + tb + 16 * half) = v;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
      }
      USTAMP();
    }
  }
  if (stamp) p.prof[31] = nst;
}

static size_t utt_smem_bytes(const SepP& p, int K) {
  const int KP4 = (K + 3) / 4;
  const int WZP = (32 + 4 * KP4 + 96 + 15) / 16 * 16;
  const int rows = (p.e.Tp + 127) / 128 * 128;                 // the GEMM reads whole groups of UT_NTW tiles
  return (size_t)rows * (p.cin_pad + 16) + 8 * 32 * UT_SP + 8 * UT_CG * (size_t)WZP;
}

bool utt_supported(int K, int dilation, int Tp, int cin_pad, int cin) {
  if (Tp > UT_TMAX || dilation != 1) return false;
  if (!(K == 0 || K == 11 || K == 13 || K == 33 || K == 39 || K == 51 || K == 63 || K == 75)) return false;
  SepP q{};
  q.e.Tp = Tp;
  q.cin_pad = cin_pad;
  q.cin = cin;
  return utt_smem_bytes(q, K) <= 160 * 1024;
}

template <int K, int EP, bool DBG>
static void launch_utt_v(hipStream_t s, const SepP& p) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)k_utt<K, EP, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  SepP q = p;
  q.prof = g_prof_mode == 0 ? g_prof : nullptr;
  hipLaunchKernelGGL((k_utt<K, EP, DBG>), dim3(p.e.B), dim3(UT_NT), utt_smem_bytes(p, K), s, q);
}

template <int K>
static void launch_utt_k(hipStream_t s, const SepP& p, int ep) {
  const bool dbg = p.e.acc_dbg || p.dw_acc_dbg;
  if (dbg) {
    if (ep == UEP_PLAIN) launch_utt_v<K, UEP_PLAIN, true>(s, p);
    else if (ep == UEP_RQ32) launch_utt_v<K, UEP_RQ32, true>(s, p);
    else launch_utt_v<K, UEP_ADD32, true>(s, p);
  } else {
    if (ep == UEP_PLAIN) launch_utt_v<K, UEP_PLAIN, false>(s, p);
    else if (ep == UEP_RQ32) launch_utt_v<K, UEP_RQ32, false>(s, p);
    else launch_utt_v<K, UEP_ADD32, false>(s, p);
  }
}

void launch_utt(hipStream_t s, const SepP& p, int ep) {
  switch (p.K) {
    case 0: launch_utt_k<0>(s, p, ep); break;
    case 11: launch_utt_k<11>(s, p, ep); break;
    case 13: launch_utt_k<13>(s, p, ep); break;
    case 33: launch_utt_k<33>(s, p, ep); break;
    case 39: launch_utt_k<39>(s, p, ep); break;
    case 51: launch_utt_k<51>(s, p, ep); break;
    case 63: launch_utt_k<63>(s, p, ep); break;
    case 75: launch_utt_k<75>(s, p, ep); break;
    default: break;
  }
}

}  // namespace qasr
