// Depthwise taps on v_mfma_i32_16x16x64_i8 through a Toeplitz operand, against k_sep2's v_mfma_i32_4x4x4_16B_i8 form
// (VERDICT round 2, item 1).  Both kernels run the WHOLE depthwise stage of a 512-channel separable layer for one
// work-group (512 threads = 8 waves, one utterance x 128 frames): global window / tap rows -> wave-private LDS rows ->
// MFMA operands -> int32 accumulators -> exact fp64 requant -> packed codes in the [tile][channel][32] image the 1x1
// GEMM reads, i.e. 64 channels x 128 frames per wave incl. operand formation and the requantisation.  The image is
// dumped and compared with a CPU convolution (the arithmetic of oracle/int_oracle.py:conv1d_int + requant).
//
// Toeplitz form, one MFMA = TWO channels x 128 frames x 32 window offsets:
//   C[i][j] += sum_u A[i][u] B[u][j],  i = frame inside a 16-frame tile, j = (channel of the pair, tile 0..7)
//   A[i][u] = w_c[u - i + 32 s]  (c = channel A for u < 32, channel B for u >= 32: the K dimension is split between the
//             two channels; lane (i, kg) reads ONE unaligned 16-byte run of the zero-margined tap row)
//   B[u][j] = x_c[16 (j & 7) + (u & 31) + 32 s] for the column's own channel, 0 for the other channel's K half
//             (lane (j, kg): one ALIGNED 16-byte run of the window row, or 16 bytes of a zero region)
//   steps s = 0 .. ceil((K + 15) / 32) - 1;  C layout: lane (j, g) holds frames 16 (j & 7) + 4 g + {0..3} of its
//   channel: one packed dword of the image.
// RESULT (toeplitz_r03.txt): bit-exact, and 3.5-4x SLOWER than the 4x4x4 form - a 16-byte LDS read that is not 16-byte
// aligned is served one lane per cycle (64 cycles per wave-instruction, ldsalign.hip / ldsalign_r03.txt), and every A
// operand of the Toeplitz form is such a read.  DESIGN.md 5.5 has the accounting of the alternatives (funnel shifts in
// registers, pre-shifted copies) - none beats the 4x4x4 stage.
// Build: hipcc --offload-arch=gfx950 -O3 -I../../q-asr_amd/csrc -o toeplitz toeplitz.hip   (output: toeplitz_r03.txt)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "qasr_sep2_impl.h"

using namespace qasr;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int NCH = 512, TPR = 256, TILE = 128;              // channels, row pitch (frames), frames per work-group

struct __attribute__((packed)) uv4 { v4i v; };             // 16 bytes at any byte address
typedef uv4 __attribute__((address_space(3))) lds_uv4;

// --------------------------------------------------------------------------------------------- Toeplitz depthwise stage
template <int K>
struct TzGeo {
  static constexpr int PAD = K / 2;
  static constexpr int NS = (K + 15 + 31) / 32;             // steps of 32 window offsets per channel
  static constexpr int HALO = (PAD + 15) / 16 * 16;         // global granules start at t0 - HALO
  static constexpr int D = HALO - PAD;                      // bytes of the first granule in front of window position 0
  static constexpr int WLEN = 16 * 7 + 32 * NS;             // window positions the B reads touch
  static constexpr int NPG = (D + WLEN + 15) / 16;          // 16-byte global granules per row
  static constexpr int WP = 16 + 16 * NPG + 16;             // LDS row pitch: window position n at byte 16 + n
  static constexpr int TW = 16 + 32 * NS;                   // tap row: 16 zero bytes, K taps, zeros up to TW, then {M, bias}
  static constexpr int TPITCH = TW + 16;
  static constexpr int NPT = (16 * NPG + 63) / 64;          // window granules per lane and group
  static constexpr int NTT = (TPITCH + 63) / 64;            // tap-row granules per lane and group (16 rows x TPITCH / 16)
  static constexpr int WREG = 16 * WP + 16 * TPITCH;        // one wave's private rows
  static constexpr int ZLEN = (14 * WP + 32 * NS + 15) / 16 * 16;   // zero region behind every immediate offset of a B read
  static constexpr size_t SMEM = (size_t)TILE * NCH + ZLEN + 8 * WREG;
  static_assert(WLEN >= TILE + K - 1, "window short");
};

template <int K, bool ILV>
__global__ void __launch_bounds__(512, 2)
k_tz(const uint8_t* __restrict__ x, const uint8_t* __restrict__ trows, int8_t* __restrict__ img, long long* __restrict__ prof,
     int T, int dw_lo, int dw_hi, int dump) {
  using G = TzGeo<K>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x, t0 = blockIdx.y * TILE;
  lds_u8* const Xd = (lds_u8*)smem;                          // [4][NCH][32]
  lds_u8* const ZB = Xd + TILE * NCH;
  lds_u8* const Wsw = ZB + G::ZLEN + wave * G::WREG;
  lds_u8* const Tlw = Wsw + 16 * G::WP;
  const long long c0 = __builtin_amdgcn_s_memtime();
  asm volatile("" : "+v"(dw_lo), "+v"(dw_hi));

  // ---- staging constants (per lane, the same for every group)
  int woff[G::NPT], toff[G::NTT];
  unsigned wkeep[G::NPT];
  lds_u8* wlds[G::NPT];
#pragma unroll
  for (int i = 0; i < G::NPT; ++i) {
    const int pi = lane + 64 * i;
    const int row = pi / G::NPG, col = pi - row * G::NPG;
    const int t = t0 - G::HALO + 16 * col;
    woff[i] = min(row, 15) * TPR + min(max(t, 0), TPR - 16);
    wkeep[i] = (t >= 0 && t < TPR) ? 0xffffffffu : 0u;
    wlds[i] = Wsw + min(row, 15) * G::WP + 16 + 16 * col - G::D;      // (unaligned 16-byte store when D % 16 != 0)
    asm volatile("" : "+v"(wkeep[i]));
  }
#pragma unroll
  for (int i = 0; i < G::NTT; ++i) toff[i] = 16 * min(lane + 64 * i, G::TPITCH - 1);
  v4i pc[G::NPT], pt[G::NTT];
  auto ld_grp = [&](int g) {
    const int cw = 128 * g + 16 * wave;
    const uint8_t* const xg = x + ((size_t)b * NCH + cw) * TPR;
    const uint8_t* const tg = trows + (size_t)cw * G::TPITCH;
#pragma unroll
    for (int i = 0; i < G::NPT; ++i) pc[i] = *(const v4i*)(xg + woff[i]);
#pragma unroll
    for (int i = 0; i < G::NTT; ++i) pt[i] = *(const v4i*)(tg + toff[i]);
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < G::NPT; ++i) {
      if (64 * i + 63 < 16 * G::NPG || lane + 64 * i < 16 * G::NPG) {
        v4i v = pc[i];
        v[0] = (v[0] & wkeep[i]) ^ 0x80808080u; v[1] = (v[1] & wkeep[i]) ^ 0x80808080u;
        v[2] = (v[2] & wkeep[i]) ^ 0x80808080u; v[3] = (v[3] & wkeep[i]) ^ 0x80808080u;
        ((lds_uv4*)wlds[i])->v = v;
      }
    }
#pragma unroll
    for (int i = 0; i < G::NTT; ++i) {
      const int gi = lane + 64 * i;
      if (64 * i + 63 < G::TPITCH || gi < G::TPITCH) *(lds_v4i*)(Tlw + 16 * gi) = pt[i];
    }
  };
  ld_grp(0);
  for (int i = tid * 16; i < G::ZLEN; i += 512 * 16) *(lds_v4i*)(ZB + i) = (v4i){0, 0, 0, 0};

  // ---- operand addresses of the lane
  const int i16 = lane & 15, kg = lane >> 4, kgl = kg & 1, kgh = kg >> 1, cj = i16 >> 3;
  const lds_u8* a_lane = Tlw + kgh * G::TPITCH + 16 + 16 * kgl - i16;
  const lds_u8* b_lane = (cj == kgh) ? Wsw + kgh * G::WP + 16 + 16 * (i16 & 7) + 16 * kgl : ZB;
  const lds_u8* p_lane = Tlw + cj * G::TPITCH + G::TW;
  const int f0 = 16 * (i16 & 7) + 4 * kg;                     // first of the lane's 4 consecutive output frames
  lds_u8* x_lane = Xd + (f0 >> 5) * (NCH * 32) + (16 * wave + cj) * 32 + (f0 & 31);
  asm volatile("" : "+v"(a_lane), "+v"(b_lane), "+v"(p_lane), "+v"(x_lane));
  const int dlim = min(T, TPR);
  unsigned fmask;
  {
    const int n = min(max(dlim - t0 - f0, 0), 4);
    fmask = n >= 4 ? 0xffffffffu : ((1u << (8 * n)) - 1u);
  }
  __syncthreads();                                            // zero region

  auto fence_w = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  commit();
  ld_grp(1);
  fence_w();
  const long long c1 = __builtin_amdgcn_s_memtime();

  struct Ops { v4i a[G::NS], bb[G::NS], pp; };
  auto rd_pair = [&](Ops& o, auto pc_) {
    constexpr int p = decltype(pc_)::value;
#pragma unroll
    for (int s = 0; s < G::NS; ++s) {
      o.a[s] = ((const lds_uv4*)(a_lane + 2 * p * G::TPITCH + 32 * s))->v;
      o.bb[s] = *(const lds_v4i*)(b_lane + 2 * p * G::WP + 32 * s);
    }
    o.pp = *(const lds_v4i*)(p_lane + 2 * p * G::TPITCH);
  };
  auto requant = [&](const v4i& acc, double M, auto gc, auto pc_) {
    constexpr int g = decltype(gc)::value, p = decltype(pc_)::value;
    const unsigned w = pack4b(rq_clamp(acc[0], M, dw_lo, dw_hi), rq_clamp(acc[1], M, dw_lo, dw_hi),
                              rq_clamp(acc[2], M, dw_lo, dw_hi), rq_clamp(acc[3], M, dw_lo, dw_hi));
    *(lds_u32*)(x_lane + (128 * g + 2 * p) * 32) = w & fmask;
  };

  sep2_for<0, 4>([&](auto gc) {
    constexpr int g = decltype(gc)::value;
    v4i acc[8];
    double Mp[8];
    if constexpr (!ILV) {
      Ops o[2];
      rd_pair(o[0], std::integral_constant<int, 0>{});
      sep2_for<0, 8>([&](auto pc_) {
        constexpr int p = decltype(pc_)::value;
        if constexpr (p + 1 < 8) rd_pair(o[(p + 1) & 1], std::integral_constant<int, p + 1>{});
        const Ops& c = o[p & 1];
        acc[p] = (v4i){c.pp[2], c.pp[2], c.pp[2], c.pp[2]};
        Mp[p] = __hiloint2double(c.pp[1], c.pp[0]);
#pragma unroll
        for (int s = 0; s < G::NS; ++s) acc[p] = __builtin_amdgcn_mfma_i32_16x16x64_i8(c.a[s], c.bb[s], acc[p], 0, 0, 0);
      });
      // every read of this group's rows has been issued: the next group's rows may overwrite them (a wave's LDS
      // instructions execute in order)
      if constexpr (g + 1 < 4) {
        fence_w();
        commit();
        if (g + 2 < 4) ld_grp(g + 2);
        fence_w();
      }
      sep2_for<0, 8>([&](auto pc_) { requant(acc[decltype(pc_)::value], Mp[decltype(pc_)::value], gc, pc_); });
    } else {
      // pair p's MFMAs, then pair p-1's requantisation (its last MFMA is NS MFMAs old), operands of pair p+1 in flight
      Ops o[2];
      rd_pair(o[0], std::integral_constant<int, 0>{});
      sep2_for<0, 8>([&](auto pc_) {
        constexpr int p = decltype(pc_)::value;
        if constexpr (p + 1 < 8) rd_pair(o[(p + 1) & 1], std::integral_constant<int, p + 1>{});
        const Ops& c = o[p & 1];
        acc[p] = (v4i){c.pp[2], c.pp[2], c.pp[2], c.pp[2]};
        Mp[p] = __hiloint2double(c.pp[1], c.pp[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < G::NS; ++s) acc[p] = __builtin_amdgcn_mfma_i32_16x16x64_i8(c.a[s], c.bb[s], acc[p], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (p > 0) requant(acc[p - 1], Mp[p - 1], gc, std::integral_constant<int, p - 1>{});
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (g + 1 < 4) {
        fence_w();
        commit();
        if (g + 2 < 4) ld_grp(g + 2);
        fence_w();
      }
      requant(acc[7], Mp[7], gc, std::integral_constant<int, 7>{});
      __builtin_amdgcn_sched_barrier(0);
    }
  });
  const long long c2 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  const long long c3 = __builtin_amdgcn_s_memtime();
  if (lane == 0) {
    long long* r = prof + 4 * ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave);
    r[0] = c1 - c0; r[1] = c2 - c1; r[2] = c3 - c0; r[3] = 0;
  }
  if (dump) {
    int8_t* const o = img + (size_t)(b * 2 + blockIdx.y) * TILE * NCH;
    for (int i = tid * 16; i < TILE * NCH; i += 512 * 16) *(v4i*)(o + i) = *(const lds_v4i*)(Xd + i);
  }
}

// --------------------------------------------------------------------------------------------- k_sep2's 4x4x4 stage
// (qasr_sep2_impl.h: wave-private rows, aligned lane streams, the previous group's requantisation one instruction behind
//  every MFMA - the production code path of round 2, lifted out of k_sep2 with the same geometry Sep2Geo<K, 128>)
// 16 six-bit codes (12 bytes: 4 codes per 3 bytes, little-endian bit order as QASR_F_W6PACK) -> 16 bytes
__device__ __forceinline__ unsigned spread6(unsigned g) {
  return (g & 0x3fu) | ((g & 0xfc0u) << 2) | ((g & 0x3f000u) << 4) | ((g & 0xfc0000u) << 6);
}
__device__ __forceinline__ v4i unpack6x16(unsigned a, unsigned b, unsigned c) {
  const unsigned g1 = __builtin_amdgcn_alignbit(b, a, 24), g2 = __builtin_amdgcn_alignbit(c, b, 16);
  return (v4i){(int)spread6(a), (int)spread6(g1), (int)spread6(g2), (int)spread6(c >> 8)};
}

// PACK6: BASELINE config 3 "to the letter" for the activations - the input rows hold 4 six-bit codes per 3 bytes
// ([C][Tp * 3 / 4]) and are unpacked where the stage touches every staged dword anyway (A/B of the stage)
template <int K, bool PACK6 = false>
__global__ void __launch_bounds__(512, 2)
k_old(const uint8_t* __restrict__ x, const int8_t* __restrict__ wdw2, const int* __restrict__ bias_dw, const double* __restrict__ m_dw,
      int8_t* __restrict__ img, long long* __restrict__ prof, int T, int dw_lo, int dw_hi, int dump) {
  constexpr int TT = TILE, CIN_PAD = NCH, NCHUNK = 4;
  using G = Sep2Geo<K, TT>;
  constexpr int NU = G::NU, S = G::S, NS = G::NS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x, t0 = blockIdx.y * TT;
  lds_u8* const Xd = (lds_u8*)smem;
  lds_u8* const Un = Xd + TT * CIN_PAD;
  lds_u8* const Wsw = Un + wave * G::WREG;
  lds_u8* const Tlw = Wsw + 16 * G::WP;
  const long long c0 = __builtin_amdgcn_s_memtime();
  asm volatile("" : "+v"(dw_lo), "+v"(dw_hi));
  const int eTp = TPR, dlim = min(T, TPR);
  const unsigned flip = 0x80808080u;
  v4i pc[G::NPT], pt[G::NTT];
  int woff[G::NPT], toff[G::NTT];
  unsigned wkeep[G::NPT];
  lds_u8* wlds[G::NPT];
#pragma unroll
  for (int i = 0; i < G::NPT; ++i) {
    const int pi = lane + 64 * i;
    const int row = pi / G::NPG, col = pi - row * G::NPG;
    const int t = t0 - G::HALO + 16 * col;
    woff[i] = PACK6 ? min(row, 15) * (eTp * 3 / 4) + min(max(t, 0), eTp - 16) / 16 * 12 : min(row, 15) * eTp + min(max(t, 0), eTp - 16);
    wkeep[i] = (t >= 0 && t < eTp) ? 0xffffffffu : 0u;
    wlds[i] = Wsw + min(row, 15) * G::WP + 16 * col;
    asm volatile("" : "+v"(wkeep[i]));
  }
#pragma unroll
  for (int i = 0; i < G::NTT; ++i) toff[i] = 16 * min(lane + 64 * i, G::KS - 1);
  auto ld_grp = [&](int g) {
    const int cw = SEP2_CH * g + 16 * wave;
    const uint8_t* const xg = x + ((size_t)b * CIN_PAD + cw) * (PACK6 ? eTp * 3 / 4 : eTp);
    const unsigned char* const tg = (const unsigned char*)wdw2 + (size_t)cw * G::KS;
#pragma unroll
    for (int i = 0; i < G::NPT; ++i) {
      if constexpr (PACK6) {
        const unsigned* q = (const unsigned*)(xg + woff[i]);   // 12 bytes, 4-byte aligned: global_load_dwordx3
        pc[i] = (v4i){(int)q[0], (int)q[1], (int)q[2], 0};
      } else {
        pc[i] = *(const v4i*)(xg + woff[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < G::NTT; ++i) pt[i] = *(const v4i*)(tg + toff[i]);
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < G::NPT; ++i) {
      if (64 * i + 63 < 16 * G::NPG || lane + 64 * i < 16 * G::NPG) {
        v4i v = pc[i];
        if constexpr (PACK6) v = unpack6x16((unsigned)v[0], (unsigned)v[1], (unsigned)v[2]);
        v[0] = (v[0] & wkeep[i]) ^ flip; v[1] = (v[1] & wkeep[i]) ^ flip; v[2] = (v[2] & wkeep[i]) ^ flip; v[3] = (v[3] & wkeep[i]) ^ flip;
        *(lds_v4i*)wlds[i] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < G::NTT; ++i) {
      const int gi = lane + 64 * i;
      if (64 * i + 63 < G::KS || gi < G::KS) *(lds_v4i*)(Tlw + 16 * gi) = pt[i];
    }
  };
  ld_grp(0);
  const int cb = lane >> 2, jl = lane & 3;
  int dbias[NCHUNK];
  double dM[NCHUNK];
#pragma unroll
  for (int gi = 0; gi < NCHUNK; ++gi) {
    const int c = SEP2_CH * gi + 16 * wave + cb;
    dbias[gi] = bias_dw[c];
    dM[gi] = m_dw[c];
  }
  constexpr int e0base = 8 + G::MS;
  const int e0 = e0base - jl, tq = e0 >> 2, tsh = e0 & 3;
  unsigned fmask[NU];
  {
    const int dl = dlim - t0 - S * jl;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int n = min(max(dl - 4 * u, 0), 4);
      fmask[u] = n >= 4 ? 0xffffffffu : ((1u << (8 * n)) - 1u);
    }
  }
  struct DwIn {
    unsigned raw[NS + 1];
    unsigned xs[G::NRD * (G::RG / 4)];
  };
  const lds_u32* tr_lane = (const lds_u32*)(Tlw + cb * G::KS + 4 * tq);
  const lds_u8* wr_lane = Wsw + cb * G::WP + S * jl + (G::A0 & ~(G::RG - 1));
  asm volatile("" : "+v"(tr_lane), "+v"(wr_lane));
  auto dw_read = [&](DwIn& in) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i <= NS; ++i) in.raw[i] = tr_lane[i];
    sep2_rd_stream<G::OFF, G::NE, G::RG>(in.xs, wr_lane);
  };
  auto dw_mfma = [&](const DwIn& in, v4i (&acc)[NU], int bias) __attribute__((always_inline)) {
    unsigned tw[NS];
#pragma unroll
    for (int st = 0; st < NS; ++st) tw[st] = __builtin_amdgcn_alignbyte(in.raw[st + 1], in.raw[st], tsh);
#pragma unroll
    for (int u = 0; u < NU; ++u) acc[u] = (v4i){bias, bias, bias, bias};
#pragma unroll
    for (int st = 0; st < NS; ++st)
#pragma unroll
      for (int u = 0; u < NU; ++u)
        acc[u] = __builtin_amdgcn_mfma_i32_4x4x4i8((int)tw[st], (int)in.xs[G::OFF + u + st], acc[u], 0, 0, 0);
  };
  auto dw_out = [&](v4i (&acc)[NU], int c0, double Mg) __attribute__((always_inline)) {
    const int c = c0 + 16 * wave + cb;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int f = S * jl + 4 * u;
      const unsigned w = pack4b(rq_clamp(acc[u][0], Mg, dw_lo, dw_hi), rq_clamp(acc[u][1], Mg, dw_lo, dw_hi),
                                rq_clamp(acc[u][2], Mg, dw_lo, dw_hi), rq_clamp(acc[u][3], Mg, dw_lo, dw_hi));
      *(lds_u32*)(Xd + (f >> 5) * (CIN_PAD * 32) + c * 32 + (f & 31)) = w & fmask[u];
    }
  };
  auto dw_mfma_ilv = [&](const DwIn& in, v4i (&acc)[NU], int bias, v4i (&prev)[NU], int cprev, double Mprev) __attribute__((always_inline)) {
    constexpr int NOPS = 17 * NU, OPM = (NOPS + NS * NU - 1) / (NS * NU);
    const int c = cprev + 16 * wave + cb;
#pragma unroll
    for (int u = 0; u < NU; ++u) acc[u] = (v4i){bias, bias, bias, bias};
    unsigned tws[2] = {__builtin_amdgcn_alignbyte(in.raw[1], in.raw[0], tsh), 0u}, P1 = 0, P2 = 0;
    double dreg = 0.0;
    int q[4] = {0, 0, 0, 0};
    __builtin_amdgcn_sched_barrier(0);
    sep2_for<0, NS * NU>([&](auto ic) {
      constexpr int i = decltype(ic)::value, st = i / NU, u = i % NU;
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
            constexpr int f0 = 4 * dw;
            const int f = S * jl + f0;
            *(lds_u32*)(Xd + (f >> 5) * (CIN_PAD * 32) + c * 32 + (f & 31)) = P1;
          }
        }
      });
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  DwIn in;
  auto stage = [&](auto chc) {
    constexpr int CH = decltype(chc)::value;
    commit();
    if constexpr (CH + 1 < NCHUNK) ld_grp(CH + 1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    dw_read(in);
    __builtin_amdgcn_sched_barrier(0);
  };
  v4i accs[2][NU];
  stage(std::integral_constant<int, 0>{});
  const long long c1 = __builtin_amdgcn_s_memtime();
  sep2_for<0, NCHUNK>([&](auto chc) {
    constexpr int CH = decltype(chc)::value;
    constexpr int cc0 = SEP2_CH * CH;
    if constexpr (CH == 0) dw_mfma(in, accs[0], dbias[0]);
    else dw_mfma_ilv(in, accs[CH & 1], dbias[CH], accs[(CH - 1) & 1], cc0 - SEP2_CH, dM[CH - 1]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (CH + 1 < NCHUNK) stage(std::integral_constant<int, CH + 1>{});
    else dw_out(accs[CH & 1], cc0, dM[CH]);
  });
  const long long c2 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  const long long c3 = __builtin_amdgcn_s_memtime();
  if (lane == 0) {
    long long* r = prof + 4 * ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave);
    r[0] = c1 - c0; r[1] = c2 - c1; r[2] = c3 - c0; r[3] = 0;
  }
  if (dump) {
    int8_t* const o = img + (size_t)(b * 2 + blockIdx.y) * TILE * NCH;
    for (int i = tid * 16; i < TILE * NCH; i += 512 * 16) *(v4i*)(o + i) = *(const lds_v4i*)(Xd + i);
  }
}

// --------------------------------------------------------------------------------------------- host
static std::vector<int8_t> cpu_image(const std::vector<uint8_t>& x, const std::vector<int8_t>& w, const std::vector<double>& M, int B,
                                     int K, int T, int lo, int hi) {
  std::vector<int8_t> img((size_t)B * 2 * TILE * NCH);
  const int pad = K / 2;
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < NCH; ++c) {
      const uint8_t* xr = &x[((size_t)b * NCH + c) * TPR];
      for (int t = 0; t < 2 * TILE; ++t) {
        int q = 0;
        if (t < T) {
          long long acc = 0;
          for (int m = 0; m < K; ++m) {
            const int tt = t - pad + m;
            if (tt >= 0 && tt < TPR) acc += (int)w[(size_t)c * K + m] * (int)xr[tt];
          }
          const double r = std::nearbyint((double)acc * M[c]);
          q = (int)std::min(std::max(r, (double)lo), (double)hi);
        }
        const int tile = t / TILE, f = t % TILE;
        img[(size_t)(b * 2 + tile) * TILE * NCH + (size_t)(f >> 5) * (NCH * 32) + c * 32 + (f & 31)] = (int8_t)q;
      }
    }
  return img;
}

struct Times { double us; long long pre, dw, all; };
template <class L>
static Times run(L&& launch, long long* dprof, int nwg, int reps) {
  launch(1);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) launch(0);
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch(0);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> p((size_t)nwg * 8 * 4);
  CK(hipMemcpy(p.data(), dprof, p.size() * 8, hipMemcpyDeviceToHost));
  Times t{ms * 1e3 / reps, 0, 0, 0};
  std::vector<long long> all;
  for (int i = 0; i < nwg; ++i) {
    long long a = 0, pr = 0, dw = 0;
    for (int w = 0; w < 8; ++w) { a = std::max(a, p[(i * 8 + w) * 4 + 2]); pr = std::max(pr, p[(i * 8 + w) * 4]); dw = std::max(dw, p[(i * 8 + w) * 4 + 1]); }
    all.push_back(a); t.pre += pr; t.dw += dw;
  }
  std::sort(all.begin(), all.end());
  t.all = all[all.size() / 2]; t.pre /= nwg; t.dw /= nwg;
  return t;
}

template <int K>
static void bench_k(int B) {
  using G = TzGeo<K>;
  using GO = Sep2Geo<K, TILE>;
  const int T = 250, lo = -128, hi = 127;
  std::mt19937 rng(K * 7919 + 1);
  std::vector<uint8_t> x((size_t)B * NCH * TPR, 0);
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < NCH; ++c)
      for (int t = 0; t < T; ++t) x[((size_t)b * NCH + c) * TPR + t] = (uint8_t)(rng() & 255);
  std::vector<int8_t> w((size_t)NCH * K);
  for (auto& v : w) v = (int8_t)((int)(rng() % 254) - 127);
  std::vector<double> M(NCH);
  std::vector<int> bias(NCH);
  for (int c = 0; c < NCH; ++c) {
    const double mant = 0.5 + (rng() % (1u << 30)) / (double)(1u << 31);          // [0.5, 1), 31 significant bits
    M[c] = std::ldexp(std::floor(mant * 2147483648.0 + 0.5), -31 - 11 - (int)(rng() % 3));
    if (c % 37 == 0) M[c] = std::ldexp(1.0, -10);                                  // exact ties: z * 2^-10 = n + 1/2
    int s = 0;
    for (int m = 0; m < K; ++m) s += w[(size_t)c * K + m];
    bias[c] = 128 * s;
  }
  // tap rows of the Toeplitz form: [16 zero bytes][K taps][zeros .. TW) {M f64, bias i32, 0}
  std::vector<uint8_t> trows((size_t)NCH * G::TPITCH + 256, 0);
  for (int c = 0; c < NCH; ++c) {
    uint8_t* r = &trows[(size_t)c * G::TPITCH];
    memcpy(r + 16, &w[(size_t)c * K], K);
    memcpy(r + G::TW, &M[c], 8);
    memcpy(r + G::TW + 8, &bias[c], 4);
  }
  // k_sep2's rows: taps behind 8 zero bytes, pitch KS
  std::vector<int8_t> wdw2((size_t)NCH * GO::KS + 256, 0);
  for (int c = 0; c < NCH; ++c) memcpy(&wdw2[(size_t)c * GO::KS + 8], &w[(size_t)c * K], K);
  const std::vector<int8_t> ref = cpu_image(x, w, M, B, K, T, lo, hi);

  uint8_t *dx, *dtr; int8_t *dw2, *dimg; int* dbias; double* dM; long long* dprof;
  CK(hipMalloc(&dx, x.size() + 256)); CK(hipMalloc(&dtr, trows.size())); CK(hipMalloc(&dw2, wdw2.size()));
  CK(hipMalloc(&dimg, ref.size())); CK(hipMalloc(&dbias, NCH * 4)); CK(hipMalloc(&dM, NCH * 8));
  CK(hipMalloc(&dprof, (size_t)B * 2 * 8 * 4 * 8));
  CK(hipMemcpy(dx, x.data(), x.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dtr, trows.data(), trows.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dw2, wdw2.data(), wdw2.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dbias, bias.data(), NCH * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dM, M.data(), NCH * 8, hipMemcpyHostToDevice));
  const dim3 grid(B, 2), blk(512);
  auto check = [&](const char* name) {
    std::vector<int8_t> got(ref.size());
    CK(hipMemcpy(got.data(), dimg, got.size(), hipMemcpyDeviceToHost));
    size_t bad = 0, first = 0;
    for (size_t i = 0; i < ref.size(); ++i)
      if (got[i] != ref[i]) { if (!bad) first = i; ++bad; }
    printf("  %-22s parity vs CPU conv + requant: %s (%zu of %zu bytes differ%s)\n", name, bad ? "FAIL" : "bit-exact", bad, ref.size(),
           bad ? "" : "");
    if (bad) printf("    first at %zu: got %d want %d\n", first, got[first], ref[first]);
    return bad == 0;
  };
  const size_t smem_old = (size_t)TILE * NCH + 8 * GO::WREG;
  CK(hipFuncSetAttribute((const void*)k_old<K, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)k_tz<K, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)k_tz<K, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  printf("K = %d, %d work-groups (512 threads, 512 channels x 128 frames each): Toeplitz NS = %d MFMA steps per channel pair, LDS %zu B; "
         "4x4x4: %d MFMAs per 16 channels, LDS %zu B\n", K, 2 * B, G::NS, G::SMEM, GO::NS * GO::NU, smem_old);
  CK(hipMemset(dimg, 0x55, ref.size()));
  Times to = run([&](int dump) { hipLaunchKernelGGL((k_old<K, false>), grid, blk, smem_old, 0, dx, dw2, dbias, dM, dimg, dprof, T, lo, hi, dump); }, dprof, 2 * B, 50);
  check("4x4x4 (k_sep2 r02)");
  CK(hipMemset(dimg, 0x55, ref.size()));
  Times t1 = run([&](int dump) { hipLaunchKernelGGL((k_tz<K, false>), grid, blk, G::SMEM, 0, dx, dtr, dimg, dprof, T, lo, hi, dump); }, dprof, 2 * B, 50);
  check("16x16x64 Toeplitz");
  CK(hipMemset(dimg, 0x55, ref.size()));
  Times t2 = run([&](int dump) { hipLaunchKernelGGL((k_tz<K, true>), grid, blk, G::SMEM, 0, dx, dtr, dimg, dprof, T, lo, hi, dump); }, dprof, 2 * B, 50);
  check("16x16x64 Toeplitz ilv");
  printf("  %-22s launch %7.2f us | work-group cycles: first rows in LDS %6lld, 4 groups (slowest wave) %6lld, until all waves done (median) %6lld\n",
         "4x4x4 (k_sep2 r02)", to.us, to.pre, to.dw, to.all);
  printf("  %-22s launch %7.2f us | work-group cycles: first rows in LDS %6lld, 4 groups (slowest wave) %6lld, until all waves done (median) %6lld\n",
         "16x16x64 Toeplitz", t1.us, t1.pre, t1.dw, t1.all);
  printf("  %-22s launch %7.2f us | work-group cycles: first rows in LDS %6lld, 4 groups (slowest wave) %6lld, until all waves done (median) %6lld\n",
         "16x16x64 Toeplitz ilv", t2.us, t2.pre, t2.dw, t2.all);
  printf("  stage speed-up (work-group cycles until all waves done): %.2fx / %.2fx\n", (double)to.all / t1.all, (double)to.all / t2.all);
  CK(hipFree(dx)); CK(hipFree(dtr)); CK(hipFree(dw2)); CK(hipFree(dimg)); CK(hipFree(dbias)); CK(hipFree(dM)); CK(hipFree(dprof));
}

// BASELINE config 3 activations: the same stage on 6-bit codes [0, 63], rows one byte per code vs 4 codes per 3 bytes
template <int K>
static void bench_pack6(int B) {
  using GO = Sep2Geo<K, TILE>;
  const int T = 250, lo = -32, hi = 31;
  std::mt19937 rng(K * 31 + 5);
  std::vector<uint8_t> x((size_t)B * NCH * TPR, 0), xp((size_t)B * NCH * TPR * 3 / 4 + 256, 0);
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < NCH; ++c) {
      uint8_t* r = &x[((size_t)b * NCH + c) * TPR];
      for (int t = 0; t < T; ++t) r[t] = (uint8_t)(rng() & 63);
      uint8_t* q = &xp[((size_t)b * NCH + c) * (TPR * 3 / 4)];
      for (int t = 0; t < TPR; t += 4) {
        const unsigned w = r[t] | (r[t + 1] << 6) | (r[t + 2] << 12) | (r[t + 3] << 18);
        q[t / 4 * 3] = w & 255; q[t / 4 * 3 + 1] = (w >> 8) & 255; q[t / 4 * 3 + 2] = (w >> 16) & 255;
      }
    }
  std::vector<int8_t> w((size_t)NCH * K);
  for (auto& v : w) v = (int8_t)((int)(rng() % 62) - 31);
  std::vector<double> M(NCH);
  std::vector<int> bias(NCH);
  for (int c = 0; c < NCH; ++c) {
    const double mant = 0.5 + (rng() % (1u << 30)) / (double)(1u << 31);
    M[c] = std::ldexp(std::floor(mant * 2147483648.0 + 0.5), -31 - 9 - (int)(rng() % 3));
    int sum = 0;
    for (int m = 0; m < K; ++m) sum += w[(size_t)c * K + m];
    bias[c] = 128 * sum;
  }
  std::vector<int8_t> wdw2((size_t)NCH * GO::KS + 256, 0);
  for (int c = 0; c < NCH; ++c) memcpy(&wdw2[(size_t)c * GO::KS + 8], &w[(size_t)c * K], K);
  const std::vector<int8_t> ref = cpu_image(x, w, M, B, K, T, lo, hi);
  uint8_t *dx, *dxp; int8_t *dw2, *dimg; int* dbias; double* dM; long long* dprof;
  CK(hipMalloc(&dx, x.size() + 256)); CK(hipMalloc(&dxp, xp.size())); CK(hipMalloc(&dw2, wdw2.size()));
  CK(hipMalloc(&dimg, ref.size())); CK(hipMalloc(&dbias, NCH * 4)); CK(hipMalloc(&dM, NCH * 8));
  CK(hipMalloc(&dprof, (size_t)B * 2 * 8 * 4 * 8));
  CK(hipMemcpy(dx, x.data(), x.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dxp, xp.data(), xp.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dw2, wdw2.data(), wdw2.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dbias, bias.data(), NCH * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dM, M.data(), NCH * 8, hipMemcpyHostToDevice));
  const dim3 grid(B, 2), blk(512);
  const size_t smem = (size_t)TILE * NCH + 8 * GO::WREG;
  CK(hipFuncSetAttribute((const void*)k_old<K, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)k_old<K, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  auto check = [&](const char* name) {
    std::vector<int8_t> got(ref.size());
    CK(hipMemcpy(got.data(), dimg, got.size(), hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < ref.size(); ++i) bad += got[i] != ref[i];
    printf("  %-34s parity vs CPU conv + requant: %s (%zu of %zu bytes differ)\n", name, bad ? "FAIL" : "bit-exact", bad, ref.size());
  };
  printf("K = %d, 6-bit codes [0, 63], %d work-groups: 4x4x4 stage, input rows one byte per code vs 4 codes per 3 bytes\n", K, 2 * B);
  CK(hipMemset(dimg, 0x55, ref.size()));
  Times t8 = run([&](int dump) { hipLaunchKernelGGL((k_old<K, false>), grid, blk, smem, 0, dx, dw2, dbias, dM, dimg, dprof, T, lo, hi, dump); }, dprof, 2 * B, 50);
  check("one byte per code");
  CK(hipMemset(dimg, 0x55, ref.size()));
  Times t6 = run([&](int dump) { hipLaunchKernelGGL((k_old<K, true>), grid, blk, smem, 0, dxp, dw2, dbias, dM, dimg, dprof, T, lo, hi, dump); }, dprof, 2 * B, 50);
  check("4 codes per 3 bytes (unpacked in staging)");
  printf("  one byte per code          launch %7.2f us | work-group cycles until all waves done (median) %6lld\n", t8.us, t8.all);
  printf("  4 codes per 3 bytes        launch %7.2f us | work-group cycles until all waves done (median) %6lld   (%+.1f %%)\n", t6.us, t6.all,
         100.0 * ((double)t6.all / t8.all - 1.0));
  CK(hipFree(dx)); CK(hipFree(dxp)); CK(hipFree(dw2)); CK(hipFree(dimg)); CK(hipFree(dbias)); CK(hipFree(dM)); CK(hipFree(dprof));
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 32;
  printf("== depthwise stage of a 512-channel layer, %d utterances x 2 tiles of 128 frames\n", B);
  bench_k<33>(B);
  bench_k<39>(B);
  bench_k<51>(B);
  bench_k<63>(B);
  bench_k<75>(B);
  printf("== sub-byte activations (BASELINE config 3): the round-2 stage with packed input rows\n");
  bench_pack6<33>(B);
  bench_pack6<75>(B);
  return 0;
}
