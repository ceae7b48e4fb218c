// Mel front-end on gfx950: AudioToMelSpectrogramPreprocessor / FilterbankFeatures.forward
// (nemo/collections/asr/parts/features.py:334-397, normalize_batch :53-67), float32 throughout.
//   k_mel   : work-group = 16 consecutive frames of one utterance, one wavefront per STFT frame at a time — the
//             pre-emphasised, reflect-padded signal stretch, window, twiddles and compact mel table staged once in LDS;
//             hann(320) window centred in 512, 512-point real FFT (256-point complex radix-4 in LDS + even/odd split,
//             butterflies and twiddles in float64), power, 64x257 mel projection, log(x + 2^-24);
//             writes un-normalised log-mel [B][n_mels][T_pad] as 64-byte row segments
//   k_norm  : one wavefront per (utterance, mel bin) row — mean / unbiased std over the valid frames,
//             (x - mean) / (std + 1e-5), zero beyond seq_len and in the pad_to padding
// Float parity with the reference is tolerance based (FFT / reduction order).  Round 1 ran the FFT in float32 with
// sincospif twiddles and landed at 2e-3 max / 2e-5 mean on the normalised log-mel: per stage, the spectrum was the whole
// error (an exact spectrum followed by the reference's float32 steps is within 5.5e-5 max / 4e-7 mean of the fixture).
// gfx950 issues f64 adds / fmas at the rate of its integer ALU ops, so the transform now runs in float64 and the result
// is rounded once to the float32 spectrum torch.stft would ideally return; everything after it (magnitude, square, mel
// dot, log, normalisation) follows features.py in float32.  tests/test_gpu_model.py: <= 1e-4.
#include "qasr_device.h"
#include "qasr_internal.h"

namespace qasr {

#define NFFT 512
#define NBIN 257
#define HOP 160
#define WIN 320
#define WOFF 96      /* (512 - 320) / 2: torch.stft centres the window inside n_fft */


// y[i] of the pre-emphasised, reflect-padded signal (features.py:347-348; torch.stft center=True, pad_mode='reflect')
#define MEL_FBMAX 768                      /* mel weights kept in LDS, runs padded to 4 (QuartzNet's 64 x 257 matrix: ~510 non-zero, <= 702 padded) */
#define MEL_MAXM 128                       /* ... for at most this many filters; larger banks are read from global memory */

// Plan (depends on the filterbank only; qasr_frontend_plan): the mel matrix is sparse — each triangular filter covers a
// short run of FFT bins — so k_melrange finds the non-zero run [lo, hi) of every filter and k_melpack lays the runs'
// weights out back to back (each padded with zero weights to a multiple of 4) in a compact table that k_mel copies into
// LDS.  Skipped terms are fma(0, P, acc) = acc, so the float result is identical to the dense product.
// Workspace: int hdr[4] = {magic, n_mels, table length or MEL_FBMAX + 1 when it does not fit, 0}; int ranges[n_mels][2];
// int offs[n_mels]; float table[MEL_FBMAX]; (16-byte aligned) double2 twiddle[512].
#define MEL_MAGIC 0x4d454c32
__global__ void __launch_bounds__(256) k_melrange(const float* __restrict__ fb, int n_mels, int* __restrict__ ranges) {
  const int lane = threadIdx.x & 63, m = blockIdx.x * 4 + (threadIdx.x >> 6);   // one wave per filter, coalesced row read
  if (m >= n_mels) return;
  int lo = NBIN, hi = 0;
  for (int k = lane; k < NBIN; k += 64)
    if (fb[(size_t)m * NBIN + k] != 0.f) {
      lo = min(lo, k);
      hi = max(hi, k + 1);
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = min(lo, __shfl_xor(lo, o));
    hi = max(hi, __shfl_xor(hi, o));
  }
  if (lane == 0) {
    ranges[2 * m] = min(lo, hi);
    ranges[2 * m + 1] = hi;
  }
}

__global__ void __launch_bounds__(256) k_melpack(const float* __restrict__ fb, int n_mels, int* __restrict__ hdr,
                                                 const int* __restrict__ ranges, int* __restrict__ offs,
                                                 float* __restrict__ table) {
  __shared__ int total, soff[MEL_MAXM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int k = tid; k < MEL_FBMAX; k += 256) table[k] = 0.f;
  if (wave == 0) {                         // exclusive scan of the padded run lengths
    int carry = 0;
    for (int mb = 0; mb < n_mels; mb += 64) {
      const int m = mb + lane, len = m < n_mels ? (ranges[2 * m + 1] - ranges[2 * m] + 3) & ~3 : 0;
      int inc = len;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(inc, o);
        if (lane >= o) inc += up;
      }
      if (m < n_mels) offs[m] = carry + inc - len;
      if (m < min(n_mels, MEL_MAXM)) soff[m] = carry + inc - len;
      carry += __shfl(inc, 63);
    }
    if (lane == 0) {
      total = (n_mels <= MEL_MAXM && carry <= MEL_FBMAX) ? carry : MEL_FBMAX + 1;
      hdr[0] = MEL_MAGIC, hdr[1] = n_mels, hdr[2] = total, hdr[3] = 0;
    }
  }
  __syncthreads();                         // (also orders this block's global writes of offs / the zero fill for itself)
  if (total > MEL_FBMAX) return;
  for (int m = wave; m < n_mels; m += 4) {
    const int lo = ranges[2 * m], hi = ranges[2 * m + 1], off = soff[m];
    for (int k = lo + lane; k < hi; k += 64) table[off + k - lo] = fb[(size_t)m * NBIN + k];
  }
}

// every wave owns its FFT buffers, so stages are separated by a wave-level fence instead of a work-group barrier:
// DS operations of one wave execute in issue order, the fence only stops the compiler from reordering them
__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifndef MEL_CUT
#define MEL_CUT 0   /* profiles/microbench/mel_cut.py: bit mask of phases left out (timing only) */
#endif
#define MEL_FPW 4                          /* frames per wave */
#define MEL_FR (4 * MEL_FPW)               /* consecutive frames of one utterance per work-group (4 waves) */
static_assert(MEL_FR == QASR_MEL_TILE, "k_stem combines k_mel's statistics per tile of QASR_MEL_TILE frames");
#define MEL_NS (HOP * (MEL_FR - 1) + WIN)  /* signal samples under those frames' (centred, 320-tap) windows */
#define MEL_NQ ((MEL_NS + 255) / 256)

__device__ __forceinline__ int rev4_256(int k) {          // reverse the four base-4 digits of k < 256
  return ((k & 3) << 6) | ((k & 12) << 2) | ((k & 48) >> 2) | ((k & 192) >> 6);
}
__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// e^{-2 pi i k / 512}, k < 512, correctly rounded from sincospi in float64: written once per call into the workspace
__global__ void __launch_bounds__(256) k_twiddle(double2* __restrict__ tw) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < NFFT) {
    double sn, cs;
    sincospi(-2.0 * (double)k / (double)NFFT, &sn, &cs);
    tw[k] = make_double2(cs, sn);
  }
}

// radix-4 butterfly without twiddles: b0..b3 = DFT4(a0..a3) with e^{-2 pi i /4} = -i
__device__ __forceinline__ void bfly4(const double2 (&a)[4], double2 (&o)[4]) {
  const double2 s02 = make_double2(a[0].x + a[2].x, a[0].y + a[2].y), d02 = make_double2(a[0].x - a[2].x, a[0].y - a[2].y);
  const double2 s13 = make_double2(a[1].x + a[3].x, a[1].y + a[3].y), d13 = make_double2(a[1].x - a[3].x, a[1].y - a[3].y);
  o[0] = make_double2(s02.x + s13.x, s02.y + s13.y);
  o[2] = make_double2(s02.x - s13.x, s02.y - s13.y);
  o[1] = make_double2(d02.x + d13.y, d02.y - d13.x);          // d02 - i d13
  o[3] = make_double2(d02.x - d13.y, d02.y + d13.x);          // d02 + i d13
}
// 16-byte unit u of a wave's FFT buffer lives at unit u + u / 4: with one unit of padding after every four, every
// access pattern below (strides 64, 16, 4, 1 units, the digit-reversed stores and the mirrored split reads) spreads
// eight neighbouring lanes over eight different bank groups (profiles/microbench/mel_cut.py has the per-phase cost).
__device__ __forceinline__ int zpad(int u) { return u + (u >> 2); }
#define MEL_ZLEN (256 + 64)

// Work-group = 16 consecutive STFT frames of one utterance, one wave per frame at a time (4 frames per wave).  The
// pre-emphasised, reflect-padded signal stretch of those frames is staged ONCE (coalesced), as are the window and the
// compact mel table; every lane keeps the twiddles of its butterflies in registers for all its frames; results leave
// as 64-byte rows (16 frames of one mel bin).
// The 512-point real FFT is computed as a 256-point complex FFT of z[n] = v[2n] + i v[2n+1] (radix-4 decimation in
// frequency: 4 stages, one butterfly per lane and stage — the first straight from the framing registers, the last
// without twiddles and stored in natural order) followed by the usual even/odd split
// X[k] = E[k] + e^{-2 pi i k/512} O[k],  k = 0..256.
__global__ void __launch_bounds__(256, 4) k_mel(const float* __restrict__ audio, int B, int S, const float* __restrict__ fb,
                                             const float* __restrict__ window, const int* __restrict__ hdr,
                                             const int* __restrict__ ranges, const int* __restrict__ goffs,
                                             const float* __restrict__ table, const double2* __restrict__ twg, int n_mels,
                                             float preemph, int n_frames, int T_pad, float* __restrict__ out,
                                             const int32_t* __restrict__ audio_lens, double* __restrict__ stats,
                                             int32_t* __restrict__ feat_lens) {
  __shared__ double2 zb[4][MEL_ZLEN];      // per wave: complex work buffer (padded, zpad)
  __shared__ float pw[4][NBIN + 3];        // per wave: power spectrum, [257..259] = 0
  __shared__ __attribute__((aligned(8))) float ys[MEL_NS];   // pre-emphasised signal under the windows, [0] <-> signal index 160 t0 - 160
  __shared__ __attribute__((aligned(8))) float win[WIN];
  __shared__ __attribute__((aligned(16))) float fbv[MEL_FBMAX];   // non-zero mel weights, filter after filter, runs padded to 4
  __shared__ short rng[MEL_MAXM][3];       // lo, hi, offset of the run in fbv
  // 40 KB in all: four work-groups (16 waves) per CU, and 32 utterances x 32 groups of 16 frames fill 256 CUs exactly once
  static_assert(sizeof(zb) + sizeof(pw) + sizeof(ys) + sizeof(win) + sizeof(fbv) + sizeof(rng) <= 40960, "k_mel: LDS budget");
  float (*ob)[MEL_FR + 1] = reinterpret_cast<float (*)[MEL_FR + 1]>(&zb[0][0]);   // log-mel results of one pass (after the FFTs)
  static_assert(sizeof(float) * 64 * (MEL_FR + 1) <= sizeof(zb), "ob aliases zb");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, t0 = blockIdx.x * MEL_FR;
  const float* x = audio + (size_t)b * S;
  // ---- staging: every global load of the work-group leaves before the first use ----
  const bool fb_lds = hdr[2] <= MEL_FBMAX;                   // (plan: n_mels <= MEL_MAXM and the table fits)
  float yv[MEL_NQ], ym[MEL_NQ];
  int yi[MEL_NQ];
#pragma unroll
  for (int q = 0; q < MEL_NQ; ++q) {       // reflect padding (features.py:350-352) folded into the index
    const int i = HOP * t0 - HOP + min(tid + 256 * q, MEL_NS - 1);
    yi[q] = min(max(i < 0 ? -i : (i >= S ? 2 * (S - 1) - i : i), 0), S - 1);
    yv[q] = x[yi[q]];
    ym[q] = x[max(yi[q] - 1, 0)];
  }
  const float4 tv = reinterpret_cast<const float4*>(table)[min(tid, MEL_FBMAX / 4 - 1)];
  const float wv0 = window[tid], wv1 = window[min(tid + 256, WIN - 1)];
  const int mm = min(tid, n_mels - 1);
  const int r_lo = ranges[2 * mm], r_hi = ranges[2 * mm + 1], r_off = goffs[mm];
  // this lane's twiddles: stage st (block length 256 >> 2 st) multiplies outputs 1..3 by W^j, W^2j, W^3j with
  // W = e^{-2 pi i / L}, j = lane mod L/4, i.e. entries (512 / L) j {1, 2, 3} of the 512-entry table; split: entry k
  double2 w[3][3], wk[4];
#pragma unroll
  for (int st = 0; st < 3; ++st) {
    const int ts = (2 << (2 * st)) * (lane & ((64 >> (2 * st)) - 1));
#pragma unroll
    for (int i = 0; i < 3; ++i) w[st][i] = twg[(i + 1) * ts];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) wk[i] = twg[lane + 64 * i];
#pragma unroll
  for (int q = 0; q < MEL_NQ; ++q) asm volatile("" : "+v"(yv[q]), "+v"(ym[q]));   // both loads unconditional, all in flight
#pragma unroll
  for (int q = 0; q < MEL_NQ; ++q)         // pre-emphasis in two float32 steps, like features.py:347-348
    if (tid + 256 * q < MEL_NS) ys[tid + 256 * q] = yi[q] > 0 ? yv[q] - mul_f32_unfused(preemph, ym[q]) : yv[q];
  if (tid < MEL_FBMAX / 4) reinterpret_cast<float4*>(fbv)[tid] = tv;
  win[tid] = wv0;
  if (tid + 256 < WIN) win[tid + 256] = wv1;
  if (tid < MEL_MAXM) rng[tid][0] = (short)r_lo, rng[tid][1] = (short)r_hi, rng[tid][2] = (short)min(r_off, MEL_FBMAX);
  if (lane < 3) pw[wave][NBIN + lane] = 0.f;
  const int r3 = ((lane & 3) << 4) | (lane & 12) | (lane >> 4);   // natural index of output 4 lane + i of the last stage: 64 i + r3
  __syncthreads();
  double2* z = zb[wave];
  float* P = pw[wave];
  for (int m0 = 0; m0 < n_mels; m0 += 64) {                  // (one pass for n_mels <= 64; the FFTs are redone otherwise)
    float res[MEL_FPW] = {};
#pragma unroll
    for (int it = 0; it < ((MEL_CUT & 16) ? 0 : MEL_FPW); ++it) {
      res[it] = 0.f;
      const int fl = MEL_FPW * wave + it, t = t0 + fl;       // frame within the group / the utterance
      const bool ok = t < n_frames;
      double2 a[4], o[4];
      // framing: sample j of frame t is ypad[160 t + j]; z[n] = v[2n] + i v[2n+1], n = lane + 64 i; windowed samples are
      // float32 products; outside the centred 320-tap window the frame is zero
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = 2 * (lane + 64 * i);
        const bool in = j >= WOFF && j < WOFF + WIN;
        const int jc = in ? j : WOFF;
        const float2 y = *reinterpret_cast<const float2*>(&ys[HOP * fl + jc - WOFF]);
        const float2 wv = *reinterpret_cast<const float2*>(&win[jc - WOFF]);
        a[i] = in && !(MEL_CUT & 1) ? make_double2((double)__fmul_rn(y.x, wv.x), (double)__fmul_rn(y.y, wv.y)) : make_double2(0.0, 0.0);
      }
      // stage 0 (span 64) in registers
      bfly4(a, o);
      wave_sync_lds();                     // previous frame's split reads are done
      z[zpad(lane)] = o[0];
#pragma unroll
      for (int i = 1; i < 4; ++i) z[zpad(lane + 64 * i)] = cmul(o[i], w[0][i - 1]);
      wave_sync_lds();
      // stages 1, 2: spans 16, 4
#pragma unroll
      for (int st = 1; st < ((MEL_CUT & 2) ? 1 : 3); ++st) {
        const int L = 256 >> (2 * st), q4 = L >> 2;            // block length, quarter
        const int base = (lane / q4) * L + (lane & (q4 - 1));
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = z[zpad(base + q4 * i)];
        bfly4(a, o);
        z[zpad(base)] = o[0];
#pragma unroll
        for (int i = 1; i < 4; ++i) z[zpad(base + q4 * i)] = cmul(o[i], w[st][i - 1]);
        wave_sync_lds();
      }
      // stage 3: span 1, no twiddles; output 4 lane + i is X[64 i + r3], stored in natural order
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = z[zpad(4 * lane + i)];
      bfly4(a, o);
      wave_sync_lds();
#pragma unroll
      for (int i = 0; i < 4; ++i) z[zpad(64 * i + r3)] = o[i];
      wave_sync_lds();
      // even/odd split and power spectrum: the reference takes sqrt(re^2+im^2) and then pow(2) (features.py:356-360)
#pragma unroll
      for (int i = 0; i < ((MEL_CUT & 4) ? 0 : 4); ++i) {
        const int k = lane + 64 * i;
        const double2 zk = z[zpad(k)], zc = z[zpad((256 - k) & 255)];
        const double2 E = make_double2(0.5 * (zk.x + zc.x), 0.5 * (zk.y - zc.y));
        const double2 O = make_double2(0.5 * (zk.y + zc.y), -0.5 * (zk.x - zc.x));   // -i (zk - conj(zc)) / 2
        // the float32 spectrum, then features.py:356-360 in float32: sqrt(re^2 + im^2), pow(2)
        const float re = (float)(E.x + O.x * wk[i].x - O.y * wk[i].y), im = (float)(E.y + O.x * wk[i].y + O.y * wk[i].x);
        const float mag = sqrtf(mul_f32_unfused(re, re) + mul_f32_unfused(im, im));   // products rounded on their own
        P[k] = mag * mag;
        if (i == 0 && lane == 0) {           // X[256] = Re z[0] - Im z[0] (twiddle -1), imaginary part 0
          const float re_n = (float)(zk.x - zk.y);
          const float mag_n = sqrtf(mul_f32_unfused(re_n, re_n) + mul_f32_unfused(0.f, 0.f));
          P[NFFT / 2] = mag_n * mag_n;
        }
      }
      wave_sync_lds();
      // mel projection + log (features.py:363-368); lane = mel bin; terms in ascending k like the round-1 kernel
      const int m = m0 + lane;
      if (m < n_mels) {
        float acc = 0.f;
        if (fb_lds) {
          const int lo = rng[m][0], hi = (MEL_CUT & 8) ? lo + 1 : rng[m][1];
          const float* f = fbv + rng[m][2];                  // run padded with zero weights to a multiple of 4
          for (int k = lo; k < hi; k += 4) {
            const float4 fw = *reinterpret_cast<const float4*>(f + (k - lo));
            const float p0 = P[k], p1 = P[k + 1], p2 = P[k + 2], p3 = P[k + 3];   // <= P[259]: finite (zero tail)
            acc = fmaf(fw.x, p0, acc);
            acc = fmaf(fw.y, p1, acc);
            acc = fmaf(fw.z, p2, acc);
            acc = fmaf(fw.w, p3, acc);
          }
        } else {
          const int lo = ranges[2 * m], hi = ranges[2 * m + 1];
          const float* f = fb + (size_t)m * NBIN;
          for (int k = lo; k < hi; ++k) acc = fmaf(f[k], P[k], acc);
        }
        res[it] = ok ? logf(acc + 5.9604644775390625e-08f) : 0.f;   // 2^-24
      }
    }
    __syncthreads();                       // every wave is done with its FFT buffer: ob may overwrite them
#pragma unroll
    for (int it = 0; it < MEL_FPW; ++it) ob[lane][MEL_FPW * wave + it] = res[it];
    __syncthreads();
    // 16 consecutive frames of a mel bin = one 64-byte row segment
    for (int i = tid; i < 64 * MEL_FR; i += 256) {
      const int ml = i / MEL_FR, fl = i - ml * MEL_FR;
      if (m0 + ml < n_mels && t0 + fl < n_frames) out[((size_t)b * n_mels + m0 + ml) * T_pad + t0 + fl] = ob[ml][fl];
    }
    // normalisation fused into the consumer (k_stem): this tile's share of normalize_batch's statistics per mel bin
    // (features.py:53-67) - sum and sum of squared deviations from the tile mean over the valid frames, float64
    if (stats && tid < 64 && m0 + tid < n_mels) {
      const int seq = (audio_lens[b] + HOP - 1) / HOP;
      const int nv = max(0, min(min(seq, n_frames) - t0, MEL_FR));
      double sum = 0.0, m2 = 0.0;
      for (int fl = 0; fl < nv; ++fl) sum += (double)ob[tid][fl];
      const double mu = nv ? sum / (double)nv : 0.0;
      for (int fl = 0; fl < nv; ++fl) {
        const double d = (double)ob[tid][fl] - mu;
        m2 += d * d;
      }
      double* st = stats + (((size_t)b * gridDim.x + blockIdx.x) * n_mels + m0 + tid) * 2;
      st[0] = sum;
      st[1] = m2;
    }
    if (feat_lens && blockIdx.x == 0 && tid == 0) feat_lens[b] = (audio_lens[b] + HOP - 1) / HOP;   // get_seq_len (features.py:327-328)
    __syncthreads();
  }
}

#define NORM_NV 16
__global__ void __launch_bounds__(64) k_norm(float* __restrict__ feats, const int32_t* __restrict__ audio_lens, int n_mels,
                                             int n_frames, int T_pad, int32_t* __restrict__ feat_lens) {
  const int row = blockIdx.x, lane = threadIdx.x;
  const int b = row / n_mels;
  const int alen = audio_lens[b];
  const int seq = (alen + HOP - 1) / HOP;              // get_seq_len: ceil(len / hop) (features.py:327-328)
  if (lane == 0 && row % n_mels == 0) feat_lens[b] = seq;
  float* x = feats + (size_t)row * T_pad;
  const int n = min(seq, n_frames);
  // sums in float64 (order-independent to float32 accuracy), results rounded to the float32 mean / std torch returns
  if (T_pad <= 64 * NORM_NV) {             // the row fits the wave's registers: one read, one write
    float v[NORM_NV];
#pragma unroll
    for (int q = 0; q < NORM_NV; ++q) v[q] = x[min(lane + 64 * q, T_pad - 1)];
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < NORM_NV; ++q) s += lane + 64 * q < n ? (double)v[q] : 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = (float)(s / (double)n);
    double var = 0.0;
#pragma unroll
    for (int q = 0; q < NORM_NV; ++q) {
      const double d = (double)v[q] - (double)mean;
      var += lane + 64 * q < n ? d * d : 0.0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o);
    const float sd = (float)sqrt(var / (double)(n - 1)) + 1e-5f;   // torch.std (unbiased) + CONSTANT (features.py:63-65)
#pragma unroll
    for (int q = 0; q < NORM_NV; ++q) {
      const int t = lane + 64 * q;
      if (t < T_pad) x[t] = t < n ? (v[q] - mean) / sd : 0.f;
    }
    return;
  }
  double s = 0.0;
  for (int t = lane; t < n; t += 64) s += (double)x[t];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = (float)(s / (double)n);
  double v = 0.0;
  for (int t = lane; t < n; t += 64) {
    const double d = (double)x[t] - (double)mean;
    v += d * d;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const float sd = (float)sqrt(v / (double)(n - 1)) + 1e-5f;  // torch.std (unbiased) + CONSTANT (features.py:63-65)
  for (int t = lane; t < T_pad; t += 64) x[t] = (t < n) ? (x[t] - mean) / sd : 0.f;
}

}  // namespace qasr

extern "C" {

int qasr_frontend_frames(int S, int pad_to) {
  int n = 1 + S / HOP;
  if (pad_to > 0 && n % pad_to) n += pad_to - n % pad_to;
  return n;
}

// workspace layout (see k_melrange): hdr[4], ranges[n_mels][2], offs[n_mels], table[MEL_FBMAX], twiddle[512]
static size_t ws_ranges(int) { return 4 * sizeof(int); }
static size_t ws_offs(int n_mels) { return ws_ranges(n_mels) + (size_t)n_mels * 2 * sizeof(int); }
static size_t ws_table(int n_mels) { return (ws_offs(n_mels) + (size_t)n_mels * sizeof(int) + 15) / 16 * 16; }
static size_t ws_tw(int n_mels) { return ws_table(n_mels) + MEL_FBMAX * sizeof(float); }
size_t qasr_frontend_workspace_bytes(int, int, int n_mels) { return ws_tw(n_mels) + NFFT * sizeof(double2); }

int qasr_frontend_plan(void* stream, const float* fb, int n_mels, void* workspace, size_t workspace_bytes) {
  if (!fb || n_mels <= 0 || !workspace || workspace_bytes < qasr_frontend_workspace_bytes(0, 0, n_mels) || ((size_t)workspace & 15))
    return QASR_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  int* ranges = (int*)(ws + ws_ranges(n_mels));
  hipLaunchKernelGGL(qasr::k_melrange, dim3((n_mels + 3) / 4), dim3(256), 0, s, fb, n_mels, ranges);
  hipLaunchKernelGGL(qasr::k_melpack, dim3(1), dim3(256), 0, s, fb, n_mels, (int*)ws, ranges, (int*)(ws + ws_offs(n_mels)),
                     (float*)(ws + ws_table(n_mels)));
  hipLaunchKernelGGL(qasr::k_twiddle, dim3(NFFT / 256), dim3(256), 0, s, (double2*)(ws + ws_tw(n_mels)));
  return hipGetLastError() == hipSuccess ? QASR_OK : QASR_ERR_HIP;
}

int qasr_frontend_mel_planned(void* stream, const float* audio, const int32_t* audio_lens, int B, int S, const float* fb,
                              const float* window, int n_mels, float preemph, int pad_to, float* feats,
                              int32_t* feat_lens, const void* workspace, size_t workspace_bytes) {
  if (!audio || !audio_lens || !fb || !window || !feats || !feat_lens || B <= 0 || S <= NFFT / 2 || n_mels <= 0 ||
      !workspace || workspace_bytes < qasr_frontend_workspace_bytes(B, S, n_mels) || ((size_t)workspace & 15))
    return QASR_ERR_ARG;
  const int n_frames = 1 + S / HOP;
  const int T_pad = qasr_frontend_frames(S, pad_to);
  hipStream_t s = (hipStream_t)stream;
  const char* ws = (const char*)workspace;
  hipLaunchKernelGGL(qasr::k_mel, dim3((n_frames + MEL_FR - 1) / MEL_FR, B), dim3(256), 0, s, audio, B, S, fb, window,
                     (const int*)ws, (const int*)(ws + ws_ranges(n_mels)), (const int*)(ws + ws_offs(n_mels)),
                     (const float*)(ws + ws_table(n_mels)), (const double2*)(ws + ws_tw(n_mels)), n_mels, preemph, n_frames,
                     T_pad, feats, nullptr, nullptr, nullptr);
  hipLaunchKernelGGL(qasr::k_norm, dim3(B * n_mels), dim3(64), 0, s, feats, audio_lens, n_mels, n_frames, T_pad,
                     feat_lens);
  return hipGetLastError() == hipSuccess ? QASR_OK : QASR_ERR_HIP;
}

}  // extern "C"

namespace qasr {
size_t frontend_stats_bytes(int B, int S, int n_mels) {
  const int n_frames = 1 + S / HOP;
  return (size_t)B * ((n_frames + MEL_FR - 1) / MEL_FR) * n_mels * 2 * sizeof(double);
}
// k_mel alone: un-normalised log-mel into `feats`, feat_lens, and per (utterance, 16-frame tile, mel bin) the partial
// statistics k_stem combines into normalize_batch's mean / std (the engine's forward_audio with the fused stem)
int frontend_mel_stats(hipStream_t s, const float* audio, const int32_t* audio_lens, int B, int S, const float* fb,
                       const float* window, int n_mels, float preemph, int pad_to, float* feats, int32_t* feat_lens,
                       const void* workspace, size_t workspace_bytes, double* stats, int* n_tiles, int* n_frames_out) {
  if (!audio || !audio_lens || !fb || !window || !feats || !feat_lens || !stats || B <= 0 || S <= NFFT / 2 || n_mels <= 0 ||
      !workspace || workspace_bytes < qasr_frontend_workspace_bytes(B, S, n_mels) || ((size_t)workspace & 15))
    return QASR_ERR_ARG;
  const int n_frames = 1 + S / HOP;
  const int T_pad = qasr_frontend_frames(S, pad_to);
  const char* ws = (const char*)workspace;
  const int nt = (n_frames + MEL_FR - 1) / MEL_FR;
  hipLaunchKernelGGL(k_mel, dim3(nt, B), dim3(256), 0, s, audio, B, S, fb, window, (const int*)ws,
                     (const int*)(ws + ws_ranges(n_mels)), (const int*)(ws + ws_offs(n_mels)),
                     (const float*)(ws + ws_table(n_mels)), (const double2*)(ws + ws_tw(n_mels)), n_mels, preemph, n_frames,
                     T_pad, feats, audio_lens, stats, feat_lens);
  *n_tiles = nt;
  *n_frames_out = n_frames;
  return hipGetLastError() == hipSuccess ? QASR_OK : QASR_ERR_HIP;
}
}  // namespace qasr

extern "C" {

int qasr_frontend_mel(void* stream, const float* audio, const int32_t* audio_lens, int B, int S, const float* fb,
                      const float* window, int n_mels, float preemph, int pad_to, float* feats, int32_t* feat_lens,
                      void* workspace, size_t workspace_bytes) {
  if (!audio || !audio_lens || !window || !feats || !feat_lens || B <= 0 || S <= NFFT / 2) return QASR_ERR_ARG;
  const int rc = qasr_frontend_plan(stream, fb, n_mels, workspace, workspace_bytes);
  return rc != QASR_OK ? rc : qasr_frontend_mel_planned(stream, audio, audio_lens, B, S, fb, window, n_mels, preemph, pad_to,
                                                        feats, feat_lens, workspace, workspace_bytes);
}
}
