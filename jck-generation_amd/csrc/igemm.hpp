// Gather-GEMM ("implicit GEMM") for every convolution-shaped product on the hot path:
//
//     out[pix m][ch] = sum_{tap t, c}  act[n, oy*sy + dy[t], ox*sx + dx[t], c] * w[ch][t*C + c]
//
//   * Conv2d k4 s2 p1        (D forward; dgrad of G's ConvTranspose):  16 taps, sy=sx=2
//   * ConvTranspose2d k4 s2 p1 (G forward; dgrad of D's Conv2d): four output-parity phases
//     (blockIdx.z), each a 2x2-tap stride-1 product - no zero insertion, no col2im atomics
//   * ConvTranspose2d k4 s1 p0 on a 1x1 input (G.conv1) and plain row-major GEMMs: 1 tap
//
// Tiling for CDNA4: 256 threads = 4 waves, v_mfma_f32_16x16x32_bf16 with the WEIGHTS as the MFMA
// A operand (rows = output channels) and the gathered activations as B (cols = pixels), so a lane
// ends up with 4 consecutive channels of one pixel (one 8/16-byte NHWC store).  Both operands are
// staged through registers into K-contiguous LDS rows padded by 16 B; global loads for k-step i+1
// are issued before the MFMAs of step i (double-buffered LDS, one barrier per k-step).
// Optional epilogue: per-channel sum / sum-of-squares for the BatchNorm that follows (wavefront
// shuffles over the 16 pixel lanes, then one atomic per channel per wave) and tanh.
#pragma once
#include "common.hpp"

struct IgemmParams {
  const void* act;        // gathered NHWC tensor, element type T
  const bf16_t* w_hi;     // packed weights [Z][NchPad][K]  (K contiguous, K % 64 == 0)
  const bf16_t* w_lo;     // low halves (PrecF32 only)
  void* out;              // NHWC output, element type T
  float* stats;           // [2][cstat] sum, sumsq (or nullptr)
  int M;                  // pixel rows per phase
  int NchStore;           // channels physically stored per output pixel (multiple of 4)
  int K;                  // ntaps << logC
  int logC;               // gathered channels per pixel (power of two >= 4)
  int H, W;               // gathered tensor spatial size
  int logOW, logOHW;      // row m -> n = m >> logOHW, oy = (m >> logOW) & .., ox = m & (OW-1)
  int sy, sx;
  int ntaps;
  signed char dy[4][16], dx[4][16];
  long long osN;          // output offset = n*osN + oy*osY + ox*osX + obase[z]   (elements)
  int osY, osX;
  int obase[4];
  int cstat_mask;         // stats channel = ch & cstat_mask
  int cstat;              // number of stats channels
  int epi;                // 0 none, 1 tanh
  long long w_phase_stride;
  double flops;           // algorithmic FLOPs of this launch (profiling only)
};

#define IG_BK 64
#define IG_LD 72          // padded LDS row (elements)

template <class P, int BCH, int BPIX> struct IgemmCfg {
  static constexpr int WCH = (BCH >= 64) ? 2 : 1;
  static constexpr int WPIX = 4 / WCH;
  static constexpr int FM = BCH / WCH / 16;
  static constexpr int FN = BPIX / WPIX / 16;
  static constexpr int NPL = P::NPLANE;
  static constexpr int WPASS = (BCH + 31) / 32;
  static constexpr int APASS = BPIX / 32;
  static constexpr int BUF_ELEMS = NPL * (BCH + BPIX) * IG_LD;
  static constexpr int LDS_BYTES = 2 * BUF_ELEMS * 2 + 64;
};

template <class P, int BCH, int BPIX, int NSUB>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmParams p) {
  typedef typename P::T T;
  typedef IgemmCfg<P, BCH, BPIX> C;
  constexpr int NPL = C::NPL, FM = C::FM, FN = C::FN;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int* toff = reinterpret_cast<int*>(smem_raw);                    // 16 ints
  bf16_t* lds = reinterpret_cast<bf16_t*>(smem_raw + 64);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int z = blockIdx.z;
  const int m0 = blockIdx.x * BPIX;
  const int ch0 = blockIdx.y * BCH;
  const int Cc = 1 << p.logC;
  if (tid < 16) toff[tid] = (tid < p.ntaps) ? (((int)p.dy[z][tid] * p.W + (int)p.dx[z][tid]) << p.logC) : 0;

  // ---- per-thread gather rows ---------------------------------------------------------------------
  const int lrow = tid >> 3, unit = tid & 7;
  int rowbase[C::APASS];
  unsigned rmask[C::APASS];
  const T* actp = reinterpret_cast<const T*>(p.act);
#pragma unroll
  for (int ps = 0; ps < C::APASS; ++ps) {
    const int m = m0 + ps * 32 + lrow;
    const int n = m >> p.logOHW;
    const int rem = m & ((1 << p.logOHW) - 1);
    const int iy0 = (rem >> p.logOW) * p.sy, ix0 = (rem & ((1 << p.logOW) - 1)) * p.sx;
    rowbase[ps] = ((n * p.H + iy0) * p.W + ix0) << p.logC;
    unsigned mk = 0;
    if (m < p.M) {
      for (int t = 0; t < p.ntaps; ++t) {
        const int iy = iy0 + p.dy[z][t], ix = ix0 + p.dx[z][t];
        if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) mk |= 1u << t;
      }
    }
    rmask[ps] = mk;
  }
  const bf16_t* whi = p.w_hi + (long long)z * p.w_phase_stride;
  const bf16_t* wlo = (NPL == 2) ? p.w_lo + (long long)z * p.w_phase_stride : nullptr;
  __syncthreads();   // toff visible

  Raw8<T> areg[C::APASS];
  u32x4 wreg[NPL][C::WPASS];

  auto load_tiles = [&](int kc) {
    const int k = kc * IG_BK + unit * 8;
#pragma unroll
    for (int ps = 0; ps < C::WPASS; ++ps) {
      const int r = ps * 32 + lrow;
      if (BCH >= 32 || r < BCH) {
        const long long o = (long long)(ch0 + r) * p.K + k;
        wreg[0][ps] = *reinterpret_cast<const u32x4*>(whi + o);
        if (NPL == 2) wreg[NPL - 1][ps] = *reinterpret_cast<const u32x4*>(wlo + o);
      }
    }
#pragma unroll
    for (int ps = 0; ps < C::APASS; ++ps) {
      if constexpr (NSUB == 1) {
        const int t = k >> p.logC, c = k & (Cc - 1);
        const bool ok = (rmask[ps] >> t) & 1u;
        const T* src = actp + (rowbase[ps] + toff[t] + c);
        if constexpr (sizeof(T) == 2) {
          u32x4 v = {0u, 0u, 0u, 0u};
          if (ok) v = *reinterpret_cast<const u32x4*>(src);
          areg[ps].v = v;
        } else {
          f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
          if (ok) { a = *reinterpret_cast<const f32x4*>(src); b = *reinterpret_cast<const f32x4*>(src + 4); }
          areg[ps].a = a; areg[ps].b = b;
        }
      } else {   // C == 4: the 8-element unit spans two taps (pixels)
        const int t0 = k >> 2, t1 = t0 + 1;
        const bool ok0 = (rmask[ps] >> t0) & 1u, ok1 = (rmask[ps] >> t1) & 1u;
        const T* s0 = actp + (rowbase[ps] + toff[t0]);
        const T* s1 = actp + (rowbase[ps] + toff[t1]);
        if constexpr (sizeof(T) == 2) {
          u32x2 a = {0u, 0u}, b = {0u, 0u};
          if (ok0) a = *reinterpret_cast<const u32x2*>(s0);
          if (ok1) b = *reinterpret_cast<const u32x2*>(s1);
          u32x4 v = {a[0], a[1], b[0], b[1]};
          areg[ps].v = v;
        } else {
          f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
          if (ok0) a = *reinterpret_cast<const f32x4*>(s0);
          if (ok1) b = *reinterpret_cast<const f32x4*>(s1);
          areg[ps].a = a; areg[ps].b = b;
        }
      }
    }
  };

  auto store_tiles = [&](int buf) {
    bf16_t* base = lds + buf * C::BUF_ELEMS;
    bf16_t* wt = base;                                   // [NPL][BCH][LD]
    bf16_t* at = base + NPL * BCH * IG_LD;               // [NPL][BPIX][LD]
#pragma unroll
    for (int ps = 0; ps < C::WPASS; ++ps) {
      const int r = ps * 32 + lrow;
      if (BCH >= 32 || r < BCH) {
        *reinterpret_cast<u32x4*>(wt + r * IG_LD + unit * 8) = wreg[0][ps];
        if (NPL == 2) *reinterpret_cast<u32x4*>(wt + (BCH + r) * IG_LD + unit * 8) = wreg[NPL - 1][ps];
      }
    }
#pragma unroll
    for (int ps = 0; ps < C::APASS; ++ps) {
      const int r = ps * 32 + lrow;
      if constexpr (sizeof(T) == 2) {
        *reinterpret_cast<u32x4*>(at + r * IG_LD + unit * 8) = areg[ps].v;
      } else {
        float f[8] = {areg[ps].a[0], areg[ps].a[1], areg[ps].a[2], areg[ps].a[3],
                      areg[ps].b[0], areg[ps].b[1], areg[ps].b[2], areg[ps].b[3]};
        u32x4 hi, lo;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          bf16_t h0, l0, h1, l1;
          split_bf(f[2 * i], h0, l0);
          split_bf(f[2 * i + 1], h1, l1);
          hi[i] = (unsigned)h0 | ((unsigned)h1 << 16);
          lo[i] = (unsigned)l0 | ((unsigned)l1 << 16);
        }
        *reinterpret_cast<u32x4*>(at + r * IG_LD + unit * 8) = hi;
        *reinterpret_cast<u32x4*>(at + (BPIX + r) * IG_LD + unit * 8) = lo;
      }
    }
  };

  const int wch = (C::WCH == 2) ? (wave >> 1) : 0;
  const int wpix = (C::WCH == 2) ? (wave & 1) : wave;
  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / IG_BK;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  for (int kc = 0; kc < nk; ++kc) {
    const bool more = kc + 1 < nk;
    if (more) load_tiles(kc + 1);
    const bf16_t* base = lds + (kc & 1) * C::BUF_ELEMS;
    const bf16_t* wt = base + (wch * FM * 16 + (lane & 15)) * IG_LD + (lane >> 4) * 8;
    const bf16_t* at = base + NPL * BCH * IG_LD + (wpix * FN * 16 + (lane & 15)) * IG_LD + (lane >> 4) * 8;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[NPL][FM], b[NPL][FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        a[0][i] = lds_frag(wt + i * 16 * IG_LD + ks * 32);
        if (NPL == 2) a[NPL - 1][i] = lds_frag(wt + (BCH + i * 16) * IG_LD + ks * 32);
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        b[0][j] = lds_frag(at + j * 16 * IG_LD + ks * 32);
        if (NPL == 2) b[NPL - 1][j] = lds_frag(at + (BPIX + j * 16) * IG_LD + ks * 32);
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          if (NPL == 2) {   // small terms first
            acc[i][j] = mfma16(a[NPL - 1][i], b[0][j], acc[i][j]);
            acc[i][j] = mfma16(a[0][i], b[NPL - 1][j], acc[i][j]);
          }
          acc[i][j] = mfma16(a[0][i], b[0][j], acc[i][j]);
        }
    }
    if (more) store_tiles((kc + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue --------------------------------------------------------------------------------------
  if (p.stats) {
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float v = acc[i][j][r]; s[r] += v; q[r] += v * v; }
#pragma unroll
      for (int r = 0; r < 4; ++r) { s[r] = row16_sum(s[r]); q[r] = row16_sum(q[r]); }
      if ((lane & 15) == 0) {
        const int ch = ch0 + wch * FM * 16 + i * 16 + (lane >> 4) * 4;
        if (ch < p.NchStore) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            atomicAdd(p.stats + ((ch + r) & p.cstat_mask), s[r]);
            atomicAdd(p.stats + p.cstat + ((ch + r) & p.cstat_mask), q[r]);
          }
        }
      }
    }
  }
  T* outp = reinterpret_cast<T*>(p.out);
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int m = m0 + wpix * FN * 16 + j * 16 + (lane & 15);
    if (m >= p.M) continue;
    const int n = m >> p.logOHW;
    const int rem = m & ((1 << p.logOHW) - 1);
    const long long off = (long long)n * p.osN + (long long)(rem >> p.logOW) * p.osY +
                          (long long)(rem & ((1 << p.logOW) - 1)) * p.osX + p.obase[z];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int ch = ch0 + wch * FM * 16 + i * 16 + (lane >> 4) * 4;
      if (ch >= p.NchStore) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (p.epi == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]);
      }
      st4(outp + off + ch, v);
    }
  }
}
