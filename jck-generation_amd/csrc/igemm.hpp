// Gather-GEMM ("implicit GEMM") for every convolution-shaped product on the hot path:
//
//     out[pix m][ch] = sum_{tap t, c}  act[n, oy*sy + dy[t], ox*sx + dx[t], c] * w[ch][t*C + c]
//
//   * Conv2d k4 s2 p1        (D forward; dgrad of G's ConvTranspose):  16 taps, sy=sx=2
//   * ConvTranspose2d k4 s2 p1 (G forward; dgrad of D's Conv2d): four output-parity phases
//     (blockIdx.z), each a 2x2-tap stride-1 product - no zero insertion, no col2im atomics
//   * ConvTranspose2d k4 s1 p0 on a 1x1 input (G.conv1) and plain row-major GEMMs: 1 tap
//
// Tiling for CDNA4: 256 threads = 4 waves; the WEIGHTS are the MFMA A operand (rows = output
// channels) and the gathered activations B (cols = pixels), so a lane ends up with 4 consecutive
// channels of one pixel (one 8/16-byte NHWC store).  Both operands are staged through registers into
// K-contiguous padded LDS rows; global loads for k-step i+1 are issued before the MFMAs of step i
// (double-buffered LDS, one barrier per k-step).
//   PrecBf16: bf16 tiles, v_mfma_f32_16x16x32_bf16          (fast path)
//   PrecF32 : fp32 tiles, v_mfma_f32_16x16x4_f32 - exact fp32 products and accumulation (parity path)
// Optional epilogue: per-channel sum / sum-of-squares of the fp32 accumulators for the BatchNorm that
// follows - reduced over the 16 pixel lanes with wavefront shuffles and stored (no atomics) into a
// per-(tile, wave) slot that jck_bn_finalize sums - and tanh.
#pragma once
#include "common.hpp"

struct IgemmParams {
  const void* act;        // gathered NHWC tensor, element type T
  const void* w;          // packed weights [Z][NchPad][K]  (K contiguous, K % 64 == 0), bf16 or fp32
  void* out;              // NHWC output, element type T
  float* stats;           // [slots][2][cstat] partial sum, sumsq (or nullptr)
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
  int cstat;              // number of stats channels; stats channel = ch % cstat (cstat power of two)
  int ytiles_per_cset;    // channel tiles (blockIdx.y) that cover one set of cstat channels
  int epi;                // 0 none, 1 tanh
  int rows_are_phases;    // 1: MFMA row r = phase*4 + channel (4-channel outputs, all four parities in one tile)
  long long w_phase_stride;
  double flops;           // algorithmic FLOPs of this launch (profiling only)
};

#define IG_BK 64

template <class P, int BCH, int BPIX> struct IgemmCfg {
  static constexpr bool F32 = P::IS_F32;
  static constexpr int WCH = (BCH >= 64) ? 2 : 1;
  static constexpr int WPIX = 4 / WCH;
  static constexpr int FM = BCH / WCH / 16;
  static constexpr int FN = BPIX / WPIX / 16;
  static constexpr int WPASS = (BCH + 31) / 32;
  static constexpr int APASS = BPIX / 32;
  // bf16: unpadded 128-byte rows, 16-byte chunk index XOR ((row >> 1) & 7): conflict-free for ds_read_b128's lane groups
  // and for the 8-lane ds_write_b128 groups.  fp32: rows padded by 16 bytes.
  static constexpr int LD = F32 ? (IG_BK + 4) : IG_BK;
  static constexpr int ESZ = F32 ? 4 : 2;
  static constexpr int BUF_BYTES = (BCH + BPIX) * LD * ESZ;
  static constexpr int LDS_BYTES = 2 * BUF_BYTES + 128;
  // slots of partial statistics written by one launch = gridDim.x * gridDim.z * (gridDim.y / ytiles_per_cset) * WPIX
};

template <class P, int BCH, int BPIX, int NSUB>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmParams p) {
  typedef typename P::T T;        // activation storage type
  typedef typename P::W W;        // LDS / packed-weight element type (bf16_t or float)
  typedef IgemmCfg<P, BCH, BPIX> C;
  constexpr bool F32 = C::F32;
  constexpr int FM = C::FM, FN = C::FN, LD = C::LD;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int* toff = reinterpret_cast<int*>(smem_raw);                    // 16 ints: element offset of each tap
  int* tdyx = toff + 16;                                           // 16 ints: (dy << 16) | (dx & 0xffff)
  unsigned char* lds = smem_raw + 128;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int z = blockIdx.z;
  const int m0 = blockIdx.x * BPIX;
  const int ch0 = blockIdx.y * BCH;
  const int Cc = 1 << p.logC;
  if (tid < 16) {
    const int dyv = tid < p.ntaps ? (int)p.dy[z][tid] : 0, dxv = tid < p.ntaps ? (int)p.dx[z][tid] : 0;
    toff[tid] = (dyv * p.W + dxv) << p.logC;
    tdyx[tid] = tid < p.ntaps ? ((dyv << 16) | (dxv & 0xffff)) : (0x4000 << 16);      // padding taps never validate
  }

  // ---- per-thread gather rows ---------------------------------------------------------------------
  const int lrow = tid >> 3, unit = tid & 7;
  int rowbase[C::APASS];
  int ryx[C::APASS];                                               // (iy0 << 16) | ix0 ; rows past M get iy0 = 0x4000
  const T* actp = reinterpret_cast<const T*>(p.act);
#pragma unroll
  for (int ps = 0; ps < C::APASS; ++ps) {
    const int m = m0 + ps * 32 + lrow;
    const int n = m >> p.logOHW;
    const int rem = m & ((1 << p.logOHW) - 1);
    const int iy0 = (rem >> p.logOW) * p.sy, ix0 = (rem & ((1 << p.logOW) - 1)) * p.sx;
    rowbase[ps] = ((n * p.H + iy0) * p.W + ix0) << p.logC;
    ryx[ps] = ((m < p.M ? iy0 : 0x4000) << 16) | ix0;
  }
  // bounds test of tap t for a gathered row, evaluated at the load (no 16-tap mask loop up front)
  auto tap_ok = [&](int r, int t) -> bool {
    const int d = tdyx[t];
    const int iy = (r >> 16) + (d >> 16), ix = (r & 0xffff) + (int)(short)(d & 0xffff);
    return (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
  };
  const W* wsrc = reinterpret_cast<const W*>(p.w) + (long long)z * p.w_phase_stride;
  __syncthreads();   // toff visible

  struct Stage { Raw8<T> a[C::APASS]; Raw8<W> w[C::WPASS]; };
  Stage sA, sB;                                                    // two register stages: global loads run 2 k-steps ahead

  auto load_tiles = [&](int kc, Stage& sg) {
    const int k = kc * IG_BK + unit * 8;
#pragma unroll
    for (int ps = 0; ps < C::WPASS; ++ps) {
      const int r = ps * 32 + lrow;
      if (BCH >= 32 || r < BCH) ldraw(wsrc + (long long)(ch0 + r) * p.K + k, sg.w[ps]);
    }
#pragma unroll
    for (int ps = 0; ps < C::APASS; ++ps) {
      if constexpr (NSUB == 1) {
        const int t = k >> p.logC, c = k & (Cc - 1);
        zero_raw(sg.a[ps]);
        if (tap_ok(ryx[ps], t)) ldraw(actp + (rowbase[ps] + toff[t] + c), sg.a[ps]);
      } else {   // C == 4: the 8-element unit spans two taps (pixels)
        const int t0 = k >> 2, t1 = t0 + 1;
        zero_raw(sg.a[ps]);
        if (tap_ok(ryx[ps], t0)) ldraw_half(actp + (rowbase[ps] + toff[t0]), sg.a[ps], 0);
        if (tap_ok(ryx[ps], t1)) ldraw_half(actp + (rowbase[ps] + toff[t1]), sg.a[ps], 1);
      }
    }
  };

  // LDS column (elements) of this thread's 8-element unit; rows ps*32 + lrow share ((row >> 1) & 7) because 32 % 16 == 0
  const int wcol = F32 ? unit * 8 : ((unit ^ ((lrow >> 1) & 7)) * 8);
  auto store_tiles = [&](int buf, const Stage& sg) {
    W* wt = reinterpret_cast<W*>(lds + buf * C::BUF_BYTES);      // [BCH][LD]
    W* at = wt + BCH * LD;                                        // [BPIX][LD]
#pragma unroll
    for (int ps = 0; ps < C::WPASS; ++ps) {
      const int r = ps * 32 + lrow;
      if (BCH >= 32 || r < BCH) straw(wt + r * LD + wcol, sg.w[ps]);
    }
#pragma unroll
    for (int ps = 0; ps < C::APASS; ++ps) straw(at + (ps * 32 + lrow) * LD + wcol, sg.a[ps]);
  };

  const int wch = (C::WCH == 2) ? (wave >> 1) : 0;
  const int wpix = (C::WCH == 2) ? (wave & 1) : wave;
  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / IG_BK;
  const int sw = ((lane & 15) >> 1) & 7;                           // read-side swizzle of this lane's rows
  auto compute = [&](int buf) {
    const W* wt0 = reinterpret_cast<const W*>(lds + buf * C::BUF_BYTES);
    const W* at0 = wt0 + BCH * LD;
    if constexpr (!F32) {
      const bf16_t* wt = wt0 + (wch * FM * 16 + (lane & 15)) * LD;
      const bf16_t* at = at0 + (wpix * FN * 16 + (lane & 15)) * LD;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int col = (((lane >> 4) + ks * 4) ^ sw) * 8;
        bf16x8 a[FM], b[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) a[i] = lds_frag(wt + i * 16 * LD + col);
#pragma unroll
        for (int j = 0; j < FN; ++j) b[j] = lds_frag(at + j * 16 * LD + col);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
      }
    } else {
      // v_mfma_f32_16x16x4_f32: lane l holds A[row l&15][k = l>>4], B[k = l>>4][col l&15]
      const float* wt = wt0 + (wch * FM * 16 + (lane & 15)) * LD + (lane >> 4);
      const float* at = at0 + (wpix * FN * 16 + (lane & 15)) * LD + (lane >> 4);
#pragma unroll 4
      for (int kk = 0; kk < IG_BK / 4; ++kk) {
        float a[FM], b[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) a[i] = wt[i * 16 * LD + kk * 4];
#pragma unroll
        for (int j = 0; j < FN; ++j) b[j] = at[j * 16 * LD + kk * 4];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
  };

  // software pipeline: LDS double buffer + two register stages (k-step kc computes from LDS while the loads of kc+2
  // are in flight and kc+1 waits in registers for its LDS slot)
  load_tiles(0, sA);
  store_tiles(0, sA);
  if (nk > 1) load_tiles(1, sA);
  __syncthreads();
  for (int kc = 0; kc < nk; kc += 2) {
    if (kc + 2 < nk) load_tiles(kc + 2, sB);
    compute(0);
    if (kc + 1 < nk) store_tiles(1, sA);
    __syncthreads();
    if (kc + 1 < nk) {
      if (kc + 3 < nk) load_tiles(kc + 3, sA);
      compute(1);
      if (kc + 2 < nk) store_tiles(0, sB);
      __syncthreads();
    }
  }

  // ---- epilogue --------------------------------------------------------------------------------------
  if (p.stats) {
    // slot = one (pixel tile, phase, channel-set replica, pixel-wave); every (slot, channel) is written exactly once
    const int yrep = blockIdx.y / p.ytiles_per_cset, nyrep = gridDim.y / p.ytiles_per_cset;
    const long long slot = (((long long)z * gridDim.x + blockIdx.x) * nyrep + yrep) * C::WPIX + wpix;
    float* sp = p.stats + slot * 2 * p.cstat;
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
          const int cc = ch & (p.cstat - 1);
          *reinterpret_cast<f32x4*>(sp + cc) = f32x4{s[0], s[1], s[2], s[3]};
          *reinterpret_cast<f32x4*>(sp + p.cstat + cc) = f32x4{q[0], q[1], q[2], q[3]};
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
    const long long off0 = (long long)n * p.osN + (long long)(rem >> p.logOW) * p.osY +
                           (long long)(rem & ((1 << p.logOW) - 1)) * p.osX;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      int ch = ch0 + wch * FM * 16 + i * 16 + (lane >> 4) * 4;
      if (ch >= p.NchStore) continue;
      long long off = off0 + p.obase[z];
      if (p.rows_are_phases) { off = off0 + p.obase[ch >> 2]; ch = 0; }
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (p.epi == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]);
      }
      st4(outp + off + ch, v);
    }
  }
}
