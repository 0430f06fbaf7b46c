// Weight-gradient product for every convolution-shaped layer:
//
//   dW[cs][t*Cb + cb] = sum_{pixel m} S[m][cs] * Big[n, oy*sy + dy[t], ox*sx + dx[t], cb]
//
// S is the small-image side (D: grad of the conv output; G: the ConvTranspose input), Big the
// large-image side (D: the conv input; G: grad of the ConvTranspose output) - one kernel serves
// Conv2d and ConvTranspose2d because both store weights as [C_small][C_big][kh][kw].
//
// The contraction runs over the PIXEL index, which is the slow index of both NHWC operands, so both
// MFMA operands need K along LDS rows: tiles are stored [pixel][channel] exactly as they arrive from
// HBM (16-byte coalesced loads) and read with ds_read_b64_tr_b16, CDNA4's transposing LDS read.
// MFMA A = gathered side (rows = (tap, cb) columns of dW), B = S side (cols = cs): a lane then owns
// 4 consecutive (tap, cb) entries of one cs row -> one float4 store into a split-K partial slab
// part[z][cs][ncols].  `wgrad_reduce_kernel` sums the slabs and transposes into the PyTorch weight
// layout [cs][cb][kh*4+kw] (deterministic; no float atomics).
#pragma once
#include "common.hpp"

struct WgradParams {
  const void* sside;      // [Mtot][CsStride] T
  const void* big;        // NHWC [N][H][W][Cb] T
  float* part;            // [Z][CsRows][ncols]
  int Mtot;               // pixel rows
  int CsStride;           // elements per S row
  int CsRows;             // rows of the slab (>= gridDim.y * BS)
  int ncols;              // ntaps << logCb (>= gridDim.x * BG)
  int logCb;
  int H, W;
  int logOW, logOHW;
  int sy, sx;
  int ntaps;
  signed char dy[16], dx[16];
  int mchunk;             // pixel rows per blockIdx.z (multiple of WG_BKP)
  double flops;           // algorithmic FLOPs of this launch (profiling only)
};

#define WG_BKP 32

template <class P, int BG, int BS, int NSUB>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
  typedef typename P::T T;
  constexpr int NPL = P::NPLANE;
  constexpr int LDG = BG + 16, LDS_ = BS + 16;                 // padded rows (elements), 32 B pad
  constexpr int WG_ = (BG >= 64) ? 2 : 1, WS_ = 4 / WG_;
  constexpr int FM = BG / WG_ / 16, FN = BS / WS_ / 16;
  constexpr int UG = BG / 8, RG = 256 / UG, PG = (WG_BKP + RG - 1) / RG;   // loader geometry, G tile
  constexpr int US = BS / 8, RS = 256 / US, PS = (WG_BKP + RS - 1) / RS;
  constexpr int GT = NPL * WG_BKP * LDG, ST = NPL * WG_BKP * LDS_;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* lds = reinterpret_cast<bf16_t*>(smem_raw);              // 2 x (G tile, S tile)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g0 = blockIdx.x * BG, s0 = blockIdx.y * BS;
  const int mz0 = blockIdx.z * p.mchunk;
  const int mz1 = min(mz0 + p.mchunk, p.Mtot);
  const T* bigp = reinterpret_cast<const T*>(p.big);
  const T* sp = reinterpret_cast<const T*>(p.sside);
  const int Cb = 1 << p.logCb;

  // fixed per-thread column unit of the gathered side
  const int gu = tid % UG, gr = tid / UG;
  const int gcol = g0 + gu * 8;
  int t0, t1, cb0;
  if constexpr (NSUB == 1) { t0 = gcol >> p.logCb; t1 = t0; cb0 = gcol & (Cb - 1); }
  else { t0 = gcol >> 2; t1 = t0 + 1; cb0 = 0; }
  const bool tv0 = t0 < p.ntaps, tv1 = t1 < p.ntaps;
  const int dy0 = tv0 ? p.dy[t0] : 0, dx0 = tv0 ? p.dx[t0] : 0;
  const int dy1 = tv1 ? p.dy[t1] : 0, dx1 = tv1 ? p.dx[t1] : 0;
  const int su = tid % US, sr = tid / US;

  Raw8<T> greg[PG], sreg[PS];

  auto load_tiles = [&](int mbase) {
#pragma unroll
    for (int ps = 0; ps < PG; ++ps) {
      const int r = ps * RG + gr;
      const int m = mbase + r;
      const bool rok = (r < WG_BKP) && (m < mz1);
      const int n = m >> p.logOHW;
      const int rem = m & ((1 << p.logOHW) - 1);
      const int iy0 = (rem >> p.logOW) * p.sy, ix0 = (rem & ((1 << p.logOW) - 1)) * p.sx;
      if constexpr (NSUB == 1) {
        const int iy = iy0 + dy0, ix = ix0 + dx0;
        const bool ok = rok && tv0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const T* src = bigp + ((((long long)n * p.H + iy) * p.W + ix) << p.logCb) + cb0;
        if constexpr (sizeof(T) == 2) {
          u32x4 v = {0u, 0u, 0u, 0u};
          if (ok) v = *reinterpret_cast<const u32x4*>(src);
          greg[ps].v = v;
        } else {
          f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
          if (ok) { a = *reinterpret_cast<const f32x4*>(src); b = *reinterpret_cast<const f32x4*>(src + 4); }
          greg[ps].a = a; greg[ps].b = b;
        }
      } else {
        const int iya = iy0 + dy0, ixa = ix0 + dx0, iyb = iy0 + dy1, ixb = ix0 + dx1;
        const bool oka = rok && tv0 && (unsigned)iya < (unsigned)p.H && (unsigned)ixa < (unsigned)p.W;
        const bool okb = rok && tv1 && (unsigned)iyb < (unsigned)p.H && (unsigned)ixb < (unsigned)p.W;
        const T* sa = bigp + ((((long long)n * p.H + iya) * p.W + ixa) << 2);
        const T* sb = bigp + ((((long long)n * p.H + iyb) * p.W + ixb) << 2);
        if constexpr (sizeof(T) == 2) {
          u32x2 a = {0u, 0u}, b = {0u, 0u};
          if (oka) a = *reinterpret_cast<const u32x2*>(sa);
          if (okb) b = *reinterpret_cast<const u32x2*>(sb);
          u32x4 v = {a[0], a[1], b[0], b[1]};
          greg[ps].v = v;
        } else {
          f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
          if (oka) a = *reinterpret_cast<const f32x4*>(sa);
          if (okb) b = *reinterpret_cast<const f32x4*>(sb);
          greg[ps].a = a; greg[ps].b = b;
        }
      }
    }
#pragma unroll
    for (int ps = 0; ps < PS; ++ps) {
      const int r = ps * RS + sr;
      const int m = mbase + r;
      const bool ok = (r < WG_BKP) && (m < mz1) && (s0 + su * 8 < p.CsStride);
      const T* src = sp + (long long)m * p.CsStride + s0 + su * 8;
      if constexpr (sizeof(T) == 2) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (ok) v = *reinterpret_cast<const u32x4*>(src);
        sreg[ps].v = v;
      } else {
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        if (ok) { a = *reinterpret_cast<const f32x4*>(src); b = *reinterpret_cast<const f32x4*>(src + 4); }
        sreg[ps].a = a; sreg[ps].b = b;
      }
    }
  };

  auto put = [&](bf16_t* tile, int ld, int rows_total, int r, int u, const Raw8<T>& rg) {
    if constexpr (sizeof(T) == 2) {
      *reinterpret_cast<u32x4*>(tile + r * ld + u * 8) = rg.v;
    } else {
      float f[8] = {rg.a[0], rg.a[1], rg.a[2], rg.a[3], rg.b[0], rg.b[1], rg.b[2], rg.b[3]};
      u32x4 hi, lo;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bf16_t h0, l0, h1, l1;
        split_bf(f[2 * i], h0, l0);
        split_bf(f[2 * i + 1], h1, l1);
        hi[i] = (unsigned)h0 | ((unsigned)h1 << 16);
        lo[i] = (unsigned)l0 | ((unsigned)l1 << 16);
      }
      *reinterpret_cast<u32x4*>(tile + r * ld + u * 8) = hi;
      *reinterpret_cast<u32x4*>(tile + (rows_total + r) * ld + u * 8) = lo;
    }
  };

  auto store_tiles = [&](int buf) {
    bf16_t* gt = lds + buf * (GT + ST);
    bf16_t* st = gt + GT;
#pragma unroll
    for (int ps = 0; ps < PG; ++ps) {
      const int r = ps * RG + gr;
      if (r < WG_BKP) put(gt, LDG, WG_BKP, r, gu, greg[ps]);
    }
#pragma unroll
    for (int ps = 0; ps < PS; ++ps) {
      const int r = ps * RS + sr;
      if (r < WG_BKP) put(st, LDS_, WG_BKP, r, su, sreg[ps]);
    }
  };

  const int wg = (WG_ == 2) ? (wave >> 1) : 0;
  const int ws = (WG_ == 2) ? (wave & 1) : wave;
  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (mz1 > mz0) ? (mz1 - mz0 + WG_BKP - 1) / WG_BKP : 0;
  if (nk > 0) {
    load_tiles(mz0);
    store_tiles(0);
  }
  __syncthreads();
  // transposed-read lane addressing: group g = lane>>4 covers pixels 8g..8g+7 of the 32-pixel step;
  // lane i = lane&15 supplies &tile[8g + (i>>2)][col0 + 4*(i&3)] and receives column col0 + i.
  const int trow = (lane >> 4) * 8 + ((lane & 15) >> 2), tcol = (lane & 3) * 4;
  for (int kc = 0; kc < nk; ++kc) {
    const bool more = kc + 1 < nk;
    if (more) load_tiles(mz0 + (kc + 1) * WG_BKP);
    const bf16_t* gt = lds + (kc & 1) * (GT + ST);
    const bf16_t* st = gt + GT;
    bf16x8 a[NPL][FM], b[NPL][FN];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const bf16_t* q = gt + (pl * WG_BKP + trow) * LDG + wg * FM * 16 + i * 16 + tcol;
        a[pl][i] = join_tr(lds_tr4(q), lds_tr4(q + 4 * LDG));
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const bf16_t* q = st + (pl * WG_BKP + trow) * LDS_ + ws * FN * 16 + j * 16 + tcol;
        b[pl][j] = join_tr(lds_tr4(q), lds_tr4(q + 4 * LDS_));
      }
    }
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        if constexpr (NPL == 2) {
          acc[i][j] = mfma16(a[1][i], b[0][j], acc[i][j]);
          acc[i][j] = mfma16(a[0][i], b[1][j], acc[i][j]);
        }
        acc[i][j] = mfma16(a[0][i], b[0][j], acc[i][j]);
      }
    if (more) store_tiles((kc + 1) & 1);
    __syncthreads();
  }

  float* part = p.part + (long long)blockIdx.z * p.CsRows * p.ncols;
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int cs = s0 + ws * FN * 16 + j * 16 + (lane & 15);
    if (cs >= p.CsRows) continue;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int col = g0 + wg * FM * 16 + i * 16 + (lane >> 4) * 4;
      if (col >= p.ncols) continue;
      *reinterpret_cast<f32x4*>(part + (long long)cs * p.ncols + col) = acc[i][j];
    }
  }
}

// grad[cs][cb][t] (+)= sum_z part[z][cs][t*CbPad + cb]      (cs < Cs, cb < Cb)
static __global__ void wgrad_reduce_kernel(const float* __restrict__ part, int Z, int CsRows, int ncols, int Cs, int Cb,
                                    int logCbPad, int ntaps, float* __restrict__ grad, int accumulate) {
  const long long total = (long long)Cs * Cb * ntaps;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(i % ntaps);
    const long long r = i / ntaps;
    const int cb = (int)(r % Cb);
    const int cs = (int)(r / Cb);
    const long long src = (long long)cs * ncols + ((long long)t << logCbPad) + cb;
    float s = 0.f;
    for (int z = 0; z < Z; ++z) s += part[(long long)z * CsRows * ncols + src];
    grad[i] = accumulate ? grad[i] + s : s;
  }
}
