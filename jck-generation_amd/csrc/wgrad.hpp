// Weight-gradient product for every convolution-shaped layer:
//
//   dW[cs][t*Cb + cb] = sum_{pixel m} S[m][cs] * Big[n, oy*sy + dy[t], ox*sx + dx[t], cb]
//
// S is the small-image side (D: grad of the conv output; G: the ConvTranspose input), Big the
// large-image side (D: the conv input; G: grad of the ConvTranspose output) - one kernel serves
// Conv2d and ConvTranspose2d because both store weights as [C_small][C_big][kh][kw].
//
// The contraction runs over the PIXEL index, the slow index of both NHWC operands.  Tiles are stored
// [pixel][channel] exactly as they arrive from HBM (16-byte coalesced loads):
//   PrecBf16: fragments are read with ds_read_b64_tr_b16, CDNA4's transposing LDS read, and feed
//             v_mfma_f32_16x16x32_bf16
//   PrecF32 : v_mfma_f32_16x16x4_f32 takes ONE element per lane (A[row l&15][k l>>4]), which is a plain
//             row read of the [pixel][channel] tile - exact fp32
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
  int big_row_elems;      // > 0: the gathered side is a plain [Mtot][big_row_elems] matrix (Linear layers), 1 tap
  int gx, gy, gz;         // logical grid of the LDS-DMA kernel (launched 1-D): column tiles, row tiles, pixel chunks
  unsigned big_bytes, s_bytes;   // sizes of the two operands (buffer descriptors of the LDS-DMA kernels; 0: use the register-staged kernel)
  int loader_prio;        // wave-specialised kernels: s_setprio of the loader waves (jck_tune "wgrad_prio")
  double flops;           // algorithmic FLOPs of this launch (profiling only)
};

#define WG_BKP 32

template <class P, int BG, int BS> struct WgradCfg {
  static constexpr bool F32 = P::IS_F32;
  static constexpr int PAD = F32 ? 4 : 16;
  static constexpr int LDG = BG + PAD, LDSS = BS + PAD;            // padded rows (elements)
  static constexpr int ESZ = F32 ? 4 : 2;
  static constexpr int BUF_BYTES = WG_BKP * (LDG + LDSS) * ESZ;
  static constexpr int LDS_BYTES = 2 * BUF_BYTES;
};

template <class P, int BG, int BS, int NSUB>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
  typedef typename P::T T;
  typedef WgradCfg<P, BG, BS> C;
  constexpr bool F32 = C::F32;
  constexpr int LDG = C::LDG, LDSS = C::LDSS;
  constexpr int WG_ = (BG >= 64) ? 2 : 1, WS_ = 4 / WG_;
  constexpr int FM = BG / WG_ / 16, FN = BS / WS_ / 16;
  constexpr int UG = BG / 8, RG = 256 / UG, PG = (WG_BKP + RG - 1) / RG;   // loader geometry, G tile
  constexpr int US = BS / 8, RS = 256 / US, PS = (WG_BKP + RS - 1) / RS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

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
      zero_raw(greg[ps]);
      if constexpr (NSUB == 1) {
        const int iy = iy0 + dy0, ix = ix0 + dx0;
        if (p.big_row_elems) {
          if (rok && gcol < p.big_row_elems) ldraw(bigp + (long long)m * p.big_row_elems + gcol, greg[ps]);
        } else if (rok && tv0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
          ldraw(bigp + ((((long long)n * p.H + iy) * p.W + ix) << p.logCb) + cb0, greg[ps]);
      } else {
        const int iya = iy0 + dy0, ixa = ix0 + dx0, iyb = iy0 + dy1, ixb = ix0 + dx1;
        if (rok && tv0 && (unsigned)iya < (unsigned)p.H && (unsigned)ixa < (unsigned)p.W)
          ldraw_half(bigp + ((((long long)n * p.H + iya) * p.W + ixa) << 2), greg[ps], 0);
        if (rok && tv1 && (unsigned)iyb < (unsigned)p.H && (unsigned)ixb < (unsigned)p.W)
          ldraw_half(bigp + ((((long long)n * p.H + iyb) * p.W + ixb) << 2), greg[ps], 1);
      }
    }
#pragma unroll
    for (int ps = 0; ps < PS; ++ps) {
      const int r = ps * RS + sr;
      const int m = mbase + r;
      zero_raw(sreg[ps]);
      if ((r < WG_BKP) && (m < mz1) && (s0 + su * 8 < p.CsStride)) ldraw(sp + (long long)m * p.CsStride + s0 + su * 8, sreg[ps]);
    }
  };

  auto store_tiles = [&](int buf) {
    T* gt = reinterpret_cast<T*>(smem_raw + buf * C::BUF_BYTES);   // [WG_BKP][LDG]
    T* st = gt + WG_BKP * LDG;                                     // [WG_BKP][LDSS]
#pragma unroll
    for (int ps = 0; ps < PG; ++ps) {
      const int r = ps * RG + gr;
      if (r < WG_BKP) straw(gt + r * LDG + gu * 8, greg[ps]);
    }
#pragma unroll
    for (int ps = 0; ps < PS; ++ps) {
      const int r = ps * RS + sr;
      if (r < WG_BKP) straw(st + r * LDSS + su * 8, sreg[ps]);
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
  // bf16 transposed-read lane addressing: group g = lane>>4 covers pixels 8g..8g+7 of the 32-pixel step;
  // lane i = lane&15 supplies &tile[8g + (i>>2)][col0 + 4*(i&3)] and receives column col0 + i.
  const int trow = (lane >> 4) * 8 + ((lane & 15) >> 2), tcol = (lane & 3) * 4;
  for (int kc = 0; kc < nk; ++kc) {
    const bool more = kc + 1 < nk;
    if (more) load_tiles(mz0 + (kc + 1) * WG_BKP);
    const T* gt = reinterpret_cast<const T*>(smem_raw + (kc & 1) * C::BUF_BYTES);
    const T* st = gt + WG_BKP * LDG;
    if constexpr (!F32) {
      bf16x8 a[FM], b[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const bf16_t* q = gt + trow * LDG + wg * FM * 16 + i * 16 + tcol;
        a[i] = join_tr(lds_tr4(q), lds_tr4(q + 4 * LDG));
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const bf16_t* q = st + trow * LDSS + ws * FN * 16 + j * 16 + tcol;
        b[j] = join_tr(lds_tr4(q), lds_tr4(q + 4 * LDSS));
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
    } else {
      const float* ga = gt + (lane >> 4) * LDG + wg * FM * 16 + (lane & 15);
      const float* sb = st + (lane >> 4) * LDSS + ws * FN * 16 + (lane & 15);
#pragma unroll 4
      for (int kk = 0; kk < WG_BKP / 4; ++kk) {
        float a[FM], b[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) a[i] = ga[kk * 4 * LDG + i * 16];
#pragma unroll
        for (int j = 0; j < FN; ++j) b[j] = sb[kk * 4 * LDSS + j * 16];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
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
// generic fallback (image layers with CbPad = 4, head): one thread per output element
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

// image layers (CbPad = 4, ncols = 64): one workgroup per cs row; 16 float4 columns x 16 slab lanes (every thread keeps
// Z/16 independent 16-byte loads in flight), LDS reduce over the slab lanes
static __global__ __launch_bounds__(256) void wgrad_reduce_img_kernel(const float* __restrict__ part, int Z, int CsRows, int Cb,
                                                                      float* __restrict__ grad, int accumulate) {
  __shared__ float red[16][64];
  const int cs = blockIdx.x, c4 = (threadIdx.x & 15) * 4, zl = threadIdx.x >> 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int z = zl; z < Z; z += 16) s += *reinterpret_cast<const f32x4*>(part + ((long long)z * CsRows + cs) * 64 + c4);
#pragma unroll
  for (int k = 0; k < 4; ++k) red[zl][c4 + k] = s[k];
  __syncthreads();
  if (threadIdx.x < 64) {
    const int col = threadIdx.x, t = col >> 2, cb = col & 3;
    if (cb < Cb) {
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) v += red[k][col];
      float* d = grad + ((long long)cs * Cb + cb) * 16 + t;
      *d = accumulate ? *d + v : v;
    }
  }
}

// 16-tap layers with Cb % 64 == 0: one workgroup per (cs, 64-channel chunk).  Reads are 256-byte runs along cb
// (float4 per thread, summed over the Z slabs in registers), the [16][64] -> [64][16] transpose goes through LDS,
// the 4 KB result is written as one contiguous run.
// ZG groups of 256 threads split the Z slabs between them (group q takes slabs q, q+ZG, ...: ZG times the loads in flight - with
// 32-64 slabs and only Cs * Cb/64 workgroups the kernel was pure load latency, 42 us for D.conv2) and are summed in group order
// through LDS: the summation order depends on (Z, ZG) only.
template <int ZG>
static __global__ __launch_bounds__(256 * ZG) void wgrad_reduce16_kernel(const float* __restrict__ part, int Z, int CsRows, int ncols,
                                                                         int Cb, int logCbPad, float* __restrict__ grad,
                                                                         int accumulate) {
  __shared__ float tile[64][17];
  __shared__ f32x4 zsum[ZG > 1 ? ZG - 1 : 1][256];
  const int cs = blockIdx.y, cb0 = blockIdx.x * 64;
  const int tid = threadIdx.x & 255, zq = threadIdx.x >> 8;
  const int t = tid >> 4, c4 = (tid & 15) * 4;
  const float* src = part + (long long)cs * ncols + ((long long)t << logCbPad) + cb0 + c4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int z = zq; z < Z; z += ZG) s += *reinterpret_cast<const f32x4*>(src + (long long)z * CsRows * ncols);
  if constexpr (ZG > 1) {
    if (zq > 0) zsum[zq - 1][tid] = s;
    __syncthreads();
    if (zq == 0) {
#pragma unroll
      for (int q = 0; q < ZG - 1; ++q) s += zsum[q][tid];
    }
  }
  if (zq == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) tile[c4 + k][t] = s[k];
  }
  __syncthreads();
  if (zq != 0) return;
  const int cbl = tid >> 2, t4 = (tid & 3) * 4;
  float* dst = grad + ((long long)cs * Cb + cb0 + cbl) * 16 + t4;
  f32x4 o = {tile[cbl][t4], tile[cbl][t4 + 1], tile[cbl][t4 + 2], tile[cbl][t4 + 3]};
  if (accumulate) o += *reinterpret_cast<const f32x4*>(dst);
  *reinterpret_cast<f32x4*>(dst) = o;
}

// ------------------------------------------------------------------------------------------------------------------
// LDS-DMA variant of the 128x128 bf16 weight-gradient tile: both operand tiles ([64 pixels][128 channels], 256-byte rows)
// go global -> LDS with global_load_lds_dwordx4 (1 KiB = 4 pixel rows per wave-instruction), two LDS stages of 32 KB
// (several workgroups per CU hide each other's latency), raw s_barrier + s_waitcnt vmcnt(0) per 64-pixel k-step.
// Rows are unpadded; the 16-byte chunk index is XORed with ((row & 3) << 2) | ((row >> 2) & 3) on the SOURCE side and
// in the ds_read_b64_tr_b16 address (the guide's 256-byte-row image that serves transposed reads without conflicts).
// Out-of-image taps / pixels past the split read a zero page.  Requires Cb >= 64 (a 128-column tile spans <= 2 taps).
// ------------------------------------------------------------------------------------------------------------------
static __device__ __attribute__((aligned(16))) unsigned int g_jck_zero_page_w[64];

#define WGD_BKP 64
// NW = 4 or 8 waves per workgroup.  The fill rate of a CU scales with the number of waves that issue LDS-DMA (an issuing wave
// stalls ~0.1 us per 1 KiB piece and cannot feed the MFMA meanwhile): 8 waves each issue half the pieces and own a 64x32
// part of the tile, without the extra split-K slabs that a second 4-wave workgroup per CU would cost.
// STAMP (development): per-wave s_memtime totals of the three parts of a k-step - [0] wait + barrier, [1] DMA issue,
// [2] LDS reads + MFMA - and [3] the whole kernel, read back with jck_debug_wgrad_stamps
static __device__ unsigned long long g_wgd_stamps[1024 * 8 * 4];
// WS (wave-specialised, 8 waves, NW = 4, NSTG = 3): the stamps show a 4-wave workgroup spending 38 % of a k-step stalled in the
// issue of its 8 LDS-DMA pieces (the fill path pushes back at ~35 B/clk/CU - its hardware rate) and 45 % in LDS reads + MFMA,
// one after the other in each wave's instruction stream, while waiting for data takes 2 %.  With WS waves 4-7 only issue the
// DMA (two stages ahead) and waves 0-3 only read fragments and feed the MFMA, one loader and one consumer per SIMD, so the
// two halves of a k-step overlap; one s_barrier per k-step hands a landed stage over and frees the stage read last.
// GT (wave-specialised form only): gathered-side tiles of 128 columns per workgroup.  GT = 2 gives a 256 x 128 output tile:
// 8 consumer waves (4 per gathered tile) + 4 loader waves = 768 threads, three 48 KB stages.  The S tile is filled once for
// twice the columns, 87 instead of 64 FLOP per filled byte - the kernels are bound by the LDS fill rate (~24 B/clk/CU), so
// bytes per FLOP is what sets their speed.
template <int NSTG, int NW, bool STAMP = false, bool WS = false, int GT = 1, bool PIPE = false, int WDBG = 0>
static __global__ __launch_bounds__(WS ? (4 + 4 * GT) * 64 : NW * 64) void wgrad_dma_kernel(const WgradParams p) {
  static_assert(!PIPE || (WS && !STAMP), "the software-pipelined consumer exists in the wave-specialised form");
  static_assert(!WS || (NW == 4 && NSTG == 3), "wave specialisation: 4 loader + 4*GT consumer waves, 3 LDS stages");
  static_assert(GT == 1 || (WS && !STAMP && GT == 2), "the 256-column tile exists in the wave-specialised form only");
  constexpr int BG = 128 * GT, BS = 128, FM = 4, FN = NW == 8 ? 2 : 4;
  constexpr int NQ = 16 / NW;                                        // DMA rounds per operand tile: 4 rows per wave and round
  constexpr int SW = 128 / (NW / 2);                                 // small-side columns per wave: 64 (4 waves) or 32 (8 waves)
  constexpr int ROWB = 256;                                          // bytes per tile row (128 bf16)
  constexpr int TILE_BYTES = WGD_BKP * ROWB;                         // 16 KB per operand tile
  constexpr int STG_BYTES = (GT + 1) * TILE_BYTES;                   // GT gathered tiles, then the S tile
  constexpr int NCW = WS ? 4 * GT : NW;                              // consumer waves
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* lds = smem_raw;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_raw = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = WS && wave_raw >= NCW;
  const int wave = loader ? wave_raw - NCW : wave_raw;               // index inside the role (loader / consumer)
  // XCD-aware order (workgroup id % 8 = XCD, each with its own 4 MB L2): every XCD takes a contiguous run of logical tiles,
  // column tile fastest, then row tile, then pixel chunk - the tiles of one pixel chunk read the same rows of both operands
  // (other taps / channel chunks), so a chunk's 1.5-4.5 MB working set is fetched into ONE L2 instead of all eight
  int wgid;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    wgid = __builtin_amdgcn_readfirstlane((xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx);
  }
  const int bx = wgid % p.gx, by = (wgid / p.gx) % p.gy, bz = wgid / (p.gx * p.gy);
  const int g0 = bx * BG, s0 = by * BS;
  const int mz0 = bz * p.mchunk;
  const int mz1 = min(mz0 + p.mchunk, p.Mtot);
  const unsigned char* bigb = reinterpret_cast<const unsigned char*>(p.big);
  const unsigned char* sb_ = reinterpret_cast<const unsigned char*>(p.sside);
  const int Cb = 1 << p.logCb;

  // this lane's place inside a wave-instruction: 4 rows x 16 chunks
  const int r4 = lane >> 4, pc = lane & 15;

  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  // Per-lane gather state of the 4 + 4 DMA pieces, advanced by 64 pixel rows per k-step instead of being re-derived: the
  // image rows of consecutive images are contiguous, so 64 more pixels are a CONSTANT byte step on both operands
  // (64/OHW images when an image has <= 64 output pixels, else 64/OW output rows = sy*64/OW input rows; H = sy*OH);
  // only the vertical bound check follows oy.  32-bit byte offsets (operands < 2 GiB).
  const int OHW = 1 << p.logOHW, OW = 1 << p.logOW, OH = OHW >> p.logOW;
  const bool small_img = OHW <= WGD_BKP;
  const unsigned binc = (unsigned)((small_img ? (WGD_BKP / OHW) * p.H * p.W : (WGD_BKP / OW) * p.sy * p.W) << p.logCb) * 2u;
  const unsigned sinc = (unsigned)(WGD_BKP * p.CsStride) * 2u;
  const int doy = small_img ? 0 : WGD_BKP / OW;
  unsigned boff[GT][NQ], soff[NQ];
  int oyq[NQ], mq[NQ], dyq[GT][NQ];
  bool bokx[GT][NQ], sokc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int row = q * (4 * NW) + wave * 4 + r4;                      // 0..63 inside the tile
    const int lc = pc ^ (((row & 3) << 2) | ((row >> 2) & 3));        // logical 16-byte chunk fetched by this lane
    const int m = mz0 + row;
    const int n = m >> p.logOHW;
    const int rem = m & (OHW - 1);
    const int oy = rem >> p.logOW, ox = rem & (OW - 1);
#pragma unroll
    for (int g = 0; g < GT; ++g) {
      // gathered side: column g0 + 128*g + lc*8 -> (tap, cb); a 128-column tile spans at most two taps (Cb >= 64)
      const int gcol0 = g0 + 128 * g;
      const int t_base = gcol0 >> p.logCb;
      const int col = (gcol0 & (Cb - 1)) + lc * 8;
      const bool second = col >= Cb;                                   // only when Cb == 64
      const int t = t_base + (second ? 1 : 0);
      const bool tv = t < p.ntaps;
      const int tc = min(t, p.ntaps - 1);
      const int dyv = p.dy[tc], dxv = p.dx[tc];
      const int cb = second ? col - Cb : col;
      const int ix = ox * p.sx + dxv;
      boff[g][q] = (unsigned)(((((n * p.H + oy * p.sy + dyv) * p.W + ix) << p.logCb) + cb) * 2);
      bokx[g][q] = tv && (unsigned)ix < (unsigned)p.W;
      dyq[g][q] = dyv;
    }
    oyq[q] = oy; mq[q] = m;
    soff[q] = (unsigned)((m * p.CsStride + s0 + lc * 8) * 2);
    sokc[q] = s0 + lc * 8 < p.CsStride;
  }
  // The pieces are buffer loads (wave-uniform descriptor + one 32-bit offset VGPR per lane; an offset past the operand returns
  // zeros - no zero page): the issue of a piece costs less than with a 64-bit address pair per lane (round 3: +4-14 %)
  const auto rs_big = make_rsrc(p.big, p.big_bytes);
  const auto rs_s = make_rsrc(p.sside, p.s_bytes);
  // WDBG (JCK_DIAG builds, timing experiments with wrong results): 1 no loads, 2 no MFMAs, 3 no slab stores, 4 no transposed reads
  auto issue = [&](int stage) {
    if constexpr (WDBG == 1) return;
    unsigned char* gt = lds + stage * STG_BYTES;
    unsigned char* st = gt + GT * TILE_BYTES;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const bool rok = mq[q] < mz1;
#pragma unroll
      for (int g = 0; g < GT; ++g) {
        const bool ok = rok && bokx[g][q] && (unsigned)(oyq[q] * p.sy + dyq[g][q]) < (unsigned)p.H;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_big, (lptr_t)(gt + g * TILE_BYTES + (q * (4 * NW) + wave * 4) * ROWB), 16,
                                                 (int)(ok ? boff[g][q] : JCK_OOB), 0, 0, 0);
        boff[g][q] += binc;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_s, (lptr_t)(st + (q * (4 * NW) + wave * 4) * ROWB), 16,
                                               (int)((rok && sokc[q]) ? soff[q] : JCK_OOB), 0, 0, 0);
      soff[q] += sinc; mq[q] += WGD_BKP;
      oyq[q] = (oyq[q] + doy) & (OH - 1);
    }
  };

  // consumer wave -> 64 x SW part of the output tile: wg = 64-column block of the gathered side (GT*2 of them), ws = S part
  const int wg = NW == 8 ? wave >> 2 : wave >> 1, ws = NW == 8 ? wave & 3 : wave & 1;
  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // transposed-read addressing: group g = lane>>4 covers pixels 8g..8g+7 of a 32-pixel sub-step; lane il = lane&15 supplies
  // &tile[row][col0 + 4*(il&3)] with row = 8g + (il>>2) (+4 for the second read) and receives column col0 + il.
  const int il = lane & 15;
  auto tr_addr = [&](const unsigned char* tile, int row, int col0) -> const bf16_t* {
    const int chunk = (col0 >> 3) + ((il & 3) >> 1);
    const int f = ((row & 3) << 2) | ((row >> 2) & 3);
    return reinterpret_cast<const bf16_t*>(tile + row * ROWB + ((chunk ^ f) << 4) + ((il & 1) << 3));
  };
  auto compute = [&](int stage) {
    const unsigned char* gt = lds + stage * STG_BYTES + (wg >> 1) * TILE_BYTES;      // this wave's gathered tile
    const unsigned char* st = lds + stage * STG_BYTES + GT * TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < WGD_BKP / 32; ++kk) {
      const int row = kk * 32 + (lane >> 4) * 8 + (il >> 2);
      bf16x8 a[FM], b[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int c0 = (wg & 1) * 64 + i * 16;
        a[i] = join_tr(lds_tr4(tr_addr(gt, row, c0)), lds_tr4(tr_addr(gt, row + 4, c0)));
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int c0 = ws * SW + j * 16;
        b[j] = join_tr(lds_tr4(tr_addr(st, row, c0)), lds_tr4(tr_addr(st, row + 4, c0)));
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
    }
  };

  const int nk = (mz1 > mz0) ? (mz1 - mz0 + WGD_BKP - 1) / WGD_BKP : 0;
  // NSTG LDS stages (2: two workgroups share a CU; 3-4: one workgroup per CU keeps 2-3 k-steps of loads in flight - the
  // split-K plan launches ~one workgroup per CU).  8 DMA pieces per stage and wave: the counted wait leaves the NSTG-2
  // youngest stages in flight; stages past the end read the zero page and are never consumed.
  unsigned long long tw = 0, ti = 0, tc = 0, t0 = 0, tk0 = 0;
  if constexpr (STAMP) tk0 = __builtin_amdgcn_s_memtime();
  if constexpr (WS) {
    if (loader) {
      if (p.loader_prio) __builtin_amdgcn_s_setprio(1);
      issue(0); issue(1);                                             // stages 0, 1 in flight
      int slot = 2;
      for (int k = 0; k < nk; ++k) {
        if constexpr (STAMP) t0 = __builtin_amdgcn_s_memtime();
        // stage k has landed (this wave's pieces): (GT + 1) * NQ = 8 or 12 younger pieces may stay in flight
        if constexpr (WDBG == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if constexpr (GT == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                 // consumers may read stage k; stage k-1 is free
        if constexpr (STAMP) { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); tw += t1 - t0; t0 = t1; }
        issue(slot);                                                  // stage k+2 (past the end: zero page, never read)
        if constexpr (STAMP) { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); ti += t1 - t0; }
        slot = slot == 2 ? 0 : slot + 1;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if constexpr (PIPE) {
      // software-pipelined consumer (round 5, as igemm_dma_persist_kernel): the transposed reads of the next 32-pixel half-step are
      // issued under the MFMAs of this one, one read per MFMA, across the k-step border too; lgkmcnt(0) in front of the barrier
      // says every read of the stage the loaders refill next has completed.  Same products in the same order: the same bits.
      const unsigned char* gt0 = lds + (wg >> 1) * TILE_BYTES;
      const unsigned char* st0 = lds + GT * TILE_BYTES;
      bf16x8 a0[FM], b0[FN], a1[FM], b1[FN];
      auto rdh = [&](bf16x8 (&a)[FM], bf16x8 (&b)[FN], int slot, int kk) __attribute__((always_inline)) {
        const unsigned char* gt = gt0 + slot * STG_BYTES;
        const unsigned char* st = st0 + slot * STG_BYTES;
        const int row = kk * 32 + (lane >> 4) * 8 + (il >> 2);
        if constexpr (WDBG == 4) {
#pragma unroll
          for (int j = 0; j < FN; ++j) asm volatile("" : "=v"(b[j]));
#pragma unroll
          for (int i = 0; i < FM; ++i) asm volatile("" : "=v"(a[i]));
          return;
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int c0 = ws * SW + j * 16;
          b[j] = join_tr(lds_tr4(tr_addr(st, row, c0)), lds_tr4(tr_addr(st, row + 4, c0)));
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          const int c0 = (wg & 1) * 64 + i * 16;
          a[i] = join_tr(lds_tr4(tr_addr(gt, row, c0)), lds_tr4(tr_addr(gt, row + 4, c0)));
        }
      };
      auto mm = [&](bf16x8 (&a)[FM], bf16x8 (&b)[FN]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) { if constexpr (WDBG == 2) asm volatile("" :: "v"(a[i]), "v"(b[j])); else acc[i][j] = mfma16(a[i], b[j], acc[i][j]); }
      };
      auto interleave = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 2 * (FM + FN); ++q) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, (FM * FN) / (2 * (FM + FN)), 0);
        }
      };
      if (nk > 0) {
        int slot = 0;
        __builtin_amdgcn_s_barrier();
        rdh(a0, b0, slot, 0);
        for (int k = 0; k < nk - 1; ++k) {
          rdh(a1, b1, slot, 1);
          mm(a0, b0);
          interleave();
          __builtin_amdgcn_sched_barrier(0);
          slot = slot == 2 ? 0 : slot + 1;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
          rdh(a0, b0, slot, 0);
          mm(a1, b1);
          interleave();
          __builtin_amdgcn_sched_barrier(0);
        }
        rdh(a1, b1, slot, 1);
        mm(a0, b0);
        interleave();
        __builtin_amdgcn_sched_barrier(0);
        mm(a1, b1);
      }
    } else {
      int slot = 0;
      for (int k = 0; k < nk; ++k) {
        if constexpr (STAMP) t0 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();
        if constexpr (STAMP) { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); tw += t1 - t0; t0 = t1; }
        compute(slot);
        if constexpr (STAMP) {
          asm volatile("s_nop 0" ::: "memory");
          const unsigned long long t1 = __builtin_amdgcn_s_memtime(); tc += t1 - t0;
        }
        slot = slot == 2 ? 0 : slot + 1;
      }
    }
    if constexpr (STAMP) {
      if (lane == 0 && wgid < 1024) {
        unsigned long long* d = g_wgd_stamps + ((long long)wgid * 8 + wave_raw) * 4;
        d[0] = tw; d[1] = ti; d[2] = tc; d[3] = __builtin_amdgcn_s_memtime() - tk0;
      }
    }
    if (loader) return;
  } else {
#pragma unroll
  for (int s0_ = 0; s0_ < NSTG - 1; ++s0_) issue(s0_);
  int st_c = 0, st_i = NSTG - 1;
  for (int k = 0; k < nk; ++k) {
    if constexpr (STAMP) t0 = __builtin_amdgcn_s_memtime();
    if constexpr (NSTG == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (2 * NQ * (NSTG - 2) == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (2 * NQ * (NSTG - 2) == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (2 * NQ * (NSTG - 2) == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else static_assert(NSTG == 2, "add the vmcnt literal");
    __builtin_amdgcn_s_barrier();
    if constexpr (STAMP) { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); tw += t1 - t0; t0 = t1; }
    issue(st_i);
    if constexpr (STAMP) { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); ti += t1 - t0; t0 = t1; }
    compute(st_c);
    if constexpr (STAMP) {
      asm volatile("s_nop 0" ::: "memory");
      const unsigned long long t1 = __builtin_amdgcn_s_memtime(); tc += t1 - t0;
    }
    st_c = (st_c + 1 == NSTG) ? 0 : st_c + 1;
    st_i = (st_i + 1 == NSTG) ? 0 : st_i + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (STAMP) {
    if (lane == 0 && wgid < 1024) {
      unsigned long long* d = g_wgd_stamps + ((long long)wgid * 8 + wave) * 4;
      d[0] = tw; d[1] = ti; d[2] = tc; d[3] = __builtin_amdgcn_s_memtime() - tk0;
    }
  }
  }   // !WS

  float* part = p.part + (long long)bz * p.CsRows * p.ncols;
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int cs = s0 + ws * SW + j * 16 + (lane & 15);
    if (cs >= p.CsRows) continue;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int col = g0 + wg * 64 + i * 16 + (lane >> 4) * 4;
      if (col >= p.ncols) continue;
      if constexpr (WDBG == 3) asm volatile("" :: "v"(acc[i][j])); else
      *reinterpret_cast<f32x4*>(part + (long long)cs * p.ncols + col) = acc[i][j];
    }
  }
}
