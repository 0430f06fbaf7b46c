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
  int tap[4][16];         // (dy << 16) | (dx & 0xffff), filled by the launcher: dword table -> scalar loads in the kernel
  long long osN;          // output offset = n*osN + oy*osY + ox*osX + obase[z]   (elements)
  int osY, osX;
  int obase[4];
  int cstat;              // number of stats channels; stats channel = ch % cstat (cstat power of two)
  int ytiles_per_cset;    // channel tiles (blockIdx.y) that cover one set of cstat channels
  int epi;                // 0 none, 1 tanh
  int rows_are_phases;    // 1: MFMA row r = phase*4 + channel (4-channel outputs, all four parities in one tile)
  int gx, gy, gz;         // logical grid: pixel tiles, channel tiles, phases (launched as a 1-D grid of gx*gy*gz)
  unsigned act_bytes, w_bytes;   // sizes of the gathered tensor and of the packed weights (buffer descriptors)
  int act_row_elems;      // > 0: plain row-major GEMM operand [M][act_row_elems] (Linear layers); taps unused
  const float* bias;      // optional per-channel bias added in the epilogue (Linear layers)
  int ksplit;             // > 1: split-K over gz = ksplit workgroup layers of `ksteps` k-steps each (plain GEMMs only);
  int ksteps;             //      layer z writes its partial tile at out + z*out_split_stride
  long long out_split_stride;
  int out_f32;            // store fp32 regardless of T (split-K slabs)
  long long w_phase_stride;
  int bn_group_rows;      // > 0: BatchNorm group = pixel row / bn_group_rows (independent batches that went through ONE launch)
  // forward statistics accumulated per workgroup and BatchNorm group by the persistent kernels (rows [group][rank][2][cstat])
  // instead of one row per (tile, wave); set by the launcher for the *_grouped entry points
  int stat_accum;
  int loader_prio;        // persistent kernels: s_setprio of the loader waves (jck_tune "igemm_prio")
  double flops;           // algorithmic FLOPs of this launch (profiling only)
};

#define IG_BK 64

template <class P, int BCH, int BPIX, int NW = 4> struct IgemmCfg {      // NW: waves that own accumulators (4, or 8 consumers)
  static constexpr bool F32 = P::IS_F32;
  static constexpr int WCH = (BCH >= 64) ? 2 : 1;
  static constexpr int WPIX = NW / WCH;
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


template <int FN>
__device__ __forceinline__ void igemm_pixel_offsets(const IgemmParams& p, int lane, int wpix, int m0, long long (&poff)[FN]) {
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int m = m0 + wpix * FN * 16 + j * 16 + (lane & 15);
    const int n = m >> p.logOHW;
    const int rem = m & ((1 << p.logOHW) - 1);
    poff[j] = m < p.M ? (long long)n * p.osN + (long long)(rem >> p.logOW) * p.osY + (long long)(rem & ((1 << p.logOW) - 1)) * p.osX
                      : -1;
  }
}

// Shared epilogue: optional BatchNorm partial statistics, bias, tanh, NHWC store of 4 consecutive channels per lane.
template <class P, int BCH, int BPIX, int FM, int FN, int WPIXN>
__device__ __forceinline__ void igemm_epilogue(const IgemmParams& p, f32x4 (&acc)[FM][FN], int lane, int wch, int wpix, int z, int zraw,
                                               int bidx, int bidy, int m0, int ch0) {
  typedef typename P::T T;
  // ---- epilogue --------------------------------------------------------------------------------------
  long long poff[FN];                                               // output offset of this lane's pixel in tile column j (-1: past M)
  igemm_pixel_offsets<FN>(p, lane, wpix, m0, poff);
  if (p.stats) {
    // slot = one (pixel tile, phase, channel-set replica, pixel-wave); every (slot, channel) is written exactly once
    const int yrep = bidy / p.ytiles_per_cset, nyrep = p.gy / p.ytiles_per_cset;
    // pixel tile slowest, so that the slots of consecutive pixel ranges (BatchNorm groups) are consecutive too
    const long long slot = (((long long)bidx * p.gz + zraw) * nyrep + yrep) * WPIXN + wpix;
    float* sp = p.stats + slot * 2 * p.cstat;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int ch = ch0 + wch * FM * 16 + i * 16 + (lane >> 4) * 4;
      const bool chok = ch < p.NchStore;
      const int cc = ch & (p.cstat - 1);
      float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float v = acc[i][j][r]; s[r] += v; q[r] += v * v; }
#pragma unroll
      for (int r = 0; r < 4; ++r) { s[r] = row16_sum(s[r]); q[r] = row16_sum(q[r]); }
      if ((lane & 15) == 0 && chok) {
        *reinterpret_cast<f32x4*>(sp + cc) = f32x4{s[0], s[1], s[2], s[3]};
        *reinterpret_cast<f32x4*>(sp + p.cstat + cc) = f32x4{q[0], q[1], q[2], q[3]};
      }
    }
  }
  T* outp = reinterpret_cast<T*>(p.out);
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    if (poff[j] < 0) continue;
    const long long off0 = poff[j];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      int ch = ch0 + wch * FM * 16 + i * 16 + (lane >> 4) * 4;
      if (ch >= p.NchStore) continue;
      long long off = off0 + p.obase[z] + (long long)zraw * p.out_split_stride;
      if (p.rows_are_phases) { off = off0 + p.obase[ch >> 2]; ch = 0; }
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (p.bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += p.bias[ch + r];
      }
      if (p.epi == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]);
      }
      if (p.out_f32) st4(reinterpret_cast<float*>(p.out) + off + ch, v);
      else st4(outp + off + ch, v);
    }
  }
}

template <class P, int BCH, int BPIX, int NSUB, int NST = 2>
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
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (id % 8 shares an XCD, each with its own
  // 4 MB L2), so give every XCD a contiguous run of logical tiles, ordered channel-tile fastest, then phase, then
  // pixel tile: the tiles that gather the same activation rows (other output channels, other parities, the halo of
  // the next pixel tile) run on one L2.  Bijective for any grid size.
  const int nwg = gridDim.x;
  int wgid;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int bidy = wgid % p.gy;
  const int zraw = (wgid / p.gy) % p.gz;
  const int z = p.ksplit > 1 ? 0 : zraw;                            // parity phase (tables, weights, output base)
  const int kc0 = p.ksplit > 1 ? zraw * p.ksteps : 0;               // first k-step of this split
  const int bidx = wgid / (p.gy * p.gz);
  const int m0 = bidx * BPIX;
  const int ch0 = bidy * BCH;
  const int Cc = 1 << p.logC;
  if (tid < 16) {
    const int dyv = tid < p.ntaps ? (int)p.dy[z][tid] : 0, dxv = tid < p.ntaps ? (int)p.dx[z][tid] : 0;
    toff[tid] = (dyv * p.W + dxv) << p.logC;
    tdyx[tid] = tid < p.ntaps ? ((dyv << 16) | (dxv & 0xffff)) : (0x4000 << 16);      // padding taps never validate
  }

  // ---- per-thread gather rows ---------------------------------------------------------------------
  // Loads are buffer loads with 32-bit byte offsets; an invalid tap / row selects JCK_OOB and reads zeros.
  const int lrow = tid >> 3, unit = tid & 7;
  constexpr unsigned ESZ = sizeof(T);
  unsigned rowoff[C::APASS];                                       // byte offset of tap (0,0) of the row (+ this unit's channels)
  int riy[C::APASS], rix[C::APASS];                                // iy0 (poisoned past M), ix0
#pragma unroll
  for (int ps = 0; ps < C::APASS; ++ps) {
    const int m = m0 + ps * 32 + lrow;
    const int n = m >> p.logOHW;
    const int rem = m & ((1 << p.logOHW) - 1);
    const int iy0 = (rem >> p.logOW) * p.sy, ix0 = (rem & ((1 << p.logOW) - 1)) * p.sx;
    rowoff[ps] = p.act_row_elems ? ((unsigned)m * (unsigned)p.act_row_elems + unit * 8) * ESZ
                                 : ((unsigned)(((n * p.H + iy0) * p.W + ix0) << p.logC) + (NSUB == 1 ? unit * 8 : 0)) * ESZ;
    riy[ps] = m < p.M ? iy0 : 0x40000000;
    rix[ps] = ix0;
  }
  unsigned wrowoff[C::WPASS];
#pragma unroll
  for (int ps = 0; ps < C::WPASS; ++ps)
    wrowoff[ps] = (unsigned)(((long long)z * p.w_phase_stride + (long long)(ch0 + ps * 32 + lrow) * p.K + unit * 8) * sizeof(W));
  const auto rs_act = make_rsrc(p.act, p.act_bytes);
  const auto rs_w = make_rsrc(p.w, p.w_bytes);
  __syncthreads();   // tap tables visible

  // NST register stages: the loads of k-step kc + NST are issued while kc is computed, so NST tiles are in flight per
  // workgroup - the loop is bound by (bytes in flight) / (L2 latency), not by MFMA issue, at these problem sizes
  struct Stage { Raw8<T> a[C::APASS]; Raw8<W> w[C::WPASS]; };
  Stage st[NST];

  auto load_tiles = [&](int kc, Stage& sg) {
    const int kbase = (kc + kc0) * IG_BK;                          // wave-uniform
#pragma unroll
    for (int ps = 0; ps < C::WPASS; ++ps) {
      const int r = ps * 32 + lrow;
      if (BCH >= 32 || r < BCH) buf_ld8(rs_w, wrowoff[ps] + kbase * (unsigned)sizeof(W), sg.w[ps]);
    }
    if constexpr (NSUB == 1) {
      // C >= 64: the whole 64-wide k-step lies in ONE tap -> tap index, its (dy, dx) and its offset are scalars
      const int t = p.act_row_elems ? 0 : (kbase >> p.logC);
      const int dyv = p.dy[z][t], dxv = p.dx[z][t];
      const int toffb = p.act_row_elems ? kbase * (int)ESZ : (((dyv * p.W + dxv) << p.logC) + (kbase & (Cc - 1))) * (int)ESZ;
#pragma unroll
      for (int ps = 0; ps < C::APASS; ++ps) {
        const bool ok = (unsigned)(riy[ps] + dyv) < (unsigned)p.H && (unsigned)(rix[ps] + dxv) < (unsigned)p.W;
        buf_ld8(rs_act, ok ? rowoff[ps] + (unsigned)toffb : JCK_OOB, sg.a[ps]);
      }
    } else {   // C == 4: the 8-element unit spans two taps (pixels), per-thread taps from the LDS tables
      const int t0 = (kbase + unit * 8) >> 2, t1 = t0 + 1;
      const int d0 = tdyx[t0], d1 = tdyx[t1], o0 = toff[t0] * (int)ESZ, o1 = toff[t1] * (int)ESZ;
#pragma unroll
      for (int ps = 0; ps < C::APASS; ++ps) {
        const bool ok0 = (unsigned)(riy[ps] + (d0 >> 16)) < (unsigned)p.H && (unsigned)(rix[ps] + (int)(short)(d0 & 0xffff)) < (unsigned)p.W;
        const bool ok1 = (unsigned)(riy[ps] + (d1 >> 16)) < (unsigned)p.H && (unsigned)(rix[ps] + (int)(short)(d1 & 0xffff)) < (unsigned)p.W;
        buf_ld4(rs_act, ok0 ? rowoff[ps] + (unsigned)o0 : JCK_OOB, sg.a[ps], 0);
        buf_ld4(rs_act, ok1 ? rowoff[ps] + (unsigned)o1 : JCK_OOB, sg.a[ps], 1);
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

  const int nk = p.ksplit > 1 ? p.ksteps : p.K / IG_BK;
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

  // software pipeline: LDS double buffer + NST register stages; k-step k lives in stage k % NST
  load_tiles(0, st[0]);
  store_tiles(0, st[0]);
#pragma unroll
  for (int j = 1; j < NST; ++j)
    if (j < nk) load_tiles(j, st[j]);
  __syncthreads();
  for (int kc = 0; kc < nk; kc += NST) {
#pragma unroll
    for (int j = 0; j < NST; ++j) {
      const int k = kc + j;
      if (k < nk) {
        if (k + NST < nk) load_tiles(k + NST, st[j]);              // stage j held k-step k, already in LDS
        compute(k & 1);
        if (k + 1 < nk) store_tiles((k + 1) & 1, st[(j + 1) % NST]);
        __syncthreads();
      }
    }
  }

  igemm_epilogue<P, BCH, BPIX, FM, FN, C::WPIX>(p, acc, lane, wch, wpix, z, zraw, bidx, bidy, m0, ch0);
}

// ------------------------------------------------------------------------------------------------------------------
// LDS-DMA variant (bf16, C >= 64 gathers): operands go global -> LDS directly with global_load_lds_dwordx4 (no VGPR
// staging, no ds_write), NSTG LDS stages, ONE raw s_barrier per k-step and counted s_waitcnt vmcnt so that the loads of
// the next stages stay in flight across the barrier.  Each wave-instruction writes 1 KiB = 8 tile rows x 128 B lane-
// linearly, so the bank swizzle (chunk ^= (row >> 1) & 7) is applied to the per-lane SOURCE address; out-of-image taps
// and rows past M read a 16-byte zero page instead.  Same tile geometry, fragment reads and epilogue as igemm_kernel.
// ------------------------------------------------------------------------------------------------------------------
static __device__ __attribute__((aligned(16))) unsigned int g_jck_zero_page[64];

// WS (wave-specialised, NSTG = 3): the last 4 waves only issue the LDS-DMA (two stages ahead), the first NCW waves only read
// fragments and feed the MFMA - see wgrad_dma_kernel in wgrad.hpp for the measurement behind it.  Used when the launch has
// about one workgroup per CU, where a 4-wave workgroup would serialise DMA issue and MFMA in every wave.
// NCW = 8 (768 threads) with a 128 x 256 tile: the weight tile is filled once for twice the pixels - 85 instead of 64 FLOP
// per filled byte (the kernels are bound by the LDS fill rate).
template <int BCH, int BPIX, int NSTG, bool WS = false, int NCW = 4>
__global__ __launch_bounds__(WS ? (NCW + 4) * 64 : 256) void igemm_dma_kernel(const IgemmParams p) {
  static_assert(!WS || NSTG == 3, "wave specialisation uses 3 LDS stages");
  static_assert(NCW == 4 || (WS && NCW == 8), "8 consumer waves exist in the wave-specialised form only");
  typedef PrecBf16 P;
  typedef IgemmCfg<P, BCH, BPIX, NCW> C;
  constexpr int FM = C::FM, FN = C::FN, LD = IG_BK;                 // unpadded 128-byte rows
  constexpr int STG_BYTES = (BCH + BPIX) * LD * 2;
  constexpr int NLD = (BCH + BPIX) / 32;                            // DMA wave-instructions per stage and wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* lds = smem_raw;                                    // the ONLY shared object (hipcc wait-insertion trap)

  const bool loader = WS && __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) >= NCW;
  const int tid = loader ? (int)threadIdx.x - NCW * 64 : (int)threadIdx.x, lane = tid & 63;   // position inside the role
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = gridDim.x;
  int wgid;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  wgid = __builtin_amdgcn_readfirstlane(wgid);                      // provably wave-uniform: tap tables come by s_load
  const int bidy = wgid % p.gy;
  const int zraw = (wgid / p.gy) % p.gz;
  const int z = p.ksplit > 1 ? 0 : zraw;                            // parity phase (tables, weights, output base)
  const int kc0 = p.ksplit > 1 ? zraw * p.ksteps : 0;               // split-K (plain GEMMs): first k-step of this layer of workgroups
  const int bidx = wgid / (p.gy * p.gz);
  const int m0 = bidx * BPIX, ch0 = bidy * BCH;
  const int Cc = 1 << p.logC;

  const int lrow = (tid & 255) >> 3, unit = tid & 7;                 // fill geometry of the 4 issuing waves
  const unsigned src_chunk = (unsigned)(unit ^ ((lrow >> 1) & 7)) * 16u;      // swizzle on the source side
  unsigned rowoff[C::APASS];
  int riy[C::APASS], rix[C::APASS];
#pragma unroll
  for (int ps = 0; ps < C::APASS; ++ps) {
    const int m = m0 + ps * 32 + lrow;
    const int n = m >> p.logOHW;
    const int rem = m & ((1 << p.logOHW) - 1);
    const int iy0 = (rem >> p.logOW) * p.sy, ix0 = (rem & ((1 << p.logOW) - 1)) * p.sx;
    rowoff[ps] = p.act_row_elems ? (unsigned)m * (unsigned)p.act_row_elems * 2u
                                 : ((unsigned)(((n * p.H + iy0) * p.W + ix0) << p.logC)) * 2u;
    riy[ps] = m < p.M ? iy0 : 0x40000000;
    rix[ps] = ix0;
  }
  unsigned wrowoff[C::WPASS];
#pragma unroll
  for (int ps = 0; ps < C::WPASS; ++ps)
    wrowoff[ps] = (unsigned)(((long long)z * p.w_phase_stride + (long long)(ch0 + ps * 32 + lrow) * p.K) * 2);
  const unsigned char* actb = reinterpret_cast<const unsigned char*>(p.act);
  const unsigned char* wb = reinterpret_cast<const unsigned char*>(p.w);

  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int nkt = p.K / IG_BK, nk = p.ksplit > 1 ? p.ksteps : nkt;   // k-steps of the product / of this workgroup
  // k-steps past the end re-load the last tile into a stage nobody reads: no predicate, no branch, exact vmcnt arithmetic
  // buffer loads (descriptor + 32-bit offset per lane, zeros past the operand) - see igemm_dma_persist_kernel
  const auto rs_a = make_rsrc(p.act, p.act_bytes);
  const auto rs_wt = make_rsrc(p.w, p.w_bytes);
  auto issue = [&](int kc, int stage) {
    const int kbase = min(kc0 + kc, nkt - 1) * IG_BK;
    unsigned char* sb = lds + stage * STG_BYTES + (wave & 3) * (8 * LD * 2);     // this wave's 8 rows of each 32-row pass
#pragma unroll
    for (int ps = 0; ps < C::WPASS; ++ps)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wt, (lptr_t)(sb + ps * (32 * LD * 2)), 16,
                                               (int)(wrowoff[ps] + (unsigned)kbase * 2u + src_chunk), 0, 0, 0);
    const int t = __builtin_amdgcn_readfirstlane(p.act_row_elems ? 0 : (kbase >> p.logC));
    const int tp = p.tap[z][t];
    const int dyv = tp >> 16, dxv = (int)(short)(tp & 0xffff);
    const unsigned toffb = p.act_row_elems ? (unsigned)kbase * 2u
                                           : (unsigned)((((dyv * p.W + dxv) << p.logC) + (kbase & (Cc - 1))) * 2);
#pragma unroll
    for (int ps = 0; ps < C::APASS; ++ps) {
      const bool ok = (unsigned)(riy[ps] + dyv) < (unsigned)p.H && (unsigned)(rix[ps] + dxv) < (unsigned)p.W;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lptr_t)(sb + (BCH + ps * 32) * (LD * 2)), 16,
                                               (int)(ok ? rowoff[ps] + toffb + src_chunk : JCK_OOB), 0, 0, 0);
    }
  };

  const int wch = (C::WCH == 2) ? (wave / C::WPIX) : 0;
  const int wpix = (C::WCH == 2) ? (wave % C::WPIX) : wave;
  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int sw = ((lane & 15) >> 1) & 7;
  auto compute = [&](int stage) {
    const bf16_t* wt0 = reinterpret_cast<const bf16_t*>(lds + stage * STG_BYTES);
    const bf16_t* wt = wt0 + (wch * FM * 16 + (lane & 15)) * LD;
    const bf16_t* at = wt0 + BCH * LD + (wpix * FN * 16 + (lane & 15)) * LD;
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
  };

  if constexpr (WS) {
    if (loader) {
      issue(0, 0); issue(1, 1);                                       // stages 0, 1 in flight
      int slot = 2;
      for (int k = 0; k < nk; ++k) {
        if constexpr (NLD == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // stage k has landed (this wave's pieces)
        else if constexpr (NLD == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if constexpr (NLD == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else static_assert(NLD == 8 || NLD == 6 || NLD == 12, "add the vmcnt literal");
        __builtin_amdgcn_s_barrier();                                 // consumers may read stage k; stage k-1 is free
        issue(k + 2, slot);                                           // past the end: re-loads the last tile, never read
        slot = slot == 2 ? 0 : slot + 1;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      return;
    }
    int slot = 0;
    for (int k = 0; k < nk; ++k) {
      __builtin_amdgcn_s_barrier();
      compute(slot);
      slot = slot == 2 ? 0 : slot + 1;
    }
    igemm_epilogue<P, BCH, BPIX, FM, FN, C::WPIX>(p, acc, lane, wch, wpix, z, zraw, bidx, bidy, m0, ch0);
    return;
  }
  // prologue: NSTG-1 stages in flight
#pragma unroll
  for (int s = 0; s < NSTG - 1; ++s) issue(s, s);
  int st_c = 0, st_i = NSTG - 1;                                    // stage to compute, stage to refill
  for (int k = 0; k < nk; ++k) {
    // all but the (NSTG-2) youngest stages have landed -> stage k is in LDS (this wave's part); the barrier then covers
    // every wave's part and also guarantees that everybody finished reading the stage that is refilled next
    if constexpr (NSTG == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (NLD * (NSTG - 2) == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (NLD * (NSTG - 2) == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (NLD * (NSTG - 2) == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (NLD * (NSTG - 2) == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else static_assert(NSTG == 2 || NLD * (NSTG - 2) == 6 || NLD * (NSTG - 2) == 8 || NLD * (NSTG - 2) == 12 || NLD * (NSTG - 2) == 16, "add the vmcnt literal");
    __builtin_amdgcn_s_barrier();
    issue(k + NSTG - 1, st_i);
    compute(st_c);
    st_c = (st_c + 1 == NSTG) ? 0 : st_c + 1;
    st_i = (st_i + 1 == NSTG) ? 0 : st_i + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // drain the dead tail loads before the epilogue reuses nothing of LDS
  igemm_epilogue<P, BCH, BPIX, FM, FN, C::WPIX>(p, acc, lane, wch, wpix, z, zraw, bidx, bidy, m0, ch0);
}

// Epilogue of the persistent kernel: 16-byte stores.  The loader fills LDS weight row r of every 32-row block with output
// channel pi(r) = 8*((r&15)>>2) + 4*(r>>4) + (r&3) of that block, so that after the product lane group g = lane>>4 holds channels
// 8g..8g+3 in fragment 2k and 8g+4..8g+7 in fragment 2k+1: ONE 16-byte store per pixel and fragment pair, 64-byte runs per
// pixel and store instruction (the shared epilogue writes 8 bytes per lane in 32-byte runs, and an instrumented build without
// the stores ran the 3B products 11-28 % faster: the store path, not the MFMA, was the tail of every tile).  bf16, no bias / tanh /
// fp32 output / BatchNorm-backward statistics - the launcher sends those launches to igemm_dma_kernel.
__device__ __forceinline__ int igemm_perm_row(int r) { return 8 * ((r & 15) >> 2) + 4 * (r >> 4) + (r & 3); }
// per-lane partial BatchNorm sums of fragment pair k over the lane's FN pixels: v[c] = sum y, v[8 + c] = sum y^2 of channel 8g + c
template <int FM, int FN>
__device__ __forceinline__ void igemm_pair_sums(const f32x4 (&acc)[FM][FN], int k, float (&v)[16]) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int j = 0; j < FN; ++j) { const float y = acc[2 * k + (c >> 2)][j][c & 3]; s += y; q += y * y; }
    v[c] = s; v[8 + c] = q;
  }
}
// One row of BatchNorm partial sums per WORKGROUP (round 5): the WPIXN pixel-waves of a channel block leave their lane values in LDS
// (sx[wave][pair][lane]) and arrive on an LDS counter; the wave that arrives last adds them in wave order - the sum does not depend
// on who is last - and stores the block's 64 channels of the row.  No barrier (the loader waves never come here), 4x / 2x fewer
// rows for whoever sums them (bn_fwd_fused_kernel's prologue, bn_finalize).  The counter is back at zero when the last wave leaves.
template <int NPAIR, int WPIXN>
__device__ __forceinline__ void igemm_wg_row(float* sx, unsigned* scnt, int wave, int lane, int wch, const float (&val)[NPAIR], float* row,
                                             int cstat, int ch_base, int nch_store) {
#pragma unroll
  for (int k = 0; k < NPAIR; ++k) sx[(wave * NPAIR + k) * 64 + lane] = val[k];
  unsigned old = 0;
  if (lane == 0) old = __hip_atomic_fetch_add(scnt + wch, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
  old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
  if (old != (unsigned)(WPIXN - 1)) return;
  if (lane == 0) __hip_atomic_store(scnt + wch, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  const int r = lane & 15, g = lane >> 4;
#pragma unroll
  for (int k = 0; k < NPAIR; ++k) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < WPIXN; ++q) t += sx[((wch * WPIXN + q) * NPAIR + k) * 64 + lane];
    const int ch = ch_base + k * 32 + 8 * g;
    if (ch < nch_store) row[(r >> 3) * cstat + (ch & (cstat - 1)) + (r & 7)] = t;
  }
}
template <int BCH, int BPIX, int FM, int FN, int WPIXN, bool NOSTORE = false>
__device__ __forceinline__ void igemm_epilogue_perm(const IgemmParams& p, f32x4 (&acc)[FM][FN], int lane, int wch, int wpix, int z,
                                                    int bidx, int bidy, int m0, int ch0, bool tile_stats, float* sx, unsigned* scnt) {
  static_assert(FM % 2 == 0, "fragment pairs");
  long long poff[FN];
  igemm_pixel_offsets<FN>(p, lane, wpix, m0, poff);
  const int g = lane >> 4;
  bf16_t* outp = reinterpret_cast<bf16_t*>(p.out);
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    if (poff[j] < 0) continue;
    const long long off = poff[j] + p.obase[z];
#pragma unroll
    for (int k = 0; k < FM / 2; ++k) {
      const int ch = ch0 + wch * FM * 16 + k * 32 + 8 * g;
      if (ch >= p.NchStore) continue;
      const float v[8] = {acc[2 * k][j][0], acc[2 * k][j][1], acc[2 * k][j][2], acc[2 * k][j][3],
                          acc[2 * k + 1][j][0], acc[2 * k + 1][j][1], acc[2 * k + 1][j][2], acc[2 * k + 1][j][3]};
      if constexpr (NOSTORE) asm volatile("" :: "v"(v[0]), "v"(v[7])); else
      st8(outp + off + ch, v);
    }
  }
  // (the tile's statistics are formed AFTER its stores have been issued: ~1 us of DPP / VALU work that runs while they drain)
  if (p.stats && tile_stats) {
    // slot = one (pixel tile, phase, channel-set replica): ONE row per tile
    const int yrep = bidy / p.ytiles_per_cset, nyrep = p.gy / p.ytiles_per_cset;
    const long long slot = ((long long)bidx * p.gz + z) * nyrep + yrep;
    float t[FM / 2];
#pragma unroll
    for (int k = 0; k < FM / 2; ++k) {
      float v[16];
      igemm_pair_sums<FM, FN>(acc, k, v);
      t[k] = row16_transpose_sum(v, lane & 15);                      // lane r: sum (r < 8) / sum of squares (r >= 8) of channel 8g + (r & 7)
    }
    igemm_wg_row<FM / 2, WPIXN>(sx, scnt, wch * WPIXN + wpix, lane, wch, t, p.stats + slot * 2 * p.cstat, p.cstat, ch0 + wch * FM * 16, p.NchStore);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Persistent form of the wave-specialised LDS-DMA kernel (round 2): gridDim.x <= (workgroups the chip holds) workgroups walk
// the logical tiles L = blockIdx.x, blockIdx.x + gridDim.x, ...  The loader waves run two k-steps ahead ACROSS tile borders:
// while the consumer waves store a tile (epilogue: 2-5 us of NHWC stores and BatchNorm partials), the first two stages of the
// workgroup's next tile are already in flight, and no workgroup launch / LDS allocation sits between two tiles of a CU.
// Same tiles and k order as igemm_dma_kernel<.., WS = true>, the output channels of a 32-row block permuted between LDS row and
// MFMA row so that the epilogue stores 16 bytes per lane (igemm_epilogue_perm) - the same products in the same order: bitwise
// the same results.  Forward statistics (p.stat_accum) are accumulated per workgroup and BatchNorm group.
// Both roles execute exactly (tiles of this workgroup) x (K / 64) barriers.
// ------------------------------------------------------------------------------------------------------------------
template <int BCH, int BPIX, int NCW, int ADIV = 1>
__global__ __launch_bounds__((NCW + 4) * 64) void igemm_dma_persist_kernel(const IgemmParams p) {
  typedef PrecBf16 P;
  typedef IgemmCfg<P, BCH, BPIX, NCW> C;
  constexpr int FM = C::FM, FN = C::FN, LD = IG_BK;
  constexpr int STG_BYTES = (BCH + BPIX) * LD * 2;
  // ADIV != 1: TIMING EXPERIMENTS ONLY (wrong results): 2 / 4 = part of the gather skipped; 101 no loads, 102 no MFMAs, 103 no epilogue, 104 barriers only
  constexpr bool NOLOAD = ADIV == 101 || ADIV == 104 || ADIV == 108, NOMFMA = ADIV == 102 || ADIV == 104, NOEPI = ADIV == 103 || ADIV == 104 || ADIV == 108;
  constexpr bool NOSTORE = ADIV == 105, NOSTAT = ADIV == 106, NOREAD = ADIV == 107 || ADIV == 108;
  constexpr int AD = ADIV > 100 ? 1 : ADIV;
  constexpr int NLD = NOLOAD ? 0 : (BCH + BPIX / AD) / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* lds = smem_raw;

  const bool loader = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) >= NCW;
  const int tid = loader ? (int)threadIdx.x - NCW * 64 : (int)threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = p.gx * p.gy * p.gz;
  const int nk = p.K / IG_BK;
  const int Cc = 1 << p.logC;
  // logical tile L -> (bidx, bidy, z): the XCD-aware order of igemm_dma_kernel over ntiles (L % 8 = XCD of every tile of this
  // workgroup, because gridDim.x is a multiple of 8 whenever a workgroup takes more than one tile)
  int bidx = 0, bidy = 0, z = 0, m0 = 0, ch0 = 0;
  auto locate = [&](int L) {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = L & 7, idx = L >> 3;
    const int wgid = __builtin_amdgcn_readfirstlane((xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx);
    bidy = wgid % p.gy;
    z = (wgid / p.gy) % p.gz;
    bidx = wgid / (p.gy * p.gz);
    m0 = bidx * BPIX; ch0 = bidy * BCH;
  };
  int my_tiles = 0;
  for (int L = blockIdx.x; L < ntiles; L += gridDim.x) ++my_tiles;
  my_tiles = __builtin_amdgcn_readfirstlane(my_tiles);

  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  if (loader) {
    // the loader waves are the younger half of the workgroup and lose the issue arbitration at equal priority (MI355X guide, "Two
    // waves per SIMD", item 4); the kernel is bound by how fast they issue their pieces: +0.3..3 % per launch alone, neutral in the step
    if (p.loader_prio) __builtin_amdgcn_s_setprio(1);
    const int lrow = (tid & 255) >> 3, unit = tid & 7;
    const unsigned src_chunk = (unsigned)(unit ^ ((lrow >> 1) & 7)) * 16u;
    unsigned rowoff[C::APASS], wrowoff[C::WPASS];
    int riy[C::APASS], rix[C::APASS];
    auto setup = [&](int L) {
      locate(L);
#pragma unroll
      for (int ps = 0; ps < C::APASS; ++ps) {
        const int m = m0 + ps * 32 + lrow;
        const int n = m >> p.logOHW;
        const int rem = m & ((1 << p.logOHW) - 1);
        const int iy0 = (rem >> p.logOW) * p.sy, ix0 = (rem & ((1 << p.logOW) - 1)) * p.sx;
        rowoff[ps] = ((unsigned)(((n * p.H + iy0) * p.W + ix0) << p.logC)) * 2u;
        riy[ps] = m < p.M ? iy0 : 0x40000000;
        rix[ps] = ix0;
      }
#pragma unroll
      for (int ps = 0; ps < C::WPASS; ++ps)          // LDS row ps*32 + lrow <- output channel ch0 + ps*32 + pi(lrow) (igemm_epilogue_perm)
        wrowoff[ps] = (unsigned)(((long long)z * p.w_phase_stride + (long long)(ch0 + ps * 32 + igemm_perm_row(lrow)) * p.K) * 2);
    };
    // The pieces are BUFFER loads: a wave-uniform descriptor in SGPRs plus ONE 32-bit offset VGPR per lane instead of a 64-bit
    // address pair (round 3: the same bytes into the same LDS slots, bit-identical results, the gather-GEMMs 3-18 % faster -
    // the loader waves spend most of their time in the ISSUE of these pieces).  An offset past the descriptor's size (JCK_OOB)
    // returns zeros, which replaces the zero page for out-of-image taps and rows past M.
    const auto rs_a = make_rsrc(p.act, p.act_bytes);
    const auto rs_wt = make_rsrc(p.w, p.w_bytes);
    auto issue = [&](int kc, int stage) {
      if constexpr (NOLOAD) return;
      const int kbase = kc * IG_BK;
      unsigned char* sb = lds + stage * STG_BYTES + (wave & 3) * (8 * LD * 2);
#pragma unroll
      for (int ps = 0; ps < C::WPASS; ++ps)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wt, (lptr_t)(sb + ps * (32 * LD * 2)), 16,
                                                 (int)(wrowoff[ps] + (unsigned)kbase * 2u + src_chunk), 0, 0, 0);
      const int t = __builtin_amdgcn_readfirstlane(kbase >> p.logC);
      const int tp = p.tap[z][t];
      const int dyv = tp >> 16, dxv = (int)(short)(tp & 0xffff);
      const unsigned toffb = (unsigned)((((dyv * p.W + dxv) << p.logC) + (kbase & (Cc - 1))) * 2);
#pragma unroll
      for (int ps = 0; ps < C::APASS / AD; ++ps) {
        const bool ok = (unsigned)(riy[ps] + dyv) < (unsigned)p.H && (unsigned)(rix[ps] + dxv) < (unsigned)p.W;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lptr_t)(sb + (BCH + ps * 32) * (LD * 2)), 16,
                                                 (int)(ok ? rowoff[ps] + toffb + src_chunk : JCK_OOB), 0, 0, 0);
      }
    };
    // (Lq, kq): the next k-step to issue; past the last tile the last k-step is issued again into a stage nobody reads
    int Lq = blockIdx.x, kq = 0, slot = 0;
    if (my_tiles > 0) setup(Lq);
    auto next = [&]() {
      issue(kq, slot);
      slot = slot == 2 ? 0 : slot + 1;
      if (kq + 1 < nk) { ++kq; return; }
      if (Lq + (int)gridDim.x < ntiles) { Lq += gridDim.x; kq = 0; setup(Lq); }   // else: stay on the last k-step (dead re-loads)
    };
    if (my_tiles > 0) { next(); next(); }
    const int steps = my_tiles * nk;
    auto wait_older = [&]() {                                         // the older of the two k-steps in flight has landed
      if constexpr (NLD == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if constexpr (NLD == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if constexpr (NLD == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if constexpr (NLD == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else if constexpr (NLD == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if constexpr (NLD == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else if constexpr (NLD == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else static_assert(NLD == 8 || NLD == 6 || NLD == 12, "add the vmcnt literal");
    };
    for (int s = 0; s < steps; ++s) {
      wait_older();
      __builtin_amdgcn_s_barrier();                                   // consumers may read this step's stage; the one before is free
      next();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  const int wch = (C::WCH == 2) ? (wave / C::WPIX) : 0;
  const int wpix = (C::WCH == 2) ? (wave % C::WPIX) : wave;
  const int sw = ((lane & 15) >> 1) & 7;
  int slot = 0;
  // Forward statistics (p.stats && p.stat_accum) are accumulated per LANE over the tiles of the workgroup that share a BatchNorm
  // group and leave the lanes only when the group changes and at the end (a workgroup keeps its channel tile and meets its tiles
  // in increasing pixel order): rows [group][rank][2][cstat] - one per workgroup (igemm_wg_row) -, rank = position of this workgroup among those with
  // its channel tile; needs the launcher's divisibility conditions.  Rows of groups a workgroup has no tile in are zeros.
  constexpr int NPAIR = FM / 2;
  float S[NPAIR];                // lane r = lane & 15: sum (r < 8) / sum of squares (r >= 8) of channel 8g + (r & 7) of pair k (row16_transpose_sum)
  int cur_group = -1, next_row_group = 0;
  const bool acc_stats = p.stats && p.stat_accum;
  const int ngroups = acc_stats ? (p.bn_group_rows > 0 ? (p.M + p.bn_group_rows - 1) / p.bn_group_rows : 1) : 0;
  int rank = 0, rows_per_group = 0;
  if (acc_stats) {
    locate(blockIdx.x);
    // first-tile wgid = base(xcd) + idx with base % gy == 0 (launcher): channel tile = idx % gy, rank = xcd * (G/8/gy) + idx / gy
    const int idx = blockIdx.x >> 3, xcd = blockIdx.x & 7, per_xcd = (int)(gridDim.x >> 3) / p.gy;
    rank = xcd * per_xcd + idx / p.gy;
    rows_per_group = (int)(gridDim.x / p.gy);
#pragma unroll
    for (int k = 0; k < NPAIR; ++k) S[k] = 0.f;
  }
  float* sx = reinterpret_cast<float*>(lds + 3 * STG_BYTES);                    // [NCW][NPAIR][64] lane values, then WCH counters (igemm_wg_row)
  unsigned* scnt = reinterpret_cast<unsigned*>(sx + NCW * NPAIR * 64);
  if (wave == 0 && lane < C::WCH) scnt[lane] = 0u;                              // (ordered before its first use by the k-loop's barriers)
  auto rows_write = [&](int grp, bool zero) __attribute__((always_inline)) {
    float* row = p.stats + ((long long)grp * rows_per_group + rank) * 2 * p.cstat;
    if (zero) {              // a group this workgroup has no tile in: its first pixel-wave writes the zeros (no LDS exchange: several of
                             // these can follow one another without a barrier in between)
      if (wpix == 0) {
        const int r = lane & 15;
#pragma unroll
        for (int k = 0; k < NPAIR; ++k) {
          const int ch = ch0 + wch * FM * 16 + k * 32 + 8 * (lane >> 4);
          if (ch < p.NchStore) row[(r >> 3) * p.cstat + (ch & (p.cstat - 1)) + (r & 7)] = 0.f;
        }
      }
      return;
    }
    igemm_wg_row<NPAIR, C::WPIX>(sx, scnt, wave, lane, wch, S, row, p.cstat, ch0 + wch * FM * 16, p.NchStore);
  };
  auto stat_switch = [&](int grp) __attribute__((always_inline)) {        // grp: the group of the next tile (ngroups at the end)
    if (cur_group >= 0) { rows_write(cur_group, false); next_row_group = cur_group + 1; }
    for (int gq = next_row_group; gq < grp; ++gq) rows_write(gq, true);       // groups this workgroup has no tile in
    if (grp > next_row_group) next_row_group = grp;
    cur_group = grp;
#pragma unroll
    for (int k = 0; k < NPAIR; ++k) S[k] = 0.f;
  };
  for (int L = blockIdx.x; L < ntiles; L += gridDim.x) {
    locate(L);
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      // Software-pipelined consumer (round 5).  The fragments of a 32-deep half-step are read into registers while the MFMAs of
      // the half-step before run, across the k-step border too: the plain loop above exposes the LDS latency of every read group
      // (6 reads, 8 MFMAs, 2 reads, 4 MFMAs, ... - the disassembly), and both consumer waves of a SIMD do so at the same moment.
      //   half-step (k,0): reads of (k,1) are issued, then 16 MFMAs on the registers of (k,0)
      //   half-step (k,1): half of its MFMAs, then lgkmcnt(0) (every read of stage k has completed: the loaders may refill it
      //                    behind the barrier), the barrier that certifies stage k+1, the reads of (k+1,0), the other half.
      // Same products in the same order per accumulator as the plain loop: bitwise the same results.
      const int offA = (wch * FM * 16 + (lane & 15)) * LD * 2, offB = (BCH + wpix * FN * 16 + (lane & 15)) * LD * 2;
      const int c0 = ((lane >> 4) ^ sw) * 16, c1 = (((lane >> 4) + 4) ^ sw) * 16;
      bf16x8 a0[FM], b0[FN], a1[FM], b1[FN];
      auto rd = [&](bf16x8 (&a)[FM], bf16x8 (&b)[FN], int sl, int col) __attribute__((always_inline)) {
        const unsigned char* base = lds + sl * STG_BYTES + col;
        if constexpr (NOREAD) {
#pragma unroll
          for (int j = 0; j < FN; ++j) asm volatile("" : "=v"(b[j]));
#pragma unroll
          for (int i = 0; i < FM; ++i) asm volatile("" : "=v"(a[i]));
          return;
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) b[j] = lds_frag(reinterpret_cast<const bf16_t*>(base + offB + j * 16 * LD * 2));
#pragma unroll
        for (int i = 0; i < FM; ++i) a[i] = lds_frag(reinterpret_cast<const bf16_t*>(base + offA + i * 16 * LD * 2));
      };
      auto mm = [&](bf16x8 (&a)[FM], bf16x8 (&b)[FN]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            if constexpr (NOMFMA) asm volatile("" :: "v"(a[i]), "v"(b[j]));
            else acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
          }
      };
      // one read, then MFMAs, ... : a burst of 8 reads per wave, issued by all 8 waves at the same moment behind the barrier, fills the
      // LDS queue and the wave sits at its next ds_read instead of issuing the MFMAs behind it
      auto interleave = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < FM + FN; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, (FM * FN) / (FM + FN), 0);
        }
      };
      __builtin_amdgcn_s_barrier();
      rd(a0, b0, slot, c0);
      for (int k = 0; k < nk - 1; ++k) {
        rd(a1, b1, slot, c1);
        mm(a0, b0);
        interleave();
        __builtin_amdgcn_sched_barrier(0);
        slot = slot == 2 ? 0 : slot + 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        rd(a0, b0, slot, c0);
        mm(a1, b1);
        interleave();
        __builtin_amdgcn_sched_barrier(0);
      }
      rd(a1, b1, slot, c1);
      mm(a0, b0);
      interleave();
      __builtin_amdgcn_sched_barrier(0);
      slot = slot == 2 ? 0 : slot + 1;
      mm(a1, b1);
    }
    const int grp = (acc_stats && p.bn_group_rows > 0) ? m0 / p.bn_group_rows : 0;
    if (acc_stats && grp != cur_group) stat_switch(grp);
    if constexpr (NOEPI) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) asm volatile("" :: "v"(acc[i][j]));
    } else
    igemm_epilogue_perm<BCH, BPIX, FM, FN, C::WPIX, NOSTORE>(p, acc, lane, wch, wpix, z, bidx, bidy, m0, ch0, !acc_stats && !NOSTAT, sx, scnt);
    if (acc_stats && !NOEPI && !NOSTAT) {     // rows past M and taps outside the image contributed zeros to acc: no masks needed
#pragma unroll
      for (int k = 0; k < NPAIR; ++k) {
        float v[16];
        igemm_pair_sums<FM, FN>(acc, k, v);
        S[k] += row16_transpose_sum(v, lane & 15);
      }
    }
  }
  if (acc_stats) stat_switch(ngroups);
}
