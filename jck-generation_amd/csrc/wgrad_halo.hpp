// Weight gradient of the stride-2 4x4 layers with TAP REUSE (round 2):
//
//   dW[cs][t*Cb + cb] = sum_{pixel m} S[m][cs] * Big[n, 2*oy + ky - 1, 2*ox + kx - 1, cb],   t = 4*ky + kx
//
// wgrad_dma_kernel (wgrad.hpp) gathers a [64 pixels][128 columns] tile of Big per (tap, channel chunk): every pixel of Big
// is pulled out of L2 sixteen times per channel, 64 FLOP per byte filled into LDS, and its loader waves spend 86 % of the
// kernel blocked in the issue of the LDS-DMA (jck_debug_wgrad_stamps): the CU's fill path (~24-28 B/clk) is the limit, so
// bytes per FLOP set the speed.  Here a workgroup owns ALL 16 taps of a 64 (cs) x 32 (cb) block of dW: per 64-pixel k-step it
// fills the S tile (8 KB) and the input PATCH of those pixels once - (2R+2) x (2*OW+2) pixels x 32 channels, 22-30 KB - and
// every tap reads its operand out of the patch: 64 x 512 outputs per workgroup, ~120 FLOP per filled byte.
//
// Patch layout in LDS (one 64-byte slot per pixel = 32 channels): a tap touches pixels of ONE parity class (py & 1, px & 1),
// so the patch is stored as four sub-grids [py&1][px&1][py>>1][px>>1 padded to RS]; the 8 pixels a 16-lane group reads for an
// MFMA k-group are then consecutive slots.  ds_read_b64_tr_b16 serves 32 lanes per cycle = two k-groups x 4 rows x 32 B, which
// must fall into eight different 32-byte columns of the 256-byte bank row: slot u sits in column (2u + (h ^ w(u))) mod 8 for
// channel half h, with the swizzle bit w(u) = (u >> WSH) & 1 and the row pitch RS chosen per geometry so that the two k-groups
// of a cycle are an odd multiple of 2^WSH slots apart (4x4 outputs: two rows = 2*RS = 12; 8x8: one row = RS = 12; >= 16 wide:
// 8 pixels).  The LDS-DMA writes 16 bytes per lane linearly, so both swizzles are applied to the per-lane SOURCE address.
//
// 8 consumer waves (wave w: taps 2w, 2w+1 -> a 64 x 64 part of the tile, 16 MFMA per 32-pixel sub-step from 8 fragments) +
// 4 loader waves, three LDS stages, one s_barrier per k-step, counted vmcnt - the structure of wgrad_dma_kernel<.., WS>.
// Same split-K slabs and reduce kernel.  Summation order differs from wgrad_dma_kernel (pixels are still summed in order
// inside a workgroup), so results agree to fp32 rounding, not bitwise; both are deterministic.
#pragma once
#include "wgrad.hpp"

template <int LOGOW> struct HaloGeo {
  static constexpr int OW = 1 << LOGOW;
  static constexpr int LOGR = LOGOW < 6 - LOGOW ? LOGOW : 6 - LOGOW;    // output rows of one image inside a 64-pixel k-step
  static constexpr int R = 1 << LOGR;
  static constexpr int NI = 64 / (R * OW);                               // images per k-step (4 for 4x4 outputs, else 1)
  static constexpr bool WHOLE = (R == OW);                               // a k-step holds whole images
  static constexpr int RS = LOGOW == 2 ? 6 : LOGOW == 3 ? 12 : OW + 1;   // sub-grid row pitch in pixel slots (>= OW + 1)
  static constexpr int WSH = LOGOW <= 3 ? 2 : 3;
  static constexpr int SLOTS_IMG = 4 * (R + 1) * RS;
  static constexpr int SLOTS = NI * SLOTS_IMG;
  static constexpr int PPIECES = (SLOTS + 15) / 16;                      // 1 KiB LDS-DMA pieces of the patch
  static constexpr int NPW = (8 + PPIECES + 3) / 4;                      // pieces per loader wave and stage (8 = the S tile)
  static constexpr int STG_BYTES = NPW * 4 * 1024;
  static constexpr int LDS_BYTES = 3 * STG_BYTES;
};

template <int LOGOW>
static __global__ __launch_bounds__(768) void wgrad_halo_kernel(const WgradParams p) {
  typedef HaloGeo<LOGOW> G;
  constexpr int OW = G::OW, R = G::R, LOGR = G::LOGR, RS = G::RS, NPW = G::NPW, STG = G::STG_BYTES, WSH = G::WSH;
  constexpr int SUB = (R + 1) * RS;                                    // slots of one parity sub-grid
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* lds = smem_raw;                                       // the ONLY shared object (hipcc wait-insertion trap)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_raw = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave_raw >= 8;
  int wgid;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    wgid = __builtin_amdgcn_readfirstlane((xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx);
  }
  const int bx = wgid % p.gx, by = (wgid / p.gx) % p.gy, bz = wgid / (p.gx * p.gy);
  const int cb0 = bx * 32, cs0 = by * 64;
  const int c0 = bz * (p.mchunk >> 6);                                 // first 64-pixel k-step of this workgroup
  const int c1 = min(c0 + (p.mchunk >> 6), (p.Mtot + 63) >> 6);
  const int nk = max(c1 - c0, 0);

  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  if (loader) {
    if (p.loader_prio) __builtin_amdgcn_s_setprio(1);
    const int wave = wave_raw - 8;
    const unsigned char* bigb = reinterpret_cast<const unsigned char*>(p.big);
    const unsigned char* sb_ = reinterpret_cast<const unsigned char*>(p.sside);
    const int OH = p.H >> 1, nimg = p.Mtot >> p.logOHW;
    // S pieces (i = 0, 1): piece j = 4i + wave = tile rows 8j .. 8j+7, 128 bytes each
    unsigned soff[2];
    int srow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (i * 4 + wave) * 8 + (lane >> 3), pc = lane & 7;
      const int f = ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
      const int lc = pc ^ (f << 1);
      srow[i] = c0 * 64 + row;
      soff[i] = (unsigned)((srow[i] * p.CsStride + cs0 + lc * 8) * 2);
    }
    const unsigned sinc = (unsigned)(64 * p.CsStride * 2);
    // patch pieces (i = 2 .. NPW-1): piece pj = 4(i-2) + wave = slots 16pj .. 16pj+15, four 16-byte chunks each
    int prel[NPW - 2], pflag[NPW - 2];
#pragma unroll
    for (int i = 0; i < NPW - 2; ++i) {
      const int slot = (i * 4 + wave) * 16 + (lane >> 2), phys = lane & 3;
      const int img = slot / G::SLOTS_IMG, s2 = slot - img * G::SLOTS_IMG;
      const int sg = s2 / SUB, s3 = s2 - sg * SUB;
      const int spy = s3 / RS, spx = s3 - spy * RS;
      const int pyp = 2 * spy + (sg >> 1), pxp = 2 * spx + (sg & 1);     // position inside the (2R+2) x (2*OW+2) patch
      bool ok = slot < G::SLOTS && pxp >= 1 && pxp <= 2 * OW;
      if (G::WHOLE) ok = ok && pyp >= 1 && pyp <= 2 * R;
      const int lc = phys ^ (((slot >> WSH) & 1) << 1);
      prel[i] = ((((img * p.H + pyp - 1) * p.W + (pxp - 1)) << p.logCb) + cb0 + lc * 8) * 2;
      pflag[i] = (ok ? 1 : 0) | (pyp == 0 ? 2 : 0) | (pyp == 2 * R + 1 ? 4 : 0) | (img << 4);
    }
    const long long binc = G::WHOLE ? ((long long)(G::NI * p.H * p.W) << p.logCb) * 2 : ((long long)(2 * R * p.W) << p.logCb) * 2;

    // buffer loads (descriptor + 32-bit offset per lane, zeros past the operand) - see wgrad_dma_kernel
    const auto rs_big = make_rsrc(p.big, p.big_bytes);
    const auto rs_s = make_rsrc(p.sside, p.s_bytes);
    auto issue = [&](int k, int stage) {
      unsigned char* sb = lds + stage * STG;
      const bool live = k < nk;
      const int cg = c0 + k;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bool sok = live && srow[i] < p.Mtot;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_s, (lptr_t)(sb + (i * 4 + wave) * 1024), 16, (int)(sok ? soff[i] : JCK_OOB), 0, 0, 0);
        soff[i] += sinc; srow[i] += 64;
      }
      const unsigned cgoff = (unsigned)((long long)cg * binc);          // wave-uniform; a patch may start one row above it (prel < 0)
      const int oy0 = (cg << LOGR) & (OH - 1);
      const bool top = oy0 == 0, bot = oy0 + R == OH;
#pragma unroll
      for (int i = 0; i < NPW - 2; ++i) {
        bool ok = live && (pflag[i] & 1);
        if (G::WHOLE) ok = ok && (cg * G::NI + (pflag[i] >> 4)) < nimg;
        else ok = ok && !((pflag[i] & 2) && top) && !((pflag[i] & 4) && bot);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_big, (lptr_t)(sb + 8192 + (i * 4 + wave) * 1024), 16,
                                                 (int)(ok ? cgoff + (unsigned)prel[i] : JCK_OOB), 0, 0, 0);
      }
    };

    issue(0, 0); issue(1, 1);
    int slot = 2;
    for (int k = 0; k < nk; ++k) {
      // stage k has landed (this wave's pieces): the NPW pieces of stage k+1 may stay in flight
      if constexpr (NPW == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if constexpr (NPW == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
      else if constexpr (NPW == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else static_assert(NPW >= 8 && NPW <= 10, "add the vmcnt literal");
      __builtin_amdgcn_s_barrier();                                    // consumers may read stage k; stage k-1 is free
      issue(k + 2, slot);
      slot = slot == 2 ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // ---- consumer wave cw: taps t = 4*ky + kxb + tt (tt = 0, 1), all 32 channels, all 64 cs ----
  const int cw = wave_raw, ky = cw >> 1, kxb = (cw & 1) * 2;
  const int g = lane >> 4, il = lane & 15, q = il >> 2, pp = il & 3;
  // transposed-read addresses: lane (g, q, pp) supplies &tile[pixel 8g + q (+4)][4*pp .. 4*pp+3 of a 16-channel block]
  unsigned paddr[2][2][2][2];                                          // [sub-step][row half][tap][channel half]
  unsigned saddr[2][2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int rh = 0; rh < 2; ++rh) {
      const int k = kk * 32 + g * 8 + rh * 4 + q;
      const int ox = k & (OW - 1), rr = k >> LOGOW, r = rr & (R - 1), img = rr >> LOGR;
      saddr[kk][rh] = (unsigned)(k * 128 + pp * 8);
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int kx = kxb + tt, sg = (ky & 1) * 2 + (kx & 1);
        const int u = ((img * 4 + sg) * (R + 1) + r + (ky >> 1)) * RS + ox + (kx >> 1);
        const int w = (u >> WSH) & 1;
#pragma unroll
        for (int h = 0; h < 2; ++h) paddr[kk][rh][tt][h] = (unsigned)(8192 + u * 64 + ((h ^ w) << 5) + pp * 8);
      }
    }
  const int fS = ((q >> 1) & 1) | ((g & 1) << 1);                      // S-tile swizzle of this lane's rows (bits 1 and 3 of k)

  f32x4 acc[4][4];                                                     // [tap tt * 2 + channel half][cs block of 16]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int stage) {
    const unsigned char* sbase = lds + stage * STG;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        a[i] = join_tr(lds_tr4(reinterpret_cast<const bf16_t*>(sbase + paddr[kk][0][i >> 1][i & 1])),
                       lds_tr4(reinterpret_cast<const bf16_t*>(sbase + paddr[kk][1][i >> 1][i & 1])));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned o = (unsigned)((j ^ fS) << 5);
        b[j] = join_tr(lds_tr4(reinterpret_cast<const bf16_t*>(sbase + saddr[kk][0] + o)),
                       lds_tr4(reinterpret_cast<const bf16_t*>(sbase + saddr[kk][1] + o)));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
    }
  };

  int slot = 0;
  for (int k = 0; k < nk; ++k) {
    __builtin_amdgcn_s_barrier();
    compute(slot);
    slot = slot == 2 ? 0 : slot + 1;
  }

  float* part = p.part + (long long)bz * p.CsRows * p.ncols;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int cs = cs0 + j * 16 + (lane & 15);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = ky * 4 + kxb + (i >> 1);
      const int col = (t << p.logCb) + cb0 + (i & 1) * 16 + (lane >> 4) * 4;
      *reinterpret_cast<f32x4*>(part + (long long)cs * p.ncols + col) = acc[i][j];
    }
  }
}
