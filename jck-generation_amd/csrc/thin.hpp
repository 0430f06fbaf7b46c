// Image-side ("thin") layers: the products with a 3(4)-channel tensor on one side.
//
//   img_down : Conv2d 4->64 k4 s2 p1 on an NHWC4 image (D.conv1 forward, model/DCGAN.py:10; dgrad of G.conv5 :58)
//   img_up   : ConvTranspose2d 64->4 k4 s2 p1 (+ tanh)   (G.conv5 forward, model/DCGAN.py:58-59,66; dgrad of D.conv1 :10)
//
// Both move ~42 MB per launch at B=256 and do 1.6 GFLOP: they are HBM-bound, and the tile kernel in igemm.hpp (one k-step
// per workgroup for img_down, a 9x re-staged operand for img_up) spent its time in prologue/epilogue latency.  Here there
// is no LDS staging of the activations at all: a lane loads its own MFMA B fragment (8 consecutive k = two neighbouring
// 4-channel pixels, or 8 consecutive channels of one neighbour) straight from global memory with buffer loads whose
// out-of-image offsets return zeros, the weights stay resident (registers / 18 KB of LDS), every wave streams several
// 16-pixel groups with all their loads in flight, and the output leaves as 16/32-byte runs per lane.  bf16 only: the
// fp32 parity path keeps the generic kernel.  Same k order as the tile kernel, so the products are bitwise the same.
#pragma once
#include "common.hpp"

struct ImgDownParams {
  const void* x;        // [N, H, W, 4] bf16
  const void* w;        // pack_down layout [64][64]: row = output channel, k = (ky*4+kx)*4 + ci
  void* out;            // [N, H/2, W/2, 64] bf16
  float* stats;         // [gridDim.x][2][64] partial sum / sum of squares, or nullptr
  int ngroups;          // 16-pixel groups: N * OH * OW / 16
  int H, W, logOH, logG;     // logG = log2(OW / 16)
  unsigned x_bytes;
};

template <int GPW>
static __global__ __launch_bounds__(256) void img_down_kernel(const ImgDownParams p) {
  __shared__ float red[4][128];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pcol = lane & 15, g = lane >> 4;
  const auto rs = make_rsrc(p.x, p.x_bytes);
  const bf16_t* wp = reinterpret_cast<const bf16_t*>(p.w);
  // A fragments: m-tile t row r holds channel 32*(t>>1) + 8*(r>>2) + 4*(t&1) + (r&3), so that after the product lane group g
  // owns channels 8g..8g+7 (tiles 0,1) and 32+8g..32+8g+7 (tiles 2,3) of its pixel: each of the two 16-byte stores of a
  // wave covers whole 64-byte runs
  bf16x8 wf[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int c = 32 * (t >> 1) + 8 * (pcol >> 2) + 4 * (t & 1) + (pcol & 3);
#pragma unroll
    for (int s = 0; s < 2; ++s) wf[t][s] = *reinterpret_cast<const bf16x8*>(wp + c * 64 + s * 32 + g * 8);
  }
  const int OH = 1 << p.logOH, OW = 16 << p.logG;
  const int g0 = (blockIdx.x * 4 + wave) * GPW;
  u32x2 xr[GPW][2][2];
#pragma unroll
  for (int i = 0; i < GPW; ++i) {
    const int gi = g0 + i;
    const int ox = ((gi & ((1 << p.logG) - 1)) << 4) + pcol;
    const int row = gi >> p.logG;
    const int oy = row & (OH - 1), n = row >> p.logOH;
    const int ix = 2 * ox + 2 * (g & 1) - 1;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int iy = 2 * oy + 2 * s + (g >> 1) - 1;
      const bool oky = gi < p.ngroups && (unsigned)iy < (unsigned)p.H;
      const unsigned off = (unsigned)(((n * p.H + iy) * p.W + ix) * 8);
      xr[i][s][0] = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)((oky && ix >= 0) ? off : JCK_OOB), 0, 0);
      xr[i][s][1] = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)((oky && ix + 1 < p.W) ? off + 8u : JCK_OOB), 0, 0);
    }
  }
  float ssum[16], ssq[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) { ssum[k] = 0.f; ssq[k] = 0.f; }
  bf16_t* outp = reinterpret_cast<bf16_t*>(p.out);
  // this lane's channels: k = 0..7 -> 8g + k, k = 8..15 -> 32 + 8g + (k - 8)
#pragma unroll
  for (int i = 0; i < GPW; ++i) {
    const int gi = g0 + i;
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const u32x4 raw = {xr[i][s][0][0], xr[i][s][0][1], xr[i][s][1][0], xr[i][s][1][1]};
      const bf16x8 b = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = mfma16(wf[t][s], b, acc[t]);
    }
    u32x4 o0, o1;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float v = acc[t][j]; ssum[4 * t + j] += v; ssq[4 * t + j] += v * v; }
    }
    o0[0] = pack2bf(acc[0][0], acc[0][1]); o0[1] = pack2bf(acc[0][2], acc[0][3]);
    o0[2] = pack2bf(acc[1][0], acc[1][1]); o0[3] = pack2bf(acc[1][2], acc[1][3]);
    o1[0] = pack2bf(acc[2][0], acc[2][1]); o1[1] = pack2bf(acc[2][2], acc[2][3]);
    o1[2] = pack2bf(acc[3][0], acc[3][1]); o1[3] = pack2bf(acc[3][2], acc[3][3]);
    if (gi < p.ngroups) {
      bf16_t* d = outp + ((long long)gi * 16 + pcol) * 64 + 8 * g;       // group gi = 16 consecutive output pixels
      *reinterpret_cast<u32x4*>(d) = o0;
      *reinterpret_cast<u32x4*>(d + 32) = o1;
    }
  }
  if (!p.stats) return;
#pragma unroll
  for (int k = 0; k < 16; ++k) { ssum[k] = row16_sum(ssum[k]); ssq[k] = row16_sum(ssq[k]); }
  if (pcol == 0) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int c = 32 * (k >> 3) + 8 * g + (k & 7);
      red[wave][c] = ssum[k]; red[wave][64 + c] = ssq[k];
    }
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    const float t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    p.stats[(long long)blockIdx.x * 128 + threadIdx.x] = t;
  }
}

struct ImgUpParams {
  const void* a;        // [N, Hs, Ws, 64] bf16
  const void* w;        // pack_up16 layout [16][9*64]: row = phase*4 + c, k = (dyi*3+dxi)*64 + cs
  void* out;            // [N, 2Hs, 2Ws, 4] bf16
  int nunits;           // strips of 16 columns x R rows: N * (Hs / R) * (Ws / 16)
  int Hs, Ws, logYB, logG;   // logYB = log2(Hs / R), logG = log2(Ws / 16)
  unsigned a_bytes;
  int epi_tanh;
  // optional epilogue of G's loss pass (train/dcgan_trainer.py:186-187): this launch computes D's gradient w.r.t. the noisy fake
  // image; the next op of the chain rule is the tanh + 0.9-mix backward, out = scale * bf16(acc) * (1 - t^2) with t = G's tanh
  // output at the same pixel (same layout).  The accumulator is rounded to bf16 first - the storage point the separate
  // tanh_bwd_kernel reads - so the result is bit for bit the two launches'.
  const void* mul_t; float mul_scale;
};

#define IMGUP_WLD 584     // padded LDS row (elements): 1168 B = 292 dwords, 292 % 64 = 36 -> conflict-free 16-byte row reads

// lane i of each 16-lane row takes `src` of lane i-1 (shr) / i+1 (shl); the row's first / last lane keeps `edge`
__device__ __forceinline__ u32x4 dpp_from_left(u32x4 edge, u32x4 src) {
  u32x4 r;
#pragma unroll
  for (int k = 0; k < 4; ++k) r[k] = (unsigned)__builtin_amdgcn_update_dpp((int)edge[k], (int)src[k], 0x111, 0xf, 0xf, false);
  return r;
}
__device__ __forceinline__ u32x4 dpp_from_right(u32x4 edge, u32x4 src) {
  u32x4 r;
#pragma unroll
  for (int k = 0; k < 4; ++k) r[k] = (unsigned)__builtin_amdgcn_update_dpp((int)edge[k], (int)src[k], 0x101, 0xf, 0xf, false);
  return r;
}

// One wave = one strip of 16 columns x R output-side rows of the small tensor.  Every input row of the strip is loaded ONCE
// (2 x 16 bytes per lane + the two halo pixels on lanes 0 / 15 of each 16-lane row), its left / right neighbours come from
// DPP row shifts, and the three rows a 3x3 neighbourhood needs roll through registers: ~1.25x the tensor is read instead
// of 9x.
template <int R>
static __global__ __launch_bounds__(256) void img_up_kernel(const ImgUpParams p) {
  __shared__ __attribute__((aligned(16))) bf16_t wl[16 * IMGUP_WLD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pcol = lane & 15, g = lane >> 4;
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(p.w);
    for (int c = threadIdx.x; c < 16 * 72; c += 256) {                   // 72 16-byte chunks per row
      const int r = c / 72, q = c - r * 72;
      *reinterpret_cast<u32x4*>(wl + r * IMGUP_WLD + q * 8) = src[c];
    }
  }
  __syncthreads();
  const int unit = blockIdx.x * 4 + wave;
  if (unit >= p.nunits) return;
  const auto rs = make_rsrc(p.a, p.a_bytes);
  const int Hs = p.Hs, Ws = p.Ws;
  const int x0 = (unit & ((1 << p.logG) - 1)) << 4;
  const int yb = (unit >> p.logG) & ((1 << p.logYB) - 1), n = unit >> (p.logG + p.logYB);
  const int y0 = yb * R, x = x0 + pcol;
  bf16_t* outp = reinterpret_cast<bf16_t*>(p.out);
  const bf16_t* wrow = wl + pcol * IMGUP_WLD + g * 8;
  // halo pixel of this lane: x0-1 on the row's first lane, x0+16 on its last one
  const int xh = pcol == 0 ? x0 - 1 : (pcol == 15 ? x0 + 16 : -1);
  const bool okh = (unsigned)xh < (unsigned)Ws;

  struct Row { u32x4 c[2], l[2], r[2]; };
  auto load_row = [&](int yy, u32x4 (&c)[2], u32x4 (&h)[2]) {
    const bool oky = (unsigned)yy < (unsigned)Hs;
    const unsigned base = (unsigned)(((n * Hs + yy) * Ws) * 128 + g * 16);
    const unsigned oc = oky ? base + (unsigned)x * 128u : JCK_OOB;
    const unsigned oh = (oky && okh) ? base + (unsigned)xh * 128u : JCK_OOB;
    c[0] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)oc, 0, 0);
    c[1] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(oc + 64u), 0, 0);        // JCK_OOB + 64 is still out of range
    h[0] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)oh, 0, 0);
    h[1] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(oh + 64u), 0, 0);
  };
  u32x4 rc[R + 2][2], rh[R + 2][2];
  constexpr int AHEAD = 4;
#pragma unroll
  for (int j = 0; j < AHEAD && j < R + 2; ++j) load_row(y0 - 1 + j, rc[j], rh[j]);
  Row win[3];
  auto shift_in = [&](int j, Row& w) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      w.c[h] = rc[j][h];
      w.l[h] = dpp_from_left(rh[j][h], rc[j][h]);
      w.r[h] = dpp_from_right(rh[j][h], rc[j][h]);
    }
  };
  shift_in(0, win[0]);
  shift_in(1, win[1]);
#pragma unroll
  for (int i = 0; i < R; ++i) {
    if (i + AHEAD < R + 2) load_row(y0 - 1 + i + AHEAD, rc[i + AHEAD], rh[i + AHEAD]);
    shift_in(i + 2, win[(i + 2) % 3]);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 3; ++d) {                                       // input row y + d - 1
      const Row& w = win[(i + d) % 3];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        acc = mfma16(lds_frag(wrow + ((d * 3 + 0) * 2 + h) * 32), __builtin_bit_cast(bf16x8, w.l[h]), acc);
        acc = mfma16(lds_frag(wrow + ((d * 3 + 1) * 2 + h) * 32), __builtin_bit_cast(bf16x8, w.c[h]), acc);
        acc = mfma16(lds_frag(wrow + ((d * 3 + 2) * 2 + h) * 32), __builtin_bit_cast(bf16x8, w.r[h]), acc);
      }
    }
    float v[4] = {acc[0], acc[1], acc[2], acc[3]};
    if (p.epi_tanh) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]);
    }
    // lane group g = output parity (py, px): rows 4g..4g+3 are its 4 channels
    const int y = y0 + i;
    const long long o = (((long long)n * 2 * Hs + 2 * y + (g >> 1)) * 2 * Ws + 2 * x + (g & 1)) * 4;
    if (p.mul_t) {
      float t[4];
      ld4(reinterpret_cast<const bf16_t*>(p.mul_t) + o, t);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = p.mul_scale * bf2f(f2bf(v[r])) * (1.f - t[r] * t[r]);
    }
    st4(outp + o, v);
  }
}
