// Memory-bound kernels of the DCGAN/CGAN step: layout conversion + instance noise, BatchNorm
// (statistics finalise, normalise+activation, backward reduce/apply), D head (4x4 valid conv to one
// logit + sigmoid + BCE with the -100 log clamp + gradient), tanh backward, gradient-penalty norm,
// Adam, weight packing.  Everything is 16-byte vectorised along the NHWC channel axis; per-channel
// reductions use registers -> LDS -> per-workgroup partial rows summed in a fixed order (no float atomics anywhere:
// two runs of a step give bitwise identical gradients and scalars).
#pragma once
#include "common.hpp"

// ------------------------------------------------------------------------------------------------------
// In-kernel instance noise (perf mode): the reference draws 0.1 * N(0,1) for every pixel of the real and of the fake batch each
// step (train/dcgan_trainer.py:160,171).  Drawing it with ATen costs a 25 MB write plus two 12.6 MB reads per step; here each
// pixel's three normals come out of ONE Philox4x32-10 block (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3":
// counter = (pixel index, tensor id, optimiser step), key = seed) and a Box-Muller transform, inside the kernel that mixes them
// in.  rng: device uint32[4] = {seed lo, seed hi, step, 0}, written per step by jck_engine_set_step (so a captured graph of the
// step carries no per-step argument).  A different stream than torch's generator - parity runs upload their noise instead.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned (&o)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
// three N(0,1) values for pixel `i` of tensor `tensor_id` at the step held in rng[2]
__device__ __forceinline__ void pixel_normals(const unsigned* __restrict__ rng, unsigned tensor_id, long long i, float (&nz)[3]) {
  unsigned o[4];
  philox4x32_10((unsigned)i, (unsigned)(i >> 32), tensor_id, rng[2], rng[0], rng[1], o);
  const float u0 = ((float)(o[0] >> 8) + 0.5f) * (1.0f / 16777216.0f), u1 = ((float)(o[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u2 = ((float)(o[2] >> 8) + 0.5f) * (1.0f / 16777216.0f), u3 = ((float)(o[3] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  // Box-Muller on the hardware transcendentals (round 5): v_log_f32 is log2, v_sin_f32 / v_cos_f32 take their argument in
  // REVOLUTIONS - sin(2 pi u) is one instruction on u itself - and v_sqrt_f32 needs no fix-up here.  The library logf / sincosf (range
  // reduction, correctly rounded) made the two noise kernels of a step ALU-bound: ~150 instructions per pixel for three normals
  // whose last bits nobody can check (the reference's noise is torch.randn of another generator; parity tests hand the noise in).
  const float r0 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u0));
  const float r1 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u2));
  nz[0] = r0 * __builtin_amdgcn_cosf(u1); nz[1] = r0 * __builtin_amdgcn_sinf(u1); nz[2] = r1 * __builtin_amdgcn_cosf(u3);
}

// ------------------------------------------------------------------------------------------------------
// image prep: out[n][p][0..3] = keep * img[n][c][p] + mix * noise[n][c][p]   (NCHW f32 -> NHWC4 T)
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void img_prep_kernel(const float* __restrict__ img, const float* __restrict__ noise, float keep, float mix,
                                T* __restrict__ out, int N, int HW, const unsigned* __restrict__ rng = nullptr, unsigned rng_tensor = 0) {
  const long long total = (long long)N * HW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / HW;
    const int px = (int)(i - n * HW);
    float v[4], nz[3] = {0.f, 0.f, 0.f};
    if (rng) pixel_normals(rng, rng_tensor, i, nz);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const long long s = (n * 3 + c) * HW + px;
      float x = __fmul_rn(keep, img[s]);
      if (noise) x = __fmaf_rn(mix, noise[s], x);                  // explicit: the same rounding in every kernel that mixes noise
      else if (rng) x = __fmaf_rn(mix, nz[c], x);
      v[c] = x;
    }
    v[3] = 0.f;
    st4(out + i * 4, v);
  }
}

// Device-resident input pipeline: a uint8 dataset [Ntot][3][Hs][Ws] (the CIFAR pickle layout) stays in HBM and a batch is
// gathered by index and pushed through the reference's transform chain on the fly (preprocess/dcgan_data_preprocessor.py:
// 38-43): Resize(2x) as PIL does it for a bilinear upscale - horizontal pass, then vertical pass, EACH rounded to uint8 with
// the 3:1 / 1:3 weights (out = (3a + b + 2) >> 2; the clamped border taps reduce to a) -, ToTensor (/255),
// Normalize(0.5, 0.5), then the instance-noise mix of train/dcgan_trainer.py:160.  Bit-exact against PIL for the image
// (tests/golden/resize_u8.json).  out_nhwc4: [B][2Hs][2Ws][4] T (or nullptr); out_nchw: [B][3][2Hs][2Ws] fp32 transformed
// image WITHOUT noise (or nullptr).
__device__ __forceinline__ int up2_pil(int o, int L, int a_km1, int a_k, int a_kp1) {
  (void)L;
  return (o & 1) ? (3 * a_k + a_kp1 + 2) >> 2 : (a_km1 + 3 * a_k + 2) >> 2;
}
template <typename T>
__global__ void img_prep_u8_kernel(const unsigned char* __restrict__ data, const long long* __restrict__ idx,
                                   const float* __restrict__ noise, float keep, float mix, T* __restrict__ out_nhwc4,
                                   float* __restrict__ out_nchw, int B, int Hs, int Ws, const unsigned* __restrict__ rng = nullptr,
                                   unsigned rng_tensor = 0) {
  const int H = 2 * Hs, W = 2 * Ws;
  const long long total = (long long)B * H * W;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const long long n = i / ((long long)W * H);
    const unsigned char* src = data + (idx ? idx[n] : n) * 3ll * Hs * Ws;
    const int kx = x >> 1, ky = y >> 1;
    const int xm = max(kx - 1, 0), xp = min(kx + 1, Ws - 1), ym = max(ky - 1, 0), yp = min(ky + 1, Hs - 1);
    float v[4], nz[3] = {0.f, 0.f, 0.f};
    if (rng) pixel_normals(rng, rng_tensor, i, nz);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const unsigned char* pl = src + (long long)c * Hs * Ws;
      int hrow[3];                                                   // horizontal pass on source rows ym, ky, yp
      const int rows[3] = {ym, ky, yp};
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const unsigned char* row = pl + rows[r] * Ws;
        hrow[r] = up2_pil(x, Ws, row[xm], row[kx], row[xp]);
      }
      const int u8 = up2_pil(y, Hs, hrow[0], hrow[1], hrow[2]);
      const float t = ((float)u8 / 255.0f - 0.5f) / 0.5f;           // ToTensor, Normalize(0.5, 0.5)
      if (out_nchw) out_nchw[((n * 3 + c) * H + y) * W + x] = t;
      float o = __fmul_rn(keep, t);
      if (noise) o = __fmaf_rn(mix, noise[((n * 3 + c) * H + y) * W + x], o);
      else if (rng) o = __fmaf_rn(mix, nz[c], o);
      v[c] = o;
    }
    v[3] = 0.f;
    if (out_nhwc4) st4(out_nhwc4 + i * 4, v);
  }
}

// Evaluation branch (train/dcgan_trainer.py:202-206): fake = 0.5*fake + 0.5; F.resize(fake, [OH, OW]) (bilinear,
// align_corners = False: src = scale*(dst + 0.5) - 0.5 clamped at 0, as aten::upsample_bilinear2d computes it in fp32);
// (fake - mean[c]) / std[c].  One pass, NCHW fp32 in and out.
static __global__ void resize_norm_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int C, int H, int W, int OH,
                                          int OW, float pre_scale, float pre_shift, const float* __restrict__ mean,
                                          const float* __restrict__ stdv) {
  const float sh = (float)H / (float)OH, sw = (float)W / (float)OW;
  const long long total = (long long)N * C * OH * OW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW), oy = (int)((i / OW) % OH);
    const long long nc = i / ((long long)OW * OH);
    const int c = (int)(nc % C);
    const float fy = fmaxf(sh * ((float)oy + 0.5f) - 0.5f, 0.f), fx = fmaxf(sw * ((float)ox + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const float* p = in + nc * H * W;
    auto at = [&](int yy, int xx) { return __fmaf_rn(pre_scale, p[yy * W + xx], pre_shift); };
    const float top = __fmaf_rn(hx, at(y0, x0), __fmul_rn(lx, at(y0, x1)));
    const float bot = __fmaf_rn(hx, at(y1, x0), __fmul_rn(lx, at(y1, x1)));
    const float v = __fmaf_rn(hy, top, __fmul_rn(ly, bot));
    out[i] = (v - mean[c]) / stdv[c];
  }
}

// NHWC4 T -> NCHW f32 (module boundary)
template <typename T>
__global__ void nhwc4_to_nchw_kernel(const T* __restrict__ in, float* __restrict__ out, int N, int HW) {
  const long long total = (long long)N * HW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / HW;
    const int px = (int)(i - n * HW);
    float v[4];
    ld4(in + i * 4, v);
#pragma unroll
    for (int c = 0; c < 3; ++c) out[(n * 3 + c) * HW + px] = v[c];
  }
}

// out = keep * x(NHWC4 T) + mix * noise(NCHW f32)
template <typename T>
__global__ void axpy_noise_kernel(const T* __restrict__ x, const float* __restrict__ noise, float keep, float mix,
                                  T* __restrict__ out, int N, int HW, const unsigned* __restrict__ rng = nullptr, unsigned rng_tensor = 0,
                                  const T* __restrict__ real = nullptr, const float* __restrict__ alpha = nullptr, T* __restrict__ xhat = nullptr) {
  // xhat != nullptr: the penalty's interpolate x_hat = alpha[n] * real + (1 - alpha[n]) * out in the same pass (interp_kernel's
  // arithmetic on the value as STORED in `out`: same bits as the two launches)
  const long long total = (long long)N * HW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / HW;
    const int px = (int)(i - n * HW);
    float v[4], nz[3] = {0.f, 0.f, 0.f};
    ld4(x + i * 4, v);
    if (rng) pixel_normals(rng, rng_tensor, i, nz);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float r = keep * v[c];
      if (noise) r += mix * noise[(n * 3 + c) * HW + px];
      else if (rng) r += mix * nz[c];
      v[c] = r;
    }
    v[3] = 0.f;
    st4(out + i * 4, v);
    if (xhat) {
      const float al = alpha[n];
      float va[4], r[4];
      ld4(real + i * 4, va);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        T t;
        stf(&t, v[c]);
        r[c] = al * va[c] + ((1.f - al) * ldf(&t));
      }
      st4(xhat + i * 4, r);
    }
  }
}

// x_hat = alpha[n] * a + (1 - alpha[n]) * b        (NHWC4, train/dcgan_trainer.py:112)
template <typename T>
__global__ void interp_kernel(const T* __restrict__ a, const T* __restrict__ b, const float* __restrict__ alpha,
                              T* __restrict__ out, int N, int HW) {
  const long long total = (long long)N * HW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const float al = alpha[i / HW];
    float va[4], vb[4], r[4];
    ld4(a + i * 4, va);
    ld4(b + i * 4, vb);
#pragma unroll
    for (int c = 0; c < 4; ++c) r[c] = al * va[c] + ((1.f - al) * vb[c]);
    st4(out + i * 4, r);
  }
}

// g_raw = scale * g * (1 - y^2)      (tanh backward fused with the 0.9 of the instance-noise mix)
template <typename T>
__global__ void tanh_bwd_kernel(const T* __restrict__ g, const T* __restrict__ y, float scale, T* __restrict__ out,
                                long long total4) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    float vg[4], vy[4], r[4];
    ld4(g + i * 4, vg);
    ld4(y + i * 4, vy);
#pragma unroll
    for (int c = 0; c < 4; ++c) r[c] = scale * vg[c] * (1.f - vy[c] * vy[c]);
    st4(out + i * 4, r);
  }
}

// gradient penalty: (||g[n]||_2 - 1)^2 -> scal[slot][n]      one workgroup per image
// scal: per-image accumulator table [slots][scal_ld] - image n writes scal[slot*scal_ld + n] (plain store; the step tail
// sums the rows in a fixed order, so the logged scalars are bitwise reproducible)
template <typename T>
__global__ __launch_bounds__(256) void gp_norm_kernel(const T* __restrict__ g, int per_image4, float* __restrict__ scal,
                                                      int slot, int scal_ld, float* __restrict__ norms) {
  __shared__ float sm[4];
  const T* p = g + (long long)blockIdx.x * per_image4 * 4;
  float s = 0.f;
  for (int i = threadIdx.x; i < per_image4; i += 256) {
    float v[4];
    ld4(p + i * 4, v);
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  s = block_sum256(s, sm);
  if (threadIdx.x == 0) {
    const float nr = sqrtf(s);
    if (norms) norms[blockIdx.x] = nr;
    if (scal && slot >= 0) scal[(long long)slot * scal_ld + blockIdx.x] = (nr - 1.f) * (nr - 1.f);
  }
}

// ------------------------------------------------------------------------------------------------------
// BatchNorm (training mode always - the reference never switches G/D to eval)
// ------------------------------------------------------------------------------------------------------
// The streaming BatchNorm kernels run at wave priority 3: in the backward passes they share the chip with the weight-gradient
// products of the second stream, whose waves otherwise win the issue arbitration by age (step 1.8535 -> 1.8466 ms, two A/B
// rounds on one box; -DJCK_BN_PRIO=0 to compare)
#ifndef JCK_BN_PRIO
#define JCK_BN_PRIO 3
#endif
// stats: [slots][2][C] partial sums / sums of squares written by the GEMM epilogue (every slot complete).
// aux layout (floats): [0,C) scale = gamma*invstd   [C,2C) shift = beta - mean*scale
//                      [2C,3C) mean                 [3C,4C) invstd
// One workgroup per 4 channels: 256 slot-lanes, 16-byte loads, double accumulation, wavefront + LDS reduction.
static __global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ stats, int slots, float count,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                 long long* __restrict__ nbt, float momentum, float eps,
                                                                 float* __restrict__ aux, int C, float* __restrict__ stat_out = nullptr) {
  __shared__ double sh[4][8];
  const int c0 = blockIdx.x * 4;
  // grouped launch (gridDim.y > 1): group g owns slots [g*slots, (g+1)*slots), aux block g and record g (independent
  // BatchNorm batches that went through ONE conv launch - the D passes that share weights)
  stats += (long long)blockIdx.y * slots * 2 * C;
  aux += (long long)blockIdx.y * 4 * C;
  if (stat_out) stat_out += (long long)blockIdx.y * 2 * C;
  double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  // eight rows of this thread's sequence in flight (the rows are 2C floats apart: every load is a line of its own, and a launch with
  // a thousand rows was four to sixteen DEPENDENT memory round trips long - 5-9 us for a kernel that moves a megabyte); the adds
  // stay in sequence order: the same sums, bit for bit
  constexpr int FU = 8;
  for (int k0 = threadIdx.x; k0 < slots; k0 += 256 * FU) {
    f32x4 a[FU], b[FU];
#pragma unroll
    for (int u = 0; u < FU; ++u) {
      const int k = k0 + u * 256;
      if (k < slots) {
        a[u] = *reinterpret_cast<const f32x4*>(stats + (long long)k * 2 * C + c0);
        b[u] = *reinterpret_cast<const f32x4*>(stats + (long long)k * 2 * C + C + c0);
      }
    }
#pragma unroll
    for (int u = 0; u < FU; ++u) {
      if (k0 + u * 256 >= slots) break;
#pragma unroll
      for (int i = 0; i < 4; ++i) { s[i] += (double)a[u][i]; q[i] += (double)b[u][i]; }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) { s[i] = wave_sum_d(s[i]); q[i] = wave_sum_d(q[i]); }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { sh[threadIdx.x >> 6][i] = s[i]; sh[threadIdx.x >> 6][4 + i] = q[i]; }
  }
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
  if (threadIdx.x >= 4) return;
  const int i = threadIdx.x, c = c0 + i;
  const double sd = sh[0][i] + sh[1][i] + sh[2][i] + sh[3][i], qd = sh[0][4 + i] + sh[1][4 + i] + sh[2][4 + i] + sh[3][4 + i];
  const double meand = sd / (double)count;
  double vard = qd / (double)count - meand * meand;
  if (vard < 0.0) vard = 0.0;
  const float mean = (float)meand, var = (float)vard;
  const float invstd = 1.0f / sqrtf(var + eps);
  const float sc = gamma[c] * invstd;
  aux[c] = sc;
  aux[C + c] = beta[c] - mean * sc;
  aux[2 * C + c] = mean;
  aux[3 * C + c] = invstd;
  const float unbiased = var * (count / fmaxf(count - 1.f, 1.f));
  if (stat_out) {             // deferred running-stat update (passes that run concurrently on different streams)
    stat_out[c] = mean;
    stat_out[C + c] = unbiased;
  }
  if (running_mean) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
  }
}

// a = act(scale[c]*y + shift[c]), act = x>0 ? x : slope*x
// pitch > 0: the output rows of 2^log_row elements are written `pitch` elements apart - the last layer of CGAN's D lands in the
// concat buffer of the Linear head ([rows][8448], model/CGAN.py:119-121) without a copy
template <typename T>
__global__ void bn_act_fwd_kernel(const T* __restrict__ y, const float* __restrict__ aux, float slope,
                                  T* __restrict__ a, long long total8, int C, int log_row = 0, long long pitch = 0) {
  __builtin_amdgcn_s_setprio(JCK_BN_PRIO);
  const long long g0 = (long long)blockIdx.y * total8 * 8;      // first element of this group
  y += g0; aux += (long long)blockIdx.y * 4 * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total8; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)((i * 8) & (C - 1));
    float v[8];
    ld8(y + i * 8, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float z = v[k] * aux[c + k] + aux[C + c + k];
      v[k] = z > 0.f ? z : slope * z;
    }
    const long long e = g0 + i * 8;
    st8(a + (pitch ? (e >> log_row) * pitch + (e & ((1ll << log_row) - 1)) : e), v);
  }
}

// BatchNorm finalize + apply as ONE launch with no hand-over between workgroups (round 5): a workgroup owns a 64-channel SLICE of a
// pixel range (128 contiguous bytes per row in bf16), sums the statistics rows of ITS 64 channels itself - rows of 2 x 64 floats,
// double accumulation in an order fixed by the geometry - forms scale / shift in LDS and streams its pixels.  What bn_finalize did for
// everybody in a launch of its own (4.2 us of the step each, measured by skipping them: DESIGN.md section 7.2) every workgroup does
// for itself in ~2 us that overlap its first loads; it pays while the rows are few (the gather-GEMMs write one per workgroup).  The
// workgroups with blockIdx.x == 0 also write the aux table the backward reads, the deferred running-statistics record / the
// running statistics themselves.  aux, stats, stat_out: as bn_finalize_kernel; a: as bn_act_fwd_kernel (pitched output included).
#define BNF_THREADS 256
template <typename T>
__global__ __launch_bounds__(BNF_THREADS) void bn_fwd_fused_kernel(const T* __restrict__ y, const float* __restrict__ stats, int slots, float count,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float slope,
                                                                   T* __restrict__ a, float* __restrict__ aux, float* __restrict__ stat_out,
                                                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                   long long* __restrict__ nbt, float momentum, long long rows, int C,
                                                                   int log_row, long long pitch) {
  __shared__ double part[8][128];
  __shared__ float tab[2][64];
  __builtin_amdgcn_s_setprio(JCK_BN_PRIO);
  const int t = threadIdx.x, s = blockIdx.y, grp = blockIdx.z;
  stats += (long long)grp * slots * 2 * C;
  aux += (long long)grp * 4 * C;
  if (stat_out) stat_out += (long long)grp * 2 * C;
  const long long g0 = (long long)grp * rows * C;              // first element of this group
  // the first rows of this thread are requested BEFORE the table is formed: their latency hides under the row sums
  constexpr int BU = 4;
  const int u8 = t & 7;
  const long long rstep = (long long)gridDim.x * 32, rfirst = (long long)blockIdx.x * 32 + (t >> 3);
  Raw8<T> raw[BU];
#pragma unroll
  for (int u = 0; u < BU; ++u)
    if (rfirst + u * rstep < rows) ldraw(y + g0 + (rfirst + u * rstep) * C + s * 64 + u8 * 8, raw[u]);
  // ---- the slice's sums: thread = (row lane 0..7, stat, 4 channels)
  {
    const int q4 = t & 31, rl = t >> 5;
    const float* col = stats + (q4 >> 4) * C + s * 64 + (q4 & 15) * 4;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    constexpr int RU = 16;                                       // rows of this thread in flight (256 rows: two round trips)
    for (int r0 = rl; r0 < slots; r0 += 8 * RU) {
      f32x4 v[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u)
        if (r0 + u * 8 < slots) v[u] = *reinterpret_cast<const f32x4*>(col + (long long)(r0 + u * 8) * 2 * C);
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        if (r0 + u * 8 >= slots) break;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += (double)v[u][i];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) part[rl][q4 * 4 + i] = acc[i];
  }
  __syncthreads();
  if (t < 64) {
    double sd = 0.0, qd = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { sd += part[k][t]; qd += part[k][64 + t]; }
    const int c = s * 64 + t;
    const double meand = sd / (double)count;
    double vard = qd / (double)count - meand * meand;
    if (vard < 0.0) vard = 0.0;
    const float mean = (float)meand, var = (float)vard;
    const float invstd = 1.0f / sqrtf(var + eps);
    const float sc = gamma[c] * invstd, sh = beta[c] - mean * sc;
    tab[0][t] = sc; tab[1][t] = sh;
    if (blockIdx.x == 0) {
      aux[c] = sc; aux[C + c] = sh; aux[2 * C + c] = mean; aux[3 * C + c] = invstd;
      const float unbiased = var * (count / fmaxf(count - 1.f, 1.f));
      if (stat_out) { stat_out[c] = mean; stat_out[C + c] = unbiased; }
      if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
      }
      if (nbt && s == 0 && grp == 0 && t == 0) *nbt += 1;
    }
  }
  __syncthreads();
  // ---- stream: 8 threads per row slice (8 channels = 16 / 32 bytes each), 32 rows per pass
  float sc[8], sh[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { sc[k] = tab[0][u8 * 8 + k]; sh[k] = tab[1][u8 * 8 + k]; }
  // four rows of this thread in flight (few workgroups - each paid for its own table - so the bytes in flight come from the unroll)
  for (long long r0 = rfirst; r0 < rows; r0 += rstep * BU) {
    if (r0 != rfirst) {
#pragma unroll
      for (int u = 0; u < BU; ++u)
        if (r0 + u * rstep < rows) ldraw(y + g0 + (r0 + u * rstep) * C + s * 64 + u8 * 8, raw[u]);
    }
#pragma unroll
    for (int u = 0; u < BU; ++u) {
      if (r0 + u * rstep >= rows) break;
      const long long e = g0 + (r0 + u * rstep) * C + s * 64 + u8 * 8;
      float v[8];
      unraw(raw[u], v);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float z = v[k] * sc[k] + sh[k];
        v[k] = z > 0.f ? z : slope * z;
      }
      st8(a + (pitch ? (e >> log_row) * pitch + (e & ((1ll << log_row) - 1)) : e), v);
    }
  }
}

// stage 1: partial[blk][0..C) = sum g_z,  partial[blk][C..2C) = sum g_z * xhat over this workgroup's rows,
// g_z = g_a * act'(z).  Register accumulation per (row-lane, 8-channel unit), one LDS pass over the row-lanes; no atomics:
// the summation order per (thread, channel) is fixed by the launch geometry alone, so the result is bitwise reproducible.
// Geometry (measured in the step, round 2): <= 256 workgroups per group.  A variant with 1024 workgroups and 4 rows in flight per
// thread (8x the bytes in flight) was SLOWER in the step (29.3 vs 22.1 us average): the kernel runs beside the weight-gradient
// product of the previous layer on the second stream and both are bound by the same memory system.  Two rows in flight at the
// same 256 workgroups (UNR = 2, the default; same summation order, same bits) is the optimum: step 1.867 -> 1.837 ms, UNR 4: 1.853.
#define BN_BWD_MAX_BLOCKS 256
template <typename T, int UNR = 1>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ ga, const T* __restrict__ y,
                                                            const float* __restrict__ aux, float slope,
                                                            float* __restrict__ partial, long long rows, int C,
                                                            long long group_stride = 0) {
  extern __shared__ float lsum[];                         // [rstep][2][C]
  __builtin_amdgcn_s_setprio(JCK_BN_PRIO);
  ga += (long long)blockIdx.y * rows * C; y += (long long)blockIdx.y * rows * C;                    // group
  aux += (long long)blockIdx.y * 4 * C; partial += (long long)blockIdx.y * group_stride;
  const int upr = C >> 3;                                 // 8-channel units per row
  const int u = threadIdx.x % upr, r0 = threadIdx.x / upr, rstep = 256 / upr;
  const int c = u * 8;
  float sc[8], sh[8], mu[8], is[8], s1[8], s2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    sc[k] = aux[c + k]; sh[k] = aux[C + c + k]; mu[k] = aux[2 * C + c + k]; is[k] = aux[3 * C + c + k];
    s1[k] = 0.f; s2[k] = 0.f;
  }
  // UNR rows of this thread's sequence are loaded together and accumulated in sequence order: the same sums, bit for bit,
  // with UNR times the bytes in flight
  const long long stride = (long long)gridDim.x * rstep;
  for (long long r = (long long)blockIdx.x * rstep + r0; r < rows; r += stride * UNR) {
    float vg[UNR][8], vy[UNR][8];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long long ru = r + u * stride;
      if (ru < rows) { ld8(ga + ru * C + c, vg[u]); ld8(y + ru * C + c, vy[u]); }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (r + u * stride >= rows) break;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float z = vy[u][k] * sc[k] + sh[k];
        const float gz = z > 0.f ? vg[u][k] : slope * vg[u][k];
        s1[k] += gz;
        s2[k] += gz * ((vy[u][k] - mu[k]) * is[k]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) { lsum[(r0 * 2) * C + c + k] = s1[k]; lsum[(r0 * 2 + 1) * C + c + k] = s2[k]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    float t = 0.f;
    for (int rr = 0; rr < rstep; ++rr) t += lsum[rr * 2 * C + i];
    partial[(long long)blockIdx.x * 2 * C + i] = t;
  }
}

// stage 2: sums[g][0..2C) = sum over workgroups for every group g; dgamma += sum_{g < grad_groups} s2, dbeta likewise (when
// given) - the groups are walked in order inside the workgroup, so the gradient does not depend on any arrival order.
// One workgroup per 4 channels.
static __global__ __launch_bounds__(256) void bn_bwd_sums_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ sums,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta, int C,
                                                                 long long group_stride = 0, int groups = 1, int grad_groups = 1,
                                                                 long long partial_group_stride = -1) {
  // groups in chunks of four: the loads of a chunk's groups are in flight together and one barrier pair serves them (the
  // grouped D pass has 3: one memory round trip instead of three; same summation order per group - same bits)
  __shared__ float sh[4][4][8];
  const int c0 = blockIdx.x * 4;
  const long long pgs = partial_group_stride >= 0 ? partial_group_stride : group_stride;
  float dg = 0.f, db = 0.f;
  for (int g0 = 0; g0 < groups; g0 += 4) {
    float s[4][4], q[4][4];
#pragma unroll
    for (int gg = 0; gg < 4; ++gg)
#pragma unroll
      for (int i = 0; i < 4; ++i) { s[gg][i] = 0.f; q[gg][i] = 0.f; }
    for (int k = threadIdx.x; k < nblk; k += 256) {
#pragma unroll
      for (int gg = 0; gg < 4; ++gg) {
        if (g0 + gg >= groups) continue;
        const float* pp = partial + (long long)(g0 + gg) * pgs;
        const f32x4 a = *reinterpret_cast<const f32x4*>(pp + (long long)k * 2 * C + c0);
        const f32x4 b = *reinterpret_cast<const f32x4*>(pp + (long long)k * 2 * C + C + c0);
#pragma unroll
        for (int i = 0; i < 4; ++i) { s[gg][i] += a[i]; q[gg][i] += b[i]; }
      }
    }
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) {
      if (g0 + gg >= groups) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i) { s[gg][i] = wave_sum(s[gg][i]); q[gg][i] = wave_sum(q[gg][i]); }
    }
    __syncthreads();                                        // the previous chunk's sh[] has been read
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int gg = 0; gg < 4; ++gg) {
        if (g0 + gg >= groups) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) { sh[gg][threadIdx.x >> 6][i] = s[gg][i]; sh[gg][threadIdx.x >> 6][4 + i] = q[gg][i]; }
      }
    }
    __syncthreads();
    if (threadIdx.x < 4) {
      const int i = threadIdx.x, c = c0 + i;
      for (int gg = 0; gg < 4 && g0 + gg < groups; ++gg) {
        const int g = g0 + gg;
        const float s1 = sh[gg][0][i] + sh[gg][1][i] + sh[gg][2][i] + sh[gg][3][i];
        const float s2 = sh[gg][0][4 + i] + sh[gg][1][4 + i] + sh[gg][2][4 + i] + sh[gg][3][4 + i];
        sums[(long long)g * group_stride + c] = s1;
        sums[(long long)g * group_stride + C + c] = s2;
        if (g < grad_groups) { dg += s2; db += s1; }
      }
    }
  }
  if (threadIdx.x < 4) {
    const int c = c0 + threadIdx.x;
    if (dgamma) dgamma[c] += dg;
    if (dbeta) dbeta[c] += db;
  }
}

// g_y = scale * (g_z - s1/n - xhat * s2/n)
template <typename T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ ga, const T* __restrict__ y, const float* __restrict__ aux,
                                    const float* __restrict__ sums, float slope, float inv_count,
                                    T* __restrict__ gy, long long total8, int C, long long group_stride = 0) {
  __builtin_amdgcn_s_setprio(JCK_BN_PRIO);
  ga += (long long)blockIdx.y * total8 * 8; y += (long long)blockIdx.y * total8 * 8; gy += (long long)blockIdx.y * total8 * 8;   // group
  aux += (long long)blockIdx.y * 4 * C; sums += (long long)blockIdx.y * group_stride;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total8; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)((i * 8) & (C - 1));
    float vg[8], vy[8];
    ld8(ga + i * 8, vg);
    ld8(y + i * 8, vy);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float sc = aux[c + k];
      const float z = vy[k] * sc + aux[C + c + k];
      const float gz = z > 0.f ? vg[k] : slope * vg[k];
      const float xh = (vy[k] - aux[2 * C + c + k]) * aux[3 * C + c + k];
      vg[k] = sc * (gz - sums[c + k] * inv_count - xh * (sums[C + c + k] * inv_count));
    }
    st8(gy + i * 8, vg);
  }
}

// Backward sums + apply as ONE launch (round 5, the backward twin of bn_fwd_fused_kernel): the three-launch form's middle launch -
// bn_bwd_sums_kernel, 10-13 us on the main stream's chain between two streaming kernels - is done by every workgroup for itself.
// A workgroup owns a 64-channel slice of a pixel range of ONE group, sums the partial rows bn_bwd_reduce_kernel wrote for ITS 64
// channels ([blk][2][C] floats: <= 256 rows of 512 bytes, L2-resident; double accumulation in an order fixed by the geometry),
// keeps s1/n, s2/n in LDS and streams its pixels.  The workgroups with blockIdx.x == 0 write the group's sums (CGAN's v-chain
// reads them); those of group 0 also form the parameter gradients: dgamma += sum_{g < grad_groups} s2_g, dbeta likewise, the
// groups walked in order.  Its first rows are requested before the sums are formed.
template <typename T>
__global__ __launch_bounds__(BNF_THREADS) void bn_bwd_apply_fused_kernel(const T* __restrict__ ga, const T* __restrict__ y, const float* __restrict__ aux,
                                                                         const float* __restrict__ partial, int nblk, float* __restrict__ sums,
                                                                         float* __restrict__ dgamma, float* __restrict__ dbeta, float slope,
                                                                         float inv_count, T* __restrict__ gy, long long rows, int C,
                                                                         long long group_stride, int grad_groups) {
  __shared__ double part[8][128];
  __shared__ float tab[2][64];
  __builtin_amdgcn_s_setprio(JCK_BN_PRIO);
  const int t = threadIdx.x, s = blockIdx.y, grp = blockIdx.z;
  const long long g0 = (long long)grp * rows * C;              // first element of this group
  constexpr int BU = 2;
  const int u8 = t & 7;
  const long long rstep = (long long)gridDim.x * 32, rfirst = (long long)blockIdx.x * 32 + (t >> 3);
  Raw8<T> rg[BU], ry[BU];
#pragma unroll
  for (int u = 0; u < BU; ++u)
    if (rfirst + u * rstep < rows) {
      ldraw(ga + g0 + (rfirst + u * rstep) * C + s * 64 + u8 * 8, rg[u]);
      ldraw(y + g0 + (rfirst + u * rstep) * C + s * 64 + u8 * 8, ry[u]);
    }
  // ---- the slice's sums of group g: thread = (row lane 0..7, stat, 4 channels); result of thread t < 64: (s1, s2) of channel s*64 + t
  const int q4 = t & 31, rl = t >> 5;
  auto slice_sums = [&](int g, double& s1, double& s2) {
    const float* col = partial + (long long)g * group_stride + (q4 >> 4) * C + s * 64 + (q4 & 15) * 4;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    constexpr int RU = 16;
    for (int r0 = rl; r0 < nblk; r0 += 8 * RU) {
      f32x4 v[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u)
        if (r0 + u * 8 < nblk) v[u] = *reinterpret_cast<const f32x4*>(col + (long long)(r0 + u * 8) * 2 * C);
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        if (r0 + u * 8 >= nblk) break;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += (double)v[u][i];
      }
    }
    __syncthreads();                                           // part[] of the previous call has been read
#pragma unroll
    for (int i = 0; i < 4; ++i) part[rl][q4 * 4 + i] = acc[i];
    __syncthreads();
    s1 = 0.0; s2 = 0.0;
    if (t < 64) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { s1 += part[k][t]; s2 += part[k][64 + t]; }
    }
  };
  double s1d, s2d;
  slice_sums(grp, s1d, s2d);
  if (t < 64) {
    const float s1 = (float)s1d, s2 = (float)s2d;
    tab[0][t] = s1 * inv_count; tab[1][t] = s2 * inv_count;
    if (blockIdx.x == 0) {
      const int c = s * 64 + t;
      sums[(long long)grp * group_stride + c] = s1;
      sums[(long long)grp * group_stride + C + c] = s2;
    }
  }
  __syncthreads();
  // ---- stream: 8 threads per row slice, 32 rows per pass, BU rows of this thread in flight
  float sc[8], sh[8], mu[8], is[8], m1[8], m2[8];
  {
    const float* ax = aux + (long long)grp * 4 * C + s * 64 + u8 * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      sc[k] = ax[k]; sh[k] = ax[C + k]; mu[k] = ax[2 * C + k]; is[k] = ax[3 * C + k];
      m1[k] = tab[0][u8 * 8 + k]; m2[k] = tab[1][u8 * 8 + k];
    }
  }
  for (long long r0 = rfirst; r0 < rows; r0 += rstep * BU) {
    if (r0 != rfirst) {
#pragma unroll
      for (int u = 0; u < BU; ++u)
        if (r0 + u * rstep < rows) {
          ldraw(ga + g0 + (r0 + u * rstep) * C + s * 64 + u8 * 8, rg[u]);
          ldraw(y + g0 + (r0 + u * rstep) * C + s * 64 + u8 * 8, ry[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < BU; ++u) {
      if (r0 + u * rstep >= rows) break;
      float vg[8], vy[8];
      unraw(rg[u], vg);
      unraw(ry[u], vy);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float z = vy[k] * sc[k] + sh[k];
        const float gz = z > 0.f ? vg[k] : slope * vg[k];
        const float xh = (vy[k] - mu[k]) * is[k];
        vg[k] = sc[k] * (gz - m1[k] - xh * m2[k]);
      }
      st8(gy + g0 + (r0 + u * rstep) * C + s * 64 + u8 * 8, vg);
    }
  }
  // ---- parameter gradients: the first workgroup of a slice in group 0 walks the gradient groups in order
  if (blockIdx.x == 0 && grp == 0 && (dgamma || dbeta) && grad_groups > 0) {      // (uniform per workgroup)
    float dg = (float)s2d, db = (float)s1d;
    for (int g = 1; g < grad_groups; ++g) {
      double a1, a2;
      slice_sums(g, a1, a2);
      dg += (float)a2; db += (float)a1;
    }
    if (t < 64) {
      const int c = s * 64 + t;
      if (dgamma) dgamma[c] += dg;
      if (dbeta) dbeta[c] += db;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// D head: logit[n] = <a4[n,:], w[:]>,  p = sigmoid, BCE(p, t) with the -100 clamp, ds = dL/dlogit
//   mode 0: loss = mean BCE;  dp = (p - t) / max(p(1-p), 1e-12) / B;  ds = dp * p(1-p)      (ATen formulas)
//   mode 1: gradient-penalty pass, grad_outputs = ones:          ds = p(1-p)
// ------------------------------------------------------------------------------------------------------
// Several batches stacked row-wise (the real | fake | penalty groups of the batched D pass) go through ONE launch: group
// g = row / rows_per_group has its own target, mode and scalar slots; prob / ds are indexed by the stacked row, the scalar table
// by the row inside the group.
struct HeadGroups { float target[4]; int mode[4], slot_loss[4], slot_p[4]; int rows_per_group; };
template <typename T>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ a4, const float* __restrict__ w, int K,
                                                       const float* __restrict__ bias, const HeadGroups hg, float invB,
                                                       float* __restrict__ prob, float* __restrict__ ds,
                                                       float* __restrict__ scal, int scal_ld, T* __restrict__ g_out = nullptr) {
  // g_out (optional): the input gradient of the row, g_out[n][k] = ds[n] * w[k] (head_bwd_fused_kernel's / head_dgrad_kernel's
  // product), written by the workgroup that has just formed ds[n] - one launch less on the step's serial chain
  __shared__ float sm[4];
  __shared__ float sds;
  const int grp = blockIdx.x / hg.rows_per_group, nrow = blockIdx.x - grp * hg.rows_per_group;
  const float target = hg.target[grp];
  const int mode = hg.mode[grp], slot_loss = hg.slot_loss[grp], slot_p = hg.slot_p[grp];
  const T* x = a4 + (long long)blockIdx.x * K;
  float s = 0.f;
  for (int i = threadIdx.x * 8; i < K; i += 256 * 8) {
    float v[8];
    ld8(x + i, v);
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + i), w1 = *reinterpret_cast<const f32x4*>(w + i + 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) s += v[k] * w0[k] + v[4 + k] * w1[k];
  }
  s = block_sum256(s, sm);
  if (threadIdx.x == 0) {
    if (bias) s += bias[0];
    const float p = 1.f / (1.f + expf(-s));
    prob[blockIdx.x] = p;
    const float pq = p * (1.f - p);
    if (mode == 0) {
      const float lp = fmaxf(logf(p), -100.f), lq = fmaxf(logf(1.f - p), -100.f);
      const float loss = -(target * lp + (1.f - target) * lq);
      const float dp = (p - target) / fmaxf(pq, 1e-12f) * invB;
      sds = dp * pq;
      if (slot_loss >= 0) scal[(long long)slot_loss * scal_ld + nrow] = loss;
    } else {
      sds = pq;
    }
    ds[blockIdx.x] = sds;
    if (slot_p >= 0) scal[(long long)slot_p * scal_ld + nrow] = p;
  }
  if (!g_out) return;
  __syncthreads();
  const float d = sds;
  T* gr = g_out + (long long)blockIdx.x * K;
  for (int i = threadIdx.x * 8; i < K; i += 256 * 8) {
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + i), w1 = *reinterpret_cast<const f32x4*>(w + i + 4);
    float o[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) { o[k] = d * w0[k]; o[4 + k] = d * w1[k]; }
    st8(gr + i, o);
  }
}

// g_a4[n][k] = ds[n] * w[k]
template <typename T>
__global__ void head_dgrad_kernel(const float* __restrict__ ds, const float* __restrict__ w, int K, T* __restrict__ g,
                                  long long total8) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total8; i += (long long)gridDim.x * blockDim.x) {
    const long long e = i * 8;
    const int k = (int)(e % K);
    const float d = ds[e / K];
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = d * w[k + j];
    st8(g + e, v);
  }
}

// part[y][k] = sum over this workgroup's images of ds[n] * a4[n][k]       (packed (h,w,c) order)
// grid (K/8/64, NS): 64 column units x 4 image lanes per workgroup, images strided over lanes and gridDim.y; the NS
// partial rows are summed in order by head_part_reduce_kernel (no float atomics: bitwise reproducible)
template <typename T>
__global__ __launch_bounds__(256) void head_wgrad_kernel(const float* __restrict__ ds, const T* __restrict__ a4, int B, int K,
                                                         float* __restrict__ part) {
  __shared__ float red[4][64][9];
  const int u = threadIdx.x & 63, ln = threadIdx.x >> 6;
  const int k = (blockIdx.x * 64 + u) * 8;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (k < K)
    for (int n = blockIdx.y * 4 + ln; n < B; n += gridDim.y * 4) {
      float v[8];
      ld8(a4 + (long long)n * K + k, v);
      const float d = ds[n];
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += d * v[j];
    }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[ln][u][j] = s[j];
  __syncthreads();
  if (ln == 0 && k < K) {
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (red[0][u][j] + red[1][u][j]) + (red[2][u][j] + red[3][u][j]);
    st8(part + (long long)blockIdx.y * K + k, o);
  }
}

// out (+)= sum_y part[y][k], y in order.  C == 0: out[k] (packed order, `accumulate` selects = / +=);  C > 0: the PyTorch
// layout of a [1][C][4][4] conv weight gradient, out[c*16 + t] += sum_y part[y][t*C + c]
static __global__ void head_part_reduce_kernel(const float* __restrict__ part, int NS, int K, int C, float* __restrict__ out,
                                               int accumulate) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  float s = 0.f;
  for (int y = 0; y < NS; ++y) s += part[(long long)y * K + k];
  if (C > 0) { const int t = k / C, c = k % C; out[c * 16 + t] += s; }
  else out[k] = accumulate ? out[k] + s : s;
}

// D.conv5 backward in one launch: g_a4[n][k] = ds[n] * w[k] and the partial weight-gradient rows
// part[y][k] = sum_{n of workgroup row y} ds[n] * a4[n][k]  (k = t*C + c is the packed (h,w,c) order); head_part_reduce_kernel
// sums the rows in order into the PyTorch-layout gradient.  grid (K/8/64, NS): 64 column units x 4 image lanes per workgroup.
template <typename T>
__global__ __launch_bounds__(256) void head_bwd_fused_kernel(const float* __restrict__ ds, const float* __restrict__ w,
                                                             const T* __restrict__ a4, int B, int K, int C,
                                                             T* __restrict__ g, float* __restrict__ grad, int Bmore = 0) {
  // Bmore: rows [B, B + Bmore) that only get their input gradient (the penalty group behind the loss groups of the batched D
  // pass): one launch for both; the rows < B meet the same lanes in the same order, so the partial sums keep their bits
  __shared__ float red[4][64][9];
  const int u = threadIdx.x & 63, ln = threadIdx.x >> 6;
  const int k = (blockIdx.x * 64 + u) * 8;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (k < K) {
    float wv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) wv[j] = w[k + j];
    for (int n = blockIdx.y * 4 + ln; n < B + Bmore; n += gridDim.y * 4) {
      const float d = ds[n];
      if (grad && n < B) {
        float v[8];
        ld8(a4 + (long long)n * K + k, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += d * v[j];
      }
      if (g) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = d * wv[j];
        st8(g + (long long)n * K + k, o);
      }
    }
  }
  if (!grad) return;
#pragma unroll
  for (int j = 0; j < 8; ++j) red[ln][u][j] = s[j];
  __syncthreads();
  if (ln == 0 && k < K) {
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (red[0][u][j] + red[1][u][j]) + (red[2][u][j] + red[3][u][j]);
    st8(grad + (long long)blockIdx.y * K + k, o);          // `grad` is the partial buffer [NS][K] here
  }
}

// ------------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam single-tensor algorithm, amsgrad=False, weight_decay=0) over a flat arena
// ------------------------------------------------------------------------------------------------------
// hp (optional): device float[2] = {step_size, bc2_sqrt} of THIS step, written by adam_hp_kernel before the step is enqueued
// - the whole-step engine passes its per-step scalars this way so that a captured hipGraph of the step carries no
// per-step kernel argument; the values are the same host-computed floats either way.
// hp[0..1] = Adam's per-step scalars; hp[4..7] (as uint32) = {noise seed lo, hi, step, 0} for the in-kernel Philox noise.
// The same launch draws the step's SMALL random inputs when the caller hands over none (perf mode): z [nz] ~ N(0,1)
// (train/dcgan_trainer.py:168), alpha [nalpha] ~ U[0,1) (:111), CGAN's Dropout keep masks [nmask] in {0,1} with P(keep) = keep_p
// (model/CGAN.py:105) - Philox4x32-10, counter = (index/4, tensor id 8 / 9 / 10, step), key = seed.  No ATen launch is left
// in the step, and a captured step replays with fresh draws without any copy into static buffers.
struct StepRng { float* z; long long nz; float* alpha; long long nalpha; float* masks; long long nmask; float keep_p; float* zero; long long nzero;
                 float* zbig[2]; long long nzbig[2];
                 void* zpad; int zd, zp, zpad_f32; };     // zpad (optional): the same z as the rows [nz / zd][zp] of G.conv1's operand (bf16 | fp32)
// (zero / nzero: a small buffer the same launch clears - the engine's per-step accumulator rows, instead of a memset node;
// zbig: up to two large 16-byte aligned ranges, counts % 4 == 0 - D's gradient arena and CGAN's permuted Linear gradient, which
// D.zero_grad() (train/dcgan_trainer.py:155) would clear with a launch of its own a few microseconds later)
__device__ __forceinline__ float u01(unsigned x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }
static __global__ void adam_hp_kernel(float* __restrict__ hp, float step_size, float bc2_sqrt, unsigned seed_lo, unsigned seed_hi,
                                      unsigned step, const StepRng r) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    hp[0] = step_size; hp[1] = bc2_sqrt;
    unsigned* w = reinterpret_cast<unsigned*>(hp + 4);
    w[0] = seed_lo; w[1] = seed_hi; w[2] = step; w[3] = 0u;
  }
  const long long q0 = (r.nz + 3) / 4, q1 = (r.nalpha + 3) / 4, q2 = (r.nmask + 3) / 4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < r.nzero; i += (long long)gridDim.x * blockDim.x) r.zero[i] = 0.f;
#pragma unroll
  for (int b = 0; b < 2; ++b)
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < (r.nzbig[b] >> 2); i += (long long)gridDim.x * blockDim.x)
      reinterpret_cast<f32x4*>(r.zbig[b])[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < q0 + q1 + q2; i += (long long)gridDim.x * blockDim.x) {
    unsigned o[4];
    if (i < q0) {                                                     // four normals: two Box-Muller pairs
      philox4x32_10((unsigned)i, (unsigned)(i >> 32), 8u, step, seed_lo, seed_hi, o);
      const float r0 = sqrtf(-2.0f * logf(u01(o[0]))), r1 = sqrtf(-2.0f * logf(u01(o[2])));
      float s0, c0, s1, c1;
      sincosf(6.283185307179586f * u01(o[1]), &s0, &c0);
      sincosf(6.283185307179586f * u01(o[3]), &s1, &c1);
      const float v[4] = {r0 * c0, r0 * s0, r1 * c1, r1 * s1};
      for (int k = 0; k < 4; ++k)
        if (i * 4 + k < r.nz) {
          r.z[i * 4 + k] = v[k];
          // ... and, in the same launch, into G.conv1's operand rows (what pad_rows_kernel would copy a launch later; the padding
          // columns [zd, zp) are never written by anybody: they keep the zeros of the zero-initialised workspace)
          if (r.zpad) {
            const long long e = i * 4 + k, b = e / r.zd;
            const int c = (int)(e - b * r.zd);
            if (r.zpad_f32) reinterpret_cast<float*>(r.zpad)[b * r.zp + c] = v[k];
            else stf(reinterpret_cast<bf16_t*>(r.zpad) + b * r.zp + c, v[k]);
          }
        }
    } else if (i < q0 + q1) {
      const long long j = i - q0;
      philox4x32_10((unsigned)j, (unsigned)(j >> 32), 9u, step, seed_lo, seed_hi, o);
      for (int k = 0; k < 4; ++k)
        if (j * 4 + k < r.nalpha) r.alpha[j * 4 + k] = (float)(o[k] >> 8) * (1.0f / 16777216.0f);      // [0, 1)
    } else {
      const long long j = i - q0 - q1;
      philox4x32_10((unsigned)j, (unsigned)(j >> 32), 10u, step, seed_lo, seed_hi, o);
      for (int k = 0; k < 4; ++k)
        if (j * 4 + k < r.nmask) r.masks[j * 4 + k] = (float)(o[k] >> 8) * (1.0f / 16777216.0f) < r.keep_p ? 1.f : 0.f;
    }
  }
}
__device__ __forceinline__ void adam_one(float& pi, float gi_raw, float& mi_io, float& vi_io, float w1, float beta2, float omb2,
                                         float eps, float step_size, float bc2_sqrt, float grad_scale) {
  const float gi = gi_raw * grad_scale;
  // exp_avg.lerp_(grad, 1-beta1): weight 0.5 takes ATen's "end - (end-start)*(1-w)" branch when w >= 0.5
  const float mi = (w1 < 0.5f) ? mi_io + w1 * (gi - mi_io) : gi - (gi - mi_io) * (1.f - w1);
  const float vi = vi_io * beta2 + (omb2 * gi) * gi;
  mi_io = mi;
  vi_io = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  pi = pi - step_size * (mi / denom);
}
// four elements per thread (16-byte loads and stores: the arenas are 16-byte aligned); the last n % 4 elements one by one
static __global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float w1 /*1-beta1*/, float beta2, float omb2 /*1-beta2*/,
                            float eps, float step_size, float bc2_sqrt, float grad_scale, const float* __restrict__ hp = nullptr,
                            int vec = 1, float* __restrict__ zero = nullptr, long long nzero4 = 0,
                            const unsigned* __restrict__ skip_if = nullptr) {
  if (hp) { step_size = hp[0]; bc2_sqrt = hp[1]; }
  // skip_if (the engine's grid-barrier error word): a resident launch of this step went on with incomplete sums - the gradients
  // are invalid, so parameters and moments stay as they are (the host learns of it at jck_engine_check); the zero range below is
  // still cleared: the next pass accumulates into it
  const bool skip = skip_if && *skip_if != 0u;
  // zero (optional): a 16-byte aligned range of nzero4 float4 the same launch clears - the OTHER network's gradient arena, whose
  // zero_grad() (train/dcgan_trainer.py:182) is the next thing in the step
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nzero4; i += (long long)gridDim.x * blockDim.x)
    reinterpret_cast<f32x4*>(zero)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (skip) return;
  const long long n4 = vec ? (n >> 2) : 0;                 // vec = 0: a pointer is not 16-byte aligned -> element by element
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i], mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
    const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float pk = pv[k], mk = mv[k], vk = vv[k];
      adam_one(pk, gv[k], mk, vk, w1, beta2, omb2, eps, step_size, bc2_sqrt, grad_scale);
      pv[k] = pk; mv[k] = mk; vv[k] = vk;
    }
    reinterpret_cast<f32x4*>(p)[i] = pv; reinterpret_cast<f32x4*>(m)[i] = mv; reinterpret_cast<f32x4*>(v)[i] = vv;
  }
  for (long long i = (n4 << 2) + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    adam_one(p[i], g[i], m[i], v[i], w1, beta2, omb2, eps, step_size, bc2_sqrt, grad_scale);
}

// ------------------------------------------------------------------------------------------------------
// weight packing (fp32 PyTorch layout [Cs][Cb][4][4] -> GEMM operand layouts in bf16 (fast) or fp32 (parity))
// ------------------------------------------------------------------------------------------------------
// down: wp[cs][ (kh*4+kw)*CbPad + cb ]   rows cs in [0, CsPad)
__device__ __forceinline__ float pack_down_val(const float* __restrict__ w, int Cs, int Cb, int logCbPad, long long i) {
  const long long K = 16ll << logCbPad;
  const int cs = (int)(i / K);
  const int k = (int)(i % K);
  const int t = k >> logCbPad, cb = k & ((1 << logCbPad) - 1);
  return (cs < Cs && cb < Cb) ? w[((long long)cs * Cb + cb) * 16 + t] : 0.f;
}
template <typename W>
__global__ void pack_down_kernel(const float* __restrict__ w, int Cs, int Cb, int CsPad, int logCbPad, W* __restrict__ wp) {
  const long long K = 16ll << logCbPad, total = (long long)CsPad * K;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    stf(wp + i, pack_down_val(w, Cs, Cb, logCbPad, i));
}

// up: wp[phase][cb][ (th*2+tw)*Cs + cs ], rows cb in [0, CbPad); phase = ph*2+pw;
// output row 2q+ph takes input rows q + DY[ph][th] through kernel rows KH[ph][th]
static __device__ __constant__ int c_up_k[2][2] = {{1, 3}, {0, 2}};
__device__ __forceinline__ float pack_up_val(const float* __restrict__ w, int Cs, int Cb, int CbPad, long long i) {
  const long long K = 4ll * Cs, per = (long long)CbPad * K;
  const int phase = (int)(i / per);
  const long long r = i % per;
  const int cb = (int)(r / K);
  const int k = (int)(r % K);
  const int t = k / Cs, cs = k % Cs;
  const int kh = c_up_k[phase >> 1][t >> 1], kw = c_up_k[phase & 1][t & 1];
  return cb < Cb ? w[((long long)cs * Cb + cb) * 16 + kh * 4 + kw] : 0.f;
}
template <typename W>
__global__ void pack_up_kernel(const float* __restrict__ w, int Cs, int Cb, int CbPad, W* __restrict__ wp) {
  const long long total = 4ll * CbPad * 4 * Cs;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    stf(wp + i, pack_up_val(w, Cs, Cb, CbPad, i));
}

// up, 3/4-channel output (G.conv5, dgrad of D.conv1): all four output parities in ONE 16-row operand,
// wp[phase*4 + c][ (dyi*3+dxi)*Cs + cs ] over the 9 input offsets dy,dx in {-1,0,1}; unused (phase, offset) pairs are 0
__device__ __forceinline__ float pack_up16_val(const float* __restrict__ w, int Cs, int Cb, long long i) {
  const long long K = 9ll * Cs;
  const int r = (int)(i / K), k = (int)(i % K);
  const int phase = r >> 2, c = r & 3, t9 = k / Cs, cs = k % Cs;
  const int dy = t9 / 3 - 1, dx = t9 % 3 - 1, ph = phase >> 1, pw = phase & 1;
  // output row 2q+ph reads input row q+dy through kernel row kh:  ph=0: (0 -> 1), (-1 -> 3);  ph=1: (+1 -> 0), (0 -> 2)
  const int kh = ph == 0 ? (dy == 0 ? 1 : (dy == -1 ? 3 : -1)) : (dy == 1 ? 0 : (dy == 0 ? 2 : -1));
  const int kw = pw == 0 ? (dx == 0 ? 1 : (dx == -1 ? 3 : -1)) : (dx == 1 ? 0 : (dx == 0 ? 2 : -1));
  return (c < Cb && kh >= 0 && kw >= 0) ? w[((long long)cs * Cb + c) * 16 + kh * 4 + kw] : 0.f;
}
template <typename W>
__global__ void pack_up16_kernel(const float* __restrict__ w, int Cs, int Cb, W* __restrict__ wp) {
  const long long total = 16 * 9ll * Cs;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    stf(wp + i, pack_up16_val(w, Cs, Cb, i));
}

// G.conv1 (ConvTranspose on a 1x1 input): wp[(kh*4+kw)*Co + co][ci], ci in [0, CiPad)
__device__ __forceinline__ float pack_g1_val(const float* __restrict__ w, int Ci, int Co, int CiPad, long long i) {
  const int ci = (int)(i % CiPad);
  const long long r = i / CiPad;
  const int co = (int)(r % Co), t = (int)(r / Co);
  return ci < Ci ? w[((long long)ci * Co + co) * 16 + t] : 0.f;
}
template <typename W>
__global__ void pack_g1_kernel(const float* __restrict__ w, int Ci, int Co, int CiPad, W* __restrict__ wp) {
  const long long total = 16ll * Co * CiPad;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    stf(wp + i, pack_g1_val(w, Ci, Co, CiPad, i));
}

// D.conv5 (a dot product per image): wp[(kh*4+kw)*C + c] fp32
static __global__ void pack_head_kernel(const float* __restrict__ w, int C, float* __restrict__ wp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 16 * C) return;
  const int t = i / C, c = i % C;
  wp[i] = w[c * 16 + t];
}

// Every packed operand of one network in ONE launch (the optimiser step is followed by 9-10 repacks; as separate launches
// their ~4.5 us floor each left the GPU idle for ~45 us twice per step).  A thread owns one (output channel, input channel)
// pair of a conv weight: it reads the pair's 16 taps (64 contiguous bytes) once and writes the 16 operand elements, with
// the lane-fastest index chosen per layout so that every store instruction covers contiguous elements.  Padding rows and
// channels are never written: they stay zero from the zero-initialised workspace.  256 pairs per workgroup.
#define PACK_MAX_JOBS 12
#define PACK_CHUNK 256
struct PackJob { const float* w; void* wp; long long total; int kind, a, b, c; };   // kind: 0 down 1 up 2 up16 3 g1 4 head
struct PackJobs { PackJob j[PACK_MAX_JOBS]; int first_chunk[PACK_MAX_JOBS + 1]; int n; };
template <typename W>
__device__ __forceinline__ void pack_multi_block(const PackJobs& jobs, int bx) {
  int ji = 0;
  while (ji + 1 < jobs.n && bx >= jobs.first_chunk[ji + 1]) ++ji;
  const PackJob& J = jobs.j[ji];
  const long long u = (long long)(bx - jobs.first_chunk[ji]) * PACK_CHUNK + threadIdx.x;
  if (u >= J.total) return;
  W* wp = reinterpret_cast<W*>(J.wp);
  float v[16];
  auto load16 = [&](const float* src) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(src + 4 * q);
      v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
    }
  };
  switch (J.kind) {
    case 0: {   // down: a = Cs, b = Cb, c = logCbPad; unit = (cs, cb), cb fastest; wp[cs][t*CbPad + cb]
      const int Cb = J.b, cs = (int)(u / Cb), cb = (int)(u % Cb);
      load16(J.w + ((long long)cs * Cb + cb) * 16);
      W* d = wp + (((long long)cs * 16) << J.c) + cb;
#pragma unroll
      for (int t = 0; t < 16; ++t) stf(d + ((long long)t << J.c), v[t]);
    } break;
    case 1: {   // up: a = Cs, b = Cb, c = CbPad; unit = (cb, cs), cs fastest; wp[phase][cb][t4*Cs + cs]
      const int Cs = J.a, Cb = J.b, cb = (int)(u / Cs), cs = (int)(u % Cs);
      load16(J.w + ((long long)cs * Cb + cb) * 16);
      const long long K = 4ll * Cs, per = (long long)J.c * K;
#pragma unroll
      for (int phase = 0; phase < 4; ++phase)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int kh = c_up_k[phase >> 1][t >> 1], kw = c_up_k[phase & 1][t & 1];
          stf(wp + phase * per + (long long)cb * K + (long long)t * Cs + cs, v[kh * 4 + kw]);
        }
    } break;
    case 2: {   // up16: a = Cs, b = Cb; unit = (c, cs), cs fastest; wp[phase*4 + c][t9*Cs + cs], unused (phase, offset) pairs stay 0
      const int Cs = J.a, Cb = J.b, c = (int)(u / Cs), cs = (int)(u % Cs);
      load16(J.w + ((long long)cs * Cb + c) * 16);
      const long long K = 9ll * Cs;
#pragma unroll
      for (int phase = 0; phase < 4; ++phase)
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9) {
          const int dy = t9 / 3 - 1, dx = t9 % 3 - 1, ph = phase >> 1, pw = phase & 1;
          const int kh = ph == 0 ? (dy == 0 ? 1 : (dy == -1 ? 3 : -1)) : (dy == 1 ? 0 : (dy == 0 ? 2 : -1));
          const int kw = pw == 0 ? (dx == 0 ? 1 : (dx == -1 ? 3 : -1)) : (dx == 1 ? 0 : (dx == 0 ? 2 : -1));
          if (kh >= 0 && kw >= 0) stf(wp + (long long)(phase * 4 + c) * K + (long long)t9 * Cs + cs, v[kh * 4 + kw]);
        }
    } break;
    case 3: {   // g1: a = Ci, b = Co, c = CiPad; unit = (co, ci), ci fastest; wp[(t*Co + co)][ci]
      const int Ci = J.a, Co = J.b, co = (int)(u / Ci), ci = (int)(u % Ci);
      load16(J.w + ((long long)ci * Co + co) * 16);
#pragma unroll
      for (int t = 0; t < 16; ++t) stf(wp + ((long long)t * Co + co) * J.c + ci, v[t]);
    } break;
    default: {  // head: a = C; unit = c; wp[t*C + c] fp32
      const int c = (int)u;
      load16(J.w + (long long)c * 16);
#pragma unroll
      for (int t = 0; t < 16; ++t) reinterpret_cast<float*>(J.wp)[(long long)t * J.a + c] = v[t];
    } break;
  }
}
template <typename W>
__global__ __launch_bounds__(256) void pack_multi_kernel(const PackJobs jobs) { pack_multi_block<W>(jobs, (int)blockIdx.x); }

template <typename T>
__global__ void cast_f32_kernel(const float* __restrict__ in, T* __restrict__ out, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    stf(out + i, in[i]);
}
template <typename T>
__global__ void cast_to_f32_kernel(const T* __restrict__ in, float* __restrict__ out, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = ldf(in + i);
}

// z [B][Ci] fp32 -> [B][CiPad] T, zero padded (G.conv1 operand)
template <typename T>
__global__ void pad_rows_kernel(const float* __restrict__ in, int B, int Ci, int CiPad, T* __restrict__ out) {
  const long long total = (long long)B * CiPad;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % CiPad);
    const long long b = i / CiPad;
    stf(out + i, c < Ci ? in[b * Ci + c] : 0.f);
  }
}

// acc rows: 0 loss_real 1 loss_fake 2 loss_g 3 p(real) 4 p(fake) 5 p(g phase) 6 (||g||-1)^2, one entry per image
// out: loss_d, loss_g, D(x), D(G(z))_1, D(G(z))_2, gp, loss_real, loss_fake      (train/dcgan_trainer.py:179,192-193)
// End of a step in ONE launch: the deferred BatchNorm running-stat records of D's four layers (blockIdx.y = layer, same
// recurrence as sequential momentum updates) and the logged scalars (blockIdx.y = number of layers).
struct TailLayer { const float* rec; float* rm; float* rv; long long* nbt; int C; };
// acc is the per-image table [7][acc_ld] (head_fwd / gp_norm write one entry per image): each row is summed here in a fixed
// order (thread-strided partial sums, wavefront shuffles, 4 wave totals), so the logged scalars are bitwise reproducible.
struct TailJobs { TailLayer l[5]; int nl; int npass; float momentum; const float* acc; int acc_ld, B; float invB, lambda_gp; float* out; };
__device__ __forceinline__ void step_tail_block(const TailJobs& t, int bx, int by) {
  if (by == t.nl) {
    if (bx != 0) return;
    __shared__ float sm[4];
    float tot[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      float s = 0.f;
      for (int n = threadIdx.x; n < t.B; n += 256) s += t.acc[(long long)q * t.acc_ld + n];
      tot[q] = block_sum256(s, sm);
    }
    if (threadIdx.x == 0) {
      const float lr = tot[0] * t.invB, lf = tot[1] * t.invB, gp = tot[6] * t.invB;
      t.out[0] = (lr + lf) + t.lambda_gp * gp;
      t.out[1] = tot[2] * t.invB;
      t.out[2] = tot[3] * t.invB;
      t.out[3] = tot[4] * t.invB;
      t.out[4] = tot[5] * t.invB;
      t.out[5] = gp;
      t.out[6] = lr;
      t.out[7] = lf;
    }
    return;
  }
  const TailLayer& L = t.l[by];
  const int c = bx * blockDim.x + threadIdx.x;
  if (c == 0 && L.nbt) *L.nbt += t.npass;
  if (c >= L.C) return;
  float rm = L.rm[c], rv = L.rv[c];
  for (int p = 0; p < t.npass; ++p) {
    rm = (1.f - t.momentum) * rm + t.momentum * L.rec[(long long)p * 2 * L.C + c];
    rv = (1.f - t.momentum) * rv + t.momentum * L.rec[(long long)p * 2 * L.C + L.C + c];
  }
  L.rm[c] = rm;
  L.rv[c] = rv;
}
// G's repack and the end of the step in one launch (two independent jobs at the very end of the step, each near the launch
// floor): workgroups [0, pack_chunks) repack, the next tail_x * (nl + 1) are step_tail_block's grid (tail_x, nl + 1) row-major
template <typename W>
__global__ __launch_bounds__(256) void pack_tail_kernel(const PackJobs jobs, int pack_chunks, const TailJobs t, int tail_x) {
  if ((int)blockIdx.x < pack_chunks) { pack_multi_block<W>(jobs, (int)blockIdx.x); return; }
  const int q = (int)blockIdx.x - pack_chunks;
  step_tail_block(t, q % tail_x, q / tail_x);
}

// ------------------------------------------------------------------------------------------------------
// CGAN pieces (model/CGAN.py:79-162): Linear layers as GEMM operands, label embedding, concat, dropout, GP gradient
// ------------------------------------------------------------------------------------------------------
// nn.Linear weight [N][K] fp32 -> wp[row][col] of type W:
//   transpose = 0: wp[n][k'] (forward operand, rows padded to NPad, K padded to KPad)
//   transpose = 1: wp[k'][n] (dgrad operand, rows padded to KRows, columns padded to NPadK)
// k' is the position in OUR activation order: the first hw*C columns of the reference are (c, hw) (NCHW flatten,
// model/CGAN.py:119-120), ours are (hw, c) (NHWC); columns >= hw*C (the label embedding) keep their place.
template <typename W>
__global__ void pack_linear_kernel(const float* __restrict__ w, int N, int K, int rows, int cols, int transpose, int permC,
                                   int permHW, W* __restrict__ wp) {
  const long long total = (long long)rows * cols;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i % cols);
    const int n = transpose ? c : r, kp = transpose ? r : c;
    float v = 0.f;
    if (n < N && kp < K) {
      int k = kp;
      if (permC > 0 && kp < permC * permHW) { const int hw = kp / permC, ch = kp % permC; k = ch * permHW + hw; }
      v = w[(long long)n * K + k];
    }
    stf(wp + i, v);
  }
}

// both layouts of one nn.Linear weight in ONE launch: blockIdx.y = 0 the forward operand wp0[rows0][cols0], 1 the dgrad operand
// wp1[rows1][cols1] (pack_linear_kernel with transpose = blockIdx.y)
template <typename W>
__global__ void pack_linear_pair_kernel(const float* __restrict__ w, int N, int K, int rows0, int cols0, W* __restrict__ wp0, int rows1,
                                        int cols1, W* __restrict__ wp1, int permC, int permHW) {
  const int transpose = blockIdx.y;
  const int rows = transpose ? rows1 : rows0, cols = transpose ? cols1 : cols0;
  W* wp = transpose ? wp1 : wp0;
  const long long total = (long long)rows * cols;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i % cols);
    const int n = transpose ? c : r, kp = transpose ? r : c;
    float v = 0.f;
    if (n < N && kp < K) {
      int k = kp;
      if (permC > 0 && kp < permC * permHW) { const int hw = kp / permC, ch = kp % permC; k = ch * permHW + hw; }
      v = w[(long long)n * K + k];
    }
    stf(wp + i, v);
  }
}

// grad[n][k] (+)= gp[n][k'] with the same column permutation (k' -> k) as pack_linear_kernel
static __global__ void unperm_linear_grad_kernel(const float* __restrict__ gp, int N, int K, int ldp, int permC, int permHW,
                                                 float* __restrict__ grad, int accumulate) {
  const long long total = (long long)N * K;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i / K), k = (int)(i % K);
    int kp = k;
    if (permC > 0 && k < permC * permHW) { const int ch = k / permHW, hw = k % permHW; kp = hw * permC + ch; }
    const float v = gp[(long long)n * ldp + kp];
    grad[i] = accumulate ? grad[i] + v : v;
  }
}

// label embedding: e[b][j] = lrelu(sum_i W[j][i] * onehot[b][i] + bias[j])   (model/CGAN.py:83-84,111)
// labels: int64 one-hot [B][100] as the reference feeds them (preprocess/cgan_data_preprocessor.py:11-16)
template <typename T>
__global__ void label_embed_fwd_kernel(const long long* __restrict__ labels, const float* __restrict__ W,
                                       const float* __restrict__ bias, float slope, int B, int NI, int NO, T* __restrict__ cbuf,
                                       int ld, int col0, float* __restrict__ pre, int label_period = 0) {
  // one workgroup per image: the label row goes to LDS once; its zeros (99 of 100 for a one-hot row) are skipped - a
  // workgroup-uniform branch, and adding an exact 0 * W changes nothing
  extern __shared__ float lab[];                      // [NI]
  const int b = blockIdx.x;
  const int lb = label_period > 0 ? b % label_period : b;       // rows of several batches that share one label tensor
  for (int k = threadIdx.x; k < NI; k += blockDim.x) lab[k] = (float)labels[(long long)lb * NI + k];
  __syncthreads();
  for (int j = threadIdx.x; j < NO; j += blockDim.x) {
    float s = bias[j];
    for (int k = 0; k < NI; ++k) {
      const float l = lab[k];
      if (l != 0.f) s += W[j * NI + k] * l;
    }
    if (pre) pre[b * NO + j] = s;
    stf(cbuf + (long long)b * ld + col0 + j, s > 0.f ? s : slope * s);
  }
}
// dW[j][i] += sum_b ue[b][j]*act'(pre) * onehot[b][i];  db[j] += sum_b ...      ue taken from gc[b][col0 + j] (type T)
// One workgroup per label column i (threads over the output units j; workgroup NI sums the bias gradient): the label
// labels[b][i] is the same for the whole workgroup, so rows with a zero label - all but ~B/NI of them for one-hot labels -
// are skipped by a uniform branch, and the masked gradient row is read contiguously along j.  (One thread per (j, i)
// looping over b took 116 us at B = 256, one workgroup per j 32 us.)
template <typename T>
__global__ __launch_bounds__(256) void label_embed_bwd_kernel(const T* __restrict__ gc, int ld, int col0, const float* __restrict__ pre,
                                                              const long long* __restrict__ labels, float slope, int B, int NI,
                                                              int NO, float* __restrict__ dW, float* __restrict__ db, int label_period = 0) {
  extern __shared__ float lcol[];                     // [B]: this label column, fetched by all threads at once
  const int i = blockIdx.x;
  if (i >= NI) {
    // bias gradient: workgroups NI.. take 16 units each, 16 row lanes per unit with four rows in flight, LDS reduce in lane order
    // (every row contributes).  Round 2's 64 units x 4 lanes walked B / 4 rows one dependent load pair at a time: 37 us of the
    // kernel's 37 at 512 rows.
    __shared__ float red[16][16];
    const int jl = threadIdx.x & 15, ln = threadIdx.x >> 4, j = (i - NI) * 16 + jl;
    float s = 0.f;
    if (j < NO) {
      auto term = [&](int b) { return ldf(gc + (long long)b * ld + col0 + j) * (pre[b * NO + j] > 0.f ? 1.f : slope); };
      int b = ln;
      for (; b + 48 < B; b += 64) {
        const float v0 = term(b), v1 = term(b + 16), v2 = term(b + 32), v3 = term(b + 48);
        s += v0; s += v1; s += v2; s += v3;
      }
      for (; b < B; b += 16) s += term(b);
    }
    red[ln][jl] = s;
    __syncthreads();
    if (ln == 0 && j < NO) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += red[k][jl];
      db[j] += t;
    }
    return;
  }
  __shared__ int nnz;
  int* rows = reinterpret_cast<int*>(lcol + B);       // [B]: indices of the rows with a non-zero label, in order
  for (int b = threadIdx.x; b < B; b += blockDim.x) lcol[b] = (float)labels[(long long)(label_period > 0 ? b % label_period : b) * NI + i];
  if (threadIdx.x == 0) nnz = 0;
  __syncthreads();
  // ordered compaction (deterministic summation order) by wavefront ballots, 256 rows per round; a serial loop of one thread
  // over B LDS words was half of this kernel's 21 us
  __shared__ int wcnt[4];
  for (int b0 = 0; b0 < B; b0 += 256) {
    const int b = b0 + (int)threadIdx.x;
    const bool hit = b < B && lcol[b] != 0.f;
    const unsigned long long m = __ballot(hit);
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    if (ln == 0) wcnt[wv] = __popcll(m);
    __syncthreads();
    int off = nnz;
    for (int w2 = 0; w2 < wv; ++w2) off += wcnt[w2];
    if (hit) rows[off + __popcll(m & ((1ull << ln) - 1ull))] = b;
    __syncthreads();
    if (threadIdx.x == 0) nnz += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    __syncthreads();
  }
  const int n = nnz;
  for (int j = threadIdx.x; j < NO; j += blockDim.x) {
    float s = 0.f;
#pragma unroll 4
    for (int k = 0; k < n; ++k) {                     // no branch: the loads of several rows are in flight together
      const int b = rows[k];
      s += lcol[b] * (ldf(gc + (long long)b * ld + col0 + j) * (pre[b * NO + j] > 0.f ? 1.f : slope));
    }
    dW[j * NI + i] += s;
  }
}

// cbuf[b][0..K0) = a4[b][0..K0)  (columns [K0, ld) are written by the label embedding / stay zero)
template <typename T>
__global__ void concat_rows_kernel(const T* __restrict__ a4, int K0, T* __restrict__ cbuf, int ld, long long total8) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total8; i += (long long)gridDim.x * blockDim.x) {
    const long long e = i * 8;
    const long long b = e / K0;
    const int k = (int)(e % K0);
    *reinterpret_cast<Raw8<T>*>(cbuf + b * ld + k) = *reinterpret_cast<const Raw8<T>*>(a4 + e);
  }
}
// the reverse: a4-shaped gradient out of the first K0 columns
template <typename T>
__global__ void split_rows_kernel(const T* __restrict__ gc, int ld, int K0, T* __restrict__ ga4, long long total8) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total8; i += (long long)gridDim.x * blockDim.x) {
    const long long e = i * 8;
    const long long b = e / K0;
    const int k = (int)(e % K0);
    *reinterpret_cast<Raw8<T>*>(ga4 + e) = *reinterpret_cast<const Raw8<T>*>(gc + b * ld + k);
  }
}

// y = x * mask * scale   (nn.Dropout(0.25) forward and backward: model/CGAN.py:105,122; mask in {0,1}, scale = 1/(1-p))
template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, const float* __restrict__ mask, float scale, T* __restrict__ y,
                               long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    stf(y + i, ldf(x + i) * mask[i] * scale);
}

// column sums of a [B][N] matrix of T into fp32 (bias gradients): db[j] += sum_b g[b][j].  16 columns x 16 row lanes per
// workgroup, four rows in flight per thread, LDS reduce over the lanes in lane order (deterministic).  (Round 2: 64 columns x 4
// lanes, one row in flight - 128 dependent load round trips for the 512 rows of CGAN's batched head, 21 us.)
#define COLSUM_COLS 16
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ g, int B, int N, int ld, float* __restrict__ db) {
  __shared__ float red[16][COLSUM_COLS];
  const int jl = threadIdx.x & 15, ln = threadIdx.x >> 4, j = blockIdx.x * COLSUM_COLS + jl;
  float s = 0.f;
  if (j < N) {
    int b = ln;
    for (; b + 48 < B; b += 64) {
      const float v0 = ldf(g + (long long)b * ld + j), v1 = ldf(g + (long long)(b + 16) * ld + j);
      const float v2 = ldf(g + (long long)(b + 32) * ld + j), v3 = ldf(g + (long long)(b + 48) * ld + j);
      s += v0; s += v1; s += v2; s += v3;
    }
    for (; b < B; b += 16) s += ldf(g + (long long)b * ld + j);
  }
  red[ln][jl] = s;
  __syncthreads();
  if (ln == 0 && j < N) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][jl];
    db[j] += t;
  }
}

// u = coef * (norm - 1) / norm * g      gradient of  lambda * mean_n (||g_n|| - 1)^2  w.r.t. g  (coef = 2*lambda/B)
template <typename T>
__global__ void gp_grad_kernel(const T* __restrict__ g, const float* __restrict__ norms, float coef, int per_image,
                               T* __restrict__ u, long long total4) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    const float nr = norms[(i * 4) / per_image];
    const float f = nr > 0.f ? coef * (nr - 1.f) / nr : 0.f;
    float v[4];
    ld4(g + i * 4, v);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] *= f;
    st4(u + i * 4, v);
  }
}

// second-order head terms of the penalty (see tests/test_gp_double_backward_math.py):
//   uds[n] = <ughd[n,:], w2>;  rs[n] = uds * (1-2p) * p(1-p);  dw2[j] += sum_n ds[n] * ughd[n][j]     (ds = p(1-p))
// (the dw2 sum is formed afterwards by head_wgrad_kernel + head_part_reduce_kernel from pq[n] = ds[n]: no atomics)
template <typename T>
__global__ __launch_bounds__(256) void gp_head2_kernel(const T* __restrict__ ughd, const float* __restrict__ w2,
                                                       const float* __restrict__ prob, int B, int K, float* __restrict__ rs,
                                                       float* __restrict__ pq_out) {
  __shared__ float sm[4];
  const int n = blockIdx.x;
  float s = 0.f;
  for (int j = threadIdx.x; j < K; j += 256) s += ldf(ughd + (long long)n * K + j) * w2[j];
  s = block_sum256(s, sm);
  if (threadIdx.x == 0) {
    const float p = prob[n], pq = p * (1.f - p);
    rs[n] = s * (1.f - 2.f * p) * pq;
    pq_out[n] = pq;
  }
}

// The same stretch of the penalty's double backward as ONE launch, one workgroup per row of 256 columns (thread = column):
// linear_finish_kernel (split-K sum of the v-chain's Linear product, Dropout) -> ughd, gp_head2_kernel (rs, pq), head_dgrad_kernel
// (g_hd = rs * w2) and the Dropout backward (g_h) - the arithmetic and the rounding points of the four launches.
template <typename T>
__global__ __launch_bounds__(256) void cg_gp_head_mid_kernel(const float* __restrict__ slab, int Z, long long zstride, const float* __restrict__ mask,
                                                             float scale, T* __restrict__ ughd, const float* __restrict__ w2,
                                                             const float* __restrict__ prob, float* __restrict__ rs, float* __restrict__ pq_out,
                                                             T* __restrict__ g_hd, T* __restrict__ g_h) {
  __shared__ float sm[4];
  __shared__ float srs;
  const int j = threadIdx.x, n = blockIdx.x;
  const long long i = (long long)n * 256 + j;
  float s = 0.f;
  for (int z = 0; z < Z; ++z) s += slab[(long long)z * zstride + i];
  T t;
  stf(&t, s * mask[i] * scale);
  ughd[i] = t;
  float d = 0.f;
  d += ldf(&t) * w2[j];
  d = block_sum256(d, sm);
  if (j == 0) {
    const float p = prob[n], pq = p * (1.f - p);
    const float r = d * (1.f - 2.f * p) * pq;
    rs[n] = r;
    pq_out[n] = pq;
    srs = r;
  }
  __syncthreads();
  T tg;
  stf(&tg, srs * w2[j]);
  g_hd[i] = tg;
  stf(g_h + i, ldf(&tg) * mask[i] * scale);
}

// ------------------------------------------------------------------------------------------------------
// Second-order BatchNorm terms of the back-propagated gradient penalty (CGAN, train/cgan_trainer.py:200-203).
// Notation (tests/test_gp_double_backward_math.py): first backward gy = (gamma/sigma) (gz - m1 - xhat*m2),
// gz = g_a*act'(z), m1 = mean gz, m2 = mean gz*xhat.  The adjoint sweep carries v = adj(gy) forward through D
// ("v-chain") and then the usual reverse sweep with two extra inputs per layer (xdir, sigexp).
// Generic per-channel reduction: NA accumulators per channel, per-workgroup partials [blk][NA][C] (no atomics).
// ------------------------------------------------------------------------------------------------------
struct BnAux { const float* aux; int C; };       // aux = scale | shift | mean | invstd

// MODE 1 (v-chain):  acc = { v, v*xhat, v*gy }             inputs a = v, b = y, c = gy
// MODE 2 (reverse):  acc = { uz, uz*xhat, xdir, xdir*xhat } inputs a = ua, b = y, c = xdir;  uz = ua*act'(z)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn2_reduce_kernel(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ c,
                                                         const float* __restrict__ aux, float slope, float* __restrict__ partial,
                                                         long long rows, int C) {
  constexpr int NA = MODE == 1 ? 3 : 4;
  extern __shared__ float lsum[];                         // [rstep][NA][C]
  const int upr = C >> 3;
  const int u = threadIdx.x % upr, r0 = threadIdx.x / upr, rstep = 256 / upr;
  const int ch = u * 8;
  float sc[8], sh[8], mu[8], is[8], acc[NA][8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    sc[k] = aux[ch + k]; sh[k] = aux[C + ch + k]; mu[k] = aux[2 * C + ch + k]; is[k] = aux[3 * C + ch + k];
#pragma unroll
    for (int q = 0; q < NA; ++q) acc[q][k] = 0.f;
  }
  // two rows of this thread's sequence in flight, accumulated in sequence order (same sums, bit for bit; see bn_bwd_reduce_kernel)
  constexpr int UNR = 2;
  const long long stride = (long long)gridDim.x * rstep;
  for (long long r = (long long)blockIdx.x * rstep + r0; r < rows; r += stride * UNR) {
    float va[UNR][8], vb[UNR][8], vc[UNR][8];
#pragma unroll
    for (int q = 0; q < UNR; ++q) {
      const long long rq = r + q * stride;
      if (rq < rows) { ld8(a + rq * C + ch, va[q]); ld8(b + rq * C + ch, vb[q]); ld8(c + rq * C + ch, vc[q]); }
    }
#pragma unroll
    for (int q = 0; q < UNR; ++q) {
      if (r + q * stride >= rows) break;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float xh = (vb[q][k] - mu[k]) * is[k];
        if (MODE == 1) {
          acc[0][k] += va[q][k]; acc[1][k] += va[q][k] * xh; acc[2][k] += va[q][k] * vc[q][k];
        } else {
          const float z = vb[q][k] * sc[k] + sh[k];
          const float uz = z > 0.f ? va[q][k] : slope * va[q][k];
          acc[0][k] += uz; acc[1][k] += uz * xh; acc[2][k] += vc[q][k]; acc[NA - 1][k] += vc[q][k] * xh;
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < NA; ++q)
#pragma unroll
    for (int k = 0; k < 8; ++k) lsum[(r0 * NA + q) * C + ch + k] = acc[q][k];
  __syncthreads();
  for (int i = threadIdx.x; i < NA * C; i += 256) {
    float t = 0.f;
    for (int r = 0; r < rstep; ++r) t += lsum[r * NA * C + i];
    partial[(long long)blockIdx.x * NA * C + i] = t;
  }
}

// sums[i] = sum over workgroups of partial[blk][i], i < NA*C (a multiple of 4): one workgroup per 4 outputs, 256 lanes over
// the partials with 16-byte loads (one thread per output looping over 256 strided partials took 64 us for C = 64)
// mode 1 (v-chain): dgamma[c] += sum(v*gy)[c] / gamma[c] (the direct parameter gradient of the v-chain);  mode 2 (reverse sweep):
// dgamma[c] += sum uz*xhat, dbeta[c] += sum uz - by the thread that owns that sum, instead of a launch of their own
static __global__ __launch_bounds__(256) void bn2_sums_kernel(const float* __restrict__ partial, int nblk, int NA, int C,
                                                              float* __restrict__ sums, int mode = 0, const float* __restrict__ gamma = nullptr,
                                                              float* __restrict__ dgamma = nullptr, float* __restrict__ dbeta = nullptr) {
  __shared__ float sh[4][4];
  const int i0 = blockIdx.x * 4, n = NA * C;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int k = threadIdx.x; k < nblk; k += 256) s += *reinterpret_cast<const f32x4*>(partial + (long long)k * n + i0);
#pragma unroll
  for (int j = 0; j < 4; ++j) s[j] = wave_sum(s[j]);
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) sh[threadIdx.x >> 6][j] = s[j];
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int i = i0 + threadIdx.x;
    const float t = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
    sums[i] = t;
    if (mode == 1 && i >= 2 * C && i < 3 * C) dgamma[i - 2 * C] += t / gamma[i - 2 * C];
    if (mode == 2 && i < C) dbeta[i] += t;
    if (mode == 2 && i >= C && i < 2 * C) dgamma[i - C] += t;
  }
}

// v-chain apply:  u = act'(z) * (gamma/sigma) (v - mean v - xhat * mean(v xhat))                 (may overwrite v)
//                 xdir = -(gamma/sigma) (v*m2 + gz * mean(v xhat)),  gz = gy*sigma/gamma + m1 + xhat*m2
// s1 = sums of the FIRST backward (sum gz, sum gz*xhat), s3 = {sum v, sum v*xhat, sum v*gy}
template <typename T>
__global__ void bn2_vchain_apply_kernel(const T* __restrict__ v, const T* __restrict__ y, const T* __restrict__ gy,
                                        const float* __restrict__ aux, const float* __restrict__ s1, const float* __restrict__ s3,
                                        float slope, float inv_count, T* __restrict__ u, T* __restrict__ xdir, long long total8, int C) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total8; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)((i * 8) & (C - 1));
    float vv[8], vy[8], vg[8], ou[8], ox[8];
    ld8(v + i * 8, vv);
    ld8(y + i * 8, vy);
    ld8(gy + i * 8, vg);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float sc = aux[c + k];                           // gamma / sigma
      const float z = vy[k] * sc + aux[C + c + k];
      const float xh = (vy[k] - aux[2 * C + c + k]) * aux[3 * C + c + k];
      const float m1 = s1[c + k] * inv_count, m2 = s1[C + c + k] * inv_count;
      const float mv = s3[c + k] * inv_count, mvx = s3[C + c + k] * inv_count;
      const float gz = vg[k] / sc + m1 + xh * m2;
      ox[k] = -sc * (vv[k] * m2 + gz * mvx);
      const float ugz = sc * (vv[k] - mv - xh * mvx);
      ou[k] = z > 0.f ? ugz : slope * ugz;
    }
    st8(u + i * 8, ou);
    st8(xdir + i * 8, ox);
  }
}
// reverse-sweep apply with the penalty's extra inputs:
//   q = gamma*uz + xdir;  uy = (q - mean q - xhat*mean(q xhat)) / sigma  -  (sum(v*gy)/sigma) * xhat / n
// s4 = {sum uz, sum uz*xhat, sum xdir, sum xdir*xhat};  vgy = sum v*gy (from the v-chain).  dgamma += sum uz*xhat, dbeta += sum uz
template <typename T>
__global__ void bn2_reverse_apply_kernel(const T* __restrict__ ua, const T* __restrict__ y, const T* __restrict__ xdir,
                                         const float* __restrict__ aux, const float* __restrict__ gamma, const float* __restrict__ s4,
                                         const float* __restrict__ vgy, float slope, float inv_count, T* __restrict__ uy,
                                         long long total8, int C) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total8; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)((i * 8) & (C - 1));
    float va[8], vy[8], vx[8], o[8];
    ld8(ua + i * 8, va);
    ld8(y + i * 8, vy);
    ld8(xdir + i * 8, vx);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float g = gamma[c + k], is = aux[3 * C + c + k];
      const float z = vy[k] * aux[c + k] + aux[C + c + k];
      const float xh = (vy[k] - aux[2 * C + c + k]) * is;
      const float uz = z > 0.f ? va[k] : slope * va[k];
      const float q = g * uz + vx[k];
      const float mq = (g * s4[c + k] + s4[2 * C + c + k]) * inv_count;
      const float mqx = (g * s4[C + c + k] + s4[3 * C + c + k]) * inv_count;
      o[k] = (q - mq - xh * mqx) * is - vgy[c + k] * is * xh * inv_count;
    }
    st8(uy + i * 8, o);
  }
}
// The second-order chains as two launches instead of three (round 5; the backward twin bn_bwd_apply_fused_kernel above): the apply
// kernel's workgroups own a 64-channel slice of a pixel range and sum the partial rows of bn2_reduce_kernel for THEIR channels
// themselves ([blk][NA][C] floats; double accumulation in an order fixed by the geometry) - bn2_sums_kernel, ~6 us on the main
// stream's chain eight times per CGAN step, is gone.  The first workgroup of a slice writes the sums (ws[q*C + c], as bn2_sums_kernel
// did: the reverse sweep reads the v-chain's) and the direct parameter gradients.
// -> LDS tot[q][j] = sum over the nblk rows of accumulator q, channel slice*64 + j   (q < NA)
template <int NA>
__device__ __forceinline__ void bn2_slice_sums(const float* __restrict__ partial, int nblk, int C, int slice, double (*part)[NA * 64], float (*tot)[64]) {
  constexpr int NCOL = NA * 16, RL = 256 / NCOL;            // float4 columns of a slice row, row lanes (5 or 4)
  const int t = threadIdx.x, q4 = t % NCOL, rl = t / NCOL;
  if (rl < RL) {
    const float* col = partial + (q4 >> 4) * C + slice * 64 + (q4 & 15) * 4;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    constexpr int RU = 8;
    for (int r0 = rl; r0 < nblk; r0 += RL * RU) {
      f32x4 v[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u)
        if (r0 + u * RL < nblk) v[u] = *reinterpret_cast<const f32x4*>(col + (long long)(r0 + u * RL) * NA * C);
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        if (r0 + u * RL >= nblk) break;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += (double)v[u][i];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) part[rl][q4 * 4 + i] = acc[i];
  }
  __syncthreads();
  if (t < NA * 64) {
    double sd = 0.0;
#pragma unroll
    for (int k = 0; k < RL; ++k) sd += part[k][t];
    tot[t >> 6][t & 63] = (float)sd;
  }
  __syncthreads();
}

// v-chain apply (bn2_vchain_apply_kernel's arithmetic) with the sums of ITS slice formed in the prologue
template <typename T>
__global__ __launch_bounds__(256) void bn2_vchain_apply_fused_kernel(const T* __restrict__ v, const T* __restrict__ y, const T* __restrict__ gy,
                                                                     const float* __restrict__ aux, const float* __restrict__ s1,
                                                                     const float* __restrict__ partial, int nblk, float* __restrict__ ws,
                                                                     const float* __restrict__ gamma, float* __restrict__ dgamma, float slope,
                                                                     float inv_count, T* __restrict__ u, T* __restrict__ xdir, long long rows, int C) {
  __shared__ double part[5][3 * 64];
  __shared__ float tot[3][64];
  __builtin_amdgcn_s_setprio(JCK_BN_PRIO);
  const int t = threadIdx.x, s = blockIdx.y, u8 = t & 7;
  const long long rstep = (long long)gridDim.x * 32, rfirst = (long long)blockIdx.x * 32 + (t >> 3);
  constexpr int BU = 2;
  Raw8<T> rv[BU], ry[BU], rg[BU];
#pragma unroll
  for (int q = 0; q < BU; ++q)
    if (rfirst + q * rstep < rows) {
      const long long e = (rfirst + q * rstep) * C + s * 64 + u8 * 8;
      ldraw(v + e, rv[q]); ldraw(y + e, ry[q]); ldraw(gy + e, rg[q]);
    }
  bn2_slice_sums<3>(partial, nblk, C, s, part, tot);
  if (blockIdx.x == 0 && t < 64) {
    const int c = s * 64 + t;
    ws[c] = tot[0][t]; ws[C + c] = tot[1][t]; ws[2 * C + c] = tot[2][t];
    if (dgamma) dgamma[c] += tot[2][t] / gamma[c];
  }
  float sc[8], sh[8], mu[8], is[8], m1[8], m2[8], mv[8], mvx[8];
  {
    const int c0 = s * 64 + u8 * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      sc[k] = aux[c0 + k]; sh[k] = aux[C + c0 + k]; mu[k] = aux[2 * C + c0 + k]; is[k] = aux[3 * C + c0 + k];
      m1[k] = s1[c0 + k] * inv_count; m2[k] = s1[C + c0 + k] * inv_count;
      mv[k] = tot[0][u8 * 8 + k] * inv_count; mvx[k] = tot[1][u8 * 8 + k] * inv_count;
    }
  }
  for (long long r0 = rfirst; r0 < rows; r0 += rstep * BU) {
    if (r0 != rfirst) {
#pragma unroll
      for (int q = 0; q < BU; ++q)
        if (r0 + q * rstep < rows) {
          const long long e = (r0 + q * rstep) * C + s * 64 + u8 * 8;
          ldraw(v + e, rv[q]); ldraw(y + e, ry[q]); ldraw(gy + e, rg[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < BU; ++q) {
      if (r0 + q * rstep >= rows) break;
      float vv[8], vy[8], vg[8], ou[8], ox[8];
      unraw(rv[q], vv); unraw(ry[q], vy); unraw(rg[q], vg);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float z = vy[k] * sc[k] + sh[k];
        const float xh = (vy[k] - mu[k]) * is[k];
        const float gz = vg[k] / sc[k] + m1[k] + xh * m2[k];
        ox[k] = -sc[k] * (vv[k] * m2[k] + gz * mvx[k]);
        const float ugz = sc[k] * (vv[k] - mv[k] - xh * mvx[k]);
        ou[k] = z > 0.f ? ugz : slope * ugz;
      }
      const long long e = (r0 + q * rstep) * C + s * 64 + u8 * 8;
      st8(u + e, ou);
      st8(xdir + e, ox);
    }
  }
}

// reverse-sweep apply (bn2_reverse_apply_kernel's arithmetic) with the sums of ITS slice formed in the prologue
template <typename T>
__global__ __launch_bounds__(256) void bn2_reverse_apply_fused_kernel(const T* __restrict__ ua, const T* __restrict__ y, const T* __restrict__ xdir,
                                                                      const float* __restrict__ aux, const float* __restrict__ gamma,
                                                                      const float* __restrict__ partial, int nblk, float* __restrict__ ws,
                                                                      const float* __restrict__ vgy, float* __restrict__ dgamma,
                                                                      float* __restrict__ dbeta, float slope, float inv_count, T* __restrict__ uy,
                                                                      long long rows, int C) {
  __shared__ double part[4][4 * 64];
  __shared__ float tot[4][64];
  __builtin_amdgcn_s_setprio(JCK_BN_PRIO);
  const int t = threadIdx.x, s = blockIdx.y, u8 = t & 7;
  const long long rstep = (long long)gridDim.x * 32, rfirst = (long long)blockIdx.x * 32 + (t >> 3);
  constexpr int BU = 2;
  Raw8<T> ra[BU], ry[BU], rx[BU];
#pragma unroll
  for (int q = 0; q < BU; ++q)
    if (rfirst + q * rstep < rows) {
      const long long e = (rfirst + q * rstep) * C + s * 64 + u8 * 8;
      ldraw(ua + e, ra[q]); ldraw(y + e, ry[q]); ldraw(xdir + e, rx[q]);
    }
  bn2_slice_sums<4>(partial, nblk, C, s, part, tot);
  if (blockIdx.x == 0 && t < 64) {
    const int c = s * 64 + t;
    ws[c] = tot[0][t]; ws[C + c] = tot[1][t]; ws[2 * C + c] = tot[2][t]; ws[3 * C + c] = tot[3][t];
    if (dbeta) dbeta[c] += tot[0][t];
    if (dgamma) dgamma[c] += tot[1][t];
  }
  float g[8], sc[8], sh[8], mu[8], is[8], mq[8], mqx[8], kv[8];
  {
    const int c0 = s * 64 + u8 * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      g[k] = gamma[c0 + k];
      sc[k] = aux[c0 + k]; sh[k] = aux[C + c0 + k]; mu[k] = aux[2 * C + c0 + k]; is[k] = aux[3 * C + c0 + k];
      mq[k] = (g[k] * tot[0][u8 * 8 + k] + tot[2][u8 * 8 + k]) * inv_count;
      mqx[k] = (g[k] * tot[1][u8 * 8 + k] + tot[3][u8 * 8 + k]) * inv_count;
      kv[k] = vgy[c0 + k];
    }
  }
  for (long long r0 = rfirst; r0 < rows; r0 += rstep * BU) {
    if (r0 != rfirst) {
#pragma unroll
      for (int q = 0; q < BU; ++q)
        if (r0 + q * rstep < rows) {
          const long long e = (r0 + q * rstep) * C + s * 64 + u8 * 8;
          ldraw(ua + e, ra[q]); ldraw(y + e, ry[q]); ldraw(xdir + e, rx[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < BU; ++q) {
      if (r0 + q * rstep >= rows) break;
      float va[8], vy[8], vx[8], o[8];
      unraw(ra[q], va); unraw(ry[q], vy); unraw(rx[q], vx);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float z = vy[k] * sc[k] + sh[k];
        const float xh = (vy[k] - mu[k]) * is[k];
        const float uz = z > 0.f ? va[k] : slope * va[k];
        const float qq = g[k] * uz + vx[k];
        o[k] = (qq - mq[k] - xh * mqx[k]) * is[k] - kv[k] * is[k] * xh * inv_count;
      }
      st8(uy + (r0 + q * rstep) * C + s * 64 + u8 * 8, o);
    }
  }
}

// h[b][j] = sum_z slab[z][b][j] + bias[j];  hd = h * mask * scale     (Linear(8392,256) finish + Dropout, model/CGAN.py:104-105,122)
template <typename T>
__global__ void linear_finish_kernel(const float* __restrict__ slab, int Z, long long zstride, const float* __restrict__ bias,
                                     const float* __restrict__ mask, float scale, T* __restrict__ h, T* __restrict__ hd, int B, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * N) return;
  float s = bias ? bias[i % N] : 0.f;
  for (int z = 0; z < Z; ++z) s += slab[(long long)z * zstride + i];
  if (h) stf(h + i, s);
  if (hd) stf(hd + i, mask ? s * mask[i] * scale : s);
}

// The middle of CGAN's head as ONE launch, one workgroup per row of N = 256 columns (thread = column): linear_finish_kernel
// (split-K sum + bias, Dropout), head_fwd_kernel (Linear(256,1) + sigmoid + BCE, ds), head_dgrad_kernel (g_hd = ds * w2) and the
// Dropout backward (g_h = g_hd * mask * scale) - four launches at the launch floor on a serial chain.  Same arithmetic in the same
// order as the four (the dot product is formed by threads 0..31 over 8 columns each, then block_sum256, as head_fwd_kernel
// does it; every value is rounded to T where the separate launches stored it).
template <typename T>
__global__ __launch_bounds__(256) void cg_head_mid_kernel(const float* __restrict__ slab, int Z, long long zstride, const float* __restrict__ bias1,
                                                          const float* __restrict__ mask, float scale, T* __restrict__ h, T* __restrict__ hd,
                                                          const float* __restrict__ w2, const float* __restrict__ bias2, const HeadGroups hg,
                                                          float invB, float* __restrict__ prob, float* __restrict__ ds, float* __restrict__ scal,
                                                          int scal_ld, T* __restrict__ g_hd, T* __restrict__ g_h) {
  __shared__ float sm[4];
  __shared__ __attribute__((aligned(16))) float shd[256];
  __shared__ float sds;
  const int j = threadIdx.x;
  const long long i = (long long)blockIdx.x * 256 + j;
  float s = bias1 ? bias1[j] : 0.f;
  for (int z = 0; z < Z; ++z) s += slab[(long long)z * zstride + i];
  stf(h + i, s);
  T t;
  stf(&t, s * mask[i] * scale);
  hd[i] = t;
  shd[j] = ldf(&t);
  __syncthreads();
  float d = 0.f;
  if (j < 32) {
    const float* v = shd + j * 8;
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(w2 + j * 8), w1 = *reinterpret_cast<const f32x4*>(w2 + j * 8 + 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) d += v[k] * w0[k] + v[4 + k] * w1[k];
  }
  d = block_sum256(d, sm);
  if (j == 0) {
    const int grp = blockIdx.x / hg.rows_per_group, nrow = blockIdx.x - grp * hg.rows_per_group;
    const float target = hg.target[grp];
    const int mode = hg.mode[grp], slot_loss = hg.slot_loss[grp], slot_p = hg.slot_p[grp];
    if (bias2) d += bias2[0];
    const float p = 1.f / (1.f + expf(-d));
    prob[blockIdx.x] = p;
    const float pq = p * (1.f - p);
    float dsv;
    if (mode == 0) {
      const float lp = fmaxf(logf(p), -100.f), lq = fmaxf(logf(1.f - p), -100.f);
      const float loss = -(target * lp + (1.f - target) * lq);
      const float dp = (p - target) / fmaxf(pq, 1e-12f) * invB;
      dsv = dp * pq;
      if (slot_loss >= 0) scal[(long long)slot_loss * scal_ld + nrow] = loss;
    } else {
      dsv = pq;
    }
    ds[blockIdx.x] = dsv;
    sds = dsv;
    if (slot_p >= 0) scal[(long long)slot_p * scal_ld + nrow] = p;
  }
  __syncthreads();
  T tg;
  stf(&tg, sds * w2[j]);
  g_hd[i] = tg;
  stf(g_h + i, ldf(&tg) * mask[i] * scale);
}

// out[0] += sum_i x[i]   (bias gradient of a 1-output Linear)
static __global__ __launch_bounds__(256) void sum_vec_kernel(const float* __restrict__ x, int n, float* __restrict__ out) {
  __shared__ float sm[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += x[i];
  s = block_sum256(s, sm);
  if (threadIdx.x == 0) out[0] += s;
}
// G input of the conditional GAN: [z | one-hot label] -> [B][CiPad] T   (model/CGAN.py:154-155: cat([x, labels], 1), int64 -> float)
template <typename T>
__global__ void cgan_z_kernel(const float* __restrict__ z, const long long* __restrict__ labels, int B, int NZ, int NL, int CiPad,
                              T* __restrict__ out) {
  const long long total = (long long)B * CiPad;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % CiPad);
    const long long b = i / CiPad;
    float v = 0.f;
    if (c < NZ) v = z[b * NZ + c];
    else if (c < NZ + NL) v = (float)labels[b * NL + (c - NZ)];
    stf(out + i, v);
  }
}
