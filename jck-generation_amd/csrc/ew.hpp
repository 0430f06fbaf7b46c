// Memory-bound kernels of the DCGAN/CGAN step: layout conversion + instance noise, BatchNorm
// (statistics finalise, normalise+activation, backward reduce/apply), D head (4x4 valid conv to one
// logit + sigmoid + BCE with the -100 log clamp + gradient), tanh backward, gradient-penalty norm,
// Adam, weight packing.  Everything is 16-byte vectorised along the NHWC channel axis; per-channel
// reductions use registers -> LDS atomics -> one global atomic per channel per workgroup.
#pragma once
#include "common.hpp"

// ------------------------------------------------------------------------------------------------------
// image prep: out[n][p][0..3] = keep * img[n][c][p] + mix * noise[n][c][p]   (NCHW f32 -> NHWC4 T)
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void img_prep_kernel(const float* __restrict__ img, const float* __restrict__ noise, float keep, float mix,
                                T* __restrict__ out, int N, int HW) {
  const long long total = (long long)N * HW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / HW;
    const int px = (int)(i - n * HW);
    float v[4];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const long long s = (n * 3 + c) * HW + px;
      float x = keep * img[s];
      if (noise) x += mix * noise[s];
      v[c] = x;
    }
    v[3] = 0.f;
    st4(out + i * 4, v);
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
                                  T* __restrict__ out, int N, int HW) {
  const long long total = (long long)N * HW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / HW;
    const int px = (int)(i - n * HW);
    float v[4];
    ld4(x + i * 4, v);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float r = keep * v[c];
      if (noise) r += mix * noise[(n * 3 + c) * HW + px];
      v[c] = r;
    }
    v[3] = 0.f;
    st4(out + i * 4, v);
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

// gradient penalty: sum_n (||g[n]||_2 - 1)^2 -> scal[slot]      one workgroup per image
template <typename T>
__global__ __launch_bounds__(256) void gp_norm_kernel(const T* __restrict__ g, int per_image4, float* __restrict__ scal,
                                                      int slot, float* __restrict__ norms) {
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
    atomicAdd(scal + slot, (nr - 1.f) * (nr - 1.f));
  }
}

// ------------------------------------------------------------------------------------------------------
// BatchNorm (training mode always - the reference never switches G/D to eval)
// ------------------------------------------------------------------------------------------------------
// stats: [slots][2][C] partial sums / sums of squares written by the GEMM epilogue (every slot complete).
// aux layout (floats): [0,C) scale = gamma*invstd   [C,2C) shift = beta - mean*scale
//                      [2C,3C) mean                 [3C,4C) invstd
// One workgroup per 4 channels: 256 slot-lanes, 16-byte loads, double accumulation, wavefront + LDS reduction.
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
static __global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ stats, int slots, float count,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                 long long* __restrict__ nbt, float momentum, float eps,
                                                                 float* __restrict__ aux, int C) {
  __shared__ double sh[4][8];
  const int c0 = blockIdx.x * 4;
  double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  for (int k = threadIdx.x; k < slots; k += 256) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(stats + (long long)k * 2 * C + c0);
    const f32x4 b = *reinterpret_cast<const f32x4*>(stats + (long long)k * 2 * C + C + c0);
#pragma unroll
    for (int i = 0; i < 4; ++i) { s[i] += (double)a[i]; q[i] += (double)b[i]; }
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
  if (running_mean) {
    const float unbiased = var * (count / fmaxf(count - 1.f, 1.f));
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
  }
}

// a = act(scale[c]*y + shift[c]), act = x>0 ? x : slope*x
template <typename T>
__global__ void bn_act_fwd_kernel(const T* __restrict__ y, const float* __restrict__ aux, float slope,
                                  T* __restrict__ a, long long total8, int C) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total8; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)((i * 8) & (C - 1));
    float v[8];
    ld8(y + i * 8, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float z = v[k] * aux[c + k] + aux[C + c + k];
      v[k] = z > 0.f ? z : slope * z;
    }
    st8(a + i * 8, v);
  }
}

// stage 1: partial[blk][0..C) = sum g_z,  partial[blk][C..2C) = sum g_z * xhat over this workgroup's rows,
// g_z = g_a * act'(z).  Register accumulation per (row-lane, 8-channel unit), one LDS pass over the row-lanes; no atomics.
#define BN_BWD_MAX_BLOCKS 256
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ ga, const T* __restrict__ y,
                                                            const float* __restrict__ aux, float slope,
                                                            float* __restrict__ partial, long long rows, int C) {
  extern __shared__ float lsum[];                         // [rstep][2][C]
  const int upr = C >> 3;                                 // 8-channel units per row
  const int u = threadIdx.x % upr, r0 = threadIdx.x / upr, rstep = 256 / upr;
  const int c = u * 8;
  float sc[8], sh[8], mu[8], is[8], s1[8], s2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    sc[k] = aux[c + k]; sh[k] = aux[C + c + k]; mu[k] = aux[2 * C + c + k]; is[k] = aux[3 * C + c + k];
    s1[k] = 0.f; s2[k] = 0.f;
  }
  for (long long r = (long long)blockIdx.x * rstep + r0; r < rows; r += (long long)gridDim.x * rstep) {
    float vg[8], vy[8];
    ld8(ga + r * C + c, vg);
    ld8(y + r * C + c, vy);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float z = vy[k] * sc[k] + sh[k];
      const float gz = z > 0.f ? vg[k] : slope * vg[k];
      s1[k] += gz;
      s2[k] += gz * ((vy[k] - mu[k]) * is[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) { lsum[(r0 * 2) * C + c + k] = s1[k]; lsum[(r0 * 2 + 1) * C + c + k] = s2[k]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    float t = 0.f;
    for (int r = 0; r < rstep; ++r) t += lsum[r * 2 * C + i];
    partial[(long long)blockIdx.x * 2 * C + i] = t;
  }
}

// stage 2: sums[0..2C) = sum over workgroups; dgamma += s2, dbeta += s1 (when given).  One workgroup per 4 channels.
static __global__ __launch_bounds__(256) void bn_bwd_sums_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ sums,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta, int C) {
  __shared__ float sh[4][8];
  const int c0 = blockIdx.x * 4;
  float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  for (int k = threadIdx.x; k < nblk; k += 256) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(partial + (long long)k * 2 * C + c0);
    const f32x4 b = *reinterpret_cast<const f32x4*>(partial + (long long)k * 2 * C + C + c0);
#pragma unroll
    for (int i = 0; i < 4; ++i) { s[i] += a[i]; q[i] += b[i]; }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) { s[i] = wave_sum(s[i]); q[i] = wave_sum(q[i]); }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { sh[threadIdx.x >> 6][i] = s[i]; sh[threadIdx.x >> 6][4 + i] = q[i]; }
  }
  __syncthreads();
  if (threadIdx.x >= 4) return;
  const int i = threadIdx.x, c = c0 + i;
  const float s1 = sh[0][i] + sh[1][i] + sh[2][i] + sh[3][i], s2 = sh[0][4 + i] + sh[1][4 + i] + sh[2][4 + i] + sh[3][4 + i];
  sums[c] = s1;
  sums[C + c] = s2;
  if (dgamma) dgamma[c] += s2;
  if (dbeta) dbeta[c] += s1;
}

// g_y = scale * (g_z - s1/n - xhat * s2/n)
template <typename T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ ga, const T* __restrict__ y, const float* __restrict__ aux,
                                    const float* __restrict__ sums, float slope, float inv_count,
                                    T* __restrict__ gy, long long total8, int C) {
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

// ------------------------------------------------------------------------------------------------------
// D head: logit[n] = <a4[n,:], w[:]>,  p = sigmoid, BCE(p, t) with the -100 clamp, ds = dL/dlogit
//   mode 0: loss = mean BCE;  dp = (p - t) / max(p(1-p), 1e-12) / B;  ds = dp * p(1-p)      (ATen formulas)
//   mode 1: gradient-penalty pass, grad_outputs = ones:          ds = p(1-p)
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ a4, const float* __restrict__ w, int K,
                                                       float target, int mode, float invB, float* __restrict__ prob,
                                                       float* __restrict__ ds, float* __restrict__ scal, int slot_loss,
                                                       int slot_p) {
  __shared__ float sm[4];
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
    const float p = 1.f / (1.f + expf(-s));
    prob[blockIdx.x] = p;
    const float pq = p * (1.f - p);
    if (mode == 0) {
      const float lp = fmaxf(logf(p), -100.f), lq = fmaxf(logf(1.f - p), -100.f);
      const float loss = -(target * lp + (1.f - target) * lq);
      const float dp = (p - target) / fmaxf(pq, 1e-12f) * invB;
      ds[blockIdx.x] = dp * pq;
      if (slot_loss >= 0) atomicAdd(scal + slot_loss, loss);
    } else {
      ds[blockIdx.x] = pq;
    }
    if (slot_p >= 0) atomicAdd(scal + slot_p, p);
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

// dw[k] += sum_n ds[n] * a4[n][k]       (packed (h,w,c) order; dw pre-zeroed or accumulating)
// grid (K/8/64, NS): 64 column units x 4 image lanes per workgroup, images strided over lanes and gridDim.y
template <typename T>
__global__ __launch_bounds__(256) void head_wgrad_kernel(const float* __restrict__ ds, const T* __restrict__ a4, int B, int K,
                                                         float* __restrict__ dw) {
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
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicAdd(dw + k + j, red[0][u][j] + red[1][u][j] + red[2][u][j] + red[3][u][j]);
  }
}

// ------------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam single-tensor algorithm, amsgrad=False, weight_decay=0) over a flat arena
// ------------------------------------------------------------------------------------------------------
static __global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float w1 /*1-beta1*/, float beta2, float omb2 /*1-beta2*/,
                            float eps, float step_size, float bc2_sqrt, float grad_scale) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float gi = g[i] * grad_scale;
    // exp_avg.lerp_(grad, 1-beta1): weight 0.5 takes ATen's "end - (end-start)*(1-w)" branch when w >= 0.5
    const float mi = (w1 < 0.5f) ? m[i] + w1 * (gi - m[i]) : gi - (gi - m[i]) * (1.f - w1);
    const float vi = v[i] * beta2 + (omb2 * gi) * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
  }
}

// ------------------------------------------------------------------------------------------------------
// weight packing (fp32 PyTorch layout [Cs][Cb][4][4] -> GEMM operand layouts in bf16 (fast) or fp32 (parity))
// ------------------------------------------------------------------------------------------------------
// down: wp[cs][ (kh*4+kw)*CbPad + cb ]   rows cs in [0, CsPad)
template <typename W>
__global__ void pack_down_kernel(const float* __restrict__ w, int Cs, int Cb, int CsPad, int logCbPad, W* __restrict__ wp) {
  const long long K = 16ll << logCbPad, total = (long long)CsPad * K;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cs = (int)(i / K);
    const int k = (int)(i % K);
    const int t = k >> logCbPad, cb = k & ((1 << logCbPad) - 1);
    float v = 0.f;
    if (cs < Cs && cb < Cb) v = w[((long long)cs * Cb + cb) * 16 + t];
    stf(wp + i, v);
  }
}

// up: wp[phase][cb][ (th*2+tw)*Cs + cs ], rows cb in [0, CbPad); phase = ph*2+pw;
// output row 2q+ph takes input rows q + DY[ph][th] through kernel rows KH[ph][th]
static __device__ __constant__ int c_up_k[2][2] = {{1, 3}, {0, 2}};
template <typename W>
__global__ void pack_up_kernel(const float* __restrict__ w, int Cs, int Cb, int CbPad, W* __restrict__ wp) {
  const long long K = 4ll * Cs, per = (long long)CbPad * K, total = 4 * per;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int phase = (int)(i / per);
    const long long r = i % per;
    const int cb = (int)(r / K);
    const int k = (int)(r % K);
    const int t = k / Cs, cs = k % Cs;
    const int kh = c_up_k[phase >> 1][t >> 1], kw = c_up_k[phase & 1][t & 1];
    float v = 0.f;
    if (cb < Cb) v = w[((long long)cs * Cb + cb) * 16 + kh * 4 + kw];
    stf(wp + i, v);
  }
}

// up, 3/4-channel output (G.conv5, dgrad of D.conv1): all four output parities in ONE 16-row operand,
// wp[phase*4 + c][ (dyi*3+dxi)*Cs + cs ] over the 9 input offsets dy,dx in {-1,0,1}; unused (phase, offset) pairs are 0
template <typename W>
__global__ void pack_up16_kernel(const float* __restrict__ w, int Cs, int Cb, W* __restrict__ wp) {
  const long long K = 9ll * Cs, total = 16 * K;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / K), k = (int)(i % K);
    const int phase = r >> 2, c = r & 3, t9 = k / Cs, cs = k % Cs;
    const int dy = t9 / 3 - 1, dx = t9 % 3 - 1, ph = phase >> 1, pw = phase & 1;
    // output row 2q+ph reads input row q+dy through kernel row kh:  ph=0: (0 -> 1), (-1 -> 3);  ph=1: (+1 -> 0), (0 -> 2)
    const int kh = ph == 0 ? (dy == 0 ? 1 : (dy == -1 ? 3 : -1)) : (dy == 1 ? 0 : (dy == 0 ? 2 : -1));
    const int kw = pw == 0 ? (dx == 0 ? 1 : (dx == -1 ? 3 : -1)) : (dx == 1 ? 0 : (dx == 0 ? 2 : -1));
    float v = 0.f;
    if (c < Cb && kh >= 0 && kw >= 0) v = w[((long long)cs * Cb + c) * 16 + kh * 4 + kw];
    stf(wp + i, v);
  }
}

// G.conv1 (ConvTranspose on a 1x1 input): wp[(kh*4+kw)*Co + co][ci], ci in [0, CiPad)
template <typename W>
__global__ void pack_g1_kernel(const float* __restrict__ w, int Ci, int Co, int CiPad, W* __restrict__ wp) {
  const long long total = 16ll * Co * CiPad;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % CiPad);
    const long long r = i / CiPad;
    const int co = (int)(r % Co), t = (int)(r / Co);
    float v = 0.f;
    if (ci < Ci) v = w[((long long)ci * Co + co) * 16 + t];
    stf(wp + i, v);
  }
}

// D.conv5 weight [1][512][4][4] -> f32 vector in (h, w, c) order
static __global__ void pack_head_kernel(const float* __restrict__ w, int C, float* __restrict__ wp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 16 * C) return;
  const int t = i / C, c = i % C;
  wp[i] = w[c * 16 + t];
}

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

// acc: 0 loss_real 1 loss_fake 2 loss_g 3 sum p(real) 4 sum p(fake) 5 sum p(g phase) 6 sum (||g||-1)^2
// out: loss_d, loss_g, D(x), D(G(z))_1, D(G(z))_2, gp, loss_real, loss_fake      (train/dcgan_trainer.py:179,192-193)
static __global__ void scalars_finalize_kernel(const float* __restrict__ acc, float invB, float lambda_gp, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float lr = acc[0] * invB, lf = acc[1] * invB, gp = acc[6] * invB;
    out[0] = (lr + lf) + lambda_gp * gp;
    out[1] = acc[2] * invB;
    out[2] = acc[3] * invB;
    out[3] = acc[4] * invB;
    out[4] = acc[5] * invB;
    out[5] = gp;
    out[6] = lr;
    out[7] = lf;
  }
}
