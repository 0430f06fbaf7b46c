// Inference kernels for the evaluation branch (reference metrics.py:46-51,80-94: the Inception-v3 classifier whose 100
// logits feed Inception Score and FID; train/dcgan_trainer.py:202-212).  The reference runs torchvision's network through
// ATen; here the network is a chain of these entry points (jck-generation_amd/inception.py holds the topology and the loader):
//
//   jck_conv2d_nhwc_f32   any kernel size / stride / padding, NHWC fp32, eval-mode BatchNorm folded into a per-channel
//                         scale + shift, optional ReLU, output written into a channel slice of a wider tensor (so the
//                         branches of an Inception block land directly in their place of the concatenation)
//   jck_pool2d_nhwc_f32   3x3 max pool (stride 2) / 3x3 average pool (stride 1, pad 1, count_include_pad) into a slice
//   jck_global_avgpool_nhwc_f32, jck_nchw_to_nhwc_f32
//   jck_mean_cov_f64      feature mean and covariance in fp64 on the device (FID: reference metrics.py:113-129 does this
//                         with numpy on the host)
//
// The convolution is an implicit GEMM on exact-fp32 MFMA (v_mfma_f32_16x16x4_f32: bitwise an fmaf chain, 1/16 of the bf16
// rate - accuracy, not speed, is what a metric network needs): 64 pixels x 64 channels per workgroup, k = (kh, kw, ci) in
// steps of 16; the activation tile is gathered with ci fastest across lanes (64-byte runs), the weight tile [k][co] with co
// fastest, both through padded LDS rows.
#include <cstdlib>

#include "ops_internal.hpp"

namespace {

struct ConvP {
  const float* x; const float* w; const float* scale; const float* shift; float* out;
  int N, H, W, Cin, KH, KW, SH, SW, PH, PW, OH, OW, Cout, K, M;
  int ocs, ocoff, relu;
};

constexpr int CB_M = 64, CB_N = 64, CB_K = 16;

__global__ __launch_bounds__(256) void conv2d_nhwc_f32_kernel(const ConvP p) {
  __shared__ float As[CB_K][CB_M + 1];
  __shared__ float Bs[CB_K][CB_N + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * CB_M, n0 = blockIdx.y * CB_N;
  // A loader: k = tid & 15 (fastest across lanes: consecutive ci), pixels (tid >> 4) + 16 * pass
  const int akq = tid & 15, apx = tid >> 4;
  int an[4], aiy[4], aix[4];
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int m = m0 + apx + 16 * ps;
    if (m < p.M) {
      const int ox = m % p.OW, t = m / p.OW, oy = t % p.OH;
      an[ps] = t / p.OH; aiy[ps] = oy * p.SH - p.PH; aix[ps] = ox * p.SW - p.PW;
    } else { an[ps] = -1; aiy[ps] = 0; aix[ps] = 0; }
  }
  // B loader: co = tid & 63, k rows (tid >> 6) + 4 * pass
  const int bco = tid & 63, bkq = tid >> 6;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;            // this wave's 32 x 32 part of the tile
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int kwc = p.KW * p.Cin;
  for (int k0 = 0; k0 < p.K; k0 += CB_K) {
    {
      const int k = k0 + akq;
      int kh = 0, kw = 0, ci = 0;
      const bool kok = k < p.K;
      if (kok) { kh = k / kwc; const int r = k - kh * kwc; kw = r / p.Cin; ci = r - kw * p.Cin; }
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        float v = 0.f;
        const int iy = aiy[ps] + kh, ix = aix[ps] + kw;
        if (kok && an[ps] >= 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
          v = p.x[(((long long)an[ps] * p.H + iy) * p.W + ix) * p.Cin + ci];
        As[akq][apx + 16 * ps] = v;
      }
    }
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int k = k0 + bkq + 4 * ps, co = n0 + bco;
      Bs[bkq + 4 * ps][bco] = (k < p.K && co < p.Cout) ? p.w[(long long)k * p.Cout + co] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < CB_K; kk += 4) {
      // v_mfma_f32_16x16x4_f32: lane l holds A[row l&15][k = l>>4], B[k = l>>4][col l&15]
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = As[kk + (lane >> 4)][wm + i * 16 + (lane & 15)];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Bs[kk + (lane >> 4)][wn + j * 16 + (lane & 15)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  // C/D: row (pixel) = 4*(l>>4) + reg, col (channel) = l & 15
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int co = n0 + wn + j * 16 + (lane & 15);
    if (co >= p.Cout) continue;
    const float sc = p.scale ? p.scale[co] : 1.f, sh = p.shift ? p.shift[co] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm + i * 16 + (lane >> 4) * 4 + r;
        if (m >= p.M) continue;
        float v = acc[i][j][r] * sc + sh;
        if (p.relu) v = v > 0.f ? v : 0.f;
        p.out[(long long)m * p.ocs + p.ocoff + co] = v;
      }
  }
}

// The same product for the layers whose input has a multiple of 16 channels (every convolution of Inception-v3 but the first):
// a 16-deep k-step then lies inside ONE filter tap, so the tap and the channel offset are wave-uniform and the gather is two
// 16-byte loads per thread and k-step (4 consecutive input channels of a pixel) instead of eight scalar loads behind an integer
// division.  128 pixels x 64 channels per workgroup (each weight value is used by twice the pixels), register double buffering:
// the loads of k-step i+1 are in flight while step i is multiplied, one barrier per step.  Same k order and the same exact-fp32
// MFMA as the generic kernel above - the results are bitwise the same.  Measured (tools/eval_prof.py, 64 images): see DESIGN.md.
constexpr int CV_M = 128, CV_N = 64, CV_K = 16, CV_LDA = 20, CV_LDB = 80;
__global__ __launch_bounds__(256) void conv2d_nhwc_f32_c16_kernel(const ConvP p) {
  __shared__ __attribute__((aligned(16))) float As[2][CV_M][CV_LDA];       // [pixel][k]: 16-byte rows; (20 m + k) % 64 is conflict-free
  __shared__ __attribute__((aligned(16))) float Bs[2][CV_K][CV_LDB];       // [k][co]: (80 k + co) % 64 is conflict-free
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * CV_M, n0 = blockIdx.y * CV_N;
  // A loader: rows (tid >> 2) and + 64, k quad (tid & 3) * 4;  B loader: k row tid >> 4, channels (tid & 15) * 4
  const int ar = tid >> 2, akq = (tid & 3) * 4;
  long long abase[2];
  int aiy[2], aix[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int m = m0 + ar + 64 * h;
    if (m < p.M) {
      const int ox = m % p.OW, t = m / p.OW, oy = t % p.OH, n = t / p.OH;
      aiy[h] = oy * p.SH - p.PH; aix[h] = ox * p.SW - p.PW;
      abase[h] = (((long long)n * p.H + aiy[h]) * p.W + aix[h]) * p.Cin + akq;
    } else { aiy[h] = -(1 << 28); aix[h] = 0; abase[h] = 0; }
  }
  const int bk = tid >> 4, bco = n0 + (tid & 15) * 4;
  const bool bok = bco < p.Cout;                                           // Cout % 4 == 0 (launcher)
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 32;                    // this wave's 64 x 32 part of the tile
  f32x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk = p.K / CV_K;
  int kh = 0, kw = 0, ci = 0;                                              // tap and channel offset of the k-step being LOADED
  f32x4 ra[2], rb;
  auto load = [&]() {
    const long long toff = ((long long)kh * p.W + kw) * p.Cin + ci;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bool ok = (unsigned)(aiy[h] + kh) < (unsigned)p.H && (unsigned)(aix[h] + kw) < (unsigned)p.W;
      ra[h] = ok ? *reinterpret_cast<const f32x4*>(p.x + abase[h] + toff) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const long long krow = ((long long)(kh * p.KW + kw) * p.Cin + ci + bk);
    rb = bok ? *reinterpret_cast<const f32x4*>(p.w + krow * p.Cout + bco) : f32x4{0.f, 0.f, 0.f, 0.f};
    ci += CV_K;
    if (ci == p.Cin) { ci = 0; if (++kw == p.KW) { kw = 0; ++kh; } }
  };
  auto store = [&](int st) {
#pragma unroll
    for (int h = 0; h < 2; ++h) *reinterpret_cast<f32x4*>(&As[st][ar + 64 * h][akq]) = ra[h];
    *reinterpret_cast<f32x4*>(&Bs[st][bk][(tid & 15) * 4]) = rb;
  };
  load();
  store(0);
  __syncthreads();
  for (int k = 0; k < nk; ++k) {
    const int st = k & 1;
    if (k + 1 < nk) load();                                                // in flight under the products below
#pragma unroll
    for (int kk = 0; kk < CV_K; kk += 4) {
      float a[4], b[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[st][wm + i * 16 + (lane & 15)][kk + (lane >> 4)];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Bs[st][kk + (lane >> 4)][wn + j * 16 + (lane & 15)];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (k + 1 < nk) store(st ^ 1);                                         // nobody reads stage st^1 during step k
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int co = n0 + wn + j * 16 + (lane & 15);
    if (co >= p.Cout) continue;
    const float sc = p.scale ? p.scale[co] : 1.f, sh = p.shift ? p.shift[co] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm + i * 16 + (lane >> 4) * 4 + r;
        if (m >= p.M) continue;
        float v = acc[i][j][r] * sc + sh;
        if (p.relu) v = v > 0.f ? v : 0.f;
        p.out[(long long)m * p.ocs + p.ocoff + co] = v;
      }
  }
}

// mode 0: max over the valid taps; mode 1: sum over the valid taps / (k*k)  (count_include_pad = True)
__global__ void pool2d_nhwc_f32_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int H, int W, int C, int k,
                                       int stride, int pad, int OH, int OW, int mode, int ocs, int ocoff) {
  const long long total = (long long)N * OH * OW * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long long t = i / C;
    const int ox = (int)(t % OW); t /= OW;
    const int oy = (int)(t % OH);
    const long long n = t / OH;
    float v = mode == 0 ? -3.402823466e38f : 0.f;
    for (int dy = 0; dy < k; ++dy)
      for (int dx = 0; dx < k; ++dx) {
        const int iy = oy * stride - pad + dy, ix = ox * stride - pad + dx;
        if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) continue;
        const float u = x[((n * H + iy) * W + ix) * C + c];
        v = mode == 0 ? fmaxf(v, u) : v + u;
      }
    if (mode == 1) v /= (float)(k * k);
    out[((n * OH + oy) * OW + ox) * ocs + ocoff + c] = v;
  }
}

__global__ void global_avgpool_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int HW, int C) {
  const long long total = (long long)N * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long n = i / C;
    float s = 0.f;
    for (int q = 0; q < HW; ++q) s += x[(n * HW + q) * C + c];
    out[i] = s / (float)HW;
  }
}

__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int C, int HW) {
  const long long total = (long long)N * HW * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long t = i / C;
    const long long n = t / HW, q = t % HW;
    out[i] = x[(n * C + c) * HW + q];
  }
}

// column means in fp64: one workgroup per column, fixed summation order
__global__ __launch_bounds__(256) void colmean_f64_kernel(const float* __restrict__ x, double* __restrict__ mean, int N, int D) {
  __shared__ double sm[256];
  const int j = blockIdx.x;
  double s = 0.0;
  for (int n = threadIdx.x; n < N; n += 256) s += (double)x[(long long)n * D + j];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) mean[j] = sm[0] / (double)N;
}
// cov[i][j] = sum_n (x[n][i] - mean[i]) (x[n][j] - mean[j]) / (N - 1): one workgroup per (i, j), fp64, fixed order
__global__ __launch_bounds__(256) void cov_f64_kernel(const float* __restrict__ x, const double* __restrict__ mean,
                                                      double* __restrict__ cov, int N, int D) {
  __shared__ double sm[256];
  const int i = blockIdx.y, j = blockIdx.x;
  if (j < i) return;                                                 // upper triangle; mirrored below
  const double mi = mean[i], mj = mean[j];
  double s = 0.0;
  for (int n = threadIdx.x; n < N; n += 256) s += ((double)x[(long long)n * D + i] - mi) * ((double)x[(long long)n * D + j] - mj);
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double v = sm[0] / (double)(N > 1 ? N - 1 : 1);
    cov[(long long)i * D + j] = v;
    cov[(long long)j * D + i] = v;
  }
}

unsigned grid1d(long long n) { return (unsigned)std::max<long long>(1, std::min<long long>((n + 255) / 256, 16384)); }

}  // namespace

extern "C" int jck_conv2d_nhwc_f32(const float* x, const float* w_kc, const float* scale, const float* shift, float* out, int N,
                                   int H, int W, int Cin, int KH, int KW, int SH, int SW, int PH, int PW, int Cout,
                                   int out_cstride, int out_coff, int relu, void* stream) {
  if (!x || !w_kc || !out || N < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1 || KH < 1 || KW < 1 || SH < 1 || SW < 1 || PH < 0 || PW < 0)
    JCK_FAIL(JCK_E_ARG, "conv2d_nhwc_f32: bad arguments");
  ConvP p = {};
  p.x = x; p.w = w_kc; p.scale = scale; p.shift = shift; p.out = out;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.KH = KH; p.KW = KW; p.SH = SH; p.SW = SW; p.PH = PH; p.PW = PW;
  p.OH = (H + 2 * PH - KH) / SH + 1; p.OW = (W + 2 * PW - KW) / SW + 1;
  if (p.OH < 1 || p.OW < 1) JCK_FAIL(JCK_E_ARG, "conv2d_nhwc_f32: kernel larger than the padded input");
  p.Cout = Cout; p.K = KH * KW * Cin;
  const long long M = (long long)N * p.OH * p.OW;
  if (M >= (1ll << 31) || (long long)N * H * W * Cin >= (1ll << 40)) JCK_FAIL(JCK_E_ARG, "conv2d_nhwc_f32: tensor too large");
  p.M = (int)M;
  if (out_cstride < out_coff + Cout || out_coff < 0) JCK_FAIL(JCK_E_ARG, "conv2d_nhwc_f32: output slice outside the channel stride");
  p.ocs = out_cstride; p.ocoff = out_coff; p.relu = relu;
  static const bool c16 = !(getenv("JCK_INFER_C16") && atoi(getenv("JCK_INFER_C16")) == 0);
  if (c16 && Cin % 16 == 0 && Cout % 4 == 0 && (((uintptr_t)x | (uintptr_t)w_kc) & 15) == 0)
    hipLaunchKernelGGL(conv2d_nhwc_f32_c16_kernel, dim3(cdiv(p.M, CV_M), cdiv(Cout, CV_N)), dim3(256), 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(conv2d_nhwc_f32_kernel, dim3(cdiv(p.M, CB_M), cdiv(Cout, CB_N)), dim3(256), 0, (hipStream_t)stream, p);
  HIPCHK(hipGetLastError());
  return JCK_OK;
}

extern "C" int jck_pool2d_nhwc_f32(const float* x, float* out, int N, int H, int W, int C, int k, int stride, int pad, int mode,
                                   int out_cstride, int out_coff, void* stream) {
  if (!x || !out || N < 1 || C < 1 || k < 1 || stride < 1 || pad < 0 || (mode != 0 && mode != 1)) JCK_FAIL(JCK_E_ARG, "pool2d: bad arguments");
  const int OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
  if (OH < 1 || OW < 1 || out_cstride < out_coff + C) JCK_FAIL(JCK_E_ARG, "pool2d: bad geometry");
  hipLaunchKernelGGL(pool2d_nhwc_f32_kernel, dim3(grid1d((long long)N * OH * OW * C)), dim3(256), 0, (hipStream_t)stream, x, out, N, H,
                     W, C, k, stride, pad, OH, OW, mode, out_cstride, out_coff);
  HIPCHK(hipGetLastError());
  return JCK_OK;
}

extern "C" int jck_global_avgpool_nhwc_f32(const float* x, float* out, int N, int HW, int C, void* stream) {
  if (!x || !out || N < 1 || HW < 1 || C < 1) JCK_FAIL(JCK_E_ARG, "global_avgpool: bad arguments");
  hipLaunchKernelGGL(global_avgpool_kernel, dim3(grid1d((long long)N * C)), dim3(256), 0, (hipStream_t)stream, x, out, N, HW, C);
  HIPCHK(hipGetLastError());
  return JCK_OK;
}

extern "C" int jck_nchw_to_nhwc_f32(const float* x, float* out, int N, int C, int H, int W, void* stream) {
  if (!x || !out || N < 1 || C < 1 || H < 1 || W < 1) JCK_FAIL(JCK_E_ARG, "nchw_to_nhwc: bad arguments");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid1d((long long)N * C * H * W)), dim3(256), 0, (hipStream_t)stream, x, out, N, C, H * W);
  HIPCHK(hipGetLastError());
  return JCK_OK;
}

extern "C" int jck_mean_cov_f64(const float* x, double* mean, double* cov, int N, int D, void* stream) {
  if (!x || !mean || !cov || N < 1 || D < 1) JCK_FAIL(JCK_E_ARG, "mean_cov: bad arguments");
  hipLaunchKernelGGL(colmean_f64_kernel, dim3(D), dim3(256), 0, (hipStream_t)stream, x, mean, N, D);
  HIPCHK(hipGetLastError());
  hipLaunchKernelGGL(cov_f64_kernel, dim3(D, D), dim3(256), 0, (hipStream_t)stream, x, (const double*)mean, cov, N, D);
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
