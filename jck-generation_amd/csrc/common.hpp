// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the DCGAN/CGAN hot path.
// Wave = 64 lanes; MFMA v_mfma_f32_16x16x32_bf16 (fast) / v_mfma_f32_16x16x4_f32 (parity); padded LDS rows.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;                                     // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;         // one MFMA A/B fragment (4 VGPR)
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(4))) float f32x4;           // one 16x16 accumulator fragment
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define JCK_WAVE 64

// ---- precision tags -----------------------------------------------------------------------------
// Bf16 : activations, gradients and GEMM operands bf16 in HBM/LDS, v_mfma_f32_16x16x32_bf16, fp32 accumulate (fast)
// F32  : everything fp32 in HBM/LDS, v_mfma_f32_16x16x4_f32 = exact fp32 products and accumulation (parity;
//        1/16 of the bf16 MFMA rate)
struct PrecBf16 { typedef bf16_t T; typedef bf16_t W; static constexpr bool IS_F32 = false; };
struct PrecF32  { typedef float  T; typedef float  W; static constexpr bool IS_F32 = true; };

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {                  // RNE, NaN stays NaN (v_cvt_pk_bf16_f32)
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
  return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
}
// the same two roundings as ONE v_cvt_pk_bf16_f32 (the scalar form above compiles to two conversions and an SDWA or)
__device__ __forceinline__ unsigned pack2bf_pk(float lo, float hi) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void stf(T* p, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

// ---- 8-element vector load/store of T as floats ------------------------------------------------------
__device__ __forceinline__ void ld8(const bf16_t* p, float (&v)[8]) {
  u32x4 r = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(r[i] << 16); v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u); }
}
__device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}
__device__ __forceinline__ void st8(bf16_t* p, const float (&v)[8]) {
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = pack2bf(v[2 * i], v[2 * i + 1]);
  *reinterpret_cast<u32x4*>(p) = r;
}
__device__ __forceinline__ void st8(float* p, const float (&v)[8]) {
  f32x4 a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
  *reinterpret_cast<f32x4*>(p) = a;
  *reinterpret_cast<f32x4*>(p + 4) = b;
}
__device__ __forceinline__ void ld4(const bf16_t* p, float (&v)[4]) {
  u32x2 r = *reinterpret_cast<const u32x2*>(p);
  v[0] = __uint_as_float(r[0] << 16); v[1] = __uint_as_float(r[0] & 0xffff0000u);
  v[2] = __uint_as_float(r[1] << 16); v[3] = __uint_as_float(r[1] & 0xffff0000u);
}
__device__ __forceinline__ void ld4(const float* p, float (&v)[4]) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p);
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
}
__device__ __forceinline__ void st4(bf16_t* p, const float (&v)[4]) {
  u32x2 r; r[0] = pack2bf(v[0], v[1]); r[1] = pack2bf(v[2], v[3]);
  *reinterpret_cast<u32x2*>(p) = r;
}
__device__ __forceinline__ void st4(float* p, const float (&v)[4]) {
  f32x4 a; a[0] = v[0]; a[1] = v[1]; a[2] = v[2]; a[3] = v[3];
  *reinterpret_cast<f32x4*>(p) = a;
}

// ---- wave reductions ---------------------------------------------------------------------------------
// (DPP row rotations + four v_readlane instead of __shfl_xor = ds_bpermute_b32: an LDS-crossbar round trip per step and value -
// the small finalize / sums / head kernels were mostly waiting on those)
template <int CTRL> __device__ __forceinline__ float dpp_row(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// sum over the 64 lanes, valid in every lane: row sums by rotation, then the four rows' totals through SGPRs in a fixed order
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_row<0x128>(v);
  v += dpp_row<0x124>(v);
  v += dpp_row<0x122>(v);
  v += dpp_row<0x121>(v);
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int ctrl = 0; ctrl < 4; ++ctrl) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    int lo = (int)(unsigned)u, hi = (int)(unsigned)(u >> 32);
    if (ctrl == 0) { lo = __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xf, 0xf, false); }
    if (ctrl == 1) { lo = __builtin_amdgcn_update_dpp(0, lo, 0x124, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x124, 0xf, 0xf, false); }
    if (ctrl == 2) { lo = __builtin_amdgcn_update_dpp(0, lo, 0x122, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x122, 0xf, 0xf, false); }
    if (ctrl == 3) { lo = __builtin_amdgcn_update_dpp(0, lo, 0x121, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x121, 0xf, 0xf, false); }
    v += __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
  }
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const int lo = (int)(unsigned)u, hi = (int)(unsigned)(u >> 32);
  double r[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned l = (unsigned)__builtin_amdgcn_readlane(lo, 16 * k), h = (unsigned)__builtin_amdgcn_readlane(hi, 16 * k);
    r[k] = __builtin_bit_cast(double, ((unsigned long long)h << 32) | l);
  }
  return (r[0] + r[1]) + (r[2] + r[3]);
}
// sum over the 16 lanes that share (lane>>4), valid in every lane.  DPP row rotations (v_add_f32 ... row_ror:n) instead of
// __shfl_xor, which hipcc lowers to ds_bpermute_b32 - an LDS-crossbar round trip of ~100 cycles per step, 4 dependent steps per
// value, hundreds of them in a GEMM tile epilogue (round 5: the per-tile statistics cost 8-23 us of a 65-100 us launch).  Lane i
// adds, step by step, partial sums over the same lane sets as the xor butterfly does, and float addition is commutative: the
// same bits.
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_row<0x128>(v);      // row_ror:8
  v += dpp_row<0x124>(v);      // row_ror:4
  v += dpp_row<0x122>(v);      // row_ror:2
  v += dpp_row<0x121>(v);      // row_ror:1
  return v;
}
// block sum for blockDim.x == 256 (4 waves); result valid in every thread
__device__ __forceinline__ float block_sum256(float v, float* sm /* >= 4 floats */) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];
}

// Transposing row reduction: lane r = lane & 15 returns the sum, over the 16 lanes of its row, of v[r].  A halving butterfly (xor
// 8, 4, 2, 1; the half a lane keeps is chosen by that bit of r): 15 DPP adds instead of the 64 that sixteen row16_sum calls take,
// and ONE live result per lane instead of sixteen - which is what lets the persistent gather-GEMM carry its BatchNorm sums across
// tiles in 2 registers instead of 32.  Fixed order: deterministic.
__device__ __forceinline__ float row16_transpose_sum(const float (&v)[16], int r) {
  float w8[8], w4[4], w2[2];
  const bool b3 = r & 8, b2 = r & 4, b1 = r & 2, b0 = r & 1;
#pragma unroll
  for (int i = 0; i < 8; ++i) w8[i] = (b3 ? v[i + 8] : v[i]) + dpp_row<0x128>(b3 ? v[i] : v[i + 8]);            // xor 8 = row_ror:8
#pragma unroll
  for (int i = 0; i < 4; ++i)                                                                                     // xor 4 = quad reverse, then half mirror
    w4[i] = (b2 ? w8[i + 4] : w8[i]) + dpp_row<0x141>(dpp_row<0x1B>(b2 ? w8[i] : w8[i + 4]));
#pragma unroll
  for (int i = 0; i < 2; ++i) w2[i] = (b1 ? w4[i + 2] : w4[i]) + dpp_row<0x4E>(b1 ? w4[i] : w4[i + 2]);          // xor 2 = quad_perm [2,3,0,1]
  return (b0 ? w2[1] : w2[0]) + dpp_row<0xB1>(b0 ? w2[0] : w2[1]);                                                // xor 1 = quad_perm [1,0,3,2]
}

// ---- MFMA --------------------------------------------------------------------------------------------
// v_mfma_f32_16x16x32_bf16: lane l holds A[row l&15][k = 8*(l>>4) + j], B[k = 8*(l>>4)+j][col l&15],
// C/D[row = 4*(l>>4) + reg][col = l&15].
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 lds_frag(const bf16_t* p) {      // 16-byte aligned LDS read
  return *reinterpret_cast<const bf16x8*>(p);
}
// ds_read_b64_tr_b16: per 16-lane group reads a 4(row) x 16(col) block of 16-bit elements and hands
// lane i of the group column i (rows 0..3).  Lane 4q+p of the group supplies &block[row q][col 4p].
__device__ __forceinline__ short4v lds_tr4(const bf16_t* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (short4v __attribute__((address_space(3)))*)(const_cast<bf16_t*>(p)));
}
__device__ __forceinline__ bf16x8 join_tr(short4v a, short4v b) {
  typedef __attribute__((ext_vector_type(8))) short short8v;
  short8v r = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, r);
}

// 8 consecutive elements of T held in registers between the global load and the LDS store
template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> { u32x4 v; };
template <> struct Raw8<float> { f32x4 a, b; };
__device__ __forceinline__ void zero_raw(Raw8<bf16_t>& r) { r.v = u32x4{0u, 0u, 0u, 0u}; }
__device__ __forceinline__ void zero_raw(Raw8<float>& r) { r.a = f32x4{0.f, 0.f, 0.f, 0.f}; r.b = r.a; }
__device__ __forceinline__ void ldraw(const bf16_t* p, Raw8<bf16_t>& r) { r.v = *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void ldraw(const float* p, Raw8<float>& r) {
  r.a = *reinterpret_cast<const f32x4*>(p);
  r.b = *reinterpret_cast<const f32x4*>(p + 4);
}
// half = 4 elements (one 4-channel pixel) into slot h of the 8-element unit
__device__ __forceinline__ void ldraw_half(const bf16_t* p, Raw8<bf16_t>& r, int h) {
  const u32x2 t = *reinterpret_cast<const u32x2*>(p);
  if (h == 0) { r.v[0] = t[0]; r.v[1] = t[1]; } else { r.v[2] = t[0]; r.v[3] = t[1]; }
}
__device__ __forceinline__ void ldraw_half(const float* p, Raw8<float>& r, int h) {
  const f32x4 t = *reinterpret_cast<const f32x4*>(p);
  if (h == 0) r.a = t; else r.b = t;
}
__device__ __forceinline__ void unraw(const Raw8<bf16_t>& r, float (&v)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(r.v[i] << 16); v[2 * i + 1] = __uint_as_float(r.v[i] & 0xffff0000u); }
}
__device__ __forceinline__ void unraw(const Raw8<float>& r, float (&v)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[i] = r.a[i]; v[4 + i] = r.b[i]; }
}
__device__ __forceinline__ void straw(bf16_t* p, const Raw8<bf16_t>& r) { *reinterpret_cast<u32x4*>(p) = r.v; }
__device__ __forceinline__ void straw(float* p, const Raw8<float>& r) {
  *reinterpret_cast<f32x4*>(p) = r.a;
  *reinterpret_cast<f32x4*>(p + 4) = r.b;
}

// ---- buffer loads: 32-bit byte offsets against a wave-uniform descriptor; an offset past num_records returns 0,
// which is how out-of-image taps and rows past M become zero operands without a branch or a zero-fill
#define JCK_OOB 0x80000000u
__device__ __forceinline__ auto make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
template <class R> __device__ __forceinline__ void buf_ld8(R rs, unsigned off, Raw8<bf16_t>& r) {
  r.v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
}
template <class R> __device__ __forceinline__ void buf_ld8(R rs, unsigned off, Raw8<float>& r) {
  const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
  const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off + 16), 0, 0);
  r.a = __builtin_bit_cast(f32x4, a);
  r.b = __builtin_bit_cast(f32x4, b);
}
// 4 elements (one 4-channel pixel) into half h of the unit
template <class R> __device__ __forceinline__ void buf_ld4(R rs, unsigned off, Raw8<bf16_t>& r, int h) {
  const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0);
  if (h == 0) { r.v[0] = t[0]; r.v[1] = t[1]; } else { r.v[2] = t[0]; r.v[3] = t[1]; }
}
template <class R> __device__ __forceinline__ void buf_ld4(R rs, unsigned off, Raw8<float>& r, int h) {
  const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
  if (h == 0) r.a = __builtin_bit_cast(f32x4, t); else r.b = __builtin_bit_cast(f32x4, t);
}

static inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
