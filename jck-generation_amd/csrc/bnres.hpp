// Resident BatchNorm(+activation) backward (round 4): ONE launch per layer instead of reduce + sums + apply, and every tensor
// byte is read ONCE.  aten::native_batch_norm_backward under model/DCGAN.py:30-33,62-65 (train/dcgan_trainer.py:164,175,187).
//
//   g_z = g_a * act'(z),  s1 = sum g_z,  s2 = sum g_z * xhat,  g_y = scale * (g_z - s1/n - xhat * s2/n)
//
// The three-launch form reads (g_a, y) twice - once for the sums, once for the apply - because the sums span the batch.  Here one
// workgroup per CU loads its share of ONE BatchNorm group (= batch) into REGISTERS (bf16 as loaded: at batch 256 the largest
// layer is 33.5 MB x 2 tensors = 256 KB per CU of the 512 KB register file), forms its partial sums, writes them as one row,
// meets the other workgroups at a grid barrier, sums the rows of its channel slice in a fixed order, and applies the result to
// the registers it still holds: passes over HBM 5 -> 3, launches 3 -> 1.  The groups of a batched pass (real | fake | penalty)
// go through the same launch: NG of them are resident together and share ONE barrier when they fit (NG * NCH <= 16 chunks of
// 16 bytes per thread), otherwise one after the other, the loads of the next batch of groups issued between the stores of this one.
//
// Geometry: workgroup = 512 threads = 64 rows x 8 units of 8 channels: a 64-channel SLICE (128 contiguous bytes per row).
// C / 64 slices; the workgroups of a slice share the rows round-robin in blocks of 64.  Only the workgroups of one slice
// exchange sums: nb / nsl rows of 128 floats.  Summation order is fixed by the geometry: two runs give the same bits.
//
// Grid barrier: placement-independent (MI355X guide, "Inter-workgroup communication"): partial rows are stored write-through
// (sc1), every storing wave drains vmcnt, workgroup barrier, one lane arrives on its group's counter (8 groups by
// blockIdx & 7 - the workgroups that share an XCD under round-robin placement; speed only), the last of a group arrives at the
// top counter, the last of those publishes the generation; one lane polls it relaxed with s_sleep, ONE agent-scope acquire,
// workgroup barrier, plain loads.  Counters are reset by their last arriver before the release and the generation is monotonic
// across launches (read once at kernel start, before the first arrival): no memset per launch, graph replay safe.  Every spin is
// bounded (s_memrealtime); a timeout sets the error word and the launch goes on with whatever rows it can read - its results are
// invalid: the engine's Adam launches read the word and leave the weights alone, the host raises at jck_engine_check (step
// scalars, replica guard, checkpoint / evaluation snapshots, end of training, bench.py).
// All workgroups must be co-resident: the launcher sizes the grid to the CU count (one 512-thread workgroup with up to 256
// VGPRs owns a CU), so two such launches must not run at the same time on one device (one stream per engine; set
// JCK_BN_RES=0 when several processes share a GPU).
#pragma once
#include "common.hpp"

#define BNRES_THREADS 512
#define BNRES_ROWS 64                       // rows per workgroup and chunk index
#define BNRES_SYNC_BYTES 2048               // 16 words, each on a 128-byte line of its own
#define BNRES_W_TOP 8
#define BNRES_W_GEN 9
#define BNRES_W_ERR 10
#define BNRES_TIMEOUT_TICKS 30000000ull     // 0.3 s of the 100 MHz s_memrealtime clock

struct BnResParams {
  const bf16_t* ga; const bf16_t* y; bf16_t* gy;
  const float* aux;               // [groups][4C]: scale | shift | mean | invstd
  float* sums; long long sums_stride;   // [groups][stride]: s1 | s2 (2C), then the partial rows (nb * 128 floats)
  float* dgamma; float* dbeta;    // += over groups < grad_groups (may be null)
  unsigned* sync;
  unsigned long long* stamps;     // debug: [nb][8] s_memrealtime stamps of workgroup leaders (null in production)
  long long rows;                 // per group
  int C, groups, grad_groups, nb, nsl;
  float slope, inv_count;
};

__device__ __forceinline__ unsigned bnres_ld(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void bnres_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Every wave has drained its stores (s_waitcnt vmcnt(0)) before the call.  sh_ok: one LDS word.  Returns false on a timeout.
__device__ __forceinline__ bool bnres_grid_sync(unsigned* st, unsigned target_gen, int nb, volatile int* sh_ok) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const int grp = blockIdx.x & 7;
    const int ngrp = nb < 8 ? nb : 8;
    const int gsize = (nb - grp + 7) >> 3;
    const unsigned old = __hip_atomic_fetch_add(st + grp * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (unsigned)(gsize - 1)) {
      bnres_st(st + grp * 32, 0u);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned o2 = __hip_atomic_fetch_add(st + BNRES_W_TOP * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (o2 == (unsigned)(ngrp - 1)) {
        bnres_st(st + BNRES_W_TOP * 32, 0u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bnres_st(st + BNRES_W_GEN * 32, target_gen);
      }
    }
    int ok = 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (bnres_ld(st + BNRES_W_GEN * 32) != target_gen) {
      __builtin_amdgcn_s_sleep(4);
      if (__builtin_amdgcn_s_memrealtime() - t0 > BNRES_TIMEOUT_TICKS) { ok = 0; bnres_st(st + BNRES_W_ERR * 32, 1u); break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    *sh_ok = ok;
  }
  __syncthreads();
  return *sh_ok != 0;
}

__device__ __forceinline__ float bnres_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bnres_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

// byte offset of chunk k; a sum that wraps 32 bits would alias the start of the group: clamp it out of range instead
__device__ __forceinline__ unsigned bnres_off(unsigned voff0, unsigned kstep, int k, unsigned gbytes) {
  const unsigned long long o = (unsigned long long)voff0 + (unsigned long long)kstep * (unsigned)k;
  return o < gbytes ? (unsigned)o : JCK_OOB;
}

#define BNRES_STAMP(i) do { if (p.stamps && threadIdx.x == 0) p.stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)

template <int NCH, int NG>
__global__ __launch_bounds__(BNRES_THREADS) void bn_bwd_res_kernel(const BnResParams p) {
  __shared__ __attribute__((aligned(16))) float sm_wave[NG][8][128];   // per-wave partial rows, later the totals in [j][0]
  __shared__ __attribute__((aligned(16))) float sm_red[16][128];       // row lanes of the cross-workgroup sum
  __shared__ __attribute__((aligned(16))) float sm_cf[NG][5][64];      // scale | shift | mean | s1/n | invstd * s2/n
  __shared__ __attribute__((aligned(16))) float sm_grad[2][64];        // dgamma, dbeta over the gradient groups (slice leader)
  __shared__ int sm_ok;
  const int C = p.C, nsl = p.nsl, nbs = p.nb / nsl;
  const int slice = blockIdx.x % nsl, wslot = blockIdx.x / nsl;
  BNRES_STAMP(0);
  unsigned gen0 = 0;
  if (threadIdx.x == 0) {
    gen0 = bnres_ld(p.sync + BNRES_W_GEN * 32);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (threadIdx.x < 64) { sm_grad[0][threadIdx.x] = 0.f; sm_grad[1][threadIdx.x] = 0.f; }
  // Buffer addressing: a wave-uniform descriptor per group and tensor (base = the group's first byte, size = the group's bytes)
  // and ONE 32-bit byte offset per lane; chunk k adds a uniform step.  Offsets past the group read zeros and drop stores, so
  // ragged tails need no predicate - and no 64-bit address pair per chunk lives in registers beside the tensor.
  const unsigned gbytes = (unsigned)(p.rows * C * 2);
  const long long gelems = p.rows * C;
  const unsigned kstep = (unsigned)((long long)nbs * BNRES_ROWS * C * 2);
  const unsigned wbase = (unsigned)((long long)wslot * BNRES_ROWS * C + slice * 64) * 2u;
  u32x4 vg[NG][NCH], vy[NG][NCH];
  {
    const unsigned voff0 = wbase + (unsigned)((threadIdx.x >> 3) * C + (threadIdx.x & 7) * 8) * 2u;
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const bool on = j < p.groups;
      const auto rg = make_rsrc(p.ga + (on ? j : 0) * gelems, on ? gbytes : 0u);
      const auto ry = make_rsrc(p.y + (on ? j : 0) * gelems, on ? gbytes : 0u);
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        const unsigned off = bnres_off(voff0, kstep, k, gbytes);
        // (g_a, y) are read exactly once - y is dead after this pass, g_a is overwritten by g_y: non-temporal loads (aux 2)
        vg[j][k] = __builtin_amdgcn_raw_buffer_load_b128(rg, (int)off, 0, 2);
        vy[j][k] = __builtin_amdgcn_raw_buffer_load_b128(ry, (int)off, 0, 2);
      }
    }
  }
  unsigned phase = 0;
  for (int g0 = 0; g0 < p.groups; g0 += NG) {
    // the thread index is laundered once per batch of groups: everything derived from it (LDS addresses, shuffle indices,
    // offsets) is recomputed here instead of being hoisted out of the loop into registers that the tensor needs
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    const int u = t & 7, lane = t & 63, wave = t >> 6;
    const int c0 = slice * 64 + u * 8;
    // ---- phase 1: partial sums over the resident chunks (rows past the end and groups past the last hold zeros: g = 0 adds nothing)
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int g = g0 + j < p.groups ? g0 + j : p.groups - 1;
      const float* aux = p.aux + (long long)g * 4 * C;
      float sc[8], sh[8], mu[8];
      {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(aux + c0), a1 = *reinterpret_cast<const f32x4*>(aux + c0 + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(aux + C + c0), b1 = *reinterpret_cast<const f32x4*>(aux + C + c0 + 4);
        const f32x4 m0 = *reinterpret_cast<const f32x4*>(aux + 2 * C + c0), m1 = *reinterpret_cast<const f32x4*>(aux + 2 * C + c0 + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { sc[i] = a0[i]; sc[4 + i] = a1[i]; sh[i] = b0[i]; sh[4 + i] = b1[i]; mu[i] = m0[i]; mu[4 + i] = m1[i]; }
      }
      float s1[8], s2[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
      // One chunk's unpacked values alive at a time - the registers hold the tensor.  The empty asm statements pin that order:
      // instruction selection is free to interleave pure arithmetic of all chunks (it did: 16 chunks unpacked at once, spills),
      // a volatile asm is ordered against the next one, and the chunk's inputs / the running sums pass through them.
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(vg[j][k][i]), "+v"(vy[j][k][i]));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float y0 = bnres_lo(vy[j][k][i]), y1 = bnres_hi(vy[j][k][i]), q0 = bnres_lo(vg[j][k][i]), q1 = bnres_hi(vg[j][k][i]);
          const float z0 = y0 * sc[2 * i] + sh[2 * i], z1 = y1 * sc[2 * i + 1] + sh[2 * i + 1];
          const float gz0 = z0 > 0.f ? q0 : p.slope * q0, gz1 = z1 > 0.f ? q1 : p.slope * q1;
          s1[2 * i] += gz0; s1[2 * i + 1] += gz1;
          s2[2 * i] += gz0 * (y0 - mu[2 * i]); s2[2 * i + 1] += gz1 * (y1 - mu[2 * i + 1]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(s1[i]), "+v"(s2[i]));
        __builtin_amdgcn_sched_barrier(0);
      }
      // the 8 row lanes of a wave that share a unit: lanes l, l^8, l^16, l^32
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) { s1[i] += __shfl_xor(s1[i], o, 64); s2[i] += __shfl_xor(s2[i], o, 64); }
      }
      if (lane < 8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { sm_wave[j][wave][lane * 8 + i] = s1[i]; sm_wave[j][wave][64 + lane * 8 + i] = s2[i]; }
      }
    }
    if (g0 == 0) BNRES_STAMP(1);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      if (g0 + j < p.groups && t < 128) {
        float* prow = p.sums + (long long)(g0 + j) * p.sums_stride + 2 * C + ((long long)slice * nbs + wslot) * 128;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += sm_wave[j][w][t];
        __hip_atomic_store(prow + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // write-through (sc1)
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (g0 == 0) BNRES_STAMP(2);
    // ---- grid barrier (a timeout is recorded in the error word; the launch runs on)
    bnres_grid_sync(p.sync, gen0 + (++phase), p.nb, &sm_ok);
    if (g0 == 0) BNRES_STAMP(3);
    // ---- totals of this slice: nbs rows of 128 floats per group, 16 row lanes x 32 float4 columns, fixed order; the loads of
    // a thread are independent (four rows in flight): a dependent chain of 16 row loads cost 5.8 us at C = 64
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      if (g0 + j >= p.groups) break;
      const int g = g0 + j;
      const float* aux = p.aux + (long long)g * 4 * C;
      {
        const int q = t & 31, rl = t >> 5;
        const float* base = p.sums + (long long)g * p.sums_stride + 2 * C + (long long)slice * nbs * 128 + q * 4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int r = rl; r < nbs; r += 64) {
          f32x4 v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (r + 16 * e < nbs) v[e] = *reinterpret_cast<const f32x4*>(base + (long long)(r + 16 * e) * 128);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) acc += v[e];
        }
        *reinterpret_cast<f32x4*>(&sm_red[rl][q * 4]) = acc;
      }
      __syncthreads();
      if (t < 128) {
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) v += sm_red[r][t];
        sm_wave[j][0][t] = v;                     // s1 total | unnormalised s2 total
      }
      __syncthreads();
      if (t < 64) {
        const int c = slice * 64 + t;
        const float is = aux[3 * C + c];
        const float t1 = sm_wave[j][0][t], t2 = sm_wave[j][0][64 + t] * is;      // s2 = invstd * sum g_z (y - mean)
        sm_cf[j][0][t] = aux[c]; sm_cf[j][1][t] = aux[C + c]; sm_cf[j][2][t] = aux[2 * C + c];
        sm_cf[j][3][t] = t1 * p.inv_count; sm_cf[j][4][t] = is * (t2 * p.inv_count);
        if (wslot == 0) {
          p.sums[(long long)g * p.sums_stride + c] = t1;
          p.sums[(long long)g * p.sums_stride + C + c] = t2;
          if (g < p.grad_groups) { sm_grad[0][t] += t2; sm_grad[1][t] += t1; }
        }
      }
    }
    __syncthreads();
    if (g0 == 0) BNRES_STAMP(4);
    // ---- phase 2: apply to the resident chunks, store, and refill the registers with the next batch of groups
    // (the chunk's registers pass through an asm statement again: otherwise the compiler keeps phase 1's unpacked g_z and
    // y - mean alive across the barrier as fp32 - four times the resident bytes - instead of recomputing them)
    const unsigned voff0 = wbase + (unsigned)((t >> 3) * C + u * 8) * 2u;
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const bool on = g0 + j < p.groups;
      const int g = on ? g0 + j : p.groups - 1;
      float sc[8], sh[8], mu[8], m1[8], k2[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        sc[i] = sm_cf[j][0][u * 8 + i]; sh[i] = sm_cf[j][1][u * 8 + i]; mu[i] = sm_cf[j][2][u * 8 + i];
        m1[i] = sm_cf[j][3][u * 8 + i]; k2[i] = sm_cf[j][4][u * 8 + i];
      }
      const auto ro = make_rsrc(p.gy + (long long)g * gelems, on ? gbytes : 0u);
      // the next batch's group into the registers this one leaves (a zero-sized descriptor past the last group: the loads
      // return zeros and touch no memory)
      const bool more = g0 + NG + j < p.groups;
      const auto rg = make_rsrc(p.ga + (more ? g0 + NG + j : 0) * gelems, more ? gbytes : 0u);
      const auto ry = make_rsrc(p.y + (more ? g0 + NG + j : 0) * gelems, more ? gbytes : 0u);
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        u32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(vg[j][k][i]), "+v"(vy[j][k][i]));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float y0 = bnres_lo(vy[j][k][i]), y1 = bnres_hi(vy[j][k][i]), q0 = bnres_lo(vg[j][k][i]), q1 = bnres_hi(vg[j][k][i]);
          const float z0 = y0 * sc[2 * i] + sh[2 * i], z1 = y1 * sc[2 * i + 1] + sh[2 * i + 1];
          const float gz0 = z0 > 0.f ? q0 : p.slope * q0, gz1 = z1 > 0.f ? q1 : p.slope * q1;
          const float r0 = sc[2 * i] * (gz0 - m1[2 * i] - (y0 - mu[2 * i]) * k2[2 * i]);
          const float r1 = sc[2 * i + 1] * (gz1 - m1[2 * i + 1] - (y1 - mu[2 * i + 1]) * k2[2 * i + 1]);
          o[i] = pack2bf_pk(r0, r1);
        }
        const unsigned off = bnres_off(voff0, kstep, k, gbytes);
        __builtin_amdgcn_raw_buffer_store_b128(o, ro, (int)off, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // (g_a, y) are read exactly once - y is dead after this pass, g_a is overwritten by g_y: non-temporal loads (aux 2)
        vg[j][k] = __builtin_amdgcn_raw_buffer_load_b128(rg, (int)off, 0, 2);
        vy[j][k] = __builtin_amdgcn_raw_buffer_load_b128(ry, (int)off, 0, 2);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (p.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); BNRES_STAMP(5); }
  if (wslot == 0 && threadIdx.x < 64) {
    const int c = slice * 64 + threadIdx.x;
    if (p.dgamma) p.dgamma[c] += sm_grad[0][threadIdx.x];
    if (p.dbeta) p.dbeta[c] += sm_grad[1][threadIdx.x];
  }
}

