// BatchNorm statistics, second form (round 3): accumulated by the kernel that PRODUCES the tensor and finalised by the last
// workgroup of that launch - no statistics-finalising launch (bn_finalize / bn_bwd_sums) and no partial-row buffers.
//
//   * Per-channel sums are kept EXACTLY: every fp32 partial (a workgroup's or a wave's sum over its pixels) is split into three
//     integer limbs - integer part, fraction bits 2^-1..2^-32, fraction bits 2^-33..2^-64 - and each limb is added to its own
//     64-bit word with an integer atomic.  No carry moves between the words while the launch runs (a limb is < 2^32 per add, the
//     word holds 2^32 of them), so the three atomics of an add are independent, need no return value and stay in flight
//     together.  Integer addition is associative: the total does not depend on the order in which workgroups arrive and two
//     runs give the same bits, which float atomics (removed in round 2) could not promise.  What reaches memory is the exact sum
//     of the fp32 partials; the partials themselves are summed in a fixed order inside their wave.
//   * `reps` replicas of every accumulator spread the adders of one address (a replica per blockIdx.x % reps); the finaliser
//     adds the replicas (exact again).
//   * The CONSUMER kernel turns the sums into coefficients in its prologue (every workgroup for itself, into LDS: a few KB of
//     loads behind the kernel boundary), and its first workgroup leaves what later launches read:
//       forward  (bn_act_fwd_x_kernel): sum y, sum y^2 -> scale | shift | mean | invstd per group (aux), the deferred (mean,
//                unbiased var) record or the running statistics - the arithmetic of bn_finalize_kernel (ew.hpp);
//       backward (bn_bwd_apply_x_kernel): sum g_z, sum g_z (y - mean) -> s1 | s2 = sum g_z xhat per group (the table the CGAN
//                double backward reads), dgamma += sum_g s2, dbeta += sum_g s1 over the gradient groups.
//     (A first form finalised in the producer's last workgroup - ticket, loads, stores: four dependent memory round trips,
//     10-14 us at the end of EVERY producer, more than the launches it replaced.  Measured and dropped.)
//   * The caller zeroes xs and flags before the producer launch (one memset per pass for every layer's accumulators).
// Reference semantics: aten::native_batch_norm / native_batch_norm_backward in training mode as reached from
// model/DCGAN.py:30-33,62-65 and train/dcgan_trainer.py:164,175,187.
#pragma once
#include "common.hpp"

struct BnStatJob {
  unsigned long long* xs;      // [reps][groups][2][3 limbs][C] u64 accumulators; nullptr = statistics off
  unsigned* flags;             // [1] poison (a non-finite or absurd partial was seen: everything reads NaN)
  int reps, groups, C, mode;   // mode 1 forward, 2 backward
  float count;                 // rows (pixels) per group
  const float* gamma;          // forward
  const float* beta;
  float eps, momentum;
  float* aux;                  // forward: written, [groups][4C]; backward: read (mean, invstd)
  float* rec;                  // forward: [groups][2C] deferred running-stat records (mean | unbiased var) or nullptr
  float* running_mean;         // forward, groups == 1: updated in place (G's BatchNorm) or nullptr
  float* running_var;
  long long* nbt;
  float* sums;                 // backward: [groups][sums_stride], the first 2C floats of a group = s1 | s2
  long long sums_stride;
  float* dgamma;               // backward: += over groups < grad_groups (or nullptr)
  float* dbeta;
  int grad_groups;
};

#define XS_LIMBS 3             /* u64 planes per (replica, group, statistic): integer part, fraction bits -1..-32, -33..-64 */
#define XS_LIMIT 4.5e15f       /* 2^52: 2048 adds of this size stay inside the 63-bit integer word */

// plane of C consecutive u64 words: the lanes of a wave add CONSECUTIVE channels, so one atomic instruction is 512 contiguous
// bytes (8 memory-side requests) - a wave whose lanes hit 64 separate 32-byte records measured 20x slower (the memory side
// processes one 64-byte request per atomic segment)
__device__ __forceinline__ unsigned long long* xs_plane(const BnStatJob& j, int rep, int group, int stat, int limb) {
  return j.xs + ((((long long)rep * j.groups + group) * 2 + stat) * XS_LIMBS + limb) * j.C;
}

// one exact add of v to channel c: up to three independent no-return atomics (zero limbs are skipped).  "Performed" = the
// issuing wave's s_waitcnt vmcnt(0) (bn_stat_arrive), which on gfx9 covers stores and atomics without return.
__device__ __forceinline__ void xsum_add(const BnStatJob& j, int rep, int group, int stat, int c, float v) {
  if (v == 0.f) return;
  if (!(fabsf(v) < XS_LIMIT)) { atomicOr(j.flags + 1, 1u); return; }
  const double d = (double)v, fl = floor(d);
  const double t = (d - fl) * 4294967296.0, th = floor(t);            // fraction * 2^32: exact power-of-two scaling
  const unsigned long long ip = (unsigned long long)(long long)fl;
  const unsigned long long fh = (unsigned long long)th, lo = (unsigned long long)((t - th) * 4294967296.0);
  if (ip) __hip_atomic_fetch_add(xs_plane(j, rep, group, stat, 0) + c, ip, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (fh) __hip_atomic_fetch_add(xs_plane(j, rep, group, stat, 1) + c, fh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (lo) __hip_atomic_fetch_add(xs_plane(j, rep, group, stat, 2) + c, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// sum of the replicas of one accumulator as a double.  Read by the CONSUMER launch: the kernel boundary behind the producer
// makes plain (L2-cached) loads correct, and they must be plain - every workgroup of the consumer reads the same few KB, and
// device-scope (sc1) loads of one hot region from 3 000 workgroups serialised at the memory side (+12..23 us per launch).
__device__ __forceinline__ double xsum_read(const BnStatJob& j, int group, int stat, int c) {
  long long ip = 0;
  unsigned long long fh = 0, lo = 0;
  unsigned long long w[4][3];
#pragma unroll
  for (int r = 0; r < 4; ++r)                    // every load issued before the first use (reps <= 4)
#pragma unroll
    for (int l = 0; l < 3; ++l)
      w[r][l] = r < j.reps ? xs_plane(j, r, group, stat, l)[c] : 0ull;
#pragma unroll
  for (int r = 0; r < 4; ++r) { ip += (long long)w[r][0]; fh += w[r][1]; lo += w[r][2]; }
  return (double)ip + ((double)fh + (double)lo * 2.3283064365386962890625e-10) * 2.3283064365386962890625e-10;     // 2^-32
}

// A workgroup's statistics leave it as ONE partial per (channel, statistic): the waves put their sums into LDS
// (scr[wave][2][CHW], CHW = channels per wave; the waves with equal wave / WPIX hold the same channels), and the first 2 * BCH
// threads add the WPIX partials of a channel in wave order and issue the atomics, 64 consecutive channels per instruction.
// tid / nthr: the calling threads (all of them have passed a barrier behind the LDS writes).
template <int BCH, int CHW, int WPIXN>
__device__ __forceinline__ void bn_wg_partials_add(const BnStatJob& j, const float* scr, int tid, int nthr, int rep, int grp, int chb,
                                                   int nch_store, int cstat) {
  for (int t = tid; t < 2 * BCH; t += nthr) {
    const int stat = t / BCH, cl = t - stat * BCH, wc = cl / CHW, local = cl - wc * CHW;
    float sum = 0.f;
#pragma unroll
    for (int wp = 0; wp < WPIXN; ++wp) sum += scr[((wc * WPIXN + wp) * 2 + stat) * CHW + local];
    const int ch = chb + cl;
    if (ch < nch_store) xsum_add(j, rep, grp, stat, ch & (cstat - 1), sum);
  }
}

// ---- what the consumers make of the sums (in their prologue; the kernel boundary behind the producer makes plain loads safe,
// the device-scope loads of xsum_read cost nothing extra) -------------------------------------------------------------------
struct BnFwdCoef { float scale, shift, mean, invstd, unbiased; };
// forward: the arithmetic of bn_finalize_kernel (ew.hpp) / aten::native_batch_norm's statistics
__device__ __forceinline__ BnFwdCoef bn_fwd_coef(const BnStatJob& j, int g, int c, bool poison) {
  double sd = xsum_read(j, g, 0, c), qd = xsum_read(j, g, 1, c);
  if (poison) { sd = __longlong_as_double(0x7ff8000000000000ll); qd = sd; }
  const double ic = 1.0 / (double)j.count;      // one division (wave-uniform), then multiplies
  const double meand = sd * ic;
  double vard = qd * ic - meand * meand;
  if (vard < 0.0) vard = 0.0;
  BnFwdCoef r;
  r.mean = (float)meand;
  const float var = (float)vard;
  r.invstd = 1.0f / sqrtf(var + j.eps);
  r.scale = j.gamma[c] * r.invstd;
  r.shift = j.beta[c] - r.mean * r.scale;
  r.unbiased = var * (j.count / fmaxf(j.count - 1.f, 1.f));
  return r;
}
// backward: s1 = sum g_z, s2 = sum g_z xhat = invstd * sum g_z (y - mean)
__device__ __forceinline__ void bn_bwd_sums(const BnStatJob& j, int g, int c, bool poison, float invstd, float& s1, float& s2) {
  double s1d = xsum_read(j, g, 0, c), q2d = xsum_read(j, g, 1, c);
  if (poison) { s1d = __longlong_as_double(0x7ff8000000000000ll); q2d = s1d; }
  s1 = (float)s1d;
  s2 = (float)(q2d * (double)invstd);
}
__device__ __forceinline__ bool bn_poisoned(const BnStatJob& j) {
  return j.flags[1] != 0;
}
