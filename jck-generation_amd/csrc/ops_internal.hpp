// Internal glue shared by ops.hip and engine.hip.
#pragma once
#include <algorithm>
#include <cmath>
#include <string>

#include "../../include/jckgan.h"
#include "common.hpp"
#include "ew.hpp"
#include "igemm.hpp"
#include "wgrad.hpp"

void jck_set_error(const std::string& s);

#define JCK_FAIL(code, msg)                                   \
  do {                                                        \
    jck_set_error(std::string(__func__) + ": " + (msg));     \
    return (code);                                            \
  } while (0)

#define HIPCHK(expr)                                                                                     \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) {                                                                              \
      jck_set_error(std::string(__func__) + ": HIP error " + hipGetErrorString(e_) + " at " #expr);     \
      return JCK_E_HIP;                                                                                  \
    }                                                                                                    \
  } while (0)

#define JCK_TRY(expr)           \
  do {                          \
    int rc_ = (expr);           \
    if (rc_ != JCK_OK) return rc_; \
  } while (0)

int launch_igemm(int prec, const IgemmParams& p, int nch_pad, int phases, int nsub, hipStream_t st, int* slots);
// Adam with {step_size, bc2_sqrt} in device memory (ops.hip): the engine's step has no per-step kernel argument
int jck_adam_set_step(float* hp, double lr, double beta1, double beta2, int step, unsigned long long seed, hipStream_t st,
                      float* rz = nullptr, long long nz = 0, float* ralpha = nullptr, long long nalpha = 0, float* rmasks = nullptr,
                      long long nmask = 0, float keep_p = 0.75f, float* zero = nullptr, long long nzero = 0, float* zbig0 = nullptr,
                      long long nzbig0 = 0, float* zbig1 = nullptr, long long nzbig1 = 0, void* zpad = nullptr, int zd = 0, int zp = 0,
                      int zpad_f32 = 0);      // zpad: the drawn z also as rows [nz / zd][zp] of type T (G.conv1's operand; padding columns untouched)
int jck_adam_hp(float* p, const float* g, float* m, float* v, long long n, double beta1, double beta2, double eps,
                float grad_scale, const float* hp, hipStream_t st, float* zero = nullptr, long long nzero = 0,
                const unsigned* skip_if = nullptr);      // skip_if: device word; non-zero = leave p, m, v untouched (a grid barrier of the step timed out)
const unsigned* jck_grid_sync_error_word(const void* sync_ws);
bool jck_prof_is_on();
// BatchNorm finalize + apply as one launch where the statistics rows are few (ops.hip); *fused = false: nothing was launched
int bn_fwd_fused(int prec, const void* y, const float* stats, int slots_per_group, float count, const float* gamma, const float* beta,
                 float eps, float slope, void* a, float* aux, float* stat_out, float* running_mean, float* running_var, int64_t* nbt,
                 float momentum, long long rows_per_group, int C, int groups, long long out_row, long long out_pitch, hipStream_t stream,
                 bool* fused);
// Internal forms of the two launches whose result another stream waits for: `done` (may be null) is completed by the launch
// that writes the result - the dispatch packet's own completion signal (hipExtLaunchKernel's stop event) instead of a
// hipEventRecord behind it, whose marker packet costs the launch stream ~6-7 us of idle time on this runtime.  The event is an
// explicit argument (round 3 handed it over through a thread-local "armed" slot that the next armable launch consumed).
int bn_act_bwd_res_ev(int prec, const void* g_a, const void* y, const float* aux, float slope, float* sums, void* g_y, float* dgamma,
                      float* dbeta, long long rows_per_group, int C, int groups, int grad_groups, void* sync_ws, hipStream_t stream,
                      hipEvent_t done);
int conv_up_tanh_bwd_ev(int prec, const void* small_in, const void* w, const void* tanh_y, float scale, void* out, int N, int Hs, int Ws,
                        int Cs, int Cb, hipStream_t stream, hipEvent_t done, bool* fused);
int head_bwd_conv2_ev(int prec, const float* ds, const float* wp, const void* a4, int B, int B_more, int C, void* g_a4, float* grad, float* ws,
                      hipStream_t stream, hipStream_t side, hipEvent_t handover);
int gp_head2_ev(int prec, const void* ughd, const float* w2, const float* prob, int B, int K, float* rs, float* dw2, float* ws,
                hipStream_t stream, hipStream_t side, hipEvent_t handover);
int pack_linear_pair(int prec, const float* w, int N, int K, int rows0, int cols0, void* wp0, int rows1, int cols1, void* wp1, int permC,
                     int permHW, hipStream_t stream);
int bn_act_fwd_pitched(int prec, const void* y, const float* aux, float slope, void* a, long long rows_per_group, int C, int groups,
                       long long out_row, long long out_pitch, hipStream_t stream);
int cg_head_mid(int prec, const float* slab, int ksplit, const float* bias1, const float* mask, float scale, void* h, void* hd, const float* w2,
                const float* bias2, int B, int G, const float* targets, const int* modes, float* prob, float* ds, float* scal,
                const int* slot_loss, const int* slot_p, int scal_ld, void* g_hd, void* g_h, hipStream_t stream);
int gp_head_mid_ev(int prec, const float* slab, int ksplit, const float* mask, float scale, void* ughd, const float* w2, const float* prob, int B,
                   float* rs, float* dw2, float* ws, void* g_hd, void* g_h, hipStream_t stream, hipStream_t side, hipEvent_t handover);
int head_fwd_grouped_ev(int prec, const void* a4, const float* wp, const float* bias, int B, int K, int G, const float* targets,
                        const int* modes, float* prob, float* ds, float* scal, const int* slot_loss, const int* slot_p, int scal_ld,
                        void* g_out, hipStream_t stream, hipEvent_t done);
int tanh_bwd_ev(int prec, const void* g, const void* y, float scale, void* out, long long numel, hipStream_t stream, hipEvent_t done);
