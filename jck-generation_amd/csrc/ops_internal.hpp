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
                      long long nmask = 0, float keep_p = 0.75f, float* zero = nullptr, long long nzero = 0);
int jck_adam_hp(float* p, const float* g, float* m, float* v, long long n, double beta1, double beta2, double eps,
                float grad_scale, const float* hp, hipStream_t st);
bool jck_prof_is_on();
// The next BatchNorm-backward apply / tanh-backward launch of this thread completes `ev` itself (hipExtLaunchKernel's stopEvent:
// the dispatch packet's own completion signal) - what a hipEventRecord behind it would do with a marker packet of its own,
// which costs the launch stream ~6-7 us of idle time per record on this runtime.  No-op for ev == nullptr.
void jck_arm_stop_event(hipEvent_t ev);
// the armed event if no launch has taken it yet (and disarms): the caller then records it the ordinary way
hipEvent_t jck_take_stop_event();
