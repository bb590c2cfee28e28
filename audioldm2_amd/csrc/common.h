// Shared helpers for the gfx950 kernel library (libaldm_hip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/aldm_hip.h"

namespace aldm {

void set_error(const char* fmt, ...);

#define ALDM_CHECK(cond, ...)                  \
    do {                                       \
        if (!(cond)) {                         \
            aldm::set_error(__VA_ARGS__);      \
            return -1;                         \
        }                                      \
    } while (0)

#define ALDM_LAUNCH_CHECK(name)                                                         \
    do {                                                                                \
        hipError_t e__ = hipGetLastError();                                             \
        if (e__ != hipSuccess) {                                                        \
            aldm::set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
            return -2;                                                                  \
        }                                                                               \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float act_apply(float v, int act, float slope) {
    switch (act) {
        case ALDM_ACT_SILU: return v / (1.0f + expf(-v));
        case ALDM_ACT_LRELU: return v > 0.0f ? v : v * slope;
        case ALDM_ACT_TANH: return tanhf(v);
        case ALDM_ACT_LOGCLAMP: return logf(fmaxf(v, slope));
        case ALDM_ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        case ALDM_ACT_GELU_TANH:  // transformers NewGELUActivation (GPT-2 "gelu_new")
            return 0.5f * v * (1.0f + tanhf(0.79788456080286535588f * (v + 0.044715f * v * v * v)));
        default: return v;
    }
}

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace aldm
