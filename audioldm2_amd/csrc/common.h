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

// erf GELU 0.5 v (1 + erf(v / sqrt 2)) without libm: Abramowitz-Stegun 7.1.26 for erfc(|z|) = poly(t) exp(-z^2), t = 1/(1 + p|z|)
// (|error| <= 1.5e-7), evaluated as 1 + erf(z) = erfc(-z) = { poly e : z < 0 ; 2 - poly e : z >= 0 } so the negative tail has
// no cancellation.  Max |error| vs the fp64 GELU 3.3e-7 over [-12, 12] (ATen's fp32 F.gelu: 1.2e-6), 1.6e-7 relative to
// max(|v|, 1); ocml erff costs ~50 VALU (both of its branches execute in a divergent wave) — the GEGLU epilogue is VALU bound on
// this (tools/dma_ablate_shapes.py).
__device__ __forceinline__ float gelu_erf_fast(float v) {
    // 16 VALU instructions (round 4: ~23 — the library is built with -ffp-contract=off, so the Horner steps are explicit FMAs here,
    // and the sign select is folded away: 0.5 v (1 + erf z) = max(v, 0) - 0.5 |v| poly(t) exp(-z^2) for either sign of v).  In the
    // GEGLU epilogues every one of these instructions is issue time the matrix pipe does not overlap (docs/experiments_r1-r6.md §3.1c).
    const float a = fabsf(v) * 0.70710678118654752440f;                       // |z|
    const float t = __frcp_rn(__builtin_fmaf(0.3275911f, a, 1.0f));
    float q = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
    q = __builtin_fmaf(t, q, 1.421413741f);
    q = __builtin_fmaf(t, q, -0.284496736f);
    q = __builtin_fmaf(t, q, 0.254829592f);
    const float pe = (t * q) * __builtin_amdgcn_exp2f((a * a) * -1.44269504088896340736f);   // erfc(|z|)
    return __builtin_fmaf(fabsf(v) * -0.5f, pe, fmaxf(v, 0.0f));
}

__device__ __forceinline__ float act_apply(float v, int act, float slope) {
    switch (act) {
        case ALDM_ACT_SILU: return v / (1.0f + expf(-v));
        case ALDM_ACT_LRELU: return v > 0.0f ? v : v * slope;
        case ALDM_ACT_TANH: return tanhf(v);
        case ALDM_ACT_LOGCLAMP: return logf(fmaxf(v, slope));
        case ALDM_ACT_GELU: return gelu_erf_fast(v);
        case ALDM_ACT_GELU_TANH:  // transformers NewGELUActivation (GPT-2 "gelu_new")
            return 0.5f * v * (1.0f + tanhf(0.79788456080286535588f * (v + 0.044715f * v * v * v)));
        default: return v;
    }
}

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace aldm
