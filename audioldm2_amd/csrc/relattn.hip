// relattn.hip — windowed relative-position self-attention of the VITS phoneme encoder:
// `MultiHeadAttention.attention` with window_size = 4 and shared heads
// (audioldm2/latent_diffusion/modules/phoneme_encoder/attentions.py:239-289), plus the per-row scale glue of its
// masked conv FFN.  A once-per-prompt conditioner (310 tokens x 192 channels, 2 heads x 96): 2.4 GFLOP per 32 prompts and
// layer — latency, not throughput, so plain fp32 FMA over LDS tiles (exact fp32 products, one block per 32 queries):
//   s[i, j] = (q_i / sqrt(d)) . k_j + [|j - i| <= W] (q_i / sqrt(d)) . Ek[j - i + W];   s = -1e4 where mask_i * mask_j == 0
//   p = softmax_j(s);   o_i = sum_j p[i, j] v_j + sum_{|r| <= W} p[i, i + r] Ev[r + W]
// (the reference's pad / reshape skews, attentions.py:317-361, are exactly these index shifts).
#include "common.h"
#include <float.h>

namespace aldm {

constexpr int RA_QB = 32;    // queries per block
constexpr int RA_KB = 32;    // keys per LDS tile
constexpr int RA_MAXD = 128;
constexpr int RA_MAXW = 8;

__global__ __launch_bounds__(256) void rel_attention_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, float* __restrict__ out, int T,
                                                            int d, int ldq, int ldk, int ldv, int ldo,
                                                            const float* __restrict__ emb_k,
                                                            const float* __restrict__ emb_v, int W,
                                                            const float* __restrict__ mask, int Tpad) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int dp = d + 1;                       // padded row pitch of the [*, d] tiles
    float* qs = sm;                             // [QB][dp]   scaled queries
    float* kv = qs + RA_QB * dp;                // [KB][dp]   current K or V tile
    float* rl = kv + RA_KB * dp;                // [QB][2W+1] relative logits, later relative weights
    float* S = rl + RA_QB * (2 * RA_MAXW + 1);  // [QB][Tpad] scores -> probabilities
    const int tid = threadIdx.x;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * RA_QB;
    const int R = 2 * W + 1;
    const float inv = sqrtf((float)d);
    const float* qb = q + (int64_t)b * T * ldq + h * d;
    const float* kb = k + (int64_t)b * T * ldk + h * d;
    const float* vb = v + (int64_t)b * T * ldv + h * d;
    const float* mb = mask + (int64_t)b * T;
    for (int i = tid; i < RA_QB * d; i += 256) {
        const int qi = i / d, c = i - qi * d;
        const int t = min(q0 + qi, T - 1);
        qs[qi * dp + c] = qb[(int64_t)t * ldq + c] / inv;                     // attentions.py:247 (query / sqrt(k_channels))
    }
    __syncthreads();
    for (int i = tid; i < RA_QB * R; i += 256) {                              // :252-255
        const int qi = i / R, r = i - qi * R;
        float a = 0.f;
        for (int c = 0; c < d; ++c) a = fmaf(qs[qi * dp + c], emb_k[r * d + c], a);
        rl[qi * (2 * RA_MAXW + 1) + r] = a;
    }
    // ---- scores ----
    const int qi = tid >> 3, kj0 = (tid & 7) * 4;
    const int tq = q0 + qi;
    const float mq = tq < T ? mb[tq] : 0.f;
    for (int j0 = 0; j0 < T; j0 += RA_KB) {
        __syncthreads();
        for (int i = tid; i < RA_KB * d; i += 256) {
            const int kj = i / d, c = i - kj * d;
            const int t = min(j0 + kj, T - 1);
            kv[kj * dp + c] = kb[(int64_t)t * ldk + c];
        }
        __syncthreads();
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < d; ++c) {
            const float x = qs[qi * dp + c];
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = fmaf(x, kv[(kj0 + e) * dp + c], a[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int tk = j0 + kj0 + e;
            if (tk < T) {
                const int off = tk - tq;
                float s = a[e];
                if (off >= -W && off <= W) s += rl[qi * (2 * RA_MAXW + 1) + off + W];   // :256-257
                if (mq * mb[tk] == 0.f) s = -1e4f;                                       // :263 (mask = mask_i * mask_j, :76)
                S[qi * Tpad + tk] = s;
            }
        }
    }
    __syncthreads();
    // ---- softmax over the T keys of each query: 8 threads per row ----
    {
        const int l8 = tid & 7;
        float mx = -FLT_MAX;
        for (int j = l8; j < T; j += 8) mx = fmaxf(mx, S[qi * Tpad + j]);
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float sum = 0.f;
        for (int j = l8; j < T; j += 8) {
            const float e = expf(S[qi * Tpad + j] - mx);
            S[qi * Tpad + j] = e;
            sum += e;
        }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float rs = 1.0f / sum;
        for (int j = l8; j < T; j += 8) S[qi * Tpad + j] *= rs;
    }
    __syncthreads();
    // relative weights: rw[i, r] = p[i, i + r - W] (0 outside the sequence)  (:277-279)
    for (int i = tid; i < RA_QB * R; i += 256) {
        const int qq = i / R, r = i - qq * R;
        const int tk = q0 + qq + r - W;
        rl[qq * (2 * RA_MAXW + 1) + r] = (tk >= 0 && tk < T && q0 + qq < T) ? S[qq * Tpad + tk] : 0.f;
    }
    // ---- o = p v + rw Ev: thread = (query, 1/8 of the d output channels, strided) ----
    constexpr int MAXE = RA_MAXD / 8;
    float acc[MAXE];
#pragma unroll
    for (int e = 0; e < MAXE; ++e) acc[e] = 0.f;
    const int c0 = tid & 7;
    for (int j0 = 0; j0 < T; j0 += RA_KB) {
        __syncthreads();
        for (int i = tid; i < RA_KB * d; i += 256) {
            const int kj = i / d, c = i - kj * d;
            const int t = min(j0 + kj, T - 1);
            kv[kj * dp + c] = vb[(int64_t)t * ldv + c];
        }
        __syncthreads();
        const int nj = min(RA_KB, T - j0);
        for (int kj = 0; kj < nj; ++kj) {
            const float pj = S[qi * Tpad + j0 + kj];
#pragma unroll
            for (int e = 0; e < MAXE; ++e) {
                const int c = c0 + 8 * e;
                if (c < d) acc[e] = fmaf(pj, kv[kj * dp + c], acc[e]);
            }
        }
    }
    if (tq < T) {
#pragma unroll
        for (int e = 0; e < MAXE; ++e) {
            const int c = c0 + 8 * e;
            if (c < d) {
                float a = acc[e];
                for (int r = 0; r < R; ++r) a = fmaf(rl[qi * (2 * RA_MAXW + 1) + r], emb_v[r * d + c], a);   // :280-283
                out[((int64_t)b * T + tq) * ldo + h * d + c] = a;
            }
        }
    }
}

// y[r, c] = x[r, c] * s[r] (+ res[r, c])   (x * x_mask around the FFN convs, attentions.py:406-413; + positional embedding)
// divide != 0: y = x / max(s[r], 1e-12) — F.normalize with s = the row norms (CLAP embeddings, clap/open_clip/model.py:745)
__global__ void rowscale_add_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                    const float* __restrict__ res, float* __restrict__ y, int64_t rows, int C4,
                                    int divide) {
    const int64_t total = rows * C4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / C4;
        f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        if (divide) v = v / fmaxf(s[r], 1e-12f);
        else v = v * s[r];
        if (res) v += reinterpret_cast<const f32x4*>(res)[i];
        reinterpret_cast<f32x4*>(y)[i] = v;
    }
}

}  // namespace aldm

using namespace aldm;

extern "C" int aldm_rel_attention(const float* q, const float* k, const float* v, float* out, int B, int heads, int T,
                                  int d, int ldq, int ldk, int ldv, int ldo, const float* emb_k, const float* emb_v,
                                  int window, const float* mask, void* stream) {
    ALDM_CHECK(q && k && v && out && emb_k && emb_v && mask, "aldm_rel_attention: null pointer");
    ALDM_CHECK(B > 0 && heads > 0 && T > 0 && d > 0 && d <= RA_MAXD && window >= 0 && window <= RA_MAXW,
               "aldm_rel_attention: bad sizes (d <= %d, window <= %d)", RA_MAXD, RA_MAXW);
    const int Tpad = (T + 3) / 4 * 4;
    const size_t lds = ((size_t)(RA_QB + RA_KB) * (d + 1) + RA_QB * (2 * RA_MAXW + 1) + (size_t)RA_QB * Tpad) * 4;
    ALDM_CHECK(lds <= 150 * 1024, "aldm_rel_attention: T=%d needs %zu bytes of LDS", T, lds);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rel_attention_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            150 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL(rel_attention_kernel, dim3(cdiv(T, RA_QB), heads, B), dim3(256), lds, (hipStream_t)stream, q, k, v,
                       out, T, d, ldq, ldk, ldv, ldo, emb_k, emb_v, window, mask, Tpad);
    ALDM_LAUNCH_CHECK("aldm_rel_attention");
    return 0;
}

extern "C" int aldm_rowscale_add(const float* x, const float* s, const float* res, float* y, int64_t rows, int C,
                                 int divide, void* stream) {
    ALDM_CHECK(x && s && y && rows > 0 && C > 0 && C % 4 == 0, "aldm_rowscale_add: bad args");
    ALDM_CHECK(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(res)) & 15) == 0,
               "aldm_rowscale_add: operands must be 16-byte aligned");
    const int64_t total = rows * (C / 4);
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(rowscale_add_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, s, res, y, rows, C / 4, divide);
    ALDM_LAUNCH_CHECK("aldm_rowscale_add");
    return 0;
}
