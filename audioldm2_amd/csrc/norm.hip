// norm.hip — GroupNorm statistics (-> per-(sample, channel) affine consumed by the igemm prologue),
// LayerNorm, row softmax.  All HBM-bound: one coalesced 16-byte read per element, fp32 math,
// deterministic reduction order (no atomics).
#include <stdlib.h>
#include "igemm_epilogue.h"

namespace aldm {

// ---- GroupNorm -------------------------------------------------------------------------------
// x = x1 ++ x2 (channels-last, [B, P, C1] / [B, P, C2]); G groups of Cg = C/G channels, Cg % 4 == 0
// so each float4 column belongs to exactly one group.
// Pass 1: grid (chunks, B).  Thread (tx, ty): float4 column tx (+ 256-strided extra columns when
//         C/4 > 256), pixels p0 + ty + rows*i.  Per-thread (sum, sumsq) -> LDS -> one thread per
//         group adds the group's thread partials in a fixed order -> ws[b][chunk][g] = {sum, sumsq}.
// Pass 2: grid (B).  double-precision combine of the chunk partials, mean/rstd, then
//         scale[b,c] = rstd*gamma[c], shift[b,c] = beta[c] - mean*rstd*gamma[c].
constexpr int GN_ITERS = 16;
constexpr int GN_UNROLL = 8;  // pixel loads in flight per thread

// Numerics: E[x^2] - E[x]^2 over raw fp32 sums loses the variance as soon as |mean| >> std (a trained checkpoint's
// post-conv activations; ATen's GroupNorm is Welford).  Here every thread accumulates sum / sum of squares of
// (x - pivot) with pivot = the first value it sees (same group, so x - pivot is of the order of the group's spread),
// turns them into (n, mean, M2 = sum (x - mean)^2) and partials are merged with Chan's parallel-variance formula in
// fp64, in a fixed order (deterministic, no atomics): M2 = sum M2_t + sum n_t (mean_t - mean)^2.
struct GnPart {
    float n, mean, m2;
};

__device__ __forceinline__ void gn_merge(double& n, double& mean, double& m2, double nb, double mb, double m2b) {
    if (nb <= 0.0) return;
    const double nt = n + nb;
    const double dlt = mb - mean;
    mean += dlt * (nb / nt);
    m2 += m2b + dlt * dlt * (n * nb / nt);
    n = nt;
}

// FUSED: statistics AND the affine in one launch, for samples small enough that the two-launch form is pure latency: block
// (gs, b) owns `gpb` whole groups of sample b — groups are independent, so it reads all P pixels of its channel slab, merges
// and finalises alone.  (Round 1 used one block per sample: 16 blocks on 256 CUs, 13 us for 100-800 KB.)
template <bool FUSED>
__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x1,
                                                         const float* __restrict__ x2, int P,
                                                         int C1, int C2, int G, int cols, int rows,
                                                         int chunk_px, float* __restrict__ ws, float eps,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta,
                                                         float* __restrict__ scale, float* __restrict__ shift, int gpb,
                                                         char* __restrict__ dst = nullptr, char* __restrict__ dst_raw = nullptr,
                                                         int parts = 3, int act = ALDM_ACT_NONE, int raw_parts = 3,
                                                         float f16_scale = 0.f) {
    const int C = C1 + C2;
    const int Cg4 = (C / G) >> 2;
    const int b = blockIdx.y;
    const int chunk = FUSED ? 0 : blockIdx.x;
    const int g_lo = FUSED ? blockIdx.x * gpb : 0;         // first group of this block; it owns groups g_lo .. g_lo + gpb - 1
    const int c4_lo = g_lo * Cg4, C4 = c4_lo + gpb * Cg4;  // its float4 columns [c4_lo, C4)
    const int p0 = chunk * chunk_px;
    const int p1 = min(P, p0 + chunk_px);
    const int tid = threadIdx.x;
    const int tx = tid % cols, ty = tid / cols;
    __shared__ GnPart ps[256];
    // columns handled in passes of `cols` (cols = min(columns of the block, 256))
    const int npass = (C4 - c4_lo + cols - 1) / cols;
    __shared__ double gacc[64][3];   // running (n, mean, M2) per group of this block (index: g - g_lo)
    if (tid < 64) gacc[tid][0] = gacc[tid][1] = gacc[tid][2] = 0.0;
    __syncthreads();
    for (int cp = 0; cp < npass; ++cp) {
        const int c4 = c4_lo + cp * cols + tx;
        GnPart part = {0.f, 0.f, 0.f};
        if (ty < rows && c4 < C4 && p0 + ty < p1) {
            const int c = c4 << 2;
            const bool first = c < C1;
            const float* src = first ? x1 + (int64_t)b * P * C1 + c : x2 + (int64_t)b * P * C2 + (c - C1);
            const int pitch = first ? C1 : C2;
            // GN_UNROLL independent loads in flight per thread: with one load per trip the fused form (one
            // block per sample, a thread walks up to 256 pixels) was a chain of ~130 ns round trips, 34 us
            // for 650 KB.  Fixed accumulator assignment and combine order -> still deterministic.
            float sa[GN_UNROLL], sq[GN_UNROLL];
#pragma unroll
            for (int u = 0; u < GN_UNROLL; ++u) sa[u] = sq[u] = 0.f;
            int p = p0 + ty;
            const float pivot = src[(int64_t)p * pitch];
            int cnt = 0;
            for (; p + (GN_UNROLL - 1) * rows < p1; p += GN_UNROLL * rows) {
                f32x4 v[GN_UNROLL];
#pragma unroll
                for (int u = 0; u < GN_UNROLL; ++u)
                    v[u] = *reinterpret_cast<const f32x4*>(src + (int64_t)(p + u * rows) * pitch) - pivot;
#pragma unroll
                for (int u = 0; u < GN_UNROLL; ++u) {
                    sa[u] += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
                    sq[u] += (v[u][0] * v[u][0] + v[u][1] * v[u][1]) + (v[u][2] * v[u][2] + v[u][3] * v[u][3]);
                }
                cnt += GN_UNROLL;
            }
            for (; p < p1; p += rows) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(src + (int64_t)p * pitch) - pivot;
                sa[0] += (v[0] + v[1]) + (v[2] + v[3]);
                sq[0] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                ++cnt;
            }
            float s = 0.f, ss = 0.f;
#pragma unroll
            for (int u = 0; u < GN_UNROLL; u += 2) {
                s += sa[u] + sa[u + 1];
                ss += sq[u] + sq[u + 1];
            }
            const float n = 4.0f * (float)cnt;
            const float md = s / n;   // mean of (x - pivot)
            part.n = n;
            part.mean = pivot + md;
            part.m2 = fmaxf(ss - s * md, 0.f);
        }
        if constexpr (FUSED) {
            // the block owns few groups (gpb <= 4) spread over all 256 threads: per group, a butterfly of Chan merges over the
            // wave (fixed pattern -> deterministic), then thread 0 folds the four wave results in order
            __shared__ double wred[4][3];
            const int mygl = (c4 - c4_lo) / Cg4;
            for (int gl = 0; gl < gpb; ++gl) {
                const bool mine = part.n > 0.f && mygl == gl;
                double n = mine ? (double)part.n : 0.0, mean = mine ? (double)part.mean : 0.0, m2 = mine ? (double)part.m2 : 0.0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const double nb = __shfl_xor(n, o), mb = __shfl_xor(mean, o), m2b = __shfl_xor(m2, o);
                    gn_merge(n, mean, m2, nb, mb, m2b);
                }
                if ((tid & 63) == 0) {
                    wred[tid >> 6][0] = n;
                    wred[tid >> 6][1] = mean;
                    wred[tid >> 6][2] = m2;
                }
                __syncthreads();
                if (tid == 0) {
                    double an = gacc[gl][0], am = gacc[gl][1], a2 = gacc[gl][2];
                    for (int w = 0; w < 4; ++w) gn_merge(an, am, a2, wred[w][0], wred[w][1], wred[w][2]);
                    gacc[gl][0] = an;
                    gacc[gl][1] = am;
                    gacc[gl][2] = a2;
                }
                __syncthreads();
            }
            continue;
        }
        ps[tid] = part;
        __syncthreads();
        // groups touched by this column pass: c4 in [c4_lo + cp*cols, c4_lo + cp*cols + cols)
        // one thread per group merges, in fixed order, all (tx, ty) of that group.
        if (tid < gpb) {
            const int g = g_lo + tid;
            const int base = c4_lo + cp * cols;
            const int lo = max(g * Cg4, base), hi = min((g + 1) * Cg4, min(C4, base + cols));
            if (lo < hi) {
                double n = gacc[tid][0], mean = gacc[tid][1], m2 = gacc[tid][2];
                for (int yy = 0; yy < rows; ++yy)
                    for (int cc = lo; cc < hi; ++cc) {
                        const GnPart& t = ps[yy * cols + (cc - base)];
                        gn_merge(n, mean, m2, (double)t.n, (double)t.mean, (double)t.m2);
                    }
                gacc[tid][0] = n;
                gacc[tid][1] = mean;
                gacc[tid][2] = m2;
            }
        }
        __syncthreads();
    }
    if (FUSED) {
        const int Cg = C / G;
        __shared__ float fin[64][2];
        if (tid < gpb) {
            const double n = gacc[tid][0];
            double var = n > 0.0 ? gacc[tid][2] / n : 0.0;
            if (var < 0.0) var = 0.0;
            fin[tid][0] = (float)gacc[tid][1];
            fin[tid][1] = (float)(1.0 / sqrt(var + (double)eps));
        }
        __syncthreads();
        __shared__ float lsc[256], lsh[256];   // this block's channel slab (gpb * Cg <= 256 channels when dst is set)
        for (int c = g_lo * Cg + tid; c < (g_lo + gpb) * Cg; c += 256) {
            const int g = c / Cg - g_lo;
            const float sc = fin[g][1] * (gamma ? gamma[c] : 1.f);
            const float sh = (beta ? beta[c] : 0.f) - fin[g][0] * sc;
            scale[(int64_t)b * C + c] = sc;
            shift[(int64_t)b * C + c] = sh;
            if (dst) {
                lsc[c - g_lo * Cg] = sc;
                lsh[c - g_lo * Cg] = sh;
            }
        }
        if (dst) {
            // GroupNorm apply + activation + operand split of the block's own slab, in the same launch (the slab was just read
            // for the statistics: it comes back from L2).  Same arithmetic, in the same order, as split_rows_kernel — the
            // image is bitwise the one aldm_groupnorm_stats + aldm_split_rows write.  One thread = 8 consecutive channels
            // of one pixel (host: slab % 8 == 0, C1 % 8 == 0, C % 32 == 0).
            __syncthreads();
            const int c_lo = g_lo * Cg, sw = (gpb * Cg) >> 3;
            for (int i = tid; i < P * sw; i += 256) {
                const int p = i / sw;
                const int cl = (i - p * sw) << 3;
                const int c = c_lo + cl;
                const int64_t row = (int64_t)b * P + p;
                const float* src = c < C1 ? x1 + row * C1 + c : x2 + row * C2 + (c - C1);
                f32x4 v0 = *reinterpret_cast<const f32x4*>(src);
                f32x4 v1 = *reinterpret_cast<const f32x4*>(src + 4);
                const int64_t off = (row * (C >> 5) + (c >> 5)) * (64 * parts) + (c & 31) * 2;
                u32x2 p0[3], p1[3];
                if (dst_raw) {   // (always a bf16 image: the raw values have no a-priori bound)
                    const int64_t roff = (row * (C >> 5) + (c >> 5)) * (64 * raw_parts) + (c & 31) * 2;
                    split4_parts(v0, p0, raw_parts);
                    split4_parts(v1, p1, raw_parts);
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        if (q < raw_parts)
                            *reinterpret_cast<u32x4*>(dst_raw + roff + q * 64) = u32x4{p0[q][0], p0[q][1], p1[q][0], p1[q][1]};
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v0[e] = __builtin_fmaf(v0[e], lsc[cl + e], lsh[cl + e]);
                    v1[e] = __builtin_fmaf(v1[e], lsc[cl + 4 + e], lsh[cl + 4 + e]);
                }
                if (act == ALDM_ACT_SILU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v0[e] = silu_fast(v0[e]);
                        v1[e] = silu_fast(v1[e]);
                    }
                }
                split4_fmt(v0, p0, parts, f16_scale);
                split4_fmt(v1, p1, parts, f16_scale);
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    if (q < parts) *reinterpret_cast<u32x4*>(dst + off + q * 64) = u32x4{p0[q][0], p0[q][1], p1[q][0], p1[q][1]};
            }
        }
    } else if (tid < G) {
        // fp64 partial of this chunk: {n, mean, M2} (mean needs the full precision: it is the pivot of the merge)
        double* w = reinterpret_cast<double*>(ws) + ((int64_t)(b * gridDim.x + chunk) * G + tid) * 3;
        w[0] = gacc[tid][0];
        w[1] = gacc[tid][1];
        w[2] = gacc[tid][2];
    }
}

__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ ws, int chunks,
                                                          int P, int C, int G, float eps,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          float* __restrict__ scale,
                                                          float* __restrict__ shift) {
    const int b = blockIdx.x;
    __shared__ float s_mean[64], s_rstd[64];
    __shared__ double s_part[256][3];
    const int Cg = C / G;
    const double* wsd = reinterpret_cast<const double*>(ws);
    // 256 threads: LPG lanes per group walk that group's chunk partials in a strided, fixed pattern
    // (independent loads in flight instead of one thread chasing `chunks` dependent loads), then a
    // fixed-order LDS combine -> deterministic.
    const int LPG = 256 / G >= 1 ? 256 / G : 1;  // G <= 64 -> LPG >= 4
    {
        const int g = threadIdx.x / LPG, l = threadIdx.x % LPG;
        double n = 0.0, mean = 0.0, m2 = 0.0;
        if (g < G) {
            for (int ch = l; ch < chunks; ch += LPG) {
                const double* w = wsd + ((int64_t)(b * chunks + ch) * G + g) * 3;
                gn_merge(n, mean, m2, w[0], w[1], w[2]);
            }
        }
        s_part[threadIdx.x][0] = n;
        s_part[threadIdx.x][1] = mean;
        s_part[threadIdx.x][2] = m2;
    }
    __syncthreads();
    if (threadIdx.x < G) {
        const int g = threadIdx.x;
        double n = 0.0, mean = 0.0, m2 = 0.0;
        for (int l = 0; l < LPG; ++l) gn_merge(n, mean, m2, s_part[g * LPG + l][0], s_part[g * LPG + l][1], s_part[g * LPG + l][2]);
        double var = n > 0.0 ? m2 / n : 0.0;
        if (var < 0.0) var = 0.0;
        s_mean[g] = (float)mean;
        s_rstd[g] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int g = c / Cg;
        const float ga = gamma ? gamma[c] : 1.f;
        const float be = beta ? beta[c] : 0.f;
        const float sc = s_rstd[g] * ga;
        scale[(int64_t)b * C + c] = sc;
        shift[(int64_t)b * C + c] = be - s_mean[g] * sc;
    }
}

// ---- LayerNorm: one wave64 per R rows, rows kept in registers (C <= 2048), exact two-pass --------
// All R rows' 16-byte loads are issued before the first reduction.  Which R is best is an occupancy question, measured
// (tools/ln_bench.py): with few rows (round 1: M = 2048) R = 4 independent rows per wave hid the latency of the two dependent
// shuffle trees; at the UNet's M = 4096..16384 one row per wave — four times the waves — does it better.
constexpr int LN_MAXV = 8;

// RMS = true: T5LayerNorm (transformers T5: y = w * x * rsqrt(mean(x^2) + eps), no mean subtraction, no bias).
template <int MAXV, int R, bool RMS = false>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x,
                                                        float* __restrict__ y, int M, int C,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        void* __restrict__ y_split, int parts, float f16_scale) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (row0 >= M) return;
    const int C4 = C >> 2;
    f32x4 v[R][MAXV];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float* xr = x + (int64_t)min(row0 + r, M - 1) * C;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c4 = lane + 64 * i;
            v[r][i] = c4 < C4 ? *reinterpret_cast<const f32x4*>(xr + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    f32x4 ga[MAXV], be[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c4 = min(lane + 64 * i, C4 - 1);
        ga[i] = *reinterpret_cast<const f32x4*>(gamma + 4 * c4);
        be[i] = RMS ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(beta + 4 * c4);
    }
    float mean[R], rstd[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) s += (v[r][i][0] + v[r][i][1]) + (v[r][i][2] + v[r][i][3]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        mean[r] = RMS ? 0.f : s / (float)C;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            if (lane + 64 * i < C4) {
                const f32x4 dlt = v[r][i] - mean[r];
                q += (dlt[0] * dlt[0] + dlt[1] * dlt[1]) + (dlt[2] * dlt[2] + dlt[3] * dlt[3]);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        rstd[r] = 1.0f / sqrtf(q / (float)C + eps);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (row0 + r >= M) break;
        float* yr = y ? y + (int64_t)(row0 + r) * C : nullptr;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c4 = lane + 64 * i;
            if (c4 < C4) {
                const f32x4 o = (v[r][i] - mean[r]) * rstd[r] * ga[i] + be[i];
                if (yr) *reinterpret_cast<f32x4*>(yr + 4 * c4) = o;
                if (y_split) {   // the next GEMM's pre-split A operand
                    if (f16_scale != 0.f) {
                        u32x2 part[3];
                        split4_f16(o, f16_scale, part);
                        char* base = reinterpret_cast<char*>(y_split) + ((int64_t)(row0 + r) * (C >> 5) + (c4 >> 3)) * 128 + (c4 & 7) * 8;
                        *reinterpret_cast<u32x2*>(base) = part[0];
                        *reinterpret_cast<u32x2*>(base + 64) = part[1];
                    } else {
                        split_store4(y_split, row0 + r, C, 4 * c4, o, parts);
                    }
                }
            }
        }
    }
}

// ---- row softmax (VAE mid attention, 4096-wide rows): one block per row, row staged in LDS ----
// MASKED (GPT-2 style attention rows of the sequence generator): row r = (b * heads + h) * q_rows + i; key j takes part
// iff keymask[b, j] != 0 and j <= q_pos0 + i (causal); excluded keys get weight exactly 0 (the reference adds finfo.min
// to them, transformers GPT2Attention: same result whenever a row keeps at least one key, which the always-unmasked
// start token guarantees).
// BIASED (T5 self-attention, transformers T5Attention): row r = (b * heads + h) * q_rows + i gets bias[h, i, :] added and
// keys with keymask[b, j] == 0 the additive finfo.min of the reference's extended attention mask (weight exactly 0 while
// the row keeps a key); no causal limit.
template <bool MASKED, bool BIASED = false>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x,
                                                           float* __restrict__ y, int N,
                                                           float scale, const float* __restrict__ keymask,
                                                           int rows_per_batch, int q_rows, int q_pos0,
                                                           const float* __restrict__ bias = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float srow[];
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const float* xr = x + row * N;
    float* yr = y + row * N;
    const int tid = threadIdx.x;
    float mx = -INFINITY;
    const float* km = nullptr;
    int jmax = N;
    const float* br = nullptr;
    if (MASKED) {
        km = keymask + (row / rows_per_batch) * N;
        jmax = BIASED ? N : q_pos0 + (int)(row % q_rows) + 1;  // keys [0, jmax) are causally visible
    }
    if (BIASED) br = bias + (row % rows_per_batch) * N;          // [heads, q_rows, N] shared by the batch
    for (int i = tid; i < N; i += 256) {
        float v = xr[i] * scale;
        if (BIASED) v += br[i];
        if (MASKED && (i >= jmax || km[i] == 0.0f)) v = -INFINITY;
        srow[i] = v;
        mx = fmaxf(mx, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int i = tid; i < N; i += 256) {
        const float e = (MASKED && srow[i] == -INFINITY) ? 0.0f : expf(srow[i] - mx);
        srow[i] = e;
        s += e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    const float inv = 1.0f / s;
    for (int i = tid; i < N; i += 256) yr[i] = srow[i] * inv;
}

static void gn_geometry(int P, int C, int* cols, int* rows, int* chunk_px, int* chunks) {
    const int C4 = C / 4;
    *cols = C4 < 256 ? C4 : 256;
    *rows = 256 / *cols;
    if (*rows < 1) *rows = 1;
    *chunk_px = *rows * GN_ITERS;
    *chunks = (P + *chunk_px - 1) / *chunk_px;
}

}  // namespace aldm

using namespace aldm;

extern "C" int64_t aldm_gn_ws_floats(int B, int P, int C, int G) {
    int cols, rows, chunk_px, chunks;
    gn_geometry(P, C, &cols, &rows, &chunk_px, &chunks);
    return (int64_t)B * chunks * G * 6 + 2;   // {n, mean, M2} in fp64 per (sample, chunk, group), 8-byte aligned
}

extern "C" int aldm_split_rows(const float* x1, const float* x2, int C1, int C2, int64_t rows, int P, const float* scale,
                               const float* shift, int act, void* dst, void* dst_raw, int parts, void* stream);
extern "C" int aldm_split_rows_f16(const float* x1, const float* x2, int C1, int C2, int64_t rows, int P, const float* scale,
                                   const float* shift, int act, float slope, void* dst, void* dst_raw, int raw_parts, float f16_scale,
                                   void* stream);

static int groupnorm_launch(const float* x1, const float* x2, int B, int P, int C1, int C2, int G, float eps,
                            const float* gamma, const float* beta, float* scale, float* shift, float* ws, void* dst,
                            void* dst_raw, int parts, int act, void* stream, int raw_parts = 0, float f16_scale = 0.f) {
    if (raw_parts == 0) raw_parts = parts;
    if (!x2) C2 = 0;
    const int C = C1 + C2;
    ALDM_CHECK(x1 && scale && shift && ws, "aldm_groupnorm_stats: null pointer");
    ws = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(ws) + 7) & ~uintptr_t(7));   // fp64 partials
    ALDM_CHECK(G > 0 && G <= 64 && C % G == 0 && (C / G) % 4 == 0 && C1 % 4 == 0,
               "aldm_groupnorm_stats: need C%%G==0, (C/G)%%4==0, C1%%4==0 (C1=%d C2=%d G=%d)", C1, C2, G);
    int cols, rows, chunk_px, chunks;
    gn_geometry(P, C, &cols, &rows, &chunk_px, &chunks);
    hipStream_t st = (hipStream_t)stream;
    static const int64_t fused_max = [] {   // A/B override: largest sample (elements) handled by the one-launch form
        const char* e = getenv("ALDM_GN_FUSED_MAX");
        return e ? (int64_t)atoll(e) : (int64_t)1 << 17;
    }();
    // (2^17 elements per sample since round 6: round 2 chose 2^20 on isolated launches; INSIDE the replayed bf16x6 step the level-1
    //  slabs — 1024 pixels x 256-640 channels — are 0.05-0.12 ms per step faster on the chunked statistics + finalize + split_rows form,
    //  f16x3 indifferent: profiles/r06_step_ab_gn_fused_max.txt)
    // group slices per sample: enough blocks to cover the chip (>= 256 / B), each owning gpb = G / gs whole groups
    int gs = 1;
    while (gs < G && gs * B < 256 && G % (gs * 2) == 0) gs *= 2;
    const int gpb = G / gs;
    // one launch, blocks own whole groups: up to 1024 pixels per sample (measured, tools/gn_bench.py / profiles/r02_gn_bench.txt:
    // 6.6-8.5 us against 7.9-12.4 for the chunked two-launch form; at 4096 pixels a block's narrow channel slab reads 32 bytes
    // per 512-byte row and loses, 24.6 vs 12.0 us)
    if ((int64_t)P * C <= fused_max && P <= 1024 && gpb <= 4) {
        const int c4b = gpb * (C / G) / 4;
        const int fcols = c4b < 256 ? c4b : 256, frows = 256 / fcols;
        const int slab = gpb * (C / G);
        static const bool fuse_split = [] {   // A/B override: ALDM_GN_SPLIT_FUSED=0 keeps statistics and split in two launches
            const char* e = getenv("ALDM_GN_SPLIT_FUSED");
            return e == nullptr || e[0] != '0';
        }();
        if (dst && fuse_split && slab % 8 == 0 && slab <= 256 && C1 % 8 == 0) {
            // statistics + apply + activation + operand split in ONE launch (round 3)
            hipLaunchKernelGGL(gn_partial_kernel<true>, dim3(gs, B), dim3(256), 0, st, x1, x2, P, C1, C2, G, fcols, frows, P, ws,
                               eps, gamma, beta, scale, shift, gpb, reinterpret_cast<char*>(dst),
                               reinterpret_cast<char*>(dst_raw), parts, act, raw_parts, f16_scale);
            ALDM_LAUNCH_CHECK("aldm_groupnorm_split");
            return 0;
        }
        hipLaunchKernelGGL(gn_partial_kernel<true>, dim3(gs, B), dim3(256), 0, st, x1, x2, P, C1, C2, G, fcols,
                           frows, P, ws, eps, gamma, beta, scale, shift, gpb, nullptr, nullptr, 3, ALDM_ACT_NONE);
    } else {
        hipLaunchKernelGGL(gn_partial_kernel<false>, dim3(chunks, B), dim3(256), 0, st, x1, x2, P, C1, C2, G,
                           cols, rows, chunk_px, ws, eps, gamma, beta, scale, shift, G);
        hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, st, ws, chunks, P, C, G, eps, gamma,
                           beta, scale, shift);
    }
    ALDM_LAUNCH_CHECK("aldm_groupnorm_stats");
    if (dst) {   // the sample is too large for the one-launch form (or the slab does not split into 16-byte pieces)
        if (f16_scale != 0.f)
            return aldm_split_rows_f16(x1, x2, C1, C2, (int64_t)B * P, P, scale, shift, act, 0.f, dst, dst_raw, raw_parts, f16_scale, stream);
        return aldm_split_rows(x1, x2, C1, C2, (int64_t)B * P, P, scale, shift, act, dst, dst_raw, parts, stream);
    }
    return 0;
}

extern "C" int aldm_groupnorm_stats(const float* x1, const float* x2, int B, int P, int C1, int C2,
                                    int G, float eps, const float* gamma, const float* beta,
                                    float* scale, float* shift, float* ws, void* stream) {
    return groupnorm_launch(x1, x2, B, P, C1, C2, G, eps, gamma, beta, scale, shift, ws, nullptr, nullptr, 3, ALDM_ACT_NONE,
                            stream);
}

extern "C" int aldm_groupnorm_split(const float* x1, const float* x2, int B, int P, int C1, int C2, int G, float eps,
                                    const float* gamma, const float* beta, int act, float* scale, float* shift, float* ws,
                                    void* dst, void* dst_raw, int parts, void* stream) {
    if (!x2) C2 = 0;
    ALDM_CHECK(dst != nullptr && (parts == 2 || parts == 3) && (C1 + C2) % 32 == 0 && C1 % 8 == 0 && C2 % 8 == 0,
               "aldm_groupnorm_split: need dst, parts 2|3, (C1+C2) %% 32 == 0, C1 %% 8 == 0 (C1=%d C2=%d)", C1, C2);
    ALDM_CHECK(act == ALDM_ACT_NONE || act == ALDM_ACT_SILU, "aldm_groupnorm_split: activation %d not supported", act);
    ALDM_CHECK(((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(dst_raw)) & 15) == 0,
               "aldm_groupnorm_split: split images must be 16-byte aligned");
    return groupnorm_launch(x1, x2, B, P, C1, C2, G, eps, gamma, beta, scale, shift, ws, dst, dst_raw, parts, act, stream);
}

// the "f16x3" forms (aldm_igemm_desc.a_fmt): the split image is the 2-part fp16 image of f16_scale * value
extern "C" int aldm_groupnorm_split_f16(const float* x1, const float* x2, int B, int P, int C1, int C2, int G, float eps,
                                        const float* gamma, const float* beta, int act, float* scale, float* shift, float* ws,
                                        void* dst, void* dst_raw, int raw_parts, float f16_scale, void* stream) {
    if (!x2) C2 = 0;
    ALDM_CHECK(dst != nullptr && f16_scale > 0.0f && (raw_parts == 2 || raw_parts == 3) && (C1 + C2) % 32 == 0 && C1 % 8 == 0 && C2 % 8 == 0,
               "aldm_groupnorm_split_f16: need dst, f16_scale > 0, raw_parts 2|3, (C1+C2) %% 32 == 0, C1 %% 8 == 0 (C1=%d C2=%d)", C1, C2);
    ALDM_CHECK(act == ALDM_ACT_NONE || act == ALDM_ACT_SILU, "aldm_groupnorm_split_f16: activation %d not supported", act);
    ALDM_CHECK(((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(dst_raw)) & 15) == 0,
               "aldm_groupnorm_split_f16: split images must be 16-byte aligned");
    return groupnorm_launch(x1, x2, B, P, C1, C2, G, eps, gamma, beta, scale, shift, ws, dst, dst_raw, 2, act, stream, raw_parts, f16_scale);
}

static int layernorm_launch(const float* x, float* y, void* y_split, int parts, int M, int C, const float* gamma,
                            const float* beta, float eps, void* stream, const char* name, bool rms = false, float f16_scale = 0.f) {
    ALDM_CHECK(parts == 2 || parts == 3, "%s: parts must be 2 or 3", name);
    ALDM_CHECK(x && (y || y_split) && gamma && (beta || rms), "%s: null pointer", name);
    ALDM_CHECK(C % 4 == 0 && C <= 256 * LN_MAXV, "%s: C=%d must be a multiple of 4 and <= %d", name, C, 256 * LN_MAXV);
    ALDM_CHECK(y_split == nullptr || (C % 32 == 0 && (reinterpret_cast<uintptr_t>(y_split) & 15) == 0),
               "%s: a split-image output needs C %% 32 == 0 and 16-byte alignment", name);
    hipStream_t st = (hipStream_t)stream;
#define ALDM_LN(V_, R_)                                                                                             \
    do {                                                                                                            \
        if (rms)                                                                                                    \
            hipLaunchKernelGGL((layernorm_kernel<V_, R_, true>), dim3(cdiv(M, 4 * R_)), dim3(256), 0, st, x, y, M, C, \
                               gamma, beta, eps, y_split, parts, f16_scale);                                        \
        else                                                                                                        \
            hipLaunchKernelGGL((layernorm_kernel<V_, R_, false>), dim3(cdiv(M, 4 * R_)), dim3(256), 0, st, x, y, M, \
                               C, gamma, beta, eps, y_split, parts, f16_scale);                                     \
    } while (0)
    const int nv = cdiv(C / 4, 64);
    static const int env_r = [] {   // A/B override (tools/ln_bench.py): rows per wave for C <= 512
        const char* e = getenv("ALDM_LN_R");
        return e ? atoi(e) : 0;
    }();
    // one row per wave up to C = 512: more, shorter waves hide the load latency better than 4 rows in flight per wave
    // (M = 16384, C = 256: 9.2 -> 7.9 us; M = 4096, C = 384: 6.9 -> 4.6 us, profiles/r02_ln_bench.txt)
    if (nv <= 1) {
        if (env_r == 4) ALDM_LN(1, 4);
        else if (env_r == 2) ALDM_LN(1, 2);
        else ALDM_LN(1, 1);
    } else if (nv <= 2) {
        if (env_r == 4) ALDM_LN(2, 4);
        else if (env_r == 2) ALDM_LN(2, 2);
        else ALDM_LN(2, 1);
    } else if (nv <= 4) ALDM_LN(4, 2);
    else ALDM_LN(8, 1);
#undef ALDM_LN
    ALDM_LAUNCH_CHECK(name);
    return 0;
}

extern "C" int aldm_layernorm(const float* x, float* y, int M, int C, const float* gamma,
                              const float* beta, float eps, void* stream) {
    return layernorm_launch(x, y, nullptr, 3, M, C, gamma, beta, eps, stream, "aldm_layernorm");
}

extern "C" int aldm_layernorm_split(const float* x, float* y, void* y_split, int M, int C, const float* gamma,
                                    const float* beta, float eps, int parts, void* stream) {
    return layernorm_launch(x, y, y_split, parts, M, C, gamma, beta, eps, stream, "aldm_layernorm_split");
}

extern "C" int aldm_layernorm_split_f16(const float* x, float* y, void* y_split, int M, int C, const float* gamma,
                                        const float* beta, float eps, float f16_scale, void* stream) {
    ALDM_CHECK(y_split != nullptr && f16_scale > 0.0f, "aldm_layernorm_split_f16: need y_split and f16_scale > 0");
    return layernorm_launch(x, y, y_split, 2, M, C, gamma, beta, eps, stream, "aldm_layernorm_split_f16", false, f16_scale);
}

extern "C" int aldm_rmsnorm(const float* x, float* y, int M, int C, const float* weight, float eps, void* stream) {
    return layernorm_launch(x, y, nullptr, 3, M, C, weight, nullptr, eps, stream, "aldm_rmsnorm", true);
}

extern "C" int aldm_softmax_rows(const float* x, float* y, int64_t M, int N, float scale,
                                 void* stream) {
    ALDM_CHECK(x && y && M > 0 && N > 0, "aldm_softmax_rows: bad args");
    ALDM_CHECK((int64_t)N * 4 <= 60 * 1024, "aldm_softmax_rows: row of %d floats exceeds the 60 KiB LDS stage", N);
    ALDM_CHECK(M < (1ll << 31), "aldm_softmax_rows: too many rows");
    hipLaunchKernelGGL(softmax_rows_kernel<false>, dim3((unsigned)M), dim3(256), (size_t)N * 4,
                       (hipStream_t)stream, x, y, N, scale, nullptr, 1, 1, 0);
    ALDM_LAUNCH_CHECK("aldm_softmax_rows");
    return 0;
}

extern "C" int aldm_softmax_rows_masked(const float* x, float* y, int B, int heads, int q_rows, int N, float scale,
                                        const float* keymask, int q_pos0, void* stream) {
    ALDM_CHECK(x && y && keymask && B > 0 && heads > 0 && q_rows > 0 && N > 0 && q_pos0 >= 0,
               "aldm_softmax_rows_masked: bad args");
    ALDM_CHECK((int64_t)N * 4 <= 60 * 1024, "aldm_softmax_rows_masked: row of %d floats exceeds the 60 KiB LDS stage", N);
    const int64_t M = (int64_t)B * heads * q_rows;
    ALDM_CHECK(M < (1ll << 31), "aldm_softmax_rows_masked: too many rows");
    hipLaunchKernelGGL(softmax_rows_kernel<true>, dim3((unsigned)M), dim3(256), (size_t)N * 4, (hipStream_t)stream, x, y,
                       N, scale, keymask, heads * q_rows, q_rows, q_pos0);
    ALDM_LAUNCH_CHECK("aldm_softmax_rows_masked");
    return 0;
}

extern "C" int aldm_softmax_rows_bias(const float* x, float* y, int B, int heads, int q_rows, int N, float scale,
                                      const float* bias, const float* keymask, void* stream) {
    ALDM_CHECK(x && y && bias && keymask && B > 0 && heads > 0 && q_rows > 0 && N > 0, "aldm_softmax_rows_bias: bad args");
    ALDM_CHECK((int64_t)N * 4 <= 60 * 1024, "aldm_softmax_rows_bias: row of %d floats exceeds the 60 KiB LDS stage", N);
    const int64_t M = (int64_t)B * heads * q_rows;
    ALDM_CHECK(M < (1ll << 31), "aldm_softmax_rows_bias: too many rows");
    hipLaunchKernelGGL((softmax_rows_kernel<true, true>), dim3((unsigned)M), dim3(256), (size_t)N * 4, (hipStream_t)stream, x,
                       y, N, scale, keymask, heads * q_rows, q_rows, 0, bias);
    ALDM_LAUNCH_CHECK("aldm_softmax_rows_bias");
    return 0;
}
