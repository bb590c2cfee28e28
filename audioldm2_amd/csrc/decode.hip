// Single-position decode step of the GPT-2 AudioMAE-token sequence generator (audiomae_gen/sequence_input.py:294-325: 512
// dependent GPT-2 forwards of ONE new position for the speech model; transformers GPT2Block = ln_1 -> c_attn -> attention over the
// key/value cache -> c_proj (+ residual) -> ln_2 -> c_fc -> gelu_new -> c_proj (+ residual)).
//
// At M <= 16 rows the GEMMs are weight streams (GPT-2 base: 340 MB of fp32 weights per token), not matrix-core work: 14 MFLOP for
// the largest.  Two kernels replace the ~17 launches per layer of the general path (LayerNorm, tile GEMM on M = 8 rows, head
// split / merge copies, cache index_copy, scores GEMM, softmax, V repack, P.V GEMM):
//   decode_linear_kernel     y = act(LN(x) W + b) + res,   W fp32 [K, N] as GPT-2's Conv1D stores it.  A block owns 32 output
//                            columns (128-byte weight row segments) and ALL of K: its 8 waves split the rows, every lane keeps up
//                            to 64 row segments in flight, partial sums meet in LDS in a fixed order.  Exact fp32 FMA: no operand
//                            splits, no MFMA, deterministic, no cross-block reduction.
//   decode_attention_kernel  one block per (sample, head): appends the new key / value to the caches at *pos, scores against
//                            every switched-on key, softmax, P.V, output already head-merged.
#include "common.h"

namespace aldm {

#ifndef ALDM_DL_CT      // A/B builds (tools/gpu/build_variant.sh): 16 = twice the blocks, 64-byte row segments
#define ALDM_DL_CT 32
#endif
#ifndef ALDM_DL_NL
#define ALDM_DL_NL 64
#endif
constexpr int DL_CT = ALDM_DL_CT;      // output columns per block: one 128-byte segment of a weight row
constexpr int DL_RPI = 64 / DL_CT;     // weight rows one wave-wide load instruction covers
constexpr int DL_WAVES = 8;
constexpr int DL_NL = ALDM_DL_NL;      // most row segments one lane keeps in flight
constexpr int DL_PASS = DL_WAVES * DL_RPI * DL_NL;   // k's per pass (1024)
constexpr int DL_MAXM = 16;
constexpr int DL_LN_MAXK = 1024;       // rows the fused LayerNorm holds in registers (GPT-2: 768)

// LDS hand-off barrier that leaves global loads in flight (__syncthreads() drains vmcnt as well: the weight rows issued at the top
// of a pass would have to land before the LayerNorm statistics could start)
__device__ __forceinline__ void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Lane = (sub = lane >> 5, c = lane & 31): column c of the block's 32, weight rows [kb + sub * nl, kb + (sub + 1) * nl) of the wave's
// 2 nl-row slab, all nl loads of the lane in flight at once; a pass covers 16 nl rows (8 waves), npass passes cover K.  No
// cross-block reduction (a device-scope release / acquire pair costs more than these launches run: measured 25 us per launch with
// a ticketed k-split over 200 blocks, profiles/r04_decode_probe.txt), all of K is summed inside the block in a fixed order.
template <int MT>
__global__ __launch_bounds__(512) void decode_linear_kernel(const float* __restrict__ x, int ldx, int M, int K,
                                                            const float* __restrict__ W, int N,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ ln_g,
                                                            const float* __restrict__ ln_b, float eps, int act,
                                                            const float* __restrict__ res, int ldr, float* __restrict__ y,
                                                            int ldy, int npass, int nl) {
    // 80 KB at MT = 16: fits gfx950's 160 KB of LDS per CU (the 64 KB of gfx90a / gfx942 would not — this library is gfx950 only,
    // and says so here instead of failing somewhere inside the build of an overridden ARCH)
    static_assert(sizeof(float) * (MT * DL_PASS + DL_WAVES * MT * DL_CT + 2 * MT) <= 160 * 1024,
                  "decode_linear_kernel: static LDS exceeds the 160 KB of a gfx950 CU");
#if !defined(__gfx950__) && defined(__HIP_DEVICE_COMPILE__)
#error "csrc/decode.hip is written for gfx950 (160 KB LDS per CU); build with --offload-arch=gfx950"
#endif
    __shared__ __attribute__((aligned(16))) float xs[MT][DL_PASS];
    __shared__ float red[DL_WAVES][MT][DL_CT];
    __shared__ float s_mean[MT], s_rstd[MT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform for the compiler: scalar branches, no exec-masked regions
    const int sub = lane / DL_CT, c = lane % DL_CT;
    const int col = blockIdx.x * DL_CT + c;
    const bool col_ok = col < N;
    const int L = DL_WAVES * DL_RPI * nl;       // k's per pass
    const int kb = w * (DL_RPI * nl) + sub * nl;

    float acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = 0.f;

    for (int p = 0; p < npass; ++p) {
        const int kc = p * L;
        // (1) the weight stream first: every row segment this lane needs in this pass is in flight before anything else happens.
        // (All passes' rows at once would make m_proj's K = 3072 one HBM round trip instead of three, but 192 row registers + their
        // addresses spill at 512 threads per block: 600-700 bytes of scratch per lane — not shipped.)
        const float* wp = W + (int64_t)(kc + kb) * N + (col_ok ? col : 0);
        float wv[DL_NL];
#pragma unroll
        for (int g = 0; g < DL_NL / 4; ++g) {
            if (4 * g < nl) {
#pragma unroll
                for (int i = 0; i < 4; ++i) wv[4 * g + i] = wp[(int64_t)(4 * g + i) * N];
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) wv[4 * g + i] = 0.f;
            }
        }
        // (2) LayerNorm statistics of the whole rows, rows held in registers: wave w takes rows w, w + 8; mean, then the centred
        // second moment, like aldm_layernorm.  No per-lane predicate on a load (a divergent branch makes hipcc wait for each one):
        // addresses are clamped, the surplus lanes are zeroed afterwards
        if (ln_g && p == 0) {
            constexpr int RPW = (MT + DL_WAVES - 1) / DL_WAVES, NV = DL_LN_MAXK / 256;
            const int C4 = K >> 2;
            f32x4 v[RPW][NV];
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const f32x4* xr = reinterpret_cast<const f32x4*>(x + (int64_t)min(w + DL_WAVES * r, M - 1) * ldx);
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    if (64 * i < C4) v[r][i] = xr[min(lane + 64 * i, C4 - 1)];
                    else v[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int m = w + DL_WAVES * r;
#pragma unroll
                for (int i = 0; i < NV; ++i)
                    if (lane + 64 * i >= C4) v[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < NV; ++i) s += (v[r][i][0] + v[r][i][1]) + (v[r][i][2] + v[r][i][3]);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
                const float mean = s / (float)K;
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const f32x4 d = v[r][i] - mean;
                    const float qq = (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
                    q += lane + 64 * i < C4 ? qq : 0.f;
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
                const float rstd = 1.0f / sqrtf(q / (float)K + eps);
                if (lane == 0 && m < MT) {
                    s_mean[m] = mean;
                    s_rstd[m] = rstd;
                }
            }
            lds_sync();
        }
        // (3) this pass's k range of the (normalised) activations -> LDS
        for (int t = tid; t < L; t += DL_WAVES * 64) {
            const int k = kc + t;
            float g = 1.f, b = 0.f;
            if (ln_g) {
                g = ln_g[k];
                b = ln_b[k];
            }
            float xv[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) xv[m] = m < M ? x[(int64_t)m * ldx + k] : 0.f;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float v = xv[m];
                if (ln_g && m < M) v = (v - s_mean[m]) * s_rstd[m] * g + b;
                xs[m][t] = v;
            }
        }
        lds_sync();
        // (4) FMAs in ascending k (fixed order)
#pragma unroll
        for (int g = 0; g < DL_NL / 4; ++g) {
            if (4 * g < nl) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(&xs[m][kb + 4 * g]);
                    acc[m] = fmaf(xv[0], wv[4 * g + 0], acc[m]);
                    acc[m] = fmaf(xv[1], wv[4 * g + 1], acc[m]);
                    acc[m] = fmaf(xv[2], wv[4 * g + 2], acc[m]);
                    acc[m] = fmaf(xv[3], wv[4 * g + 3], acc[m]);
                }
            }
        }
        lds_sync();
    }
    // the two row groups of a wave, then the eight waves in wave order
#pragma unroll
    for (int o = 32; o >= DL_CT; o >>= 1) {
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] += __shfl_xor(acc[m], o);
    }
    if (lane < DL_CT) {
#pragma unroll
        for (int m = 0; m < MT; ++m) red[w][m][c] = acc[m];
    }
    lds_sync();
    if (tid < MT * DL_CT) {
        const int m = tid / DL_CT, cc = tid % DL_CT;
        const int ocol = blockIdx.x * DL_CT + cc;
        if (m < M && ocol < N) {
            float v = red[0][m][cc];
#pragma unroll
            for (int i = 1; i < DL_WAVES; ++i) v += red[i][m][cc];
            if (bias) v += bias[ocol];
            v = act_apply(v, act, 0.f);
            if (res) v += res[(int64_t)m * ldr + ocol];
            y[(int64_t)m * ldy + ocol] = v;
        }
    }
}

// One 1024-thread block per (sample b, head h); head dim 64.  qkv: [B, 3 E] rows (q | k | v, E = heads * 64) of the NEW position.
// Wave w, lane = (ksub = lane >> 4, d4 = lane & 15): key j = 4 (w + 16 i) + ksub, head dims 4 d4 .. 4 d4 + 3 — one load instruction
// covers four consecutive cache rows (1 KB contiguous), and all of a wave's <= 16 row loads are in flight at once.
__global__ __launch_bounds__(1024) void decode_attention_kernel(const float* __restrict__ qkv, int ldq,
                                                                const int64_t* __restrict__ pos_ptr,
                                                                float* __restrict__ kc, float* __restrict__ vc,
                                                                const float* __restrict__ keymask, int n_tot, int heads,
                                                                float scale, float* __restrict__ out, int ldo,
                                                                int qparts = 1, int64_t qstride = 0,
                                                                const float* __restrict__ qbias = nullptr) {
    __shared__ __attribute__((aligned(16))) float sq[64], sk[64], sv[64];
    __shared__ float sp[1024];
    __shared__ float red_max[16], red_sum[16];
    __shared__ __attribute__((aligned(16))) float so[16][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ksub = lane >> 4, d4 = lane & 15;
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int E = heads * 64;
    const int64_t z = blockIdx.x;
    int64_t pos64 = *pos_ptr;
    const int pos = (int)(pos64 < 0 ? 0 : (pos64 >= n_tot ? n_tot - 1 : pos64));
    const int n_hi = pos + 1;                  // keys 0 .. pos exist; later cache slots are not part of the sequence yet
    float* kz = kc + z * n_tot * 64;
    float* vz = vc + z * n_tot * 64;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    f32x4 kv[16];                              // the cached key rows first: they do not depend on the new position
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        kv[i] = zero4;
        if (4 * (w + 16 * i) < n_hi)   // uniform; no per-lane predicate on the load (clamped row, unused lanes are discarded below)
            kv[i] = *reinterpret_cast<const f32x4*>(kz + (int64_t)min(4 * (w + 16 * i) + ksub, pos) * 64 + 4 * d4);
    }
    if (tid < 64) {
        const float* r = qkv + (int64_t)b * ldq + h * 64 + tid;
        float q = r[0], k = r[E], v = r[2 * E];
        for (int j0 = 1; j0 < qparts; j0 += 4) {   // (round 6) c_attn's K-slices, summed in slab order (four loads in flight), then its bias
            float pq[4], pk[4], pvv[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const bool jon = j0 + jj < qparts;
                const float* rp = r + (jon ? j0 + jj : 0) * qstride;
                pq[jj] = jon ? rp[0] : 0.f;
                pk[jj] = jon ? rp[E] : 0.f;
                pvv[jj] = jon ? rp[2 * E] : 0.f;
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                q += pq[jj];
                k += pk[jj];
                v += pvv[jj];
            }
        }
        if (qbias) {
            q += qbias[h * 64 + tid];
            k += qbias[E + h * 64 + tid];
            v += qbias[2 * E + h * 64 + tid];
        }
        sq[tid] = q;
        sk[tid] = k;
        sv[tid] = v;
        kz[(int64_t)pos * 64 + tid] = k;       // the new position joins the cache (transformers GPT2Attention: torch.cat on layer_past)
        vz[(int64_t)pos * 64 + tid] = v;
    }
    lds_sync();
    const f32x4 qv = *reinterpret_cast<const f32x4*>(&sq[4 * d4]);
    const f32x4 knew = *reinterpret_cast<const f32x4*>(&sk[4 * d4]);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (4 * (w + 16 * i) < n_hi) {
            const int j = 4 * (w + 16 * i) + ksub;
            const f32x4 kk = j == pos ? knew : kv[i];
            float d = fmaf(kk[3], qv[3], fmaf(kk[2], qv[2], fmaf(kk[1], qv[1], kk[0] * qv[0])));
            d += __shfl_xor(d, 1);
            d += __shfl_xor(d, 2);
            d += __shfl_xor(d, 4);
            d += __shfl_xor(d, 8);
            if (d4 == 0 && j < n_hi) sp[j] = d * scale;
        }
    }
    asm volatile("" ::: "memory");             // keep the value loads behind the scores: 16 x 4 registers each, not both at once
    const float kmt = keymask[(int64_t)b * n_tot + min(tid, n_tot - 1)];   // thread t = key t in the softmax phase
    f32x4 vv[16];                              // the cached value rows, in flight under the softmax
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        vv[i] = zero4;
        if (4 * (w + 16 * i) < n_hi)
            vv[i] = *reinterpret_cast<const f32x4*>(vz + (int64_t)min(4 * (w + 16 * i) + ksub, pos) * 64 + 4 * d4);
    }
    lds_sync();
    const float s = (tid < n_hi && kmt != 0.0f) ? sp[tid] : -INFINITY;   // thread t = key t; masked keys weigh 0
    float mx = s;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) red_max[w] = mx;
    lds_sync();
    mx = red_max[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, red_max[i]);
    const float e = s == -INFINITY ? 0.0f : expf(s - mx);
    if (tid < n_hi) sp[tid] = e;
    float sum = e;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) red_sum[w] = sum;
    lds_sync();
    float tot = red_sum[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) tot += red_sum[i];
    const float inv = 1.0f / tot;
    const f32x4 vnew = *reinterpret_cast<const f32x4*>(&sv[4 * d4]);
    f32x4 o4 = zero4;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (4 * (w + 16 * i) < n_hi) {
            const int j = 4 * (w + 16 * i) + ksub;
            const float p = j < n_hi ? sp[j] : 0.0f;
            const f32x4 v4 = j == pos ? vnew : vv[i];
            o4[0] = fmaf(p, v4[0], o4[0]);
            o4[1] = fmaf(p, v4[1], o4[1]);
            o4[2] = fmaf(p, v4[2], o4[2]);
            o4[3] = fmaf(p, v4[3], o4[3]);
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        o4[c] += __shfl_xor(o4[c], 16);
        o4[c] += __shfl_xor(o4[c], 32);
    }
    if (lane < 16) *reinterpret_cast<f32x4*>(&so[w][4 * d4]) = o4;
    lds_sync();
    if (tid < 64) {
        float t = so[0][tid];
#pragma unroll
        for (int i = 1; i < 16; ++i) t += so[i][tid];
        out[(int64_t)b * ldo + h * 64 + tid] = t * inv;
    }
}


// ---- round 6: the same decode step on the WHOLE chip --------------------------------------------------------------------------
// decode_linear_kernel gives one block all of K for its 32 columns: 24 - 96 blocks per launch, each pulling 96 - 393 KB of weights
// through ONE compute unit's memory path (tens of GB/s per CU) — 8 us per 1024-row pass whatever the HBM could deliver, 24.7 us
// for m_proj.  A dependent kernel boundary costs ~1.5 us on this chip (MI355X_MICROARCH.md, row "boundary"), far less than that:
// so the layer is cut into MORE launches that each use every CU for one HBM round trip:
//   decode_gemv_kernel       a block = 32 columns x ONE K-slice of R <= 512 rows (<= 64 KB of weights, all in flight at once,
//                            non-temporal: each byte is read once per token), grid = column tiles x K-slices ~ 288 blocks; writes
//                            the slice's partial sums ypart[slice][m][n].  Its activations are read as act(bias + sum of the
//                            PREVIOUS launch's partial slabs) — that is how m_proj consumes c_fc without a launch in between.
//   decode_reduce_ln_kernel  one block per row: h = res + bias + sum of the partial slabs in slab order (deterministic), and the
//                            LayerNorm of the result for the next GEMV (mean, then the centred second moment, like aldm_layernorm).
//   decode_attention_kernel  takes q | k | v as bias + sum of c_attn's partial slabs.
// 7 launches per GPT-2 block instead of 5, each a fraction of the old ones' time.
constexpr int DG_CT = 32;        // output columns per block
constexpr int DG_WAVES = 4;
constexpr int DG_MAXR = 512;     // rows per K-slice: 8 row groups x <= 64 rows in flight per lane
constexpr int DG_NL = DG_MAXR / (2 * DG_WAVES);

template <int MT>
__global__ __launch_bounds__(256) void decode_gemv_kernel(const float* __restrict__ x, int ldx, int xparts, int64_t xstride,
                                                          const float* __restrict__ xbias, int xact, int M, int K,
                                                          const float* __restrict__ W, int N, float* __restrict__ ypart, int R) {
    __shared__ __attribute__((aligned(16))) float xs[MT][DG_MAXR];
    __shared__ float red[DG_WAVES][MT][DG_CT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = lane >> 5, c = lane & 31;
    const int col = blockIdx.x * DG_CT + c;
    const bool col_ok = col < N;
    const int k0 = blockIdx.y * R;
    const int nl = R / (2 * DG_WAVES);            // rows per lane: a multiple of 4
    const int kb = (2 * w + sub) * nl;
    // (1) the slice's weights: every row segment of this lane in flight before anything else
    const float* wp = W + (int64_t)(k0 + kb) * N + (col_ok ? col : 0);
    float wv[DG_NL];
#pragma unroll
    for (int g = 0; g < DG_NL / 4; ++g) {
        if (4 * g < nl) {
#pragma unroll
            for (int i = 0; i < 4; ++i) wv[4 * g + i] = __builtin_nontemporal_load(wp + (int64_t)(4 * g + i) * N);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) wv[4 * g + i] = 0.f;
        }
    }
    // (2) the slice of the activations -> LDS: plain rows, or act(bias + the previous launch's partial slabs in slab order)
    for (int t = tid; t < R; t += DG_WAVES * 64) {
        const int k = k0 + t;
        float xv[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) xv[m] = m < M ? x[(int64_t)m * ldx + k] : 0.f;
        // the other slabs four at a time: all of a batch's loads in flight, then added in slab order (a slab past the last adds 0)
        for (int j0 = 1; j0 < xparts; j0 += 4) {
            float pv[4][MT];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const bool jon = j0 + jj < xparts;
                const float* xp = x + (jon ? j0 + jj : 0) * xstride + k;
#pragma unroll
                for (int m = 0; m < MT; ++m) pv[jj][m] = (jon && m < M) ? xp[(int64_t)m * ldx] : 0.f;
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
                for (int m = 0; m < MT; ++m) xv[m] += pv[jj][m];
            }
        }
        const float b = xbias ? xbias[k] : 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m) xs[m][t] = m < M ? act_apply(xv[m] + b, xact, 0.f) : 0.f;
    }
    lds_sync();
    // (3) FMAs in ascending k
    float acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = 0.f;
#pragma unroll
    for (int g = 0; g < DG_NL / 4; ++g) {
        if (4 * g < nl) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(&xs[m][kb + 4 * g]);
                acc[m] = fmaf(xv[0], wv[4 * g + 0], acc[m]);
                acc[m] = fmaf(xv[1], wv[4 * g + 1], acc[m]);
                acc[m] = fmaf(xv[2], wv[4 * g + 2], acc[m]);
                acc[m] = fmaf(xv[3], wv[4 * g + 3], acc[m]);
            }
        }
    }
    // (4) the two row groups of a wave, then the four waves in wave order
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] += __shfl_xor(acc[m], 32);
    if (lane < DG_CT) {
#pragma unroll
        for (int m = 0; m < MT; ++m) red[w][m][c] = acc[m];
    }
    lds_sync();
    for (int t = tid; t < MT * DG_CT; t += DG_WAVES * 64) {
        const int m = t / DG_CT, cc = t % DG_CT;
        const int ocol = blockIdx.x * DG_CT + cc;
        if (m < M && ocol < N) {
            float v = red[0][m][cc];
#pragma unroll
            for (int i = 1; i < DG_WAVES; ++i) v += red[i][m][cc];
            ypart[((int64_t)blockIdx.y * M + m) * N + ocol] = v;
        }
    }
}

// One block per row m: v = part[0][m] + part[1][m] + ... (slab order), + bias (row *bias_row of a table when bias_row != null:
// the position embedding), + res; h_out = v (may alias res); xn = LayerNorm(v) gamma + beta when ln_g.  N <= 1024, N % 4 == 0.
__global__ __launch_bounds__(256) void decode_reduce_ln_kernel(const float* __restrict__ part, int nparts, int64_t pstride, int ldp,
                                                               const float* __restrict__ bias, const int64_t* __restrict__ bias_row,
                                                               const float* __restrict__ res, int ldr, int N,
                                                               float* __restrict__ h_out, int ldh,
                                                               const float* __restrict__ ln_g, const float* __restrict__ ln_b, float eps,
                                                               float* __restrict__ xn, int ldn, float* __restrict__ xn2, int ldn2) {
    __shared__ float s_red[4];
    const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c4 = tid * 4;
    const bool on = c4 < N;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (on) {
        // eight slabs at a time: a batch's loads all in flight, then added in slab order (a slab past the last adds 0)
        for (int j0 = 0; j0 < nparts; j0 += 8) {
            f32x4 pv[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                pv[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (j0 + jj < nparts) pv[jj] = *reinterpret_cast<const f32x4*>(part + (j0 + jj) * pstride + (int64_t)m * ldp + c4);
            }
            if (j0 == 0) v = pv[0];
            else v += pv[0];
#pragma unroll
            for (int jj = 1; jj < 8; ++jj) v += pv[jj];
        }
        if (bias) v += *reinterpret_cast<const f32x4*>(bias + (bias_row ? *bias_row * (int64_t)N : 0) + c4);
        if (res) v += *reinterpret_cast<const f32x4*>(res + (int64_t)m * ldr + c4);
        if (h_out) *reinterpret_cast<f32x4*>(h_out + (int64_t)m * ldh + c4) = v;
    }
    if (!ln_g) return;
    float s = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) s_red[w] = s;
    __syncthreads();
    const float mean = ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) / (float)N;
    __syncthreads();
    const f32x4 d = v - mean;
    float q = on ? (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]) : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    if (lane == 0) s_red[w] = q;
    __syncthreads();
    const float rstd = 1.0f / sqrtf(((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) / (float)N + eps);
    if (on) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(ln_g + c4), b = *reinterpret_cast<const f32x4*>(ln_b + c4);
        const f32x4 y = d * rstd * g + b;
        *reinterpret_cast<f32x4*>(xn + (int64_t)m * ldn + c4) = y;
        if (xn2) *reinterpret_cast<f32x4*>(xn2 + (int64_t)m * ldn2 + c4) = y;
    }
}

}  // namespace aldm

using namespace aldm;

extern "C" int aldm_decode_linear(const float* x, int ldx, int M, int K, const float* w_kn, int N, const float* bias,
                                  const float* ln_gamma, const float* ln_beta, float ln_eps, int act, const float* res,
                                  int ldr, float* y, int ldy, void* stream) {
    ALDM_CHECK(x && w_kn && y && M > 0 && M <= DL_MAXM && K > 0 && N > 0, "aldm_decode_linear: bad args (1 <= M <= %d rows)",
               DL_MAXM);
    const int npass = cdiv(K, DL_PASS);
    ALDM_CHECK(K % (64 * npass) == 0, "aldm_decode_linear: K = %d must be a multiple of %d", K, 64 * npass);
    ALDM_CHECK(ldx >= K && ldy >= N && (!res || ldr >= N), "aldm_decode_linear: row pitch shorter than the row");
    ALDM_CHECK(!ln_gamma == !ln_beta, "aldm_decode_linear: LayerNorm needs gamma and beta");
    ALDM_CHECK(!ln_gamma || (K <= DL_LN_MAXK && ldx % 4 == 0 && ((uintptr_t)x & 15) == 0),
               "aldm_decode_linear: the fused LayerNorm holds rows of <= %d floats, 16-byte aligned", DL_LN_MAXK);
    ALDM_CHECK(act == ALDM_ACT_NONE || act == ALDM_ACT_GELU || act == ALDM_ACT_GELU_TANH || act == ALDM_ACT_SILU ||
                   act == ALDM_ACT_TANH,
               "aldm_decode_linear: unsupported activation %d", act);
    const int MT = M <= 1 ? 1 : M <= 2 ? 2 : M <= 4 ? 4 : M <= 8 ? 8 : 16;
    const int nl = K / (npass * DL_WAVES * DL_RPI);   // row segments per lane and pass: a multiple of 4, <= 64
    const dim3 grid(cdiv(N, DL_CT));
#define ALDM_DL(MT_)                                                                                                   \
    hipLaunchKernelGGL((decode_linear_kernel<MT_>), grid, dim3(DL_WAVES * 64), 0, (hipStream_t)stream, x, ldx, M, K, w_kn, N, \
                       bias, ln_gamma, ln_beta, ln_eps, act, res, ldr, y, ldy, npass, nl)
    switch (MT) {
        case 1: ALDM_DL(1); break;
        case 2: ALDM_DL(2); break;
        case 4: ALDM_DL(4); break;
        case 8: ALDM_DL(8); break;
        default: ALDM_DL(16); break;
    }
#undef ALDM_DL
    ALDM_LAUNCH_CHECK("aldm_decode_linear");
    return 0;
}

extern "C" int aldm_decode_attention(const float* qkv, int ldq, const int64_t* pos, float* k_cache, float* v_cache,
                                     const float* keymask, int B, int heads, int n_tot, float scale, float* out, int ldo,
                                     void* stream) {
    ALDM_CHECK(qkv && pos && k_cache && v_cache && keymask && out && B > 0 && heads > 0, "aldm_decode_attention: bad args");
    ALDM_CHECK(n_tot > 0 && n_tot <= 1024, "aldm_decode_attention: %d cache positions (1..1024 supported)", n_tot);
    ALDM_CHECK(ldq >= 3 * heads * 64 && ldo >= heads * 64, "aldm_decode_attention: row pitch shorter than the row");
    // the kernel reads qkv rows and the caches as f32x4 and writes out rows the same way
    ALDM_CHECK(((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(k_cache) | reinterpret_cast<uintptr_t>(v_cache) |
                 reinterpret_cast<uintptr_t>(out)) & 15) == 0 && (ldq & 3) == 0 && (ldo & 3) == 0,
               "aldm_decode_attention: qkv / k_cache / v_cache / out must be 16-byte aligned with row pitches that are multiples of 4");
    hipLaunchKernelGGL(decode_attention_kernel, dim3(B * heads), dim3(1024), 0, (hipStream_t)stream, qkv, ldq, pos, k_cache,
                       v_cache, keymask, n_tot, heads, scale, out, ldo);
    ALDM_LAUNCH_CHECK("aldm_decode_attention");
    return 0;
}

// ---- round 6: split-K decode step (7 launches per GPT-2 block, every one on the whole chip) ----
static int decode_gemv_slices(int K, int N) {
    // K-slices: enough blocks to cover the chip (~288), slices of R = K / S rows with R % 32 == 0 and R <= DG_MAXR
    const int tiles = cdiv(N, DG_CT);
    int best = 0;
    for (int S = 1; S <= 64; ++S) {
        if (K % S) continue;
        const int R = K / S;
        if (R % 32 || R > DG_MAXR) continue;
        best = S;
        if (tiles * S >= 256) break;
    }
    return best;
}

extern "C" int aldm_decode_gemv_slices(int K, int N) { return decode_gemv_slices(K, N); }

extern "C" int aldm_decode_gemv(const float* x, int ldx, int xparts, int64_t xstride, const float* xbias, int xact, int M, int K,
                                const float* w_kn, int N, float* ypart, void* stream) {
    ALDM_CHECK(x && w_kn && ypart && M > 0 && M <= DL_MAXM && K > 0 && N > 0, "aldm_decode_gemv: bad args (1 <= M <= %d rows)", DL_MAXM);
    ALDM_CHECK(xparts >= 1 && ldx >= K, "aldm_decode_gemv: xparts >= 1, row pitch >= K");
    ALDM_CHECK(xact == ALDM_ACT_NONE || xact == ALDM_ACT_GELU || xact == ALDM_ACT_GELU_TANH || xact == ALDM_ACT_SILU || xact == ALDM_ACT_TANH,
               "aldm_decode_gemv: unsupported activation %d", xact);
    const int S = decode_gemv_slices(K, N);
    ALDM_CHECK(S > 0, "aldm_decode_gemv: K = %d has no slicing into multiples of 32 rows of at most %d", K, DG_MAXR);
    const int MT = M <= 4 ? 4 : M <= 8 ? 8 : 16;
    const dim3 grid(cdiv(N, DG_CT), S);
#define ALDM_DG(MT_)                                                                                                          \
    hipLaunchKernelGGL((decode_gemv_kernel<MT_>), grid, dim3(DG_WAVES * 64), 0, (hipStream_t)stream, x, ldx, xparts, xstride, xbias, \
                       xact, M, K, w_kn, N, ypart, K / S)
    switch (MT) {
        case 4: ALDM_DG(4); break;
        case 8: ALDM_DG(8); break;
        default: ALDM_DG(16); break;
    }
#undef ALDM_DG
    ALDM_LAUNCH_CHECK("aldm_decode_gemv");
    return 0;
}

extern "C" int aldm_decode_reduce_ln(const float* part, int nparts, int64_t pstride, int ldp, const float* bias, const int64_t* bias_row,
                                     const float* res, int ldr, int M, int N, float* h_out, int ldh, const float* ln_gamma,
                                     const float* ln_beta, float ln_eps, float* xn, int ldn, float* xn2, int ldn2, void* stream) {
    ALDM_CHECK(M > 0 && N > 0 && N <= 1024 && N % 4 == 0, "aldm_decode_reduce_ln: 1 <= N <= 1024, N %% 4 == 0");
    ALDM_CHECK(nparts >= 0 && (nparts == 0 || part) && (h_out || ln_gamma), "aldm_decode_reduce_ln: bad args");
    ALDM_CHECK(!ln_gamma == !ln_beta && (!ln_gamma || xn), "aldm_decode_reduce_ln: LayerNorm needs gamma, beta and an output");
    ALDM_CHECK(((reinterpret_cast<uintptr_t>(part) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(res) |
                 reinterpret_cast<uintptr_t>(h_out) | reinterpret_cast<uintptr_t>(xn) | reinterpret_cast<uintptr_t>(xn2) |
                 reinterpret_cast<uintptr_t>(ln_gamma) | reinterpret_cast<uintptr_t>(ln_beta)) & 15) == 0 &&
                   ((ldp | ldr | ldh | ldn | ldn2 | (int)(pstride & 3)) & 3) == 0,
               "aldm_decode_reduce_ln: operands must be 16-byte aligned with pitches that are multiples of 4");
    hipLaunchKernelGGL(decode_reduce_ln_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, part, nparts, pstride, ldp, bias, bias_row,
                       res, ldr, N, h_out, ldh, ln_gamma, ln_beta, ln_eps, xn, ldn, xn2, ldn2);
    ALDM_LAUNCH_CHECK("aldm_decode_reduce_ln");
    return 0;
}

extern "C" int aldm_decode_attention_parts(const float* qkv_part, int ldq, int qparts, int64_t qstride, const float* qbias,
                                           const int64_t* pos, float* k_cache, float* v_cache, const float* keymask, int B, int heads,
                                           int n_tot, float scale, float* out, int ldo, void* stream) {
    ALDM_CHECK(qkv_part && pos && k_cache && v_cache && keymask && out && B > 0 && heads > 0 && qparts >= 1,
               "aldm_decode_attention_parts: bad args");
    ALDM_CHECK(n_tot > 0 && n_tot <= 1024, "aldm_decode_attention_parts: %d cache positions (1..1024 supported)", n_tot);
    ALDM_CHECK(ldq >= 3 * heads * 64 && ldo >= heads * 64, "aldm_decode_attention_parts: row pitch shorter than the row");
    ALDM_CHECK(((reinterpret_cast<uintptr_t>(k_cache) | reinterpret_cast<uintptr_t>(v_cache) | reinterpret_cast<uintptr_t>(out)) & 15) == 0 &&
                   (ldo & 3) == 0,
               "aldm_decode_attention_parts: k_cache / v_cache / out must be 16-byte aligned, ldo a multiple of 4");
    hipLaunchKernelGGL(decode_attention_kernel, dim3(B * heads), dim3(1024), 0, (hipStream_t)stream, qkv_part, ldq, pos, k_cache,
                       v_cache, keymask, n_tot, heads, scale, out, ldo, qparts, qstride, qbias);
    ALDM_LAUNCH_CHECK("aldm_decode_attention_parts");
    return 0;
}
