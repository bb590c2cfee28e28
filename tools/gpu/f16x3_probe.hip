// f16x3_probe.hip — microbenchmark (not product code), VERDICT r5 next #4: what would an fp32-grade product cost on THREE matrix
// instructions instead of six?  fp16 has 11 significant bits: the 2-part round-to-nearest split x = hi + lo keeps 22 of the 24
// bits of an fp32 (|x - hi - lo| <= 2^-22 |x| while lo stays out of fp16's subnormal range), and hi.hi + hi.lo + lo.hi on
// v_mfma_f32_32x32x16_f16 (the bf16 instruction's rate) drops only lo.lo (<= 2^-22 of the product).  fp16's 5 exponent bits are the
// catch: operands must be scaled by exact powers of two into [2^-14 .. 65504] by whoever writes them, and un-scaled in the epilogue.
// This tool measures, on the GPU, for C = A.B (128 x K x 128, one wave per 32 x 32 tile, fp32 MFMA accumulators, ascending k):
//   f32     v_mfma_f32_32x32x2_f32 (the reference's own arithmetic)
//   bf16x6  3-part truncation split, 6 products, smallest first (the library's default)
//   bf16x3  (hi, mid) round-to-nearest, 3 products (the opt-in fast mode)
//   f16x3   (hi, lo) fp16 round-to-nearest of the SCALED operands, 3 products, smallest first; A scaled per tensor so that its
//           bound (the caller's: max |a|, or a GroupNorm-style a-priori bound) maps to <= 2^15, B so that max |b| maps to [2^14, 2^15)
//   f16x3u  the same without scaling (what goes wrong: lo parts in the subnormal range, overflow past 65504)
// against an fp64 host reference: rms and max error relative to max |C| and to rms |C|, on (a) N(0,1) x N(0, 1/K) operands, (b) the
// test suite's stress operands (Student-t(3) weights, activations with per-channel mean / std = 10^3), (c) GroupNorm + SiLU-like
// activations under an a-priori bound 128x too wide, (d) activations 2^-12 .. 2^12 in magnitude (wide dynamic range in one tensor).
// It also checks what the matrix core does with fp16 SUBNORMAL inputs (flushed or honoured) and times 24 x MFMA loops of the f16 and
// the bf16 instruction on split images of random data (sustained TFLOP/s and core clock, as tools/gpu/mfma_peak.hip does).
// Build + run (GPU box): hipcc --offload-arch=gfx950 -O3 tools/gpu/f16x3_probe.hip -o /tmp/f16x3_probe && /tmp/f16x3_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <random>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
using bf16x8 = __bf16 __attribute__((ext_vector_type(8)));
using f16x8 = _Float16 __attribute__((ext_vector_type(8)));

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

enum { M_F32 = 0, M_BF16X6 = 1, M_BF16X3 = 2, M_F16X3 = 3 };

__device__ inline unsigned short bf16_trunc_bits(float x) { return (unsigned short)(__builtin_bit_cast(unsigned, x) >> 16); }
__device__ inline float bf16_bits_to_f(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }

// one wave = one 32 x 32 tile of C; A [M][K] row major, B [K][N] row major; sa / sb: power-of-two operand scales (f16x3)
template <int MODE>
__global__ __launch_bounds__(64) void gemm_probe(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M,
                                                 int N, int K, float sa, float sb) {
    const int lane = threadIdx.x, l31 = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    if constexpr (MODE == M_F32) {
        for (int k = 0; k < K; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(int64_t)(m0 + l31) * K + k + lh], B[(int64_t)(k + lh) * N + n0 + l31], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 16) {
            float a[8], b[8];
            for (int e = 0; e < 8; ++e) {
                a[e] = A[(int64_t)(m0 + l31) * K + k + 8 * lh + e];
                b[e] = B[(int64_t)(k + 8 * lh + e) * N + n0 + l31];
            }
            if constexpr (MODE == M_F16X3) {
                f16x8 ah, al, bh, bl;
                for (int e = 0; e < 8; ++e) {
                    const float xa = a[e] * sa, xb = b[e] * sb;
                    ah[e] = (_Float16)xa;
                    al[e] = (_Float16)(xa - (float)ah[e]);
                    bh[e] = (_Float16)xb;
                    bl[e] = (_Float16)(xb - (float)bh[e]);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
            } else if constexpr (MODE == M_BF16X3) {
                bf16x8 ah, am, bh, bm;
                for (int e = 0; e < 8; ++e) {
                    ah[e] = (__bf16)a[e];
                    am[e] = (__bf16)(a[e] - (float)ah[e]);
                    bh[e] = (__bf16)b[e];
                    bm[e] = (__bf16)(b[e] - (float)bh[e]);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
            } else {
                unsigned short pa[3][8], pb[3][8];
                for (int e = 0; e < 8; ++e) {
                    float r = a[e];
                    for (int q = 0; q < 3; ++q) {
                        pa[q][e] = bf16_trunc_bits(r);
                        r -= bf16_bits_to_f(pa[q][e]);
                    }
                    r = b[e];
                    for (int q = 0; q < 3; ++q) {
                        pb[q][e] = bf16_trunc_bits(r);
                        r -= bf16_bits_to_f(pb[q][e]);
                    }
                }
                bf16x8 fa[3], fb[3];
                for (int q = 0; q < 3; ++q) {
                    memcpy(&fa[q], pa[q], 16);
                    memcpy(&fb[q], pb[q], 16);
                }
                constexpr int PA_[6] = {0, 2, 1, 0, 1, 0}, PB_[6] = {2, 0, 1, 1, 0, 0};   // smallest partial products first
                for (int q = 0; q < 6; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA_[q]], fb[PB_[q]], acc, 0, 0, 0);
            }
        }
    }
    const float un = MODE == M_F16X3 ? 1.0f / (sa * sb) : 1.0f;   // exact: powers of two
    for (int e = 0; e < 16; ++e) C[(int64_t)(m0 + (e & 3) + 8 * (e >> 2) + 4 * lh) * N + n0 + l31] = acc[e] * un;
}

// what does the matrix core do with fp16 subnormal inputs?  a = 2^-20 (subnormal), b = 2^10: exact product 2^-10 per k
__global__ void denorm_probe(float* out) {
    const int lane = threadIdx.x;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = __builtin_bit_cast(_Float16, (unsigned short)0x0010);   // 16 * 2^-24 = 2^-20
        b[e] = (_Float16)1024.0f;
    }
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];   // 16 k x 2^-10 = 2^-6 when subnormals are honoured, 0 when flushed
    // and a subnormal RESULT of the conversion: (_Float16)(2^-20) must keep its value (v_cvt_f16_f32 with denormals on)
    const volatile float tiny = 9.5367431640625e-07f;
    if (lane == 0) out[1] = (float)(_Float16)tiny;
}

template <bool F16>
__global__ __launch_bounds__(256) void rate_probe(const uint4* __restrict__ opsrc, float* __restrict__ out, int iters,
                                                  unsigned long long* __restrict__ clk) {
    const int lane = threadIdx.x & 63;
    uint4 a[2][2], b[2][2];
    for (int i = 0; i < 2; ++i)
        for (int q = 0; q < 2; ++q) {
            a[i][q] = opsrc[(i * 2 + q) * 64 + lane];
            b[i][q] = opsrc[(4 + i * 2 + q) * 64 + lane];
        }
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    constexpr int PA_[3] = {1, 0, 0}, PB_[3] = {0, 1, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if constexpr (F16)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i][PA_[q]]),
                                                                               __builtin_bit_cast(f16x8, b[j][PB_[q]]), acc[i][j], 0, 0, 0);
                        else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i][PA_[q]]),
                                                                                __builtin_bit_cast(bf16x8, b[j][PB_[q]]), acc[i][j], 0, 0, 0);
                    }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = t1 - t0;
        clk[1] = r1 - r0;
    }
}

static float pow2_floor(float x) { return ldexpf(1.0f, (int)floorf(log2f(x))); }

struct Err {
    double rms_rel_max, max_rel_max, rms_rel_rms;
    int nonfinite;
};

static Err compare(const std::vector<float>& c, const std::vector<double>& ref) {
    double mx = 0, ss = 0, es = 0, em = 0;
    int bad = 0;
    for (size_t i = 0; i < ref.size(); ++i) {
        mx = fmax(mx, fabs(ref[i]));
        ss += ref[i] * ref[i];
        if (!isfinite(c[i])) {
            ++bad;
            continue;
        }
        const double e = (double)c[i] - ref[i];
        es += e * e;
        em = fmax(em, fabs(e));
    }
    const double n = (double)ref.size();
    return {sqrt(es / n) / mx, em / mx, sqrt(es / n) / sqrt(ss / n), bad};
}

int main() {
    const int M = 128, N = 128;
    std::mt19937_64 rng(1234);
    std::normal_distribution<double> nd(0.0, 1.0);
    auto student3 = [&]() {
        const double z = nd(rng), a = nd(rng), b = nd(rng), c = nd(rng);
        return z / sqrt((a * a + b * b + c * c) / 3.0);
    };
    float* dres;
    CK(hipMalloc(&dres, 64));
    denorm_probe<<<1, 64>>>(dres);
    float hres[2];
    CK(hipMemcpy(hres, dres, 8, hipMemcpyDeviceToHost));
    printf("# fp16 subnormal INPUTS of v_mfma_f32_32x32x16_f16: 16 x (2^-20 x 2^10) = %.9g (honoured: 0.015625, flushed: 0); "
           "(_Float16)2^-20 -> %.9g (kept: 9.5367e-07)\n",
           hres[0], hres[1]);

    printf("# C = A.B, 128 x K x 128; errors of each mode against fp64: rms / max relative to max|C|, rms relative to rms|C|\n");
    printf("%-34s %5s %-7s %12s %12s %12s %s\n", "operands", "K", "mode", "rms/max|C|", "max/max|C|", "rms/rms|C|", "non-finite");
    for (int dist = 0; dist < 4; ++dist) {
        for (int K : {256, 1152, 4608}) {
            std::vector<float> A((size_t)M * K), B((size_t)K * N);
            float a_bound = 0.f;   // what the PRODUCER would know about max |a| without looking at the data
            const char* name = "";
            if (dist == 0) {
                name = "a ~ N(0,1), w ~ N(0,1/K)";
                for (auto& v : A) v = (float)nd(rng);
                for (auto& v : B) v = (float)(nd(rng) / sqrt((double)K));
            } else if (dist == 1) {
                name = "stress: channel mean/std 1e3, t(3) w";
                std::vector<double> mu(K);
                for (auto& v : mu) v = 1000.0 * nd(rng);
                for (int m = 0; m < M; ++m)
                    for (int k = 0; k < K; ++k) A[(size_t)m * K + k] = (float)(mu[k] + nd(rng));
                for (auto& v : B) v = (float)(student3() / sqrt((double)K));
            } else if (dist == 2) {
                name = "SiLU(GroupNorm)-like, bound 128x wide";
                for (auto& v : A) {
                    const double y = 1.3 * nd(rng) + 0.2;
                    v = (float)(y / (1.0 + exp(-y)));
                }
                for (auto& v : B) v = (float)(nd(rng) / sqrt((double)K));
                a_bound = 128.0f * 1.3f + 0.2f;   // sqrt(group size) * max|gamma| + max|beta|: a hard bound of GroupNorm's output
            } else {
                name = "a magnitudes 2^-12 .. 2^12";
                std::uniform_real_distribution<double> ue(-12.0, 12.0);
                for (auto& v : A) v = (float)(nd(rng) * exp2(ue(rng)));
                for (auto& v : B) v = (float)(nd(rng) / sqrt((double)K));
            }
            float amax = 0.f, bmax = 0.f;
            for (auto v : A) amax = fmaxf(amax, fabsf(v));
            for (auto v : B) bmax = fmaxf(bmax, fabsf(v));
            if (a_bound == 0.f) a_bound = amax;
            const float sa = pow2_floor(32768.0f / a_bound), sb = pow2_floor(32768.0f / bmax) ;   // bound -> (2^14, 2^15]
            std::vector<double> ref((size_t)M * N, 0.0);
            for (int m = 0; m < M; ++m)
                for (int k = 0; k < K; ++k) {
                    const double a = A[(size_t)m * K + k];
                    for (int n = 0; n < N; ++n) ref[(size_t)m * N + n] += a * (double)B[(size_t)k * N + n];
                }
            float *dA, *dB, *dC;
            CK(hipMalloc(&dA, A.size() * 4));
            CK(hipMalloc(&dB, B.size() * 4));
            CK(hipMalloc(&dC, (size_t)M * N * 4));
            CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
            std::vector<float> C((size_t)M * N);
            const dim3 grid(N / 32, M / 32);
            for (int mode = 0; mode < 5; ++mode) {
                const char* mn[5] = {"f32", "bf16x6", "bf16x3", "f16x3", "f16x3u"};
                if (mode == 0) gemm_probe<M_F32><<<grid, 64>>>(dA, dB, dC, M, N, K, 1.f, 1.f);
                if (mode == 1) gemm_probe<M_BF16X6><<<grid, 64>>>(dA, dB, dC, M, N, K, 1.f, 1.f);
                if (mode == 2) gemm_probe<M_BF16X3><<<grid, 64>>>(dA, dB, dC, M, N, K, 1.f, 1.f);
                if (mode == 3) gemm_probe<M_F16X3><<<grid, 64>>>(dA, dB, dC, M, N, K, sa, sb);
                if (mode == 4) gemm_probe<M_F16X3><<<grid, 64>>>(dA, dB, dC, M, N, K, 1.f, 1.f);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
                const Err e = compare(C, ref);
                printf("%-34s %5d %-7s %12.3e %12.3e %12.3e %d%s\n", name, K, mn[mode], e.rms_rel_max, e.max_rel_max, e.rms_rel_rms, e.nonfinite,
                       mode == 3 ? (std::string("   (scales 2^") + std::to_string((int)log2f(sa)) + ", 2^" + std::to_string((int)log2f(sb)) + ")").c_str() : "");
            }
            CK(hipFree(dA));
            CK(hipFree(dB));
            CK(hipFree(dC));
        }
    }

    // sustained rate of the two instructions on 2-part split images of random data (3 products x 2 x 2 tiles x 2 = 24 MFMAs / iteration)
    {
        const int blocks = 256, iters = 20000;
        std::vector<unsigned short> hb(8 * 64 * 8), hf(8 * 64 * 8);
        for (int f = 0; f < 8; f += 2)
            for (int i = 0; i < 64 * 8; ++i) {
                const float x = (float)nd(rng) * 256.0f;
                const _Float16 h16 = (_Float16)x, l16 = (_Float16)(x - (float)h16);
                memcpy(&hf[(size_t)f * 512 + i], &h16, 2);
                memcpy(&hf[(size_t)(f + 1) * 512 + i], &l16, 2);
                const __bf16 hb16 = (__bf16)x, mb16 = (__bf16)(x - (float)hb16);
                memcpy(&hb[(size_t)f * 512 + i], &hb16, 2);
                memcpy(&hb[(size_t)(f + 1) * 512 + i], &mb16, 2);
            }
        uint4* dop;
        float* dout;
        unsigned long long* dclk;
        CK(hipMalloc(&dop, 8 * 64 * 16));
        CK(hipMalloc(&dout, (size_t)blocks * 256 * 4));
        CK(hipMalloc(&dclk, 16));
        for (int f16 = 0; f16 < 2; ++f16) {
            CK(hipMemcpy(dop, f16 ? hf.data() : hb.data(), 8 * 64 * 16, hipMemcpyHostToDevice));
            double best = 0, mhz = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEvent_t e0, e1;
                CK(hipEventCreate(&e0));
                CK(hipEventCreate(&e1));
                CK(hipEventRecord(e0));
                if (f16) rate_probe<true><<<blocks, 256>>>(dop, dout, iters, dclk);
                else rate_probe<false><<<blocks, 256>>>(dop, dout, iters, dclk);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                unsigned long long clk[2];
                CK(hipMemcpy(clk, dclk, 16, hipMemcpyDeviceToHost));
                const double tf = (double)blocks * 4 * iters * 24 * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
                if (tf > best) {
                    best = tf;
                    mhz = (double)clk[0] / ((double)clk[1] / 100e6) / 1e6;
                }
            }
            printf("# sustained %s: %.0f TFLOP/s (dense), core clock %.0f MHz — 2-part split images of N(0, 256^2) data, one wave per SIMD on 256 CUs\n",
                   f16 ? "v_mfma_f32_32x32x16_f16 " : "v_mfma_f32_32x32x16_bf16", best, mhz);
        }
    }
    return 0;
}
