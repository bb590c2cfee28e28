// clap_audio.hip — the glue kernels of the CLAP audio tower (re-ranking the n_candidate_gen_per_text
// candidates, ddpm.py:1554-1568): everything of `CLAPAudioEmbeddingClassifierFreev2.forward("audio")` / `HTSAT_Swin_Transformer`
// that is not a GEMM, a LayerNorm or a softmax (those run on the shared engine).  All HBM / latency bound, fp32.
#include "common.h"

namespace aldm {

static inline int ca_blocks(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

// torchaudio.functional.resample's polyphase FIR (encoders/modules.py:700-703): y[b, n*up + i] = sum_j xpad[b, n*down + j] *
// k[i, j], xpad = x zero padded by `width` on the left; k: [up, taps] (taps = 2*width + down).
__global__ void resample_sinc_kernel(const float* __restrict__ x, const float* __restrict__ k, float* __restrict__ y, int B,
                                     int T, int Tout, int down, int up, int taps, int width) {
    const int64_t total = (int64_t)B * Tout;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / Tout);
        const int o = (int)(i - (int64_t)b * Tout);
        const int n = o / up, ph = o - n * up;
        const float* xb = x + (int64_t)b * T;
        const float* kp = k + ph * taps;
        float acc = 0.f;
        const int s0 = n * down - width;
        for (int j = 0; j < taps; ++j) {
            const int s = s0 + j;
            const float v = (s >= 0 && s < T) ? xb[s] : 0.f;
            acc = fmaf(v, kp[j], acc);
        }
        y[i] = acc;
    }
}

// |STFT|^2: spec rows [re(0..F-1) | im(0..F-1)] -> out[m, 0..ld_out) (zero padded): torchlibrosa Spectrogram(power = 2)
__global__ void power_spec_kernel(const float* __restrict__ spec, float* __restrict__ out, int64_t M, int F, int ld_spec,
                                  int ld_out) {
    const int64_t total = M * ld_out;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / ld_out;
        const int f = (int)(i - m * ld_out);
        float v = 0.f;
        if (f < F) {
            const float re = spec[m * ld_spec + f], im = spec[m * ld_spec + F + f];
            v = re * re + im * im;
        }
        out[i] = v;
    }
}

// y[r, c] = x[r, c] * scale[c] + shift[c]  (bn0 in eval mode, htsat.py:1118-1120: scale = w / sqrt(var + eps), shift = b - mean*scale)
__global__ void col_affine_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                  const float* __restrict__ shift, float* __restrict__ y, int64_t rows, int C) {
    const int64_t total = rows * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        y[i] = fmaf(x[i], scale[c], shift[c]);
    }
}

__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

// reshape_wav2img (htsat.py:1064-1090) + the im2col of PatchEmbed's 4x4 / stride-4 conv (:150, :193), in one gather:
// x [B, T, Fm] (log-mel after bn0) is stretched along time to Tt = S * ratio frames with F.interpolate(mode="bicubic",
// align_corners=True) (ATen: A = -0.75, source index scale*(dst), clamped taps), folded into the S x S image
// img[q*Fm + f][tt] = x'[q*S + tt][f], and cut into (S/p)^2 patches of p*p pixels: out[b, pi*(S/p) + pj, di*p + dj].
__global__ void bicubic_patchify_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int T, int Fm, int S,
                                        int p) {
    const int Tt = S * (S / Fm);
    const int G = S / p;
    const int64_t total = (int64_t)B * S * S;
    const float scale = Tt > 1 ? (float)(T - 1) / (float)(Tt - 1) : 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        // i = ((b*G + pi)*G + pj)*p*p + di*p + dj
        const int dj = (int)(i % p), di = (int)((i / p) % p);
        const int pj = (int)((i / (p * p)) % G), pi = (int)((i / ((int64_t)p * p * G)) % G);
        const int b = (int)(i / ((int64_t)S * S));
        const int r = pi * p + di, c = pj * p + dj;
        const int q = r / Fm, f = r - q * Fm;
        const int td = q * S + c;                      // time index in the stretched signal
        const float* xb = x + (int64_t)b * T * Fm + f;
        float v;
        if (T == Tt) {
            v = xb[(int64_t)td * Fm];
        } else {
            const float src = scale * (float)td;
            const float fl = floorf(src);
            const int i0 = (int)fl;
            const float t = src - fl;
            const float A = -0.75f;
            const float w0 = cubic2(t + 1.f, A), w1 = cubic1(t, A), w2 = cubic1(1.f - t, A), w3 = cubic2(2.f - t, A);
            auto at = [&](int k) { return xb[(int64_t)min(max(k, 0), T - 1) * Fm]; };
            v = at(i0 - 1) * w0 + at(i0) * w1 + at(i0 + 1) * w2 + at(i0 + 2) * w3;
        }
        out[i] = v;
    }
}

// y[b, c] = mean_l x[b, l, c]   (HTSAT's "embedding": avgpool over every position, htsat.py:1034-1035)
__global__ void token_mean_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int L, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C;
    const float* xb = x + (int64_t)b * L * C + c;
    float s = 0.f;
    for (int l = 0; l < L; ++l) s += xb[(int64_t)l * C];
    y[i] = s / (float)L;
}

// out[m] = a[m].b[m] / (max(|a[m]|, eps) * max(|b[m]|, eps))   (F.cosine_similarity, encoders/modules.py:651): one wave per row
__global__ __launch_bounds__(256) void row_cosine_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ out, int M, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float ab = 0.f, aa = 0.f, bb = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float u = a[(int64_t)row * C + c], v = b[(int64_t)row * C + c];
        ab = fmaf(u, v, ab);
        aa = fmaf(u, u, aa);
        bb = fmaf(v, v, bb);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ab += __shfl_xor(ab, o);
        aa += __shfl_xor(aa, o);
        bb += __shfl_xor(bb, o);
    }
    if (lane == 0) out[row] = ab / (fmaxf(sqrtf(aa), eps) * fmaxf(sqrtf(bb), eps));
}

}  // namespace aldm

using namespace aldm;

extern "C" int aldm_resample_sinc(const float* x, const float* kernel, float* y, int B, int T, int Tout, int down, int up,
                                  int taps, int width, void* stream) {
    ALDM_CHECK(x && kernel && y && B > 0 && T > 0 && Tout > 0 && down > 0 && up > 0 && taps > 0 && width >= 0,
               "aldm_resample_sinc: bad args");
    hipLaunchKernelGGL(resample_sinc_kernel, dim3(ca_blocks((int64_t)B * Tout)), dim3(256), 0, (hipStream_t)stream, x, kernel,
                       y, B, T, Tout, down, up, taps, width);
    ALDM_LAUNCH_CHECK("aldm_resample_sinc");
    return 0;
}

extern "C" int aldm_power_spec(const float* spec, float* out, int64_t M, int F, int ld_spec, int ld_out, void* stream) {
    ALDM_CHECK(spec && out && M > 0 && F > 0 && ld_spec >= 2 * F && ld_out >= F, "aldm_power_spec: bad args");
    hipLaunchKernelGGL(power_spec_kernel, dim3(ca_blocks(M * ld_out)), dim3(256), 0, (hipStream_t)stream, spec, out, M, F,
                       ld_spec, ld_out);
    ALDM_LAUNCH_CHECK("aldm_power_spec");
    return 0;
}

extern "C" int aldm_col_affine(const float* x, const float* scale, const float* shift, float* y, int64_t rows, int C,
                               void* stream) {
    ALDM_CHECK(x && scale && shift && y && rows > 0 && C > 0, "aldm_col_affine: bad args");
    hipLaunchKernelGGL(col_affine_kernel, dim3(ca_blocks(rows * C)), dim3(256), 0, (hipStream_t)stream, x, scale, shift, y,
                       rows, C);
    ALDM_LAUNCH_CHECK("aldm_col_affine");
    return 0;
}

extern "C" int aldm_bicubic_patchify(const float* x, float* out, int B, int T, int Fm, int S, int p, void* stream) {
    ALDM_CHECK(x && out && B > 0 && T > 0 && Fm > 0 && S > 0 && p > 0 && S % Fm == 0 && S % p == 0 && T <= S * (S / Fm),
               "aldm_bicubic_patchify: need S %% mel == 0, S %% patch == 0, T <= S*S/mel (T=%d mel=%d S=%d)", T, Fm, S);
    hipLaunchKernelGGL(bicubic_patchify_kernel, dim3(ca_blocks((int64_t)B * S * S)), dim3(256), 0, (hipStream_t)stream, x, out,
                       B, T, Fm, S, p);
    ALDM_LAUNCH_CHECK("aldm_bicubic_patchify");
    return 0;
}

extern "C" int aldm_token_mean(const float* x, float* y, int B, int L, int C, void* stream) {
    ALDM_CHECK(x && y && B > 0 && L > 0 && C > 0, "aldm_token_mean: bad args");
    hipLaunchKernelGGL(token_mean_kernel, dim3((B * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, y, B, L, C);
    ALDM_LAUNCH_CHECK("aldm_token_mean");
    return 0;
}

extern "C" int aldm_row_cosine(const float* a, const float* b, float* out, int M, int C, float eps, void* stream) {
    ALDM_CHECK(a && b && out && M > 0 && C > 0, "aldm_row_cosine: bad args");
    hipLaunchKernelGGL(row_cosine_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, b, out, M, C, eps);
    ALDM_LAUNCH_CHECK("aldm_row_cosine");
    return 0;
}
