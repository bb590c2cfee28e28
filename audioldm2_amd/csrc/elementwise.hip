// elementwise.hip — HBM-bound glue kernels of the sampling path: GEGLU gate, timestep embedding,
// layout changes at the UNet boundary, the fused CFG + DDIM update, reflect padding and STFT
// magnitude/phase.  16-byte accesses wherever the layout allows, grid-stride, fp32.
#include "common.h"

namespace aldm {

static inline int ew_blocks(int64_t n, int per_thread = 1) {
    int64_t b = cdiv64(n, 256ll * per_thread);
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}

// y[m, c] = x[m, c] * gelu(x[m, C + c]);  C % 4 == 0
__global__ void geglu_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t M, int C) {
    const int C4 = C >> 2;
    const int64_t total = M * C4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / C4;
        const int c4 = (int)(i - m * C4);
        const float* xr = x + m * 2 * C;
        const f32x4 a = *reinterpret_cast<const f32x4*>(xr + 4 * c4);
        const f32x4 g = *reinterpret_cast<const f32x4*>(xr + C + 4 * c4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = a[e] * act_apply(g[e], ALDM_ACT_GELU, 0.f);
        *reinterpret_cast<f32x4*>(y + m * C + 4 * c4) = o;
    }
}

// out[b, :half] = cos(t[b]*f_i), out[b, half:2half] = sin(t[b]*f_i), f_i = exp(-ln(P)*i/half)
__global__ void timestep_embedding_kernel(const float* __restrict__ t, float* __restrict__ out, int B,
                                          int dim, float max_period) {
    const int half = dim / 2;
    const int total = B * dim;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int b = i / dim, c = i - b * dim;
        float v = 0.f;
        if (c < 2 * half) {
            const int j = c < half ? c : c - half;
            const float freq = expf(-logf(max_period) * (float)j / (float)half);
            const float arg = t[b] * freq;
            v = c < half ? cosf(arg) : sinf(arg);
        }
        out[i] = v;
    }
}

// x [B, C, HW] -> y [rep, B, HW, C]
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C,
                                    int HW, int rep) {
    const int64_t total = (int64_t)B * C * HW;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        // i indexes the OUTPUT (coalesced stores): ((b*HW + p)*C + c)
        const int c = (int)(i % C);
        const int64_t bp = i / C;
        const int p = (int)(bp % HW);
        const int b = (int)(bp / HW);
        const float v = x[((int64_t)b * C + c) * HW + p];
        for (int r = 0; r < rep; ++r) y[r * total + i] = v;
    }
}

// x [B, HW, C] -> y [B, C, HW]
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C,
                                    int HW) {
    const int64_t total = (int64_t)B * C * HW;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % HW);
        const int64_t bc = i / HW;
        const int c = (int)(bc % C);
        const int b = (int)(bc / C);
        y[i] = x[((int64_t)b * HW + p) * C + c];
    }
}

__global__ void ddim_step_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                 const float* __restrict__ noise, const float* __restrict__ coef,
                                 float* __restrict__ x_prev, float* __restrict__ pred_x0, int64_t n) {
    const float c0 = coef[0], c1 = coef[1], c2 = coef[2], c3 = coef[3], c4 = coef[4], gs = coef[5];
    const bool cfg = coef[6] != 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        float e;
        if (cfg) {
            const float eu = eps[i], ec = eps[n + i];
            e = eu + gs * (ec - eu);
        } else {
            e = eps[i];
        }
        const float xv = x[i];
        const float p0 = (xv - c0 * e) / c1;
        const float dir = c2 * e;
        const float nz = c4 * noise[i];
        x_prev[i] = c3 * p0 + dir + nz;
        if (pred_x0) pred_x0[i] = p0;
    }
}

// The same update with the step's coefficient row and noise row selected ON THE DEVICE by a step counter, in place on x:
// one captured graph serves every step with no host-issued copies between replays.
__global__ void ddim_step_indexed_kernel(float* __restrict__ x, const float* __restrict__ eps,
                                         const float* __restrict__ noise_tab, const float* __restrict__ coef_tab,
                                         const int* __restrict__ step_idx, float* __restrict__ pred_x0, int64_t n,
                                         int coef_ld) {
    const int s = *step_idx;
    const float* coef = coef_tab + (int64_t)s * coef_ld;
    const float* noise = noise_tab + (int64_t)s * n;
    const float c0 = coef[0], c1 = coef[1], c2 = coef[2], c3 = coef[3], c4 = coef[4], gs = coef[5];
    const bool cfg = coef[6] != 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        float e;
        if (cfg) {
            const float eu = eps[i], ec = eps[n + i];
            e = eu + gs * (ec - eu);
        } else {
            e = eps[i];
        }
        const float xv = x[i];
        const float p0 = (xv - c0 * e) / c1;
        const float dir = c2 * e;
        const float nz = c4 * noise[i];
        x[i] = c3 * p0 + dir + nz;
        if (pred_x0) pred_x0[i] = p0;
    }
}

// last node of the step graph: counter += 1 and the NEXT step's timestep row into the UNet's static input (one block: every
// thread reads the old counter before thread 0 stores the new one)
__global__ void step_advance_kernel(int* __restrict__ step_idx, const float* __restrict__ t_tab, float* __restrict__ t_cur,
                                    int nt, int steps) {
    // the counter saturates at the last row: a replay past the end of the run (an extra run_step()) re-reads the last step's
    // coefficient / noise / timestep rows instead of indexing the tables out of bounds
    const int s = *step_idx + 1;
    const int row = s < steps ? s : steps - 1;
    for (int j = threadIdx.x; j < nt; j += blockDim.x) t_cur[j] = t_tab[(int64_t)row * nt + j];
    __syncthreads();
    if (threadIdx.x == 0) *step_idx = row;
}

__global__ void axpby_kernel(const float* __restrict__ a, const float* __restrict__ b,
                             float* __restrict__ y, float alpha, float beta, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        float v = alpha * a[i];
        if (b) v += beta * b[i];
        y[i] = v;
    }
}

// y[b, i] = x[b, reflect(i - pad)], i in [0, T + 2*pad)
__global__ void reflect_pad_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int T,
                                   int pad, int ld_out) {
    const int To = T + 2 * pad;
    const int64_t total = (int64_t)B * To;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / To);
        int j = (int)(i - (int64_t)b * To) - pad;
        if (j < 0) j = -j;
        if (j >= T) j = 2 * (T - 1) - j;
        y[(int64_t)b * ld_out + (i - (int64_t)b * To)] = x[(int64_t)b * T + j];
    }
}

__global__ void mag_phase_kernel(const float* __restrict__ spec, float* __restrict__ mag,
                                 float* __restrict__ phase, int64_t M, int F, int ld_spec, int ld_mag) {
    const int64_t total = M * ld_mag;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / ld_mag;
        const int f = (int)(i - m * ld_mag);
        float mg = 0.f;
        if (f < F) {
            const float re = spec[m * ld_spec + f], im = spec[m * ld_spec + F + f];
            mg = sqrtf(re * re + im * im);
            if (phase) phase[m * F + f] = atan2f(im, re);
        }
        mag[i] = mg;
    }
}

// out[m] = sqrt(sum_f x[m, f]^2): one wave64 per row (torch.norm(magnitudes, dim=1), stft.py:176)
__global__ __launch_bounds__(256) void row_l2norm_kernel(const float* __restrict__ x,
                                                         float* __restrict__ out, int64_t M, int F,
                                                         int ld) {
    const int lane = threadIdx.x & 63;
    const int64_t row = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (row >= M) return;
    float s = 0.f;
    for (int f = lane; f < F; f += 64) {
        const float v = x[row * ld + f];
        s += v * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[row] = sqrtf(s);
}

}  // namespace aldm

using namespace aldm;

extern "C" int aldm_row_l2norm(const float* x, float* out, int64_t M, int F, int ld, void* stream) {
    ALDM_CHECK(x && out && M > 0 && F > 0 && ld >= F, "aldm_row_l2norm: bad args");
    hipLaunchKernelGGL(row_l2norm_kernel, dim3((unsigned)cdiv64(M, 4)), dim3(256), 0,
                       (hipStream_t)stream, x, out, M, F, ld);
    ALDM_LAUNCH_CHECK("aldm_row_l2norm");
    return 0;
}

extern "C" int aldm_geglu(const float* x, float* y, int64_t M, int C, void* stream) {
    ALDM_CHECK(x && y && M > 0 && C > 0 && C % 4 == 0, "aldm_geglu: bad args (C=%d)", C);
    hipLaunchKernelGGL(geglu_kernel, dim3(ew_blocks(M * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                       x, y, M, C);
    ALDM_LAUNCH_CHECK("aldm_geglu");
    return 0;
}

extern "C" int aldm_timestep_embedding(const float* t, float* out, int B, int dim, float max_period,
                                       void* stream) {
    ALDM_CHECK(t && out && B > 0 && dim > 0, "aldm_timestep_embedding: bad args");
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3(ew_blocks((int64_t)B * dim)), dim3(256), 0,
                       (hipStream_t)stream, t, out, B, dim, max_period);
    ALDM_LAUNCH_CHECK("aldm_timestep_embedding");
    return 0;
}

extern "C" int aldm_nchw_to_nhwc(const float* x, float* y, int B, int C, int HW, int rep,
                                 void* stream) {
    ALDM_CHECK(x && y && B > 0 && C > 0 && HW > 0 && rep > 0, "aldm_nchw_to_nhwc: bad args");
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ew_blocks((int64_t)B * C * HW)), dim3(256), 0,
                       (hipStream_t)stream, x, y, B, C, HW, rep);
    ALDM_LAUNCH_CHECK("aldm_nchw_to_nhwc");
    return 0;
}

extern "C" int aldm_nhwc_to_nchw(const float* x, float* y, int B, int C, int HW, void* stream) {
    ALDM_CHECK(x && y && B > 0 && C > 0 && HW > 0, "aldm_nhwc_to_nchw: bad args");
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ew_blocks((int64_t)B * C * HW)), dim3(256), 0,
                       (hipStream_t)stream, x, y, B, C, HW);
    ALDM_LAUNCH_CHECK("aldm_nhwc_to_nchw");
    return 0;
}

extern "C" int aldm_ddim_step(const float* x, const float* eps, const float* noise, const float* coef,
                              float* x_prev, float* pred_x0, int64_t n, void* stream) {
    ALDM_CHECK(x && eps && noise && coef && x_prev && n > 0, "aldm_ddim_step: bad args");
    hipLaunchKernelGGL(ddim_step_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, eps,
                       noise, coef, x_prev, pred_x0, n);
    ALDM_LAUNCH_CHECK("aldm_ddim_step");
    return 0;
}

extern "C" int aldm_ddim_step_indexed(float* x, const float* eps, const float* noise_tab, const float* coef_tab,
                                      const int* step_idx, float* pred_x0, int64_t n, int coef_ld, void* stream) {
    ALDM_CHECK(x && eps && noise_tab && coef_tab && step_idx && n > 0 && coef_ld >= 7, "aldm_ddim_step_indexed: bad args");
    hipLaunchKernelGGL(ddim_step_indexed_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, eps, noise_tab,
                       coef_tab, step_idx, pred_x0, n, coef_ld);
    ALDM_LAUNCH_CHECK("aldm_ddim_step_indexed");
    return 0;
}

extern "C" int aldm_step_advance(int* step_idx, const float* t_tab, float* t_cur, int nt, int steps, void* stream) {
    ALDM_CHECK(step_idx && t_tab && t_cur && nt > 0 && steps > 0, "aldm_step_advance: bad args");
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, step_idx, t_tab, t_cur, nt, steps);
    ALDM_LAUNCH_CHECK("aldm_step_advance");
    return 0;
}

// ancestral DDPM step (ddpm.py:357-373, 1127-1181), same operation order as the reference:
//   x_recon = a*x - b*eps ; mean = c1*x_recon + c2*x ; x_prev = mean + s*noise
// coef = {a = sqrt(1/abar_t), b = sqrt(1/abar_t - 1), c1, c2 (posterior mean), s = nonzero*exp(0.5*logvar)}
__global__ void ddpm_step_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                 const float* __restrict__ noise, const float* __restrict__ coef,
                                 float* __restrict__ x_prev, int64_t n) {
    const float a = coef[0], b = coef[1], c1 = coef[2], c2 = coef[3], sg = coef[4];
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float xv = x[i];
        const float x0 = a * xv - b * eps[i];
        const float mean = c1 * x0 + c2 * xv;
        x_prev[i] = mean + sg * noise[i];
    }
}

// inpainting blend (ddim.py:226-231 with q_sample ddpm.py:430-436):
//   x = (sa*x0 + so*qnoise)*mask + (1 - mask)*x ;  coef = {sa = sqrt(abar_t), so = sqrt(1 - abar_t)}
__global__ void inpaint_blend_kernel(const float* __restrict__ x0, const float* __restrict__ qnoise,
                                     const float* __restrict__ mask, const float* __restrict__ coef,
                                     float* __restrict__ x, int64_t n) {
    const float sa = coef[0], so = coef[1];
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float m = mask[i];
        const float orig = sa * x0[i] + so * qnoise[i];
        x[i] = orig * m + (1.0f - m) * x[i];
    }
}

extern "C" int aldm_ddpm_step(const float* x, const float* eps, const float* noise, const float* coef,
                              float* x_prev, int64_t n, void* stream) {
    ALDM_CHECK(x && eps && noise && coef && x_prev && n > 0, "aldm_ddpm_step: bad args");
    hipLaunchKernelGGL(ddpm_step_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, eps,
                       noise, coef, x_prev, n);
    ALDM_LAUNCH_CHECK("aldm_ddpm_step");
    return 0;
}

extern "C" int aldm_inpaint_blend(const float* x0, const float* qnoise, const float* mask,
                                  const float* coef, float* x, int64_t n, void* stream) {
    ALDM_CHECK(x0 && qnoise && mask && coef && x && n > 0, "aldm_inpaint_blend: bad args");
    hipLaunchKernelGGL(inpaint_blend_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x0,
                       qnoise, mask, coef, x, n);
    ALDM_LAUNCH_CHECK("aldm_inpaint_blend");
    return 0;
}

extern "C" int aldm_axpby(const float* a, const float* b, float* y, float alpha, float beta, int64_t n,
                          void* stream) {
    ALDM_CHECK(a && y && n > 0, "aldm_axpby: bad args");
    hipLaunchKernelGGL(axpby_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, y,
                       alpha, beta, n);
    ALDM_LAUNCH_CHECK("aldm_axpby");
    return 0;
}

extern "C" int aldm_reflect_pad_1d(const float* x, float* y, int B, int T, int pad, int ld_out,
                                   void* stream) {
    ALDM_CHECK(x && y && B > 0 && T > 1 && pad >= 0 && pad < T && ld_out >= T + 2 * pad,
               "aldm_reflect_pad_1d: bad args");
    hipLaunchKernelGGL(reflect_pad_kernel, dim3(ew_blocks((int64_t)B * (T + 2 * pad))), dim3(256), 0,
                       (hipStream_t)stream, x, y, B, T, pad, ld_out);
    ALDM_LAUNCH_CHECK("aldm_reflect_pad_1d");
    return 0;
}

extern "C" int aldm_mag_phase(const float* spec, float* mag, float* phase, int64_t M, int F,
                              int ld_spec, int ld_mag, void* stream) {
    ALDM_CHECK(spec && mag && M > 0 && F > 0 && ld_spec >= 2 * F && ld_mag >= F,
               "aldm_mag_phase: bad args");
    hipLaunchKernelGGL(mag_phase_kernel, dim3(ew_blocks(M * ld_mag)), dim3(256), 0,
                       (hipStream_t)stream, spec, mag, phase, M, F, ld_spec, ld_mag);
    ALDM_LAUNCH_CHECK("aldm_mag_phase");
    return 0;
}
