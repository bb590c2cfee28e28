// mfma_peak.hip — microbenchmark (not product code): what the bf16 matrix pipe of THIS chip sustains on the igemm
// engine's own instruction mix (24 x v_mfma_f32_32x32x16_bf16 over 4 accumulators, the 6-partial-product order) with
// NO memory traffic, for zero operands and for operands that are split images of random fp32 data; and the shader clock
// under that load (s_memtime core cycles against the constant-rate s_memrealtime).  SURVEY.md §8(d): "confirm the
// nominal peaks on the box with a microbenchmark before dividing".
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
using bf16x8 = __bf16 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void mfma_peak_kernel(const uint4* __restrict__ opsrc, float* __restrict__ out, int iters,
                                                        unsigned long long* __restrict__ clk) {
    const int lane = threadIdx.x & 63;
    bf16x8 a[2][3], b[2][3];
    for (int i = 0; i < 2; ++i)
        for (int q = 0; q < 3; ++q) {
            a[i][q] = __builtin_bit_cast(bf16x8, opsrc[((i * 3 + q) * 64 + lane)]);
            b[i][q] = __builtin_bit_cast(bf16x8, opsrc[((6 + i * 3 + q) * 64 + lane)]);
        }
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    constexpr int PA_[6] = {0, 2, 1, 0, 1, 0}, PB_[6] = {2, 0, 1, 1, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA_[q]], b[j][PB_[q]], acc[i][j], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
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

static unsigned short bf16_trunc(float x) {
    unsigned u;
    memcpy(&u, &x, 4);
    return (unsigned short)(u >> 16);
}

extern "C" int mfma_peak_run(int blocks_per_cu, int iters, int random_data, double* tflops, double* core_mhz) {
    const int blocks = 256 * blocks_per_cu;
    std::vector<unsigned short> h(12 * 64 * 8, 0);
    if (random_data) {
        srand(1);
        // operand fragments = (hi, mid, lo) parts of random fp32 values in [-1, 1), like the engine's split images
        for (int frag = 0; frag < 4; ++frag)       // a0, a1, b0, b1
            for (int l = 0; l < 64 * 8; ++l) {
                float x = (float)rand() / RAND_MAX * 2.f - 1.f;
                float hi, mid;
                unsigned short p0 = bf16_trunc(x);
                unsigned u = (unsigned)p0 << 16; memcpy(&hi, &u, 4);
                float r1 = x - hi;
                unsigned short p1 = bf16_trunc(r1);
                u = (unsigned)p1 << 16; memcpy(&mid, &u, 4);
                unsigned short p2 = bf16_trunc(r1 - mid);
                h[((frag * 3 + 0) * 64 * 8) + l] = p0;
                h[((frag * 3 + 1) * 64 * 8) + l] = p1;
                h[((frag * 3 + 2) * 64 * 8) + l] = p2;
            }
    }
    uint4* d_ops; float* d_out; unsigned long long* d_clk;
    hipMalloc(&d_ops, h.size() * 2); hipMalloc(&d_out, blocks * 256 * 4); hipMalloc(&d_clk, 16);
    hipMemcpy(d_ops, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, 0, d_ops, d_out, iters / 10, d_clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, 0, d_ops, d_out, iters, d_clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2]; hipMemcpy(c, d_clk, 16, hipMemcpyDeviceToHost);
    *tflops = (double)blocks * 4 * iters * 24 * 32768.0 / (ms * 1e-3) * 1e-12;
    *core_mhz = c[1] ? (double)c[0] / (double)c[1] * 100.0 : 0.0;   // s_memrealtime ticks at 100 MHz
    hipFree(d_ops); hipFree(d_out); hipFree(d_clk);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
