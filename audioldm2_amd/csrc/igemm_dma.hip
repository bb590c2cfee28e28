// igemm_dma.hip — instantiations of the DMA-fed bf16-split implicit-GEMM kernel (igemm_dma.h) and the split-image
// producers that feed it: aldm_split_rows (GroupNorm apply + activation + 3-way split in one pass over the tensor).
#include "igemm_dma.h"
#include <atomic>

namespace aldm {

// (tile, ring depth) instantiations; the 2-part ("bf16x3") images are 2/3 the size, so their rings can be deeper
bool igemm_dma_config_ok(int BM, int BN, int nst, int parts) {
    if (BM == 256 && BN == 128) return parts == 3 ? nst == 2 : (nst == 2 || nst == 3);
    if (BM == 128 && BN == 128) return parts == 3 ? (nst == 2 || nst == 3) : (nst == 2 || nst == 4);
    // (6-deep rings, 2-part images only: the small-M launches run one block per CU and are bound by the LDS-DMA latency)
    if ((BM == 64 && BN == 128) || (BM == 128 && BN == 64)) return nst == 2 || nst == 4 || (nst == 6 && parts == 2);
    if (BM == 64 && BN == 64) return nst == 2 || nst == 3 || (nst == 6 && parts == 2);
    return false;
}

#ifdef ALDM_TEST_HOOKS
// test hook (aldm_debug_drop_product, include/aldm_hip.h; libaldm_hip_testhooks.so only): non-zero -> DMA-fed launches run the
// 5-product kernel or fail.  The release library has neither the switch nor the broken instantiation.
std::atomic<int> g_debug_drop_product{0};
#endif

int igemm_launch_dma(int BM, int BN, int nst, int parts, bool f16, dim3 grid, hipStream_t st, const IgemmK& p) {
#ifdef ALDM_TEST_HOOKS
    if (g_debug_drop_product.load(std::memory_order_relaxed)) {   // (aldm_igemm has checked the tile)
        if (!(BM == 64 && BN == 128 && nst == 2 && parts == 3)) return -1;
        hipLaunchKernelGGL((igemm_dma_kernel<64, 128, 2, 2, 3, true>), grid, dim3(256), 0, st, p);
        return 0;
    }
#endif
#define ALDM_DMA(BM_, BN_, NST_, NP_) \
    hipLaunchKernelGGL((igemm_dma_kernel<BM_, BN_, NST_, 2, NP_>), grid, dim3(256), 0, st, p)
#define ALDM_DMA8(BM_, BN_, NST_, NP_) \
    hipLaunchKernelGGL((igemm_dma_kernel<BM_, BN_, NST_, 4, NP_>), grid, dim3(512), 0, st, p)
    if (f16) {   // "f16x3" images: the 2-part instantiations on the fp16 matrix instruction
        if (parts != 2) return -1;
#define ALDM_DMA_H(BM_, BN_, NST_) \
    hipLaunchKernelGGL((igemm_dma_kernel<BM_, BN_, NST_, 2, 2, false, true>), grid, dim3(256), 0, st, p)
#define ALDM_DMA8_H(BM_, BN_, NST_) \
    hipLaunchKernelGGL((igemm_dma_kernel<BM_, BN_, NST_, 4, 2, false, true>), grid, dim3(512), 0, st, p)
        if (BM == 256 && BN == 128 && nst == 3) ALDM_DMA8_H(256, 128, 3);
        else if (BM == 256 && BN == 128 && nst == 2) ALDM_DMA8_H(256, 128, 2);
        else if (BM == 128 && BN == 128 && nst == 4) ALDM_DMA_H(128, 128, 4);
        else if (BM == 128 && BN == 128 && nst == 2) ALDM_DMA_H(128, 128, 2);
        else if (BM == 64 && BN == 128 && nst == 6) ALDM_DMA_H(64, 128, 6);
        else if (BM == 64 && BN == 128 && nst == 4) ALDM_DMA_H(64, 128, 4);
        else if (BM == 64 && BN == 128 && nst == 2) ALDM_DMA_H(64, 128, 2);
        else if (BM == 128 && BN == 64 && nst == 6) ALDM_DMA_H(128, 64, 6);
        else if (BM == 128 && BN == 64 && nst == 4) ALDM_DMA_H(128, 64, 4);
        else if (BM == 128 && BN == 64 && nst == 2) ALDM_DMA_H(128, 64, 2);
        else if (BM == 64 && BN == 64 && nst == 6) ALDM_DMA_H(64, 64, 6);
        else if (BM == 64 && BN == 64 && nst == 3) ALDM_DMA_H(64, 64, 3);
        else if (BM == 64 && BN == 64 && nst == 2) ALDM_DMA_H(64, 64, 2);
        else return -1;
#undef ALDM_DMA_H
#undef ALDM_DMA8_H
        return 0;
    }
    if (parts == 3) {
        if (BM == 256 && BN == 128 && nst == 2) ALDM_DMA8(256, 128, 2, 3);
        else if (BM == 128 && BN == 128 && nst == 3) ALDM_DMA(128, 128, 3, 3);
        else if (BM == 128 && BN == 128 && nst == 2) ALDM_DMA(128, 128, 2, 3);
        else if (BM == 64 && BN == 128 && nst == 4) ALDM_DMA(64, 128, 4, 3);
        else if (BM == 64 && BN == 128 && nst == 2) ALDM_DMA(64, 128, 2, 3);
        else if (BM == 128 && BN == 64 && nst == 4) ALDM_DMA(128, 64, 4, 3);
        else if (BM == 128 && BN == 64 && nst == 2) ALDM_DMA(128, 64, 2, 3);
        else if (BM == 64 && BN == 64 && nst == 3) ALDM_DMA(64, 64, 3, 3);
        else if (BM == 64 && BN == 64 && nst == 2) ALDM_DMA(64, 64, 2, 3);
        else return -1;
    } else {
        if (BM == 256 && BN == 128 && nst == 3) ALDM_DMA8(256, 128, 3, 2);
        else if (BM == 256 && BN == 128 && nst == 2) ALDM_DMA8(256, 128, 2, 2);
        else if (BM == 128 && BN == 128 && nst == 4) ALDM_DMA(128, 128, 4, 2);
        else if (BM == 128 && BN == 128 && nst == 2) ALDM_DMA(128, 128, 2, 2);
        else if (BM == 64 && BN == 128 && nst == 6) ALDM_DMA(64, 128, 6, 2);
        else if (BM == 64 && BN == 128 && nst == 4) ALDM_DMA(64, 128, 4, 2);
        else if (BM == 64 && BN == 128 && nst == 2) ALDM_DMA(64, 128, 2, 2);
        else if (BM == 128 && BN == 64 && nst == 6) ALDM_DMA(128, 64, 6, 2);
        else if (BM == 128 && BN == 64 && nst == 4) ALDM_DMA(128, 64, 4, 2);
        else if (BM == 128 && BN == 64 && nst == 2) ALDM_DMA(128, 64, 2, 2);
        else if (BM == 64 && BN == 64 && nst == 6) ALDM_DMA(64, 64, 6, 2);
        else if (BM == 64 && BN == 64 && nst == 3) ALDM_DMA(64, 64, 3, 2);
        else if (BM == 64 && BN == 64 && nst == 2) ALDM_DMA(64, 64, 2, 2);
        else return -1;
    }
#undef ALDM_DMA
#undef ALDM_DMA8
    return 0;
}

// ---- split-image producer -------------------------------------------------------------------------------------
// dst[row][c] = split(act(x[row][c] * scale[b, c] + shift[b, c])), x = x1 ++ x2 along C (channels-last rows, P rows
// per sample); scale == NULL: no affine.  dst_raw (optional) = split(x) of the same concatenated rows (the operand of
// a ResBlock's 1x1 skip conv).  One thread = 8 consecutive channels of one row: two 16-byte loads, three (six)
// 16-byte stores; 4 neighbouring lanes cover one 192-byte block.
template <int ACT, bool AFF>
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                         int C1, int C2, int64_t rows, int P,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift, char* __restrict__ dst,
                                                         char* __restrict__ dst_raw, int parts, float slope, int raw_parts,
                                                         float f16_scale) {
    const int C = C1 + C2;
    const int C8 = C >> 3;
    const int64_t total = rows * C8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / C8;
        const int c = (int)(i - row * C8) << 3;
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
                if (q < raw_parts) *reinterpret_cast<u32x4*>(dst_raw + roff + q * 64) = u32x4{p0[q][0], p0[q][1], p1[q][0], p1[q][1]};
        }
        if constexpr (AFF) {
            const int64_t so = (row / P) * C + c;
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(scale + so), s1 = *reinterpret_cast<const f32x4*>(scale + so + 4);
            const f32x4 h0 = *reinterpret_cast<const f32x4*>(shift + so), h1 = *reinterpret_cast<const f32x4*>(shift + so + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {   // one rounding per element: matters when |mean| >> std (shift ~ -mean * scale)
                v0[e] = __builtin_fmaf(v0[e], s0[e], h0[e]);
                v1[e] = __builtin_fmaf(v1[e], s1[e], h1[e]);
            }
        }
        if constexpr (ACT == ALDM_ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v0[e] = silu_fast(v0[e]);
                v1[e] = silu_fast(v1[e]);
            }
        }
        if constexpr (ACT == ALDM_ACT_LRELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v0[e] = v0[e] > 0.0f ? v0[e] : v0[e] * slope;
                v1[e] = v1[e] > 0.0f ? v1[e] : v1[e] * slope;
            }
        }
        split4_fmt(v0, p0, parts, f16_scale);
        split4_fmt(v1, p1, parts, f16_scale);
#pragma unroll
        for (int q = 0; q < 3; ++q)
            if (q < parts) *reinterpret_cast<u32x4*>(dst + off + q * 64) = u32x4{p0[q][0], p0[q][1], p1[q][0], p1[q][1]};
    }
}

}  // namespace aldm

using namespace aldm;

#ifdef ALDM_TEST_HOOKS
extern "C" int aldm_debug_drop_product(int on) { return aldm::g_debug_drop_product.exchange(on ? 1 : 0); }
#endif

extern "C" int64_t aldm_split_image_bytes(int64_t rows, int C, int parts) { return rows * (int64_t)(C / 32) * 64 * parts; }

extern "C" int aldm_split_rows_act(const float* x1, const float* x2, int C1, int C2, int64_t rows, int P, const float* scale,
                                   const float* shift, int act, float slope, void* dst, void* dst_raw, int parts, void* stream);
extern "C" int aldm_split_rows_f16(const float* x1, const float* x2, int C1, int C2, int64_t rows, int P, const float* scale,
                                   const float* shift, int act, float slope, void* dst, void* dst_raw, int raw_parts, float f16_scale,
                                   void* stream);

extern "C" int aldm_split_rows(const float* x1, const float* x2, int C1, int C2, int64_t rows, int P, const float* scale,
                               const float* shift, int act, void* dst, void* dst_raw, int parts, void* stream) {
    ALDM_CHECK(act == ALDM_ACT_NONE || act == ALDM_ACT_SILU, "aldm_split_rows: activation %d not supported", act);
    return aldm_split_rows_act(x1, x2, C1, C2, rows, P, scale, shift, act, 0.f, dst, dst_raw, parts, stream);
}

static int split_rows_launch(const float* x1, const float* x2, int C1, int C2, int64_t rows, int P, const float* scale,
                             const float* shift, int act, float slope, void* dst, void* dst_raw, int parts, int raw_parts,
                             float f16_scale, void* stream);

extern "C" int aldm_split_rows_act(const float* x1, const float* x2, int C1, int C2, int64_t rows, int P, const float* scale,
                                   const float* shift, int act, float slope, void* dst, void* dst_raw, int parts, void* stream) {
    return split_rows_launch(x1, x2, C1, C2, rows, P, scale, shift, act, slope, dst, dst_raw, parts, parts, 0.f, stream);
}

extern "C" int aldm_split_rows_f16(const float* x1, const float* x2, int C1, int C2, int64_t rows, int P, const float* scale,
                                   const float* shift, int act, float slope, void* dst, void* dst_raw, int raw_parts, float f16_scale,
                                   void* stream) {
    ALDM_CHECK(f16_scale > 0.0f && (raw_parts == 2 || raw_parts == 3), "aldm_split_rows_f16: need f16_scale > 0 and raw_parts 2 | 3");
    return split_rows_launch(x1, x2, C1, C2, rows, P, scale, shift, act, slope, dst, dst_raw, 2, raw_parts, f16_scale, stream);
}

static int split_rows_launch(const float* x1, const float* x2, int C1, int C2, int64_t rows, int P, const float* scale,
                             const float* shift, int act, float slope, void* dst, void* dst_raw, int parts, int raw_parts,
                             float f16_scale, void* stream) {
    if (!x2) C2 = 0;
    const int C = C1 + C2;
    ALDM_CHECK(x1 && dst && rows > 0 && P > 0 && (parts == 2 || parts == 3), "aldm_split_rows: bad args");
    ALDM_CHECK(C % 32 == 0 && C1 % 8 == 0 && C2 % 8 == 0, "aldm_split_rows: need (C1+C2) %% 32 == 0, C1 %% 8 == 0 (C1=%d C2=%d)",
               C1, C2);
    ALDM_CHECK((scale == nullptr) == (shift == nullptr), "aldm_split_rows: scale/shift must come together");
    ALDM_CHECK(act == ALDM_ACT_NONE || act == ALDM_ACT_SILU || act == ALDM_ACT_LRELU, "aldm_split_rows: activation %d not supported",
               act);
    ALDM_CHECK(((reinterpret_cast<uintptr_t>(x1) | reinterpret_cast<uintptr_t>(x2) | reinterpret_cast<uintptr_t>(dst) |
                 reinterpret_cast<uintptr_t>(dst_raw) | reinterpret_cast<uintptr_t>(scale) |
                 reinterpret_cast<uintptr_t>(shift)) & 15) == 0,
               "aldm_split_rows: operands must be 16-byte aligned");
    const int64_t total = rows * (C >> 3);
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 256 * 16);
    hipStream_t st = (hipStream_t)stream;
#define ALDM_SPLIT(A_, F_)                                                                                        \
    hipLaunchKernelGGL((split_rows_kernel<A_, F_>), dim3(blocks), dim3(256), 0, st, x1, x2, C1, C2, rows, P, scale, \
                       shift, reinterpret_cast<char*>(dst), reinterpret_cast<char*>(dst_raw), parts, slope, raw_parts, f16_scale)
    if (scale) {
        if (act == ALDM_ACT_SILU) ALDM_SPLIT(ALDM_ACT_SILU, true);
        else if (act == ALDM_ACT_LRELU) ALDM_SPLIT(ALDM_ACT_LRELU, true);
        else ALDM_SPLIT(ALDM_ACT_NONE, true);
    } else {
        if (act == ALDM_ACT_SILU) ALDM_SPLIT(ALDM_ACT_SILU, false);
        else if (act == ALDM_ACT_LRELU) ALDM_SPLIT(ALDM_ACT_LRELU, false);
        else ALDM_SPLIT(ALDM_ACT_NONE, false);
    }
#undef ALDM_SPLIT
    ALDM_LAUNCH_CHECK("aldm_split_rows");
    return 0;
}
