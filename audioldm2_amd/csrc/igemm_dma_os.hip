// igemm_dma_os.hip — instantiations of the operand-stationary DMA-fed GEMM (igemm_dma_os.h).
#include "igemm_dma_os.h"

namespace aldm {

// (k-tiles, ring depth) per image format: what fits 160 KB of LDS next to the 8 KB epilogue staging; the weight slab takes
// KT x NP x 4 VGPRs per wave (96 at KT = 8 / 144 at KT = 12 with 3-part images, 64 / 96 with 2-part ones) of the 256 a wave has at
// two waves per SIMD
bool igemm_dma_os_config_ok(int KT, int nst, int parts) {
    if (parts == 3) return (KT == 8 && (nst == 2 || nst == 3)) || (KT == 12 && nst == 2);
    return (KT == 8 && (nst == 2 || nst == 3 || nst == 4)) || (KT == 12 && (nst == 2 || nst == 3));
}

// the deepest ring that fits: the default when a launch is routed here without an explicit depth
int igemm_dma_os_default_stages(int KT, int parts) {
    if (parts == 3) return KT == 8 ? 3 : 2;
    return KT == 8 ? 4 : 3;
}

int igemm_launch_dma_os(int KT, int nst, int parts, bool f16, dim3 grid, hipStream_t st, const IgemmK& p) {
    // the epilogue form is a template parameter (igemm_dma_os.h, EPI): chosen here from the descriptor
    const int epi = p.d.epi_mode == ALDM_EPI_GEGLU ? OS_EPI_GEGLU : (p.d.epi_mode == ALDM_EPI_QKV ? OS_EPI_QKV : OS_EPI_PLAIN);
    if (f16) {   // "f16x3" operands (2 fp16 parts); the epilogue writes 3-part bf16 images (K / V^T, the GEGLU output)
        const bool fo = p.d.out_split_fmt == ALDM_FMT_F16;
        if (parts != 2 || (!fo && p.d.out_split_parts != 3)) return -1;
#define ALDM_OS_H(KT_, NST_)                                                                                                   \
    if (KT == KT_ && nst == NST_) {                                                                                            \
        if (fo && epi == OS_EPI_GEGLU) hipLaunchKernelGGL((igemm_dma_os_kernel<KT_, NST_, 2, OS_EPI_GEGLU, 3, true, true>), grid, dim3(512), 0, st, p); \
        else if (fo && epi == OS_EPI_QKV) hipLaunchKernelGGL((igemm_dma_os_kernel<KT_, NST_, 2, OS_EPI_QKV, 2, true, true>), grid, dim3(512), 0, st, p); \
        else if (fo) hipLaunchKernelGGL((igemm_dma_os_kernel<KT_, NST_, 2, OS_EPI_PLAIN, 3, true, true>), grid, dim3(512), 0, st, p); \
        else if (epi == OS_EPI_GEGLU) hipLaunchKernelGGL((igemm_dma_os_kernel<KT_, NST_, 2, OS_EPI_GEGLU, 3, true>), grid, dim3(512), 0, st, p); \
        else if (epi == OS_EPI_QKV) hipLaunchKernelGGL((igemm_dma_os_kernel<KT_, NST_, 2, OS_EPI_QKV, 3, true>), grid, dim3(512), 0, st, p); \
        else hipLaunchKernelGGL((igemm_dma_os_kernel<KT_, NST_, 2, OS_EPI_PLAIN, 3, true>), grid, dim3(512), 0, st, p);        \
        return 0;                                                                                                              \
    }
        ALDM_OS_H(8, 2)
        ALDM_OS_H(8, 3)
        ALDM_OS_H(8, 4)
        ALDM_OS_H(12, 2)
        ALDM_OS_H(12, 3)
#undef ALDM_OS_H
        return -1;
    }
    if (p.d.out_split_parts != parts || p.d.out_split_fmt != ALDM_FMT_BF16) return -1;   // (bf16 launches write the image format they read)
#define ALDM_OS_E(KT_, NST_, NP_, E_)                                                                            \
    hipLaunchKernelGGL((igemm_dma_os_kernel<KT_, NST_, NP_, E_>), grid, dim3(512), 0, st, p)
#define ALDM_OS(KT_, NST_, NP_)                                                                                  \
    if (KT == KT_ && nst == NST_ && parts == NP_) {                                                              \
        if (epi == OS_EPI_GEGLU) ALDM_OS_E(KT_, NST_, NP_, OS_EPI_GEGLU);                                        \
        else if (epi == OS_EPI_QKV) ALDM_OS_E(KT_, NST_, NP_, OS_EPI_QKV);                                       \
        else ALDM_OS_E(KT_, NST_, NP_, OS_EPI_PLAIN);                                                            \
        return 0;                                                                                                \
    }
    ALDM_OS(8, 2, 3)
    ALDM_OS(8, 3, 3)
    ALDM_OS(12, 2, 3)
    ALDM_OS(8, 2, 2)
    ALDM_OS(8, 3, 2)
    ALDM_OS(8, 4, 2)
    ALDM_OS(12, 2, 2)
    ALDM_OS(12, 3, 2)
#undef ALDM_OS
#undef ALDM_OS_E
    return -1;
}

}  // namespace aldm
