// igemm_dma_lw.hip — instantiations of the DMA-fed GEMM with loader waves (igemm_dma_lw.h).
#include "igemm_dma_lw.h"
#include <stdlib.h>

namespace aldm {

static int lw_bpc(int BM, int BN, int nst, int parts) {   // blocks per CU the LDS ring allows (at most 2)
    return (160 * 1024) / dma_lds_bytes(BM, BN, nst, parts) >= 2 ? 2 : 1;
}

bool igemm_dma_lw_config_ok(int BM, int BN, int nst, int parts) {
    if (parts == 2) {
        if (BM == 128 && BN == 128) return nst == 2 || nst == 4;
        if ((BM == 64 && BN == 128) || (BM == 128 && BN == 64)) return nst == 2 || nst == 3 || nst == 4;
        if (BM == 64 && BN == 64) return nst == 2 || nst == 3 || nst == 4;
        return false;
    }
    if (BM == 128 && BN == 128) return nst == 2 || nst == 3;
    if ((BM == 64 && BN == 128) || (BM == 128 && BN == 64)) return nst == 2 || nst == 4;
    if (BM == 64 && BN == 64) return nst == 2 || nst == 3;
    return false;
}

int igemm_launch_dma_lw(int BM, int BN, int nst, int parts, bool f16, dim3 grid, hipStream_t st, const IgemmK& p) {
    // blocks per CU the registers are budgeted for: 2 (128 VGPRs per wave: the 64-row / 64-column tiles fit, bar a few
    // epilogue spills) unless the ring leaves room for one block only or the tile needs more registers (128x128).
    // $ALDM_LW_BPC=1 forces one block per CU everywhere (A/B).
    static const int env_bpc = [] {
        const char* e = getenv("ALDM_LW_BPC");
        return e ? atoi(e) : 0;
    }();
#define ALDM_LW_H(BM_, BN_, NST_, WM_)                                                                                  \
    if (f16 && BM == BM_ && BN == BN_ && nst == NST_ && parts == 2) {                                                  \
        constexpr bool two = (160 * 1024) / dma_lds_bytes(BM_, BN_, NST_, 2) >= 2 && BM_ * BN_ < 128 * 128;            \
        if (two && env_bpc != 1)                                                                                       \
            hipLaunchKernelGGL((igemm_dma_lw_kernel<BM_, BN_, NST_, WM_, 2, two ? 2 : 1, true>), grid, dim3(256 * WM_), 0, st, p); \
        else                                                                                                           \
            hipLaunchKernelGGL((igemm_dma_lw_kernel<BM_, BN_, NST_, WM_, 2, 1, true>), grid, dim3(256 * WM_), 0, st, p); \
        return 0;                                                                                                      \
    }
    ALDM_LW_H(128, 128, 2, 2)
    ALDM_LW_H(128, 128, 4, 2)
    ALDM_LW_H(64, 128, 2, 2)
    ALDM_LW_H(64, 128, 3, 2)
    ALDM_LW_H(64, 128, 4, 2)
    ALDM_LW_H(128, 64, 2, 2)
    ALDM_LW_H(128, 64, 3, 2)
    ALDM_LW_H(128, 64, 4, 2)
    ALDM_LW_H(64, 64, 2, 2)
    ALDM_LW_H(64, 64, 3, 2)
    ALDM_LW_H(64, 64, 4, 2)
#undef ALDM_LW_H
    if (f16) return -1;
#define ALDM_LW(BM_, BN_, NST_, WM_, NP_)                                                                              \
    if (BM == BM_ && BN == BN_ && nst == NST_ && parts == NP_) {                                                       \
        constexpr bool two = (160 * 1024) / dma_lds_bytes(BM_, BN_, NST_, NP_) >= 2 && BM_ * BN_ < 128 * 128;          \
        if (two && env_bpc != 1)                                                                                       \
            hipLaunchKernelGGL((igemm_dma_lw_kernel<BM_, BN_, NST_, WM_, NP_, two ? 2 : 1>), grid, dim3(256 * WM_), 0, st, p); \
        else                                                                                                           \
            hipLaunchKernelGGL((igemm_dma_lw_kernel<BM_, BN_, NST_, WM_, NP_, 1>), grid, dim3(256 * WM_), 0, st, p);   \
        return 0;                                                                                                      \
    }
    ALDM_LW(128, 128, 2, 2, 2)
    ALDM_LW(128, 128, 4, 2, 2)
    ALDM_LW(64, 128, 2, 2, 2)
    ALDM_LW(64, 128, 3, 2, 2)
    ALDM_LW(64, 128, 4, 2, 2)
    ALDM_LW(128, 64, 2, 2, 2)
    ALDM_LW(128, 64, 3, 2, 2)
    ALDM_LW(128, 64, 4, 2, 2)
    ALDM_LW(64, 64, 2, 2, 2)
    ALDM_LW(64, 64, 3, 2, 2)
    ALDM_LW(64, 64, 4, 2, 2)
    ALDM_LW(128, 128, 2, 2, 3)
    ALDM_LW(128, 128, 3, 2, 3)
    ALDM_LW(64, 128, 2, 2, 3)
    ALDM_LW(64, 128, 4, 2, 3)
    ALDM_LW(128, 64, 2, 2, 3)
    ALDM_LW(128, 64, 4, 2, 3)
    ALDM_LW(64, 64, 2, 2, 3)
    ALDM_LW(64, 64, 3, 2, 3)
#undef ALDM_LW
    (void)lw_bpc;
    return -1;
}

}  // namespace aldm
