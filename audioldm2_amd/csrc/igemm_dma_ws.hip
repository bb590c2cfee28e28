// igemm_dma_ws.hip — instantiations of the persistent, wave-specialised DMA-fed GEMM (igemm_dma_ws.h).
#include "igemm_dma_ws.h"

namespace aldm {

// (tile, ring depth) instantiations: one 512-thread block per CU, so the ring is as deep as the 160 KiB of LDS allow next to
// the accumulator hand-off buffer
bool igemm_dma_ws_config_ok(int BM, int BN, int nst, int parts) {
    if (parts == 2) {
        if (BM == 64 && BN == 128) return nst >= 2 && nst <= 5;   // 2: two blocks per CU
        if (BM == 128 && BN == 64) return nst >= 2 && nst <= 4;   // 2: two blocks per CU
        if (BM == 64 && BN == 64) return nst == 3 || nst == 4 || nst == 6;   // 3, 4: two blocks per CU
        return false;
    }
    if (BM == 64 && BN == 128) return nst == 2 || nst == 3;
    if (BM == 128 && BN == 64) return nst == 2 || nst == 3;
    if (BM == 64 && BN == 64) return nst == 2 || nst == 3 || nst == 4;   // 2, 3: two blocks per CU
    return false;
}

int igemm_dma_ws_blocks_per_cu(int BM, int BN, int nst, int parts) {
    return 2 * (nst * dma_stage_slots(BM, BN, parts) * 16 + ws_acc_floats(BM, BN) * 4) <= 160 * 1024 ? 2 : 1;
}

int igemm_launch_dma_ws(int BM, int BN, int nst, int parts, int blocks, hipStream_t st, const IgemmK& p) {
#define ALDM_WS(BM_, BN_, NST_, NP_)                                                                         \
    if (BM == BM_ && BN == BN_ && nst == NST_ && parts == NP_) {                                             \
        hipLaunchKernelGGL((igemm_dma_ws_kernel<BM_, BN_, NST_, NP_>), dim3(blocks), dim3(512), 0, st, p);   \
        return 0;                                                                                            \
    }
    ALDM_WS(64, 128, 2, 2)
    ALDM_WS(64, 128, 3, 2)
    ALDM_WS(64, 128, 4, 2)
    ALDM_WS(64, 128, 5, 2)
    ALDM_WS(128, 64, 2, 2)
    ALDM_WS(128, 64, 3, 2)
    ALDM_WS(128, 64, 4, 2)
    ALDM_WS(64, 64, 3, 2)
    ALDM_WS(64, 64, 4, 2)
    ALDM_WS(64, 64, 6, 2)
    ALDM_WS(64, 128, 2, 3)
    ALDM_WS(64, 128, 3, 3)
    ALDM_WS(128, 64, 2, 3)
    ALDM_WS(128, 64, 3, 3)
    ALDM_WS(64, 64, 2, 3)
    ALDM_WS(64, 64, 3, 3)
    ALDM_WS(64, 64, 4, 3)
#undef ALDM_WS
    return -1;
}

}  // namespace aldm
