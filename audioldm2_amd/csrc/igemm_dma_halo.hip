// igemm_dma_halo.hip — instantiations of the halo-patch 3x3 convolution kernel (igemm_dma_halo.h).
#include "igemm_dma_halo.h"

namespace aldm {

// (tile, weight-ring depth, waves, parts, patch chunks) instantiations.  MAXCH = 16-pixel chunks per part the patch buffers hold:
// BM / 16 + 2 covers W <= 16 (the UNet's levels), 16 the 128-row tile at W = 32 / 64 (the VAE decoder's upper levels).
#define ALDM_HALO_LIST(X)          \
    X(256, 128, 2, 4, 3, 18)       \
    X(128, 128, 3, 2, 3, 10)       \
    X(128, 128, 4, 2, 3, 10)       \
    X(128, 128, 2, 2, 3, 10)       \
    X(128, 128, 2, 4, 3, 10)       \
    X(128, 128, 3, 4, 3, 10)       \
    X(128, 128, 2, 2, 3, 16)       \
    X(128, 128, 2, 4, 3, 16)       \
    X(256, 128, 2, 4, 2, 18)       \
    X(256, 128, 3, 4, 2, 18)       \
    X(128, 128, 3, 2, 2, 10)       \
    X(128, 128, 4, 2, 2, 10)       \
    X(128, 128, 3, 4, 2, 10)       \
    X(128, 128, 3, 2, 2, 16)       \
    X(128, 128, 3, 4, 2, 16)

// the instantiation that runs (BM x BN, ring depth nstb, WM x 2 waves, parts) on a patch of `nch` chunks per part: its MAXCH, or 0
int igemm_dma_halo_maxch(int BM, int BN, int nstb, int wm, int parts, int nch) {
    int best = 0;
#define X(BM_, BN_, NST_, WM_, NP_, CH_)                                                                   \
    if (BM == BM_ && BN == BN_ && nstb == NST_ && wm == WM_ && parts == NP_ && nch <= CH_ &&             \
        nch * NP_ <= (11 - NST_) * WM_ * 2 && (best == 0 || CH_ < best))                                   \
        best = CH_;
    ALDM_HALO_LIST(X)
#undef X
    return best;
}

int igemm_launch_dma_halo(int BM, int BN, int nstb, int wm, int parts, bool f16, int maxch, dim3 grid, hipStream_t st, const IgemmK& p) {
    if (f16) {   // "f16x3" images: the 2-part instantiations on the fp16 matrix instruction
#define X(BM_, BN_, NST_, WM_, NP_, CH_)                                                                                   \
    if constexpr (NP_ == 2) {                                                                                              \
        if (BM == BM_ && BN == BN_ && nstb == NST_ && wm == WM_ && parts == 2 && maxch == CH_) {                           \
            hipLaunchKernelGGL((igemm_dma_halo_kernel<BM_, BN_, NST_, WM_, 2, CH_, true>), grid, dim3(128 * WM_), 0, st, p); \
            return 0;                                                                                                      \
        }                                                                                                                  \
    }
        ALDM_HALO_LIST(X)
#undef X
        return -1;
    }
#define X(BM_, BN_, NST_, WM_, NP_, CH_)                                                                             \
    if (BM == BM_ && BN == BN_ && nstb == NST_ && wm == WM_ && parts == NP_ && maxch == CH_) {                      \
        hipLaunchKernelGGL((igemm_dma_halo_kernel<BM_, BN_, NST_, WM_, NP_, CH_>), grid, dim3(128 * WM_), 0, st, p); \
        return 0;                                                                                                    \
    }
    ALDM_HALO_LIST(X)
#undef X
    return -1;
}

}  // namespace aldm
