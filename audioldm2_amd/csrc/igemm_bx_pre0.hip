// igemm_bx_pre0.hip — bf16-split (BX) instantiations of the implicit-GEMM kernel for prologue mode 0
// (see igemm_kernel.h).  The 128x32 tile and the generic prologue have no BX variant (host falls back to fp32).
#include "igemm_kernel.h"

namespace aldm {
int igemm_launch_bx_pre0(int BM, int BN, int kgroups, bool uni, bool w8, dim3 grid, hipStream_t st, const IgemmK& p) {
    constexpr int PRE = 0;
    constexpr bool HAS_UNI = PRE == PRE_AFFINE || PRE == PRE_AFFINE_SILU;
#define ALDM_IG1(BM_, BN_, WM_, WN_, KG_, U_) \
    hipLaunchKernelGGL((igemm_kernel<BM_, BN_, WM_, WN_, PRE, KG_, U_, true>), grid, dim3(64 * WM_ * WN_ * KG_), 0, st, p)
#define ALDM_IG(BM_, BN_, WM_, WN_, KG_)                            \
    do {                                                            \
        if (HAS_UNI && uni) ALDM_IG1(BM_, BN_, WM_, WN_, KG_, HAS_UNI); \
        else ALDM_IG1(BM_, BN_, WM_, WN_, KG_, false);              \
    } while (0)
    if (kgroups == 2) {
        if (BM == 64 && BN == 64) ALDM_IG(64, 64, 2, 2, 2);
        else return -1;
    } else if (BM == 128 && BN == 128 && p.d.epi_mode == ALDM_EPI_GEGLU) {
        if constexpr (PRE == PRE_NONE) ALDM_IG(128, 128, 4, 2, 1);  // 8 waves as 4x2: two 32-column tiles per wave
        else ALDM_IG(128, 128, 2, 2, 1);                            // (value + gate), what the GEGLU epilogue needs
    } else if (BM == 128 && BN == 128 && w8) ALDM_IG(128, 128, 2, 4, 1);  // one block per CU: 8 waves
    else if (BM == 128 && BN == 128) ALDM_IG(128, 128, 2, 2, 1);
    else if (BM == 128 && BN == 64) ALDM_IG(128, 64, 2, 2, 1);
    else if (BM == 64 && BN == 128) ALDM_IG(64, 128, 2, 2, 1);
    else if (BM == 64 && BN == 64) ALDM_IG(64, 64, 2, 2, 1);
    else return -1;
#undef ALDM_IG
#undef ALDM_IG1
    return 0;
}
}  // namespace aldm
