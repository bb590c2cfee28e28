// igemm_dma_lw.h — igemm_dma_kernel (igemm_dma.h) with LOADER WAVES: the block doubles its wave count and splits it into
//   waves 0 .. NW-1   ("MMA"):    fragment reads (ds_read_b128) and MFMAs only, then the epilogue;
//   waves NW .. 2NW-1 ("loader"): address generation and the LDS-DMA issues (global_load_lds) only; they leave after the K loop.
// Why: one global_load_lds issue holds the issuing wave for ~60-185 cycles (MI355X_MICROARCH.md, "LDS-DMA piece issue cost"),
// and a wave issues in order — in igemm_dma_kernel the 6-10 DMA issues of a k-tile sit in the same instruction stream as its
// 12-24 MFMAs, so the matrix pipe waits behind them (ablation, profiles/r02_dma_ablate.txt: MFMAs alone 85 us, DMA alone 62 us,
// together 119.5 us instead of ~85).  With the issues in their own waves the SIMD's arbiter runs the MMA wave's MFMAs while the
// loader wave is held in an issue.  Same ring, same barrier per k-tile (both roles execute the same barrier sequence), same
// fragment layout, same K order and epilogue -> results are bitwise those of igemm_dma_kernel on the same tile.
#pragma once
#include "igemm_dma.h"

namespace aldm {

template <int BM, int BN, int NST, int WM, int NP, int BPC, bool F16 = false>
__global__ __launch_bounds__(256 * WM, BPC * WM)
void igemm_dma_lw_kernel(const IgemmK p) {
    constexpr int WN = 2, NW = WM * WN;
    constexpr int MT = BM / (32 * WM), NT = BN / 64;
    constexpr int STG = dma_stage_slots(BM, BN, NP);
    constexpr int PB = 64 * NP;
    constexpr int RA = BM / (16 * NW);
    constexpr int NB = 4 * NP * (BN / 64) / NW;
    constexpr int D = NP * RA + NB;
    constexpr int NPROD = NP == 3 ? 6 : 3;
    static_assert(NP == 2 || NP == 3, "2 or 3 parts");
    static_assert(BM % (16 * NW) == 0 && (4 * NP * (BN / 64)) % NW == 0, "DMA chunks must divide among the loader waves");
    static_assert(NST >= 2 && NST <= 8 && (NST - 1) * D <= 63, "ring depth / vmcnt range");
    static_assert(NW * 32 * (NT * 32 + 4) * 4 <= NST * STG * 16, "epilogue staging must fit the ring");
    __shared__ u32x4 smem[NST * STG];

    const aldm_igemm_desc& d = p.d;
    const int lane = threadIdx.x & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool loader = wave_all >= NW;
    const int wave = loader ? wave_all - NW : wave_all;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;

    int tile_m, tile_n;
    {   // XCD-aware bijective remap of the linear block id (block b runs on XCD b % 8)
        const int nwg = gridDim.x;
        const int bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        tile_n = logical % p.tiles_n;
        tile_m = logical / p.tiles_n;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int split = blockIdx.y;
    const int nk_all = d.K >> 5;
    const int kt0 = split * p.kt_per_split;
    const int kt1 = min(nk_all, kt0 + p.kt_per_split);
    const int nk = kt1 - kt0;

    if (loader) {
        // ============================================ loader role ==================================================
        const char* zero = reinterpret_cast<const char*>(g_dma_zero_page);
        const char* abase = reinterpret_cast<const char*>(d.a_split);
        const char* wbase = reinterpret_cast<const char*>(d.w_split);
        const int cpb = p.Cin >> 5;
        const int taps = d.KH * d.KW;
        const int64_t rowbytes = (int64_t)cpb * PB;
        const int lane_off = ((lane & 3) ^ ((lane >> 4) & 3)) * 16;
        int a_pix[RA], a_h[RA], a_w[RA];
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int m = m0 + (wave * RA + i) * 16 + (lane >> 2);
            if (m < p.M) {
                const int b = m / p.OHW;
                const int rem = m - b * p.OHW;
                const int oh = rem / d.OW;
                const int ow = rem - oh * d.OW;
                a_pix[i] = b * d.H;
                a_h[i] = oh * d.SH - d.PH;
                a_w[i] = ow * d.SW - d.PW;
            } else {
                a_pix[i] = 0;
                a_h[i] = -(1 << 28);
                a_w[i] = 0;
            }
        }
        const char* a_ptr[RA];
        int a_step[RA];
        int t_kh, t_kw, t_cb;
        auto set_tap = [&]() {
            const int dh = t_kh * d.DH, dw = t_kw * d.DW;
            const int cboff = t_cb * PB + lane_off;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const int ihv = a_h[i] + dh, iwv = a_w[i] + dw;
                const bool ok = (unsigned)ihv < (unsigned)p.HV && (unsigned)iwv < (unsigned)p.WV;
                const int pix = (a_pix[i] + (ihv >> p.shh)) * d.W + (iwv >> p.shw);
                a_ptr[i] = ok ? abase + ((int64_t)pix * rowbytes + cboff) : zero + lane_off;
                a_step[i] = ok ? PB : 0;
            }
        };
        {
            const int tap = kt0 / cpb;
            t_cb = kt0 - tap * cpb;
            t_kh = tap / d.KW;
            t_kw = tap - t_kh * d.KW;
            set_tap();
        }
        const char* b_ptr[NB];
        int64_t b_tile[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int c = wave * NB + j;
            const int srow = c / (BN / 64), half = c % (BN / 64);
            const int col = n0 + half * 64 + lane;
            const bool ok = col < p.Npad;
            b_tile[j] = ok ? (int64_t)4 * NP * p.Npad * 16 : 0;
            const int tile0 = (t_kh * d.KW + t_kw) * cpb + t_cb;
            b_ptr[j] = ok ? wbase + (((int64_t)tile0 * 4 * NP + srow) * p.Npad + col) * 16 : zero;
        }
        using gptr_t = const __attribute__((address_space(1))) void*;
        using lptr_t = __attribute__((address_space(3))) void*;
        auto issue = [&](int st) {   // the next k-tile into ring stage st, then advance the gather state
            u32x4* sa = &smem[st * STG];
#pragma unroll
            for (int i = 0; i < RA; ++i)
#pragma unroll
                for (int q = 0; q < NP; ++q)
                    __builtin_amdgcn_global_load_lds((gptr_t)(a_ptr[i] + q * 64), (lptr_t)(sa + ((wave * RA + i) * NP + q) * 64),
                                                     16, 0, 0);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int c = wave * NB + j;
                const int srow = c / (BN / 64), half = c % (BN / 64);
                __builtin_amdgcn_global_load_lds((gptr_t)b_ptr[j], (lptr_t)(sa + BM * 4 * NP + srow * BN + half * 64), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < RA; ++i) a_ptr[i] += a_step[i];
#pragma unroll
            for (int j = 0; j < NB; ++j) b_ptr[j] += b_tile[j];
            if (++t_cb == cpb && taps > 1) {
                t_cb = 0;
                if (++t_kw == d.KW) {
                    t_kw = 0;
                    ++t_kh;
                }
                set_tap();
            }
        };
        auto wait_tiles = [&](int n) {
            constexpr int MX = NST - 1;
            if (n <= 0) wait_vmcnt<0>();
            else if (n == 1) wait_vmcnt<D>();
            else if (n == 2) wait_vmcnt<(MX >= 2 ? 2 : MX) * D>();
            else if (n == 3) wait_vmcnt<(MX >= 3 ? 3 : MX) * D>();
            else if (n == 4) wait_vmcnt<(MX >= 4 ? 4 : MX) * D>();
            else if (n == 5) wait_vmcnt<(MX >= 5 ? 5 : MX) * D>();
            else wait_vmcnt<(MX >= 6 ? 6 : MX) * D>();
        };
#pragma unroll
        for (int s = 0; s < NST; ++s)
            if (s < nk) issue(s);
        wait_tiles(min(nk, NST) - 1);
        __builtin_amdgcn_s_barrier();                       // P: k-tile 0 has landed
        int st = 0, t = 0;
        for (; t + NST < nk; ++t) {                         // steady: k-tile t + NST exists
            wait_vmcnt<(NST - 2) * D>();                    // k-tile t + 1 has landed (NST - 2 younger ones may fly)
            __builtin_amdgcn_s_barrier();                   // every MMA wave is done with stage st (k-tile t)
            issue(st);
            st = st + 1 == NST ? 0 : st + 1;
        }
        for (; t + 1 < nk; ++t) {
            wait_tiles(min(NST - 2, nk - 2 - t));
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_s_barrier();                       // the MMA waves' final barrier (before their epilogue)
        return;
    }

    // ================================================= MMA role =====================================================
    struct Frag {
        bf16x8 a[MT][NP], b[NT][NP];
    };
    const int a_sw = (l31 >> 2) & 3;
    auto read_frags = [&](Frag& f, int st, int step) {
        const u32x4* sa = &smem[st * STG];
        const int o = 2 * step + lh;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = (wm * MT + i) * 32 + l31;
#pragma unroll
            for (int q = 0; q < NP; ++q)
                f.a[i][q] = __builtin_bit_cast(bf16x8, sa[((row >> 4) * NP + q) * 64 + (row & 15) * 4 + (o ^ a_sw)]);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int q = 0; q < NP; ++q)
                f.b[j][q] = __builtin_bit_cast(bf16x8, sa[BM * 4 * NP + (o * NP + q) * BN + (wn * NT + j) * 32 + l31]);
    };
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    auto mma_frags = [&](const Frag& f) {
        constexpr int PA_[6] = {NP == 3 ? 0 : 1, NP == 3 ? 2 : 0, NP == 3 ? 1 : 0, 0, 1, 0};
        constexpr int PB_[6] = {NP == 3 ? 2 : 0, NP == 3 ? 0 : 1, NP == 3 ? 1 : 0, 1, 0, 0};
#pragma unroll
        for (int q = 0; q < NPROD; ++q)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = mfma_32x32x16<F16>(f.a[i][PA_[q]], f.b[j][PB_[q]], acc[i][j]);
    };
    constexpr int NMF = NPROD * MT * NT, NRD = NP * (MT + NT);
    __builtin_amdgcn_s_barrier();                           // P
    Frag f0, f1;
    read_frags(f0, 0, 0);
    int st = 0;
    for (int t = 0; t + 1 < nk; ++t) {
        read_frags(f1, st, 1);
        mma_frags(f0);
#pragma unroll
        for (int q = 0; q < NMF; ++q) {   // one fragment read behind each of the first MFMAs
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (q < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        const int st1 = st + 1 == NST ? 0 : st + 1;
        __builtin_amdgcn_s_waitcnt((7 << 4) | (3 << 14) | 15);   // lgkmcnt(0): this wave is done reading stage st
        __builtin_amdgcn_s_barrier();
        read_frags(f0, st1, 0);
        mma_frags(f1);
#pragma unroll
        for (int q = 0; q < NMF; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
            if (q < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        st = st1;
    }
    read_frags(f1, st, 1);   // last k-tile
    mma_frags(f0);
    mma_frags(f1);
    __builtin_amdgcn_s_waitcnt((7 << 4) | (3 << 14) | 15);   // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();   // every MMA wave is past its last fragment read: the ring becomes epilogue staging
    unscale_acc<F16>(acc, d.acc_scale);
    igemm_epilogue<MT, NT>(p, acc, reinterpret_cast<float*>(&smem[0]), m0, n0, wave, wm, wn, lane, 0, split);
}

}  // namespace aldm
