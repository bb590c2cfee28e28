// igemm_dma.h — the DMA-fed bf16-split implicit-GEMM kernel: both operands arrive PRE-SPLIT.
//
//   out[m, n] = epi( sum_k A[m, k] * W[k, n] ),  m = (b, oh, ow), k = (kh, kw, ci)
//
// Same contraction, same "BF16x6" arithmetic (fp32 product = 6 bf16 partial products of exact 3-way operand splits,
// fp32 accumulate on v_mfma_f32_32x32x16_bf16) and same epilogue as igemm_kernel<.., BX = true>, but the K loop holds
// nothing except LDS-DMA issues, fragment reads and MFMAs:
//  * A is a split image [pixel][C/32][3 parts][32] bf16 written by its producer (aldm_split_rows = GroupNorm apply +
//    SiLU + split in ONE pass per element instead of once per tap and N-tile inside the K loop; aldm_layernorm_split;
//    the attention and GEMM epilogues).  A conv tap is still pure address generation: per row and tap one pointer,
//    zero padding = a pointer to a zero page;
//  * W is the split image aldm_pack_split_bf16 writes, [k-octet][part][Npad][8 bf16];
//  * both go global -> LDS with global_load_lds_dwordx4 (no staging registers, no ds_write, no VALU).  An LDS-DMA
//    instruction writes 64 lanes x 16 B contiguously, so the LDS image is whatever order the lanes fetch in:
//      A chunk (16 rows, one part) = [16 rows][4 k-octets] 16-byte slots, the octet XOR-swizzled by (row >> 2) & 3 on the
//        SOURCE address and on the fragment read (a ds_read_b128 lane group covers rows {r, r+12.., r+20..}: 16 distinct
//        slots mod 16 -> conflict free), each row reading 64 contiguous bytes;
//      B chunk = 64 consecutive columns of one (octet, part) row = 1 KB contiguous in HBM and in LDS;
//  * NST-deep LDS ring (3 for the 128x128 tile: 144 KB), ONE raw s_barrier per 32-wide k-tile, placed between the
//    tile's two 16-wide k-steps: before it the wave has read all its fragments of tile t and waits (counted vmcnt,
//    never 0 in steady state) for its own DMA of tile t+1; after it tile t's buffer is free, so tile t+NST is issued
//    there and the next tile's first fragments are read under the second k-step's MFMAs.  A wave is never without
//    queued MFMAs except across the barrier itself;
//  * WM x 2 waves, each a 64 x 64 (or 32-wide, for the 64-row / 64-column blocks) patch of 32x32 MFMA tiles: 12
//    ds_read_b128 per 24 MFMAs (0.5 per MFMA; the register-staged kernel's 8-wave tile: 0.75).  128x128 = 4 waves, one
//    per SIMD; 256x128 = 8 waves (two per SIMD, 2-deep ring of 72 KB stages), 25 % less LDS-DMA traffic per MFMA.
// Restrictions (host checked): C1 % 32 == 0 (every k-tile lies in one tap), no second source tensor, no prologue,
// packed + split weights, batch 1.  Everything else (stride, dilation, padding, nearest upsample, split-K, row remap,
// GEGLU, split-image output) as in igemm_kernel.h.
#pragma once
#include "igemm_epilogue.h"
#include <type_traits>

namespace aldm {

// zero padding / out-of-range columns are fetched from here (one copy per translation unit)
static __device__ __attribute__((aligned(256))) unsigned g_dma_zero_page[256 + 64];

constexpr int dma_stage_slots(int BM, int BN, int NP) { return (BM + BN) * 4 * NP; }   // 16-byte slots of one k-tile image
constexpr int dma_lds_bytes(int BM, int BN, int NST, int NP) { return NST * dma_stage_slots(BM, BN, NP) * 16; }
constexpr int dma_blocks_per_cu(int BM, int BN, int NST, int NP) {
    return (160 * 1024) / dma_lds_bytes(BM, BN, NST, NP) >= 2 ? 2 : 1;
}

// s_waitcnt vmcnt(N) lgkmcnt(0) as the BUILTIN (gfx9 encoding: vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt << 8 | vmcnt[5:4] << 14):
// hipcc's own counter model sees it, so the fragment reads issued after the barrier are not waited for together with
// the (already retired) older ones — an inline-asm wait is invisible to that model and cost an lgkmcnt(0) in front of
// the second k-step's MFMAs.
// the 32x32x16 matrix instruction of an operand format: bf16 parts, or the IEEE fp16 parts of the "f16x3" images (F16; the
// registers hold either — the split images are typed by the launch, not by the kernel's operand arrays)
template <bool F16>
__device__ __forceinline__ f32x16 mfma_32x32x16(const bf16x8 a, const bf16x8 b, const f32x16 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// F16 launches: the accumulators hold (s_a x) . (s_w w); acc_scale = 1 / (s_a s_w) (exact: powers of two) before the epilogue
template <bool F16, int MT, int NT>
__device__ __forceinline__ void unscale_acc(f32x16 (&acc)[MT][NT], float acc_scale) {
    if constexpr (F16) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] *= acc_scale;
    }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (0 << 8) | ((N >> 4) << 14));
}

// K order of a conv's k-tiles: 0 = tap outer, channel block inner (row pointers recomputed once per tap: the shipped
// order); 1 = channel block outer, tap inner (the 9 taps of a block re-read one L2-resident patch, but every k-tile pays
// the row-pointer arithmetic).  Measured on MI355X (profiles/r02_dma_taporder_ab.txt): no gain from 1 — the tap
// re-reads are served by the 256 MiB Infinity Cache at the same rate — and the 64x64 tiles lose.
#ifndef ALDM_DMA_TAP_INNER
#define ALDM_DMA_TAP_INNER 0
#endif
#ifndef ALDM_DMA_ABLATE
#define ALDM_DMA_ABLATE 0  // debug builds only (tools/gpu/build_variant.sh): 1 no A DMA, 2 no B DMA, 4 no MFMA, 8 no fragment reads,
                           // 16 no epilogue (accumulators kept alive by a never-taken store), 32 K loop cut to one k-tile,
                           // 64 the epilogue computes but never stores
#endif

// WM x 2 waves (WM = 2: 256 threads, one wave per SIMD and block; WM = 4: 512 threads, the 256-row tiles).
// NP = parts per operand: 3 = "bf16x6" (exact 3-way split, 6 partial products, fp32-grade products), 2 = "bf16x3"
// (hi + mid, both rounded to nearest: 16 significant bits per operand, 3 partial products hi*hi + hi*mid + mid*hi).
// DROP (test hook, aldm_debug_drop_product): leave out the first — smallest — partial product (hi_a x lo_w).  The result is a
// deliberately broken "5-product" GEMM, ~1e-5 off: tests/test_dma_gpu.py asserts that it FAILS the fp32-grade bar, i.e. that the
// bar would catch a kernel that silently lost a product.  Instantiated for ONE tile only (64x128, 2 stages, 3 parts).
template <int BM, int BN, int NST, int WM = 2, int NP = 3, bool DROP = false, bool F16 = false>
__global__ __launch_bounds__(128 * WM, dma_blocks_per_cu(BM, BN, NST, NP) * (WM / 2))
void igemm_dma_kernel(const IgemmK p) {
    constexpr int WN = 2, NW = WM * WN;
    constexpr int MT = BM / (32 * WM), NT = BN / 64;
    constexpr int STG = dma_stage_slots(BM, BN, NP);
    constexpr int PB = 64 * NP;               // bytes of one (row, 32-channel block) of a split image
    constexpr int RA = BM / (16 * NW);        // A row groups (16 rows x NP parts) per wave = A rows per thread
    constexpr int NB = 4 * NP * (BN / 64) / NW;   // B chunks per wave
    constexpr int D = NP * RA + NB;           // LDS-DMA instructions per thread and k-tile
    constexpr int NPROD = NP == 3 ? 6 : 3;    // bf16 partial products per fp32 product
    static_assert(NP == 2 || NP == 3, "2 or 3 parts");
    static_assert(!F16 || NP == 2, "fp16 images have two parts");
    static_assert(BM % (16 * NW) == 0 && (4 * NP * (BN / 64)) % NW == 0, "DMA chunks must divide among the waves");
    static_assert(NST >= 2 && NST <= 8 && (NST - 1) * D <= 63, "ring depth / vmcnt range");
    static_assert(NW * 32 * (NT * 32 + 4) * 4 <= NST * STG * 16, "epilogue staging must fit the ring");
    __shared__ u32x4 smem[NST * STG];   // the ONLY LDS object (a second one makes hipcc drain vmcnt before every ds_read)

    const aldm_igemm_desc& d = p.d;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;

    // XCD-aware bijective remap of the linear block id (block b runs on XCD b % 8)
    int tile_m, tile_n;
    {
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
#if ALDM_DMA_ABLATE & 32
    const int nk = min(kt1 - kt0, 1);
#else
    const int nk = kt1 - kt0;
#endif

    const char* zero = reinterpret_cast<const char*>(g_dma_zero_page);
    const char* abase = reinterpret_cast<const char*>(d.a_split);
    const char* wbase = reinterpret_cast<const char*>(d.w_split);
    const int cpb = p.Cin >> 5;                       // 32-channel blocks per tap
    const int taps = d.KH * d.KW;
    const int64_t rowbytes = (int64_t)cpb * PB;
    // K order: see ALDM_DMA_TAP_INNER above (k-tile kt = tap * cpb + cb in the shipped order, matching the weights).

    // ---- A bookkeeping: this thread fetches row (wave*RA + i)*16 + lane/4, slot lane%4 of each of the NP parts ----
    const int lane_off = ((lane & 3) ^ ((lane >> 4) & 3)) * 16;   // swizzled source octet of this lane's LDS slot
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
    int a_step[RA];         // bytes to the next channel block of the same pixel (0 for rows on the zero page)
    int t_kh, t_kw, t_cb;   // (tap, channel block) of the NEXT k-tile to issue
    auto set_tap = [&]() {  // row pointers of tap (t_kh, t_kw), channel block t_cb; zero padding -> the zero page
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
        if (ALDM_DMA_TAP_INNER) {
            t_cb = kt0 / taps;
            const int tap = kt0 - t_cb * taps;
            t_kh = tap / d.KW;
            t_kw = tap - t_kh * d.KW;
        } else {
            const int tap = kt0 / cpb;
            t_cb = kt0 - tap * cpb;
            t_kh = tap / d.KW;
            t_kw = tap - t_kh * d.KW;
        }
        set_tap();
    }
    // ---- B bookkeeping: chunk c = wave*NB + j -> (slot row = octet*NP + part, 64-column half).  The weights are
    // stored tap-major (k = tap * Cin + ci): k-tile (cb, tap) is tile index tap * cpb + cb. ----
    const char* b_ptr[NB];
    int64_t b_tile[NB];   // bytes between consecutive weight k-tiles (0 for out-of-range columns -> zero page)
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
    // issues the next k-tile of this block into ring stage `st` (branch free: it is interleaved with MFMAs) ...
    auto issue_dma = [&](int st) {
        u32x4* sa = &smem[st * STG];
#pragma unroll
        for (int i = 0; i < RA; ++i)
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const char* src = (ALDM_DMA_ABLATE & 1) ? zero + lane * 16 : a_ptr[i] + q * 64;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sa + ((wave * RA + i) * NP + q) * 64), 16, 0, 0);
            }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int c = wave * NB + j;
            const int srow = c / (BN / 64), half = c % (BN / 64);
            const char* src = (ALDM_DMA_ABLATE & 2) ? zero + lane * 16 : b_ptr[j];
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sa + BM * 4 * NP + srow * BN + half * 64), 16, 0, 0);
        }
    };
    // ... and advances the gather state to the k-tile after it
    auto advance = [&]() {
        if (taps == 1 || !ALDM_DMA_TAP_INNER) {
            // the next channel block of the same pixels (rows on the zero page keep reading it)
#pragma unroll
            for (int i = 0; i < RA; ++i) a_ptr[i] += a_step[i];
#pragma unroll
            for (int j = 0; j < NB; ++j) b_ptr[j] += b_tile[j];
            if (++t_cb == cpb && taps > 1) {   // next tile starts another tap: new row pointers
                t_cb = 0;
                if (++t_kw == d.KW) {
                    t_kw = 0;
                    ++t_kh;
                }
                set_tap();
            }
            return;
        }
        if (++t_kw == d.KW) {
            t_kw = 0;
            ++t_kh;
        }
        if (t_kh == d.KH) {   // all taps of this channel block done: first tap of the next block
            t_kh = 0;
            ++t_cb;
#pragma unroll
            for (int j = 0; j < NB; ++j) b_ptr[j] += b_tile[j] * (1 - (int64_t)(taps - 1) * cpb);
        } else {
#pragma unroll
            for (int j = 0; j < NB; ++j) b_ptr[j] += b_tile[j] * cpb;
        }
        set_tap();
    };
    auto issue = [&](int st) {
        issue_dma(st);
        advance();
    };

    // fragments of one 16-wide k-step: lane half lh owns k-octet 2*step + lh of both operands
    struct Frag {
        bf16x8 a[MT][NP], b[NT][NP];
    };
    const int a_sw = (l31 >> 2) & 3;
    auto read_frags = [&](Frag& f, int st, int step) {
#if ALDM_DMA_ABLATE & 8
        return;
#endif
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
        // smallest partial products first; NP = 2 uses the first three of {mid*hi, hi*mid, hi*hi}
        constexpr int PA_[6] = {NP == 3 ? 0 : 1, NP == 3 ? 2 : 0, NP == 3 ? 1 : 0, 0, 1, 0};
        constexpr int PB_[6] = {NP == 3 ? 2 : 0, NP == 3 ? 0 : 1, NP == 3 ? 1 : 0, 1, 0, 0};
#if ALDM_DMA_ABLATE & 4
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int q = 0; q < NP; ++q) asm volatile("" ::"v"(f.a[i][q]));
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int q = 0; q < NP; ++q) asm volatile("" ::"v"(f.b[j][q]));
        return;
#endif
#pragma unroll
        for (int q = DROP ? 1 : 0; q < NPROD; ++q)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = mfma_32x32x16<F16>(f.a[i][PA_[q]], f.b[j][PB_[q]], acc[i][j]);
    };
    // wait until at most `n` k-tiles of this thread's DMA are still in flight (n is wave uniform)
    auto wait_tiles = [&](int n) {
        constexpr int MX = NST - 1;   // at most NST - 1 tiles are ever in flight
        if (n <= 0) wait_vmcnt<0>();
        else if (n == 1) wait_vmcnt<D>();
        else if (n == 2) wait_vmcnt<(MX >= 2 ? 2 : MX) * D>();
        else if (n == 3) wait_vmcnt<(MX >= 3 ? 3 : MX) * D>();
        else if (n == 4) wait_vmcnt<(MX >= 4 ? 4 : MX) * D>();
        else if (n == 5) wait_vmcnt<(MX >= 5 ? 5 : MX) * D>();
        else wait_vmcnt<(MX >= 6 ? 6 : MX) * D>();
    };

    // ---- K loop ----------------------------------------------------------------------------------------------
#pragma unroll
    for (int s = 0; s < NST; ++s)
        if (s < nk) issue(s);
    wait_tiles(min(nk, NST) - 1);
    __builtin_amdgcn_s_barrier();
    Frag f0, f1;
#if ALDM_DMA_ABLATE & 8
    for (int q = 0; q < NP; ++q) {
        for (int i = 0; i < MT; ++i) f0.a[i][q] = f1.a[i][q] = __builtin_bit_cast(bf16x8, smem[lane + q]);
        for (int j = 0; j < NT; ++j) f0.b[j][q] = f1.b[j][q] = __builtin_bit_cast(bf16x8, smem[lane + 64 + q]);
    }
#endif
    read_frags(f0, 0, 0);
    int st = 0, t = 0;
    // one k-tile with a successor: k-step 0's MFMAs | wait + barrier | (issue tile t+NST) | next tile's first fragments
    // under k-step 1's MFMAs.  STEADY: tile t+NST exists, so NST-2 tiles stay in flight across the barrier — no
    // data-dependent branch and no control-flow join between the fragment reads and their MFMAs (at a join hipcc
    // merges its LDS counters to lgkmcnt(0), i.e. the fresh reads would be waited for before the first MFMA).
    auto body = [&](auto steady) {
        constexpr bool ST = decltype(steady)::value;
        constexpr int NMF = NPROD * MT * NT, NRD = NP * (MT + NT);
        read_frags(f1, st, 1);
        mma_frags(f0);
#pragma unroll
        for (int q = 0; q < NMF; ++q) {   // one fragment read behind each of the first MFMAs
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (q < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        const int st1 = st + 1 == NST ? 0 : st + 1;
        // the wait includes lgkmcnt(0): this wave is done reading stage `st`
        if constexpr (ST) wait_vmcnt<(NST - 2) * D>();
        else wait_tiles(min(NST - 2, nk - 2 - t));   // tiles t+2 .. nk-1 may stay in flight
        __builtin_amdgcn_s_barrier();
        if constexpr (ST) issue_dma(st);
        read_frags(f0, st1, 0);
        mma_frags(f1);
#pragma unroll
        for (int q = 0; q < NMF; ++q) {   // MFMA first, the DMA issues and the next fragments in its shadow
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
            if (ST && q == 0) __builtin_amdgcn_sched_group_barrier(0x020, D, 1);
            if (q < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ST) advance();
        st = st1;
    };
    for (; t + NST < nk; ++t) body(std::true_type{});
    for (; t + 1 < nk; ++t) body(std::false_type{});
    read_frags(f1, st, 1);   // last tile
    mma_frags(f0);
    mma_frags(f1);

    __syncthreads();   // every wave is past its last fragment read; nothing is in flight
#if ALDM_DMA_ABLATE & 16
    if (p.M == -12345) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) d.ws[(i * NT + j) * 16 + e + threadIdx.x * 256] = acc[i][j][e];
    }
    return;
#endif
    unscale_acc<F16>(acc, d.acc_scale);
    igemm_epilogue<MT, NT>(p, acc, reinterpret_cast<float*>(&smem[0]), m0, n0, wave, wm, wn, lane, 0, split);
}

}  // namespace aldm
