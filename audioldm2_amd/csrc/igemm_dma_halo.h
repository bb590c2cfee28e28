// igemm_dma_halo.h — the DMA-fed bf16-split implicit-GEMM kernel for 3x3 / stride-1 / pad-1 convolutions with the A operand
// staged ONCE per 32-channel block: a halo patch in LDS that all nine taps read.
//
//   out[m, n] = epi( sum_{cb} sum_{kh, kw} sum_{c in block cb} A[pix(m) + (kh - 1, kw - 1), c] * W[(kh, kw, c), n] )
//
// igemm_dma_kernel (igemm_dma.h) treats a conv tap as address generation: every one of the 9 taps of a channel block streams its
// own BM x 32 A tile L2 -> LDS, i.e. each A element crosses the L2 -> LDS path nine times (PMC: 4.0x the algorithmic bytes on the
// 256x128 tile, profiles/r05_pmc_traffic_bf16x6.json; with that feed ablated the launch ran 26 % faster, r02_dma_ablate.txt).
// Here a block's BM output rows are TR = BM / W whole image rows of ONE image (host checked), so the A data of ALL nine taps of a
// channel block is the (TR + 2) x W pixel patch around them:
//  * the patch of channel block cb + 1 streams into the second of two LDS patch buffers while the nine taps of block cb run — one
//    1 KB LDS-DMA piece (16 pixels x one bf16 part, the A chunk layout of igemm_dma.h: [16 pixels][4 k-octets] 16-byte slots,
//    octet XOR-swizzled by (pixel >> 2) & 3) per wave and k-tile instead of NP * BM / (16 * waves).  Rows above / below the
//    image come from the zero page;
//  * a tap is a SHIFT of the fragment-read address: output pixel pt (tile-local, row-major) reads patch pixel
//    pp = pt + kh * W + kw - 1; the left / right zero padding (kw = 0 at column 0, kw = 2 at column W - 1) reads a zero slot — the one
//    of the SAME 16-byte bank group as the slot it replaces (one shared zero slot cost 10-15 % LDS bank-conflict cycles: it broke the
//    permutation below).
//    The ds_read_b128 lane groups {0-3, 12-15, 20-27} + shift are a permutation of 0 .. 15 mod 16 for EVERY shift, so the
//    swizzled layout stays conflict free for all nine taps and any W;
//  * the weights stream exactly as in igemm_dma_kernel (NSTB-deep ring of [4 octets x NP parts][BN] k-tiles, 1 KB pieces), in
//    the order (channel block outer, tap inner): k-tile (cb, tap) is weight tile tap * cpb + cb;
//  * same MFMA tiling (WM x 2 waves, MT x NT 32x32 tiles per wave, v_mfma_f32_32x32x16_bf16, NP parts -> 6 / 3 partial
//    products, smallest first), same one-barrier-per-k-tile software pipeline, same epilogue (igemm_epilogue.h).
// L2 -> LDS bytes per k-tile of the 256x128 / 3-part tile: 24 KB of weights + 6 KB of patch instead of 24 + 48.
// The K order differs from igemm_dma_kernel's default (tap outer): results agree to fp32 summation order, not bitwise.
//
// LDS-DMA bookkeeping.  Every wave issues D = NB + 1 pieces per k-tile: its NB weight pieces of tile t + NSTB and ONE patch
// piece (or a dummy piece from the zero page into the zero region, so that the counted s_waitcnt stays uniform).  After the
// barrier of tile t' the patch piece slot v = t' + 1 is issued: patch c = v / 9 + 1, slot s = v % 9, real for s < ATILES =
// 11 - NSTB (the counted wait at the barrier of tile t' + 1 covers everything issued up to tile t' + 2 - NSTB, and patch c must
// be complete at the barrier of tile 9c - 1; its buffer is free after the barrier of tile 9c - 10).  Piece id = s * waves + wave
// -> (chunk, part) = (id / NP, id % NP); host checks chunks * NP <= ATILES * waves.
#pragma once
#include "igemm_dma.h"

#ifndef ALDM_HALO_ABLATE
#define ALDM_HALO_ABLATE 0   // timing-only builds (tools/gpu/build_variant.sh): 1 every patch piece is a dummy (no A traffic after the
                             // prologue), 2 weight pieces read the zero page, 32 fragment addresses are not recomputed per tap, 64 every weight k-tile is tile 0
                             // (REAL operand bits from a cache-resident address: 1 and 2 feed ZEROS to the matrix pipe, which lowers its power
                             // draw and raises the clock — they overstate what the feed costs), 128 every patch is channel block cb0's;
                             // ALDM_DMA_ABLATE's 4 (no MFMA), 8 (no fragment reads), 16 (no epilogue) apply as in igemm_dma.h
#endif

namespace aldm {

constexpr int halo_lds_slots(int BN, int NSTB, int NP, int MAXCH) {
    return 2 * MAXCH * NP * 64 + NSTB * BN * 4 * NP + NP * 64;   // two patches, the weight ring, the zero region
}

template <int BM, int BN, int NSTB, int WM, int NP, int MAXCH, bool F16 = false>
__global__ __launch_bounds__(128 * WM, WM / 2)
void igemm_dma_halo_kernel(const IgemmK p) {
    constexpr int WN = 2, NW = WM * WN;
    constexpr int MT = BM / (32 * WM), NT = BN / 64;
    constexpr int PB = 64 * NP;                    // bytes of one (pixel, 32-channel block) of a split image
    constexpr int PATCH = MAXCH * NP * 64;         // 16-byte slots of one patch buffer
    constexpr int BSTG = BN * 4 * NP;              // ... of one weight k-tile
    constexpr int B0 = 2 * PATCH;                  // first slot of the weight ring
    constexpr int Z0 = B0 + NSTB * BSTG;           // first slot of the zero region (NP KB: the part offset q * 1 KB of a read stays inside)
    constexpr int NB = 4 * NP * (BN / 64) / NW;    // weight pieces per wave and k-tile
    constexpr int D = NB + 1;                      // LDS-DMA instructions per wave and k-tile
    constexpr int ATILES = 11 - NSTB;              // patch-piece slots per channel block
    constexpr int NPROD = NP == 3 ? 6 : 3;
    static_assert(NP == 2 || NP == 3, "2 or 3 parts");
    static_assert((4 * NP * (BN / 64)) % NW == 0, "weight pieces must divide among the waves");
    static_assert(NSTB >= 2 && NSTB <= 6 && (NSTB - 1) * D <= 63, "ring depth / vmcnt range");
    static_assert(NW * 32 * (NT * 32 + 4) * 4 <= Z0 * 16, "epilogue staging must fit");
    static_assert((Z0 + NP * 64) * 16 <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(1024))) u32x4 smem[Z0 + NP * 64];   // the ONLY LDS object (see igemm_dma.h)

    const aldm_igemm_desc& d = p.d;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;

    int tile_m, tile_n;
    {   // XCD-aware bijective remap of the linear block id (block b runs on XCD b % 8): an XCD walks neighbouring image rows
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
    const int cpb = p.Cin >> 5;                        // 32-channel blocks
    const int cbps = p.kt_per_split / 9;               // ... per split (host: kt_per_split % 9 == 0)
    const int cb0 = split * cbps;
    const int ncb = min(cpb, cb0 + cbps) - cb0;
    const int nk = ncb * 9;

    const int W = d.W, lgW = __builtin_ctz((unsigned)W);
    const int img = m0 / p.OHW;                        // the tile lies inside one image (host: OHW % BM == 0, BM % W == 0)
    const int r0 = (m0 - img * p.OHW) >> lgW;          // its first image row
    const int PP = ((BM >> lgW) + 2) << lgW;           // patch pixels: image rows r0 - 1 .. r0 + TR
    const int NCH = (PP + 15) >> 4;                    // 16-pixel chunks per part

    const char* zero = reinterpret_cast<const char*>(g_dma_zero_page);
    const char* abase = reinterpret_cast<const char*>(d.a_split);
    const char* wbase = reinterpret_cast<const char*>(d.w_split);
    const int64_t rowbytes = (int64_t)cpb * PB;
    const int lane_off = ((lane & 3) ^ ((lane >> 4) & 3)) * 16;   // swizzled source octet of this lane's LDS slot

    using gptr_t = const __attribute__((address_space(1))) void*;
    using lptr_t = __attribute__((address_space(3))) void*;

    // ---- patch pieces ---------------------------------------------------------------------------------------------------
    // Source / destination of piece `id` of the patch of relative channel block ci (live: that block exists and the slot is a real
    // one).  Branch free (both candidates are computed, then selected: it runs among the MFMAs in front of the barrier): a wave-
    // uniform 64-bit base — the patch's first pixel, channel block ci — plus a 32-bit lane offset; lanes outside the image, beyond
    // the patch or of a dummy piece read the zero page, and a dummy piece lands in the zero region.
    const int64_t patch_pix0 = (int64_t)(img * d.H + r0 - 1) * W;   // (row r0 - 1 may lie outside the image: never dereferenced then)
    const char* patch_base0 = abase + (patch_pix0 * rowbytes + (int64_t)cb0 * PB);
    const unsigned rowbytes32 = (unsigned)rowbytes;                  // host: the patch spans < 2^31 bytes
    struct Piece {
        const char* src;
        int dst;   // LDS slot (wave uniform)
    };
    auto prep_patch_piece = [&](int id, int ci, bool live) -> Piece {
        const int j = id / NP, q = id - j * NP;
        const bool valid = live & (j < NCH);   // (& not &&: no short-circuit branch — a control-flow join in the loop makes hipcc
        const int px = j * 16 + (lane >> 2);   //  wait for the fresh fragment reads before the first MFMA, see igemm_dma.h)
        const int irow = r0 - 1 + (px >> lgW);
        const bool ok = valid & (px < PP) & ((unsigned)irow < (unsigned)d.H);
        const unsigned off = (unsigned)px * rowbytes32 + (unsigned)(q * 64 + lane_off);
        const char* real = patch_base0 + (int64_t)((ALDM_HALO_ABLATE & 128) ? 0 : ci) * PB + off;
        const char* zsrc = zero + lane_off;
        Piece pc;
        pc.src = ok ? real : zsrc;
        const int dreal = (ci & 1) * PATCH + (j * NP + q) * 64;
        pc.dst = valid ? dreal : Z0;
        return pc;
    };
    auto issue_piece = [&](const Piece& pc) {
        __builtin_amdgcn_global_load_lds((gptr_t)pc.src, (lptr_t)&smem[pc.dst], 16, 0, 0);
    };
    int av_s = 0, av_c = 1;   // next patch-piece slot: slot av_s of the patch of relative channel block av_c
    auto prep_a = [&]() {
        return prep_patch_piece(av_s * NW + wave, av_c, (ALDM_HALO_ABLATE & 1) ? false : (av_s < ATILES) & (av_c < ncb));
    };
    auto advance_a = [&]() {
        const bool wrap = av_s == 8;
        av_s = wrap ? 0 : av_s + 1;
        av_c += wrap ? 1 : 0;
    };

    // ---- weight pieces: chunk c = wave*NB + j -> (slot row = octet*NP + part, 64-column half) ---------------------------------
    // k-tile (cb, tap) is weight tile tap * cpb + cb.  A wave-uniform tile pointer advances by scalar adds; the lane part is
    // constant.  Columns beyond Npad (a 128-column tile over N = 192) re-read the last real column: finite values whose
    // accumulators the epilogue never stores.
    unsigned b_off[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int c = wave * NB + j;
        const int srow = c / (BN / 64), half = c % (BN / 64);
        const int col = min(n0 + half * 64 + lane, p.Npad - 1);
        b_off[j] = (unsigned)((srow * p.Npad + col) * 16);
    }
    const int64_t b_tile = (int64_t)4 * NP * p.Npad * 16;       // bytes of one weight k-tile
    const char* b_base = wbase + (int64_t)cb0 * b_tile;        // tile (tap 0, cb0)
    int bt_tap = 0;   // tap of the next weight k-tile to issue
    auto issue_b = [&](int st) {
        u32x4* sb = &smem[B0 + st * BSTG];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int c = wave * NB + j;
            const int srow = c / (BN / 64), half = c % (BN / 64);
            const char* src = (ALDM_HALO_ABLATE & 2) ? zero + lane * 16 : b_base + b_off[j];
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sb + srow * BN + half * 64), 16, 0, 0);
        }
    };
    auto advance_b = [&]() {   // tap inner: + cpb tiles; after tap 8 the next channel block's tap 0: + 1 - 8 cpb tiles
        const bool wrap = bt_tap == 8;
        bt_tap = wrap ? 0 : bt_tap + 1;
        if (!(ALDM_HALO_ABLATE & 64)) b_base += b_tile * (wrap ? 1 - 8 * cpb : cpb);
    };

    // ---- fragments ------------------------------------------------------------------------------------------------------
    struct Frag {
        bf16x8 a[MT][NP], b[NT][NP];
    };
    int pt[MT];            // tile-local output pixel of this lane's row in MFMA row tile i
#pragma unroll
    for (int i = 0; i < MT; ++i) pt[i] = (wm * MT + i) * 32 + l31;
    unsigned aaddr[MT];    // LDS byte address of this lane's k-octet lh fragment (step 0) for the tile being read; step 1: ^ 32
    unsigned aaddr_n[MT];  // ... for the tile after it (computed among the MFMAs in front of the barrier)
    int c_kh = 0, c_kw = 0, c_buf = 0;   // tap / patch buffer of the tile `aaddr_n` was last computed for
    auto set_aaddr_n = [&]() {   // branch free: both candidates, then a select
        const int toff = c_kh * W + c_kw - 1;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int pp = pt[i] + toff;
            const int cc = (pt[i] & (W - 1)) + c_kw - 1;
            const unsigned off = (unsigned)(pp >> 4) * (NP * 1024) + (unsigned)(pp & 15) * 64 + (unsigned)((((pp >> 2) & 3) ^ lh) << 4);
            const unsigned real = (unsigned)(c_buf * PATCH * 16) + off;
            const unsigned zslot = (unsigned)(Z0 * 16) + (off & 0xF0u);   // the zero slot of the SAME 16-byte bank group as the real slot
            aaddr_n[i] = (unsigned)cc < (unsigned)W ? real : zslot;
        }
    };
    auto next_tap = [&]() {
        const bool w1 = c_kw == 2;
        c_kw = w1 ? 0 : c_kw + 1;
        const bool w2 = w1 && c_kh == 2;
        c_kh = w2 ? 0 : (w1 ? c_kh + 1 : c_kh);
        c_buf ^= w2 ? 1 : 0;
        if (!(ALDM_HALO_ABLATE & 32)) set_aaddr_n();
    };
    auto take_aaddr = [&]() {
#pragma unroll
        for (int i = 0; i < MT; ++i) aaddr[i] = aaddr_n[i];
    };
    const char* lds = reinterpret_cast<const char*>(smem);
    auto read_frags = [&](Frag& f, int st, int step) {
#if ALDM_DMA_ABLATE & 8
        return;
#endif
        const int o = 2 * step + lh;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const u32x4* pa = reinterpret_cast<const u32x4*>(lds + (aaddr[i] ^ (unsigned)(step << 5)));
#pragma unroll
            for (int q = 0; q < NP; ++q) f.a[i][q] = __builtin_bit_cast(bf16x8, pa[q * 64]);
        }
        const u32x4* sb = &smem[B0 + st * BSTG];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int q = 0; q < NP; ++q)
                f.b[j][q] = __builtin_bit_cast(bf16x8, sb[(o * NP + q) * BN + (wn * NT + j) * 32 + l31]);
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
        for (int q = 0; q < NPROD; ++q)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = mfma_32x32x16<F16>(f.a[i][PA_[q]], f.b[j][PB_[q]], acc[i][j]);
    };
    // wait until at most `n` k-tile groups of this wave's DMA are still in flight (n is wave uniform)
    auto wait_tiles = [&](int n) {
        constexpr int MX = NSTB - 1;
        if (n <= 0) wait_vmcnt<0>();
        else if (n == 1) wait_vmcnt<D>();
        else if (n == 2) wait_vmcnt<(MX >= 2 ? 2 : MX) * D>();
        else if (n == 3) wait_vmcnt<(MX >= 3 ? 3 : MX) * D>();
        else wait_vmcnt<(MX >= 4 ? 4 : MX) * D>();
    };

    // ---- prologue: the zero region, the whole patch of the first channel block, slot 0 of the second, NSTB weight tiles ----------
    if (wave < NP)
        __builtin_amdgcn_global_load_lds((gptr_t)(zero + lane * 16), (lptr_t)&smem[Z0 + wave * 64], 16, 0, 0);
    for (int id = wave; id < NCH * NP; id += NW) issue_piece(prep_patch_piece(id, 0, true));
    issue_piece(prep_a());
    advance_a();
#pragma unroll
    for (int s = 0; s < NSTB; ++s)
        if (s < nk) {
            issue_b(s);
            advance_b();
        }
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    set_aaddr_n();
    take_aaddr();
    Frag f0, f1;
#if ALDM_DMA_ABLATE & 8
    for (int q = 0; q < NP; ++q) {
        for (int i = 0; i < MT; ++i) f0.a[i][q] = f1.a[i][q] = __builtin_bit_cast(bf16x8, smem[lane + q]);
        for (int j = 0; j < NT; ++j) f0.b[j][q] = f1.b[j][q] = __builtin_bit_cast(bf16x8, smem[lane + 64 + q]);
    }
#endif
    read_frags(f0, 0, 0);
    int st = 0, t = 0;
    // one k-tile with a successor: k-step 0's MFMAs | wait + barrier | (issue weight tile t + NSTB and a patch piece) | the next
    // tile's first fragments under k-step 1's MFMAs — the pipeline of igemm_dma_kernel
    auto body = [&](auto steady) {
        constexpr bool ST = decltype(steady)::value;
        constexpr int NMF = NPROD * MT * NT, NRD = NP * (MT + NT);
        read_frags(f1, st, 1);
        // what the second half needs right behind the barrier is computed here, in the shadow of k-step 0's MFMAs: the next
        // tile's fragment addresses and the patch piece's source pointer
        next_tap();
        Piece pc;
        if constexpr (ST) pc = prep_a();
        mma_frags(f0);
#pragma unroll
        for (int q = 0; q < NMF; ++q) {   // one fragment read and a few address instructions behind each of the first MFMAs
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (q < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        const int st1 = st + 1 == NSTB ? 0 : st + 1;
        // the wait includes lgkmcnt(0): this wave is done reading stage `st` (and, at a channel block's last tap, its patch)
        if constexpr (ST) wait_vmcnt<(NSTB - 2) * D>();
        else wait_tiles(min(NSTB - 2, nk - 2 - t));
        __builtin_amdgcn_s_barrier();
        if constexpr (ST) {
            issue_b(st);
            issue_piece(pc);
        }
        take_aaddr();
        read_frags(f0, st1, 0);
        mma_frags(f1);
#pragma unroll
        for (int q = 0; q < NMF; ++q) {   // MFMA first, the DMA issues and the next fragments in its shadow
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
            if (ST && q == 0) __builtin_amdgcn_sched_group_barrier(0x020, D, 1);
            if (q < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ST) {
            advance_b();
            advance_a();
        }
        st = st1;
    };
    for (; t + NSTB < nk; ++t) body(std::true_type{});
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
