// igemm_dma_ws.h — the PERSISTENT, WAVE-SPECIALISED form of the DMA-fed bf16-split implicit GEMM (igemm_dma.h).
//
// Why (profiles/r02_dma_ablate_shapes.txt): for the transformer blocks' short-K GEMMs the three phases of a
// block — launch + first LDS-DMA round trip, K loop, epilogue (LDS transpose, bias / GELU / gate / split, stores) — ADD UP
// (11 + 40 + 28 us for the GEGLU projection), because a wave issues in order: while it runs its epilogue its SIMD's matrix
// pipe idles, and co-resident blocks of one launch run in lockstep.  Here one 512-thread block per CU stays resident and
// walks a contiguous run of output tiles with its waves split into two ROLES:
//   waves 0-3 ("MMA", one per SIMD): nothing but LDS-DMA issues, fragment reads and MFMAs — the K loop of igemm_dma.h,
//       tile after tile.  When a tile's accumulators are complete they are written to an LDS hand-off buffer (64
//       ds_write_b32 per wave) and the next tile's first ring stages are already in flight.  These waves never issue a global
//       load or store of their own, so their vmcnt counts LDS-DMA only and every wait stays a COUNTED one (on gfx9 loads and
//       stores share vmcnt, which is what kept the epilogue's stores out of a single-role persistent loop);
//   waves 4-7 ("EPI", one per SIMD): the epilogue of the PREVIOUS tile — they read the hand-off buffer as float4 rows, apply
//       bias / row bias / activation / GEGLU / residual / alpha, and store fp32 and / or the split image — while the MMA
//       waves run the current tile's K loop on the same SIMDs: VALU and memory instructions of one wave issue beside the
//       MFMAs of the other (separate pipes).  Residual / row-bias / bias operands are fetched ONE TILE AHEAD.
// Synchronisation is the workgroup barrier only (gfx950 has no named barriers), so BOTH roles execute exactly the same
// barrier sequence per tile:  nk - 1 in-loop barriers (the K loop's ring hand-offs; the EPI waves interleave their items
// between them, a fixed quota per barrier, so they never hold the MMA waves up for longer than one item), then
//     X   every MMA wave is past its last fragment read (ring free) AND the EPI waves are done with the hand-off buffer,
//     YP  the accumulators of the tile just finished are in the hand-off buffer AND stage 0 of the next tile has landed.
// Restrictions (host checked, everything else runs on igemm_dma_kernel): no split-K, no row remap, no accumulate, no
// epilogue activation other than the GEGLU erf gate, whole tiles (M % BM == 0, N % BN == 0), 16-byte aligned N / pitches.
#pragma once
#include "igemm_dma.h"

namespace aldm {

constexpr int ws_lds_bytes_(int BM, int BN, int NST, int NP) {
    return NST * dma_stage_slots(BM, BN, NP) * 16 + 4 * (BM / 64) * 32 * ((BN / 64) * 32) * 4;
}
// hand-off buffer: the block's BM x BN accumulators, row pitch = the wave slab's width (no padding: a half wave writes 32
// consecutive floats of one row, a ds_read_b128 lane group reads 16 distinct float4 of two rows — conflict free as it is)
constexpr int ws_acc_floats(int BM, int BN) { return 4 * (BM / 64) * 32 * ((BN / 64) * 32); }
constexpr int ws_blocks_per_cu(int BM, int BN, int NST, int NP) { return 2 * ws_lds_bytes_(BM, BN, NST, NP) <= 160 * 1024 ? 2 : 1; }
constexpr int ws_lds_bytes(int BM, int BN, int NST, int NP) { return ws_lds_bytes_(BM, BN, NST, NP); }

// Two blocks share a CU (16 waves, 128 registers each) when two rings + hand-off buffers fit the 160 KiB of LDS.
template <int BM, int BN, int NST, int NP>
__global__ __launch_bounds__(512, 2 * ws_blocks_per_cu(BM, BN, NST, NP))
void igemm_dma_ws_kernel(const IgemmK p) {
    constexpr int WN = 2, NW = 4;
    constexpr int MT = BM / 64, NT = BN / 64;
    constexpr int STG = dma_stage_slots(BM, BN, NP);
    constexpr int PB = 64 * NP;
    constexpr int RA = BM / (16 * NW);
    constexpr int NB = 4 * NP * (BN / 64) / NW;
    constexpr int D = NP * RA + NB;
    constexpr int NPROD = NP == 3 ? 6 : 3;
    constexpr int SP = NT * 32;              // hand-off row pitch (floats)
    constexpr int SLAB = 32 * SP;            // one 32-row slab of one wave
    static_assert(NP == 2 || NP == 3, "2 or 3 parts");
    static_assert(BM % (16 * NW) == 0 && (4 * NP * (BN / 64)) % NW == 0, "DMA chunks must divide among the waves");
    static_assert(NST >= 2 && NST <= 8 && (NST - 1) * D <= 63, "ring depth / vmcnt range");
    static_assert(ws_lds_bytes(BM, BN, NST, NP) <= 160 * 1024, "LDS budget");
    __shared__ u32x4 smem[NST * STG + ws_acc_floats(BM, BN) / 4];   // ring, then the hand-off buffer: ONE LDS object
    float* accbuf = reinterpret_cast<float*>(&smem[NST * STG]);

    const aldm_igemm_desc& d = p.d;
    const int lane = threadIdx.x & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = wave8 & 3;                 // index inside the role
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;

    // this block's contiguous run of output tiles (tile_n fastest: consecutive tiles share their A rows in L1 / L2)
    const int ntiles = p.tiles_m * p.tiles_n;
    const int per = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int t_begin = blockIdx.x * per;
    const int t_end = min(ntiles, t_begin + per);
    if (t_begin >= t_end) return;               // (whole block; before any barrier)
    const int nk = d.K >> 5;

    if (wave8 < 4) {
        // =============================================== MMA role ===================================================
        __builtin_amdgcn_s_setprio(2);
        const char* zero = reinterpret_cast<const char*>(g_dma_zero_page);
        const char* abase = reinterpret_cast<const char*>(d.a_split);
        const char* wbase = reinterpret_cast<const char*>(d.w_split);
        const int cpb = p.Cin >> 5;
        const int taps = d.KH * d.KW;
        const int64_t rowbytes = (int64_t)cpb * PB;
        const int lane_off = ((lane & 3) ^ ((lane >> 4) & 3)) * 16;
        int a_pix[RA], a_h[RA], a_w[RA];
        const char* a_ptr[RA];
        int a_step[RA];
        int t_kh, t_kw, t_cb;
        const char* b_ptr[NB];
        int64_t b_tile[NB];
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
        auto setup_tile = [&](int ti) {   // gather state of k-tile 0 of output tile ti
            const int tile_m = ti / p.tiles_n, tile_n = ti - tile_m * p.tiles_n;
            const int m0 = tile_m * BM, n0 = tile_n * BN;
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
            t_kh = t_kw = t_cb = 0;
            set_tap();
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int c = wave * NB + j;
                const int srow = c / (BN / 64), half = c % (BN / 64);
                const int col = n0 + half * 64 + lane;
                const bool ok = col < p.Npad;
                b_tile[j] = ok ? (int64_t)4 * NP * p.Npad * 16 : 0;
                b_ptr[j] = ok ? wbase + ((int64_t)srow * p.Npad + col) * 16 : zero;
            }
        };
        using gptr_t = const __attribute__((address_space(1))) void*;
        using lptr_t = __attribute__((address_space(3))) void*;
        auto issue_dma = [&](int st) {
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
        };
        auto advance = [&]() {   // tap outer, channel block inner (the shipped K order of igemm_dma.h)
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
        auto zero_acc = [&]() {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        };
        auto mma_frags = [&](const Frag& f) {
            constexpr int PA_[6] = {NP == 3 ? 0 : 1, NP == 3 ? 2 : 0, NP == 3 ? 1 : 0, 0, 1, 0};
            constexpr int PB_[6] = {NP == 3 ? 2 : 0, NP == 3 ? 0 : 1, NP == 3 ? 1 : 0, 1, 0, 0};
#pragma unroll
            for (int q = 0; q < NPROD; ++q)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][PA_[q]], f.b[j][PB_[q]], acc[i][j], 0, 0, 0);
        };
        auto wait_tiles = [&](int n) {   // at most n k-tiles of this thread's DMA still in flight (+ lgkmcnt(0))
            constexpr int MX = NST - 1;
            if (n <= 0) wait_vmcnt<0>();
            else if (n == 1) wait_vmcnt<D>();
            else if (n == 2) wait_vmcnt<(MX >= 2 ? 2 : MX) * D>();
            else if (n == 3) wait_vmcnt<(MX >= 3 ? 3 : MX) * D>();
            else if (n == 4) wait_vmcnt<(MX >= 4 ? 4 : MX) * D>();
            else if (n == 5) wait_vmcnt<(MX >= 5 ? 5 : MX) * D>();
            else wait_vmcnt<(MX >= 6 ? 6 : MX) * D>();
        };
        auto issue_first_stages = [&]() {
#pragma unroll
            for (int s = 0; s < NST; ++s)
                if (s < nk) {
                    issue_dma(s);
                    advance();
                }
        };

        setup_tile(t_begin);
        issue_first_stages();
        zero_acc();
        wait_tiles(min(nk, NST) - 1);
        __builtin_amdgcn_s_barrier();   // P0
        for (int ti = t_begin; ti < t_end; ++ti) {
            // ---- K loop of tile ti: identical to igemm_dma_kernel's (nk - 1 barriers) ----
            Frag f0, f1;
            read_frags(f0, 0, 0);
            int st = 0, t = 0;
            auto body = [&](auto steady) {
                constexpr bool ST = decltype(steady)::value;
                constexpr int NMF = NPROD * MT * NT, NRD = NP * (MT + NT);
                read_frags(f1, st, 1);
                mma_frags(f0);
#pragma unroll
                for (int q = 0; q < NMF; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (q < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                const int st1 = st + 1 == NST ? 0 : st + 1;
                if constexpr (ST) wait_vmcnt<(NST - 2) * D>();
                else wait_tiles(min(NST - 2, nk - 2 - t));
                __builtin_amdgcn_s_barrier();
                if constexpr (ST) issue_dma(st);
                read_frags(f0, st1, 0);
                mma_frags(f1);
#pragma unroll
                for (int q = 0; q < NMF; ++q) {
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
            read_frags(f1, st, 1);
            mma_frags(f0);
            mma_frags(f1);
            // ---- hand-over ----
            const bool more = ti + 1 < t_end;
            if (more) setup_tile(ti + 1);          // address arithmetic under the draining MFMAs
            wait_vmcnt<0>();                       // (nothing is in flight; lgkmcnt(0): the last fragments are in registers)
            __builtin_amdgcn_s_barrier();          // X: ring free, hand-off buffer free
            if (more) issue_first_stages();        // the next tile's first stages fly while the accumulators move out
            float* stg = accbuf + wave * (MT * SLAB);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        stg[i * SLAB + ((e & 3) + 8 * (e >> 2) + 4 * lh) * SP + j * 32 + l31] = acc[i][j][e];
            zero_acc();
            if (more) wait_tiles(min(nk, NST) - 1);   // stage 0 landed (and lgkmcnt(0): the hand-off writes are done)
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();          // YP
        }
        return;
    }

    // ================================================= EPI role =====================================================
    // Host guarantees: plain epilogue with act NONE (alpha applied) or GEGLU with the erf gate; M % BM == 0 and N % BN == 0
    // (no ragged tiles: the item code is straight-line); N % 4 == 0 and 16-byte aligned pointers / pitches; a row bias only
    // when every tile lies inside one sample (OHW % BM == 0).
    // Schedule of one tile-slot (the MMA waves run the K loop of tile ti, this role owns the accumulators of tile ti - 1):
    //     `lead` idle barriers | items, `quota` per barrier | fetch the NEXT tile's bias / residual rows | idle barriers | X | YP
    // The operand loads are issued AFTER the slot's stores and used one slot later: loads and stores share vmcnt on gfx9 and
    // may retire out of order, so the only safe wait for a load that is older than stores is vmcnt(0) — this order makes that
    // wait (inserted by hipcc at the first item of the next slot) cover nothing that is still young.  Absent operands are
    // read from the zero page: no data-dependent control flow between the first item and the last.
    {
        constexpr int C4 = NT * 8;        // float4 per hand-off row
        constexpr int RPI = 64 / C4;      // rows covered by one wave-wide float4 read
        constexpr int IT = 32 / RPI;      // reads per 32-row slab
        constexpr int W = MT * IT;        // items per tile, plain epilogue: item k = rows mrow0 + k*RPI of the wave's slabs
        constexpr int WG = MT * 4;        // items per tile, GEGLU epilogue (8 rows x 32 outputs per item)
        const float* stg = accbuf + wave * (MT * SLAB);
        const bool geglu = NT == 2 && d.epi_mode == ALDM_EPI_GEGLU;
        const int sr = lane / C4, sc = (lane % C4) * 4;
        const int gr = lane >> 3, gc = (lane & 7) * 4;
        float* const outp = d.out;
        char* const simg = reinterpret_cast<char*>(d.out_split);
        const float* const zero = reinterpret_cast<const float*>(g_dma_zero_page);
        const int ldo = d.ldo;
        const float alpha = d.alpha;
        const int nb_slots = nk - 1;                        // in-loop barriers per tile
        const int lead = nb_slots >> 2;
        const int items = geglu ? WG : W;
        const int work_slots = nb_slots - lead;
        const int quota = work_slots > 0 ? (items + work_slots - 1) / work_slots : items;
        int bars = 0, cnt = 0;
        auto tick = [&]() {   // after every item: one barrier per `quota` items while in-loop barriers are left
            if (++cnt >= quota && bars > 0) {
                __builtin_amdgcn_s_barrier();
                --bars;
                cnt = 0;
            }
        };
        f32x4 aux[W];   // residual rows of the next tile's items (plain epilogue)
        f32x4 b0, b1;   // plain: bias quad, row-bias quad; GEGLU: value and gate bias quads
        auto prefetch = [&](int ti) {
            const int tile_m = ti / p.tiles_n, tile_n = ti - tile_m * p.tiles_n;
            const int m0 = tile_m * BM, n0 = tile_n * BN;
            if (geglu) {
                const int ncol_p = n0 + wn * 64 + gc;
                const float* bp = d.bias ? d.bias + ncol_p : zero;
                b0 = *reinterpret_cast<const f32x4*>(bp);
                b1 = *reinterpret_cast<const f32x4*>(bp + 32);
                return;
            }
            const int ncol = n0 + wn * NT * 32 + sc;
            b0 = *reinterpret_cast<const f32x4*>(d.bias ? d.bias + ncol : zero);
            b1 = *reinterpret_cast<const f32x4*>(d.rowbias ? d.rowbias + (int64_t)(m0 / p.OHW) * p.rb_ld + ncol : zero);
            const int mrow0 = m0 + wm * MT * 32 + sr;
            const float* rp = d.res ? d.res + (int64_t)mrow0 * ldo + ncol : zero;
            const int64_t rstep = d.res ? (int64_t)RPI * ldo : 0;
#pragma unroll
            for (int k = 0; k < W; ++k) aux[k] = *reinterpret_cast<const f32x4*>(rp + k * rstep);
        };
        auto process = [&](int ti, auto has_out, auto has_split) {
            constexpr bool HO = decltype(has_out)::value, HS = decltype(has_split)::value;
            const int tile_m = ti / p.tiles_n, tile_n = ti - tile_m * p.tiles_n;
            const int m0 = tile_m * BM, n0 = tile_n * BN;
            if (geglu) {
                const int ncol_o = ((n0 + wn * 64) >> 1) + gc;
#pragma unroll
                for (int k = 0; k < WG; ++k) {
                    const int i = k >> 2, it = k & 3;
                    const int r = it * 8 + gr;
                    f32x4 xv = *reinterpret_cast<const f32x4*>(&stg[i * SLAB + r * SP + gc]) + b0;
                    const f32x4 xg = *reinterpret_cast<const f32x4*>(&stg[i * SLAB + r * SP + 32 + gc]) + b1;
#pragma unroll
                    for (int c = 0; c < 4; ++c) xv[c] *= gelu_erf_fast(xg[c]);
                    const int m = m0 + (wm * MT + i) * 32 + r;
                    if constexpr (HO) *reinterpret_cast<f32x4*>(outp + (int64_t)m * ldo + ncol_o) = xv;
                    if constexpr (HS) split_store4(simg, m, d.out_split_c, ncol_o, xv, NP);
                    tick();
                }
                return;
            }
            const int ncol = n0 + wn * NT * 32 + sc;
            const int mrow0 = m0 + wm * MT * 32 + sr;
#pragma unroll
            for (int k = 0; k < W; ++k) {
                const int m = mrow0 + k * RPI;   // slab k / IT, row (k % IT) * RPI + sr: the slabs are 32 rows apart
                f32x4 v = *reinterpret_cast<const f32x4*>(&stg[(k / IT) * SLAB + ((k % IT) * RPI + sr) * SP + sc]);
                v = ((v + b0) + b1) + aux[k];
                v *= alpha;
                if constexpr (HO) *reinterpret_cast<f32x4*>(outp + (int64_t)m * ldo + ncol) = v;
                if constexpr (HS) split_store4(simg, m, d.out_split_c, ncol, v, NP);
                tick();
            }
        };
        auto process_any = [&](int ti) {
            if (outp && simg) process(ti, std::true_type{}, std::true_type{});
            else if (simg) process(ti, std::false_type{}, std::true_type{});
            else process(ti, std::true_type{}, std::false_type{});
        };
        __builtin_amdgcn_s_barrier();   // P0
        prefetch(t_begin);
        for (int ti = t_begin; ti < t_end; ++ti) {
            bars = nb_slots;
            cnt = 0;
            if (ti > t_begin) {
                for (int l = 0; l < lead; ++l) {
                    __builtin_amdgcn_s_barrier();
                    --bars;
                }
                process_any(ti - 1);
                prefetch(ti);
            }
            while (bars > 0) {
                __builtin_amdgcn_s_barrier();
                --bars;
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();   // X: every item of the previous tile has left the hand-off buffer
            __builtin_amdgcn_s_barrier();   // YP: the accumulators of tile ti are in the hand-off buffer
            asm volatile("" ::: "memory");
        }
        bars = 0;
        process_any(t_end - 1);
    }
}

}  // namespace aldm
