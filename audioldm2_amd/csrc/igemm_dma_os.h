// igemm_dma_os.h — OPERAND-STATIONARY form of the DMA-fed bf16-split GEMM for short K: the transformer
// blocks' K = C projections (q/k/v, GEGLU, to_out, proj_in / proj_out: M = tokens x samples rows, K = 256 / 384).
//
// Why.  igemm_dma_kernel streams BOTH operands of every k-tile through LDS.  On its 64x128 tile a k-tile moves
// (64 + 128) x 4 x NP x 16 B = 36.9 KB (NP = 3) for 768 matrix-pipe cycles: 48 B per clock and CU against the ~56 B/clk/CU the
// L2 -> LDS path delivers — the K = 256 launches are bound by that feed and by their per-tile phases (8 k-tiles, then an
// epilogue), not by the MFMAs (profiles/r04_pmc_sq_bf16x6.txt: matrix pipe 30-39 % busy).  With K this short a wave can keep its
// whole weight slab in REGISTERS instead:
//   * block = 4 waves = 128 output columns; wave w owns columns [32w, 32w + 32) and loads their K x 32 weight fragments ONCE
//     (2K/16 x NP 16-byte pieces per lane: 192 VGPRs at K = 256, NP = 3; the block runs one wave per SIMD with the 512-register
//     budget), straight from the split weight image — no LDS, no re-fetch;
//   * the block then walks a chunk of rows, 32 at a time: a STAGE = 32 rows x all of K of the pre-split A image (49 KB at
//     K = 256, NP = 3) goes global -> LDS with global_load_lds_dwordx4 in the fragment order of igemm_dma.h (16-row x 4-octet
//     groups, octet XOR-swizzled by (row >> 2) & 3), NST stages in a ring;
//   * per stage every wave runs the WHOLE K loop on one 32x32 accumulator (2K/16 k-steps x NPROD MFMAs, 3 ds_read_b128 each):
//     L2 -> LDS traffic per matrix-pipe cycle drops 3x (16 B/clk/CU), there is one epilogue per 96 MFMAs instead of one per 24,
//     and one s_barrier per stage.
// K order, product order and epilogue arithmetic are those of igemm_dma_kernel, so results are BITWISE those of the classic
// kernel (tests/test_dma_gpu.py asserts torch.equal), including the GEGLU form (a value wave and its gate wave exchange their
// tiles through the staging area) and ALDM_EPI_QKV.
// Host-checked restrictions: 1x1 / linear launches (one tap, stride 1, no padding, no upsample), K = 32 KT with an instantiated
// KT, no split-K, no row remap.
#pragma once
#include "igemm_dma.h"

#ifndef ALDM_OS_ABLATE
#define ALDM_OS_ABLATE 0   // debug builds only (tools/gpu/build_variant.sh; results are wrong by construction): 1 no stage DMA inside
                           // the loop, 2 no MFMA, 4 no epilogue, 8 no fragment reads
#endif

namespace aldm {

constexpr int os_stage_slots(int KT, int NP) { return KT * 128 * NP; }   // 16-byte slots of one stage: 32 rows x K x NP parts
constexpr int OS_EPI_SLOTS = 1024;                                        // epilogue staging: 4 waves x 32 x 32 floats = 16 KB
constexpr int os_lds_bytes(int KT, int NST, int NP) { return (NST * os_stage_slots(KT, NP) + OS_EPI_SLOTS) * 16; }

template <int KT, int NST, int NP>
__global__ __launch_bounds__(256, 1)
void igemm_dma_os_kernel(const IgemmK p) {
    constexpr int KS = 2 * KT;                 // 16-wide k-steps
    constexpr int STG = os_stage_slots(KT, NP);
    constexpr int PB = 64 * NP;                // bytes of one (row, 32-channel block) of the A image
    constexpr int NPROD = NP == 3 ? 6 : 3;
    constexpr int ND = KT * NP / 2;            // LDS-DMA instructions per wave and stage (KT x 2 row groups x NP over 4 waves)
    static_assert(NP == 2 || NP == 3, "2 or 3 parts");
    static_assert(KT % 2 == 0 && KT >= 2, "the k-tiles of a stage are split between two wave pairs");
    static_assert(NST >= 2 && NST <= 4 && (NST - 2) * ND <= 63, "ring depth / vmcnt range");
    static_assert(os_lds_bytes(KT, NST, NP) <= 160 * 1024, "ring + staging must fit the CU's LDS");
    __shared__ u32x4 smem[NST * STG + OS_EPI_SLOTS];   // the ONLY LDS object (see igemm_dma.h)

    const aldm_igemm_desc& d = p.d;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, lh = lane >> 5;

    // XCD-aware remap (block b runs on XCD b % 8): consecutive logical ids — the column slabs of one row chunk, which re-read
    // the same A rows — share an XCD's L2
    int tile_n, chunk;
    {
        const int nwg = gridDim.x;
        const int bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        tile_n = logical % p.tiles_n;
        chunk = logical / p.tiles_n;
    }
    const int n0 = tile_n * 128;
    const int r0 = chunk * p.os_rows;
    const int r1 = min(p.M, r0 + p.os_rows);
    const int nstg = (r1 - r0 + 31) >> 5;
    if (nstg <= 0) return;

    const char* abase = reinterpret_cast<const char*>(d.a_split);
    const char* wbase = reinterpret_cast<const char*>(d.w_split);
    using gptr_t = const __attribute__((address_space(1))) void*;
    using lptr_t = __attribute__((address_space(3))) void*;

    // ---- A stage DMA: wave w fetches row group rg = w & 1 (16 rows) of the k-tiles [h KT/2, (h + 1) KT/2), h = w >> 1 ----
    const int rg = wave & 1, hk = wave >> 1;
    const int lane_off = ((lane & 3) ^ ((lane >> 4) & 3)) * 16;   // swizzled source octet of this lane's LDS slot
    const int64_t rowbytes = (int64_t)KT * PB;
    auto issue_stage = [&](int s) {
        const int m = min(r0 + s * 32 + rg * 16 + (lane >> 2), p.M - 1);   // rows past the end re-read the last row (discarded)
        const char* src = abase + (int64_t)m * rowbytes + hk * (KT / 2) * PB + lane_off;
        u32x4* dst = &smem[(s % NST) * STG + (hk * (KT / 2) * 2 * NP + rg * NP) * 64];
#pragma unroll
        for (int jj = 0; jj < KT / 2; ++jj)
#pragma unroll
            for (int q = 0; q < NP; ++q)
                __builtin_amdgcn_global_load_lds((gptr_t)(src + jj * PB + q * 64), (lptr_t)(dst + (jj * 2 * NP + q) * 64), 16, 0, 0);
    };

    issue_stage(0);
    // ---- the wave's weight slab, all of K, into registers: k-step ks, half lh -> k-octet 2 ks + lh of column n0 + 32 w + l31 ----
    bf16x8 bw[KS][NP];
    {
        const int col = min(n0 + wave * 32 + l31, p.Npad - 1);   // out-of-range columns duplicate the last one (discarded)
        const char* wp = wbase + ((int64_t)lh * NP * p.Npad + col) * 16;
        const int64_t ostep = (int64_t)2 * NP * p.Npad * 16;     // two k-octets = one k-step
        const int64_t pstep = (int64_t)p.Npad * 16;
        // Loaded straight into ACCUMULATION registers (gfx950: one 512-entry file per SIMD lane, MFMA A / B operands may be AGPRs):
        // the slab would otherwise fill the 256 architectural VGPRs, and the scheduler then serialises the A-fragment reads behind
        // the last use of their registers (one exposed LDS round trip per k-step).  Inline asm: hipcc does not count these loads —
        // they are all waited for by the explicit vmcnt(0) below, before anything uses them.
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                u32x4 t;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(t) : "v"(wp + ks * ostep + q * pstep) : "memory");
                bw[ks][q] = __builtin_bit_cast(bf16x8, t);
            }
    }
#pragma unroll
    for (int s = 1; s < NST - 1; ++s)
        if (s < nstg) issue_stage(s);

    // GEGLU: the bias of this wave pair's value / gate quads (columns are fixed for the block's lifetime)
    const bool geglu = d.epi_mode == ALDM_EPI_GEGLU;
    const int pair = wave >> 1, role = wave & 1;
    const int gr = lane >> 3, gc = (lane & 7) * 4;
    const int g_ncol_p = n0 + pair * 64 + gc;
    const int g_ncol_o = ((n0 + pair * 64) >> 1) + gc;
    const bool g_cok = g_ncol_p < d.N;
    f32x4 g_bv = {0.f, 0.f, 0.f, 0.f}, g_bg = {0.f, 0.f, 0.f, 0.f};
    if (geglu && d.bias && g_cok) {
        g_bv = *reinterpret_cast<const f32x4*>(d.bias + g_ncol_p);
        g_bg = *reinterpret_cast<const f32x4*>(d.bias + g_ncol_p + 32);
    }

    // fragment addressing: row l31 of the stage, k-step ks -> k-tile ks >> 1, octet 2 (ks & 1) + lh
    const int a_sw = (l31 >> 2) & 3;
    const int fbase = (l31 >> 4) * NP * 64 + (l31 & 15) * 4;
    const int foff0 = fbase + ((0 + lh) ^ a_sw), foff1 = fbase + ((2 + lh) ^ a_sw);
    constexpr int PA_[6] = {NP == 3 ? 0 : 1, NP == 3 ? 2 : 0, NP == 3 ? 1 : 0, 0, 1, 0};   // smallest partial products first
    constexpr int PB_[6] = {NP == 3 ? 2 : 0, NP == 3 ? 0 : 1, NP == 3 ? 1 : 0, 1, 0, 0};   // (the order of igemm_dma_kernel)
    float* const epi_lds = reinterpret_cast<float*>(&smem[NST * STG]);

    // everything issued so far has landed (stage 0 and the weights; the first wait also covers stages 1 .. NST - 2, which were
    // issued behind the weight loads: they are a few KB and needed one stage from now)
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();

    for (int t = 0; t < nstg; ++t) {
        const bool more = t + NST - 1 < nstg;
        if (more && !(ALDM_OS_ABLATE & 1)) issue_stage(t + NST - 1);   // into the buffer of stage t - 1: every wave is past the barrier that ended it
        const u32x4* sa = &smem[(t % NST) * STG];
        f32x16 acc[1][1];
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][0][e] = 0.f;
        bf16x8 af[2][NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) af[0][q] = __builtin_bit_cast(bf16x8, sa[q * 64 + foff0]);
#if ALDM_OS_ABLATE & 8
#pragma unroll
        for (int q = 0; q < NP; ++q) af[1][q] = af[0][q];
#endif
        // One k-step = NPROD dependent MFMAs on the tile's accumulator; the NEXT k-step's NP fragment reads go right behind the
        // first of them, so an LDS round trip (~130 cycles) has the other NPROD - 1 MFMAs (32 cycles each) to land under.  The
        // order is pinned with sched_barrier: left to itself hipcc sinks the last read behind the fragments' last use.
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (!(ALDM_OS_ABLATE & 2))
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][PA_[0]], bw[ks][PB_[0]], acc[0][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < KS && !(ALDM_OS_ABLATE & 8)) {
                const int kn = ks + 1;
#pragma unroll
                for (int q = 0; q < NP; ++q)
                    af[kn & 1][q] = __builtin_bit_cast(bf16x8, sa[(kn >> 1) * 128 * NP + q * 64 + ((kn & 1) ? foff1 : foff0)]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 1; q < NPROD; ++q)
                if (!(ALDM_OS_ABLATE & 2))
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][PA_[q]], bw[ks][PB_[q]], acc[0][0], 0, 0, 0);
#if ALDM_OS_ABLATE & 2
#pragma unroll
            for (int q = 0; q < NP; ++q) asm volatile("" ::"v"(af[ks & 1][q]), "a"(bw[ks][q]));
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        // Everything older than the newest NST - 2 stage issues has landed: stage t + 1 (issued one whole stage ago) and the
        // previous epilogue's stores.  Placed BEFORE this tile's epilogue so that its loads / stores are not waited for.
        if (t + 1 < nstg) {
            if (more) wait_vmcnt<(NST - 2) * ND>();
            else wait_vmcnt<0>();
        }
        const int m0 = r0 + t * 32;
        if ((ALDM_OS_ABLATE & 4) && p.M != -12345) {
            asm volatile("" ::"a"(acc[0][0]));
        } else if (geglu) {
            // value wave (role 0) and gate wave (role 1) of a 64-column group stage their tiles side by side: rows of
            // [32 value | 32 gate] floats, exactly the slab igemm_epilogue's GEGLU form sees in one wave
            float* stg = epi_lds + pair * (32 * 64);
#pragma unroll
            for (int e = 0; e < 16; ++e) stg[((e & 3) + 8 * (e >> 2) + 4 * lh) * 64 + role * 32 + l31] = acc[0][0][e];
            __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's staging writes are in LDS
            __builtin_amdgcn_s_barrier();
            float* go = d.out;
            const int gate_act = d.act == ALDM_ACT_GELU_TANH ? ALDM_ACT_GELU_TANH : ALDM_ACT_GELU;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int r = role * 16 + it * 8 + gr;
                f32x4 xv = *reinterpret_cast<const f32x4*>(&stg[r * 64 + gc]) + g_bv;
                const f32x4 xg = *reinterpret_cast<const f32x4*>(&stg[r * 64 + 32 + gc]) + g_bg;
#pragma unroll
                for (int c = 0; c < 4; ++c) xv[c] *= act_apply(xg[c], gate_act, 0.f);
                const int m = m0 + r;
                if (m < p.M && g_cok) {
                    if (go) *reinterpret_cast<f32x4*>(go + (int64_t)m * d.ldo + g_ncol_o) = xv;
                    if (d.out_split) split_store4(d.out_split, m, d.out_split_c, g_ncol_o, xv, d.split_parts);
                }
            }
        } else {
            igemm_epilogue<1, 1, 0>(p, acc, epi_lds, m0, n0, wave, 0, wave, lane, 0, 0);
        }
        __builtin_amdgcn_s_barrier();   // stage t + 1 is complete in LDS for every wave; stage t's buffer and the staging are free
    }
}

}  // namespace aldm
