// igemm_dma_os.h — OPERAND-STATIONARY form of the DMA-fed bf16-split GEMM for short K: the transformer
// blocks' K = C projections (q/k/v, GEGLU, to_out, proj_in / proj_out: M = tokens x samples rows, K = 256 / 384).
//
// Why.  igemm_dma_kernel streams BOTH operands of every k-tile through LDS.  On its 64x128 tile a k-tile moves
// (64 + 128) x 4 x NP x 16 B = 36.9 KB (NP = 3) for 768 matrix-pipe cycles: 48 B per clock and CU against the ~56 B/clk/CU the
// L2 -> LDS path delivers — the K = 256 launches are bound by that feed and by their per-tile phases (8 k-tiles, then an
// epilogue), not by the MFMAs (profiles/r04_before_pmc_sq_bf16x6.txt: matrix pipe 30-39 % busy).  With K this short a wave can
// keep its whole weight slab in REGISTERS instead:
//   * block = 8 waves = 128 output columns; wave w owns 16 of them and loads their K x 16 weight fragments ONCE (K/32 x NP
//     16-byte pieces per lane: 96 VGPRs at K = 256, NP = 3 — two waves per SIMD fit the register file), straight from the split
//     weight image: no LDS, no re-fetch;
//   * the block walks a chunk of rows, 32 at a time: a STAGE = 32 rows x all of K of the pre-split A image (49 KB at K = 256,
//     NP = 3) goes global -> LDS with global_load_lds_dwordx4 in 16-row x 4-octet groups (octet swizzled so that the 16x16x32
//     fragment reads are conflict free), NST stages in a ring, ONE s_barrier per stage;
//   * per stage every wave runs the WHOLE K loop on two 16x16 accumulators (v_mfma_f32_16x16x32_bf16: K/32 k-tiles x NPROD
//     MFMAs x 2 row tiles, 2 NP ds_read_b128 per k-tile): L2 -> LDS traffic per matrix-pipe cycle drops 3x, and there is one
//     epilogue per 96 MFMAs instead of one per 24;
//   * the two waves of a SIMD run the stage's two phases in OPPOSITE order — waves 0-3 multiply stage t and then store it, waves
//     4-7 store stage t - 1 and then multiply stage t — so that one wave's epilogue (loads of the residual, LDS transpose, GELU,
//     split, stores) runs under its partner's MFMAs although both meet at the same barrier.  (The first version, one wave per
//     SIMD with 32-column slabs on 32x32x16 MFMAs, spent 30 % of its time in exposed epilogues and 35 % in MFMAs:
//     profiles/r04_os_ablate_v1.txt.)
// The GEGLU form needs no exchange between waves: a wave's 16 columns are 8 value columns and THEIR 8 gate columns of the
// classic packed weight image (a lane loads any column it likes), combined in registers with one DPP row rotation.
// Products and their order per k-tile are those of igemm_dma_kernel; the accumulation inside a 16x16x32 MFMA differs from two
// chained 32x32x16 ones, so results agree with the classic kernel to fp32 rounding (tests/test_dma_gpu.py: 2e-6 max-norm), not
// bitwise.  Host-checked restrictions (igemm.hip, dma_os_eligible): 1x1 / linear launches (one tap, stride 1, no padding, no
// upsample), K = 32 KT with an instantiated KT, no split-K / row remap / row bias / output activation / accumulate, 16-byte
// aligned fp32 operands.
#pragma once
#include "igemm_dma.h"

#ifndef ALDM_OS_ABLATE
#define ALDM_OS_ABLATE 0   // debug builds only (tools/gpu/build_variant.sh; results are wrong by construction): 1 no stage DMA inside
                           // the loop, 2 no MFMA, 4 no epilogue, 8 the epilogue computes but never stores to global memory, 16 no
                           // fragment reads
#endif

#ifndef ALDM_OS_GEGLU_PACK
#define ALDM_OS_GEGLU_PACK 1   // 0: one GELU per accumulator register (round 4), for A/Bs
#endif

namespace aldm {

constexpr int os_stage_slots(int KT, int NP) { return KT * 128 * NP; }   // 16-byte slots of one stage: 32 rows x K x NP parts
constexpr int OS_EPI_SLOTS = 512;                                         // epilogue staging: 8 waves x 16 x 16 floats = 8 KB
constexpr int os_lds_bytes(int KT, int NST, int NP) { return (NST * os_stage_slots(KT, NP) + OS_EPI_SLOTS) * 16; }

// EPI (round 5): the epilogue a launch needs is known on the host, so it is a template parameter — the round-4 kernel chose among
// the plain / GEGLU / QKV forms, 2- or 3-part images and the gate activation with wave-uniform RUNTIME branches inside every
// 32-row stage (a dozen s_cbranch per 16x16 tile, each with its v_cmp / s_and and the scheduling fence a branch is), and loaded the
// residual with a global_load -> s_waitcnt vmcnt(0) pair in the middle of the epilogue: a full drain (the ring's LDS-DMA included)
// plus an L2 round trip per tile, per stage.  Now: one straight-line epilogue per instantiation, the image format is NP, and the
// residual rows of stage t are fetched at the TOP of stage t — they land under the stage's MFMAs.
enum { OS_EPI_PLAIN = 0, OS_EPI_GEGLU = 1, OS_EPI_QKV = 2 };

// NPO: parts of the split images the EPILOGUE writes (bf16 always); F16: the operands are "f16x3" images (NP = 2 fp16 parts of
// power-of-two scaled values, acc_scale undoes the scaling in front of the epilogue) — such launches write 3-part images.
// FO: out_split is the 2-part fp16 image of out_split_scale * result (the GEGLU output feeding an f16x3 FF-out GEMM).
template <int KT, int NST, int NP, int EPI, int NPO = NP, bool F16 = false, bool FO = false>
__global__ __launch_bounds__(512, 2)
void igemm_dma_os_kernel(const IgemmK p) {
    constexpr int STG = os_stage_slots(KT, NP);
    constexpr int PB = 64 * NP;                // bytes of one (row, 32-channel block) of the A image
    constexpr int NPROD = NP == 3 ? 6 : 3;
    constexpr int ND = KT * NP / 4;            // LDS-DMA instructions per wave and stage (KT x 2 row groups x NP over 8 waves)
    static_assert(NP == 2 || NP == 3, "2 or 3 parts");
    static_assert((NPO == 2 || NPO == 3) && (!F16 || NP == 2) && (!FO || EPI != OS_EPI_QKV || NPO == 2), "image formats");
    static_assert(KT % 4 == 0 && KT >= 4, "the k-tiles of a stage are split between four wave pairs");
    static_assert(NST >= 2 && NST <= 4 && (NST - 2) * ND <= 63, "ring depth / vmcnt range");
    static_assert(os_lds_bytes(KT, NST, NP) <= 160 * 1024, "ring + staging must fit the CU's LDS");
    __shared__ u32x4 smem[NST * STG + OS_EPI_SLOTS];   // the ONLY LDS object (see igemm_dma.h)

    const aldm_igemm_desc& d = p.d;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lc = lane & 15, lg = lane >> 4;   // MFMA 16x16x32: operand row / column, k-group (8 consecutive k = one octet)
    const bool late = wave >= 4;                // waves 4-7 share the SIMDs of waves 0-3: they store stage t - 1 BEFORE multiplying t

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

    // ---- A stage DMA: wave w fetches row group rg = w & 1 (16 rows) of the k-tiles [h KT/4, (h + 1) KT/4), h = w >> 1.  One
    // instruction = 16 rows x 4 octet slots of one (k-tile, part); the slot of row r holding octet o is o ^ sw(r) with
    // sw(r) = (4 - (r >> 2)) & 3: under it every ds_read_b128 lane group of the 16x16x32 fragment read (rows 0-3 / 12-15 of one
    // k-group with rows 4-11 of another) touches 16 different 16-byte bank groups ----
    const int rg = wave & 1, hk = wave >> 1;
    const int lane_off = ((lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3)) * 16;
    const int64_t rowbytes = (int64_t)KT * PB;
    auto issue_stage = [&](int s) {
        const int m = min(r0 + s * 32 + rg * 16 + (lane >> 2), p.M - 1);   // rows past the end re-read the last row (discarded)
        const char* src = abase + (int64_t)m * rowbytes + hk * (KT / 4) * PB + lane_off;
        u32x4* dst = &smem[(s % NST) * STG + (hk * (KT / 4) * 2 * NP + rg * NP) * 64];
#pragma unroll
        for (int jj = 0; jj < KT / 4; ++jj)
#pragma unroll
            for (int q = 0; q < NP; ++q)
                __builtin_amdgcn_global_load_lds((gptr_t)(src + jj * PB + q * 64), (lptr_t)(dst + (jj * 2 * NP + q) * 64), 16, 0, 0);
    };

    issue_stage(0);
    // ---- the wave's 16 weight columns, all of K, into registers: k-tile kt, k-group lg -> octet 4 kt + lg of column wcol ----
    constexpr bool geglu = EPI == OS_EPI_GEGLU;
    const int g64 = wave >> 2, j8 = wave & 3;   // GEGLU: 64-column group of the packed image, 8-column sub-slab
    bf16x8 bw[KT][NP];
    {
        int wcol = n0 + wave * 16 + lc;
        if (geglu) wcol = n0 + g64 * 64 + j8 * 8 + (lc < 8 ? lc : 32 + (lc - 8));   // 8 value columns ++ their 8 gate columns
        wcol = min(wcol, p.Npad - 1);                                                // out of range: duplicates (discarded)
        const char* wp = wbase + ((int64_t)lg * NP * p.Npad + wcol) * 16;
        const int64_t tstep = (int64_t)4 * NP * p.Npad * 16;   // four k-octets = one k-tile
        const int64_t pstep = (int64_t)p.Npad * 16;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int q = 0; q < NP; ++q)
                bw[kt][q] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wp + kt * tstep + q * pstep));
    }
#pragma unroll
    for (int s = 1; s < NST - 1; ++s)
        if (s < nstg) issue_stage(s);

    // ---- epilogue constants: the columns of a block never change ----
    constexpr bool qkv = EPI == OS_EPI_QKV;
    const int seg = qkv ? n0 / d.qkv_c : 0;               // q | k | v segment of the fused projection (whole 128-column slabs)
    float* outp = d.out;
    void* simg = d.out_split;
    int simg_c = d.out_split_c, col_shift = 0;
    if (qkv) {
        if (seg == 1) {
            outp = nullptr;
            simg = d.k_split;
            simg_c = d.qkv_c;
            col_shift = d.qkv_c;
        } else {
            simg = nullptr;
        }
    }
    // plain path: this lane's row / column quad of a staged 16x16 tile.  The tile is staged with its 4x4 row index transposed (row
    // 4 lg + i at position 4 i + lg): the four k-groups of one ds_write_b32 then hit four different 64-byte bank quarters (in row
    // order they are 256 bytes apart — the same 16 banks, a 4-way conflict on every accumulator write), and lane l reads position
    // l >> 2 = row 4 ((l >> 2) & 3) + (l >> 4), still one contiguous KB per wave
    const int er = ((lane >> 2) & 3) * 4 + (lane >> 4), ec = (lane & 3) * 4;
    const int ncol = n0 + wave * 16 + ec;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (!geglu && d.bias) {
#pragma unroll
        for (int c = 0; c < 4; ++c) bias4[c] = d.bias[min(ncol + c, d.N - 1)];
    }
    const bool has_res = EPI == OS_EPI_PLAIN && d.res != nullptr;
    const bool lrelu_img = EPI == OS_EPI_PLAIN && d.out_split_act == ALDM_ACT_LRELU;
    // GEGLU: lane column lc < 8 -> value column, lc >= 8 -> its gate column; outputs leave through lanes of an 8-column tile
    const int g_pcol = n0 + g64 * 64 + j8 * 8 + (lc < 8 ? lc : 32 + (lc - 8));
    const float g_bias = (geglu && d.bias && g_pcol < d.N) ? d.bias[g_pcol] : 0.f;
    // (staged like the plain tile: row 4 lg + i of a 16-row half at position 4 i + lg, so that the k-groups of one write do not share
    //  banks; lane l reads position l >> 1)
    const int g_pos = lane >> 1, g_ec = (lane & 1) * 4;
    const int g_er = (g_pos & 16) + (g_pos & 3) * 4 + ((g_pos & 15) >> 2);
    const int g_ncol_o = ((n0 + g64 * 64) >> 1) + j8 * 8 + g_ec;
    const bool g_cok = n0 + g64 * 64 + j8 * 8 + g_ec < d.N;

    // fragment addressing: row lc of row tile rt, k-tile kt, k-group lg -> octet lg of the k-tile
    const int foff = lc * 4 + (lg ^ ((4 - ((lc >> 2) & 3)) & 3));
    constexpr int PA_[6] = {NP == 3 ? 0 : 1, NP == 3 ? 2 : 0, NP == 3 ? 1 : 0, 0, 1, 0};   // smallest partial products first
    constexpr int PB_[6] = {NP == 3 ? 2 : 0, NP == 3 ? 0 : 1, NP == 3 ? 1 : 0, 1, 0, 0};   // (the order of igemm_dma_kernel)
    float* const stg = reinterpret_cast<float*>(&smem[NST * STG]) + wave * 256;             // this wave's 1 KB of staging

    // residual rows of a stage, fetched a whole K loop before their epilogue reads them (clamped address when out of range)
    f32x4 resv[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    auto prefetch_res = [&](int t) {
        if constexpr (EPI == OS_EPI_PLAIN) {
            if (has_res) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const int m = r0 + t * 32 + rt * 16 + er;
                    const bool ok = m < p.M && ncol < d.N;
                    resv[rt] = *reinterpret_cast<const f32x4*>(d.res + (ok ? (int64_t)m * d.ldo + ncol : 0));
                }
            }
        }
    };

    f32x4 acc[2];
    auto mfma16 = [&](const bf16x8 a, const bf16x8 b, const f32x4 c) -> f32x4 {
        if constexpr (F16)
            return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
        else
            return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    };
    auto epilogue = [&](int t) {
        if ((ALDM_OS_ABLATE & 4) && p.M != -12345) {
            asm volatile("" ::"v"(acc[0]), "v"(acc[1]));
            return;
        }
        const int m0 = (ALDM_OS_ABLATE & 8) ? p.M + (p.M == -12345 ? 0 : 64) : r0 + t * 32;   // (ablation: every row out of range)
        if constexpr (geglu) {
            // y = (value + b_v) * gelu(gate + b_g) (attention.py:37-45): the gate of value lane lc sits 8 lanes up in the same
            // 16-lane row -> one DPP row rotation; lanes lc < 8 then hold a 32 x 8 tile of outputs.  (erf GELU: the tanh-gated
            // form of T5 has K = 1024 and never runs here — host checked.)
#if ALDM_OS_GEGLU_PACK
            // Every lane runs every VALU instruction, but only the value lanes (lc < 8) keep a result: evaluating gelu(gate) once
            // per accumulator wastes half of each GELU.  Instead ONE GELU per register index i serves both row tiles: the gate lanes
            // evaluate their own gate of row tile 0, the value lanes the gate of row tile 1 (fetched from 8 lanes up by a DPP row
            // rotation whose bank mask writes the value lanes only) — 4 GELUs per stage instead of 8, same values, same results.
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float x0 = acc[0][i] + g_bias, x1 = acc[1][i] + g_bias;
                // lanes lc < 8 (banks 0, 1 of every 16-lane row): x1 of lane lc + 8 = the gate of row tile 1; lanes lc >= 8: own x0
                const float z = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, x0), __builtin_bit_cast(int, x1),
                                                                                      0x128 /* row_ror:8 */, 0xf, 0x3, false));
                const float gz = gelu_erf_fast(z);
                const float g0 = __builtin_bit_cast(   // gelu(gate of row tile 0) from the gate lanes
                    float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, gz), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
                if (lc < 8) {
                    stg[(i * 4 + lg) * 8 + lc] = x0 * g0;
                    stg[(16 + i * 4 + lg) * 8 + lc] = x1 * gz;
                }
            }
#else
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float x = acc[rt][i] + g_bias;
                    const float xg = __builtin_bit_cast(
                        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
                    const float y = x * gelu_erf_fast(xg);
                    if (lc < 8) stg[(rt * 16 + i * 4 + lg) * 8 + lc] = y;
                }
#endif
            const f32x4 v = *reinterpret_cast<const f32x4*>(&stg[g_pos * 8 + g_ec]);
            const int m = m0 + g_er;
            if (m < p.M && g_cok) {
                if (d.out) *reinterpret_cast<f32x4*>(d.out + (int64_t)m * d.ldo + g_ncol_o) = v;
                if (d.out_split) {
                    if constexpr (FO) split_store4_f16(d.out_split, m, d.out_split_c, g_ncol_o, v, d.out_split_scale);
                    else split_store4_t<NPO>(d.out_split, m, d.out_split_c, g_ncol_o, v);
                }
            }
            return;
        } else {
            if constexpr (qkv) {
                if (seg == 2) {
                    // v: transposed per (sample, head, 32-key tile) straight from the accumulators.  The image keeps the 32 keys of a
                    // tile in the order the attention kernel's P operand has them (the 32x32 MFMA accumulator rows: 16-byte chunk 2 s + h
                    // of a dim's 64-byte row = keys 16 s + 4 h + {0..3, 8..11}); lane (column lc, k-group lg) holds keys 4 lg .. 4 lg + 3
                    // of row tile rt = half of chunk 2 rt + (lg & 1)
                    char* vt = reinterpret_cast<char*>(d.vt_split);
                    const int heads = d.qkv_c >> 5;
                    const int tiles = d.qkv_rows >> 5;
                    if (m0 >= p.M) return;
                    const int b = m0 / d.qkv_rows, tl = (m0 - b * d.qkv_rows) >> 5;
                    const int cc = n0 - 2 * d.qkv_c + wave * 16 + lc;
                    const int h = cc >> 5, dd = cc & 31;
                    char* base = vt + ((((int64_t)b * heads + h) * tiles + tl) * NPO) * 2048 + dd * 64 + (lg & 1) * 16 + (lg >> 1) * 8;
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) {
                        u32x2 part[3];
                        if constexpr (FO) split4_f16(acc[rt], d.vt_scale, part);
                        else split4_parts(acc[rt], part, NPO);
#pragma unroll
                        for (int q = 0; q < NPO; ++q) *reinterpret_cast<u32x2*>(base + q * 2048 + rt * 32) = part[q];
                    }
                    return;
                }
            }
            // plain / q / k: transpose each 16x16 tile through the wave's staging so that a lane owns 4 consecutive columns, then
            //   v = acc + bias; v += res; v *= alpha; out = v; out_split = split([leaky_relu] v)        (igemm_epilogue's order)
            // Both row tiles' values are finished BEFORE the first store: a store counts in vmcnt like a load, so reading the second
            // tile's prefetched residual after the first tile's stores would wait for those stores to be acknowledged.
            f32x4 v[2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
                for (int i = 0; i < 4; ++i) stg[(i * 4 + lg) * 16 + lc] = acc[rt][i];
                v[rt] = *reinterpret_cast<const f32x4*>(&stg[(lane >> 2) * 16 + ec]);
                v[rt] += bias4;
                if (has_res) v[rt] += resv[rt];
                v[rt] *= d.alpha;
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int m = m0 + rt * 16 + er;
                const bool ok = m < p.M && ncol < d.N;
                const int64_t off = ok ? (int64_t)m * d.ldo + (ncol - col_shift) : 0;
                if (ok) {
                    if (outp) *reinterpret_cast<f32x4*>(outp + off) = v[rt];
                    if (simg) {
                        if (lrelu_img) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) v[rt][c] = v[rt][c] > 0.0f ? v[rt][c] : v[rt][c] * d.out_split_slope;
                        }
                        if constexpr (FO) split_store4_f16(simg, m, simg_c, ncol - col_shift, v[rt], d.out_split_scale);
                        else split_store4_t<NPO>(simg, m, simg_c, ncol - col_shift, v[rt]);
                    }
                }
            }
        }
    };

    // everything issued so far has landed (stage 0 and the weights; the first wait also covers stages 1 .. NST - 2, issued
    // behind the weight loads)
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();

    for (int t = 0; t < nstg; ++t) {
        if (late && t > 0) epilogue(t - 1);   // under the partner wave's MFMAs of stage t
        prefetch_res(t);                      // this stage's residual rows: in flight under its K loop (older than the stage DMA below)
        const bool more = t + NST - 1 < nstg;
        if (more && !(ALDM_OS_ABLATE & 1)) issue_stage(t + NST - 1);   // into the buffer of stage t - 1: every wave is past its barrier
        const u32x4* sa = &smem[(t % NST) * STG];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[rt][e] = 0.f;
        bf16x8 af[2][2][NP];   // [k-tile parity][row tile][part]
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int q = 0; q < NP; ++q) af[0][rt][q] = __builtin_bit_cast(bf16x8, sa[(rt * NP + q) * 64 + foff]);
#if ALDM_OS_ABLATE & 16
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int q = 0; q < NP; ++q) af[1][rt][q] = af[0][rt][q];
#endif
        // One k-tile = NPROD MFMAs on each of the two row tiles' accumulators (two independent chains); the NEXT k-tile's 2 NP
        // fragment reads go right behind the first pair, so an LDS round trip has the other 2 NPROD - 2 MFMAs (16 cycles each,
        // plus the partner wave's) to land under.  The order is pinned with sched_barrier: left to itself hipcc sinks the reads
        // behind the fragments' last use.
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            if (!(ALDM_OS_ABLATE & 2)) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
                    acc[rt] = mfma16(af[kt & 1][rt][PA_[0]], bw[kt][PB_[0]], acc[rt]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 1 < KT && !(ALDM_OS_ABLATE & 16)) {
                const int kn = kt + 1;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int q = 0; q < NP; ++q)
                        af[kn & 1][rt][q] = __builtin_bit_cast(bf16x8, sa[kn * 128 * NP + (rt * NP + q) * 64 + foff]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!(ALDM_OS_ABLATE & 2)) {
#pragma unroll
                for (int q = 1; q < NPROD; ++q)
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
                        acc[rt] = mfma16(af[kt & 1][rt][PA_[q]], bw[kt][PB_[q]], acc[rt]);
            } else {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int q = 0; q < NP; ++q) asm volatile("" ::"v"(af[kt & 1][rt][q]), "v"(bw[kt][q]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (F16) {   // (s_a x) . (s_w w) -> x . w: exact, powers of two
            acc[0] *= d.acc_scale;
            acc[1] *= d.acc_scale;
        }
        // Everything older than the newest NST - 2 stage issues has landed: stage t + 1 (issued one whole stage ago) and the
        // previous epilogue's stores.  Placed BEFORE an early wave's epilogue so that its loads / stores are not waited for.
        if (t + 1 < nstg) {
            if (more) wait_vmcnt<(NST - 2) * ND>();
            else wait_vmcnt<0>();
        }
        if (!late) epilogue(t);
        __builtin_amdgcn_s_barrier();   // stage t + 1 is complete in LDS for every wave; stage t's buffer is free
    }
    if (late) epilogue(nstg - 1);
}

}  // namespace aldm
