// igemm_kernel.h — the implicit-GEMM convolution / GEMM kernel, on the fp32 MFMA (v_mfma_f32_32x32x2_f32) or,
// with BX = true, on the bf16 matrix cores with every fp32 product evaluated as six bf16 partial products of
// exact 3-way operand splits (v_mfma_f32_32x32x16_bf16, "BF16x6"; the default path — see the notes at Frag /
// the BX branch of the K loop below and docs/experiments_r1-r6.md §3.1b).
//
//   out[m, n] = epi( sum_k A[m, k] * W[k, n] ),  m = (b, oh, ow), k = (kh, kw, ci)
//
// Design (MI355X-first, not a port of any cuDNN/ATen algorithm):
//  * activations are channels-last, so 4 consecutive k of one tap are one 16-byte load and a
//    1x1 conv, a Linear and a conv tap are the same gather;
//  * block tile BM x BN x 32, WM x WN wave64 (4 = one per SIMD; 8 on the 128x128 prologue convs); each
//    wave owns MT x NT MFMA 32x32 tiles (16 accumulator VGPRs each);
//  * LDS image is k-group major: As[kg][row] / Bs[kg][col] hold float4 = 4 consecutive k, so one
//    conflict-free ds_read_b128 feeds 4 MFMAs.  The K index inside the 8-wide sub-step is
//    permuted (lane half h takes k = 4h..4h+3) — legal because A and B use the same permutation;
//  * software pipeline, ONE barrier per k-tile: LDS is double buffered; the global loads of tile
//    t+1 are issued (branch-free: clamped addresses + a validity mask) before the MFMAs of tile t,
//    their prologue transform + ds_write into the other buffer happens between the two halves of
//    the MFMA block.  fp32 MFMA is 64 cycles/instruction, so one tile = 1024..4096 MFMA cycles per
//    SIMD: HBM/L2 latency and the prologue VALU sit in that shadow;
//  * the prologue is a template parameter (no runtime switch in the loop): GroupNorm apply
//    (+SiLU), leaky_relu; skip-concat (two source tensors), nearest upsample, stride, dilation
//    and padding are part of the address generation; (tap, ci) are tracked incrementally;
//  * split-K (blockIdx.y): raw partial tiles go to a workspace, igemm_reduce_kernel sums them
//    in a fixed order and applies the epilogue (deterministic, no atomics);
//  * epilogue fusion: bias, timestep-embedding row bias, activation, residual, scale, accumulate,
//    strided row remap (polyphase transposed conv);
//  * blockIdx is remapped so each XCD (private 4 MiB L2) walks a contiguous range of tiles.
// The bullets on the LDS image and the software pipeline describe the fp32-MFMA instantiations; the BX ones keep
// gather, prologues, epilogues and split-K but use a 12-slot bf16 image and a deeper pipeline (below).
#pragma once
#include "igemm_epilogue.h"
#include <type_traits>

namespace aldm {


#ifndef ALDM_BX_INTERLEAVE
#define ALDM_BX_INTERLEAVE 1
#endif
// Experimental (not validated on hardware yet, compiled out): the pre-split B operand of the bf16-split kernels
// goes global -> LDS directly (global_load_lds_dwordx4: no staging registers, no ds_write), one k-tile ahead.
#ifndef ALDM_BX_GLDS
#define ALDM_BX_GLDS 0
#endif
#ifndef ALDM_ABLATE
#define ALDM_ABLATE 0  // debug builds only (tools/gpu/build_ablate.sh): drop pieces of the BX K loop to time the rest
#endif
constexpr int BK = 32;
constexpr int KG = BK / 4;

// resident blocks per CU the register budget must allow (LDS allows 2 / 3 / 4 for the three tile sizes)
constexpr int igemm_min_blocks(int BM, int BN) {
    return BM * BN >= 128 * 128 ? 2 : ((BM * BN >= 64 * 128 || BN == 32) ? 3 : 4);
}

// LDS bytes of one block (A/B double buffers; BX = bf16-split image, see igemm_kernel)
constexpr int igemm_lds_bytes(int BM, int BN, int kgrp, bool BX) {
    return BX ? kgrp * 2 * 12 * (BM + 2 + BN + 2) * 16 : kgrp * 2 * 8 * (BM + 1 + BN + 1) * 16;
}
// waves per SIMD the register budget must allow = what the LDS lets be resident (waves = WM*WN*KGRP)
constexpr int igemm_waves_per_simd(int BM, int BN, int waves, bool BX) {
    if (!BX) return waves == 8 ? 4 : igemm_min_blocks(BM, BN);
    const int kgrp = (waves == 8 && BM == 64 && BN == 64) ? 2 : 1;
    const int blocks = (160 * 1024) / igemm_lds_bytes(BM, BN, kgrp, true);
    const int w = blocks * waves / 4;
    return w < 1 ? 1 : (w > 4 ? 4 : w);
}

// Measured and rejected (profiles/r01_igemm_pipeline_variants_ab.txt, r01_igemm_two_load_stages_ab.txt):
// folding commit() into the second half's MFMAs with sched_barrier fences, and a second register stage
// of global loads (first for the 64x64 tile, later for every tile) — all within +-2 % of this simpler
// pipeline on the UNet's shapes: neither load latency nor prologue VALU is what limits the fp32-MFMA kernels
// (their 64-cycle MFMAs hide both).  The bf16-split kernels are the opposite case and use both ideas.
//
// KGRP = wave groups per block.  KGRP = 2 (512 threads): two 4-wave groups work on the SAME output tile,
// each with its own LDS double buffer, group g taking k-tiles g, g+2, ...; their accumulators are added
// through LDS before the epilogue (fixed order: deterministic).  For launches with fewer than two blocks
// per CU this puts two waves on every SIMD, so one group's LDS/barrier/address turnaround (≈1700 cycles
// per k-tile when alone, measured) hides behind the other group's MFMAs — an in-block split-K with no
// workspace and no reduce kernel.
//
// UNI (affine prologues only): every block tile lies inside ONE sample (OH*OW % BM == 0, checked by the
// host), so the per-(sample, channel) GroupNorm scale/shift of a k-group is loaded once per k-tile
// instead of once per row pass: 2 instead of 2*PA extra loads and 2 instead of 2*PA float4 registers.
template <int BM, int BN, int WM, int WN, int PRE, int KGRP, bool UNI, bool BX = false>
__global__ __launch_bounds__(64 * WM * WN * KGRP, igemm_waves_per_simd(BM, BN, WM * WN * KGRP, BX))
void igemm_kernel(const IgemmK p) {
    constexpr int MT = BM / (32 * WM);
    constexpr int NT = BN / (32 * WN);
    // WM x WN waves share the tile: 4 (one per SIMD) or 8 (two per SIMD from ONE block: the 128x128 tile's
    // LDS allows only two blocks per CU, so 8-wave blocks are how four waves per SIMD get to overlap
    // their staging / barrier phases with each other's MFMAs)
    constexpr int GT = 64 * WM * WN;   // threads of one wave group
    constexpr int RPP = GT / 8;        // rows (A) / columns (NT-mode B) gathered per loader pass
    constexpr int PA = BM / RPP;       // A-loader passes (RPP rows x 8 k-groups per pass)
    constexpr int PB = (BX ? 12 : KG) * BN / GT;   // B-loader passes
    static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves");
    static_assert(WM * WN == 4 || KGRP == 1, "8-wave tiles have one wave group");
    static_assert(BM % RPP == 0 && ((BX ? 12 : KG) * BN) % GT == 0 && BN % RPP == 0, "loader tiling");
    // one raw LDS buffer: A/B double buffers during the K loop, per-wave output staging afterwards.
    // fp32 image: LG = 8 k-groups (float4 = 4 consecutive k) per row and k-tile.  BX image: LG = 12 16-byte
    // slots per row / column and k-tile = 4 k-octets x 3 bf16 parts (8 bf16 = 8 consecutive k per slot);
    // A is indexed [part*4 + octet], B [octet*3 + part] (the order the split weights are stored in).
    constexpr int LG = BX ? 12 : KG;
    constexpr int LP = BX ? 2 : 1;     // row padding: BX writes 8-byte halves, pitch = 2 mod 16 keeps them apart
    constexpr int A_F4 = 2 * LG * (BM + LP);
    constexpr int B_F4 = 2 * LG * (BN + LP);
    __shared__ f32x4 smem[KGRP * (A_F4 + B_F4)];
    const int grp = KGRP == 1 ? 0 : (int)(threadIdx.x / GT);
    f32x4 (*As)[LG][BM + LP] = reinterpret_cast<f32x4 (*)[LG][BM + LP]>(&smem[grp * (A_F4 + B_F4)]);
    f32x4 (*Bs)[LG][BN + LP] = reinterpret_cast<f32x4 (*)[LG][BN + LP]>(&smem[grp * (A_F4 + B_F4) + A_F4]);

    const aldm_igemm_desc& d = p.d;
    const int tid = KGRP == 1 ? (int)threadIdx.x : (int)(threadIdx.x % GT);  // position inside the wave group
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

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
    const int z = blockIdx.z;
    const int split = blockIdx.y;
    const float* x1 = d.x1 + (int64_t)z * d.stride_x;
    const float* x2 = d.x2 ? d.x2 + (int64_t)z * d.stride_x : x1;
    // BX: the weights arrive pre-split into bf16 parts (aldm_pack_split_bf16), one shared matrix
    const float* wgt = BX ? reinterpret_cast<const float*>(d.w_split) : d.w + (int64_t)z * d.stride_w;

    const int nk_all = (d.K + BK - 1) / BK;
    const int kt0 = split * p.kt_per_split;
    const int kt1 = min(nk_all, kt0 + p.kt_per_split);

    // ---- A loader bookkeeping: this thread gathers rows ar0 + RPP*pp (RPP = threads/8), k-group akg ----
    const int akg = tid & 7;
    const int ar0 = tid >> 3;
    int a_pix[PA], a_h[PA], a_w[PA], a_b[PA];
#pragma unroll
    for (int pp = 0; pp < PA; ++pp) {
        const int m = m0 + ar0 + RPP * pp;
        if (m < p.M) {
            const int b = m / p.OHW;
            const int rem = m - b * p.OHW;
            const int oh = rem / d.OW;
            const int ow = rem - oh * d.OW;
            a_b[pp] = b;
            a_pix[pp] = b * d.H;
            a_h[pp] = oh * d.SH - d.PH;
            a_w[pp] = ow * d.SW - d.PW;
        } else {
            a_b[pp] = 0;
            a_pix[pp] = 0;
            a_h[pp] = -(1 << 28);
            a_w[pp] = 0;
        }
    }
    // incremental (kh, kw, ci) of this thread's k = kt*BK + 4*akg
    int t_ci, t_kh, t_kw;
    {
        const int k = (kt0 + grp) * BK + 4 * akg;  // group g starts at k-tile kt0 + g
        const int tap = k / p.Cin;
        t_ci = k - tap * p.Cin;
        t_kh = tap / d.KW;
        t_kw = tap - t_kh * d.KW;
    }
    const int pix1 = d.pix1, pix2 = d.pix2;

    // ---- B loader bookkeeping (branch-free over both layouts) ----
    // load pp of k-tile kt reads offset kt*b_kt + pp*b_pp + b_base; it is valid iff
    //   kt*b_ks + pp*b_kps + b_k0 < b_klim  and  pp*b_nps + b_n0 < b_nlim ;
    // it is stored at Bs[buf][b_skg + pp*b_skgs][b_sc + pp*b_scs].
    //   PACKED [Kg][Npad][4]: thread -> column tid % BN, k-groups tid / BN + (GT/BN)*pp
    //   NT     Bmat[N][ldb] : thread -> row tid / 8 + RPP*pp, k-group tid % 8
    //   BX     [octet][part][Npad][8 bf16]: as PACKED with 12 always-valid 16-byte slots per k-tile (the
    //          split weights are zero padded to whole k-tiles)
    constexpr int stepb = GT / BN;
    const bool packed = BX || d.b_mode == ALDM_B_PACKED;
    const int64_t b_kt = packed ? (int64_t)LG * p.Npad * 4 : BK;
    const int64_t b_pp = packed ? (int64_t)stepb * p.Npad * 4 : (int64_t)RPP * d.ldb;
    const int64_t b_base = packed ? ((int64_t)(tid / BN) * p.Npad + n0 + tid % BN) * 4
                                  : (int64_t)(n0 + ar0) * d.ldb + 4 * akg;
    const int b_ks = BX ? 0 : (packed ? KG : BK), b_kps = BX ? 0 : (packed ? stepb : 0);
    const int b_k0 = BX ? 0 : (packed ? tid / BN : 4 * akg), b_klim = BX ? 1 : (packed ? p.Kg : d.K);
    const int b_nps = packed ? 0 : RPP, b_n0 = packed ? n0 + tid % BN : n0 + ar0;
    const int b_nlim = packed ? p.Npad : d.N;
    const int b_skg = packed ? tid / BN : akg, b_skgs = packed ? stepb : 0;
    const int b_sc = packed ? tid % BN : ar0, b_scs = packed ? 0 : RPP;

    // ---- running gather state -----------------------------------------------------------------
    // The address of a gathered float4 only moves by a constant (BK*KGRP channels) from one k-tile to
    // the next while the tile stays inside one (tap, source tensor) segment; the full index arithmetic
    // (bounds checks, upsample shift, 64-bit pixel*pitch products) is redone only when the segment
    // changes.  Measured with s_memtime stamps: address generation was 800 (64x64) to 1500+ (128x128)
    // cycles of every k-tile, as much as half the tile's MFMA time when a wave is alone on its SIMD.
    constexpr bool AFF = PRE == PRE_AFFINE || PRE == PRE_AFFINE_SILU || PRE == PRE_GENERIC;
    constexpr int KSTEP = BK * KGRP;           // channels between two k-tiles of this wave group
    const float* a_ptr[PA];                    // next tile's source of row pass pp (always a safe address)
    const float* sc_ptr[(AFF && !UNI) ? PA : 1];
    const float* sh_ptr[(AFF && !UNI) ? PA : 1];
    unsigned a_okmask = 0;                     // bit pp: that source is real data (else zero padding)
    int seg_hi = 0;                            // t_ci < seg_hi <=> still inside the current segment
    auto recompute_gather = [&]() {
        const bool kval = t_kh < d.KH;  // k < K
        const bool first = t_ci < d.C1;
        const float* src = first ? x1 : x2;
        const int c = first ? t_ci : t_ci - d.C1;
        const int pitch = first ? pix1 : pix2;
        const int dh = t_kh * d.DH, dw = t_kw * d.DW;
        seg_hi = kval ? (first ? d.C1 : p.Cin) : (1 << 30);  // past K: stay "inside" (always invalid)
        a_okmask = 0;
#pragma unroll
        for (int pp = 0; pp < PA; ++pp) {
            const int ihv = a_h[pp] + dh;
            const int iwv = a_w[pp] + dw;
            const bool ok = kval && (unsigned)ihv < (unsigned)p.HV && (unsigned)iwv < (unsigned)p.WV;
            const int ih = ihv >> p.shh, iw = iwv >> p.shw;
            const int pix = ok ? (a_pix[pp] + ih) * d.W + iw : 0;
            // invalid -> pixel 0 of the source at channel c (kval) or its base (k >= K): readable memory
            a_ptr[pp] = src + (int64_t)pix * pitch + (kval ? c : 0);
            a_okmask |= (ok ? 1u : 0u) << pp;
            if constexpr (AFF && !UNI) {
                if (PRE != PRE_GENERIC || d.pre_scale != nullptr) {
                    const int64_t so = (int64_t)a_b[pp] * p.Cin + (kval ? t_ci : 0);
                    sc_ptr[pp] = d.pre_scale + so;
                    sh_ptr[pp] = d.pre_shift + so;
                }
            }
        }
        if constexpr (AFF && UNI) {
            const int64_t so = (int64_t)(m0 / p.OHW) * p.Cin + (kval ? t_ci : 0);
            sc_ptr[0] = d.pre_scale + so;
            sh_ptr[0] = d.pre_shift + so;
        }
    };
    recompute_gather();
    // B: running pointers too; validity is a compare against the running k index
    const float* b_ptr[PB];
    int b_kidx[PB];
#pragma unroll
    for (int pp = 0; pp < PB; ++pp) {
        b_ptr[pp] = wgt + ((int64_t)(kt0 + grp) * b_kt + pp * b_pp + b_base);
        b_kidx[pp] = (kt0 + grp) * b_ks + pp * b_kps + b_k0;
    }

    constexpr bool GLDS = BX && ALDM_BX_GLDS != 0;
    // one in-flight k-tile of this thread's global loads
    struct Stage {
        f32x4 ra[PA], rb[GLDS ? 1 : PB];
        f32x4 rsc[(AFF && !UNI) ? PA : 1], rsh[(AFF && !UNI) ? PA : 1];
        unsigned avalid, bvalid;
    };

    // loads the NEXT k-tile of this wave group (the gather state points at it), then advances the state
    auto issue_loads = [&](Stage& r) {
        r.avalid = a_okmask;
#pragma unroll
        for (int pp = 0; pp < PA; ++pp) r.ra[pp] = *reinterpret_cast<const f32x4*>(a_ptr[pp]);
        if constexpr (AFF) {
            if (PRE != PRE_GENERIC || d.pre_scale != nullptr) {
#pragma unroll
                for (int pp = 0; pp < ((AFF && !UNI) ? PA : 1); ++pp) {
                    r.rsc[pp] = *reinterpret_cast<const f32x4*>(sc_ptr[pp]);
                    r.rsh[pp] = *reinterpret_cast<const f32x4*>(sh_ptr[pp]);
                }
            }
        }
        r.bvalid = 0;
        if constexpr (!GLDS) {
#pragma unroll
            for (int pp = 0; pp < PB; ++pp) {
                const bool ok = b_kidx[pp] < b_klim && pp * b_nps + b_n0 < b_nlim;
                r.rb[pp] = *reinterpret_cast<const f32x4*>(ok ? b_ptr[pp] : wgt);
                r.bvalid |= (ok ? 1u : 0u) << pp;
                b_ptr[pp] += b_kt * KGRP;
                b_kidx[pp] += b_ks * KGRP;
            }
        }
        // advance A to this group's next k-tile
        t_ci += KSTEP;
        if (t_ci < seg_hi) {
#pragma unroll
            for (int pp = 0; pp < PA; ++pp) a_ptr[pp] += KSTEP;
            if constexpr (AFF) {
                if (PRE != PRE_GENERIC || d.pre_scale != nullptr) {
#pragma unroll
                    for (int pp = 0; pp < ((AFF && !UNI) ? PA : 1); ++pp) {
                        sc_ptr[pp] += KSTEP;
                        sh_ptr[pp] += KSTEP;
                    }
                }
            }
        } else {  // next tile starts in another tap / source tensor (or past K): full index arithmetic
            while (t_ci >= p.Cin) {
                t_ci -= p.Cin;
                if (++t_kw == d.KW) {
                    t_kw = 0;
                    ++t_kh;
                }
            }
            recompute_gather();
        }
    };

    // ---- commit = operand prologue on the loaded registers + LDS stores ----
    constexpr bool ELEMWISE = PRE == PRE_AFFINE || PRE == PRE_AFFINE_SILU || PRE == PRE_LRELU;
    auto xform_elem = [&](Stage& r, int pp, int c) {  // one component, in place
        float v = r.ra[pp][c];
        constexpr int si = UNI ? 0 : 1;  // scale/shift register index = pp * si
        if constexpr (PRE == PRE_AFFINE) {
            v = v * r.rsc[pp * si][c] + r.rsh[pp * si][c];
        } else if constexpr (PRE == PRE_AFFINE_SILU) {
            v = silu_fast(v * r.rsc[pp * si][c] + r.rsh[pp * si][c]);
        } else if constexpr (PRE == PRE_LRELU) {
            v = v > 0.0f ? v : v * d.pre_slope;
        }
        r.ra[pp][c] = v;
    };
    auto store_a = [&](Stage& r, int buf, int pp) {
        f32x4 v = r.ra[pp];
        if constexpr (PRE == PRE_GENERIC) {
            if (d.pre_scale != nullptr) v = v * r.rsc[pp] + r.rsh[pp];
            if (d.pre_act != ALDM_ACT_NONE) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = act_apply(v[j], d.pre_act, d.pre_slope);
            }
        }
        if (!((r.avalid >> pp) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};  // zero padding stays zero
        if constexpr (BX) {
            // exact 3-way split x = hi + mid + lo, each part the top 16 bits of an fp32 (8 significant bits,
            // truncation: the parts cover x's 24-bit significand exactly); two parts packed per dword
            unsigned u0[4], u1[4], u2[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float x = v[c];  // (bit-casting the vector element directly reads element 0 on this compiler)
                u0[c] = __builtin_bit_cast(unsigned, x);
                const float r1 = x - __builtin_bit_cast(float, u0[c] & 0xFFFF0000u);
                u1[c] = __builtin_bit_cast(unsigned, r1);
                const float r2 = r1 - __builtin_bit_cast(float, u1[c] & 0xFFFF0000u);
                u2[c] = __builtin_bit_cast(unsigned, r2);
            }
#if ALDM_ABLATE & 4
#pragma unroll
            for (int c = 0; c < 4; ++c) u1[c] = u2[c] = u0[c];
#endif
            const int o = akg >> 1, hf = akg & 1, row = ar0 + RPP * pp;
#if ALDM_ABLATE & 8
            asm volatile("" ::"v"(u0[0]), "v"(u0[1]), "v"(u0[2]), "v"(u0[3]), "v"(u1[0]), "v"(u1[1]), "v"(u1[2]), "v"(u1[3]));
            asm volatile("" ::"v"(u2[0]), "v"(u2[1]), "v"(u2[2]), "v"(u2[3]));
            return;
#endif
            reinterpret_cast<u32x2*>(&As[buf][0 + o][row])[hf] = u32x2{hi16_pair(u0[0], u0[1]), hi16_pair(u0[2], u0[3])};
            reinterpret_cast<u32x2*>(&As[buf][4 + o][row])[hf] = u32x2{hi16_pair(u1[0], u1[1]), hi16_pair(u1[2], u1[3])};
            reinterpret_cast<u32x2*>(&As[buf][8 + o][row])[hf] = u32x2{hi16_pair(u2[0], u2[1]), hi16_pair(u2[2], u2[3])};
        } else {
            As[buf][akg][ar0 + RPP * pp] = v;
        }
    };
    auto store_b = [&](Stage& r, int buf, int pp) {
        f32x4 v = r.rb[pp];
        if (!((r.bvalid >> pp) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
#if ALDM_ABLATE & 16
        asm volatile("" ::"v"(v));
        return;
#endif
        Bs[buf][b_skg + pp * b_skgs][b_sc + pp * b_scs] = v;
    };
    // GLDS: the next B k-tile of this wave group, global -> LDS.  A wave's 64 lanes cover 64 consecutive columns of
    // one 16-byte slot row, i.e. 1 KB contiguous in LDS = what one global_load_lds_dwordx4 writes (M0 = the wave's
    // slot base, lane i lands at +16*i).  Columns past Npad read a valid dummy address; they only feed output
    // columns >= N, which are never stored.
    auto dma_b = [&](int buf) {
        if constexpr (GLDS) {
#pragma unroll
            for (int pp = 0; pp < PB; ++pp) {
                const bool ok = b_n0 < b_nlim;
                const int slot = __builtin_amdgcn_readfirstlane(b_skg + pp * b_skgs);
                const int col0 = __builtin_amdgcn_readfirstlane(b_sc & ~63);
                __builtin_amdgcn_global_load_lds(ok ? b_ptr[pp] : wgt,
                                                 (__attribute__((address_space(3))) void*)&Bs[buf][slot][col0], 16, 0, 0);
                b_ptr[pp] += b_kt * KGRP;
            }
        }
    };
    auto commit = [&](Stage& r, int buf) {
#pragma unroll
        for (int pp = 0; pp < PA; ++pp) {
            if constexpr (ELEMWISE) {
#pragma unroll
                for (int c = 0; c < 4; ++c) xform_elem(r, pp, c);
            }
            store_a(r, buf, pp);
        }
        if constexpr (!GLDS) {
#pragma unroll
            for (int pp = 0; pp < PB; ++pp) store_b(r, buf, pp);
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int l31 = lane & 31;
    const int lh = lane >> 5;

    // BX: fragments of one 16-wide k-step (lane half lh owns k-octet 2*step + lh of both operands, the same
    // 8 k on both sides) and their product: fp32 a*b = 6 bf16 partial products, smallest first, accumulated
    // in fp32 by the MFMA.  Reading and multiplying are separate so the K loop can keep one step of
    // fragments in flight behind the other step's MFMAs.
    struct Frag {
        bf16x8 a[MT][3], b[NT][3];
    };
    auto read_frags = [&](Frag& f, int buf, int step) {
        if constexpr (!BX) return;
        const int o = 2 * step + lh;
#if !(ALDM_ABLATE & 2)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int q = 0; q < 3; ++q)
                f.a[i][q] = __builtin_bit_cast(bf16x8, As[buf][q * 4 + o][(wm * MT + i) * 32 + l31]);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int q = 0; q < 3; ++q)
                f.b[j][q] = __builtin_bit_cast(bf16x8, Bs[buf][o * 3 + q][(wn * NT + j) * 32 + l31]);
#endif
    };
    auto mma_frags = [&](Frag& f) {
        if constexpr (!BX) return;
#if ALDM_ABLATE & 2
        return;
#elif ALDM_ABLATE & 1
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int q = 0; q < 3; ++q) asm volatile("" ::"v"(f.a[i][q]));
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int q = 0; q < 3; ++q) asm volatile("" ::"v"(f.b[j][q]));
        return;
#else
        constexpr int PA_[6] = {0, 2, 1, 0, 1, 0}, PB_[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][PA_[q]], f.b[j][PB_[q]], acc[i][j],
                                                                        0, 0, 0);
#endif
    };

    // fp32 MFMA: one half of a k-tile = 2 sub-steps of 8 k: one ds_read_b128 per fragment, 4 MFMAs per
    // fragment pair and tile
    auto mma_half = [&](int buf, int half) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int kg = 2 * (2 * half + s2) + lh;
            f32x4 af[MT], bf[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = As[buf][kg][(wm * MT + i) * 32 + l31];
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[j] = Bs[buf][kg][(wn * NT + j) * 32 + l31];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e],
                                                                          acc[i][j], 0, 0, 0);
        }
    };
    if (kt0 < kt1) {
        // group g multiplies k-tiles kt0 + g + it*KGRP; every group runs the same number of iterations
        // (and barriers), an iteration without a tile only synchronises
        const int n_it = (kt1 - kt0 + KGRP - 1) / KGRP;
        auto tile_of = [&](int it) { return kt0 + grp + it * KGRP; };
        if constexpr (!BX) {
            Stage r0;
            if (tile_of(0) < kt1) {
                issue_loads(r0);
                commit(r0, 0);
            }
            __syncthreads();
            int buf = 0;
            for (int it = 0; it < n_it; ++it) {
                const bool cur = tile_of(it) < kt1, nxt = tile_of(it + 1) < kt1;
                if (nxt) issue_loads(r0);
                if (cur) mma_half(buf, 0);
                __builtin_amdgcn_sched_barrier(0);  // keep the loads' first use behind half the MFMAs
                if (nxt) commit(r0, buf ^ 1);
                if (cur) mma_half(buf, 1);
                if (it + 1 < n_it) __syncthreads();
                buf ^= 1;
            }
        } else {
            // BX: a k-tile's MFMAs last 768 (8-wave tile) .. 1536 cycles - less than a global load's latency and
            // about as long as the tile's LDS traffic - so everything is software pipelined inside the wave:
            //  * global loads run TWO k-tiles ahead in two register stages (iteration `it` issues tile it+2,
            //    commits tile it+1, multiplies tile it);
            //  * the fragments of k-step 1 are read while k-step 0's MFMAs run, and - after the barrier that
            //    publishes the next buffer - the next tile's k-step 0 fragments while k-step 1's MFMAs run.
            // `steady` iterations (tiles it .. it+2 exist for every wave group) are branch-free: a conditional
            // issue_loads makes the compiler's s_waitcnt insertion assume the committed stage holds the
            // newest loads and drain the whole queue (vmcnt(0)), i.e. no prefetch at all.
            Stage r0, r1;
            Frag f0, f1;
            if (tile_of(0) < kt1) dma_b(0);
            if (tile_of(0) < kt1) issue_loads(r0);
            if (tile_of(1) < kt1) issue_loads(r1);
            if (tile_of(0) < kt1) commit(r0, 0);
            __syncthreads();
            if (tile_of(0) < kt1) read_frags(f0, 0, 0);
            auto body = [&](Stage& ld, Stage& cm, int it, int buf, auto steady) {
                constexpr bool ST = decltype(steady)::value;
                const bool cur = ST || tile_of(it) < kt1, nxt = ST || tile_of(it + 1) < kt1;
                const bool nx2 = ST || tile_of(it + 2) < kt1;
                if (nxt) dma_b(buf ^ 1);  // (GLDS only) B of tile it+1: `buf ^ 1` is free since the last barrier
                if (nx2) issue_loads(ld);
                __builtin_amdgcn_sched_barrier(0);
                if (cur) read_frags(f1, buf, 1);
                if (cur) mma_frags(f0);
                if (nxt) commit(cm, buf ^ 1);
                if constexpr (ST && ALDM_BX_INTERLEAVE) {
                    // one wave's MFMA stream leaves ~28 of every 32 cycles of its issue slot free: spread the
                    // k-step-1 fragment reads and the commit (prologue + split VALU, LDS stores) between the
                    // MFMAs instead of running them as separate phases
                    constexpr int NMF = 6 * MT * NT, NRD = 3 * (MT + NT);
                    constexpr int NVA = PA * (PRE == PRE_AFFINE_SILU ? 84 : (PRE == PRE_NONE ? 32 : 44)) + 8;
                    constexpr int VPM = (NVA + NMF - 1) / NMF, NWR = 3 * PA + PB;
#pragma unroll
                    for (int q = 0; q < NMF; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (q < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
                        if (q >= NMF - NWR) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (ST || it + 1 < n_it) __syncthreads();  // everyone has read `buf` and written `buf ^ 1`
                if (nxt) read_frags(f0, buf ^ 1, 0);
                if (cur) mma_frags(f1);
                if constexpr (ST && ALDM_BX_INTERLEAVE) {
                    constexpr int NMF = 6 * MT * NT, NRD = 3 * (MT + NT);
#pragma unroll
                    for (int q = 0; q < NMF; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                        if (q < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            const int n_full = (kt1 - kt0) / KGRP;  // iterations in which every wave group has a tile
            int it = 0;
            for (; it + 3 < n_full; it += 2) {
                body(r0, r1, it, 0, std::true_type{});
                body(r1, r0, it + 1, 1, std::true_type{});
            }
            for (; it < n_it; it += 2) {  // `it` is even here: same stage roles as in the steady loop
                body(r0, r1, it, 0, std::false_type{});
                if (it + 1 < n_it) body(r1, r0, it + 1, 1, std::false_type{});
            }
        }
    }

    // ---- epilogue (igemm_epilogue.h): each wave transposes its slab through a private LDS region -----
    constexpr int SP = NT * 32 + 4;   // staging row pitch (floats); +4 keeps 16-byte alignment
    static_assert(WM * WN * 32 * SP * 4 <= (A_F4 + B_F4) * 16, "staging must fit the K-loop LDS");
    __syncthreads();  // every wave is done reading As/Bs
    if constexpr (KGRP == 2) {
        // add the two groups' accumulators: group 1 parks its registers in LDS (lane-linear, conflict
        // free, behind the transpose staging area), group 0 adds them and runs the epilogue alone
        static_assert(4 * 32 * SP * 4 + 4 * MT * NT * 16 * 64 * 4 <= KGRP * (A_F4 + B_F4) * 16, "reduction area");
        float* red = reinterpret_cast<float*>(&smem[0]) + 4 * 32 * SP + wave * (MT * NT * 16 * 64) + lane;
        if (grp == 1) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) red[((i * NT + j) * 16 + e) * 64] = acc[i][j][e];
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] += red[((i * NT + j) * 16 + e) * 64];
    }
    igemm_epilogue<MT, NT>(p, acc, reinterpret_cast<float*>(&smem[0]), m0, n0, wave, wm, wn, lane, z, split);
}

}  // namespace aldm
