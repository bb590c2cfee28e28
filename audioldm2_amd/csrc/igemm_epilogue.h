// igemm_epilogue.h — what the two implicit-GEMM K loops (igemm_kernel.h: register-staged fp32 / bf16-split;
// igemm_dma.h: DMA-fed bf16-split over pre-split operands) share: the launch parameter block, the operand split,
// and the fused epilogue
//   v = act(acc + bias[n] + rowbias[b, n]); v = alpha*(v + res[m, n]); out = accumulate ? out + v : v
// with its optional GEGLU form and its optional second output, the bf16-split image of the result (the A operand
// of the next DMA-fed GEMM, see igemm_dma.h).
#pragma once
#include "common.h"

#ifndef ALDM_DMA_ABLATE
#define ALDM_DMA_ABLATE 0  // debug builds only, see igemm_dma.h; bit 64: the epilogue computes but never stores
#endif

namespace aldm {

struct IgemmK {
    aldm_igemm_desc d;
    int Cin, M, OHW, HV, WV, shh, shw, Kg, Npad, tiles_m, tiles_n;
    int splits, kt_per_split;  // split-K: k-tiles [s*kt_per_split, ...) per blockIdx.y
    int kgroups;               // wave groups per block (1 or 2, see igemm_kernel)
    int rb_ld;                 // row-bias pitch
    int bx;                    // 1: bf16-split kernels (d.w_split), 0: fp32 MFMA
    int pre;                   // PRE_* prologue mode of the descriptor
    int dma;                   // 1: DMA-fed kernel over a pre-split A image (d.a_split)
    int nst;                   // DMA kernel: LDS ring depth of the chosen instantiation
    int ws;                    // 1: the persistent wave-specialised DMA kernel (igemm_dma_ws.h), ws_blocks blocks; 2: loader waves
                               // (igemm_dma_lw.h); 3: the operand-stationary kernel (igemm_dma_os.h); 4: the halo-patch kernel
                               // (igemm_dma_halo.h: ws_blocks = WM, os_rows = MAXCH of the instantiation)
    int ws_blocks;
    int os_rows;               // operand-stationary kernel: rows per block (a multiple of 32)
};

enum { PRE_NONE = 0, PRE_AFFINE = 1, PRE_AFFINE_SILU = 2, PRE_LRELU = 3, PRE_GENERIC = 4 };

using bf16x8 = __bf16 __attribute__((ext_vector_type(8)));
using u32x2 = unsigned __attribute__((ext_vector_type(2)));
using u32x4 = unsigned __attribute__((ext_vector_type(4)));
// the upper halves of two dwords as one dword (lo half from a): two truncated bf16 side by side
__device__ __forceinline__ unsigned hi16_pair(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// ---- the split image --------------------------------------------------------------------------------------------
// A "split image" of a channels-last fp32 tensor [rows, C] (C % 32 == 0) stores every value as its exact 3-way
// truncation split x = hi + mid + lo (each part the top 16 bits of an fp32, i.e. a bf16 bit pattern), blocked so
// that one 32-channel block of one row is 192 contiguous bytes:
//     img[row][C/32][part 0..2][32]  bf16          (SPLIT_BLOCK_BYTES per (row, 32-channel block))
// One 16-byte piece = 8 consecutive channels of one part = one MFMA operand fragment of one lane.
// A 2-part image ("bf16x3" mode) keeps hi + mid only, both ROUNDED TO NEAREST (hi = RN_bf16(x), mid = RN_bf16(x - hi)):
// |x - hi - mid| <= 2^-16 |x| (2^-18 rms), unbiased; 128 bytes per block.
constexpr int SPLIT_BLOCK_BYTES = 192;   // 3-part image; a P-part image has 64 * P

__device__ __forceinline__ unsigned bf16_rn_bits(float x) {   // bf16 bit pattern (in the upper half), round to nearest even
    const unsigned u = __builtin_bit_cast(unsigned, x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
}

// exact 3-way split of 4 values -> per part two dwords (4 bf16)
__device__ __forceinline__ void split4(const f32x4 v, u32x2 (&part)[3]) {
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
    part[0] = u32x2{hi16_pair(u0[0], u0[1]), hi16_pair(u0[2], u0[3])};
    part[1] = u32x2{hi16_pair(u1[0], u1[1]), hi16_pair(u1[2], u1[3])};
    part[2] = u32x2{hi16_pair(u2[0], u2[1]), hi16_pair(u2[2], u2[3])};
}

// 2-part split (hi, mid), round to nearest even: gfx950 converts two floats per v_cvt_pk_bf16_f32, already packed — 2.5 VALU
// per element against ~10 for the integer emulation (bf16_rn_bits + repacking)
using bf16x4 = __bf16 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split4_rn2(const f32x4 v, u32x2 (&part)[3]) {
    const bf16x4 h = __builtin_convertvector(v, bf16x4);
    const bf16x4 m = __builtin_convertvector(v - __builtin_convertvector(h, f32x4), bf16x4);
    part[0] = __builtin_bit_cast(u32x2, h);
    part[1] = __builtin_bit_cast(u32x2, m);
    part[2] = u32x2{0u, 0u};
}

__device__ __forceinline__ void split4_parts(const f32x4 v, u32x2 (&part)[3], int parts) {
    if (parts == 2) split4_rn2(v, part);
    else split4(v, part);
}

// "f16x3" operands (aldm_igemm_desc.a_fmt = ALDM_FMT_F16): 2 parts, IEEE fp16, of the value scaled by an exact power of two:
// hi = RN_f16(s x), lo = RN_f16(s x - hi) — 22 of the 24 significand bits while lo is a normal fp16, and an absolute error of at
// most 2^-25 / s below that (the matrix core honours fp16 subnormals: profiles/r06_f16x3_accuracy.txt).  The producer's s puts
// |s x| under 65504 by construction (a GroupNorm / LayerNorm output cannot exceed sqrt(n) max|gamma| + max|beta|); the clamp is a
// seat belt that never engages inside that bound.
using f16x4 = _Float16 __attribute__((ext_vector_type(4)));
using f16x8 = _Float16 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split4_f16(const f32x4 v, float s, u32x2 (&part)[3]) {
    f32x4 x;
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = __builtin_fminf(__builtin_fmaxf(v[e] * s, -65504.0f), 65504.0f);
    const f16x4 h = __builtin_convertvector(x, f16x4);
    const f16x4 l = __builtin_convertvector(x - __builtin_convertvector(h, f32x4), f16x4);
    part[0] = __builtin_bit_cast(u32x2, h);
    part[1] = __builtin_bit_cast(u32x2, l);
    part[2] = u32x2{0u, 0u};
}
// the producers' store: f16_scale != 0 -> the 2-part fp16 image of f16_scale * v, else the `parts`-part bf16 image of v
__device__ __forceinline__ void split4_fmt(const f32x4 v, u32x2 (&part)[3], int parts, float f16_scale) {
    if (f16_scale != 0.f) split4_f16(v, f16_scale, part);
    else split4_parts(v, part, parts);
}

// channels c .. c+3 (c % 4 == 0) of row `row` of a split image with Cs channels per row and `parts` parts
__device__ __forceinline__ void split_store4(void* img, int64_t row, int Cs, int c, const f32x4 v, int parts = 3) {
    u32x2 part[3];
    split4_parts(v, part, parts);
    char* base = reinterpret_cast<char*>(img) + (row * (Cs >> 5) + (c >> 5)) * (64 * parts) + (c & 31) * 2;
#pragma unroll
    for (int q = 0; q < 3; ++q)
        if (q < parts) *reinterpret_cast<u32x2*>(base + q * 64) = part[q];
}

// channels c .. c+3 of row `row` of a 2-part fp16 ("f16x3") image of scale * v
__device__ __forceinline__ void split_store4_f16(void* img, int64_t row, int Cs, int c, const f32x4 v, float scale) {
    u32x2 part[3];
    split4_f16(v, scale, part);
    char* base = reinterpret_cast<char*>(img) + (row * (Cs >> 5) + (c >> 5)) * 128 + (c & 31) * 2;
    *reinterpret_cast<u32x2*>(base) = part[0];
    *reinterpret_cast<u32x2*>(base + 64) = part[1];
}
// the epilogues' image store: the descriptor's output format (bf16 with out_split_parts parts, or fp16 with out_split_scale)
__device__ __forceinline__ void split_store4_out(const aldm_igemm_desc& d, void* img, int64_t row, int Cs, int c, const f32x4 v) {
    if (d.out_split_fmt == ALDM_FMT_F16) split_store4_f16(img, row, Cs, c, v, d.out_split_scale);
    else split_store4(img, row, Cs, c, v, d.out_split_parts);
}

// ... with the image format known at compile time: no `q < parts` branches around the stores, one split form compiled
template <int P>
__device__ __forceinline__ void split_store4_t(void* img, int64_t row, int Cs, int c, const f32x4 v) {
    static_assert(P == 2 || P == 3, "2 or 3 parts");
    u32x2 part[3];
    if constexpr (P == 2) split4_rn2(v, part);
    else split4(v, part);
    char* base = reinterpret_cast<char*>(img) + (row * (Cs >> 5) + (c >> 5)) * (64 * P) + (c & 31) * 2;
#pragma unroll
    for (int q = 0; q < P; ++q) *reinterpret_cast<u32x2*>(base + q * 64) = part[q];
}

// one output element through the fused epilogue (split-K reduce and ragged-N fallbacks)
__device__ __forceinline__ float epi_value(const aldm_igemm_desc& d, int rb_ld, const float* __restrict__ outp,
                                           const float* __restrict__ resp, int b, int64_t o, int n, float v) {
    if (d.bias) v += d.bias[n];
    if (d.rowbias) v += d.rowbias[(int64_t)b * rb_ld + n];
    v = act_apply(v, d.act, d.act_slope);
    if (resp) v += resp[o];
    v *= d.alpha;
    if (d.accumulate) v += outp[o];
    return v;
}

__device__ __forceinline__ float silu_fast(float v) {
    // x * sigmoid(x); v_exp_f32 / v_rcp_f32 are <= 1 ulp each: ~1e-7 relative, far inside the
    // parity tolerance, and 5 VALU ops instead of the ~25 of expf + IEEE division.
    return v * __frcp_rn(1.0f + __expf(-v));
}

// (hi, mid) round-to-nearest split of 8 values -> two bf16x8 (the attention kernel's split8_rn2), or the exact 3-part split
__device__ __forceinline__ void split8_to_parts(const float (&x)[8], u32x4 (&part)[3], int parts) {
    f32x4 a = {x[0], x[1], x[2], x[3]}, b = {x[4], x[5], x[6], x[7]};
    u32x2 pa[3], pb[3];
    split4_parts(a, pa, parts);
    split4_parts(b, pb, parts);
#pragma unroll
    for (int q = 0; q < 3; ++q) part[q] = u32x4{pa[q][0], pa[q][1], pb[q][0], pb[q][1]};
}

// ---- epilogue ------------------------------------------------------------------------------------------------------
// The MFMA accumulator layout gives a lane 4-byte pieces of 16 different rows; storing those
// directly is store-issue bound (one dword store instruction per element).  Instead each wave
// transposes its 32 x (NT*32) slab through its private LDS region and every lane then owns
// float4s along N: bias / residual / previous-output loads and the stores are 16 bytes wide,
// all optional operands are fetched with unconditional loads from clamped addresses.
//   v = act(acc + bias + rowbias); v = alpha*(v + res); out = accumulate ? out + v : v
// `lds` = the block's LDS (free: every wave is past its last read of the K-loop image); wave w stages in
// lds[w * 32 * (NT*32+4) ...].  wm / wn = this wave's position in the block's wave grid.
// PAD: extra floats per staged row (default 4: the float4 read-back is conflict free and the two half waves' writes hit
// different banks; 0 = unpadded, 4 KB per wave at NT = 1 — the half waves' ds_write_b32 then conflict 2-way, which costs a
// ds_write_b32 nothing (MI355X_MICROARCH.md, LDS) — for the operand-stationary kernel, whose ring leaves exactly 16 KB).
template <int MT, int NT, int PAD = 4>
__device__ __forceinline__ void igemm_epilogue(const IgemmK& p, f32x16 (&acc)[MT][NT], float* lds, int m0, int n0,
                                               int wave, int wm, int wn, int lane, int z, int split) {
    const aldm_igemm_desc& d = p.d;
    const bool epi_st = (ALDM_DMA_ABLATE & 64) ? p.M == -12345 : true;   // (ablation builds: compute, never store)
    constexpr int SP = NT * 32 + PAD;   // staging row pitch (floats); a multiple of 4 keeps 16-byte alignment
    constexpr int C4 = NT * 8;        // float4 per staged row
    constexpr int RPI = 64 / C4;      // rows covered by one wave-wide float4 read
    constexpr int IT = 32 / RPI;      // reads per 32-row slab
    constexpr int ITC = IT < 4 ? IT : 4;  // ... processed ITC at a time
    const int l31 = lane & 31;
    const int lh = lane >> 5;
    float* stg = lds + wave * (32 * SP);
    const int sr = lane / C4;          // row within an RPI group
    const int sc = (lane % C4) * 4;    // column within the wave's slab
    const int ncol = n0 + wn * NT * 32 + sc;
    const bool split_out = p.splits > 1;
    const bool vec = split_out || ((d.ldo & 3) == 0 && (d.N & 3) == 0 &&
                                   ((reinterpret_cast<uintptr_t>(d.out) | reinterpret_cast<uintptr_t>(d.res)) & 15) == 0 &&
                                   ((d.stride_o & 3) == 0));
    float* outp = split_out ? d.ws + ((int64_t)z * p.splits + split) * (int64_t)p.M * d.N
                            : (d.out ? d.out + (int64_t)z * d.stride_o : nullptr);
    const float* resp = (!split_out && d.res) ? d.res + (int64_t)z * d.stride_o : nullptr;
    const bool need_b = !split_out && (d.rowbias != nullptr || d.out_mul > 0);
    const int ld_out = split_out ? d.N : d.ldo;
    void* simg = split_out ? nullptr : d.out_split;
    int col_shift = 0;        // ALDM_EPI_QKV: the k columns are stored from column 0 of their image
    int simg_c = d.out_split_c;
    if (d.epi_mode == ALDM_EPI_QKV) {
        // the block's columns lie in ONE of the three C-wide segments (host: the tile width divides C)
        const int seg = n0 / d.qkv_c;
        if (seg == 2) {
            // v: transposed per (sample, head, 32-key tile), straight from the accumulator layout.  Lane (column l31 of
            // 32-column tile j, half lh) holds rows (e & 3) + 8 (e >> 2) + 4 lh of a 32-row slab in registers e = 0..15:
            // registers 8s .. 8s + 7 are the 8 keys of chunk 2s + lh of its dim's 64-byte row.
            char* vt = reinterpret_cast<char*>(d.vt_split);
            const int heads = d.qkv_c >> 5, parts = d.out_split_parts;
            const int tiles = d.qkv_rows >> 5;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int mb = m0 + (wm * MT + i) * 32;
                if (mb >= p.M) continue;
                const int b = mb / d.qkv_rows, t = (mb - b * d.qkv_rows) >> 5;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int h = (n0 - 2 * d.qkv_c + (wn * NT + j) * 32) >> 5;
                    char* base = vt + ((((int64_t)b * heads + h) * tiles + t) * parts) * 2048 + l31 * 64 + lh * 16;
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        float x8[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) x8[e] = acc[i][j][8 * s + e];
                        u32x4 part[3];
                        if (d.out_split_fmt == ALDM_FMT_F16) {
                            f32x4 a = {x8[0], x8[1], x8[2], x8[3]}, bq = {x8[4], x8[5], x8[6], x8[7]};
                            u32x2 pa[3], pb[3];
                            split4_f16(a, d.vt_scale, pa);
                            split4_f16(bq, d.vt_scale, pb);
#pragma unroll
                            for (int q = 0; q < 3; ++q) part[q] = u32x4{pa[q][0], pa[q][1], pb[q][0], pb[q][1]};
                        } else {
                            split8_to_parts(x8, part, parts);
                        }
#pragma unroll
                        for (int q = 0; q < 3; ++q)
                            if (q < parts && epi_st) *reinterpret_cast<u32x4*>(base + q * 2048 + s * 32) = part[q];
                    }
                }
            }
            return;
        }
        if (seg == 1) {        // k: the split image only
            outp = nullptr;
            simg = d.k_split;
            simg_c = d.qkv_c;
            col_shift = d.qkv_c;
        } else {               // q: fp32 only
            simg = nullptr;
        }
    }
    if constexpr (NT == 2) {
        if (d.epi_mode == ALDM_EPI_GEGLU) {
            // fused GEGLU (attention.py:42-44): the wave's slab holds 32 value columns then their 32
            // gate columns; 8 lanes cover a row's 32 outputs, one wave-wide read covers 8 rows.
            const int gr = lane >> 3, gc = (lane & 7) * 4;
            const int ncol_p = n0 + wn * 64 + gc;              // packed column of the value quad
            const int ncol_o = ((n0 + wn * 64) >> 1) + gc;     // output column
            const bool cok = ncol_p < d.N;  // N % 64 == 0: a wave's 64-column slab is all in or all out
            f32x4 bv = {0.f, 0.f, 0.f, 0.f}, bg = {0.f, 0.f, 0.f, 0.f};
            if (d.bias && cok) {
                bv = *reinterpret_cast<const f32x4*>(d.bias + ncol_p);
                bg = *reinterpret_cast<const f32x4*>(d.bias + ncol_p + 32);
            }
            float* go = d.out ? d.out + (int64_t)z * d.stride_o : nullptr;
            // gate activation: erf GELU (attention.py:44) unless the descriptor asks for the tanh form (T5 gated-gelu FF)
            const int gate_act = d.act == ALDM_ACT_GELU_TANH ? ALDM_ACT_GELU_TANH : ALDM_ACT_GELU;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        stg[((e & 3) + 8 * (e >> 2) + 4 * lh) * SP + j * 32 + l31] = acc[i][j][e];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int r = it * 8 + gr;
                    f32x4 xv = *reinterpret_cast<const f32x4*>(&stg[r * SP + gc]) + bv;
                    const f32x4 xg = *reinterpret_cast<const f32x4*>(&stg[r * SP + 32 + gc]) + bg;
#pragma unroll
                    for (int c = 0; c < 4; ++c) xv[c] *= act_apply(xg[c], gate_act, 0.f);
                    const int m = m0 + (wm * MT + i) * 32 + r;
                    if (m < p.M && cok && epi_st) {
                        if (go) *reinterpret_cast<f32x4*>(go + (int64_t)m * d.ldo + ncol_o) = xv;
                        if (simg) split_store4_out(d, simg, m, d.out_split_c, ncol_o, xv);
                    }
                }
            }
            return;
        }
    }
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (!split_out && d.bias) {
#pragma unroll
        for (int c = 0; c < 4; ++c) bias4[c] = d.bias[min(ncol + c, d.N - 1)];
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        // registers -> LDS (wave private, conflict free: 32 consecutive columns per half wave)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                stg[((e & 3) + 8 * (e >> 2) + 4 * lh) * SP + j * 32 + l31] = acc[i][j][e];
        // LDS -> float4 per lane, ITC wave-wide reads at a time (bounds the live registers)
#pragma unroll
        for (int itc = 0; itc < IT; itc += ITC) {
        f32x4 v[ITC];
        int64_t rowoff[ITC];
        int64_t srow[ITC];
        int rboff[ITC];
        unsigned okmask = 0;
#pragma unroll
        for (int it = 0; it < ITC; ++it) {
            const int r = (itc + it) * RPI + sr;
            v[it] = *reinterpret_cast<const f32x4*>(&stg[r * SP + sc]);
            const int m = m0 + (wm * MT + i) * 32 + r;
            bool ok = m < p.M && ncol < d.N;
            int b = 0;
            int64_t orow = m;
            if (need_b) {
                b = m / p.OHW;
                if (d.out_mul > 0) {
                    const int qq = m - b * p.OHW;
                    const int t = qq * d.out_mul + d.out_off;
                    ok = ok && (unsigned)t < (unsigned)d.out_len;
                    orow = (int64_t)b * d.out_len + t;
                }
            }
            rowoff[it] = ok ? orow * ld_out + (ncol - col_shift) : 0;
            srow[it] = orow;
            rboff[it] = ok ? b * p.rb_ld + ncol : 0;
            okmask |= ((ok && epi_st) ? 1u : 0u) << it;
        }
        if (split_out) {
#pragma unroll
            for (int it = 0; it < ITC; ++it)
                if ((okmask >> it) & 1u) *reinterpret_cast<f32x4*>(outp + rowoff[it]) = v[it];
            continue;
        }
#pragma unroll
        for (int it = 0; it < ITC; ++it) v[it] += bias4;
        if (d.rowbias) {
#pragma unroll
            for (int it = 0; it < ITC; ++it)
#pragma unroll
                for (int c = 0; c < 4; ++c) v[it][c] += d.rowbias[rboff[it] + (ncol + c < d.N ? c : 0)];
        }
        switch (d.act) {
            case ALDM_ACT_NONE: break;
            case ALDM_ACT_SILU:
#pragma unroll
                for (int it = 0; it < ITC; ++it)
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[it][c] = act_apply(v[it][c], ALDM_ACT_SILU, 0.f);
                break;
            case ALDM_ACT_GELU:
#pragma unroll
                for (int it = 0; it < ITC; ++it)
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[it][c] = act_apply(v[it][c], ALDM_ACT_GELU, 0.f);
                break;
            default:
#pragma unroll
                for (int it = 0; it < ITC; ++it)
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[it][c] = act_apply(v[it][c], d.act, d.act_slope);
                break;
        }
        if (vec) {
            if (resp) {
#pragma unroll
                for (int it = 0; it < ITC; ++it) v[it] += *reinterpret_cast<const f32x4*>(resp + rowoff[it]);
            }
#pragma unroll
            for (int it = 0; it < ITC; ++it) v[it] *= d.alpha;
            if (d.accumulate) {
#pragma unroll
                for (int it = 0; it < ITC; ++it) v[it] += *reinterpret_cast<const f32x4*>(outp + rowoff[it]);
            }
            if (outp) {
#pragma unroll
                for (int it = 0; it < ITC; ++it)
                    if ((okmask >> it) & 1u) *reinterpret_cast<f32x4*>(outp + rowoff[it]) = v[it];
            }
            if (simg) {  // second output: the result as a split image (host checks N % 4 == 0, vec)
                if (d.out_split_act == ALDM_ACT_LRELU) {   // ... of leaky_relu(result): the next conv's pre-activated operand
#pragma unroll
                    for (int it = 0; it < ITC; ++it)
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[it][c] = v[it][c] > 0.0f ? v[it][c] : v[it][c] * d.out_split_slope;
                }
#pragma unroll
                for (int it = 0; it < ITC; ++it)
                    if ((okmask >> it) & 1u) {
                        split_store4_out(d, simg, srow[it], simg_c, ncol - col_shift, v[it]);   // (QKV: the k image, scale out_split_scale)
                    }
            }
        } else {  // unaligned / ragged N (e.g. the 1-channel HiFi-GAN output conv): per component
#pragma unroll
            for (int it = 0; it < ITC; ++it)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (!((okmask >> it) & 1u) || ncol + c >= d.N) continue;
                    float x = v[it][c];
                    if (resp) x += resp[rowoff[it] + c];
                    x *= d.alpha;
                    if (d.accumulate) x += outp[rowoff[it] + c];
                    outp[rowoff[it] + c] = x;
                }
        }
        }  // itc
    }
}

}  // namespace aldm
