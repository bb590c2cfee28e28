// attn.hip — multi-head attention with head dim 32 (every UNet attention: attention.py:343-367),
// flash-style online softmax on fp32 MFMA 32x32x2.  No score matrix ever reaches HBM (the
// reference materialises 8x1024x1024 fp32 = 32 MB per layer per sample).
//
// Wave-level dataflow (one wave64 = 32 queries, all operands stay in registers):
//   S^T = K * Q^T   (A = K rows (keys), B = Q^T)   -> lane (query = lane&31, half = lane>>5) holds the
//                                                      scores of its query against 16 keys of the tile
//   softmax is therefore lane-local + one xor-32 exchange (no LDS, no transposes),
//   O^T += V^T * P^T (A = V^T, B = P^T = the score registers as they are)
//                                                   -> lane holds 16 output dims of ITS query, so the
//                                                      online-softmax rescale is lane-local too.
// The K-index permutation inside each MFMA is free (d index s pairs with 16+s; key (r&3)+8(r>>2)
// pairs with the same +4), which is what makes both products transpose-free.
#include "igemm_epilogue.h"
#include <float.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>

namespace aldm {


// max(x of this lane, x of lane ^ 32): v_permlane32_swap exchanges the upper 32 lanes of its first operand with the lower 32 of its
// second, so (x, x) comes back as (lower half's x everywhere, upper half's x everywhere).  The two results MUST be copied into
// scalars before they are bit-cast: __builtin_bit_cast(float, sw[1]) on the vector element reads element 0 on this compiler (hipcc,
// ROCm 7.2 — the quirk igemm_epilogue.h's split4 notes), and the "row maximum" of the online softmax was then the maximum over the
// LOWER lane half's 16 keys of each tile only.  Found in round 5 (tools/attn_extreme.py): harmless while every key lies within
// 2^128 of that maximum, because softmax does not depend on its reference — and non-finite output once a key of the other half
// lies 128 log2 units (89 nats) above it.
__device__ __forceinline__ float max_across_halves(float x) {
    const unsigned a = __builtin_bit_cast(unsigned, x);
    const auto sw = __builtin_amdgcn_permlane32_swap(a, a, false, false);
    const unsigned lo = sw[0], hi = sw[1];
    return fmaxf(__builtin_bit_cast(float, lo), __builtin_bit_cast(float, hi));
}

// Exact 3-way split of 8 fp32 values (x = hi + mid + lo, each part the top 16 bits of an fp32) into three bf16x8
// MFMA operands; element j of an operand is x[j].  Same arithmetic as the igemm engine's A-side split
// (igemm_kernel.h, docs/experiments_r1-r6.md §3.1b).
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8 (&part)[3]) {
    unsigned u[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = x[j];
        u[0][j] = __builtin_bit_cast(unsigned, v);
        const float r1 = v - __builtin_bit_cast(float, u[0][j] & 0xFFFF0000u);
        u[1][j] = __builtin_bit_cast(unsigned, r1);
        const float r2 = r1 - __builtin_bit_cast(float, u[1][j] & 0xFFFF0000u);
        u[2][j] = __builtin_bit_cast(unsigned, r2);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        u32x4 w;
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = __builtin_amdgcn_perm(u[q][2 * i + 1], u[q][2 * i], 0x07060302u);
        part[q] = __builtin_bit_cast(bf16x8, w);
    }
}

// (hi, mid) rounded to nearest — the "bf16x3" operands (igemm_epilogue.h split4_rn2): plain __bf16 conversions, which hipcc
// lowers to v_cvt_pk_bf16_f32 (two values per instruction, already packed): 3 VALU per element instead of the truncation
// split's 5.5, and two parts instead of three.
__device__ __forceinline__ void split8_rn2(const float (&x)[8], bf16x8 (&part)[3]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)x[j];
        part[0][j] = h;
        part[1][j] = (__bf16)(x[j] - (float)h);
    }
}

template <int NP>
__device__ __forceinline__ void split8_np(const float (&x)[8], bf16x8 (&part)[3]) {
    if constexpr (NP == 2) split8_rn2(x, part);
    else split8(x, part);
}

// QT = 32-query tiles per wave.  With QT = 2 one wave reuses every K / V fragment it loads for 64
// queries (half the L1/L2 traffic per MFMA); used when the grid still fills the chip.
// BX: both products on the bf16 matrix cores as 6 partial products of exact operand splits (Q once, K / V / P per
// key tile, all in registers): 24 MFMAs of 32 cycles per 32x32 tile pair instead of 32 of 64.  The operand layouts
// carry over unchanged: a 16-wide MFMA k-step takes 8 consecutive entries of what the fp32 kernel feeds one at a time.
// NP (BX only): 3 = "bf16x6" (exact 3-part splits, 6 partial products), 2 = "bf16x3" ((hi, mid) rounded to nearest, 3 partial
// products: half the MFMAs and a cheaper split — the kernel is VALU bound on the splits and the exponentials).
template <bool HAS_MASK, int QT, bool BX = false, int NP = 3>
__global__ __launch_bounds__(256, (QT == 1 && BX && NP == 2 && !HAS_MASK) ? 4 : 2) void attention_d32_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    float* __restrict__ out, int Lq, int Lk, int ldq, int ldk, int ldv, int ldo,
    const float* __restrict__ mask, float scale, void* __restrict__ out_split, int split_c, int parts) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int l31 = lane & 31;
    const int lh = lane >> 5;
    const int h = blockIdx.y;
    const int b = blockIdx.z;
    const int q0 = (blockIdx.x * 4 + wave) * 32 * QT;
    if (q0 >= Lq) return;  // wave-uniform

    // Q fragments: Q[q0 + 32*t + l31][h*32 + 16*lh + s], pre-scaled by scale * log2(e): the scores come out in log2 units
    // and every exponential of the online softmax is one v_exp_f32 (rows past Lq are clamped, never stored)
    const float qscale = scale * 1.44269504088896340736f;
    float qf[QT][16];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qi = min(q0 + 32 * t + l31, Lq - 1);
        const float* qp = q + ((int64_t)b * Lq + qi) * ldq + h * 32 + 16 * lh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(qp + 4 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) qf[t][4 * g + e] = x[e] * qscale;
        }
    }

    bf16x8 qx[BX ? QT : 1][2][3];  // BX: Q^T operands, k-step s covers d = 16*lh + 8*s .. + 7
    if constexpr (BX) {
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float x8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x8[j] = qf[t][8 * s + j];
                split8_np<NP>(x8, qx[t][s]);
            }
    }

    f32x16 oT[QT];
    float m_run[QT], l_run[QT];  // l_run: this lane's half of the row sum
#pragma unroll
    for (int t = 0; t < QT; ++t) {
#pragma unroll
        for (int e = 0; e < 16; ++e) oT[t][e] = 0.f;
        m_run[t] = -INFINITY;
        l_run[t] = 0.f;
    }

    const float* kb = k + (int64_t)b * Lk * ldk + h * 32;
    const float* vb = v + (int64_t)b * Lk * ldv + h * 32;
    const float* mb = HAS_MASK ? mask + (int64_t)b * Lk : nullptr;

    // K / V / mask tiles come through raw buffer loads: the descriptor (wave uniform) carries the base and the byte range of
    // this (batch, head) slice, the key-tile position is a scalar offset, the lane's place in the tile a loop-invariant
    // 32-bit offset — no per-load address arithmetic on the VALU (it was 19 % of the loop's instructions).
    // All 20 loads of a tile are issued back to back and waited for once.
    const __amdgpu_buffer_rsrc_t rk =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(kb), 0, ((Lk - 1) * ldk + 32) * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(vb), 0, ((Lk - 1) * ldv + 32) * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rm =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(HAS_MASK ? mb : vb), 0, (HAS_MASK ? Lk : 1) * 4, 0x00020000);
    const int koff = (l31 * ldk + 16 * lh) * 4;    // byte offsets of this lane inside a tile
    const int voff = (4 * lh * ldv + l31) * 4;
    const int moff = 4 * lh * 4;
    f32x4 kraw[4];
    float vf[16];
    float mk[16];
    const int j_last = ((Lk - 1) >> 5) << 5;
    auto load_tile = [&](int j0) {
        // The prefetch of the tile after the last one re-reads the last tile.  The descriptor's range check is NOT relied on
        // for correctness (it compares the lane offset with num_records - scalar offset, which protects nothing once the
        // scalar part is itself past the end, and 4 rows of a lane offset can run past a short buffer unnoticed): every
        // offset formed here addresses a key < Lk.
        j0 = min(j0, j_last);
        if (j0 + 32 <= Lk) {   // full tile (wave uniform): scalar tile offset + loop-invariant lane offset
            // (readfirstlane: keep the tile offsets in scalar registers whatever the loop optimiser makes of j0)
            const int sk = __builtin_amdgcn_readfirstlane(j0 * ldk * 4);
#pragma unroll
            for (int g = 0; g < 4; ++g)
                kraw[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, koff + 16 * g, sk, 0));
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kr = j0 + (r & 3) + 8 * (r >> 2);   // + 4*lh rows: in the lane offset
                vf[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                      rv, voff, __builtin_amdgcn_readfirstlane(kr * ldv * 4), 0));
                if (HAS_MASK)
                    mk[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                          rm, moff, __builtin_amdgcn_readfirstlane(kr * 4), 0));
            }
        } else {               // ragged last tile: key indices clamped per lane (the duplicates are masked to -inf below)
            const int kj = min(j0 + l31, Lk - 1);
#pragma unroll
            for (int g = 0; g < 4; ++g)
                kraw[g] = __builtin_bit_cast(
                    f32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, (kj * ldk + 16 * lh + 4 * g) * 4, 0, 0));
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kr = min(j0 + (r & 3) + 8 * (r >> 2) + 4 * lh, Lk - 1);
                vf[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, (kr * ldv + l31) * 4, 0, 0));
                if (HAS_MASK) mk[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rm, kr * 4, 0, 0));
            }
        }
    };

    // one 32-key tile
    auto key_tile = [&](int j0) {
        // The tile's K / V (loaded during the previous tile) move out of the landing registers — as split operands on the
        // bf16 paths — and the NEXT tile's loads are issued at once: they have this whole tile's MFMAs and softmax to
        // land in (a prefetch past the last tile re-reads the last tile and is never used).
        bf16x8 kx[2][3], vx[2][3];
        f32x4 kc[4];
        float vc[16], mkc[16];
        if constexpr (BX) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float x8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x8[j] = kraw[2 * s + (j >> 2)][j & 3];
                split8_np<NP>(x8, kx[s]);
#pragma unroll
                for (int j = 0; j < 8; ++j) x8[j] = vf[8 * s + j];
                split8_np<NP>(x8, vx[s]);
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) kc[g] = kraw[g];
#pragma unroll
            for (int r = 0; r < 16; ++r) vc[r] = vf[r];
        }
        if (HAS_MASK) {
#pragma unroll
            for (int r = 0; r < 16; ++r) mkc[r] = mk[r];
        }
        if (j0 + 32 < Lk) load_tile(j0 + 32);   // (wave uniform; a single-tile launch — cross-attention to 8 or 32 keys — has nothing to prefetch)
        // S^T tile = K Q^T: lane (query l31, half lh) gets its query's scores against keys
        // j0 + (r&3) + 8(r>>2) + 4*lh
        f32x16 st[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) st[t][e] = 0.f;
        // (lo-order products first; NP = 2: mid*hi, hi*mid, hi*hi)
        constexpr int NPROD = NP == 3 ? 6 : 3;
        constexpr int PA_[6] = {NP == 3 ? 0 : 1, NP == 3 ? 2 : 0, NP == 3 ? 1 : 0, 0, 1, 0};
        constexpr int PB_[6] = {NP == 3 ? 2 : 0, NP == 3 ? 0 : 1, NP == 3 ? 1 : 0, 1, 0, 0};
        if constexpr (BX) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int p = 0; p < NPROD; ++p)
#pragma unroll
                    for (int t = 0; t < QT; ++t)
                        st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kx[s][PA_[p]], qx[t][s][PB_[p]], st[t], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int t = 0; t < QT; ++t)
                    st[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(kc[s >> 2][s & 3], qf[t][s], st[t], 0, 0, 0);
        }

        // masked keys get -FLT_MAX exactly as masked_fill_(~(mask == 1), -finfo.max) does in the reference ...
        if (HAS_MASK) {
#pragma unroll
            for (int t = 0; t < QT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[t][r] = mkc[r] != 1.0f ? -FLT_MAX : st[t][r];
        }
        if (j0 + 32 > Lk) {   // ... and on the ragged last tile (wave uniform branch) keys past Lk are excluded (-inf)
#pragma unroll
            for (int t = 0; t < QT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    st[t][r] = j0 + (r & 3) + 8 * (r >> 2) + 4 * lh >= Lk ? -INFINITY : st[t][r];
        }
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float tmax = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                tmax = fmaxf(tmax, st[t][r]);
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
            const float m_new = fmaxf(m_run[t], tmax);
            const float alpha = __builtin_amdgcn_exp2f(m_run[t] - m_new);  // 0 on the first tile (m_run = -inf)
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(st[t][r] - m_new);
                st[t][r] = pv;
                psum += pv;
            }
            l_run[t] = l_run[t] * alpha + psum;
            m_run[t] = m_new;
#pragma unroll
            for (int e = 0; e < 16; ++e) oT[t][e] *= alpha;
        }
        if constexpr (BX) {
            // k-step s takes the 8 keys of registers r = 8*s .. 8*s + 7 — the same keys in vf (V^T rows) and in st
            // (P^T columns) of a lane half, so the pairing inside the MFMA is consistent
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    float x8[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) x8[j] = st[t][8 * s + j];
                    bf16x8 px[3];
                    split8_np<NP>(x8, px);
#pragma unroll
                    for (int p = 0; p < NPROD; ++p)
                        oT[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vx[s][PA_[p]], px[PB_[p]], oT[t], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int t = 0; t < QT; ++t)
                    oT[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vc[r], st[t][r], oT[t], 0, 0, 0);
        }
    };
    load_tile(0);
    for (int j0 = 0; j0 < Lk; j0 += 32) key_tile(j0);

#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const float l_tot = l_run[t] + __shfl_xor(l_run[t], 32);
        const float inv = 1.0f / l_tot;
        const int qi = q0 + 32 * t + l31;
        if (qi < Lq) {
            float* op = out ? out + ((int64_t)b * Lq + qi) * ldo + h * 32 + 4 * lh : nullptr;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 x;
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = oT[t][4 * g + e] * inv;
                if (op) *reinterpret_cast<f32x4*>(op + 8 * g) = x;
                // a head = one 32-channel block of the split image (the out-projection GEMM's pre-split A operand)
                if (out_split) split_store4(out_split, (int64_t)b * Lq + qi, split_c, h * 32 + 8 * g + 4 * lh, x, parts);
            }
        }
    }
}

// ---- software-pipelined variant (bf16 paths, Lk % 32 == 0) ------------------------------------------------------------
// A wave issues in order, and the kernel above runs a key tile as dependent phases — 12 MFMAs (S = K Q^T), ~200 VALU (mask,
// max, exp2, sums, the P split), 12 MFMAs (O += V^T P) — so the matrix pipe idles through the softmax and the VALU through
// the MFMAs; with two waves per SIMD in no particular phase relation the counters show the two pipes' busy times simply
// adding up (rocprofv3 --pmc: MFMA busy 33 % + VALU 51 % + waits; profiles/r02_attn_pmc.txt).  Here every MFMA is followed
// IN PROGRAM ORDER by a chunk of VALU work that does not depend on it (an MFMA holds the matrix pipe for 32 cycles = 16 VALU
// issue slots), fenced with sched_barrier so the compiler keeps the order:
//     tile j, phase 1:  MFMA  S_j = K_j Q^T               |  VALU  split P_{j-1}, split V_{j-1}, split K_{j+1}, O *= alpha_{j-1},
//                                                          |        issue the loads of V_j and K_{j+2}
//             phase 2:  MFMA  O += V_{j-1}^T P_{j-1}      |  VALU  softmax of S_j (mask, max, alpha_j, P_j = 2^(S_j - m_j), sums)
// i.e. the P V product trails its softmax by one tile.  Tile 0's phase 1/2 work on P_{-1} = 0.  After the loop the last
// tile's product is finished un-overlapped.
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// PRE (round 3): K and V^T arrive PRE-SPLIT from the qkv projection's epilogue (ALDM_EPI_QKV, include/aldm_hip.h) — `k` is the
// split image [key][heads][NP][32] bf16 of the k columns (ldk = heads), `v` the per-(sample, head, key tile) transposed image
// [tile][NP][32 dims][32 keys] bf16 — so a key tile costs 4 + 4 16-byte loads that ARE the MFMA operands and no split
// arithmetic (96 of the 328 VALU instructions of a tile); the products and their order are unchanged: bit-identical results.
template <bool HAS_MASK, int QT, int NP, bool PRE = false>
__global__ __launch_bounds__(256) void attention_d32_pipe_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    float* __restrict__ out, int Lq, int Lk, int ldq, int ldk, int ldv, int ldo,
    const float* __restrict__ mask, float scale, void* __restrict__ out_split, int split_c, int parts) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int l31 = lane & 31;
    const int lh = lane >> 5;
    const int h = blockIdx.y;
    const int b = blockIdx.z;
    const int q0 = (blockIdx.x * 4 + wave) * 32 * QT;
    if (q0 >= Lq) return;  // wave-uniform

    constexpr int NPROD = NP == 3 ? 6 : 3;
    constexpr int PA_[6] = {NP == 3 ? 0 : 1, NP == 3 ? 2 : 0, NP == 3 ? 1 : 0, 0, 1, 0};
    constexpr int PB_[6] = {NP == 3 ? 2 : 0, NP == 3 ? 0 : 1, NP == 3 ? 1 : 0, 1, 0, 0};
    constexpr int NMF = NPROD * 2 * QT;   // MFMAs of one product of a tile; MFMA i = (k-step i / (NPROD*QT), product, query tile i % QT)

    // Q^T operands, pre-scaled by scale * log2(e) (scores in log2 units), k-step s covers d = 16*lh + 8*s .. + 7
    const float qscale = scale * 1.44269504088896340736f;
    bf16x8 qx[QT][2][3];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qi = min(q0 + 32 * t + l31, Lq - 1);
        const float* qp = q + ((int64_t)b * Lq + qi) * ldq + h * 32 + 16 * lh;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(qp + 8 * s), x1 = *reinterpret_cast<const f32x4*>(qp + 8 * s + 4);
            float x8[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                x8[e] = x0[e] * qscale;
                x8[4 + e] = x1[e] * qscale;
            }
            split8_np<NP>(x8, qx[t][s]);
        }
    }

    // PRE: ldk = heads (the k image's blocks per row); the images' strides follow from heads, NP and Lk
    const int heads_ = PRE ? ldk : 0;
    const char* kimg = reinterpret_cast<const char*>(k) + (PRE ? ((int64_t)b * Lk * heads_ + h) * (64 * NP) : 0);
    const char* vimg = reinterpret_cast<const char*>(v) + (PRE ? ((int64_t)b * heads_ + h) * (Lk >> 5) * (int64_t)(NP * 2048) : 0);
    const float* kb = PRE ? reinterpret_cast<const float*>(kimg) : k + (int64_t)b * Lk * ldk + h * 32;
    const float* vb = PRE ? reinterpret_cast<const float*>(vimg) : v + (int64_t)b * Lk * ldv + h * 32;
    const float* mb = HAS_MASK ? mask + (int64_t)b * Lk : nullptr;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(kb), 0, PRE ? ((Lk - 1) * heads_ + 1) * (64 * NP) : ((Lk - 1) * ldk + 32) * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(vb), 0, PRE ? (Lk >> 5) * NP * 2048 : ((Lk - 1) * ldv + 32) * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rm =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(HAS_MASK ? mb : vb), 0, (HAS_MASK ? Lk : 1) * 4, 0x00020000);
    const int koff = PRE ? l31 * heads_ * (64 * NP) + lh * 32 : (l31 * ldk + 16 * lh) * 4;
    const int voff = PRE ? l31 * 64 + lh * 16 : (4 * lh * ldv + l31) * 4;
    const int moff = 4 * lh * 4;

    f32x4 kraw[4];
    float vf[16], mk[16];
    u32x4 kpre[2][NP], vpre[2][NP];   // PRE: the landing registers hold MFMA operands already
    const int j_last = Lk - 32;   // prefetches past the end re-read the last tile (see load_tile above: the descriptor's
                                  // range check is not relied on; with Lk % 32 == 0 every offset formed here is < Lk)
    auto load_k = [&](int j0) {
        if constexpr (PRE) {
            const int sk = __builtin_amdgcn_readfirstlane(min(j0, j_last) * heads_ * (64 * NP));
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    kpre[s][p] = __builtin_amdgcn_raw_buffer_load_b128(rk, koff + 16 * s + 64 * p, sk, 0);
        } else {
            const int sk = __builtin_amdgcn_readfirstlane(min(j0, j_last) * ldk * 4);
#pragma unroll
            for (int g = 0; g < 4; ++g)
                kraw[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, koff + 16 * g, sk, 0));
        }
    };
    auto load_v = [&](int j0) {
        if constexpr (PRE) {
            const int sv = __builtin_amdgcn_readfirstlane((min(j0, j_last) >> 5) * (NP * 2048));
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    vpre[s][p] = __builtin_amdgcn_raw_buffer_load_b128(rv, voff + 32 * s + 2048 * p, sv, 0);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                vf[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    rv, voff, __builtin_amdgcn_readfirstlane((min(j0, j_last) + (r & 3) + 8 * (r >> 2)) * ldv * 4), 0));
        }
    };
    auto load_m = [&](int j0) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            mk[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                rm, moff, __builtin_amdgcn_readfirstlane((min(j0, j_last) + (r & 3) + 8 * (r >> 2)) * 4), 0));
    };
    bf16x8 kx[2][3], kxn[2][3], vx[2][3], px[QT][2][3];
    f32x16 oT[QT], sc[QT], pr_[QT];   // sc: scores of the tile in flight; pr_: probabilities of the previous tile (fp32)
    float m_run[QT], l_run[QT], alpha[QT], m_new[QT], tmax[QT], psum[QT];

    auto split_k_step = [&](int s, bf16x8 (&dst)[2][3]) {
        if constexpr (PRE) {
#pragma unroll
            for (int p = 0; p < NP; ++p) dst[s][p] = __builtin_bit_cast(bf16x8, kpre[s][p]);
        } else {
            float x8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x8[j] = kraw[2 * s + (j >> 2)][j & 3];
            split8_np<NP>(x8, dst[s]);
        }
    };
    auto split_v_step = [&](int s) {
        if constexpr (PRE) {
#pragma unroll
            for (int p = 0; p < NP; ++p) vx[s][p] = __builtin_bit_cast(bf16x8, vpre[s][p]);
        } else {
            float x8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x8[j] = vf[8 * s + j];
            split8_np<NP>(x8, vx[s]);
        }
    };
    auto split_p_step = [&](int t, int s) {
        float x8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x8[j] = pr_[t][8 * s + j];
        split8_np<NP>(x8, px[t][s]);
    };
    auto rescale = [&](int t) {
#pragma unroll
        for (int e = 0; e < 16; ++e) oT[t][e] *= alpha[t];
    };
    // phase-1 VALU work, chunk c of 3*QT + 4
    auto chunk1 = [&](auto cc, int j) {
        constexpr int c = decltype(cc)::value;
        if constexpr (c < 2 * QT) split_p_step(c >> 1, c & 1);
        else if constexpr (c == 2 * QT) split_v_step(0);
        else if constexpr (c == 2 * QT + 1) {
            split_v_step(1);
            load_v(32 * j);                        // V_j -> split in tile j+1
        } else if constexpr (c == 2 * QT + 2) split_k_step(0, kxn);
        else if constexpr (c == 2 * QT + 3) {
            split_k_step(1, kxn);
            load_k(32 * (j + 2));                  // K_{j+2} -> split in tile j+1
        } else if constexpr (c < 3 * QT + 4) rescale(c - (2 * QT + 4));
    };
    // phase-2 VALU work (softmax of sc), chunk c of 4*QT
    auto chunk2 = [&](auto cc) {
        constexpr int c = decltype(cc)::value;
        if constexpr (c < 4 * QT) {
            constexpr int t = c >> 2, part = c & 3;
            if constexpr (part == 0) {
                if (HAS_MASK) {   // masked keys: -FLT_MAX exactly as masked_fill_(~(mask == 1), -finfo.max) in the reference
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[t][r] = mk[r] != 1.0f ? -FLT_MAX : sc[t][r];
                }
                float mx = sc[t][0];
#pragma unroll
                for (int r = 1; r < 8; ++r) mx = fmaxf(mx, sc[t][r]);
                tmax[t] = mx;
            } else if constexpr (part == 1) {
                float mx = tmax[t];
#pragma unroll
                for (int r = 8; r < 16; ++r) mx = fmaxf(mx, sc[t][r]);
                mx = max_across_halves(mx);
                m_new[t] = fmaxf(m_run[t], mx);
                alpha[t] = __builtin_amdgcn_exp2f(m_run[t] - m_new[t]);   // 0 on the first tile (m_run = -inf)
                m_run[t] = m_new[t];
            } else {
                float ps = part == 2 ? 0.f : psum[t];
#pragma unroll
                for (int r = 8 * (part - 2); r < 8 * (part - 1); ++r) {
                    const float pvv = __builtin_amdgcn_exp2f(sc[t][r] - m_new[t]);
                    sc[t][r] = pvv;
                    ps += pvv;
                }
                psum[t] = ps;
                if constexpr (part == 3) l_run[t] = l_run[t] * alpha[t] + ps;
            }
        }
    };
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto mfma_qk = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int s = i / (NPROD * QT), pr = (i / QT) % NPROD, t = i % QT;
        sc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kx[s][PA_[pr]], qx[t][s][PB_[pr]], (s == 0 && pr == 0) ? zero16 : sc[t],
                                                        0, 0, 0);
    };
    auto mfma_pv = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int s = i / (NPROD * QT), pr = (i / QT) % NPROD, t = i % QT;
        oT[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vx[s][PA_[pr]], px[t][s][PB_[pr]], oT[t], 0, 0, 0);
    };

#pragma unroll
    for (int t = 0; t < QT; ++t) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            oT[t][e] = 0.f;
            pr_[t][e] = 0.f;     // P_{-1} = 0
        }
        m_run[t] = -INFINITY;
        l_run[t] = 0.f;
        alpha[t] = 1.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) vf[r] = 0.f;   // V_{-1} = 0
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int p = 0; p < NP; ++p) vpre[s][p] = u32x4{0u, 0u, 0u, 0u};

    // prologue: K_0 split, K_1 in flight; mask tile 0 in flight
    const int nt = Lk >> 5;
    load_k(0);
    if (HAS_MASK) load_m(0);
    split_k_step(0, kx);
    split_k_step(1, kx);
    load_k(32);

    constexpr int NC1 = 3 * QT + 4, NC2 = 4 * QT;
    for (int j = 0; j < nt; ++j) {
        // phase 1
        static_for<0, NMF>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            mfma_qk(ic);
            if constexpr (i < NMF - 1) {
                if constexpr (i < NC1) chunk1(ic, j);
            } else {   // the last MFMA takes whatever chunks are left (QT = 1 with 3 products has fewer MFMAs than chunks)
                static_for<NMF - 1, NC1>([&](auto cc) { chunk1(cc, j); });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        // phase 2
        static_for<0, NMF>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            mfma_pv(ic);
            if constexpr (i < NMF - 1) {
                if constexpr (i < NC2) chunk2(ic);
            } else {
                static_for<NMF - 1, NC2>([&](auto cc) { chunk2(cc); });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (HAS_MASK) load_m(32 * (j + 1));
#pragma unroll
        for (int t = 0; t < QT; ++t) pr_[t] = sc[t];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int pr = 0; pr < 3; ++pr) kx[s][pr] = kxn[s][pr];
    }
    // the last tile's product
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        split_p_step(t, 0);
        split_p_step(t, 1);
        rescale(t);
    }
    split_v_step(0);
    split_v_step(1);
    static_for<0, NMF>([&](auto ic) { mfma_pv(ic); });

#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const float l_tot = l_run[t] + __shfl_xor(l_run[t], 32);
        const float inv = 1.0f / l_tot;
        const int qi = q0 + 32 * t + l31;
        if (qi < Lq) {
            float* op = out ? out + ((int64_t)b * Lq + qi) * ldo + h * 32 + 4 * lh : nullptr;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 x;
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = oT[t][4 * g + e] * inv;
                if (op) *reinterpret_cast<f32x4*>(op + 8 * g) = x;
                if (out_split) split_store4(out_split, (int64_t)b * Lq + qi, split_c, h * 32 + 8 * g + 4 * lh, x, parts);
            }
        }
    }
}


// ---- round 5: the pre-split self-attention loop, re-scheduled -----------------------------------------------------------
// The pipelined kernel above hangs its VALU work behind a FEW of a phase's MFMAs in logical units (a whole 8-value operand
// split = 44 VALU, a whole rescale, a 16-value exponential block): in the shipped ISA the first five MFMAs of phase 1 are
// followed by 34 / 45 / 51 / 33 / 21 VALU instructions, the next fifteen by none, and an 81-instruction register-copy tail
// (pr_ = sc, kx = kxn) runs with the matrix pipe empty.  The launch holds ONE wave per SIMD (316 registers), so nothing else
// fills those holes: SQ counters 32 % matrix-pipe busy, ~4300 cycles per key tile against 1536 of MFMA work
// (profiles/r04_pmc_sq_bf16x6.txt, r05_attn_*).  An in-order wave overlaps the two pipes only when EVERY MFMA (32 cycles of
// matrix pipe) is followed by about its own 32 cycles of independent VALU issue (~7 instructions).  This kernel is the same
// arithmetic in the same order (bit-identical results) with
//   * work ITEMS of <= ~11 VALU instructions — the split of one PAIR of probabilities into its three dwords, half a rescale,
//     four exponentials with their running sum, half a running-max — dealt one (rarely two) per MFMA slot over the whole phase;
//   * the operand split of the SECOND k-step's probabilities moved under the first k-step's P.V MFMAs (which do not read it),
//     the running max behind it, the exponentials under the second k-step: phase 2 no longer opens on a wait for the scores;
//   * no register copies: scores ping-pong between two buffers (tile j's Q.K^T lands in the one whose probabilities the
//     previous P.V has consumed), K and V^T tiles land in the two register sets their MFMAs read, alternately — the key loop is
//     unrolled by two on the tile parity;
//   * tile 0's Q.K^T and softmax run as a prologue instead of a whole iteration multiplying a zero P_{-1}.
// Only for what the UNet's self-attention launches: K / V^T pre-split by the QKV epilogue, no mask, Lk % 32 == 0.
template <int N, int S>
constexpr int item_slot(int k, int base) { return base + (k * S) / N; }

// LDSKV (round 6, aldm_attention_sched 3): the four waves of a block need the SAME K / V^T tiles — without it each of them streams its
// own copy of every tile from L2 (805 MB of L2 -> register traffic per 16 x 8 x 1024 x 1024 launch for 50 MB of distinct data; the
// ablation of profiles/r05_attn_ablate.txt prices that at 28 of the launch's 95 us).  With it a tile crosses L2 -> LDS ONCE per
// block: its 4 NP fragment sets (K / V^T x k-step x part, 1 KB each: exactly the 64-lane register image the MFMAs consume) are
// LDS-DMA pieces dealt to the four waves (global_load_lds_dwordx4: the lanes fetch what load_k / load_v would have fetched, the
// piece lands lane-contiguous), into an NST-deep ring of tiles; load_k / load_v become conflict-free ds_read_b128 at the same
// places of the schedule.  One s_barrier per key tile at the top of the body: behind it the pieces of tile j + 2 (issued NST - 3
// tiles earlier, waited for with a counted vmcnt) are visible to every wave and the stage of tile j - 1 is free for tile
// j + NST - 1.  Same arithmetic in the same order: bit-identical results.  Needs every wave of the block live (Lq % (128 QT) == 0).
// F16 (round 6, the "f16x3" mode): K and V^T arrive as 2-part IEEE-fp16 images of k_scale * k and v_scale * v (the QKV epilogue
// knows a bound of a LayerNorm-fed projection before the data: R c), q is split here into fp16 parts of q_mul * q (q_mul = softmax
// scale * log2(e) * a power of two under the same bound), the probabilities into fp16 parts of 2^15 p, and both products run
// hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16: 24 matrix instructions per key tile instead of 48.  The scores come out of the
// accumulators in units of 1 / sc_c; the softmax reference is the INTEGER ceil(max score in log2 units) — so the exponent's offset
// 15 - m and the rescale factor 2^(m_old - m_new) are exact — and everything the scalings multiplied in leaves with the final
// out_mul / l.  (A softmax does not depend on its reference: any value >= the row maximum that keeps 2^15 p <= 65504 will do.)
#ifndef ALDM_F16_POFF
#define ALDM_F16_POFF 15.0f   // log2 of the factor the fp16 probabilities carry
#endif
template <int QT, int NP, bool LDSKV = false, bool F16 = false>
__global__ __launch_bounds__(256) void attention_d32_presplit2_kernel(
    const float* __restrict__ q, const void* __restrict__ k_img, const void* __restrict__ vt_img, float* __restrict__ out,
    int Lq, int Lk, int ldq, int heads, int ldo, float scale, void* __restrict__ out_split, int split_c, int parts,
    float q_mul = 0.f, float sc_c = 1.f, float out_mul = 1.f, float out_img_scale = 0.f) {
    static_assert(!F16 || NP == 2, "fp16 images have two parts");
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int l31 = lane & 31;
    const int lh = lane >> 5;
    const int h = blockIdx.y;
    const int b = blockIdx.z;
    const int q0 = (blockIdx.x * 4 + wave) * 32 * QT;
    if (!LDSKV && q0 >= Lq) return;  // wave-uniform (LDSKV: the host launches whole blocks only — every wave joins the barriers)

    constexpr int NPROD = NP == 3 ? 6 : 3;
    constexpr int PA_[6] = {NP == 3 ? 0 : 1, NP == 3 ? 2 : 0, NP == 3 ? 1 : 0, 0, 1, 0};
    constexpr int PB_[6] = {NP == 3 ? 2 : 0, NP == 3 ? 0 : 1, NP == 3 ? 1 : 0, 1, 0, 0};
    constexpr int NMF = NPROD * 2 * QT;   // MFMAs of one product of a tile; MFMA i = (k-step i / (NPROD*QT), product, query tile i % QT)
    constexpr int HALF = NMF / 2;         // ... the first HALF of them are k-step 0

    // Q^T operands, pre-scaled by scale * log2(e) (scores in log2 units), k-step s covers d = 16*lh + 8*s .. + 7
    const float qscale = F16 ? q_mul : scale * 1.44269504088896340736f;
    bf16x8 qx[QT][2][3];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qi = min(q0 + 32 * t + l31, Lq - 1);
        const float* qp = q + ((int64_t)b * Lq + qi) * ldq + h * 32 + 16 * lh;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(qp + 8 * s), x1 = *reinterpret_cast<const f32x4*>(qp + 8 * s + 4);
            float x8[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                x8[e] = x0[e] * qscale;
                x8[4 + e] = x1[e] * qscale;
            }
            if constexpr (F16) {
                // (hi first, as a whole vector; the remainder from the converted-back hi: see item_sp below)
                using f32x8 = float __attribute__((ext_vector_type(8)));
                f32x8 xv;
#pragma unroll
                for (int e = 0; e < 8; ++e) xv[e] = x8[e];
                const f16x8 hv = __builtin_convertvector(xv, f16x8);
                const f16x8 lv = __builtin_convertvector(xv - __builtin_convertvector(hv, f32x8), f16x8);
                qx[t][s][0] = __builtin_bit_cast(bf16x8, hv);
                qx[t][s][1] = __builtin_bit_cast(bf16x8, lv);
            } else {
                split8_np<NP>(x8, qx[t][s]);
            }
        }
    }

    // the images of this (sample, head): k rows [key][heads][NP][32] bf16, v tiles [tile][NP][32 dims][32 keys] bf16
    const char* kimg = reinterpret_cast<const char*>(k_img) + ((int64_t)b * Lk * heads + h) * (64 * NP);
    const char* vimg = reinterpret_cast<const char*>(vt_img) + ((int64_t)b * heads + h) * (Lk >> 5) * (int64_t)(NP * 2048);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(kimg), 0, ((Lk - 1) * heads + 1) * (64 * NP), 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(vimg), 0, (Lk >> 5) * NP * 2048, 0x00020000);
    const int koff = l31 * heads * (64 * NP) + lh * 32;
    const int voff = l31 * 64 + lh * 16;
    const int nt = Lk >> 5;
    const int t_last = nt - 1;    // prefetches past the end re-read the last tile (never used)

    u32x4 Kr[2][2][NP], Vr[2][2][NP];   // [tile parity][k-step][part]: the landing registers ARE the MFMA operands
    u32x4 px[QT][2][NP];                // P^T operands of the tile whose P.V runs next
    f32x16 S[2][QT];                    // [tile parity]: scores, then (in place) probabilities
    f32x16 oT[QT];
    float m_run[QT], l_run[QT], alpha[QT], m_new[QT], tmax[QT], psum[QT];

    // LDSKV: ring of NST tiles x (2 NP K fragment sets + 2 NP V^T fragment sets) x 64 lanes x 16 bytes
    constexpr int NST = 6, TILE_SLOTS = 4 * NP * 64, PPW = NP;   // pieces per wave and tile: 4 NP / 4 waves
    __shared__ u32x4 kv_lds[LDSKV ? NST * TILE_SLOTS : 1];
    auto dma_tile = [&](int tile) {   // this wave's PPW pieces of `tile` (past the end: the last tile again, keeps vmcnt uniform)
        using gptr_t = const __attribute__((address_space(1))) void*;
        using lptr_t = __attribute__((address_space(3))) void*;
        const int tc = min(tile, t_last);
        u32x4* stg = &kv_lds[(tile % NST) * TILE_SLOTS];
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int pi = wave * PPW + i;             // 0 .. 4 NP - 1: (K | V, k-step s, part p)
            const int isv = pi / (2 * NP), sp = pi - isv * (2 * NP), ks = sp / NP, pp = sp - ks * NP;
            const char* ksrc = kimg + ((int64_t)tc * 32 * heads * (64 * NP) + koff + 16 * ks + 64 * pp);
            const char* vsrc = vimg + ((int64_t)tc * (NP * 2048) + voff + 32 * ks + 2048 * pp);
            __builtin_amdgcn_global_load_lds((gptr_t)(isv ? vsrc : ksrc), (lptr_t)(stg + pi * 64), 16, 0, 0);
        }
    };
    auto load_k = [&](auto pc, int tile) {
        constexpr int P = decltype(pc)::value;
        if constexpr (LDSKV) {
            const u32x4* stg = &kv_lds[(tile % NST) * TILE_SLOTS + lane];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int p = 0; p < NP; ++p) Kr[P][s][p] = stg[(s * NP + p) * 64];
        } else {
            const int sk = __builtin_amdgcn_readfirstlane(min(tile, t_last) * 32 * heads * (64 * NP));
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int p = 0; p < NP; ++p) Kr[P][s][p] = __builtin_amdgcn_raw_buffer_load_b128(rk, koff + 16 * s + 64 * p, sk, 0);
        }
    };
    auto load_v = [&](auto pc, int tile) {
        constexpr int P = decltype(pc)::value;
        if constexpr (LDSKV) {
            const u32x4* stg = &kv_lds[(tile % NST) * TILE_SLOTS + 2 * NP * 64 + lane];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int p = 0; p < NP; ++p) Vr[P][s][p] = stg[(s * NP + p) * 64];
        } else {
            const int sv = __builtin_amdgcn_readfirstlane(min(tile, t_last) * (NP * 2048));
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int p = 0; p < NP; ++p) Vr[P][s][p] = __builtin_amdgcn_raw_buffer_load_b128(rv, voff + 32 * s + 2048 * p, sv, 0);
        }
    };
    // LDSKV, top of the body of tile j: my pieces of tile j + 2 have landed (the PPW (NST - 4) pieces of tiles j + 3 .. j + NST - 2
    // may stay in flight), my reads of tile j - 1's stage are complete; behind the barrier that holds for every wave
    auto ring_sync = [&]() {
        constexpr int N = PPW * (NST - 4);
        __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (0 << 8) | ((N >> 4) << 14));
        __builtin_amdgcn_s_barrier();
    };

    // ---- work items (each <= ~11 VALU instructions) ----
    // SP: probabilities 8s + 2p, 8s + 2p + 1 of query tile t -> dword p of the NP parts of px[t][s] (split8's arithmetic)
    auto item_sp = [&](auto pc, int t, int s, int p) {
        constexpr int P = decltype(pc)::value;
        const float x0 = S[P][t][8 * s + 2 * p], x1 = S[P][t][8 * s + 2 * p + 1];
        if constexpr (NP == 3) {
            const unsigned a0 = __builtin_bit_cast(unsigned, x0), a1 = __builtin_bit_cast(unsigned, x1);
            const float r0 = x0 - __builtin_bit_cast(float, a0 & 0xFFFF0000u), r1 = x1 - __builtin_bit_cast(float, a1 & 0xFFFF0000u);
            const unsigned b0 = __builtin_bit_cast(unsigned, r0), b1 = __builtin_bit_cast(unsigned, r1);
            const float s0 = r0 - __builtin_bit_cast(float, b0 & 0xFFFF0000u), s1 = r1 - __builtin_bit_cast(float, b1 & 0xFFFF0000u);
            px[t][s][0][p] = __builtin_amdgcn_perm(a1, a0, 0x07060302u);
            px[t][s][1][p] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
            px[t][s][2][p] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, s1), __builtin_bit_cast(unsigned, s0), 0x07060302u);
        } else if constexpr (F16) {   // (hi, lo) fp16 parts of the (2^15-scaled) probabilities
            // The low part MUST be the remainder of the hi that is stored: written with scalar conversions, hipcc packed the pair
            // with v_cvt_pk_f16_f32 while it derived the remainder from a separate v_cvt_f16_f32 of the same input — and the two
            // instructions do not round every input alike (2 of 4096 query rows came out 7e-5 off: one fp16 ulp of a hi part).
            // Converting BACK from the packed value makes hi + lo exact whatever the instruction rounds to.
            using f16x2 = _Float16 __attribute__((ext_vector_type(2)));
            using f32x2 = float __attribute__((ext_vector_type(2)));
            const f32x2 xx = {x0, x1};
            const f16x2 hh = __builtin_convertvector(xx, f16x2);
            const f16x2 ll = __builtin_convertvector(xx - __builtin_convertvector(hh, f32x2), f16x2);
            px[t][s][0][p] = __builtin_bit_cast(unsigned, hh);
            px[t][s][1][p] = __builtin_bit_cast(unsigned, ll);
        } else {   // (hi, mid) rounded to nearest: split8_rn2's arithmetic
            using bf16x2 = __bf16 __attribute__((ext_vector_type(2)));
            const __bf16 h0 = (__bf16)x0, h1 = (__bf16)x1;
            const __bf16 m0 = (__bf16)(x0 - (float)h0), m1 = (__bf16)(x1 - (float)h1);
            px[t][s][0][p] = __builtin_bit_cast(unsigned, bf16x2{h0, h1});
            px[t][s][1][p] = __builtin_bit_cast(unsigned, bf16x2{m0, m1});
        }
    };
    // RS: half hh of O^T of query tile t times alpha
    auto item_rs = [&](int t, int hh) {
#pragma unroll
        for (int e = 8 * hh; e < 8 * hh + 8; ++e) oT[t][e] *= alpha[t];
    };
    // MX: running max of query tile t, first / second 8 scores (+ the cross-half exchange, alpha, m_run)
    auto item_mx = [&](auto pc, int t, int a) {
        constexpr int P = decltype(pc)::value;
        if (a == 0) {
            float mx = S[P][t][0];
#pragma unroll
            for (int r = 1; r < 8; ++r) mx = fmaxf(mx, S[P][t][r]);
            tmax[t] = mx;
        } else {
            float mx = tmax[t];
#pragma unroll
            for (int r = 8; r < 16; ++r) mx = fmaxf(mx, S[P][t][r]);
            mx = max_across_halves(mx);
            if constexpr (F16) mx = __builtin_ceilf(mx * sc_c);        // accumulator units -> log2 units, up to an integer (exact offsets)
            m_new[t] = fmaxf(m_run[t], mx);
            alpha[t] = __builtin_amdgcn_exp2f(m_run[t] - m_new[t]);   // 0 on the first tile (m_run = -inf)
            m_run[t] = m_new[t];
            if constexpr (F16) m_new[t] = ALDM_F16_POFF - m_new[t];            // the exponent's offset: probabilities come out as 2^15 p
        }
    };
    // EX: probabilities 4g .. 4g + 3 of query tile t (in place) and their part of the row sum, in the scores' order
    auto item_ex = [&](auto pc, int t, int g) {
        constexpr int P = decltype(pc)::value;
        float ps = g == 0 ? 0.f : psum[t];
#pragma unroll
        for (int r = 4 * g; r < 4 * g + 4; ++r) {
            const float pv = F16 ? __builtin_amdgcn_exp2f(__builtin_fmaf(S[P][t][r], sc_c, m_new[t]))
                                 : __builtin_amdgcn_exp2f(S[P][t][r] - m_new[t]);
            S[P][t][r] = pv;
            ps += pv;
        }
        psum[t] = ps;
        if (g == 3) l_run[t] = l_run[t] * alpha[t] + ps;
    };

    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto mfma_qk = [&](auto pc, auto ic) {
        constexpr int P = decltype(pc)::value;
        constexpr int i = decltype(ic)::value;
        constexpr int s = i / (NPROD * QT), pr = (i / QT) % NPROD, t = i % QT;
        if constexpr (F16)
            S[P][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, Kr[P][s][PA_[pr]]),
                                                             __builtin_bit_cast(f16x8, qx[t][s][PB_[pr]]),
                                                             (s == 0 && pr == 0) ? zero16 : S[P][t], 0, 0, 0);
        else
            S[P][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Kr[P][s][PA_[pr]]), qx[t][s][PB_[pr]],
                                                              (s == 0 && pr == 0) ? zero16 : S[P][t], 0, 0, 0);
    };
    auto mfma_pv = [&](auto vc, auto ic) {   // vc: parity of the V tile (= of the P tile)
        constexpr int P = decltype(vc)::value;
        constexpr int i = decltype(ic)::value;
        constexpr int s = i / (NPROD * QT), pr = (i / QT) % NPROD, t = i % QT;
        if constexpr (F16)
            oT[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, Vr[P][s][PA_[pr]]),
                                                           __builtin_bit_cast(f16x8, px[t][s][PB_[pr]]), oT[t], 0, 0, 0);
        else
            oT[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Vr[P][s][PA_[pr]]),
                                                            __builtin_bit_cast(bf16x8, px[t][s][PB_[pr]]), oT[t], 0, 0, 0);
    };

    // item lists of a steady-state tile j (parity P; Q = the other parity = tile j - 1), dealt evenly over the phase's MFMA slots:
    //   phase 1 (Q.K^T of tile j)  : SP(t, 0, p) of P_{j-1} x 4 QT, the first MV of the SP(t, 1, p), RS(t, hh) x 2 QT, load V_j
    //   phase 2 (P.V of tile j - 1): load K_{j+2}, the other 4 QT - MV SP(t, 1, p) — before the first k-step-1 MFMA (slot HALF)
    //                                reads them —, MX(t, a) of S_j x 2 QT, EX(t, g) of S_j x 4 QT
    // MV = 2 QT balances the two phases (QT = 2, 3 parts: 17 items = ~172 / ~184 VALU instructions per 24 slots).
#ifndef ALDM_ATTN_MVQ
#define ALDM_ATTN_MVQ 2   // (A/B builds: 0 .. 4 of the four k-step-1 pairs of every query tile)
#endif
    constexpr int MV = ALDM_ATTN_MVQ * QT;
    static_assert(MV >= 0 && MV <= 4 * QT, "SP items of k-step 1 moved into phase 1");
    constexpr int N1 = 4 * QT + MV + 2 * QT + 1;
    constexpr int N2 = 1 + (4 * QT - MV) + 2 * QT + 4 * QT;
    static_assert(MV == 4 * QT || item_slot<N2, NMF>(4 * QT - MV, 0) < HALF, "P operands of k-step 1 must be complete before its first MFMA");
    auto run_item1 = [&](auto pc, auto kc, int j) {
        constexpr int P = decltype(pc)::value, Q = 1 - P;
        constexpr int k = decltype(kc)::value;
        using QC = std::integral_constant<int, Q>;
        if constexpr (k < 4 * QT) item_sp(QC{}, k / 4, 0, k % 4);
        else if constexpr (k < 4 * QT + MV) item_sp(QC{}, (k - 4 * QT) / 4, 1, (k - 4 * QT) % 4);
        else if constexpr (k < 6 * QT + MV) item_rs((k - 4 * QT - MV) / 2, (k - 4 * QT - MV) % 2);
        else load_v(pc, j);
    };
    auto run_item2 = [&](auto pc, auto kc, int j) {
        constexpr int P = decltype(pc)::value, Q = 1 - P;
        constexpr int k = decltype(kc)::value;
        constexpr int NSP = 4 * QT - MV;
        using QC = std::integral_constant<int, Q>;
        if constexpr (k == 0) load_k(pc, j + 2);   // Kr[P] is free: every Q.K^T MFMA of tile j has been issued and has read it
        else if constexpr (k < 1 + NSP) item_sp(QC{}, (MV + k - 1) / 4, 1, (MV + k - 1) % 4);
        else if constexpr (k < 1 + NSP + 2 * QT) item_mx(pc, (k - 1 - NSP) / 2, (k - 1 - NSP) % 2);
        else item_ex(pc, (k - 1 - NSP - 2 * QT) / 4, (k - 1 - NSP - 2 * QT) % 4);
    };

    auto body = [&](auto pc, int j) {
        constexpr int P = decltype(pc)::value, Q = 1 - P;
        if constexpr (LDSKV) ring_sync();
        // phase 1
        static_for<0, NMF>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            mfma_qk(pc, ic);
            if constexpr (LDSKV && i == 1) dma_tile(j + NST - 1);   // the stage of tile j - 1 is free behind the barrier
            static_for<0, N1>([&](auto kc) {
                if constexpr (item_slot<N1, NMF>(decltype(kc)::value, 0) == i) run_item1(pc, kc, j);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        // phase 2
        static_for<0, NMF>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            mfma_pv(std::integral_constant<int, Q>{}, ic);
            static_for<0, N2>([&](auto kc) {
                if constexpr (item_slot<N2, NMF>(decltype(kc)::value, 0) == i) run_item2(pc, kc, j);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
    };

#pragma unroll
    for (int t = 0; t < QT; ++t) {
#pragma unroll
        for (int e = 0; e < 16; ++e) oT[t][e] = 0.f;
        m_run[t] = -INFINITY;
        l_run[t] = 0.f;
        alpha[t] = 1.f;
    }
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    // prologue: tile 0 (parity 0): its Q.K^T and softmax, nothing to overlap with yet; K_1, V_0 in flight behind K_0
    if constexpr (LDSKV) {   // tiles 0 .. NST - 1 fill the ring (the body of tile j adds tile j + NST - 1); 0 .. 2 are read below
#pragma unroll
        for (int tt = 0; tt < NST; ++tt) dma_tile(tt);
        ring_sync();
    }
    load_k(P0{}, 0);
    load_k(P1{}, 1);
    load_v(P0{}, 0);
    static_for<0, NMF>([&](auto ic) { mfma_qk(P0{}, ic); });
    __builtin_amdgcn_sched_barrier(0);
    load_k(P0{}, 2);
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        item_mx(P0{}, t, 0);
        item_mx(P0{}, t, 1);
#pragma unroll
        for (int g = 0; g < 4; ++g) item_ex(P0{}, t, g);
    }
    __builtin_amdgcn_sched_barrier(0);

    int j = 1;
    for (; j + 1 < nt; j += 2) {
        body(P1{}, j);
        body(P0{}, j + 1);
    }
    // the last tile's probabilities: split, rescale, P.V — un-overlapped
    auto tail = [&](auto pc) {   // pc: parity of the LAST tile
#pragma unroll
        for (int t = 0; t < QT; ++t) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int p = 0; p < 4; ++p) item_sp(pc, t, s, p);
            item_rs(t, 0);
            item_rs(t, 1);
        }
        static_for<0, NMF>([&](auto ic) { mfma_pv(pc, ic); });
    };
    if (j < nt) {   // odd tile left (parity 1)
        body(P1{}, j);
        tail(P1{});
    } else {
        tail(P0{});
    }

#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const float l_tot = l_run[t] + __shfl_xor(l_run[t], 32);
        const float inv = (F16 ? out_mul : 1.0f) / l_tot;   // (F16: O carries 2^15 v_scale, l carries 2^15)
        const int qi = q0 + 32 * t + l31;
        if (qi < Lq) {
            float* op = out ? out + ((int64_t)b * Lq + qi) * ldo + h * 32 + 4 * lh : nullptr;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 x;
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = oT[t][4 * g + e] * inv;
                if (op) *reinterpret_cast<f32x4*>(op + 8 * g) = x;
                if constexpr (F16) {   // (a convex combination of the values: |out| <= max|v|, the image may carry v's scale)
                    if (out_split && out_img_scale != 0.f)
                        split_store4_f16(out_split, (int64_t)b * Lq + qi, split_c, h * 32 + 8 * g + 4 * lh, x, out_img_scale);
                    else if (out_split)
                        split_store4(out_split, (int64_t)b * Lq + qi, split_c, h * 32 + 8 * g + 4 * lh, x, parts);
                } else {
                    if (out_split) split_store4(out_split, (int64_t)b * Lq + qi, split_c, h * 32 + 8 * g + 4 * lh, x, parts);
                }
            }
        }
    }
}

// ---- round 5 (second step, OPT-IN: ALDM_ATTN_SCHED=2): one pass over the scores — a FIXED softmax reference per query row ----------
// An experiment, kept for its ablation builds (ALDM_ATTN3_ABLATE, profiles/r05_attn_ablate.txt) — it answered what bounds this loop.
// attention_d32_presplit2_kernel's loop carries ~316 VALU instructions per key tile beside 48 MFMAs.  A third of them exist only
// because the row maximum must be known before the first exponential: the running max over all scores, a subtract per score, the
// alpha bookkeeping and the rescale of O^T, the probabilities written back in place and read again by the operand split.  None of
// this is needed for the RESULT: softmax is invariant to the reference m in p = 2^(s - m) as long as nothing overflows, and
// O^T / l is formed once at the end.  This kernel therefore
//   * takes the row's reference from tile 0 (its exact maximum) and keeps it: -m sits in a 16-register block that is the C operand
//     of every tile's first Q.K^T MFMA, so the accumulators hold s - m and no subtract is issued;
//   * reads each score ONCE: exponential, row sum and the 3-part operand split of a PAIR of probabilities are one work item in
//     three parts of 4 - 6 VALU instructions, one part per MFMA slot (exactly 24 parts per 24-MFMA phase); the P^T operands of
//     k-step 0 are double-buffered (the previous tile's P.V is reading its own), the scores themselves are never rewritten;
//   * has no per-tile running max, alpha or O^T rescale.  Instead the tile's row sum is compared with 2^60: only if some row of
//     the wave collected a probability that large (a score ~40 nats above everything tile 0 held) the wave takes a slow path that
//     raises those rows' reference to the tile's maximum — rescales O^T and l, recomputes the tile's probabilities from the intact
//     scores, shifts the next tile's scores — exactly what the online softmax does every tile.  2^60 leaves 2^67 of headroom
//     to fp32 overflow for the row sum of 1024 keys and for O^T; an overflowed exponential (inf) trips the same test;
//   * parks the Q^T operands in accumulator registers (MFMAs read them there; the allocator otherwise reloads them 56 times a pair
//     of tiles) and pins every finished operand dword to its slot (the compiler sinks the splits behind the overflow test otherwise).
// 226 VALU instructions per key tile (-28 %), 4.7 per MFMA, evenly dealt.  MEASURED (16 x 8 heads x 1024 x 1024, one box): 95.3
// against 98.7 us (-3.5 %), the UNet step +-0.05 ms — the loop was NOT bound by its VALU instruction count any more.  The ablation
// builds say by what: MFMAs alone, operands cache-resident, 57 us (the matrix pipe at the power-capped clock); + the real K / V
// loads 85 us (every wave streams its own copy of K and V^T from L2: 805 MB per launch); + the VALU work 94 - 97 us; the VALU work
// alone 47 us; four accumulator chains instead of two: slower.  What would move it is K / V shared by a block's four waves through
// LDS AND less VALU — neither alone (removing either leaves the other: 88 / 87 us).  Error vs fp64 1.2e-6 (exact-max: 7e-7): -m
// enters the accumulator first, so the 2^-16-sized partial products are rounded at the magnitude of m.  Not the default.
// Results equal the exact-max kernels' to fp32 rounding (the probabilities differ by one common factor per row that cancels in
// O^T / l), not bitwise: tests bound the difference to the exact-max path and compare against fp64
// (test_presplit_kernels_selected_by_env_agree_with_the_default).
// Work items of tile j: 8 QT pairs; the k-step-0 half runs under the P.V MFMAs of tile j - 1 (phase 2 of iteration j), the
// k-step-1 half under the Q.K^T MFMAs of tile j + 1 (phase 1 of iteration j + 1); the overflow test sits between the phases.
#ifndef ALDM_ATTN3_QX_AGPR
#define ALDM_ATTN3_QX_AGPR 1
#endif
#ifndef ALDM_ATTN3_ABLATE
#define ALDM_ATTN3_ABLATE 0   // timing-only ablation builds (wrong results): 1 every K / V load reads tile 0, 2 no operand splits (parts B, C), 4 no MFMAs in the loop, 8 no exponentials (part A), 16 four MFMA accumulator chains instead of two
#endif
template <int QT, int NP>
__global__ __launch_bounds__(256) void attention_d32_presplit3_kernel(
    const float* __restrict__ q, const void* __restrict__ k_img, const void* __restrict__ vt_img, float* __restrict__ out,
    int Lq, int Lk, int ldq, int heads, int ldo, float scale, void* __restrict__ out_split, int split_c, int parts) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int l31 = lane & 31;
    const int lh = lane >> 5;
    const int h = blockIdx.y;
    const int b = blockIdx.z;
    const int q0 = (blockIdx.x * 4 + wave) * 32 * QT;
    if (q0 >= Lq) return;  // wave-uniform

    constexpr int NPROD = NP == 3 ? 6 : 3;
    constexpr int PA_[6] = {NP == 3 ? 0 : 1, NP == 3 ? 2 : 0, NP == 3 ? 1 : 0, 0, 1, 0};
    constexpr int PB_[6] = {NP == 3 ? 2 : 0, NP == 3 ? 0 : 1, NP == 3 ? 1 : 0, 1, 0, 0};
    constexpr int NMF = NPROD * 2 * QT;   // MFMAs of one product of a tile; MFMA i = (k-step i / (NPROD*QT), product, query tile i % QT)

    // Q^T operands, pre-scaled by scale * log2(e) (scores in log2 units), k-step s covers d = 16*lh + 8*s .. + 7
    const float qscale = scale * 1.44269504088896340736f;
    bf16x8 qx[QT][2][3];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qi = min(q0 + 32 * t + l31, Lq - 1);
        const float* qp = q + ((int64_t)b * Lq + qi) * ldq + h * 32 + 16 * lh;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(qp + 8 * s), x1 = *reinterpret_cast<const f32x4*>(qp + 8 * s + 4);
            float x8[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                x8[e] = x0[e] * qscale;
                x8[4 + e] = x1[e] * qscale;
            }
            split8_np<NP>(x8, qx[t][s]);
#if ALDM_ATTN3_QX_AGPR
            // the Q^T operands are read by MFMAs only, for the whole key loop: park them in accumulator registers (an MFMA reads
            // its B operand from there directly) instead of letting the allocator spill and reload them around the loop's VALU work
#pragma unroll
            for (int pr = 0; pr < NP; ++pr) {
                u32x4 w = __builtin_bit_cast(u32x4, qx[t][s][pr]);
                asm volatile("" : "+a"(w));
                qx[t][s][pr] = __builtin_bit_cast(bf16x8, w);
            }
#endif
        }
    }

    // the images of this (sample, head): k rows [key][heads][NP][32] bf16, v tiles [tile][NP][32 dims][32 keys] bf16
    const char* kimg = reinterpret_cast<const char*>(k_img) + ((int64_t)b * Lk * heads + h) * (64 * NP);
    const char* vimg = reinterpret_cast<const char*>(vt_img) + ((int64_t)b * heads + h) * (Lk >> 5) * (int64_t)(NP * 2048);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(kimg), 0, ((Lk - 1) * heads + 1) * (64 * NP), 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(vimg), 0, (Lk >> 5) * NP * 2048, 0x00020000);
    const int koff = l31 * heads * (64 * NP) + lh * 32;
    const int voff = l31 * 64 + lh * 16;
    const int nt = Lk >> 5;
    const int t_last = nt - 1;    // prefetches past the end re-read the last tile (never used)

    u32x4 Kr[2][2][NP], Vr[2][2][NP];   // [tile parity][k-step][part]: the landing registers ARE the MFMA operands
    u32x4 px0[2][QT][NP];               // [tile parity]: P^T operands of k-step 0 (written while the previous tile's P.V reads its own)
    u32x4 px1[QT][NP];                  // ... of k-step 1 (written after the previous tile's P.V has been issued: one set)
    f32x16 S[2][QT];                    // [tile parity]: scores minus the row's reference (log2 units); written by MFMAs only
    f32x16 negm[QT];                    // minus the row's reference, in all 16 registers: C of a tile's first Q.K^T MFMA
    f32x16 oT[QT];
    float l_run[QT], psum[QT];
    float cx0 = 0.f, cx1 = 0.f, cr0 = 0.f, cr1 = 0.f;   // the pair in flight between the three parts of a work item
#if ALDM_ATTN3_ABLATE & 16
    f32x16 oX[QT], oY[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) oX[t][e] = oY[t][e] = 0.f;
#endif
    constexpr float BIG = 1152921504606846976.0f;       // 2^60

    auto load_k = [&](auto pc, int tile, int s) {   // k-step s of a K tile
        constexpr int P = decltype(pc)::value;
        if (ALDM_ATTN3_ABLATE & 1) tile = 0;
        const int sk = __builtin_amdgcn_readfirstlane(min(tile, t_last) * 32 * heads * (64 * NP));
#pragma unroll
        for (int p = 0; p < NP; ++p) Kr[P][s][p] = __builtin_amdgcn_raw_buffer_load_b128(rk, koff + 16 * s + 64 * p, sk, 0);
    };
    auto load_v = [&](auto pc, int tile, int s) {
        constexpr int P = decltype(pc)::value;
        if (ALDM_ATTN3_ABLATE & 1) tile = 0;
        const int sv = __builtin_amdgcn_readfirstlane(min(tile, t_last) * (NP * 2048));
#pragma unroll
        for (int p = 0; p < NP; ++p) Vr[P][s][p] = __builtin_amdgcn_raw_buffer_load_b128(rv, voff + 32 * s + 2048 * p, sv, 0);
    };

    // One work item = scores 8s + 2p, 8s + 2p + 1 of query tile t -> probabilities, their part of the row sum, dword p of the NP
    // parts of the tile's P^T operand of k-step s (split8's / split8_rn2's arithmetic), in three parts of 4 - 6 VALU instructions,
    // one per MFMA slot:  A exponentials + row sum;  B the hi part and the first residuals;  C the mid / lo parts.
    // The empty asm pins a finished dword to ITS slot: the slow path below rewrites every operand register, which makes these
    // writes dead on that edge, and the compiler would otherwise sink the splits out of the MFMA phase behind the overflow test.
    auto px_of = [&](auto pc, int t, int s) -> u32x4(&)[NP] {
        constexpr int P = decltype(pc)::value;
        return s == 0 ? px0[P][t] : px1[t];
    };
    auto item_a = [&](auto pc, int t, int s, int p) {
        constexpr int P = decltype(pc)::value;
        cx0 = __builtin_amdgcn_exp2f(S[P][t][8 * s + 2 * p]);
        cx1 = __builtin_amdgcn_exp2f(S[P][t][8 * s + 2 * p + 1]);
        psum[t] = (s == 0 && p == 0) ? cx0 + cx1 : psum[t] + (cx0 + cx1);
    };
    auto item_b = [&](auto pc, int t, int s, int p) {
        u32x4(&dst)[NP] = px_of(pc, t, s);
        if constexpr (NP == 3) {
            const unsigned a0 = __builtin_bit_cast(unsigned, cx0), a1 = __builtin_bit_cast(unsigned, cx1);
            cr0 = cx0 - __builtin_bit_cast(float, a0 & 0xFFFF0000u);
            cr1 = cx1 - __builtin_bit_cast(float, a1 & 0xFFFF0000u);
            unsigned d0 = __builtin_amdgcn_perm(a1, a0, 0x07060302u);
            asm volatile("" : "+v"(d0));
            dst[0][p] = d0;
        } else {
            using bf16x2 = __bf16 __attribute__((ext_vector_type(2)));
            const __bf16 h0 = (__bf16)cx0, h1 = (__bf16)cx1;
            cr0 = cx0 - (float)h0;
            cr1 = cx1 - (float)h1;
            unsigned d0 = __builtin_bit_cast(unsigned, bf16x2{h0, h1});
            asm volatile("" : "+v"(d0));
            dst[0][p] = d0;
        }
    };
    auto item_c = [&](auto pc, int t, int s, int p) {
        u32x4(&dst)[NP] = px_of(pc, t, s);
        if constexpr (NP == 3) {
            const unsigned b0 = __builtin_bit_cast(unsigned, cr0), b1 = __builtin_bit_cast(unsigned, cr1);
            const float s0 = cr0 - __builtin_bit_cast(float, b0 & 0xFFFF0000u), s1 = cr1 - __builtin_bit_cast(float, b1 & 0xFFFF0000u);
            unsigned d1 = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
            unsigned d2 = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, s1), __builtin_bit_cast(unsigned, s0), 0x07060302u);
            asm volatile("" : "+v"(d1), "+v"(d2));
            dst[1][p] = d1;
            dst[2][p] = d2;
        } else {
            using bf16x2 = __bf16 __attribute__((ext_vector_type(2)));
            unsigned d1 = __builtin_bit_cast(unsigned, bf16x2{(__bf16)cr0, (__bf16)cr1});
            asm volatile("" : "+v"(d1));
            dst[1][p] = d1;
        }
    };
    // part u = 0 .. 12 QT - 1 of the 4 QT items of k-step s of a tile: item u / 3 (query tiles alternate: their row sums are
    // independent chains; pair p = item / QT), part u % 3
    constexpr int NU = 12 * QT;
    auto run_part = [&](auto pc, int s, auto uc, auto loopc) {
        constexpr int u = decltype(uc)::value;
        constexpr int abl = decltype(loopc)::value ? ALDM_ATTN3_ABLATE : 0;   // (ablation builds strip the key LOOP only)
        constexpr int k = u / 3, t = k % QT, p = k / QT;
        if constexpr (u % 3 == 0) {
            if constexpr (!(abl & 8)) item_a(pc, t, s, p);
        } else if constexpr (!(abl & 2)) {
            if constexpr (u % 3 == 1) item_b(pc, t, s, p);
            else item_c(pc, t, s, p);
        }
    };
    // the slow path: raise the reference of the rows of tile parity P to the tile's maximum where that is above it
    auto raise_reference = [&](auto pc, auto nextc) {
        constexpr int P = decltype(pc)::value;
        constexpr bool NEXT = decltype(nextc)::value;   // the other parity already holds the NEXT tile's scores
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float mx = S[P][t][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, S[P][t][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float d = fmaxf(mx, 0.f);              // rows that did not overflow keep their reference (d = 0: no change)
            const float a = __builtin_amdgcn_exp2f(-d);
#pragma unroll
            for (int e = 0; e < 16; ++e) oT[t][e] *= a;
            l_run[t] *= a;
            float ps = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    cx0 = __builtin_amdgcn_exp2f(S[P][t][8 * s + 2 * p] - d);
                    cx1 = __builtin_amdgcn_exp2f(S[P][t][8 * s + 2 * p + 1] - d);
                    ps += cx0 + cx1;
                    item_b(pc, t, s, p);
                    item_c(pc, t, s, p);
                }
            psum[t] = ps;
#pragma unroll
            for (int r = 0; r < 16; ++r) negm[t][r] -= d;
            if constexpr (NEXT) {
#pragma unroll
                for (int r = 0; r < 16; ++r) S[1 - P][t][r] -= d;
            }
        }
    };
    // after the last part of a tile (parity P): the overflow test, then the row sum joins l
    auto close_tile = [&](auto pc, auto nextc) {
        bool big = false;
#pragma unroll
        for (int t = 0; t < QT; ++t) big |= psum[t] > BIG;
        if (__builtin_amdgcn_ballot_w64(big) != 0) raise_reference(pc, nextc);
#pragma unroll
        for (int t = 0; t < QT; ++t) l_run[t] += psum[t];
    };

    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto mfma_qk = [&](auto pc, auto ic, auto firstc) {
        constexpr int P = decltype(pc)::value;
        constexpr int i = decltype(ic)::value;
        constexpr bool FIRST = decltype(firstc)::value;   // tile 0: no reference yet
        constexpr int s = i / (NPROD * QT), pr = (i / QT) % NPROD, t = i % QT;
#if ALDM_ATTN3_ABLATE & 16
        // (timing only: four independent accumulator chains instead of two — is the loop bound by the dependent-MFMA latency?)
        if constexpr (!FIRST && ((i / QT) & 1)) {
            oX[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Kr[P][s][PA_[pr]]), qx[t][s][PB_[pr]], oX[t], 0, 0, 0);
            return;
        }
#endif
        S[P][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Kr[P][s][PA_[pr]]), qx[t][s][PB_[pr]],
                                                          (s == 0 && pr == 0) ? (FIRST ? zero16 : negm[t]) : S[P][t], 0, 0, 0);
    };
    auto mfma_pv = [&](auto vc, auto ic) {   // vc: parity of the V tile (= of the P tile)
        constexpr int P = decltype(vc)::value;
        constexpr int i = decltype(ic)::value;
        constexpr int s = i / (NPROD * QT), pr = (i / QT) % NPROD, t = i % QT;
#if ALDM_ATTN3_ABLATE & 16
        if constexpr ((i / QT) & 1) {
            oY[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                __builtin_bit_cast(bf16x8, Vr[P][s][PA_[pr]]),
                __builtin_bit_cast(bf16x8, s == 0 ? px0[P][t][PB_[pr]] : px1[t][PB_[pr]]), oY[t], 0, 0, 0);
            return;
        }
#endif
        oT[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
            __builtin_bit_cast(bf16x8, Vr[P][s][PA_[pr]]),
            __builtin_bit_cast(bf16x8, s == 0 ? px0[P][t][PB_[pr]] : px1[t][PB_[pr]]), oT[t], 0, 0, 0);
    };

    // Iteration j (tile parity P, Q = the other one):
    //   phase 1, NMF slots: Q.K^T of tile j -> S[P];  the k-step-1 parts of tile j - 1 (-> px1; P.V of tile j - 2 has been issued);
    //                        V^T of tile j in slots 0 and 1
    //   the overflow test of tile j - 1
    //   phase 2, NMF slots: P.V of tile j - 1 (px0[Q], px1);  the k-step-0 parts of tile j (-> px0[P]);  K of tile j + 2 in slots
    //                        0 and 1 (Kr[P] is free: every Q.K^T MFMA of tile j has been issued and has read it)
    // NU parts over NMF slots: one part per slot in the 3-part form (24 / 24), two in the 2-part form.
    auto phase_parts = [&](auto pc, int s, auto ic) {
        constexpr int i = decltype(ic)::value;
        static_for<0, NU>([&](auto uc) {
            if constexpr (item_slot<NU, NMF>(decltype(uc)::value, 0) == i) run_part(pc, s, uc, std::true_type{});
        });
    };
    auto body = [&](auto pc, int j) {
        constexpr int P = decltype(pc)::value, Q = 1 - P;
        using QC = std::integral_constant<int, Q>;
        constexpr bool MFMA = !(ALDM_ATTN3_ABLATE & 4);
#if (ALDM_ATTN3_ABLATE & 4) && defined(__HIP_DEVICE_COMPILE__)
        // keep the ablated loop a loop: what the removed MFMAs would have produced is opaque to the optimiser
#pragma unroll
        for (int t = 0; t < QT; ++t) asm volatile("" : "+v"(S[P][t]), "+v"(oT[t]));
#endif
        static_for<0, NMF>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (MFMA) mfma_qk(pc, ic, std::false_type{});
            phase_parts(QC{}, 1, ic);
            if constexpr (i < 2) load_v(pc, j, i);
            __builtin_amdgcn_sched_barrier(0);
        });
#if (ALDM_ATTN3_ABLATE & 10) == 10
        // (no work item reads the scores: keep their MFMAs alive)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int t = 0; t < QT; ++t) asm volatile("" ::"v"(S[P][t]));
#endif
#else
        close_tile(QC{}, std::true_type{});
#endif
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, NMF>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (MFMA) mfma_pv(QC{}, ic);
            phase_parts(pc, 0, ic);
            if constexpr (i < 2) load_k(pc, j + 2, i);
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    // prologue: tile 0 (parity 0) — its exact row maximum becomes the reference; nothing to overlap with yet
    load_k(P0{}, 0, 0);
    load_k(P0{}, 0, 1);
    load_k(P1{}, 1, 0);
    load_k(P1{}, 1, 1);
    load_v(P0{}, 0, 0);
    load_v(P0{}, 0, 1);
    static_for<0, NMF>([&](auto ic) { mfma_qk(P0{}, ic, std::true_type{}); });
    __builtin_amdgcn_sched_barrier(0);
    load_k(P0{}, 2, 0);
    load_k(P0{}, 2, 1);
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        float mx = S[0][t][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, S[0][t][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            S[0][t][r] -= mx;
            negm[t][r] = -mx;
            oT[t][r] = 0.f;
        }
        l_run[t] = 0.f;
    }
    static_for<0, NU>([&](auto uc) { run_part(P0{}, 0, uc, std::false_type{}); });
    __builtin_amdgcn_sched_barrier(0);

    int j = 1;
    for (; j + 1 < nt; j += 2) {
        body(P1{}, j);
        body(P0{}, j + 1);
    }
    // the last tile: its k-step-1 parts, the overflow test, P.V — un-overlapped
    auto tail = [&](auto pc) {   // pc: parity of the LAST tile
        static_for<0, NU>([&](auto uc) { run_part(pc, 1, uc, std::false_type{}); });
        close_tile(pc, std::false_type{});
        static_for<0, NMF>([&](auto ic) { mfma_pv(pc, ic); });
    };
    if (j < nt) {   // odd tile left (parity 1)
        body(P1{}, j);
        tail(P1{});
    } else {
        tail(P0{});
    }

#if ALDM_ATTN3_ABLATE & 16
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) oT[t][e] += oX[t][e] + oY[t][e];
#endif
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const float l_tot = l_run[t] + __shfl_xor(l_run[t], 32);
        const float inv = 1.0f / l_tot;
        const int qi = q0 + 32 * t + l31;
        if (qi < Lq) {
            float* op = out ? out + ((int64_t)b * Lq + qi) * ldo + h * 32 + 4 * lh : nullptr;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 x;
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = oT[t][4 * g + e] * inv;
                if (op) *reinterpret_cast<f32x4*>(op + 8 * g) = x;
                if (out_split) split_store4(out_split, (int64_t)b * Lq + qi, split_c, h * 32 + 8 * g + 4 * lh, x, parts);
            }
        }
    }
}


}  // namespace aldm

using namespace aldm;

// matrix-core path of the attention kernel: -1 = default (bf16x3 since round 2: 1024x1024 self-attention 202 us on the
// fp32 MFMA, 139 us as bf16x6, see profiles/r02_attn_ab*.txt; $ALDM_ATTN_MMA = f32 | bf16x6 | bf16x3 overrides),
// 1 = fp32 MFMA, 2 = bf16x6, 3 = bf16x3
// PROCESS-wide (a thread_local setting silently did not reach other launch threads).
static std::atomic<int> g_attn_mma{-1};
static int default_attn_mode() {
    // $ALDM_ATTN_MMA pins the attention kernel's path; otherwise it follows the engine's $ALDM_MMA (with
    // ALDM_MMA=bf16x6 in the environment the attention used to stay on bf16x3, so "strict" runs measured a mixed mode)
    static const int v = [] {
        auto parse = [](const char* e) {
            if (e == nullptr || e[0] == 0) return 0;
            if (strcmp(e, "f32") == 0) return 1;
            if (strcmp(e, "bf16x6") == 0) return 2;
            if (strcmp(e, "bf16x3") == 0) return 3;
            fprintf(stderr, "[libaldm_hip] ignoring unknown matrix-core mode \"%s\" (f32 | bf16x6 | bf16x3)\n", e);
            return 0;
        };
        int m = parse(getenv("ALDM_ATTN_MMA"));
        if (m == 0) m = parse(getenv("ALDM_MMA"));
        return m ? m : 2;   // the library default is the fp32-grade mode (bf16x6), like audioldm2_amd.ops.MMA_MODE
    }();
    return v;
}
extern "C" int aldm_attention_mma(int mode) {
    if (mode == -1 || (mode >= 1 && mode <= 3)) return g_attn_mma.exchange(mode);
    return g_attn_mma.load();
}

// schedule of the pre-split self-attention kernel (aldm_attention_d32_presplit): -1 = default ($ALDM_ATTN_SCHED, read once, else 1);
// 0 = the round-3 / 4 pipelined kernel, 1 = the re-scheduled exact-max loop, 2 = the one-pass fixed-reference loop.  Process wide,
// like the product mode (an env-only switch could not be A/B-ed inside one test process).
static std::atomic<int> g_attn_sched{-1};
extern "C" int aldm_attention_sched(int sched) {
    if (sched >= -1 && sched <= 3) return g_attn_sched.exchange(sched);
    return g_attn_sched.load();
}

static int attention_launch(const float* q, const float* k, const float* v, float* out, void* out_split, int parts, int B,
                            int heads, int Lq, int Lk, int ldq, int ldk, int ldv, int ldo,
                            const float* mask, float scale, void* stream) {
    ALDM_CHECK(q && k && v && (out || out_split), "aldm_attention_d32: null pointer");
    ALDM_CHECK(parts == 2 || parts == 3, "aldm_attention_d32: parts must be 2 or 3");
    ALDM_CHECK((reinterpret_cast<uintptr_t>(out_split) & 15) == 0, "aldm_attention_d32: out_split must be 16-byte aligned");
    const int split_c = heads * 32;
    if (!out) ldo = heads * 32;
    ALDM_CHECK(B > 0 && heads > 0 && Lq > 0 && Lk > 0, "aldm_attention_d32: bad sizes");
    ALDM_CHECK(ldq % 4 == 0 && ldk % 4 == 0 && ldo % 4 == 0 && ldq >= heads * 32 &&
                   ldk >= heads * 32 && ldv >= heads * 32 && ldo >= heads * 32,
               "aldm_attention_d32: row pitches must be multiples of 4 and >= heads*32");
    ALDM_CHECK(((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) |
                 reinterpret_cast<uintptr_t>(out)) & 15) == 0,
               "aldm_attention_d32: q/k/out must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    // 64 queries per wave (every K / V fragment and its split reused for two query tiles) whenever the grid still has >= 128
    // blocks: 1024x1024 self-attention 85.7 us vs 106 (QT = 1), 256x256 (192 blocks) 14.7 vs 16.5 us, 1024x32 cross-attention
    // 11.2 vs 13.2 us on MI355X (profiles/r02_attn_ab_pipelined.txt; round 1's kernel preferred QT = 1 for short key lists)
    static const int env_qt = [] {
        const char* e = getenv("ALDM_ATTN_QT");  // A/B override: 1 or 2 query tiles per wave
        return e ? atoi(e) : 0;
    }();
    static const bool pipe = [] {   // A/B override: ALDM_ATTN_PIPE=0 runs the phase-by-phase kernel on the bf16 paths too
        const char* e = getenv("ALDM_ATTN_PIPE");
        return e == nullptr || e[0] != '0';
    }();
    // (not for Lq < 128: with 64 queries per wave a 64-query launch keeps ONE wave of each block busy — 16 x 20 heads x 64 x 64 in
    //  bf16x6: 13.0 us with 64 queries per wave, 8.0 us with 32, profiles/r04_attn_probe_fp32_pv.txt)
    const bool qt2 = env_qt ? env_qt == 2 : (Lq >= 128 && (int64_t)cdiv(Lq, 256) * heads * B >= 128);
    const int gm = g_attn_mma.load();
    const int amode = gm < 0 ? default_attn_mode() : gm;
    dim3 grid(cdiv(Lq, qt2 ? 256 : 128), heads, B);
#define ALDM_ATTN(M_, Q_, X_, P_)                                                                                 \
    hipLaunchKernelGGL((attention_d32_kernel<M_, Q_, X_, P_>), grid, dim3(256), 0, st, q, k, v, out, Lq, Lk, ldq, \
                       ldk, ldv, ldo, mask, scale, out_split, split_c, parts)
#define ALDM_ATTN_P(M_, Q_, P_)                                                                                        \
    hipLaunchKernelGGL((attention_d32_pipe_kernel<M_, Q_, P_>), grid, dim3(256), 0, st, q, k, v, out, Lq, Lk, ldq, ldk, \
                       ldv, ldo, mask, scale, out_split, split_c, parts)
#define ALDM_ATTN_X(M_, Q_)                                       \
    do {                                                          \
        if (amode == 3 && pipe && Lk % 32 == 0) ALDM_ATTN_P(M_, Q_, 2);           \
        else if (amode == 2 && pipe && Lk % 32 == 0) ALDM_ATTN_P(M_, Q_, 3);      \
        else if (amode == 3) ALDM_ATTN(M_, Q_, true, 2);          \
        else if (amode == 2) ALDM_ATTN(M_, Q_, true, 3);          \
        else ALDM_ATTN(M_, Q_, false, 3);                         \
    } while (0)
    if (mask) {
        if (qt2) ALDM_ATTN_X(true, 2);
        else ALDM_ATTN_X(true, 1);
    } else {
        if (qt2) ALDM_ATTN_X(false, 2);
        else ALDM_ATTN_X(false, 1);
    }
#undef ALDM_ATTN_X
#undef ALDM_ATTN_P
#undef ALDM_ATTN
    ALDM_LAUNCH_CHECK("aldm_attention_d32");
    return 0;
}

extern "C" int aldm_attention_d32_presplit(const float* q, const void* k_split, const void* vt_split, float* out,
                                           void* out_split, int parts, int B, int heads, int Lq, int Lk, int ldq, int ldo,
                                           float scale, void* stream) {
    ALDM_CHECK(q && k_split && vt_split && (out || out_split), "aldm_attention_d32_presplit: null pointer");
    ALDM_CHECK(parts == 2 || parts == 3, "aldm_attention_d32_presplit: parts must be 2 or 3");
    ALDM_CHECK(B > 0 && heads > 0 && Lq > 0 && Lk > 0 && Lk % 32 == 0, "aldm_attention_d32_presplit: Lk must be a multiple of 32");
    if (!out) ldo = heads * 32;
    ALDM_CHECK(ldq % 4 == 0 && ldo % 4 == 0 && ldq >= heads * 32 && ldo >= heads * 32,
               "aldm_attention_d32_presplit: row pitches must be multiples of 4 and >= heads*32");
    ALDM_CHECK(((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k_split) | reinterpret_cast<uintptr_t>(vt_split) |
                 reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(out_split)) & 15) == 0,
               "aldm_attention_d32_presplit: operands must be 16-byte aligned");
    const int gm = g_attn_mma.load();
    const int amode = gm < 0 ? default_attn_mode() : gm;
    ALDM_CHECK((amode == 3 && parts == 2) || (amode == 2 && parts == 3),
               "aldm_attention_d32_presplit: %d-part images need the %s attention mode (mode in force: %d)", parts,
               parts == 2 ? "bf16x3" : "bf16x6", amode);
    static const int env_qt = [] {
        const char* e = getenv("ALDM_ATTN_QT");
        return e ? atoi(e) : 0;
    }();
    // (not for Lq < 128: with 64 queries per wave a 64-query launch keeps ONE wave of each block busy — 16 x 20 heads x 64 x 64 in
    //  bf16x6: 13.0 us with 64 queries per wave, 8.0 us with 32, profiles/r04_attn_probe_fp32_pv.txt)
    const bool qt2 = env_qt ? env_qt == 2 : (Lq >= 128 && (int64_t)cdiv(Lq, 256) * heads * B >= 128);
    dim3 grid(cdiv(Lq, qt2 ? 256 : 128), heads, B);
    hipStream_t st = (hipStream_t)stream;
    const float* kf = reinterpret_cast<const float*>(k_split);
    const float* vf = reinterpret_cast<const float*>(vt_split);
    const int split_c = heads * 32;
    // ALDM_ATTN_SCHED (read once; A/Bs with tools/attn_probe.py): unset / 1 = the re-scheduled exact-max loop
    // (attention_d32_presplit2_kernel; bitwise the fp32-K/V path), 0 = the round-3 / 4 pipelined kernel (bitwise too), 2 = the
    // one-pass fixed-reference loop (attention_d32_presplit3_kernel: -26 % VALU instructions for -3 % time and 1.7x the error —
    // the experiment that showed what bounds this loop; kept opt-in for its ablation builds)
    static const int env_sched = [] {
        const char* e = getenv("ALDM_ATTN_SCHED");
        return e == nullptr ? 1 : (e[0] == '0' ? 0 : (e[0] == '2' ? 2 : (e[0] == '3' ? 3 : 1)));
    }();
    const int gs = g_attn_sched.load();
    const int sched = gs < 0 ? env_sched : gs;
#define ALDM_ATTN_PRE(Q_, P_)                                                                                          \
    hipLaunchKernelGGL((attention_d32_pipe_kernel<false, Q_, P_, true>), grid, dim3(256), 0, st, q, kf, vf, out, Lq, Lk, ldq, \
                       heads, 0, ldo, nullptr, scale, out_split, split_c, parts)
#define ALDM_ATTN_PRE2(K_, Q_, P_)                                                                                     \
    hipLaunchKernelGGL((K_<Q_, P_>), grid, dim3(256), 0, st, q, k_split, vt_split, out, Lq, Lk, ldq, heads, ldo, scale,  \
                       out_split, split_c, parts)
    if (sched == 2) {
        if (parts == 2) {
            if (qt2) ALDM_ATTN_PRE2(attention_d32_presplit3_kernel, 2, 2);
            else ALDM_ATTN_PRE2(attention_d32_presplit3_kernel, 1, 2);
        } else {
            if (qt2) ALDM_ATTN_PRE2(attention_d32_presplit3_kernel, 2, 3);
            else ALDM_ATTN_PRE2(attention_d32_presplit3_kernel, 1, 3);
        }
    } else if (sched == 3 && Lq % (qt2 ? 256 : 128) == 0) {
        // K / V^T through LDS once per block (whole blocks only: every wave joins the per-tile barrier)
#define ALDM_ATTN_PRE4(Q_, P_)                                                                                               \
    hipLaunchKernelGGL((attention_d32_presplit2_kernel<Q_, P_, true>), grid, dim3(256), 0, st, q, k_split, vt_split, out, Lq, Lk, ldq, \
                       heads, ldo, scale, out_split, split_c, parts)
        if (parts == 2) {
            if (qt2) ALDM_ATTN_PRE4(2, 2);
            else ALDM_ATTN_PRE4(1, 2);
        } else {
            if (qt2) ALDM_ATTN_PRE4(2, 3);
            else ALDM_ATTN_PRE4(1, 3);
        }
#undef ALDM_ATTN_PRE4
    } else if (sched == 1 || sched == 3) {
        if (parts == 2) {
            if (qt2) ALDM_ATTN_PRE2(attention_d32_presplit2_kernel, 2, 2);
            else ALDM_ATTN_PRE2(attention_d32_presplit2_kernel, 1, 2);
        } else {
            if (qt2) ALDM_ATTN_PRE2(attention_d32_presplit2_kernel, 2, 3);
            else ALDM_ATTN_PRE2(attention_d32_presplit2_kernel, 1, 3);
        }
    } else if (parts == 2) {
        if (qt2) ALDM_ATTN_PRE(2, 2);
        else ALDM_ATTN_PRE(1, 2);
    } else {
        if (qt2) ALDM_ATTN_PRE(2, 3);
        else ALDM_ATTN_PRE(1, 3);
    }
#undef ALDM_ATTN_PRE2
#undef ALDM_ATTN_PRE
    ALDM_LAUNCH_CHECK("aldm_attention_d32_presplit");
    return 0;
}

// "f16x3" self-attention: K / V^T as 2-part fp16 images of k_scale * k, v_scale * v (ALDM_EPI_QKV with out_split_fmt = ALDM_FMT_F16),
// q split in the kernel as fp16 parts of q_scale * softmax-scale * log2(e) * q; out_split (optional) is a bf16 image with
// out_parts parts, or — out_parts = 0 — the 2-part fp16 image of out_scale * out (the output is a convex combination of the
// values: out_scale = v_scale keeps it inside fp16 whenever the V image is).  Three matrix instructions per product in both
// contractions.
extern "C" int aldm_attention_d32_presplit_f16(const float* q, const void* k_split, const void* vt_split, float* out, void* out_split,
                                               int out_parts, float out_scale, int B, int heads, int Lq, int Lk, int ldq, int ldo,
                                               float scale, float q_scale, float k_scale, float v_scale, void* stream) {
    ALDM_CHECK(q && k_split && vt_split && (out || out_split), "aldm_attention_d32_presplit_f16: null pointer");
    ALDM_CHECK(out_parts == 2 || out_parts == 3 || (out_parts == 0 && out_scale > 0.f),
               "aldm_attention_d32_presplit_f16: out_parts must be 2 or 3 (bf16 image) or 0 with out_scale > 0 (fp16 image)");
    const float out_img_scale = out_parts == 0 ? out_scale : 0.f;
    ALDM_CHECK(q_scale > 0.f && k_scale > 0.f && v_scale > 0.f, "aldm_attention_d32_presplit_f16: scales must be positive");
    ALDM_CHECK(B > 0 && heads > 0 && Lq > 0 && Lk > 0 && Lk % 32 == 0, "aldm_attention_d32_presplit_f16: Lk must be a multiple of 32");
    if (!out) ldo = heads * 32;
    ALDM_CHECK(ldq % 4 == 0 && ldo % 4 == 0 && ldq >= heads * 32 && ldo >= heads * 32,
               "aldm_attention_d32_presplit_f16: row pitches must be multiples of 4 and >= heads*32");
    ALDM_CHECK(((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k_split) | reinterpret_cast<uintptr_t>(vt_split) |
                 reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(out_split)) & 15) == 0,
               "aldm_attention_d32_presplit_f16: operands must be 16-byte aligned");
    static const int env_qt = [] {
        const char* e = getenv("ALDM_ATTN_QT");
        return e ? atoi(e) : 0;
    }();
    const bool qt2 = env_qt ? env_qt == 2 : (Lq >= 128 && (int64_t)cdiv(Lq, 256) * heads * B >= 128);
    dim3 grid(cdiv(Lq, qt2 ? 256 : 128), heads, B);
    hipStream_t st = (hipStream_t)stream;
    const float q_mul = scale * 1.44269504088896340736f * q_scale, sc_c = 1.0f / (q_scale * k_scale), out_mul = 1.0f / v_scale;
    const int split_c = heads * 32;
    if (qt2)
        hipLaunchKernelGGL((attention_d32_presplit2_kernel<2, 2, false, true>), grid, dim3(256), 0, st, q, k_split, vt_split, out, Lq, Lk,
                           ldq, heads, ldo, scale, out_split, split_c, out_parts, q_mul, sc_c, out_mul, out_img_scale);
    else
        hipLaunchKernelGGL((attention_d32_presplit2_kernel<1, 2, false, true>), grid, dim3(256), 0, st, q, k_split, vt_split, out, Lq, Lk,
                           ldq, heads, ldo, scale, out_split, split_c, out_parts, q_mul, sc_c, out_mul, out_img_scale);
    ALDM_LAUNCH_CHECK("aldm_attention_d32_presplit_f16");
    return 0;
}

extern "C" int aldm_attention_d32(const float* q, const float* k, const float* v, float* out, int B,
                                  int heads, int Lq, int Lk, int ldq, int ldk, int ldv, int ldo,
                                  const float* mask, float scale, void* stream) {
    ALDM_CHECK(out != nullptr, "aldm_attention_d32: null pointer");
    return attention_launch(q, k, v, out, nullptr, 3, B, heads, Lq, Lk, ldq, ldk, ldv, ldo, mask, scale, stream);
}

extern "C" int aldm_attention_d32_split(const float* q, const float* k, const float* v, float* out, void* out_split,
                                        int parts, int B, int heads, int Lq, int Lk, int ldq, int ldk, int ldv, int ldo,
                                        const float* mask, float scale, void* stream) {
    return attention_launch(q, k, v, out, out_split, parts, B, heads, Lq, Lk, ldq, ldk, ldv, ldo, mask, scale, stream);
}
