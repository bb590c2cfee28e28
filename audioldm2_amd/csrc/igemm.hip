// igemm.hip — implicit-GEMM convolution / GEMM on fp32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
// One engine for every contraction of the AudioLDM2 sampling path (see include/aldm_hip.h):
//   out[m, n] = epi( sum_k A[m, k] * W[k, n] ),  m = (b, oh, ow), k = (kh, kw, ci)
//
// Design (MI355X-first, not a port of any cuDNN/ATen algorithm):
//  * activations are channels-last, so 4 consecutive k of one tap are one 16-byte load and a
//    1x1 conv, a Linear and a conv tap are the same gather;
//  * block tile BM x BN x 32, 4 wave64 (one per SIMD); each wave owns MT x NT MFMA 32x32 tiles
//    (16 accumulator VGPRs each);
//  * LDS image is k-group major: As[kg][row] / Bs[kg][col] hold float4 = 4 consecutive k, so one
//    conflict-free ds_read_b128 feeds 4 MFMAs.  The K index inside the 8-wide sub-step is
//    permuted (lane half h takes k = 4h..4h+3) — legal because A and B use the same permutation;
//  * global -> register prefetch of tile t+1 is issued before the MFMA block of tile t
//    (fp32 MFMA is 64 cycles/instruction: one tile = 4096 MFMA cycles per SIMD hides HBM/L2);
//  * prologue fusion: GroupNorm apply (+SiLU) / leaky_relu on the gathered operand, skip-concat
//    (two source tensors), nearest upsample; epilogue fusion: bias, timestep-embedding row bias,
//    activation, residual, accumulate, strided row remap (polyphase transposed conv);
//  * blockIdx is remapped so each XCD (private 4 MiB L2) walks a contiguous range of tiles.
#include "common.h"

namespace aldm {

struct IgemmK {
    aldm_igemm_desc d;
    int Cin, M, OHW, HV, WV, shh, shw, Kg, Npad, tiles_m, tiles_n;
};

constexpr int BK = 32;
constexpr int KG = BK / 4;

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmK p) {
    constexpr int MT = BM / (32 * WM);
    constexpr int NT = BN / (32 * WN);
    constexpr int PA = BM / 32;  // A-loader passes (32 rows x 8 k-groups per pass)
    constexpr int PB = BN / 32;  // B-loader passes
    static_assert(WM * WN == 4, "4 waves");
    __shared__ f32x4 As[KG][BM + 1];
    __shared__ f32x4 Bs[KG][BN + 1];

    const aldm_igemm_desc& d = p.d;
    const int tid = threadIdx.x;
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
    const float* x1 = d.x1 + (int64_t)z * d.stride_x;
    const float* x2 = d.x2 ? d.x2 + (int64_t)z * d.stride_x : nullptr;
    const float* wgt = d.w + (int64_t)z * d.stride_w;

    // ---- A loader bookkeeping: this thread gathers rows r0 + 32*pp, k-group akg ----
    const int akg = tid & 7;
    const int ar0 = tid >> 3;
    int a_b[PA], a_h[PA], a_w[PA];
#pragma unroll
    for (int pp = 0; pp < PA; ++pp) {
        const int m = m0 + ar0 + 32 * pp;
        if (m < p.M) {
            const int b = m / p.OHW;
            const int rem = m - b * p.OHW;
            const int oh = rem / d.OW;
            const int ow = rem - oh * d.OW;
            a_b[pp] = b;
            a_h[pp] = oh * d.SH - d.PH;
            a_w[pp] = ow * d.SW - d.PW;
        } else {
            a_b[pp] = 0;
            a_h[pp] = -(1 << 28);
            a_w[pp] = 0;
        }
    }
    const int pix1 = d.pix1, pix2 = d.pix2;
    const bool has_pre = d.pre_scale != nullptr;
    const int pre_act = d.pre_act;
    const float pre_slope = d.pre_slope;

    f32x4 ra[PA], rb[PB];

    auto load_a = [&](int k0) {
        const int k = k0 + 4 * akg;
        const bool kval = k < d.K;
        const int tap = kval ? k / p.Cin : 0;
        const int ci = kval ? k - tap * p.Cin : 0;
        const int kh = tap / d.KW;
        const int kw = tap - kh * d.KW;
        const bool first = ci < d.C1;
        const float* src = first ? x1 : x2;
        const int c = first ? ci : ci - d.C1;
        const int pitch = first ? pix1 : pix2;
        const int dh = kh * d.DH, dw = kw * d.DW;
#pragma unroll
        for (int pp = 0; pp < PA; ++pp) {
            const int ihv = a_h[pp] + dh;
            const int iwv = a_w[pp] + dw;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (kval && (unsigned)ihv < (unsigned)p.HV && (unsigned)iwv < (unsigned)p.WV) {
                const int ih = ihv >> p.shh, iw = iwv >> p.shw;
                const int64_t off = ((int64_t)(a_b[pp] * d.H + ih) * d.W + iw) * pitch + c;
                v = *reinterpret_cast<const f32x4*>(src + off);
                if (has_pre) {
                    const int64_t so = (int64_t)a_b[pp] * p.Cin + ci;
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(d.pre_scale + so);
                    const f32x4 sh = *reinterpret_cast<const f32x4*>(d.pre_shift + so);
                    v = v * sc + sh;
                }
                if (pre_act != ALDM_ACT_NONE) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = act_apply(v[j], pre_act, pre_slope);
                }
            }
            ra[pp] = v;
        }
    };

    auto load_b = [&](int k0) {
        if (d.b_mode == ALDM_B_PACKED) {
            const int n = tid % BN;
            const int kg0 = tid / BN;
            constexpr int step = 256 / BN;
#pragma unroll
            for (int pp = 0; pp < PB; ++pp) {
                const int kg = (k0 >> 2) + kg0 + step * pp;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (kg < p.Kg && n0 + n < p.Npad)
                    v = *reinterpret_cast<const f32x4*>(wgt + ((int64_t)kg * p.Npad + n0 + n) * 4);
                rb[pp] = v;
            }
        } else {  // NT: Bmat[N][ldb]
            const int kg = tid & 7;
            const int r0 = tid >> 3;
#pragma unroll
            for (int pp = 0; pp < PB; ++pp) {
                const int n = n0 + r0 + 32 * pp;
                const int k = k0 + 4 * kg;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (n < d.N && k < d.K)
                    v = *reinterpret_cast<const f32x4*>(wgt + (int64_t)n * d.ldb + k);
                rb[pp] = v;
            }
        }
    };

    auto store_lds = [&]() {
#pragma unroll
        for (int pp = 0; pp < PA; ++pp) As[akg][ar0 + 32 * pp] = ra[pp];
        if (d.b_mode == ALDM_B_PACKED) {
            const int n = tid % BN;
            const int kg0 = tid / BN;
            constexpr int step = 256 / BN;
#pragma unroll
            for (int pp = 0; pp < PB; ++pp) Bs[kg0 + step * pp][n] = rb[pp];
        } else {
            const int kg = tid & 7;
            const int r0 = tid >> 3;
#pragma unroll
            for (int pp = 0; pp < PB; ++pp) Bs[kg][r0 + 32 * pp] = rb[pp];
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = (d.K + BK - 1) / BK;
    const int l31 = lane & 31;
    const int lh = lane >> 5;

    load_a(0);
    load_b(0);
    for (int kt = 0; kt < nk; ++kt) {
        store_lds();
        __syncthreads();
        if (kt + 1 < nk) {
            load_a((kt + 1) * BK);
            load_b((kt + 1) * BK);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int kg = 2 * s + lh;
            f32x4 af[MT], bf[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = As[kg][(wm * MT + i) * 32 + l31];
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[j] = Bs[kg][(wn * NT + j) * 32 + l31];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e],
                                                                          acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue ----
    float* outp = d.out + (int64_t)z * d.stride_o;
    const float* resp = d.res ? d.res + (int64_t)z * d.stride_o : nullptr;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (wm * MT + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            const int m = m0 + row;
            if (m >= p.M) continue;
            const int b = m / p.OHW;
            int64_t orow = m;
            if (d.out_mul > 0) {
                const int qq = m - b * p.OHW;
                const int t = qq * d.out_mul + d.out_off;
                if ((unsigned)t >= (unsigned)d.out_len) continue;
                orow = (int64_t)b * d.out_len + t;
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = n0 + (wn * NT + j) * 32 + l31;
                if (n >= d.N) continue;
                float v = acc[i][j][e];
                if (d.bias) v += d.bias[n];
                if (d.rowbias) v += d.rowbias[(int64_t)b * d.N + n];
                v = act_apply(v, d.act, d.act_slope);
                const int64_t o = orow * d.ldo + n;
                if (resp) v += resp[o];
                v *= d.alpha;
                if (d.accumulate) v += outp[o];
                outp[o] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------------------------
// conv / linear: src [N, Cin, KH, KW] -> dst[(k/4)][n][k%4], k = (kh*KW + kw)*Cin + ci
__global__ void pack_weight_kernel(const float* __restrict__ src, float* __restrict__ dst, int N,
                                   int Cin, int KH, int KW, int Kg, int Npad) {
    const int64_t total = (int64_t)Kg * Npad * 4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int j = i & 3;
        const int n = (i >> 2) % Npad;
        const int kg = (i >> 2) / Npad;
        const int k = kg * 4 + j;
        float v = 0.f;
        if (n < N && k < KH * KW * Cin) {
            const int tap = k / Cin, ci = k - tap * Cin;
            const int kh = tap / KW, kw = tap - kh * KW;
            v = src[(((int64_t)n * Cin + ci) * KH + kh) * KW + kw];
        }
        dst[i] = v;
    }
}

// ConvTranspose1d polyphase: src [Cin, N, KWfull]; phase p uses taps kk = p + j*stride,
// j = 0..T-1; stored flipped: kw' = T-1-j  (so ih = q - (T-1) + kw' = q - j).
__global__ void pack_weight_tr_kernel(const float* __restrict__ src, float* __restrict__ dst, int N,
                                      int Cin, int KWfull, int T, int phase, int stride, int Kg,
                                      int Npad) {
    const int64_t total = (int64_t)Kg * Npad * 4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int jj = i & 3;
        const int n = (i >> 2) % Npad;
        const int kg = (i >> 2) / Npad;
        const int k = kg * 4 + jj;
        float v = 0.f;
        if (n < N && k < T * Cin) {
            const int kwp = k / Cin, ci = k - kwp * Cin;
            const int j = T - 1 - kwp;
            const int kk = phase + j * stride;
            if (kk < KWfull) v = src[((int64_t)ci * N + n) * KWfull + kk];
        }
        dst[i] = v;
    }
}

// [K, N] (row pitch lds) -> packed
__global__ void pack_kn_kernel(const float* __restrict__ src, float* __restrict__ dst, int K, int N,
                               int lds, int Kg, int Npad, int64_t stride_src, int64_t stride_dst) {
    const int64_t total = (int64_t)Kg * Npad * 4;
    src += blockIdx.y * stride_src;
    dst += blockIdx.y * stride_dst;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int j = i & 3;
        const int n = (i >> 2) % Npad;
        const int kg = (i >> 2) / Npad;
        const int k = kg * 4 + j;
        dst[i] = (n < N && k < K) ? src[(int64_t)k * lds + n] : 0.f;
    }
}

static int log2_exact(int v) {
    int s = 0;
    while ((1 << s) < v) ++s;
    return (1 << s) == v ? s : -1;
}

}  // namespace aldm

using namespace aldm;

// validation + derived quantities + tile selection, shared by aldm_igemm and aldm_igemm_plan
static int igemm_prepare(const aldm_igemm_desc* dd, IgemmK& p, int& BM, int& BN) {
    ALDM_CHECK(dd != nullptr, "aldm_igemm: null descriptor");
    p.d = *dd;
    aldm_igemm_desc& d = p.d;
    ALDM_CHECK(d.x1 && d.w && d.out, "aldm_igemm: null x1/w/out");
    if (!d.x2) d.C2 = 0;
    if (d.pix1 == 0) d.pix1 = d.C1;
    if (d.pix2 == 0) d.pix2 = d.C2;
    if (d.up_h == 0) d.up_h = 1;
    if (d.up_w == 0) d.up_w = 1;
    if (d.batch == 0) d.batch = 1;
    p.Cin = d.C1 + d.C2;
    ALDM_CHECK(d.C1 > 0 && d.C1 % 4 == 0 && d.C2 % 4 == 0,
               "aldm_igemm: C1=%d/C2=%d must be multiples of 4", d.C1, d.C2);
    ALDM_CHECK(d.pix1 % 4 == 0 && d.pix2 % 4 == 0, "aldm_igemm: pixel pitch must be a multiple of 4");
    ALDM_CHECK(d.K == d.KH * d.KW * p.Cin, "aldm_igemm: K=%d != KH*KW*Cin=%d", d.K,
               d.KH * d.KW * p.Cin);
    ALDM_CHECK(d.N > 0 && d.ldo >= d.N, "aldm_igemm: bad N=%d ldo=%d", d.N, d.ldo);
    ALDM_CHECK(d.B > 0 && d.H > 0 && d.W > 0 && d.OH > 0 && d.OW > 0, "aldm_igemm: bad extents");
    ALDM_CHECK(d.SH > 0 && d.SW > 0 && d.DH > 0 && d.DW > 0, "aldm_igemm: bad stride/dilation");
    p.shh = log2_exact(d.up_h);
    p.shw = log2_exact(d.up_w);
    ALDM_CHECK(p.shh >= 0 && p.shw >= 0, "aldm_igemm: upsample factors must be powers of two");
    ALDM_CHECK((reinterpret_cast<uintptr_t>(d.x1) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(d.w) & 15) == 0 &&
                   (d.x2 == nullptr || (reinterpret_cast<uintptr_t>(d.x2) & 15) == 0),
               "aldm_igemm: operands must be 16-byte aligned");
    if (d.b_mode == ALDM_B_NT) {
        ALDM_CHECK(d.ldb % 4 == 0 && d.ldb >= d.K, "aldm_igemm: NT ldb=%d invalid", d.ldb);
        p.Npad = d.N;
    } else {
        p.Npad = (d.N + 31) / 32 * 32;
        ALDM_CHECK(d.ldb == 0 || d.ldb == p.Npad, "aldm_igemm: packed ldb must equal Npad");
    }
    if (d.out_mul > 0) ALDM_CHECK(d.OH == 1, "aldm_igemm: row remap requires OH == 1");
    ALDM_CHECK((d.pre_scale == nullptr) == (d.pre_shift == nullptr),
               "aldm_igemm: pre_scale/pre_shift must come together");
    p.OHW = d.OH * d.OW;
    const int64_t M64 = (int64_t)d.B * p.OHW;
    ALDM_CHECK(M64 < (1ll << 31) - 256, "aldm_igemm: M too large");
    p.M = (int)M64;
    p.HV = d.H * d.up_h;
    p.WV = d.W * d.up_w;
    p.Kg = (d.K + 3) / 4;

    // tile selection: widest N tile that the layer fills; drop to BM=64 when the 128-row grid
    // would leave most of the 256 CUs idle (deep UNet levels at small batch).
    BN = d.N > 64 ? 128 : (d.N > 32 ? 64 : 32);
    BM = 128;
    if (BN >= 64) {
        const int64_t blocks128 = cdiv64(p.M, 128) * cdiv(d.N, BN) * d.batch;
        if (blocks128 < 512) BM = 64;
    }
    if (BM == 64 && BN == 128) {
        const int64_t blocks = cdiv64(p.M, 64) * cdiv(d.N, 128) * d.batch;
        if (blocks < 256 && d.N % 64 == 0) BN = 64;
    }
    p.tiles_m = cdiv(p.M, BM);
    p.tiles_n = cdiv(d.N, BN);
    return 0;
}

extern "C" int aldm_igemm_plan(const aldm_igemm_desc* dd, int* bm, int* bn, int64_t* flops) {
    IgemmK p;
    int BM, BN;
    const int rc = igemm_prepare(dd, p, BM, BN);
    if (rc) return rc;
    if (bm) *bm = BM;
    if (bn) *bn = BN;
    if (flops) *flops = 2ll * p.M * p.d.N * p.d.K * p.d.batch;
    return 0;
}

extern "C" int aldm_igemm(const aldm_igemm_desc* dd, void* stream) {
    IgemmK p;
    int BM, BN;
    const int rc = igemm_prepare(dd, p, BM, BN);
    if (rc) return rc;
    const aldm_igemm_desc& d = p.d;
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n), 1, (unsigned)d.batch);
    hipStream_t st = (hipStream_t)stream;
#define ALDM_IG(BM_, BN_, WM_, WN_) \
    hipLaunchKernelGGL((igemm_kernel<BM_, BN_, WM_, WN_>), grid, dim3(256), 0, st, p)
    if (BM == 128 && BN == 128) ALDM_IG(128, 128, 2, 2);
    else if (BM == 128 && BN == 64) ALDM_IG(128, 64, 2, 2);
    else if (BM == 128 && BN == 32) ALDM_IG(128, 32, 4, 1);
    else if (BM == 64 && BN == 128) ALDM_IG(64, 128, 2, 2);
    else if (BM == 64 && BN == 64) ALDM_IG(64, 64, 2, 2);
    else {
        set_error("aldm_igemm: no kernel for tile %dx%d", BM, BN);
        return -1;
    }
#undef ALDM_IG
    ALDM_LAUNCH_CHECK("aldm_igemm");
    return 0;
}

extern "C" int aldm_pack_weight(const float* src, float* dst, int N, int Cin, int KH, int KW,
                                int transposed, int phase, int stride, void* stream) {
    ALDM_CHECK(src && dst && N > 0 && Cin > 0 && KH > 0 && KW > 0, "aldm_pack_weight: bad args");
    const int Npad = (N + 31) / 32 * 32;
    hipStream_t st = (hipStream_t)stream;
    if (!transposed) {
        const int K = KH * KW * Cin;
        const int Kg = (K + 3) / 4;
        const int64_t total = (int64_t)Kg * Npad * 4;
        const int blocks = (int)std::min<int64_t>(cdiv64(total, 256), 4096);
        hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, st, src, dst, N, Cin, KH,
                           KW, Kg, Npad);
    } else {
        ALDM_CHECK(KH == 1 && stride > 0 && phase >= 0 && phase < stride,
                   "aldm_pack_weight: bad transposed args");
        const int T = (KW + stride - 1) / stride;
        const int K = T * Cin;
        const int Kg = (K + 3) / 4;
        const int64_t total = (int64_t)Kg * Npad * 4;
        const int blocks = (int)std::min<int64_t>(cdiv64(total, 256), 4096);
        hipLaunchKernelGGL(pack_weight_tr_kernel, dim3(blocks), dim3(256), 0, st, src, dst, N, Cin,
                           KW, T, phase, stride, Kg, Npad);
    }
    ALDM_LAUNCH_CHECK("aldm_pack_weight");
    return 0;
}

extern "C" int aldm_pack_kn(const float* src, float* dst, int K, int N, int lds, int batch,
                            int64_t stride_src, int64_t stride_dst, void* stream) {
    ALDM_CHECK(src && dst && K > 0 && N > 0 && lds >= N && batch > 0, "aldm_pack_kn: bad args");
    const int Npad = (N + 31) / 32 * 32;
    const int Kg = (K + 3) / 4;
    const int64_t total = (int64_t)Kg * Npad * 4;
    const int blocks = (int)std::min<int64_t>(cdiv64(total, 256), 4096);
    hipLaunchKernelGGL(pack_kn_kernel, dim3(blocks, batch), dim3(256), 0, (hipStream_t)stream, src,
                       dst, K, N, lds, Kg, Npad, stride_src, stride_dst);
    ALDM_LAUNCH_CHECK("aldm_pack_kn");
    return 0;
}
