// igemm.hip — host side of the implicit-GEMM engine: validation, tile / split-K selection, launch,
// split-K reduce kernel, weight packing.  The device kernel lives in igemm_kernel.h and is
// instantiated per prologue mode in igemm_pre{0..4}.hip (parallel compilation).
#include "igemm_kernel.h"
#include <algorithm>
#include <atomic>
#include <stdlib.h>

namespace aldm {

// per-prologue launchers (igemm_pre*.hip)
int igemm_launch_pre0(int BM, int BN, int kgroups, bool uni, bool w8, dim3 grid, hipStream_t st, const IgemmK& p);
int igemm_launch_pre1(int BM, int BN, int kgroups, bool uni, bool w8, dim3 grid, hipStream_t st, const IgemmK& p);
int igemm_launch_pre2(int BM, int BN, int kgroups, bool uni, bool w8, dim3 grid, hipStream_t st, const IgemmK& p);
int igemm_launch_pre3(int BM, int BN, int kgroups, bool uni, bool w8, dim3 grid, hipStream_t st, const IgemmK& p);
int igemm_launch_pre4(int BM, int BN, int kgroups, bool uni, bool w8, dim3 grid, hipStream_t st, const IgemmK& p);
// bf16-split ("BF16x6") variants (igemm_bx_pre*.hip)
int igemm_launch_bx_pre0(int BM, int BN, int kgroups, bool uni, bool w8, dim3 grid, hipStream_t st, const IgemmK& p);
int igemm_launch_bx_pre1(int BM, int BN, int kgroups, bool uni, bool w8, dim3 grid, hipStream_t st, const IgemmK& p);
int igemm_launch_bx_pre2(int BM, int BN, int kgroups, bool uni, bool w8, dim3 grid, hipStream_t st, const IgemmK& p);
int igemm_launch_bx_pre3(int BM, int BN, int kgroups, bool uni, bool w8, dim3 grid, hipStream_t st, const IgemmK& p);

// DMA-fed kernel over pre-split operands (igemm_dma.hip)
int igemm_launch_dma(int BM, int BN, int nst, int parts, bool f16, dim3 grid, hipStream_t st, const IgemmK& p);
bool igemm_dma_config_ok(int BM, int BN, int nst, int parts);
#ifdef ALDM_TEST_HOOKS
extern std::atomic<int> g_debug_drop_product;   // test hook, igemm_dma.hip (libaldm_hip_testhooks.so only)
#endif
// ... and its persistent wave-specialised form (igemm_dma_ws.hip)
int igemm_launch_dma_ws(int BM, int BN, int nst, int parts, int blocks, hipStream_t st, const IgemmK& p);
bool igemm_dma_ws_config_ok(int BM, int BN, int nst, int parts);
int igemm_dma_ws_blocks_per_cu(int BM, int BN, int nst, int parts);
// ... and its loader-wave form (igemm_dma_lw.hip): same grid, twice the waves per block
int igemm_launch_dma_lw(int BM, int BN, int nst, int parts, bool f16, dim3 grid, hipStream_t st, const IgemmK& p);
bool igemm_dma_lw_config_ok(int BM, int BN, int nst, int parts);
// ... and the operand-stationary form for short K (igemm_dma_os.hip): weight slab in registers, 32-row stages of the whole K
int igemm_launch_dma_os(int KT, int nst, int parts, bool f16, dim3 grid, hipStream_t st, const IgemmK& p);
bool igemm_dma_os_config_ok(int KT, int nst, int parts);
int igemm_dma_os_default_stages(int KT, int parts);
// ... and the halo-patch form for 3x3 / stride-1 / pad-1 convolutions (igemm_dma_halo.hip): the A patch of a 32-channel block is
// staged in LDS once for all nine taps
int igemm_launch_dma_halo(int BM, int BN, int nstb, int wm, int parts, bool f16, int maxch, dim3 grid, hipStream_t st, const IgemmK& p);
int igemm_dma_halo_maxch(int BM, int BN, int nstb, int wm, int parts, int nch);

// split-K reduce: out = epi(sum_s ws[z][s][m][n]) — fixed summation order, one thread per element
// quad (N % 4 == 0 is required for split-K).
__global__ __launch_bounds__(256) void igemm_reduce_kernel(const IgemmK p) {
    const aldm_igemm_desc& d = p.d;
    const int z = blockIdx.z;
    const int N4 = d.N >> 2;
    const int64_t total = (int64_t)p.M * N4;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int m = (int)(i / N4);
    const int n = (int)(i - (int64_t)m * N4) << 2;
    const int64_t slab = (int64_t)p.M * d.N;
    const float* w = d.ws + (int64_t)z * p.splits * slab + (int64_t)m * d.N + n;
    f32x4 v = *reinterpret_cast<const f32x4*>(w);
    int s = 1;
    for (; s + 3 < p.splits; s += 4) {  // four partial tiles in flight; same summation order as the plain loop
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(w + (int64_t)s * slab);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(w + (int64_t)(s + 1) * slab);
        const f32x4 a2 = *reinterpret_cast<const f32x4*>(w + (int64_t)(s + 2) * slab);
        const f32x4 a3 = *reinterpret_cast<const f32x4*>(w + (int64_t)(s + 3) * slab);
        v += a0;
        v += a1;
        v += a2;
        v += a3;
    }
    for (; s < p.splits; ++s) v += *reinterpret_cast<const f32x4*>(w + (int64_t)s * slab);
    const int b = m / p.OHW;
    int64_t orow = m;
    if (d.out_mul > 0) {
        const int qq = m - b * p.OHW;
        const int t = qq * d.out_mul + d.out_off;
        if ((unsigned)t >= (unsigned)d.out_len) return;
        orow = (int64_t)b * d.out_len + t;
    }
    float* outp = d.out ? d.out + (int64_t)z * d.stride_o : nullptr;
    const float* resp = d.res ? d.res + (int64_t)z * d.stride_o : nullptr;
    const int64_t o = orow * d.ldo + n;
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = epi_value(d, p.rb_ld, outp, resp, b, o + j, n + j, v[j]);
    if (outp) {
#pragma unroll
        for (int j = 0; j < 4; ++j) outp[o + j] = r[j];
    }
    if (d.out_split) {
        if (d.out_split_act == ALDM_ACT_LRELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = r[j] > 0.0f ? r[j] : r[j] * d.out_split_slope;
        }
        split_store4_out(d, d.out_split, orow, d.out_split_c, n, r);
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

// packed fp32 [Kg][Npad][4] -> bf16-split image [Ko][3][Npad][8 bf16], Ko = 4*ceil(Kg/8) k-octets (zero
// padded to whole k-tiles): part 0/1/2 = hi/mid/lo of the exact truncation split w = hi + mid + lo (each
// part = the top 16 bits of an fp32), element j of a slot = k 8*ko + j.  Same split as the kernel's A side.
template <int NP>
__global__ void pack_split_kernel(const float* __restrict__ src, uint4* __restrict__ dst, int Kg, int Npad,
                                  int Ko) {
    const int64_t total = (int64_t)Ko * Npad;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(i % Npad);
        const int ko = (int)(i / Npad);
        u32x2 part[2][3];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kg = 2 * ko + h;
            f32x4 f = {0.f, 0.f, 0.f, 0.f};
            if (kg < Kg) f = *reinterpret_cast<const f32x4*>(src + ((int64_t)kg * Npad + n) * 4);
            split4_parts(f, part[h], NP);   // same split as the activations' (igemm_epilogue.h)
        }
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            uint4 o;
            o.x = part[0][q][0];
            o.y = part[0][q][1];
            o.z = part[1][q][0];
            o.w = part[1][q][1];
            dst[((int64_t)ko * NP + q) * Npad + n] = o;
        }
    }
}

// packed fp32 [Kg][Npad][4] -> "f16x3" weight image [Ko][2][Npad][8 fp16]: hi / lo of scale * w (split4_f16)
__global__ void pack_split_f16_kernel(const float* __restrict__ src, uint4* __restrict__ dst, int Kg, int Npad, int Ko, float scale) {
    const int64_t total = (int64_t)Ko * Npad;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(i % Npad);
        const int ko = (int)(i / Npad);
        u32x2 part[2][3];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kg = 2 * ko + h;
            f32x4 f = {0.f, 0.f, 0.f, 0.f};
            if (kg < Kg) f = *reinterpret_cast<const f32x4*>(src + ((int64_t)kg * Npad + n) * 4);
            split4_f16(f, scale, part[h]);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            uint4 o;
            o.x = part[0][q][0];
            o.y = part[0][q][1];
            o.z = part[1][q][0];
            o.w = part[1][q][1];
            dst[((int64_t)ko * 2 + q) * Npad + n] = o;
        }
    }
}

static int log2_exact(int v) {
    int s = 0;
    while ((1 << s) < v) ++s;
    return (1 << s) == v ? s : -1;
}

}  // namespace aldm

using namespace aldm;

static thread_local int g_force_bm = 0, g_force_bn = 0, g_force_splits = 0, g_force_kgroups = 0, g_force_stages = 0;

extern "C" void aldm_igemm_force(int bm, int bn, int splits, int kgroups) {
    g_force_bm = bm;
    g_force_bn = bn;
    g_force_splits = splits;
    g_force_kgroups = kgroups;
    g_force_stages = 0;
}
extern "C" void aldm_igemm_force_stages(int stages) { g_force_stages = stages; }

static int default_wave8_mask() {
    static const int m = [] {
        const char* e = getenv("ALDM_IGEMM_W8");
        return e ? atoi(e) : 1;
    }();
    return m;
}
static thread_local int g_wave8_mask = -1;

extern "C" int aldm_igemm_wave8_mask(int mask) {
    g_wave8_mask = mask;
    return mask < 0 ? default_wave8_mask() : mask;
}

// matrix-core path override (tests / tools): 0 = automatic (bf16-split when the descriptor carries w_split),
// 1 = fp32 MFMA always, 2 = bf16-split wherever an instantiation exists
static thread_local int g_force_mma = 0;
extern "C" int aldm_igemm_mma(int mode) {
    const int prev = g_force_mma;
    if (mode >= 0 && mode <= 2) g_force_mma = mode;
    return prev;
}

static int pre_mode_of(const aldm_igemm_desc& d) {
    if (d.pre_scale == nullptr && d.pre_act == ALDM_ACT_NONE) return PRE_NONE;
    if (d.pre_scale != nullptr && d.pre_act == ALDM_ACT_NONE) return PRE_AFFINE;
    if (d.pre_scale != nullptr && d.pre_act == ALDM_ACT_SILU) return PRE_AFFINE_SILU;
    if (d.pre_scale == nullptr && d.pre_act == ALDM_ACT_LRELU) return PRE_LRELU;
    return PRE_GENERIC;
}

// compute units of the current device (the persistent kernels launch one block per CU)
static int device_cus() {
    static const int n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            return 256;
        return cus;
    }();
    return n;
}

// the persistent wave-specialised DMA kernel handles the vector epilogue without ragged tiles, activation, row remap,
// accumulation or split-K (igemm_dma_ws.h); everything else stays on igemm_dma_kernel
static bool dma_ws_eligible(const IgemmK& p, int BM, int BN) {
    const aldm_igemm_desc& d = p.d;
    const bool geglu = d.epi_mode == ALDM_EPI_GEGLU;
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (d.out_mul > 0 || d.accumulate || d.batch != 1 || d.act != ALDM_ACT_NONE || d.out_split_act != ALDM_ACT_NONE ||
        d.epi_mode == ALDM_EPI_QKV)
        return false;
    if (p.M % BM != 0 || d.N % BN != 0 || (d.N & 3) != 0 || (d.ldo & 3) != 0) return false;
    if (!al16(d.out) || !al16(d.res) || !al16(d.bias) || !al16(d.rowbias) || !al16(d.out_split)) return false;
    if (d.rowbias && (p.OHW % BM != 0 || (p.rb_ld & 3) != 0)) return false;
    if (geglu && (BN != 128 || d.res || d.rowbias)) return false;
    return true;
}

// the operand-stationary DMA kernel (igemm_dma_os.h) runs 1x1 / linear launches whose output row m reads input row m: one tap,
// stride 1, no padding / upsample / row remap, all of K in one tap, no split-K; the QKV epilogue needs its three segments to be
// whole 128-column slabs
static bool dma_os_eligible(const IgemmK& p) {
    const aldm_igemm_desc& d = p.d;
    if (d.KH != 1 || d.KW != 1 || d.SH != 1 || d.SW != 1 || d.PH != 0 || d.PW != 0 || d.up_h != 1 || d.up_w != 1) return false;
    if (d.OH != d.H || d.OW != d.W || d.K != p.Cin || d.K % 32 != 0 || d.out_mul > 0 || d.batch > 1) return false;
    if (d.epi_mode == ALDM_EPI_QKV && d.qkv_c % 128 != 0) return false;
    // its epilogue is the vector form only: no row bias / output activation / accumulation, float4-addressable fp32 operands
    if (d.rowbias || d.accumulate || d.act != ALDM_ACT_NONE) return false;   // (GEGLU: the erf gate only — act == NONE; T5's tanh gate has K = 1024)
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if ((d.N & 3) != 0 || (d.ldo & 3) != 0 || !al16(d.out) || !al16(d.res)) return false;
    return true;
}

// the halo-patch kernel (igemm_dma_halo.h) runs 3x3 convolutions with stride 1, padding 1, no dilation / upsample / row remap whose
// BM-row tiles are whole image rows of one image (W a power of two dividing BM, OH * OW a multiple of BM); returns the 16-pixel
// chunks per part of its patch, 0 when the launch does not qualify
static int dma_halo_chunks(const IgemmK& p, int BM) {
    const aldm_igemm_desc& d = p.d;
    if (d.KH != 3 || d.KW != 3 || d.SH != 1 || d.SW != 1 || d.PH != 1 || d.PW != 1 || d.DH != 1 || d.DW != 1) return 0;
    if (d.up_h != 1 || d.up_w != 1 || d.OH != d.H || d.OW != d.W || d.out_mul > 0 || d.batch > 1) return 0;
    if (d.W <= 0 || (d.W & (d.W - 1)) != 0 || BM % d.W != 0 || p.OHW % BM != 0) return 0;
    return ((BM / d.W + 2) * d.W + 15) / 16;
}

static bool tile_supported(int BM, int BN) {
    return (BM == 128 && (BN == 128 || BN == 64 || BN == 32)) || (BM == 64 && (BN == 128 || BN == 64));
}

// validation + derived quantities + tile / split-K selection, shared by aldm_igemm and the queries
static int igemm_prepare(const aldm_igemm_desc* dd, IgemmK& p, int& BM, int& BN) {
    ALDM_CHECK(dd != nullptr, "aldm_igemm: null descriptor");
    p.d = *dd;
    aldm_igemm_desc& d = p.d;
    ALDM_CHECK((d.x1 || d.a_split) && d.w && (d.out || d.out_split), "aldm_igemm: null x1/w/out");
    if (!d.x1) d.x1 = reinterpret_cast<const float*>(d.a_split);  // never dereferenced on the DMA path
    if (d.split_parts == 0) d.split_parts = 3;
    ALDM_CHECK(d.split_parts == 2 || d.split_parts == 3, "aldm_igemm: split_parts must be 0 / 3 or 2");
    if (d.out_split_parts == 0) d.out_split_parts = d.split_parts;
    ALDM_CHECK(d.out_split_parts == 2 || d.out_split_parts == 3, "aldm_igemm: out_split_parts must be 0, 2 or 3");
    ALDM_CHECK(d.a_fmt == ALDM_FMT_BF16 || (d.a_fmt == ALDM_FMT_F16 && d.a_split != nullptr && d.split_parts == 2 && d.acc_scale > 0.0f),
               "aldm_igemm: a_fmt = ALDM_FMT_F16 needs a pre-split operand (a_split), 2-part images and acc_scale > 0");
    if (d.a_fmt == ALDM_FMT_BF16) d.acc_scale = 1.0f;
    ALDM_CHECK(d.out_split_fmt == ALDM_FMT_BF16 ||
                   (d.out_split_fmt == ALDM_FMT_F16 && d.out_split_scale > 0.0f &&
                    (d.epi_mode == ALDM_EPI_QKV ? (d.k_split != nullptr && d.vt_split != nullptr && d.vt_scale > 0.0f) : d.out_split != nullptr)),
               "aldm_igemm: out_split_fmt = ALDM_FMT_F16 needs out_split (QKV: k_split, vt_split and vt_scale) and out_split_scale > 0");
    if (d.out_split_fmt == ALDM_FMT_F16) d.out_split_parts = 2;
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
    ALDM_CHECK(d.N > 0 && (d.ldo >= d.N || d.epi_mode == ALDM_EPI_GEGLU || d.epi_mode == ALDM_EPI_QKV),
               "aldm_igemm: bad N=%d ldo=%d", d.N, d.ldo);
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
    ALDM_CHECK(d.pre_scale == nullptr ||
                   ((reinterpret_cast<uintptr_t>(d.pre_scale) | reinterpret_cast<uintptr_t>(d.pre_shift)) & 15) == 0,
               "aldm_igemm: pre_scale/pre_shift must be 16-byte aligned");
    const bool geglu = d.epi_mode == ALDM_EPI_GEGLU;
    const bool qkv = d.epi_mode == ALDM_EPI_QKV;
    ALDM_CHECK(d.epi_mode == ALDM_EPI_PLAIN || geglu || qkv, "aldm_igemm: unknown epi_mode %d", d.epi_mode);
    if (qkv) {
        auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
        ALDM_CHECK(d.a_split && d.qkv_c > 0 && d.qkv_c % 64 == 0 && d.N == 3 * d.qkv_c && d.qkv_rows > 0 && d.qkv_rows % 32 == 0 &&
                       ((int64_t)d.B * d.OH * d.OW) % d.qkv_rows == 0 && d.out && d.k_split && d.vt_split && al16(d.out) &&
                       al16(d.k_split) && al16(d.vt_split) && d.ldo >= d.qkv_c && (d.ldo & 3) == 0 && !d.bias && !d.rowbias &&
                       !d.res && !d.out_split && !d.accumulate && d.out_mul == 0 && d.act == ALDM_ACT_NONE && d.alpha == 1.0f &&
                       d.batch <= 1,
                   "aldm_igemm: ALDM_EPI_QKV needs a pre-split operand, N = 3*qkv_c (qkv_c %% 64 == 0), qkv_rows %% 32 == 0, "
                   "aligned out / k_split / vt_split and a plain epilogue");
    }
    if (geglu) {
        ALDM_CHECK(d.N % 64 == 0 && d.ldo >= d.N / 2 && d.ldo % 4 == 0 && d.b_mode == ALDM_B_PACKED &&
                       d.out_mul == 0 && !d.res && !d.rowbias && !d.accumulate &&
                       (d.act == ALDM_ACT_NONE || d.act == ALDM_ACT_GELU_TANH) &&
                       (reinterpret_cast<uintptr_t>(d.out) & 15) == 0 && d.stride_o % 4 == 0 &&
                       (d.bias == nullptr || (reinterpret_cast<uintptr_t>(d.bias) & 15) == 0),
                   "aldm_igemm: GEGLU epilogue needs N %% 64 == 0, ldo >= N/2, packed weights, a plain epilogue");
    }
    p.OHW = d.OH * d.OW;
    const int64_t M64 = (int64_t)d.B * p.OHW;
    ALDM_CHECK(M64 < (1ll << 31) - 256, "aldm_igemm: M too large");
    p.M = (int)M64;
    p.HV = d.H * d.up_h;
    p.WV = d.W * d.up_w;
    ALDM_CHECK((int64_t)d.B * d.H * d.W < (1ll << 31), "aldm_igemm: input has too many pixels");
    p.Kg = (d.K + 3) / 4;
    p.rb_ld = d.rowbias_ld > 0 ? d.rowbias_ld : d.N;

    // ---- tile + split-K selection: a small cost model over (tile, split) candidates ------------
    // All quantities in MFMA cycles of one SIMD.  A block alone on its 4 SIMDs needs
    // L = kt * BM*BN/4 cycles of matrix pipe (+ per-tile staging S and a fixed prologue/epilogue F);
    // `occ` co-resident blocks share the pipe.  Blocks go round-robin over the 256 CUs; the last,
    // partially filled round costs as much as its fullest CU.  Split-K adds a reduce pass.
    const int nk = (d.K + BK - 1) / BK;
    const int64_t Mz = p.M;
    // Constants fitted on MI355X to a (tile x split) sweep over the UNet's shapes (tools/igemm_tune.py,
    // profiles/r01_igemm_tune.txt): S = un-overlapped staging cycles per k-tile (per 32 A rows / B
    // columns; x1.5 with a GroupNorm/SiLU prologue), F = fixed block prologue + epilogue, reduce =
    // 10k cycles + (splits + 1) * M * N * 4 B at 2000 B/cycle.
    // bf16-split path (BX): 6 bf16 MFMAs of 32 cycles replace 8 fp32 MFMAs of 64 per 16 k (x0.375 matrix-pipe
    // time), the LDS image is 1.5x larger (1 / 2 / 3 resident blocks for the three tile sizes) and staging
    // pays the operand split.
    const int pre = pre_mode_of(d);
    p.dma = 0;
    p.nst = 0;
    p.ws = 0;
    p.ws_blocks = 0;
    p.os_rows = 0;
    ALDM_CHECK(d.out_split_act == ALDM_ACT_NONE || (d.out_split_act == ALDM_ACT_LRELU && d.out_split != nullptr &&
                                                   d.epi_mode == ALDM_EPI_PLAIN),
               "aldm_igemm: out_split_act must be ALDM_ACT_NONE or ALDM_ACT_LRELU with a split-image output of the plain epilogue");
    if (d.out_split) {
        ALDM_CHECK(d.out_split_c > 0 && d.out_split_c % 32 == 0 && d.out_split_c >= (geglu ? d.N / 2 : d.N) &&
                       d.N % 4 == 0 && d.batch == 1 && (reinterpret_cast<uintptr_t>(d.out_split) & 15) == 0 &&
                       ((reinterpret_cast<uintptr_t>(d.out) | reinterpret_cast<uintptr_t>(d.res)) & 15) == 0 &&
                       (d.out == nullptr || (d.ldo & 3) == 0),
                   "aldm_igemm: out_split needs out_split_c %% 32 == 0 and >= N, N %% 4 == 0, batch 1, aligned out/res");
        ALDM_CHECK(d.out != nullptr || !d.accumulate, "aldm_igemm: accumulate needs an fp32 output");
    }
    if (d.a_split) {
        // ---- DMA-fed kernel over pre-split operands (igemm_dma.h) ----
        ALDM_CHECK(d.w_split != nullptr && d.b_mode == ALDM_B_PACKED && d.stride_w == 0 && d.batch == 1,
                   "aldm_igemm: a_split needs packed, split weights and batch 1");
        ALDM_CHECK(d.C2 == 0 && d.C1 % 32 == 0 && pre == PRE_NONE,
                   "aldm_igemm: a_split needs one source with C1 %% 32 == 0 and no prologue (C1=%d C2=%d)", d.C1, d.C2);
        ALDM_CHECK(((reinterpret_cast<uintptr_t>(d.a_split) | reinterpret_cast<uintptr_t>(d.w_split)) & 15) == 0,
                   "aldm_igemm: split images must be 16-byte aligned");
        ALDM_CHECK((int64_t)d.B * d.H * d.W * p.Cin * 2 * d.split_parts < (1ll << 40), "aldm_igemm: split image too large");
        const bool can_split = d.N % 4 == 0 && nk >= 8 && !geglu && !qkv;
        const bool have_ws = d.ws != nullptr && (reinterpret_cast<uintptr_t>(d.ws) & 15) == 0;
        auto dma_cost = [&](int bm, int bn, int nst, int sp, int* sp_eff) -> double {
            const int kt = cdiv(nk, sp);
            sp = cdiv(nk, kt);
            *sp_eff = sp;
            const double blocks = (double)cdiv64(Mz, bm) * cdiv(d.N, bn) * sp;
            const int lds = nst * (bm + bn) * 64 * d.split_parts;
            const int o = std::min(2, (160 * 1024) / lds);
            const double tile_c = (bm / 64) * (bn / 64) * 64.0 * (d.split_parts == 3 ? 6 : 3);   // MFMA cycles per k-tile
            const double dma_c = (bm + bn) * 64.0 * d.split_parts / 56.0;   // L2 -> LDS at ~56 B/clk/CU
            const double L = kt * std::max(tile_c, dma_c);
            const double one = kt * std::max(tile_c + 200.0, dma_c) + 6000.0;
            const int64_t nb = (int64_t)((blocks + 255.0) / 256.0);
            const int64_t full = nb / o, last = nb - full * o;
            double T = full * std::max(one, o * L);
            if (last) T += std::max(one, last * L);
            if (sp > 1) T += 10000.0 + (double)(sp + 1) * Mz * d.N * 4.0 / 2000.0;
            return T;
        };
        int splits = 1, nst = 0;
        // a tuned hint is keyed by geometry, not by epilogue: one whose tile the QKV / GEGLU epilogue cannot use (tile width not
        // dividing qkv_c; not 128 columns) is dropped and the cost search below picks the tile — only a FORCED tile fails
        const bool hint_fits = d.hint_bm > 0 && (!qkv || (d.hint_bn > 0 && d.qkv_c % d.hint_bn == 0)) && (!geglu || d.hint_bn == 128);
        const int f_bm = g_force_bm ? g_force_bm : (hint_fits ? d.hint_bm : 0), f_bn = g_force_bm ? g_force_bn : d.hint_bn;
        const int f_sp = g_force_bm ? g_force_splits : d.hint_splits;
        int f_st = g_force_bm ? g_force_stages : (hint_fits ? d.hint_stages : 0);
        int ws_nst = 0;   // stages in [100, 200): the persistent wave-specialised kernel with a ring of (stages - 100)
        int lw_nst = 0;   // stages in [200, 300): igemm_dma_kernel with loader waves, ring of (stages - 200)
        if (f_st >= 300 && f_st < 400) {
            // stages 300 .. 399: the operand-stationary kernel, ring of (stages - 300) 32-row stages (300 itself: the deepest ring that
            // fits).  A launch it cannot run keeps the cost search's tile on igemm_dma_kernel when the request was a tuned hint
            // (tables are keyed by geometry, not by epilogue) and fails when it was forced.
            const int KT = d.K / 32;
            int os_nst = f_st - 300;
            if (os_nst == 0) os_nst = igemm_dma_os_default_stages(KT, d.split_parts);
            if (dma_os_eligible(p) && igemm_dma_os_config_ok(KT, os_nst, d.split_parts)) {
                // tuning override (tools/os_probe.py): rows per block.  Read once (this runs per launch and per query)
                static const int env_rows = [] {
                    const char* e = getenv("ALDM_OS_ROWS");
                    return e ? atoi(e) : 0;
                }();
                const int tn = cdiv(d.N, 128);
                const int stages = cdiv(p.M, 32);
                const int chunks = std::max(1, device_cus() / tn);   // one block per CU: the largest chunk count that still is one round
                int rows = cdiv(stages, chunks) * 32;
                if (env_rows > 0) rows = cdiv(env_rows, 32) * 32;
                BM = 32;
                BN = 128;
                p.tiles_n = tn;
                p.tiles_m = cdiv(p.M, rows);
                p.os_rows = rows;
                p.kt_per_split = nk;
                p.splits = 1;
                p.kgroups = 1;
                p.bx = 1;
                p.pre = pre;
                p.dma = 1;
                p.nst = os_nst;
                p.ws = 3;
                p.ws_blocks = 0;
                return 0;
            }
            ALDM_CHECK(!g_force_bm, "aldm_igemm: the operand-stationary DMA kernel cannot run this launch (K = %d, %d stages, %d parts)",
                       d.K, os_nst, d.split_parts);
            f_st = 0;
        }
        if (f_st >= 400) {
            // stages 400 + 10 * w8 + depth: the halo-patch kernel (3x3 stride-1 convolutions; depth = weight-ring stages, w8 = 1: the
            // 128-row tile on 8 waves).  Split-K must cut between channel blocks.  As above: a hinted launch it cannot run falls
            // back to the cost search, a forced one fails.
            const int hb = f_st - 400, nstb = hb % 10, w8 = hb / 10;
            const int bm = f_bm ? f_bm : 256, bn = f_bn ? f_bn : 128;
            const int wm = bm == 256 ? 4 : (w8 ? 4 : 2);
            const int nch = dma_halo_chunks(p, bm);
            const int maxch = nch > 0 && !geglu && !qkv ? igemm_dma_halo_maxch(bm, bn, nstb, wm, d.split_parts, nch) : 0;
            const int cpb = p.Cin / 32;
            int sp = f_sp > 0 ? f_sp : 1;
            if (sp > 1 && (!can_split || !have_ws || cpb % sp != 0 || d.ws_floats < (int64_t)sp * p.M * d.N)) sp = 0;
            if (maxch > 0 && sp > 0) {
                BM = bm;
                BN = bn;
                p.tiles_m = p.M / BM;
                p.tiles_n = cdiv(d.N, BN);
                p.splits = sp;
                p.kt_per_split = 9 * (cpb / sp);
                p.kgroups = 1;
                p.bx = 1;
                p.pre = pre;
                p.dma = 1;
                p.nst = nstb;
                p.ws = 4;
                p.ws_blocks = wm;
                p.os_rows = maxch;
                return 0;
            }
            ALDM_CHECK(!g_force_bm, "aldm_igemm: the halo-patch DMA kernel cannot run this launch (tile %dx%d, %d stages, %d waves, "
                       "%d parts, %d splits; 3x3 / stride 1 / pad 1, W a power of two dividing the tile, OH*OW %% tile == 0)", bm, bn,
                       nstb, 2 * wm, d.split_parts, f_sp);
            f_st = 0;
        }
        if (f_st >= 200) {
            lw_nst = f_st - 200;
            f_st = 0;
        } else if (f_st >= 100) {
            ws_nst = f_st - 100;
            f_st = 0;
        }
        if (f_bm && !(f_bm == 32 && !g_force_bm)) {   // (a hinted 32-row "tile" is an operand-stationary entry that fell back)
            BM = f_bm;
            BN = f_bn;
            nst = f_st > 0 ? f_st
                           : (BM == 256 ? (d.split_parts == 3 ? 2 : 3)
                                        : (BM == 128 && BN == 128 ? (d.split_parts == 3 ? 3 : 4)
                                                                  : (BM == 64 && BN == 64 ? 3 : 4)));
            ALDM_CHECK(igemm_dma_config_ok(BM, BN, nst, d.split_parts), "aldm_igemm: no DMA kernel for tile %dx%d, %d stages", BM, BN, nst);
            if (f_sp > 0 && can_split && nk / f_sp >= 1) splits = f_sp;
        } else {
            const int cand[5][3] = {{256, 128, d.split_parts == 3 ? 2 : 3}, {128, 128, d.split_parts == 3 ? 3 : 4},
                                    {64, 128, 2}, {128, 64, 2}, {64, 64, 2}};
            static const int sps[8] = {1, 2, 3, 4, 6, 8, 12, 16};
            double best = 1e300;
            BM = BN = 64;
            nst = 3;
            for (int c = 0; c < 5; ++c) {
                const int bm = cand[c][0], bn = cand[c][1];
                if (geglu ? bn != 128 : (bn > 64 && d.N <= 64)) continue;
                if (qkv && d.qkv_c % bn != 0) continue;
                if (bm == 256 && Mz < 4096) continue;
                for (int si = 0; si < 8; ++si) {
                    const int sp = sps[si];
                    if (sp > 1 && (!can_split || !have_ws || nk / sp < 3)) break;
                    int spe;
                    const double t = dma_cost(bm, bn, cand[c][2], sp, &spe);
                    if (spe > 1 && (int64_t)spe * Mz * d.N > d.ws_floats) continue;
                    if (t < best) {
                        best = t;
                        BM = bm;
                        BN = bn;
                        nst = cand[c][2];
                        splits = spe;
                    }
                }
            }
        }
        ALDM_CHECK(!geglu || BN == 128, "aldm_igemm: the GEGLU epilogue needs a 128-column tile");
        ALDM_CHECK(!qkv || d.qkv_c % BN == 0, "aldm_igemm: the QKV epilogue needs a tile width dividing qkv_c");
        if (qkv) splits = 1;
        p.tiles_m = cdiv(p.M, BM);
        p.tiles_n = cdiv(d.N, BN);
        p.kt_per_split = cdiv(nk, splits);
        splits = cdiv(nk, p.kt_per_split);
        if (splits > 1 && (!have_ws || d.ws_floats < (int64_t)splits * p.M * d.N)) {
            splits = 1;
            p.kt_per_split = nk;
        }
        p.splits = splits;
        p.kgroups = 1;
        p.bx = 1;
        p.pre = pre;
        p.dma = 1;
        p.nst = nst;
        p.ws = 0;
        p.ws_blocks = 0;
        if (lw_nst > 0) {
            ALDM_CHECK(igemm_dma_lw_config_ok(BM, BN, lw_nst, d.split_parts), "aldm_igemm: no loader-wave DMA kernel for tile %dx%d, %d stages", BM, BN, lw_nst);
            p.ws = 2;
            p.nst = lw_nst;
        }
        if (ws_nst > 0) {
            // hinted / forced persistent form; a launch the persistent kernel cannot run (tuned tables are keyed by geometry,
            // not by epilogue flags) keeps the tile on igemm_dma_kernel with that tile's default ring
            if (splits == 1 && d.a_fmt == ALDM_FMT_BF16 && d.out_split_parts == d.split_parts && d.out_split_fmt == ALDM_FMT_BF16 &&
                igemm_dma_ws_config_ok(BM, BN, ws_nst, d.split_parts) && dma_ws_eligible(p, BM, BN)) {
                const int ntiles = p.tiles_m * p.tiles_n;
                const int per = cdiv(ntiles, device_cus() * igemm_dma_ws_blocks_per_cu(BM, BN, ws_nst, d.split_parts));
                p.ws = 1;
                p.nst = ws_nst;
                p.ws_blocks = cdiv(ntiles, per);
            } else {
                ALDM_CHECK(!g_force_bm, "aldm_igemm: the persistent DMA kernel cannot run this launch (tile %dx%d, %d stages)", BM,
                           BN, ws_nst);
            }
        }
        return 0;
    }
    const bool bx_ok = d.w_split != nullptr && d.split_parts == 3 && d.b_mode == ALDM_B_PACKED && d.stride_w == 0 && d.batch == 1 &&
                       pre != PRE_GENERIC && (reinterpret_cast<uintptr_t>(d.w_split) & 15) == 0 &&
                       (g_force_mma == 2 || (g_force_mma == 0 && d.hint_mma != 1));
    auto occ_of = [&](int bm, int bn) {
        if (bx_ok && bn != 32) return bm * bn >= 128 * 128 ? 1 : (bm * bn >= 64 * 128 ? 2 : 3);
        return bm * bn >= 128 * 128 ? 2 : ((bm * bn >= 64 * 128 || bn == 32) ? 3 : 4);
    };
    const double pre_w = (d.pre_scale != nullptr || d.pre_act != ALDM_ACT_NONE) ? 1.5 : 1.0;
    auto cost_of = [&](int bm, int bn, int sp, int kg, int* sp_eff) -> double {
        const bool bx = bx_ok && bn != 32;
        const double mf = bx ? 0.375 : 1.0;
        const int kt = cdiv(nk, sp);
        sp = cdiv(nk, kt);
        *sp_eff = sp;
        const double blocks = (double)cdiv64(Mz, bm) * cdiv(d.N, bn) * d.batch * sp;
        const double L = (double)kt * bm * bn / 4.0 * mf;
        const int o = kg == 2 ? (bx ? 1 : 2) : occ_of(bm, bn);  // 512-thread blocks: two per CU (one with BX)
        // BX staging: + the operand split, but the 8-wave 128x128 tile halves the per-thread share and its
        // interleaved schedule hides most of it behind the MFMAs (profiles/r01_mma_ab.txt)
        const double S = (300.0 * (bm / 32) * pre_w + 200.0 * (bn / 32)) *
                         (bx ? (bm * bn >= 128 * 128 && !geglu ? 0.45 : 1.3) : 1.0);
        const double F = 4000.0 + (kg == 2 ? 600.0 : 0.0);
        const int64_t nb = (int64_t)((blocks + 255.0) / 256.0);
        const int64_t full = nb / o, last = nb - full * o;
        // critical path of one wave group: it multiplies every kg-th k-tile
        const double one = (double)cdiv(kt, kg) * ((double)bm * bn / 4.0 * mf + S) + F;
        double T = full * std::max(one, o * L);
        if (last) T += std::max(one, last * L);
        if (sp > 1) T += 10000.0 + (double)(sp + 1) * Mz * d.N * d.batch * 4.0 / 2000.0;
        return T;
    };
    const bool can_split = d.N % 4 == 0 && nk >= 8 && !geglu;
    const bool have_ws = d.ws != nullptr && (reinterpret_cast<uintptr_t>(d.ws) & 15) == 0;
    int splits = 1, kgroups = 1;
    // explicit choice: thread-local override (tools/tests) first, then the descriptor's tuned hint
    const int f_bm = g_force_bm ? g_force_bm : d.hint_bm, f_bn = g_force_bm ? g_force_bn : d.hint_bn;
    const int f_sp = g_force_bm ? g_force_splits : d.hint_splits;
    const int f_kg = g_force_bm ? g_force_kgroups : d.hint_kgroups;
    if (f_bm) {
        BM = f_bm;
        BN = f_bn;
        ALDM_CHECK(tile_supported(BM, BN), "aldm_igemm: unsupported forced/hinted tile %dx%d", BM, BN);
        ALDM_CHECK(!geglu || BN == 128, "aldm_igemm: the GEGLU epilogue needs a 128-column tile");
        if (f_sp > 0 && can_split && nk / f_sp >= 1) splits = f_sp;
        if (f_kg == 2) {
            ALDM_CHECK(BM == 64 && BN == 64, "aldm_igemm: two wave groups exist for the 64x64 tile only");
            kgroups = 2;
        }
    } else if (d.N <= 32 && !geglu) {
        BM = 128;
        BN = 32;
    } else {
        static const int cand[4][2] = {{128, 128}, {64, 128}, {128, 64}, {64, 64}};
        static const int sps[8] = {1, 2, 3, 4, 6, 8, 12, 16};
        double best = 1e300;
        BM = 64;
        BN = 64;
        for (int c = 0; c < 4; ++c) {
            const int bm = cand[c][0], bn = cand[c][1];
            if (geglu ? bn != 128 : (bn > 64 && d.N <= 64)) continue;  // 
            for (int si = 0; si < 8; ++si) {
                const int sp = sps[si];
                if (sp > 1 && (!can_split || !have_ws || nk / sp < 3)) break;
                for (int kg = 1; kg <= ((bm == 64 && bn == 64) ? 2 : 1); ++kg) {
                    int spe;
                    const double t = cost_of(bm, bn, sp, kg, &spe);
                    if (spe > 1 && (int64_t)d.batch * spe * Mz * d.N > d.ws_floats) continue;
                    if (kg == 2 && cdiv(nk, spe) < 4) continue;
                    if (t < best) {
                        best = t;
                        BM = bm;
                        BN = bn;
                        splits = spe;
                        kgroups = kg;
                    }
                }
            }
        }
    }
    ALDM_CHECK(!geglu || BN == 128, "aldm_igemm: internal: GEGLU epilogue without a 128-column tile");
    p.tiles_m = cdiv(p.M, BM);
    p.tiles_n = cdiv(d.N, BN);
    if (splits < 1) splits = 1;
    p.kt_per_split = cdiv(nk, splits);
    splits = cdiv(nk, p.kt_per_split);  // no empty split
    if (splits > 1) {
        const int64_t need = (int64_t)d.batch * splits * p.M * d.N;
        if (!have_ws || d.ws_floats < need) {
            splits = 1;
            p.kt_per_split = nk;
        }
    }
    p.splits = splits;
    p.kgroups = kgroups;
    p.bx = bx_ok && BN != 32 ? 1 : 0;
    p.pre = pre;
    return 0;
}

extern "C" int aldm_igemm_plan(const aldm_igemm_desc* dd, int* bm, int* bn, int64_t* flops, int* splits,
                               int* kgroups, int* mma) {
    IgemmK p;
    int BM, BN;
    const int rc = igemm_prepare(dd, p, BM, BN);
    if (rc) return rc;
    if (bm) *bm = BM;
    if (bn) *bn = BN;
    if (flops) *flops = 2ll * p.M * p.d.N * p.d.K * p.d.batch;
    if (splits) *splits = p.splits;
    if (kgroups) *kgroups = p.kgroups;
    if (mma) *mma = p.bx ? ALDM_MMA_BF16X6 : ALDM_MMA_F32;
    return 0;
}

extern "C" int aldm_igemm_plan_stages(const aldm_igemm_desc* dd) {
    IgemmK p;
    int BM, BN;
    if (igemm_prepare(dd, p, BM, BN)) return -1;
    if (p.dma && p.ws == 4) return 400 + (BM == 128 && p.ws_blocks == 4 ? 10 : 0) + p.nst;
    return p.dma ? p.nst + 100 * p.ws : 0;
}

extern "C" int64_t aldm_igemm_ws_floats(const aldm_igemm_desc* dd) {
    if (!dd) return 0;
    // ask with an "infinite" dummy workspace to learn the split the heuristic wants
    aldm_igemm_desc t = *dd;
    t.ws = reinterpret_cast<float*>(uintptr_t(16));
    t.ws_floats = INT64_MAX;
    IgemmK p;
    int BM, BN;
    if (igemm_prepare(&t, p, BM, BN)) return 0;
    return p.splits > 1 ? (int64_t)p.d.batch * p.splits * p.M * p.d.N : 0;
}

extern "C" int aldm_igemm(const aldm_igemm_desc* dd, void* stream) {
    IgemmK p;
    int BM, BN;
    int rc = igemm_prepare(dd, p, BM, BN);
    if (rc) return rc;
    const aldm_igemm_desc& d = p.d;
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n), (unsigned)p.splits, (unsigned)d.batch);
    hipStream_t st = (hipStream_t)stream;
    const int pre = p.pre;
    // every block tile inside one sample -> scale/shift loaded once per k-tile (UNI kernels)
    const bool uni = p.OHW % BM == 0;
    // 8 waves (512 threads, 4 waves/SIMD at 2 blocks/CU) on a tile: measured 1-7 % faster than the 4-wave
    // 128x128 tile for the GroupNorm(+SiLU) prologue convs (profiles/r01_w8_ab.txt), a wash or worse without a
    // prologue.  ALDM_IGEMM_W8 = bit mask of the tiles that use it (1: 128x128, 2: 64x128, 4: 128x64; bit 3:
    // also launches without a GroupNorm prologue, 128x128 only); default 1, aldm_igemm_wave8_mask overrides.
    const int env_w8 = g_wave8_mask < 0 ? default_wave8_mask() : g_wave8_mask;
    const bool gn_pre = pre == PRE_AFFINE || pre == PRE_AFFINE_SILU;
    const int tile_bit = (BM == 128 && BN == 128) ? 1 : (BM == 64 && BN == 128) ? 2 : (BM == 128 && BN == 64) ? 4 : 0;
    // BX: the 128x128 image leaves room for one block per CU, so that tile takes 8 waves for every prologue
    const bool w8 = p.kgroups == 1 && d.epi_mode != ALDM_EPI_GEGLU && (env_w8 & tile_bit) != 0 &&
                    (p.bx ? tile_bit == 1 : (gn_pre || (tile_bit == 1 && (env_w8 & 8) != 0)));
#ifdef ALDM_TEST_HOOKS
    if (g_debug_drop_product.load(std::memory_order_relaxed))   // test hook: only the classic DMA-fed 64x128 tile has a 5-product form
        ALDM_CHECK(p.dma && p.ws == 0 && BM == 64 && BN == 128 && p.nst == 2 && d.split_parts == 3,
                   "aldm_igemm: aldm_debug_drop_product is set: force the classic DMA-fed 64x128 tile with 2 stages and 3-part images "
                   "(got dma=%d form=%d tile %dx%d, %d stages, %d parts) or clear the switch", p.dma, p.ws, BM, BN, p.nst, d.split_parts);
#endif
    if (p.dma && p.ws == 4) {
        rc = igemm_launch_dma_halo(BM, BN, p.nst, p.ws_blocks, d.split_parts, d.a_fmt == ALDM_FMT_F16, p.os_rows, grid, st, p);
    } else if (p.dma && p.ws == 3) {
        rc = igemm_launch_dma_os(d.K / 32, p.nst, d.split_parts, d.a_fmt == ALDM_FMT_F16, grid, st, p);
    } else if (p.dma && p.ws == 2) {
        rc = igemm_launch_dma_lw(BM, BN, p.nst, d.split_parts, d.a_fmt == ALDM_FMT_F16, grid, st, p);
    } else if (p.dma && p.ws) {
        rc = igemm_launch_dma_ws(BM, BN, p.nst, d.split_parts, p.ws_blocks, st, p);
    } else if (p.dma) {
        rc = igemm_launch_dma(BM, BN, p.nst, d.split_parts, d.a_fmt == ALDM_FMT_F16, grid, st, p);
    } else if (p.bx) {
        switch (pre) {
            case PRE_NONE: rc = igemm_launch_bx_pre0(BM, BN, p.kgroups, uni, w8, grid, st, p); break;
            case PRE_AFFINE: rc = igemm_launch_bx_pre1(BM, BN, p.kgroups, uni, w8, grid, st, p); break;
            case PRE_AFFINE_SILU: rc = igemm_launch_bx_pre2(BM, BN, p.kgroups, uni, w8, grid, st, p); break;
            default: rc = igemm_launch_bx_pre3(BM, BN, p.kgroups, uni, w8, grid, st, p); break;
        }
    } else
    switch (pre) {
        case PRE_NONE: rc = igemm_launch_pre0(BM, BN, p.kgroups, uni, w8, grid, st, p); break;
        case PRE_AFFINE: rc = igemm_launch_pre1(BM, BN, p.kgroups, uni, w8, grid, st, p); break;
        case PRE_AFFINE_SILU: rc = igemm_launch_pre2(BM, BN, p.kgroups, uni, w8, grid, st, p); break;
        case PRE_LRELU: rc = igemm_launch_pre3(BM, BN, p.kgroups, uni, w8, grid, st, p); break;
        default: rc = igemm_launch_pre4(BM, BN, p.kgroups, uni, w8, grid, st, p); break;
    }
    if (rc) {
        set_error("aldm_igemm: no kernel for tile %dx%d", BM, BN);
        return -1;
    }
    if (p.splits > 1) {
        const int64_t total = (int64_t)p.M * (d.N >> 2);
        hipLaunchKernelGGL(igemm_reduce_kernel, dim3((unsigned)cdiv64(total, 256), 1, (unsigned)d.batch),
                           dim3(256), 0, st, p);
    }
    ALDM_LAUNCH_CHECK("aldm_igemm");
    return 0;
}

extern "C" int64_t aldm_split_bytes_parts(int K, int N, int parts) {
    if (K <= 0 || N <= 0 || (parts != 2 && parts != 3)) return 0;
    const int Npad = (N + 31) / 32 * 32;
    const int64_t Ko = 4ll * ((K + 31) / 32);  // k-octets, padded to whole 32-wide k-tiles
    return Ko * parts * Npad * 16;
}
extern "C" int64_t aldm_split_bytes(int K, int N) { return aldm_split_bytes_parts(K, N, 3); }

extern "C" int aldm_pack_split_f16(const float* packed, void* dst, int K, int N, float scale, void* stream) {
    ALDM_CHECK(packed && dst && K > 0 && N > 0 && scale > 0.0f, "aldm_pack_split_f16: bad args");
    ALDM_CHECK(((reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0,
               "aldm_pack_split_f16: operands must be 16-byte aligned");
    const int Npad = (N + 31) / 32 * 32;
    const int Kg = (K + 3) / 4;
    const int Ko = 4 * ((K + 31) / 32);
    const int64_t total = (int64_t)Ko * Npad;
    hipLaunchKernelGGL(pack_split_f16_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 4096)), dim3(256), 0,
                       (hipStream_t)stream, packed, reinterpret_cast<uint4*>(dst), Kg, Npad, Ko, scale);
    ALDM_LAUNCH_CHECK("aldm_pack_split_f16");
    return 0;
}

extern "C" int aldm_pack_split_bf16_parts(const float* packed, void* dst, int K, int N, int parts, void* stream) {
    ALDM_CHECK(packed && dst && K > 0 && N > 0 && (parts == 2 || parts == 3), "aldm_pack_split_bf16: bad args");
    ALDM_CHECK(((reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0,
               "aldm_pack_split_bf16: operands must be 16-byte aligned");
    const int Npad = (N + 31) / 32 * 32;
    const int Kg = (K + 3) / 4;
    const int Ko = 4 * ((K + 31) / 32);
    const int64_t total = (int64_t)Ko * Npad;
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 65535);
    if (parts == 3)
        hipLaunchKernelGGL(pack_split_kernel<3>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, packed,
                           reinterpret_cast<uint4*>(dst), Kg, Npad, Ko);
    else
        hipLaunchKernelGGL(pack_split_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, packed,
                           reinterpret_cast<uint4*>(dst), Kg, Npad, Ko);
    ALDM_LAUNCH_CHECK("aldm_pack_split_bf16");
    return 0;
}
extern "C" int aldm_pack_split_bf16(const float* packed, void* dst, int K, int N, void* stream) {
    return aldm_pack_split_bf16_parts(packed, dst, K, N, 3, stream);
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
