/*
 * aldm_hip.h — C ABI of libaldm_hip.so, the MI355X (gfx950) kernel library behind the
 * AudioLDM2 sampling hot path (DDIM/UNet -> VAE decode -> HiFi-GAN, + STFT/mel).
 *
 * The reference (haoheliu/AudioLDM2) has NO native code: every "kernel" on this path is an
 * ATen op reached from Python.  The entry points below are therefore the op inventory of
 * SURVEY.md §2.3, one C function per fused op; each comment cites the reference call site
 * (file:line under /root/reference/audioldm2/) whose arithmetic the entry point replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (e.g. torch tensors), fp32,
 *     channels-last ("tokens x C": [B, H, W, C] / [B, L, C]); no torch types cross this ABI;
 *   - `stream` is a hipStream_t passed as void*; nothing here synchronises or allocates;
 *   - return value: 0 on success, negative on error; aldm_last_error() gives the message
 *     (thread local).  Host wrappers turn that into a Python RuntimeError, mirroring the
 *     reference's plain-exception convention (e.g. openaimodel.py:858-860 asserts).
 *   - operands, accumulators and every stored tensor are IEEE fp32; HOW an fp32 x fp32 product of a contraction is
 *     evaluated depends on the matrix-core mode: "f32" = v_mfma_f32_32x32x2_f32 (exact fp32 products,
 *     bit-identical to an fmaf chain); "bf16x6" (the DEFAULT) = 6 bf16 MFMA partial products of exact 3-part operand
 *     splits (fp32 grade, 2.4e-7 rms vs fp64); "bf16x3" (opt-in fast mode for launches over pre-split operands and for
 *     attention) = 3 partial products of (hi, mid) parts rounded to nearest: 16 significant bits per operand, 4.4e-6 rms.
 */
#ifndef ALDM_HIP_H
#define ALDM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- library / error ------------------------------------------------------------------ */
int aldm_version(void);              /* ABI version (9), bumped on any struct / entry change */
const char* aldm_last_error(void);   /* message of the last failing call on this thread     */

/* ---- activations usable as prologue (applied to the gathered input) or epilogue -------- */
enum {
    ALDM_ACT_NONE = 0,
    ALDM_ACT_SILU = 1,      /* x*sigmoid(x): util.py:219-221, model.py:33-35                */
    ALDM_ACT_LRELU = 2,     /* leaky_relu(x, slope): hifigan/models.py:98,151,161           */
    ALDM_ACT_TANH = 3,      /* hifigan/models.py:163                                        */
    ALDM_ACT_LOGCLAMP = 4,  /* log(max(x, slope)): audio_processing.py:85-91 (clip 1e-5)    */
    ALDM_ACT_GELU = 5,      /* exact erf GELU: attention.py:44                              */
    ALDM_ACT_GELU_TANH = 6  /* tanh GELU ("gelu_new"): GPT-2 MLP of the sequence generator, sequence_input.py:69 */
};

/* epilogue modes of aldm_igemm.
 * ALDM_EPI_GEGLU: the GEMM computes the 2*C-wide GEGLU projection, but only C columns are
 * stored: out[m, g*32 + l] = (acc[m, g*64 + l] + bias) * gelu_erf(acc[m, g*64 + 32 + l] + bias),
 * i.e. the weight's output channels are interleaved in groups of 32 (value block, gate block)
 * — pack with aldm_pack_weight from a weight whose rows were permuted that way.  N (the packed
 * width, 2*C) must be a multiple of 64, ldo >= N/2, no split-K, rowbias unused; act = ALDM_ACT_GELU_TANH selects the tanh
 * GELU for the gate (the gated-GELU FF of the FLAN-T5 conditioner: wo(gelu_new(wi_0 x) * wi_1 x)), otherwise erf GELU. */
/* ALDM_EPI_QKV (ABI v6, DMA-fed launches only): the fused self-attention projection [q | k | v] = x W^T (N = 3*qkv_c, no
 * bias; attention.py:335-342) whose epilogue hands the attention kernel its operands in the form it multiplies them in:
 *   columns [0, C)    q   -> out, fp32 [M, ldo >= C]
 *   columns [C, 2C)   k   -> k_split: the split image [M][C/32][parts][32] of the k columns (a head = one 32-channel block)
 *   columns [2C, 3C)  v   -> vt_split: v TRANSPOSED per (sample, head, 32-key tile), written straight from the MFMA accumulator
 *                            layout (no LDS transposition): [b][head][tile][part][32 dims][32 keys] bf16, the keys of a tile in
 *                            the order the attention kernel's P^T operand holds them (chunk 2s + lh = accumulator registers 8s ..
 *                            8s + 7 of lane half lh, i.e. tile rows (r & 3) + 8 (r >> 2) + 4 lh).
 * Requires C % 32 == 0, the block tile's width dividing C, qkv_rows (rows per sample) % 32 == 0, no split-K.  The split is the
 * one aldm_attention_d32 applies to fp32 K / V in registers (split_parts = 2: (hi, mid) round to nearest; 3: exact), so
 * aldm_attention_d32_presplit over these images gives bit-identical results with ~30 % fewer VALU instructions per key tile. */
enum { ALDM_EPI_PLAIN = 0, ALDM_EPI_GEGLU = 1, ALDM_EPI_QKV = 2 };

/* B-operand layouts of aldm_igemm */
enum {
    ALDM_B_PACKED = 0, /* weights pre-packed by aldm_pack_weight: [ceil(K/4)][Npad][4]      */
    ALDM_B_NT = 1      /* row-major activations Bmat[N][ldb] (C = A * Bmat^T)               */
};

/*
 * aldm_igemm — the one contraction engine: implicit-GEMM convolution on fp32 MFMA.
 *   out[m, n] = epi( sum_k A[m, k] * W[k, n] ),  m = (b, oh, ow),  k = (kh, kw, ci)
 * A is gathered on the fly from one or two channels-last tensors (x1 ++ x2 along C = the
 * UNet skip concat, openaimodel.py:879), optionally through a virtual nearest-neighbour
 * upsample (openaimodel.py:133, model.py:54), optionally through a per-(sample, channel)
 * affine + activation (= GroupNorm apply + SiLU fused into the conv: openaimodel.py:227-231,
 * model.py:156-158; or leaky_relu for HiFi-GAN).  Zero padding stays zero.
 * Replaces: nn.Conv2d (openaimodel.py:122,172,230,256,267,572,810; attention.py:439,452;
 * model.py:49,139,147,195-203,596,650; autoencoder.py:112), nn.Conv1d / ConvTranspose1d
 * (hifigan/models.py:24-84,114-141), nn.Linear (attention.py:40,54,335-342;
 * openaimodel.py:244-250,537-541), F.conv1d DFT basis (stft.py:67-72), torch.bmm
 * (model.py:219,226), torch.matmul mel basis (stft.py:174).
 */
enum { ALDM_MMA_F32 = 0, ALDM_MMA_BF16X6 = 1 };

typedef struct aldm_igemm_desc {
    /* A operand: gathered input */
    const float* x1;       /* [B, H, W, C1] channels-last                                   */
    const float* x2;       /* optional second tensor, concatenated after x1 along C         */
    int32_t C1, C2;        /* channels of x1 / x2 (C2 = 0 when x2 == NULL); multiples of 4  */
    int32_t pix1, pix2;    /* element pitch between consecutive pixels (0 => C1 / C2)       */
    int32_t B, H, W;       /* stored input extent                                           */
    int32_t up_h, up_w;    /* virtual nearest upsample factors (1 = none)                   */
    int32_t KH, KW, SH, SW, PH, PW, DH, DW;
    int32_t OH, OW;        /* output extent; M = B*OH*OW                                    */
    /* prologue on A */
    const float* pre_scale; /* [B, C1+C2] or NULL: a = a*scale + shift                      */
    const float* pre_shift;
    int32_t pre_act;        /* ALDM_ACT_* applied after the affine                          */
    float pre_slope;
    /* B operand */
    const float* w;
    int32_t b_mode;        /* ALDM_B_PACKED / ALDM_B_NT                                     */
    int32_t ldb;           /* NT: row pitch of Bmat; PACKED: Npad                           */
    int32_t K, N;          /* K = KH*KW*(C1+C2) (multiple of 4), N = output channels        */
    /* epilogue: v = act(acc + bias[n] + rowbias[b, n]); v = alpha*(v + res[m, n]);
       out = accumulate ? out + v : v   (HiFi-GAN: xs += (resblock_j(x))/num_kernels)        */
    const float* bias;     /* [N] or NULL                                                   */
    const float* rowbias;  /* [B, rowbias_ld] or NULL (timestep-embedding add, openaimodel.py:298) */
    const float* res;      /* [Mout, ldo] or NULL (residual)                                */
    float* out;            /* [Mout, ldo]                                                   */
    int32_t ldo;           /* output row pitch in elements (>= N)                           */
    int32_t act;           /* ALDM_ACT_* epilogue                                           */
    float act_slope;
    float alpha;
    int32_t accumulate;
    /* output row remap (polyphase ConvTranspose1d, hifigan/models.py:127-134): when
       out_mul > 0 the GEMM row (b, q) is stored at row b*out_len + q*out_mul + out_off and
       dropped when that position falls outside [0, out_len).  Requires OH == 1.            */
    int32_t out_mul, out_off, out_len;
    /* batched GEMM: blockIdx.z = batch; element strides added per batch (0 = shared)       */
    int32_t batch;
    int64_t stride_x, stride_w, stride_o;
    /* ABI v2 */
    int32_t rowbias_ld;    /* row pitch of rowbias (0 => N): lets every ResBlock read its slice of
                              ONE batched timestep-embedding projection                          */
    int32_t epi_mode;      /* ALDM_EPI_*: 0 = plain; ALDM_EPI_GEGLU fuses attention.py:42-44 (GEGLU)
                              into the projection GEMM — see below                               */
    float* ws;             /* optional split-K workspace (caller-owned scratch, see
                              aldm_igemm_ws_floats); NULL or too small => no split-K            */
    int64_t ws_floats;     /* capacity of ws in floats                                          */
    /* ABI v3: tuned launch configuration for this shape (0 = pick with the built-in cost model):
       block tile hint_bm x hint_bn in {128x128,128x64,64x128,64x64,128x32} and split-K factor.
       audioldm2_amd/tuning/ (JSON) holds the table measured on MI355X (tools/igemm_autotune.py).  */
    int32_t hint_bm, hint_bn, hint_splits;
    int32_t hint_kgroups;  /* 2 = two 4-wave groups per block share the 64x64 tile's K loop (in-block
                              split-K through LDS: no workspace, no reduce kernel); 0/1 = one group */
    /* ABI v4: optional bf16-split image of the packed weights (aldm_pack_split_bf16).  When set (packed
       weights, stride_w == 0) the product runs on the bf16 matrix cores as fp32 = 6 bf16 partial products
       of exact 3-way operand splits with fp32 accumulation ("BF16x6": per-product error 0.7 * 2^-24 on average, <= 2^-21 worst case, i.e. fp32
       grade) instead of the fp32 MFMA.  NULL => fp32 MFMA.                         */
    const void* w_split;
    int32_t hint_mma;      /* tuned table: 1 = fp32 MFMA even when w_split is set, 0 = automatic          */
    int32_t hint_stages;   /* DMA-fed kernel: LDS ring depth (0 = automatic); 100 + depth = the persistent wave-
                              specialised form (one 512-thread block per CU walks a run of tiles: 4 waves run the K
                              loops, 4 waves the previous tile's epilogue) when the launch qualifies: no split-K /
                              activation / row remap / accumulate, whole tiles, 16-byte aligned operands            */
    /* ABI v5: pre-split operands ("split images", see aldm_split_rows).  When a_split is set the A operand is NOT
       gathered from x1/x2 but from the split image of the [B, H, W, C1] input (C1 % 32 == 0, C2 = 0, no prologue:
       normalisation / activation were applied by whoever wrote the image) and both operands go global -> LDS by
       DMA (global_load_lds) with no staging registers and no operand arithmetic in the K loop; requires w_split.
       out_split: the epilogue also (out != NULL) or only (out == NULL) writes its result as a split image with
       out_split_c channels per row — the A operand of the next GEMM (N % 4 == 0, 16-byte aligned out / res).    */
    const void* a_split;
    void* out_split;
    int32_t out_split_c;
    int32_t split_parts;   /* parts of a_split, of out_split and of the w_split the DMA-fed kernel reads: 3 (or 0) =
                              exact 3-way split, 6 partial products ("bf16x6"); 2 = (hi, mid) rounded to nearest, 3
                              partial products ("bf16x3": ~16 significant bits per operand, unbiased).  The register-
                              staged kernels always use the 3-part w_split.                                      */
    /* ABI v6: an activation applied to the SPLIT-IMAGE output only (ALDM_ACT_NONE | ALDM_ACT_LRELU with out_split_slope):
       out = v (fp32, e.g. a HiFi-GAN ResBlock's running sum x + conv(...), hifigan/models.py:96-103) while out_split =
       split(leaky_relu(v)), the pre-activated operand of the next conv — the leaky_relu that the register-staged kernels
       apply in their operand gather.                                                                            */
    int32_t out_split_act;
    float out_split_slope;
    /* ABI v6: ALDM_EPI_QKV outputs (see the enum) */
    void* k_split;
    void* vt_split;
    int32_t qkv_c;         /* C = heads * 32                                                                         */
    int32_t qkv_rows;      /* rows (keys) per sample                                                                  */
    /* ABI v9: "f16x3" operands.  a_fmt = ALDM_FMT_F16: a_split and w_split are 2-part images (split_parts = 2) whose parts are
       IEEE fp16 — hi = RN_f16(s x), lo = RN_f16(s x - hi), s an exact power of two chosen by the producer so that |s x| <= 65504
       holds by construction (GroupNorm / LayerNorm outputs: sqrt(n) max|gamma| + max|beta|; weights: their maximum) — and the fp32
       product is hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16 / 16x16x32_f16: three matrix instructions instead of bf16x6's
       six, 22 of 24 significand bits per operand (measured 0.65 - 0.7x the fp32 MFMA's error against fp64:
       profiles/r06_f16x3_accuracy.txt).  acc_scale = 1 / (s_a s_w) multiplies the accumulators before the epilogue (0 = 1).
       out_split_parts: parts of the split images the EPILOGUE writes (out_split, k_split, vt_split), always bf16; 0 = split_parts.
       DMA-fed launches only (aldm_split_rows_act / aldm_groupnorm_split / aldm_layernorm_split with f16_scale != 0 write such
       images, aldm_pack_split_f16 the weights').                                                                      */
    int32_t a_fmt;
    float acc_scale;
    int32_t out_split_parts;
    /* out_split_fmt = ALDM_FMT_F16: out_split (plain and GEGLU epilogues; not k_split / vt_split) is written as the 2-part fp16
       image of out_split_scale * result — the A operand of a following f16x3 launch.  The caller vouches for |out_split_scale *
       result| <= 65504 (e.g. the GEGLU output of a LayerNorm-fed projection: (R c + b)^2 with R the row 2-norm bound of the
       normalised rows, c the largest column 2-norm of the weight, b the largest |bias|).                                  */
    int32_t out_split_fmt;
    float out_split_scale;
    /* ALDM_EPI_QKV with out_split_fmt = ALDM_FMT_F16: k_split is the fp16 image of out_split_scale * k, vt_split of vt_scale * v
       (aldm_attention_d32_presplit_f16 reads them)                                                                       */
    float vt_scale;
} aldm_igemm_desc;

int aldm_igemm(const aldm_igemm_desc* d, void* stream);
/* Floats of split-K workspace this descriptor would use (0 = the launch fills the chip without
 * splitting K).  Deep UNet levels have M = B*H*W of only 1024..4096 rows: their K loop is split
 * over blockIdx.y, partial tiles go to ws and a second kernel reduces them in a fixed order and
 * applies the epilogue (deterministic, no atomics).                                            */
int64_t aldm_igemm_ws_floats(const aldm_igemm_desc* d);
/* Host-only query (no launch): the block tile / split-K factor aldm_igemm would pick for this
 * descriptor and the algorithmic FLOPs of the call (2*M*N*K*batch) — used by bench.py's roofline
 * accounting.  splits / kgroups may be NULL.                                                              */
enum { ALDM_FMT_BF16 = 0, ALDM_FMT_F16 = 1 };
int aldm_igemm_plan(const aldm_igemm_desc* d, int* bm, int* bn, int64_t* flops, int* splits, int* kgroups,
                    int* mma);   /* mma: ALDM_MMA_* the launch would run on (may be NULL) */
/* Host-only query: LDS ring depth of the DMA-fed kernel this descriptor launches (a_split set), 0 for the register-
 * staged kernels, negative on an invalid descriptor.                                                            */
int aldm_igemm_plan_stages(const aldm_igemm_desc* d);
/* Tuning override (tests / tools): force the block tile and split-K factor of subsequent
 * aldm_igemm calls on this thread; bm = 0 => automatic.  bm x bn in {128x128,128x64,64x128,64x64,128x32}. */
void aldm_igemm_force(int bm, int bn, int splits, int kgroups);
/* ... and the LDS ring depth of the DMA-fed kernel (a_split descriptors): 128x128 {2,3}, 64x128 / 128x64 {2,4},
 * 64x64 {2,3}; 0 = default for the tile; 100 + depth = the persistent wave-specialised kernel (64x128 {3,4,5}, 128x64
 * {3,4}, 64x64 {4,6} on 2-part images; 64x128 / 128x64 {2,3}, 64x64 {3,4} on 3-part images); 200 + depth = the loader-wave
 * form; 300 + depth = the OPERAND-STATIONARY kernel for short K (csrc/igemm_dma_os.h: 1x1 / linear launches with K = 256 or
 * 384, the weight slab of a 128-column block held in registers, A streamed in 32-row stages of the whole K; depth 2-3 on
 * 3-part images (K = 384: 2), 2-4 on 2-part ones; 300 = the deepest ring that fits; forced with bm = 32, bn = 128) — a forced
 * launch one of them cannot run fails, a hinted one falls back to aldm_igemm's own choice.  aldm_igemm_force() resets it to 0. */
void aldm_igemm_force_stages(int stages);
/* Tuning override (tests / tools): bit mask of the block tiles that run with 8 instead of 4 wavefronts per
 * tile on this thread (1: 128x128, 2: 64x128, 4: 128x64 — GroupNorm-prologue launches; 8: 128x128 launches
 * without a prologue too).  mask < 0 restores the default (1, or $ALDM_IGEMM_W8).  Returns the mask in force. */
int aldm_igemm_wave8_mask(int mask);
/* Matrix-core path override (tests / tools) for this thread: 0 = automatic (bf16-split when the descriptor
 * carries w_split and the tuned hint allows), 1 = fp32 MFMA always, 2 = bf16-split wherever an instantiation
 * exists.  Returns the previous mode; other values only query.                                            */
int aldm_igemm_mma(int mode);
#ifdef ALDM_TEST_HOOKS
/* TEST HOOK — NOT in libaldm_hip.so (ABI v9).  It exists only in the variant library libaldm_hip_testhooks.so, which the build
 * compiles from the same sources with -DALDM_TEST_HOOKS and which tests/test_dma_gpu.py loads in a subprocess through
 * $ALDM_LIB_PATH.  Process wide: on != 0 makes DMA-fed launches leave out the smallest of the six bf16 partial products (hi_a x
 * lo_w) — a deliberately broken "5-product" GEMM, ~1e-5 off — so that the test can show that the fp32-grade tolerance WOULD catch
 * a kernel that silently lost a product.  Only the classic 64x128 tile with 2 stages and 3-part images has
 * that instantiation: while the switch is on every other aldm_igemm launch FAILS (nothing runs at full precision by accident).
 * Returns the previous setting.                                                                                          */
int aldm_debug_drop_product(int on);
#endif
/* bf16-split image of a packed weight [ceil(K/4)][Npad][4] (aldm_pack_weight / aldm_pack_kn output) for
 * aldm_igemm_desc.w_split: [4*ceil(K/32) k-octets][3 parts][Npad][8 bf16], w = hi + mid + lo exactly.
 * aldm_split_bytes = size of that image.                                                                 */
int64_t aldm_split_bytes(int K, int N);
int aldm_pack_split_bf16(const float* packed, void* dst, int K, int N, void* stream);
/* the same with `parts` = 2 | 3 parts per weight ([k-octet][parts][Npad][8 bf16]; 2 = (hi, mid) rounded to nearest,
 * the "bf16x3" image of the DMA-fed kernel)                                                                     */
int64_t aldm_split_bytes_parts(int K, int N, int parts);
int aldm_pack_split_bf16_parts(const float* packed, void* dst, int K, int N, int parts, void* stream);
/* the "f16x3" weight image (ABI v9): [k-octet][2 parts][Npad][8 fp16] of scale * w, scale an exact power of two with
 * |scale * w| <= 65504 (the caller's: 2^floor(log2(32768 / max|w|))); same size as the 2-part bf16 image (aldm_split_bytes_parts(K, N, 2)) */
int aldm_pack_split_f16(const float* packed, void* dst, int K, int N, float scale, void* stream);

/* ---- split images: pre-split activations for the DMA-fed GEMM (aldm_igemm_desc.a_split, ABI v5) -----------------
 * A split image of channels-last fp32 rows [rows, C] (C % 32 == 0) holds every value as its exact 3-way truncation
 * split x = hi + mid + lo (bf16 bit patterns), blocked [row][C/32][part][32]: 6 bytes per element, 192 contiguous
 * bytes per (row, 32-channel block); with parts = 2 ("bf16x3") only (hi, mid), both rounded to nearest: 4 bytes per
 * element, 128 bytes per block.  Producers: aldm_split_rows (below), aldm_layernorm_split,
 * aldm_attention_d32_split and aldm_igemm's out_split.                                                            */
int64_t aldm_split_image_bytes(int64_t rows, int C, int parts);
/* dst = split(act(x*scale[b, c] + shift[b, c])) with x = x1 ++ x2 along C ([rows, C1] / [rows, C2], P rows per
 * sample; scale/shift [rows/P, C1+C2] or NULL = no affine; act = ALDM_ACT_NONE | ALDM_ACT_SILU): GroupNorm apply +
 * SiLU (openaimodel.py:227-231,280-283; attention.py:459) and the skip concat (openaimodel.py:879) done ONCE per
 * element instead of once per conv tap inside the GEMM.  dst_raw (optional): split(x) of the same rows (the 1x1 skip
 * conv's operand, openaimodel.py:267).                                                                            */
int aldm_split_rows(const float* x1, const float* x2, int C1, int C2, int64_t rows, int P, const float* scale,
                    const float* shift, int act, void* dst, void* dst_raw, int parts, void* stream);
/* the same with act = ALDM_ACT_LRELU(slope) allowed as well (ABI v6): the leaky_relu in front of every HiFi-GAN conv
 * (hifigan/models.py:98, 151) applied once while writing the operand image                                          */
int aldm_split_rows_act(const float* x1, const float* x2, int C1, int C2, int64_t rows, int P, const float* scale,
                        const float* shift, int act, float slope, void* dst, void* dst_raw, int parts, void* stream);
/* "f16x3" producers (ABI v9; aldm_igemm_desc.a_fmt): dst is the 2-part IEEE-fp16 image of f16_scale * value — hi = RN_f16, lo =
 * RN_f16 of the remainder, 128 bytes per (row, 32-channel block) like a 2-part bf16 image — where f16_scale is an exact power of two
 * under which the CALLER guarantees |f16_scale * value| <= 65504 (a GroupNorm output cannot exceed sqrt(elements per group) *
 * max|gamma| + max|beta|, a LayerNorm output sqrt(C) * max|gamma| + max|beta|, SiLU only shrinks; values beyond are clamped).
 * dst_raw stays a bf16 image with raw_parts parts: raw activations have no a-priori bound.                                  */
int aldm_split_rows_f16(const float* x1, const float* x2, int C1, int C2, int64_t rows, int P, const float* scale,
                        const float* shift, int act, float slope, void* dst, void* dst_raw, int raw_parts, float f16_scale,
                        void* stream);

/* Pack a weight for ALDM_B_PACKED.  src is the PyTorch layout:
 *   conv:      [N, Cin, KH, KW] (Conv2d / Conv1d with KH = 1),  linear: KH = KW = 1
 *   transposed=1: ConvTranspose1d layout [Cin, N, KW]; phase/stride select the polyphase
 *   taps kw = phase + j*stride (j = 0..T-1, T = ceil(KWfull/stride)), stored flipped so the
 *   phase runs as an ordinary conv with PW = T-1 (polyphase decomposition, ops.pack_convtr1d).
 * dst has ceil(K/4) * Npad * 4 floats, Npad = round_up(N, 32), K = KH*KWeff*Cin.           */
int aldm_pack_weight(const float* src, float* dst, int N, int Cin, int KH, int KW,
                     int transposed, int phase, int stride, void* stream);
/* [K, N] row-major activations -> packed [ceil(K/4)][Npad][4] (P*V of the VAE attention)   */
int aldm_pack_kn(const float* src, float* dst, int K, int N, int lds, int batch,
                 int64_t stride_src, int64_t stride_dst, void* stream);

/* ---- normalisation -------------------------------------------------------------------- */
/* GroupNorm statistics over channels-last x = x1 ++ x2 -> per-(b, c) scale/shift such that
 * GroupNorm(x)[b, p, c] = x*scale[b, c] + shift[b, c]  (scale = rstd*gamma,
 * shift = beta - mean*rstd*gamma).  util.py:224-241 (eps 1e-5), attention.py:75-78 and
 * model.py:38-41 (eps 1e-6).  ws: scratch of aldm_gn_ws_floats() floats.
 * Statistics are accumulated about a per-thread pivot and merged with the parallel-variance formula in fp64 (fixed
 * order), so |mean| >> std inputs keep their variance (ATen: Welford). */
int aldm_groupnorm_stats(const float* x1, const float* x2, int B, int P, int C1, int C2,
                         int G, float eps, const float* gamma, const float* beta,
                         float* scale, float* shift, float* ws, void* stream);
int64_t aldm_gn_ws_floats(int B, int P, int C, int G);
/* GroupNorm + activation (ALDM_ACT_NONE | ALDM_ACT_SILU) + operand split in one call (ABI v6): dst = split(act(GroupNorm(x1 ++
 * x2))) as a split image with `parts` parts, dst_raw (optional) = split(x1 ++ x2) — the image aldm_groupnorm_stats followed by
 * aldm_split_rows writes, bit for bit.  Samples of up to 1024 pixels run as ONE launch (a block owns whole groups: it reads its
 * channel slab for the statistics, then normalises and splits it while it is still in L2); larger ones as the two launches.
 * scale / shift [B, C1+C2] are written as by aldm_groupnorm_stats.                                                       */
int aldm_groupnorm_split(const float* x1, const float* x2, int B, int P, int C1, int C2, int G, float eps,
                         const float* gamma, const float* beta, int act, float* scale, float* shift, float* ws,
                         void* dst, void* dst_raw, int parts, void* stream);
int aldm_groupnorm_split_f16(const float* x1, const float* x2, int B, int P, int C1, int C2, int G, float eps,
                             const float* gamma, const float* beta, int act, float* scale, float* shift, float* ws,
                             void* dst, void* dst_raw, int raw_parts, float f16_scale, void* stream);
/* LayerNorm over the last dim of [M, C] (attention.py:393-395), eps 1e-5                   */
int aldm_layernorm(const float* x, float* y, int M, int C, const float* gamma,
                   const float* beta, float eps, void* stream);
/* same, writing the result as a split image (y_split, C % 32 == 0) and optionally also as fp32 (y may be NULL)    */
int aldm_layernorm_split(const float* x, float* y, void* y_split, int M, int C, const float* gamma,
                         const float* beta, float eps, int parts, void* stream);
int aldm_layernorm_split_f16(const float* x, float* y, void* y_split, int M, int C, const float* gamma,
                             const float* beta, float eps, float f16_scale, void* stream);

/* ---- attention ------------------------------------------------------------------------ */
/* Multi-head attention, head dim 32, flash-style online softmax on fp32 MFMA:
 *   out[b, i, h*32:(h+1)*32] = softmax_j(scale * q_i.k_j  [masked -> -FLT_MAX]) @ v
 * attention.py:343-367.  q/k/v/out are [B, L, *] with independent row pitches so a fused
 * QKV projection buffer can be consumed in place.  mask: [B, Lk] floats (1 = keep) or NULL
 * (attention.py:357-361: masked scores are set to -finfo.max, NOT -inf).                   */
int aldm_attention_d32(const float* q, const float* k, const float* v, float* out,
                       int B, int heads, int Lq, int Lk, int ldq, int ldk, int ldv, int ldo,
                       const float* mask, float scale, void* stream);
/* same, writing the result (also / only: out may be NULL) as a split image with heads*32 channels per row — the
 * pre-split A operand of the to_out projection (attention.py:366)                                               */
int aldm_attention_d32_split(const float* q, const float* k, const float* v, float* out, void* out_split,
                             int parts, int B, int heads, int Lq, int Lk, int ldq, int ldk, int ldv, int ldo,
                             const float* mask, float scale, void* stream);
/* Self-attention over the pre-split K / V^T images an ALDM_EPI_QKV launch wrote (ABI v6): q fp32 [B, Lq, *] (row pitch ldq),
 * k_split / vt_split as described at ALDM_EPI_QKV with `parts` parts, Lk % 32 == 0, no mask; out / out_split as
 * aldm_attention_d32_split.  `parts` must match the attention mode in force (2 for bf16x3, 3 for bf16x6).              */
int aldm_attention_d32_presplit(const float* q, const void* k_split, const void* vt_split, float* out, void* out_split,
                                int parts, int B, int heads, int Lq, int Lk, int ldq, int ldo, float scale, void* stream);
/* Matrix-core path of aldm_attention_d32, PROCESS wide: 1 = fp32 MFMA, 2 = "bf16x6" (both products as 6 bf16 partial
 * products of exact 3-part operand splits), 3 = "bf16x3" ((hi, mid) rounded to nearest, 3 partial products), -1 =
 * default: $ALDM_ATTN_MMA if set, else the engine's $ALDM_MMA, else bf16x6 ("f32" | "bf16x6" | "bf16x3"; anything else is
 * reported on stderr and ignored).  Returns the previous mode; other values only query.                         */
int aldm_attention_mma(int mode);
/* Schedule of aldm_attention_d32_presplit (tests / tools; process wide): -1 = default ($ALDM_ATTN_SCHED, else 1), 0 = the round-3/4
 * pipelined kernel, 1 = the re-scheduled exact-max loop (bitwise the fp32-K/V path), 2 = the one-pass loop with a fixed softmax
 * reference per row (opt-in experiment, 1.7x the error), 3 = K / V^T tiles shared by a block's waves through LDS (ABI v9).
 * Returns the previous setting; other values only query.                                                                  */
int aldm_attention_sched(int sched);
/* "f16x3" self-attention (ABI v9): k_split / vt_split are the 2-part fp16 images an ALDM_EPI_QKV launch with out_split_fmt =
 * ALDM_FMT_F16 wrote (of k_scale * k and v_scale * v); q is split in the kernel into fp16 parts of q_scale * scale * log2(e) * q
 * (the caller vouches for |q_scale * scale * log2(e) * q| <= 65504: a LayerNorm-fed projection is bounded by R c), the
 * probabilities into fp16 parts of 2^15 p; Q.K^T and P.V run hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 (24 matrix
 * instructions per 32-key tile and 64 queries instead of 48).  The softmax reference is the integer ceiling of the running
 * maximum in log2 units, so its offsets are exact.  out_split (optional): a bf16 image with out_parts (2 | 3) parts, or with
 * out_parts = 0 the 2-part fp16 image of out_scale * out (|out| <= max|v|: out_scale = v_scale is always safe) — the operand
 * of a three-product to_out projection.                                                                                   */
int aldm_attention_d32_presplit_f16(const float* q, const void* k_split, const void* vt_split, float* out, void* out_split,
                                    int out_parts, float out_scale, int B, int heads, int Lq, int Lk, int ldq, int ldo,
                                    float scale, float q_scale, float k_scale, float v_scale, void* stream);
/* Windowed relative-position self-attention of the VITS phoneme encoder (phoneme_encoder/attentions.py:239-289,
 * window_size = `window` <= 8, shared heads): per head h (channels [h*d, (h+1)*d), d <= 128)
 *   s[i, j] = (q_i/sqrt(d)).k_j + [|j-i| <= window] (q_i/sqrt(d)).emb_k[j-i+window];  s = -1e4 where mask_i*mask_j == 0;
 *   out_i = softmax_j(s) v + sum_r p[i, i+r-window] emb_v[r].   q/k/v/out: [B, T, *] rows with pitches ld*; emb_*:
 * [2*window+1, d]; mask [B, T] (1 = token).  Exact fp32 FMA (a once-per-prompt conditioner, latency bound).       */
int aldm_rel_attention(const float* q, const float* k, const float* v, float* out, int B, int heads, int T, int d,
                       int ldq, int ldk, int ldv, int ldo, const float* emb_k, const float* emb_v, int window,
                       const float* mask, void* stream);
/* y[r, c] = x[r, c] * s[r] (+ res[r, c] when res != NULL): the x * x_mask of the encoder's conv FFN
 * (attentions.py:406-413) and the final + positional embedding (encoders/modules.py:103); divide != 0:
 * y = x / max(s[r], 1e-12) (F.normalize with s = row norms: CLAP embeddings, clap/open_clip/model.py:745)           */
int aldm_rowscale_add(const float* x, const float* s, const float* res, float* y, int64_t rows, int C, int divide,
                      void* stream);
/* T5LayerNorm (FLAN-T5 conditioner, encoders/modules.py:173-198 via transformers): y = weight * x * rsqrt(mean(x^2) + eps) */
int aldm_rmsnorm(const float* x, float* y, int M, int C, const float* weight, float eps, void* stream);
/* Row softmax of attention scores [B, heads, q_rows, N] with an additive bias [heads, q_rows, N] shared by the batch (T5's
 * bucketed relative-position bias) and a key padding mask [B, N] (0 = padded key: weight 0, the reference adds finfo.min) */
int aldm_softmax_rows_bias(const float* x, float* y, int B, int heads, int q_rows, int N, float scale, const float* bias,
                           const float* keymask, void* stream);
/* row softmax with pre-scale: y = softmax(scale * x) over the last dim of [M, N]
 * (model.py:220-221)                                                                        */
int aldm_softmax_rows(const float* x, float* y, int64_t M, int N, float scale, void* stream);
/* Causal + key-padding masked row softmax for attention score rows laid out [B, heads, q_rows, N] (GPT-2 blocks of the
 * AudioMAE-token sequence generator, audiomae_gen/sequence_input.py:308-323 via transformers GPT2Attention): key j of
 * batch b takes part iff keymask[b, j] != 0 and j <= q_pos0 + i for query row i; excluded keys get weight 0.       */
int aldm_softmax_rows_masked(const float* x, float* y, int B, int heads, int q_rows, int N, float scale,
                             const float* keymask, int q_pos0, void* stream);

/* ---- single-position decode step of the GPT-2 sequence generator (ABI v7) ----------------------------------------------
 * audiomae_gen/sequence_input.py:294-325 calls transformers' GPT2Model once per generated AudioMAE token (512 dependent calls for
 * the speech model, 8 for text-to-audio); with a key/value cache each call is ONE new position per sample: M = batch <= 16 rows
 * through every Linear.  At that M the layers are weight streams, not matrix-core work, and the general path spends its time in
 * launches (~17 per GPT-2 block).  These two entry points are the decode step's layer body in 5 launches per block.
 *
 * aldm_decode_linear: y[M, N] = act(LN(x)[M, K] . W[K, N] + bias) + res — transformers Conv1D (weight stored [in, out] = [K, N],
 *   used as is, fp32) with the preceding LayerNorm (ln_gamma / ln_beta both NULL: none; statistics over the whole row, biased
 *   variance, eps inside the root; K <= 1024 then) and the following activation (ALDM_ACT_NONE | GELU | GELU_TANH | SILU | TANH) /
 *   residual add fused.  1 <= M <= 16; K a multiple of 64 * ceil(K / 1024).  Exact fp32 FMA in a fixed order (deterministic,
 *   no workspace): one block per 32 output columns sums all of K.                                                            */
int aldm_decode_linear(const float* x, int ldx, int M, int K, const float* w_kn, int N, const float* bias,
                       const float* ln_gamma, const float* ln_beta, float ln_eps, int act, const float* res, int ldr,
                       float* y, int ldy, void* stream);
/* aldm_decode_attention: attention of the new position over the key/value cache (transformers GPT2Attention._attn with
 *   layer_past, head dim 64): qkv [B, 3E] = the c_attn output rows (q | k | v, E = heads * 64) of the new position; *pos (a
 *   DEVICE int64, so a captured graph advances it) = its cache slot.  Writes k / v into k_cache / v_cache
 *   [B * heads, n_tot, 64] at slot *pos, scores q.k_j * scale against every key j <= *pos with keymask[b, j] != 0 (the caller
 *   switches slot *pos on first; slots behind it are not part of the sequence yet), softmax, P.V; out [B, E] head-merged.
 *   n_tot <= 1024 (GPT-2's n_positions).                                                                                     */
int aldm_decode_attention(const float* qkv, int ldq, const int64_t* pos, float* k_cache, float* v_cache, const float* keymask,
                          int B, int heads, int n_tot, float scale, float* out, int ldo, void* stream);
/* ---- the same decode step on the whole chip (ABI v9, round 6): 7 launches per GPT-2 block -----------------------------------
 * aldm_decode_linear gives one block all of K for its 32 columns: 24 - 96 blocks per launch, each streaming 96 - 393 KB of
 * weights through one compute unit's memory path.  These entry points cut K into S slices so that every launch covers the chip
 * with one HBM round trip per block, and hand the partial sums to the NEXT launch instead of reducing them across blocks:
 *
 * aldm_decode_gemv_slices(K, N): the S the library uses for a [K, N] weight (K / S rows per slice, a multiple of 32, <= 512;
 *   column tiles x S >= 256 blocks where K allows) — the caller sizes ypart [S, M, N] with it; 0 = K cannot be sliced.
 * aldm_decode_gemv: ypart[s][m][n] = sum over slice s of xe[m][k] W[k][n], W = Conv1D's own [K, N] fp32 weight (non-temporal
 *   loads: read once per token), xe = xact(xbias + x[0] + x[1] + ... + x[xparts-1]) with x[j] = x + j * xstride (partial slabs
 *   [M, ldx] of the previous aldm_decode_gemv, summed in slab order; xparts = 1, xbias = NULL, xact = NONE: plain rows).
 * aldm_decode_reduce_ln: per row m: v = part[0][m] + ... + part[nparts-1][m] (slab order; nparts = 0: none) + bias (row
 *   *bias_row of a [rows, N] table when bias_row != NULL — GPT-2's position embedding at a DEVICE-side position) + res;
 *   h_out = v (may alias res; NULL: not stored); with ln_gamma: xn (and xn2 when != NULL) = LayerNorm(v) gamma + beta
 *   (mean, then the centred second moment, biased variance, eps inside the root).  N <= 1024, N % 4 == 0.
 * aldm_decode_attention_parts: aldm_decode_attention with q | k | v = qbias + the qparts slabs of a sliced c_attn.
 * Deterministic (fixed slab order), no atomics, no workspace besides the slabs the caller owns.                           */
int aldm_decode_gemv_slices(int K, int N);
int aldm_decode_gemv(const float* x, int ldx, int xparts, int64_t xstride, const float* xbias, int xact, int M, int K,
                     const float* w_kn, int N, float* ypart, void* stream);
int aldm_decode_reduce_ln(const float* part, int nparts, int64_t pstride, int ldp, const float* bias, const int64_t* bias_row,
                          const float* res, int ldr, int M, int N, float* h_out, int ldh, const float* ln_gamma,
                          const float* ln_beta, float ln_eps, float* xn, int ldn, float* xn2, int ldn2, void* stream);
int aldm_decode_attention_parts(const float* qkv_part, int ldq, int qparts, int64_t qstride, const float* qbias,
                                const int64_t* pos, float* k_cache, float* v_cache, const float* keymask, int B, int heads,
                                int n_tot, float scale, float* out, int ldo, void* stream);

/* ---- elementwise ---------------------------------------------------------------------- */
/* GEGLU gate: y[m, c] = x[m, c] * gelu_erf(x[m, C + c]), x: [M, 2C] (attention.py:42-44)   */
int aldm_geglu(const float* x, float* y, int64_t M, int C, void* stream);
/* sinusoidal timestep embedding [cos | sin], util.py:172-196; t: [B] floats               */
int aldm_timestep_embedding(const float* t, float* out, int B, int dim, float max_period,
                            void* stream);
/* NCHW <-> NHWC (the reference's rearrange at attention.py:462,465 disappears; only the
 * 8/16-channel latent crosses layouts at the UNet boundary). rep: write `rep` copies.     */
int aldm_nchw_to_nhwc(const float* x, float* y, int B, int C, int HW, int rep, void* stream);
int aldm_nhwc_to_nchw(const float* x, float* y, int B, int C, int HW, void* stream);
/* Fused classifier-free guidance + DDIM step (ddim.py:298-355).  eps holds [e_uncond ; e_cond]
 * ([2, n] when cfg != 0 path is used, else [1, n]); coef (device) = {sqrt(1-a_t), sqrt(a_t),
 * sqrt(1-a_prev-sigma^2), sqrt(a_prev), sigma_t, guidance_scale, use_cfg, 0}.
 *   e = e_u + s*(e_c - e_u); pred_x0 = (x - c0*e)/c1; x_prev = c3*pred_x0 + c2*e + c4*noise  */
int aldm_ddim_step(const float* x, const float* eps, const float* noise, const float* coef,
                   float* x_prev, float* pred_x0, int64_t n, void* stream);
/* The same step with its inputs selected on the DEVICE (ABI v6): coef = coef_tab[*step_idx] (rows of coef_ld >= 7 floats),
 * noise = noise_tab[*step_idx] (rows of n floats), x updated IN PLACE (x_prev overwrites x).  Together with
 * aldm_step_advance it lets one captured HIP graph serve every DDIM step with no host-issued copy between replays.   */
int aldm_ddim_step_indexed(float* x, const float* eps, const float* noise_tab, const float* coef_tab,
                           const int* step_idx, float* pred_x0, int64_t n, int coef_ld, void* stream);
/* *step_idx += 1 and t_cur[0..nt) = t_tab[min(*step_idx, steps-1)] (the next step's timestep row, the UNet's static input;
 * the time_range of ddim.py:205-213 stored as floats, one row per step in loop order)                                  */
int aldm_step_advance(int* step_idx, const float* t_tab, float* t_cur, int nt, int steps, void* stream);
/* Ancestral DDPM step (LatentDiffusion.sample -> p_sample, ddpm.py:357-373,1127-1181), reference
 * operation order: x_recon = a*x - b*eps; mean = c1*x_recon + c2*x; x_prev = mean + s*noise with
 * coef (device) = {sqrt(1/abar_t), sqrt(1/abar_t - 1), posterior_mean_coef1, posterior_mean_coef2,
 * nonzero_mask*exp(0.5*posterior_log_variance_clipped)}.                                        */
int aldm_ddpm_step(const float* x, const float* eps, const float* noise, const float* coef,
                   float* x_prev, int64_t n, void* stream);
/* Inpainting blend, in place on x (ddim.py:226-231, q_sample ddpm.py:430-436):
 *   x = (sa*x0 + so*qnoise)*mask + (1 - mask)*x,  coef (device) = {sqrt(abar_t), sqrt(1 - abar_t)} */
int aldm_inpaint_blend(const float* x0, const float* qnoise, const float* mask, const float* coef,
                       float* x, int64_t n, void* stream);
/* generic y = alpha*a + beta*b (b may be NULL) */
int aldm_axpby(const float* a, const float* b, float* y, float alpha, float beta, int64_t n,
               void* stream);
/* reflect padding of [B, T] by `pad` on both sides into rows of pitch ld_out (stft.py:60-64) */
int aldm_reflect_pad_1d(const float* x, float* y, int B, int T, int pad, int ld_out,
                        void* stream);
/* STFT post: spec [M, ld_spec] = [re(0..F-1) | im(0..F-1)] -> mag [M, ld_mag] (zero padded to
 * ld_mag), phase [M, F] (atan2(im, re)) (stft.py:74-79)                                     */
int aldm_mag_phase(const float* spec, float* mag, float* phase, int64_t M, int F, int ld_spec,
                   int ld_mag, void* stream);

/* out[m] = ||x[m, 0:F]||_2 over rows of pitch ld (energy = torch.norm(magnitudes, dim=1),
 * stft.py:176)                                                                              */
int aldm_row_l2norm(const float* x, float* out, int64_t M, int F, int ld, void* stream);

/* ---- CLAP audio tower glue (re-ranking of n_candidate_gen_per_text candidates, ddpm.py:1554-1568) ---------------------- */
/* torchaudio.functional.resample's polyphase windowed-sinc FIR (encoders/modules.py:700-703): y[b, n*up + i] =
 * sum_j xpad[b, n*down + j] * kernel[i, j] with x zero padded by `width` on the left; kernel [up, taps].           */
int aldm_resample_sinc(const float* x, const float* kernel, float* y, int B, int T, int Tout, int down, int up, int taps,
                       int width, void* stream);
/* |STFT|^2 of rows [re | im] (torchlibrosa Spectrogram, power 2: clap/open_clip/htsat.py:889-897), zero padded to ld_out */
int aldm_power_spec(const float* spec, float* out, int64_t M, int F, int ld_spec, int ld_out, void* stream);
/* y[r, c] = x[r, c]*scale[c] + shift[c]: BatchNorm2d over the mel bins in eval mode (htsat.py:1118-1120)             */
int aldm_col_affine(const float* x, const float* scale, const float* shift, float* y, int64_t rows, int C, void* stream);
/* reshape_wav2img (htsat.py:1064-1090: bicubic stretch of T frames to S*S/mel, align_corners, then the fold into an S x S
 * image) fused with the im2col of the 4x4 / stride-4 PatchEmbed conv: x [B, T, mel] -> out [B, (S/p)^2, p*p]         */
int aldm_bicubic_patchify(const float* x, float* out, int B, int T, int mel, int S, int p, void* stream);
/* y[b, c] = mean over the L tokens of x[b, :, c] (HTSAT "embedding", htsat.py:1034-1035)                             */
int aldm_token_mean(const float* x, float* y, int B, int L, int C, void* stream);
/* out[m] = cos(a[m], b[m]) with F.cosine_similarity's eps clamp (encoders/modules.py:651)                            */
int aldm_row_cosine(const float* a, const float* b, float* out, int M, int C, float eps, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ALDM_HIP_H */
