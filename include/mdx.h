/* mdx.h -- C-ABI of libmdx.so: the MI355X (gfx950) kernels under the minddiffusion
 * UNet-denoising hot path.
 *
 * The reference (mindspore-lab/minddiffusion) has no FFI / operator-plugin interface:
 * its hot path is Python calling stock MindSpore primitives.  Each
 * entry point below therefore replaces one *primitive call site group* of the reference;
 * the file:line it replaces is cited per function (paths under
 * vision/stablediffusionv2/ unless noted).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *  - All pointers are DEVICE pointers (HBM) unless noted; the caller owns every buffer
 *    including workspaces; the library never allocates device memory and never
 *    synchronises -- with ONE exception: the arrival counters of the in-kernel split-K reduce
 *    (16 KiB per distinct workspace address, carved from 1 MiB chunks that are zeroed when they
 *    are allocated; see mdx_gemm_desc.workspace and mdx_gemm_release_counters) for a workspace the
 *    caller has NOT bound counters of its own to (mdx_gemm_bind_counters: with it, no exception).  All work is
 *    enqueued on the caller's stream (hipStream_t passed as void*), so calls are capturable
 *    into a hipGraph.
 *  - Activations are NHWC ("token-major") fp16: [B][H*W][C]; the reference's NCHW fp32
 *    tensors exist only at the apply_model boundary (mdx_nchw_to_nhwc_f16 /
 *    mdx_nhwc_to_nchw_f32).
 *  - Weights are fp16 in a kernel-native packed format (minddiffusion_amd/ops.py: pack_gemm_weight /
 *    pack_conv_weight produce it; it is the only format mdx_gemm_f16 accepts):
 *      1. K order: nn.Dense weight [out][in] as is; nn.Conv2d weight [out][in][kh][kw] becomes
 *         [out][in/64][kh*kw][64] when in % 64 == 0 (channel-chunk major, tap minor: consecutive K tiles touch
 *         the same pixels), else [out][kh*kw][in] (only conv_in, in = 4 padded to 8);
 *      2. N and K zero-padded to multiples of 64 and stored tile-major, [N/64][K/64][64 rows][8 chunks][8 halves],
 *         with the 16-byte chunk at position q of row r holding logical chunk q ^ ((r >> 1) & 7) (the LDS
 *         bank-conflict swizzle, applied once offline), so each DMA instruction reads 1 KiB of contiguous HBM.
 *  - Return value: 0 = ok, negative = error (MDX_E_*); mdx_last_error() returns a
 *    thread-local message.  Nothing throws or exits across the ABI.
 */
#ifndef MDX_H_
#define MDX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDX_OK 0
#define MDX_E_INVALID (-1)     /* bad argument / unsupported shape */
#define MDX_E_WORKSPACE (-2)   /* workspace too small */
#define MDX_E_HIP (-3)         /* HIP runtime error at launch */

typedef void* mdx_stream_t; /* hipStream_t */

int mdx_version(void);
const char* mdx_last_error(void);
/* Tuning / experiment switches of the library.  PROCESS-GLOBAL MUTABLE STATE -- the one place the library departs from "re-entrant
 * per call, no global state" (SURVEY 8(b)): plain ints shared by every thread, stream and device of the process, read when a
 * launch is resolved (so a captured graph holds the values of its capture).  Defaults = what the product runs; they exist for A/B
 * measurements and the tuner.  Set them before the first launch they affect, from one thread; a host that needs two settings at
 * once needs two processes.  The library itself reads NO environment variables.  Names:
 *   gemm_tuned (1)  gemm_bm (0 = auto)  gemm_bn (0)  gemm_ring (0 = auto, 2..5)  gemm_halo (1)  gemm_halo8 (1)
 *   gemm_splitk_fixup_max (4)  gemm_spread (1)  halo_nsb (0 = auto)  gn_min_blocks (512)  gn_fused (1)
 *   attn_fast_stage (1): the attention kernel issues full KV tiles with fixed per-lane offsets + a scalar tile offset
 *   gemm_dense_issue (1): dense launches of the generic GEMM kernel issue their K tiles with a fixed per-lane offset + a scalar K offset
 *   gn_prefetch (1)  gemm_ln_prefetch (1): round-5 "latency diet" -- parameters that are cold in HBM (GroupNorm gamma / beta /
 *   FiLM rows, LayerNorm-fold S[n] and row statistics) are fetched at the top of the kernel instead of behind the dependency
 *   they used to follow; 0 restores the round-4 order for A/B runs (same results bit for bit either way)
 *   gemm_lean_dense (1): round 6 -- dense row-major launches run the lean kernel of csrc/dense.hip (same tile program and bits as the
 *   generic kernel; division-free prologue, the epilogue's global reads prefetched before the K loop); 0 = generic kernel, for A/B runs
 *   (the full list with defaults: csrc/mdx_common.h enum MdxOpt)
 * Unknown names return MDX_E_INVALID. */
int mdx_set_option(const char* name, int value);
int mdx_get_option(const char* name, int* value);

/* ---- layout boundary: LatentDiffusion.apply_model casts/wraps (ldm/models/diffusion/ddpm.py:290-306) */
/* x [B][C][H][W] fp32 -> y [B][H*W][Cpad] fp16, channels C..Cpad-1 zero-filled. */
int mdx_nchw_to_nhwc_f16(const float* x, void* y, int B, int C, int H, int W, int Cpad, mdx_stream_t s);
/* x [B][H*W][Cstride] fp16 -> y [B][C][H][W] fp32 (first C channels). */
int mdx_nhwc_to_nchw_f32(const void* x, float* y, int B, int C, int H, int W, int Cstride, mdx_stream_t s);

/* ---- nn.GroupNorm(32,C,eps) [+ SiLU]  (ldm/modules/diffusionmodules/util.py:87-108,
 *      openaimodel.py:136-137,159-160,521-523; attention.py:83-84 eps 1e-6)
 * Input is the channel-concatenation of x1 [B][HW][C1] and optional x2 [B][HW][C2]
 * (openaimodel.py:568 Concat is never materialised).  Statistics in fp32.
 * y [B][HW][C1+C2] fp16.  gamma/beta fp32 [C1+C2].
 * ws: fp32 workspace of at least mdx_groupnorm_ws_floats(B, HW, C1+C2, groups) floats. */
size_t mdx_groupnorm_ws_floats(int B, int HW, int C, int groups);
int mdx_groupnorm_f16(const void* x1, int C1, const void* x2, int C2, const float* gamma, const float* beta,
                      void* y, int B, int HW, int groups, float eps, int silu, float* ws, mdx_stream_t s);

/* GroupNorm whose statistics come from its producers' column partials (mdx_gemm_desc.colstats_out) instead of a reduction
 * pass over the tensor: ONE launch (normalise + affine [+ SiLU]) and one read of x instead of two launches and two reads.
 * cs1 / cs2: the producers' colstats buffers of x1 / x2 ([B * nrb][C][2] fp32), nrb1 / nrb2 = row blocks per sample
 * (HW / rows per block).  scale / shift: optional FiLM rows as in mdx_groupnorm_scaleshift_f16 (NULL for the plain norm).
 * Deterministic (fixed-order folds, no atomics). */
int mdx_groupnorm_colstats_f16(const void* x1, int C1, const float* cs1, int nrb1, const void* x2, int C2, const float* cs2,
                               int nrb2, const float* gamma, const float* beta, const float* scale, const float* shift,
                               int mod_ld, void* y, int B, int HW, int groups, float eps, int silu, mdx_stream_t s);

/* Two-level fold of a producer's column partials: dst[b][j][c] = sum of the f = ceil(nrb / nrb2) consecutive row blocks
 * j f .. j f + f - 1 of src[b][.][c] (src [B * nrb][C][2], dst [B * nrb2][C][2], nrb2 == ceil(nrb / f)).  Tensors with hundreds of
 * row blocks per sample (Taichu-GLIDE's 128 x 128 / 256 x 256 levels) are folded once to <= 64 blocks and then normalised by
 * mdx_groupnorm_colstats_f16 -- one small launch instead of the statistics pass over the tensor. */
int mdx_colstats_fold_f32(const float* src, int nrb, float* dst, int nrb2, int B, int C, mdx_stream_t s);

/* Same with the FiLM modulation of GLIDE's ResBlock (Taichu-GLIDE/.../unet.py:203-208):
 *   y = silu?( GN(x) * (1 + scale[b][c]) + shift[b][c] ),  scale/shift fp32 rows of stride mod_ld. */
int mdx_groupnorm_scaleshift_f16(const void* x1, int C1, const void* x2, int C2, const float* gamma,
                                 const float* beta, const float* scale, const float* shift, int mod_ld, void* y,
                                 int B, int HW, int groups, float eps, int silu, float* ws, mdx_stream_t s);

/* ---- nn.LayerNorm([C], eps) (attention.py:176-178): rows x C fp16 -> fp16, fp32 statistics. */
int mdx_layernorm_f16(const void* x, const float* gamma, const float* beta, void* y, int rows, int C, float eps,
                      mdx_stream_t s);

/* ---- nn.Conv2d 3x3/1x1 and nn.Dense as one MFMA implicit-GEMM
 *      (openaimodel.py:50,81,138,163,174,352,524; attention.py:44,66,108-112,212,231)
 * out[m][n] = sum_k A[m][k] * W[n][k]  (+bias[n]) (+rowbias[b(m)][n]) (+residual[m][n])
 *   m = (b, yo, xo) output pixel / token;  k = (tap, cin);  A gathered on the fly from the
 *   NHWC source(s) with zero padding, optional stride 2 (Downsample) and optional nearest-2x
 *   upsample of the source (Upsample, openaimodel.py:57) folded into the gather.             */
typedef struct mdx_gemm_desc {
    const void* a;        /* source 1, NHWC fp16 [B][H][W][c1] */
    const void* a2;       /* optional source 2 (channel concat), [B][H][W][c2], or NULL */
    int c1, c2;           /* Cin = c1 + c2 (each a multiple of 8) */
    const void* w;        /* packed weights (format above), logical shape [N][ksize*ksize*Cin] */
    const float* bias;    /* [N] fp32 or NULL */
    const float* rowbias; /* [B][rowbias_ld] fp32 per-sample bias (ResBlock emb add, openaimodel.py:188-200) or NULL */
    int rowbias_ld;
    const void* residual; /* fp16 [M][residual_ld] or NULL (skip / transformer residual adds) */
    int residual_ld;
    void* out;            /* fp16 */
    int out_ld;           /* row stride (elements) of out for out_mode 0; token stride for out_mode 1 */
    int B, H, W;          /* source spatial extent per sample (Dense: H = tokens, W = 1) */
    int N;                /* output channels (multiple of 8; GEGLU: 2x the produced width, interleaved packing) */
    int ksize;            /* 1 or 3 (3 => pad 1) */
    int stride;           /* 1 or 2 */
    int upsample;         /* 1: nearest-2x upsample the source before the conv */
    int epilogue;         /* MDX_EPI_* */
    int out_mode;         /* MDX_OUT_* */
    int splitk;           /* 0 = auto, >=1 = number of K splits */
    void* workspace;      /* split-K workspace, 16-byte aligned (may be NULL when splitk <= 1).  Row-major launches reduce IN the
                             kernel: every (tile, split) block parks its fp32 partial in the workspace and takes a ticket on
                             the tile's arrival counter; the block that completes a tile sums the partials in split order and
                             runs the epilogue (no reduce launch).  The counters are LIBRARY-owned, one set per (device,
                             workspace address), always zero between launches: the workspace needs no initialisation (rounds 2
                             and 3 kept them in the first MDX_GEMM_WS_HEAD bytes of the workspace, which made "zeroed by the
                             caller" -- later "zeroed on first sight of an address" -- part of the contract; both broke on
                             buffers an allocator hands out twice).  The first MDX_GEMM_WS_HEAD bytes stay reserved.  Launches
                             that can run concurrently (two streams) need distinct workspaces, as their partials always did.
                             Transposed-output split launches use [split][M][N] slabs + a reduce
                             launch as before (no counters). */
    size_t workspace_bytes;
    long out_bs;          /* row-major only: element stride between samples (0 = dense); lets a projection write
                             into a token sub-range of a larger [B][tokens][C] buffer (GLIDE text|image keys) */
    void* out2;           /* split output (n_split > 0): columns [0, n_split) go row-major to `out` as usual, columns      */
    int out2_ld;          /* [n_split, N) go TRANSPOSED to out2[(b * (N - n_split) + n - n_split) * out2_ld + tok]:      */
    int n_split;          /* q|k and V^T of a self-attention in ONE launch (attention.py:108-112; Taichu-GLIDE unet.py:289-297 with
                             out_bs: q | k of the image tokens behind the text keys).  Multiple of 128; bias only (no rowbias /
                             residual / epilogue). */
    int asym_pad;         /* 3x3 stride-2 only: zero-pad bottom / right instead of all around (VAE Encoder Downsample,
                             ldm/modules/diffusionmodules/model.py:55-78) */
    /* nn.LayerNorm folded into the two GEMMs around it (BasicTransformerBlock, attention.py:176-185):
     *   LN(x) W^T + b  =  rstd_m * ( x (gamma (.) W)^T  -  mean_m * S )  +  (W beta + b),   S[n] = sum_k (gamma (.) W)[n][k]
     * PRODUCER (the GEMM that writes the token rows x; row-major, N % 64 == 0): stats_out[m][N / 64][2] fp32 receives
     * {sum, sum of squares} of every 64-column slice of the fp16 row it stores.
     * CONSUMER (dense 1x1 GEMM over those rows, K = the producer's N): `w` holds fp16(gamma (.) W), `bias` holds
     * W beta + b, ln_stats = the producer's stats_out, ln_nt = K / 64, ln_s = S computed from the fp16 weights.
     * The correction is applied to the fp32 accumulators (or to the split-K sum) before bias / activation. */
    float* stats_out;
    const float* ln_stats;
    const float* ln_s;
    int ln_nt;
    float ln_eps;
    /* GroupNorm statistics from the producer (openaimodel.py:136,159: every GroupNorm input is a conv / Dense output): when
     * set, the launch also writes, for every ROW BLOCK of its output and every output column n,
     *   colstats_out[(row_block * N + n) * 2 + {0,1}] = {sum, sum of squares} of the fp16 values stored in that column,
     * rows per block = the M tile (a HALO conv tile = one 8x16 pixel patch), 64 for a split-K launch that still uses the
     * separate reduce kernel; mdx_gemm_query reports the number.  Per COLUMN, so that any consumer grouping -- channel
     * concats, groups that are not aligned to the store granule -- can fold them (mdx_groupnorm_colstats_f16).  Plain row-major
     * launches only (no GEGLU / transposed / n_split / LayerNorm fold / out_bs); tokens per sample % rows per block == 0. */
    float* colstats_out;
    int colstats_cap;     /* row blocks colstats_out has room for: a launch that would write more fails (MDX_E_INVALID) */
    int defer_reduce;     /* split-K launches only: write the fp32 slabs and do NOT launch the reduce -- the consumer,
                             mdx_groupnorm_from_splitk_f16, sums them while it normalises (it must run before any other launch
                             reuses the workspace).  mdx_gemm_f16 fails if the launch does not split. */
    int tile_m;           /* 0 = auto (tuned table, then the cost model); 64 | 128 forces the M tile.  For tools/tune_gemm.py,
                             which measures the (tile_m, splitk) candidates of every UNet shape on the device. */
    int tile_n;           /* 0 = auto; 64 | 128 forces the N tile (same purpose; GEGLU always uses 128) */
    /* nn.GroupNorm(32) [+ SiLU] of the conv's INPUT applied inside the conv (openaimodel.py:136-138, 159-163: GroupNorm -> SiLU ->
     * Conv2d): gn_colstats = the column partials the input's producer emitted (mdx_gemm_desc.colstats_out /
     * mdx_st_tail_desc.colstats_out: [B * gn_nrb][Cin][2]), gn_gamma / gn_beta fp32 [Cin].  Every block folds its sample's
     * partials into a per-channel {scale, shift} table and normalises the halo slices in LDS after they land; `a` is then the RAW
     * tensor and no GroupNorm launch runs.  Single-source 3x3 stride-1 convs with Cin %% 64 == 0, Cin <= 640, images larger than
     * 8 x 8, that resolve to the HALO kernel with 64-column tiles (mdx_gemm_query: out7[3] == 1, out7[1] == 64).
     * DENSE launches (ksize = 1; round 5): nn.GroupNorm(32) WITHOUT an activation in front of a Dense / 1x1 conv
     * (SpatialTransformer.norm -> proj_in, attention.py:243-247; GLIDE AttentionBlock.norm -> qkv): set gn_silu = 0.  The block folds
     * the partials into fp16 {scale, shift} tables and applies them to the A fragments between LDS and the matrix pipe (one packed fma
     * per two values).  Single source, Cin %% 64 == 0, Cin <= 2560, tokens per sample %% tile_m == 0 (mdx_gemm_query out7[0]). */
    const float* gn_colstats;
    const float* gn_gamma;
    const float* gn_beta;
    int gn_nrb;
    int gn_silu;
    float gn_eps;
    /* ResBlock skip_connection fused into the block's second conv (openaimodel.py:174, 201-205): when skip_w is set,
     *   out = conv3x3(a) + conv1x1(cat(skip_a, skip_a2)) (+ bias + rowbias + residual ...),
     * the 1x1 conv over the block's RAW input riding on this launch as extra K tiles (one accumulator, one epilogue; `bias` must
     * hold the SUM of the two convs' biases).  skip_a / skip_a2: NHWC fp16 [B][H][W][skip_c1 / skip_c2] (same H, W), channels
     * multiples of 64; skip_w: packed tile-major weights of logical shape [N][skip_c1 + skip_c2].  Single-source 3x3 stride-1
     * convs that resolve to the HALO kernel only (mdx_gemm_query out7[3] == 1); not with w_frag. */
    const void* skip_a;
    const void* skip_a2;
    int skip_c1, skip_c2;
    const void* skip_w;
    int w_frag;           /* 1: `w` is packed in MFMA-FRAGMENT order instead of the tile-major format above -- [N / 32 column tiles]
                             [K / 16 k-steps][64 lanes][8 halves], piece (ct, s)[lane] = W[32 ct + lane % 32][16 s + 8 (lane / 32)
                             + 0..7], K in the conv order of item 1 (ops.pack_conv_weight_frag) -- and the launch streams it
                             HBM -> registers (weight-streaming form of the HALO 3x3 conv for M <= 512: no weight tiles in LDS,
                             one barrier per 64-channel chunk).  3x3 / stride 1 / single source / Cin % 64 == 0 convs that
                             resolve to the HALO kernel only; set tile_m = 128.  Results are bit-identical to the tile-major form. */
    int stages;           /* 0 = auto; 2 .. 6 forces the depth of the LDS ring the K tiles are DMA'd through (same purpose; 64-row
                             tiles up to 6, 128-row tiles up to 5, HALO weight ring 2 | 3); 10 | 11 = depth 2 | 3 with EIGHT
                             waves per block (generic kernel, tile_m = 128 only); 8 | 9 with tile_m = 256 = the eight-wave 3x3 conv
                             core (conv8p.hip; 9: one phase per 32-deep k-step) */
    const void* w_sub;    /* upsample = 1 only, optional: the SUB-PIXEL weights of the nearest-2x + 3x3 conv (openaimodel.py:57-60).
                             For output parity (dy, dx) the conv is a 2 x 2 conv of the low-resolution source with pre-summed taps --
                             rows {y-1+dy, y+dy}: dy = 0 -> {w[0], w[1]+w[2]}, dy = 1 -> {w[0]+w[1], w[2]}; columns likewise -- packed like
                             a conv weight of logical shape [4 N][Cin][2][2], parity-major rows (ops.pack_subpixel_conv_weight).  When set
                             and the shape suits the eight-wave conv core (Cin % 64 == 0, N % 64 == 0, H % 16 == 0, W % 16 == 0, single
                             source) the launch computes 4 Cin instead of 9 Cin products per output on the un-upsampled tensor; otherwise
                             `w` and the upsampling gather are used.  The summed taps are rounded to fp16 once: results differ from the
                             nine-product form by fp16 rounding of the weights (parity-tested against the oracle at the usual 1e-3). */
    /* Cross-attention over a short cached context as the EPILOGUE of its query projection (round 6; BasicTransformerBlock.attn2,
     * attention.py:108, 138-152 with the context keys / values of :119-121 cached per context tensor): when xattn_k is set the launch
     * computes q = a W^T (with the LayerNorm fold if ln_stats is set), keeps each 64-column tile of it -- ONE HEAD: the head dim must be
     * 64 -- on chip, and writes  out[m][64 h ..] = softmax(q_h K_h^T * xattn_scale) V_h  instead of q: the separate attention launch and
     * the fp16 round trip of q disappear, the arithmetic is that of mdx_attention_f16 on the fp16-rounded q (bit-identical).
     *   xattn_k  [B][xattn_cap][N] fp16 (keys, row-major: head h in columns 64 h ..), xattn_vt [B][N][xattn_cap] fp16 (values, transposed),
     *   xattn_len <= 128 keys of xattn_cap rows are attended to.
     * Dense row-major launches only (ksize 1, one source, no epilogue / residual / statistics / n_split / out_bs), N % 64 == 0, tokens per
     * sample % 128 == 0 (or % 64 == 0 with tile_m = 64); set tile_n = 64 and splitk = 1. */
    const void* xattn_k;
    const void* xattn_vt;
    int xattn_len;
    int xattn_cap;
    float xattn_scale;
} mdx_gemm_desc;

#define MDX_GEMM_WS_HEAD 16384 /* reserved bytes at the head of mdx_gemm_desc.workspace (the arrival counters' former home; their size) */
#define MDX_EPI_NONE 0
#define MDX_EPI_GEGLU 1 /* out[m][j] = a * gelu_tanh(g); packed so that each 128-wide N tile = 64 'a' | 64 'gate' cols
                           (attention.py:41-51) */
#define MDX_EPI_GELU 2  /* out = gelu_tanh(acc + bias)  (GLIDE text-transformer MLP, xf.py:52-59; SDv2 text encoder MLP) */
#define MDX_EPI_QUICKGELU 3 /* out = x * sigmoid(1.702 x), x = acc + bias  (Wukong text encoder, WK text_encoder.py:67-74) */
#define MDX_OUT_ROWMAJOR 0   /* out[m * out_ld + n] */
#define MDX_OUT_TRANSPOSED 1 /* out[(b * N + n) * out_ld + tok]: V^T for mdx_attention_f16 */

int mdx_gemm_f16(const mdx_gemm_desc* d, mdx_stream_t s);
/* Frees the library-owned split-K arrival counters (nothing may be in flight); later launches allocate them again.  Every hipGraph
 * captured with a split launch holds the address of its workspace's counters: destroy such graphs first. */
int mdx_gemm_release_counters(void);
/* Hands the counters of ONE workspace (16 KiB per distinct workspace address) back for reuse by the next workspace that needs a set:
 * call it when a workspace is freed, so that a long-lived process that plans at many resolutions does not accumulate them.  Same
 * precondition (nothing in flight, no live graph captured with this workspace).  Returns the number of sets released (0 = the
 * workspace never carried a split launch).  The counters' device is taken from the workspace POINTER, not the calling thread. */
int mdx_gemm_release_workspace(const void* workspace);
/* Caller-owned arrival counters: binds `counters` -- MDX_GEMM_WS_HEAD bytes of ZEROED device memory on the workspace's device,
 * 16-byte aligned, alive and untouched by the caller until mdx_gemm_release_workspace(workspace) -- to the workspace ADDRESS.
 * Launches on a bound workspace take their tickets there, and the library then allocates NOTHING for them: the ownership rule
 * above holds without its exception (what a host that captures graphs on its own allocator wants; minddiffusion_amd/ops.py
 * new_gemm_workspace binds a set to every workspace it creates).  Every launch leaves the counters zero.  Rebinding the same
 * pair is a no-op; binding another set replaces the previous one (nothing may be in flight, no live graph captured with it). */
int mdx_gemm_bind_counters(const void* workspace, void* counters);
/* Bytes of split-K workspace mdx_gemm_f16 wants for this problem under its auto heuristic (0 if none). */
size_t mdx_gemm_workspace_bytes(const mdx_gemm_desc* d);
/* Host-only validation of a descriptor (no launch): MDX_OK or MDX_E_INVALID with mdx_last_error() set. */
int mdx_gemm_check(const mdx_gemm_desc* d);
/* What mdx_gemm_f16 WOULD launch for this descriptor (host only, nothing is launched):
 * out7 = {tile_m, tile_n, splitk, kernel (0 = generic implicit GEMM, 1 = HALO 3x3 conv, 2 = the lean dense kernel), 1 if the choice came from the
 * measured tile table csrc/gemm_tuned.inc, rows per colstats_out row block (0 = this launch cannot produce column
 * statistics), 1 if a split launch reduces in the kernel (no reduce launch follows)}.  The parity tests assert with it that
 * the table rows are hit at the benchmarked shapes. */
int mdx_gemm_query(const mdx_gemm_desc* d, int* out7);
/* First-use tuner for shapes the measured tile table does not list (no reference counterpart; the reference's graph compiler
 * picks its kernels).  Times every distinct launch form the library has for this descriptor (tile_m x tile_n x split-K) on
 * stream `s` -- median of `reps` (1 .. 31, 0 = 5) launches each, after a hipMemsetAsync of `flush` (may be NULL; >= 512 MiB
 * evicts L2 and the Infinity Cache, i.e. weights arrive cold as they do inside a UNet evaluation) -- and returns the fastest in
 * best4 = {tile_m, tile_n, splitk, stages} for the descriptor's override fields; all zero = keep the library's choice (it won, or
 * lost by < 2 %).  us2 (may be NULL) = {library's choice, best} in microseconds.  The CALLER keeps the answer (ops.tune_cache);
 * the library stores nothing.  This is the one entry that SYNCHRONISES (event waits on `s`): not capturable; every trial
 * overwrites d->out.  Descriptors with colstats_out, defer_reduce or w_frag are refused (their consumer / weight packing is
 * tied to the launch form). */
int mdx_gemm_tune(const mdx_gemm_desc* d, mdx_stream_t s, void* flush, size_t flush_bytes, int reps, int* best4, float* us2);

/* GroupNorm fused with the split-K reduce of its producer (the small tensors of the deep UNet levels, where one block
 * normalises a whole (sample, column block) and the conv in front of it is always split): `prod` is the descriptor of a conv /
 * Dense that ran with defer_reduce = 1.  One launch sums the slabs in slab order, applies the producer's bias / time-embedding
 * row / residual, stores the producer's fp16 output, computes the statistics of those fp16 values and writes
 * y = GroupNorm(out) [+ SiLU] -- bit-identical to the reduce kernel followed by mdx_groupnorm_f16.  Returns MDX_E_INVALID when
 * the tensor does not fit the one-block-per-column-block scheme (the caller then must not defer). */
int mdx_groupnorm_from_splitk_f16(const mdx_gemm_desc* prod, const float* gamma, const float* beta, void* y, int groups,
                                  float eps, int silu, mdx_stream_t s);
/* 1 if mdx_groupnorm_from_splitk_f16 can consume this producer (host only). */
int mdx_groupnorm_from_splitk_ok(const mdx_gemm_desc* prod, int groups);

/* ---- CrossAttention core: softmax(q k^T * scale) v, flash-style (attention.py:138-152);
 *      the [b*h, N, N] score tensor of the reference is never materialised.
 * q  : fp16, element (b, i, h*D + d) at q[b*q_bs + i*q_ld + h*D + d]
 * k  : fp16, element (b, j, h*D + d) at k[b*k_bs + j*k_ld + h*D + d]
 * vt : fp16 TRANSPOSED values, element (b, h*D + d, j) at vt[b*vt_bs + (h*D+d)*vt_ld + j];
 *      columns Nk..vt_ld-1 must be finite (zero-filled by the caller)
 * o  : fp16, same addressing as q with o_bs/o_ld.   D in {40, 64, 80, 160}
 *      (SDv2 / GLIDE: 64; Wukong-Huahua num_heads=8: 40 / 80 / 160, WK/configs/v1-inference-chinese.yaml:31). */
int mdx_attention_f16(const void* q, long q_bs, int q_ld, const void* k, long k_bs, int k_ld, const void* vt,
                      long vt_bs, int vt_ld, void* o, long o_bs, int o_ld, int B, int heads, int D, int Nq, int Nk,
                      float scale, mdx_stream_t s);

/* ---- The same attention with the KEY tiles of every (batch, head, 128-query block) item dealt to `kv_splits` blocks
 *      (attention.py:138-152; flash-decoding style).  For launches whose items do not fill the chip once -- the 64 x 64 and
 *      32 x 32 self-attentions of a UNet evaluation at batch 2: 320 / 160 items on 256 CUs -- every split block keeps its own
 *      running (reference, row sum, O), parks the normalised fp16 partial in `ws`, and the last arriver of an item stores
 *      sum_s w_s O_s / sum_s w_s with w_s = l_s 2^(m_s - max m), summed in split order (schedule-independent bits).
 * kv_splits : 0 = the library chooses (mdx_attention_ws_bytes tells how much workspace that needs; option attn_kv_split),
 *             1 = no split (same launch as mdx_attention_f16), 2..8 = that many (each split needs >= 2 key tiles of 64).
 * ws        : device memory, ZERO on the first use, at least mdx_attention_ws_bytes() bytes for kv_splits = 0 or
 *             65536 + items * kv_splits * (128 * D * 2 + 1024) bytes otherwise (items = B * heads * ceil(Nq / 128) <= 16384);
 *             its first 64 KiB hold one arrival counter per item, which every launch leaves at zero -- so one workspace serves all attention launches of a
 *             stream in turn, but never two streams at once.  NULL: no split. */
size_t mdx_attention_ws_bytes(int B, int heads, int D, int Nq, int Nk);
int mdx_attention_splitkv_f16(const void* q, long q_bs, int q_ld, const void* k, long k_bs, int k_ld, const void* vt,
                              long vt_bs, int vt_ld, void* o, long o_bs, int o_ld, int B, int heads, int D, int Nq, int Nk,
                              float scale, int kv_splits, void* ws, size_t ws_bytes, mdx_stream_t s);

/* ---- Row-local fused tail of a SpatialTransformer block: everything BasicTransformerBlock.construct does after the
 *      self-attention core, plus SpatialTransformer's proj_out, as ONE launch (attention.py:151-152 to_out, :177 norm2,
 *      :108 to_q, :138-150 cross-attention over the cached context keys, :183, :178 norm3, :41-70 GEGLU feed-forward, :184,
 *      :231 proj_out, :256 residual):
 *        t1 = attn_out Wo1^T + bo1 + tok;   q2 = LN2(t1) Wq2^T;   o2 = softmax(q2 K_ctx^T scale) V_ctx;
 *        t2 = o2 Wo2^T + bo2 + t1;   h = GEGLU(LN3(t2) W1^T + b1);   t3 = h W2^T + b2 + t2;   out = t3 Wpo^T + bpo + x_in
 *      A block owns `tile_rows` token rows for the whole chain (A operands resident in LDS, weights streamed HBM/L2 -> registers
 *      in MFMA fragment order); none of q2, o2, t1, t2, the [M][4C] GEGLU intermediate or t3 goes through HBM.
 * attn_out / tok / x_in / out: fp16 [B * tokens][C] dense.  ctx_k: fp16 [B][ctx_cap][C] = to_k(context), rows >= ctx_len
 * finite; ctx_vt: fp16 [B][C][ctx_cap] = to_v(context) transposed (what the cached-context GEMMs of the unfused path write).
 * wstream: the six weight matrices packed by minddiffusion_amd/ops.py: pack_st_tail (per wave w of the C / 32 waves, ONE
 *   contiguous stream of 1 KiB pieces in consumption order; piece (matrix, column tile ct, k-step s)[lane][8] =
 *   W[32 ct + lane % 32][16 s + 8 (lane / 32) + 0..7]); mdx_st_tail_stream_bytes(C) bytes.
 * vec: fp32 [16 C] = [bo1 | gamma2 | beta2 | bo2 | gamma3 | beta3 | b1 (8C: a | gate, reference order) | b2 | bpo].
 * colstats_out: optional [B * tokens / tile_rows][C][2] fp32 = per row block and column {sum, sum of squares} of the fp16
 *   values stored to `out` (the layout of mdx_gemm_desc.colstats_out, rows per block = tile_rows), for the next GroupNorm.
 * debug_out (tests only): when non-NULL the launch stops after stage `debug_stage` and writes that stage's [M][C] fp16 rows
 *   there instead of finishing (1 t1, 2 LN2(t1), 3 q2, 4 o2, 5 t2, 6 LN3(t2), 7 t3). */
typedef struct mdx_st_tail_desc {
    const void* attn_out;
    const void* tok;
    const void* x_in;
    void* out;
    const void* ctx_k;
    const void* ctx_vt;
    const void* wstream;
    const float* vec;
    float* colstats_out;
    void* debug_out;
    int debug_stage;
    int B, tokens;        /* M = B * tokens rows; tokens % tile_rows == 0 */
    int C, heads, dim_head;
    int ctx_len, ctx_cap; /* keys used / row capacity of ctx_k (ctx_vt row length); ctx_cap % 8 == 0, <= 96 */
    float scale;          /* dim_head ** -0.5 */
    float ln_eps;
    int tile_rows;        /* 32 | 64 */
    int warm;             /* 0 = no L2 warmer wave (A/B switch), anything else = default */
} mdx_st_tail_desc;
#define MDX_ST_DEBUG_COUNT_BARRIERS 100 /* debug_stage: run the whole chain WITHOUT the L2 warmer wave and write the number of block
                                           barriers the compute waves executed (int32) to debug_out[0] */
/* Barriers in the schedule the warmer wave walks beside the compute waves (hand-mirrored in csrc/stchain.hip): the two must agree
 * or the product launch deadlocks; tests compare them. */
int mdx_st_tail_sched_barriers(int C, int tile_rows);
int mdx_st_head_sched_barriers(int C, int tile_rows);
int mdx_st_tail_f16(const mdx_st_tail_desc* d, mdx_stream_t s);
/* 1 if mdx_st_tail_f16 has a kernel for this shape (host only): C = 320 with 5 x 64 or 8 x 40 heads today. */
int mdx_st_tail_supported(int C, int heads, int dim_head, int tokens_per_sample, int tile_rows);
size_t mdx_st_tail_stream_bytes(int C);

/* ---- Row-local fused head of a SpatialTransformer block, one launch: SpatialTransformer.norm (GroupNorm(32, eps 1e-6), its
 *      statistics folded from the producer's column partials) -> proj_in -> BasicTransformerBlock.norm1 -> attn1.to_q | to_k |
 *      to_v (attention.py:83-84, 212, 241-247, 176, 108-112).
 * x: fp16 [B * tokens][C] (the block's NHWC input); colstats: [B * nrb][C][2] fp32 = per row block and column {sum, sum of
 * squares} of x as its producer emitted them (mdx_gemm_desc.colstats_out / mdx_st_tail_desc.colstats_out), nrb row blocks per
 * sample.  Outputs: tok [M][C] (the token stream = residual of attn1), qk [M][2C] row-major (q | k), vt [B][C][vt_ld] = V
 * transposed -- the operands of mdx_attention_f16.
 * wstream: ops.pack_st_head (per wave: proj_in, to_q, to_k, to_v column tiles in MFMA fragment order);
 * vec fp32 [5C] = [gn gamma | gn beta | b_proj_in | ln1 gamma | ln1 beta].
 * debug_out / debug_stage (tests): 1 GroupNorm(x), 2 tok, 3 LN1(tok). */
typedef struct mdx_st_head_desc {
    const void* x;
    const float* colstats;
    int nrb;
    const void* wstream;
    const float* vec;
    void* tok;
    void* qk;
    void* vt;
    int vt_ld;
    void* debug_out;
    int debug_stage;
    int B, tokens, C;
    float gn_eps, ln_eps;
    int tile_rows;        /* 32 | 64 */
    int warm;             /* 0 = no L2 warmer wave (A/B switch) */
} mdx_st_head_desc;
int mdx_st_head_f16(const mdx_st_head_desc* d, mdx_stream_t s);
int mdx_st_head_supported(int C, int tokens_per_sample, int tile_rows);
size_t mdx_st_head_stream_bytes(int C);

/* ---- timestep_embedding (util.py:111-131): t [M] fp32 -> out [M][dim] fp32 = [cos | sin]. */
int mdx_timestep_embedding_f32(const float* t, float* out, int M, int dim, float max_period, mdx_stream_t s);

/* ---- small-M nn.Dense for the time-embedding MLP and the 22 emb_layers
 *      (openaimodel.py:341-345,150-157): out[M][N] = act_out(act_in(x)[M][K] @ W[N][K]^T + b)
 * x, out fp32; W fp16; b fp32.  act: 0 none, 1 SiLU. */
int mdx_dense_small_f32(const float* x, int x_ld, const void* w, const float* b, float* out, int out_ld, int M,
                        int N, int K, int act_in, int act_out, mdx_stream_t s);

/* ---- fused sampler update (ldm/models/diffusion/plms.py:188-197, 210-244)
 * eps_u / eps_c: UNet outputs, NHWC fp16 [B][HW][eps_ld] (first C channels used).
 *   e_t = eps_c                                  if eps_u == NULL
 *       = eps_u + cfg_scale * (eps_c - eps_u)    otherwise            (plms.py:197)
 *   e'  = coef[0]*e_t + coef[1]*old1 + coef[2]*old2 + coef[3]*old3     (plms.py:231-244; DDIM: coef = {1,0,0,0};
 *         the 'pseudo improved Euler' second call passes coef = {.5,.5,0,0} with old1 = first e_t)
 *   pred_x0 = (x - sqrt_one_minus_at * e') / sqrt_at                   (plms.py:218)
 *   x_prev  = sqrt_a_prev * pred_x0 + dir_coef * e' + sigma * noise    (plms.py:222-226)
 * x, old*, noise, e_t_out, x_prev, pred_x0: NCHW fp32 [B][C][H][W] (noise may be NULL when sigma == 0;
 * old* may be NULL when their coef is 0; e_t_out / pred_x0 may be NULL).  x_prev may alias x.
 * coef4 is a HOST pointer to 4 floats (read at call time). */
int mdx_sampler_step_f32(const float* x, const void* eps_u, const void* eps_c, int eps_ld, float cfg_scale,
                         const float* old1, const float* old2, const float* old3, const float* coef4,
                         float sqrt_at, float sqrt_one_minus_at, float sqrt_a_prev, float dir_coef, float sigma,
                         const float* noise, float* e_t_out, float* x_prev, float* pred_x0, int B, int C, int H,
                         int W, mdx_stream_t s);

/* ---- GLIDE (Taichu-GLIDE/model/glide_text2im) specifics --------------------------------------------------
 * AvgPool2d(2,2) / ResizeNearestNeighbor x2 of a ResBlock's skip path (unet.py:46-49,74,180-185); NHWC fp16. */
int mdx_avgpool2x2_f16(const void* x, void* y, int B, int H, int W, int C, mdx_stream_t s);
int mdx_upsample_nearest2x_f16(const void* x, void* y, int B, int H, int W, int C, mdx_stream_t s);
/* token + positional embedding with padding replacement (text2im_model.py:88-92):
 * out[b][t][:] = mask[b][t] ? tok_emb[tokens[b][t]] + pos[t] : pad[t];  fp16 tables, int32 tokens/mask. */
int mdx_glide_text_embed_f16(const int* tokens, const int* mask, const void* tok_emb, const void* pos,
                             const void* pad, void* out, int B, int T, int width, int n_vocab, mdx_stream_t s);
/* super-res UNet input (text2im_model.py:214-216 + gaussian_diffusion.py:307-313):
 * out NHWC fp16 [B][S*S][8] = [x (3ch) | legacy-bilinear(round((low+1)*127.5)/127.5-1, s->S) (3ch) | 0 0];
 * x [B][3][S][S] fp32, low [B][3][s][s] fp32. */
int mdx_glide_superres_input_f16(const float* x, const float* low, void* out, int B, int S, int s_low,
                                 mdx_stream_t s);
/* one fused sampler update (guider.py:73-86, gaussian_diffusion.py:79-142, 229-254):
 *   eps = out_u ? out_u[:3] + scale*(out_c[:3] - out_u[:3]) : out_c[:3];  v = out_c[3:6]
 *   logvar = (v+1)/2*log_beta + (1-(v+1)/2)*post_logvar;  x0 = clip(sqrt_recip*x - sqrt_recipm1*eps, -1, 1)
 *   mode 0 (ancestral): x_next = coef1*x0 + coef2*x + noise_scale*exp(logvar/2)*noise   (noise_scale = 0 at t = 0)
 *   mode 1 (DDIM eta=0): eps' = (sqrt_recip*x - x0)/sqrt_recipm1; x_next = sqrt_ab_prev*x0 + sqrt(1-ab_prev)*eps'
 * out_c/out_u: UNet outputs NHWC fp16 [B][HW][ld] (6 channels used); x, noise, x_next, pred_x0: NCHW fp32 [B][3][H][W].
 * coef: HOST pointer to 8 floats {log_beta, post_logvar, sqrt_recip, sqrt_recipm1, coef1, coef2, sqrt_ab_prev,
 * sqrt_one_minus_ab_prev}. */
int mdx_glide_step_f32(const float* x, const void* out_c, const void* out_u, int ld, float guidance_scale,
                       const float* coef8, int mode, float noise_scale, const float* noise, float* x_next,
                       float* pred_x0, int B, int H, int W, mdx_stream_t s);

/* Taichu-GLIDE sampling loops know every prompt before the first step (main_funcs.py:21-44 draws the unconditional prompt of
 * step k inside the loop, but from a stream that does not depend on the model): the text transformer (xf.py:126-154) and every
 * AttentionBlock's encoder_kv projection (unet.py:289-297) of ALL steps' prompts run once per loop into tables, and each step
 * copies its prompt's rows into the text slots of the blocks' key / value buffers -- ONE launch for all blocks:
 *   for slot in [0, nslots), b in [b0, b0 + nb):  entry e = entry0 + (b - b0) * entry_per_b (entry_per_b in {0, 1})
 *     dst[b * dst_batch_bytes + r * dst_pitch + 0..row_bytes) = src[e * src_entry_bytes + r * src_pitch + 0..row_bytes),  r < rows
 * (keys: rows = text tokens, row_bytes = 2 C, both pitches 2 C; transposed values: rows = C, row_bytes = 2 text tokens,
 * dst_pitch = 2 (text + image tokens)).  `slots_dev` is a DEVICE array (caller-owned); every pointer and byte count a multiple
 * of 16.  blocks_per_copy = grid blocks that share one (slot, b) copy. */
typedef struct mdx_glide_kv_slot {
    const void* src;
    void* dst;
    long src_entry_bytes;
    long dst_batch_bytes;
    long src_pitch, dst_pitch;
    int rows, row_bytes;
} mdx_glide_kv_slot;
int mdx_glide_kv_select_f16(const mdx_glide_kv_slot* slots_dev, int nslots, long entry0, int entry_per_b, int b0, int nb,
                            int blocks_per_copy, mdx_stream_t s);

/* Causal self-attention (key j visible to query i iff j <= i): the text encoder's triu(-inf) mask
 * (ldm/modules/encoders/text_encoder.py:136-139, MultiheadAttention :43-66).  Same arguments; Nq must equal Nk. */
int mdx_attention_causal_f16(const void* q, long q_bs, int q_ld, const void* k, long k_bs, int k_ld, const void* vt,
                             long vt_bs, int vt_ld, void* o, long o_bs, int o_ld, int B, int heads, int D, int Nq,
                             int Nk, float scale, mdx_stream_t s);

/* ---- VAE decoder attention (AutoencoderKL.decode -> Decoder.mid.attn_1, ldm/modules/diffusionmodules/model.py:151-206):
 * one head of d = C = 512, scores materialised as in the reference: S = Q K^T and O = P V run through mdx_gemm_f16. */
/* Re-lay a row-major fp16 ACTIVATION matrix src[rows][K] (row stride src_ld elements) as the packed B operand of
 * mdx_gemm_f16 (the format ops.pack_gemm_weight builds for weights): dst holds ceil(rows/64)*ceil(K/64)*4096 halves. */
int mdx_pack_b_operand_f16(const void* src, long src_ld, int rows, int K, void* dst, mdx_stream_t s);
/* In place: x[r][0..cols) = softmax(scale * x[r][0..cols)) for fp16 scores (P.Softmax(axis=2), model.py:192-194);
 * fp32 max / sum.  cols % 8 == 0, cols <= 16384. */
int mdx_softmax_rows_f16(void* x, long ld, int rows, int cols, float scale, mdx_stream_t s);

/* AutoencoderKL.encode's DiagonalGaussianDistribution sample (autoencoder.py:70-78): moments NHWC fp16 [B][HW][ld] =
 * [mean (zc) | logvar (zc) | ..]; out NCHW fp32 [B][zc][HW] = mean + exp(0.5 * clip(logvar, -30, 20)) * noise
 * (noise NCHW fp32, or NULL for the mode). */
int mdx_vae_gaussian_sample_f32(const void* moments, int ld, const float* noise, float* out, int B, int zc, int HW,
                                mdx_stream_t s);

/* ---- probes used by tests to pin hardware layout assumptions (not on the hot path) */
int mdx_probe_mfma_32x32x16_f16(const void* a, const void* b, float* c, mdx_stream_t s);
/* Same for v_mfma_f32_16x16x32_f16 (the conv8p core): a, b = 64 lanes x 8 halves, c = 64 lanes x 4 floats. */
int mdx_probe_mfma_16x16x32_f16(const void* a, const void* b, float* c, mdx_stream_t s);
/* streaming-bandwidth probe of the HBM/L2 -> LDS DMA path (mode 0) vs plain vector loads (mode 1) */
int mdx_probe_dma_stream(const void* src, size_t bytes_per_block, int nblocks, int waves, int per, int ns, int mode,
                         int stride_tiles, float* sink, mdx_stream_t s);

/* streaming probe of the L2 -> VGPR path the fused-chain kernels use for their weights: each wave reads its own contiguous
 * bytes_per_wave region in 1 KiB pieces, pf in flight; shared = 1: every block reads the same regions (tools/l2_probe.py) */
int mdx_probe_l2_stream(const void* src, size_t total_bytes, unsigned bytes_per_wave, int nblocks, int waves, int pf,
                        int shared, float* sink, mdx_stream_t s);

/* diagnostics: register a device buffer (bytes >= 64 x blocks) that the blocks of subsequent mdx_gemm_f16 launches
 * fill with phase timestamps (8 x u64 per block, 100 MHz realtime counter); NULL unregisters.  tools/gemm_trace.py */
int mdx_probe_gemm_trace(void* buf, size_t bytes);
/* VALU issue-rate probe (no reference counterpart; diagnostics): `iters` rounds of eight independent chains per lane of one
 * instruction kind -- 0 v_fma_f32, 1 v_exp_f32, 2 v_pk_fma_f32, 3 v_cvt_pk_f16_f32 (+ two converts back), 4 v_max3_f32, 5 v_exp_f16,
 * 6 v_pk_fma_f16, 7 v_pk_max_f16 -- on
 * `nblocks` blocks of 256 threads; time it from the host (tools/exp/r04n_valu_probe.py). */
int mdx_probe_valu_rate(int kind, int iters, int nblocks, float* sink, mdx_stream_t s);
/* Diagnostics (round 6): four slots per iteration of { one 32x32x16 MFMA } and / or VALU work -- kind 0 MFMA alone, 1 MFMA + 2 v_exp_f32,
 * 2 MFMA + 8 v_fma_f32, 3 the 2 exps alone, 4 the 8 fmas alone, 5 MFMA + the softmax mix of one score pair, 6 that mix alone: does VALU
 * work run in the shadow of an MFMA?  (tools/exp/r06_mix_probe.py) */
int mdx_probe_mix_rate(int kind, int iters, int nblocks, float* sink, mdx_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* MDX_H_ */
