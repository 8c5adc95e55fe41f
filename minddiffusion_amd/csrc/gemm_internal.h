// Internals shared by the GEMM / conv translation units of libmdx (gemm.hip, conv8p.hip): the launch parameter block and the
// row-wise epilogue helpers.  Everything sits in an anonymous namespace: each TU gets its own copy.
#pragma once
#include "mdx_common.h"

namespace mdx_int {

struct GemmParams {
    const f16* a;
    const f16* a2;
    const f16* w;
    const float* bias;
    const float* rowbias;
    const f16* residual;
    f16* out;
    f16* out2;       // transposed destination of the columns n >= n_split (0 = none): q|k row-major + V^T in ONE launch
    float* ws;
    unsigned* tickets;       // in-kernel split-K reduce: one arrival counter per output tile (library-owned, ticket_slot()), else null
    int c1, c2, cin;
    int rowbias_ld, residual_ld, out_ld, out2_ld, n_split;
    // LayerNorm folded into the GEMMs around it (mdx.h): the PRODUCER of the token stream writes per-row {sum, sumsq}
    // partials of the fp16 values it stores, one pair per N tile; the CONSUMER multiplies the raw tokens by gamma (.) W
    // and turns acc into rstd * (acc - mean * S[n]) in the accumulator registers before the usual epilogue.
    float* stats_out;        // producer: [M][N / 64][2]
    float* colstats_out;     // GroupNorm statistics for the consumer: [row block][N][2] (mdx.h)
    const float* ln_stats;   // consumer: [M][ln_nt][2] written by the producer
    const float* ln_s;       // consumer: S[n] = sum_k (gamma (.) W)[n][k], fp32 [N]
    int ln_nt;
    int bn_hint;             // mdx_gemm_desc.tile_n
    int st_hint;             // mdx_gemm_desc.stages
    int spread;              // gemm_kernel: 1-D grid of (tile, split) items dealt round-robin to the XCDs
    float ln_eps;
    long out_bs;   // element stride between samples of a row-major output (0 = dense [M][out_ld])
    int B, H, W, Ho, Wo, HoWo, M, N, K;
    int ksize, stride, upsample, pad;
    int epilogue, out_mode;
    int ktiles, ktiles_per_split, nsplit;
    int tiles_m, tiles_n, tiles_per_xcd, n_fastest;
    int kt64;   // 64-wide K tiles in the packed weight storage
    unsigned a_bytes, a2_bytes, w_bytes;
    int bk;
    unsigned long long* trace;   // diagnostics: per-block phase timestamps (mdx_probe_gemm_trace), else null
    // GroupNorm (+ SiLU) of the conv's INPUT applied inside the conv (mdx_gemm_desc.gn_colstats)
    const float* gn_cs;
    const float* gn_gamma;
    const float* gn_beta;
    int gn_nrb, gn_silu;
    float gn_eps;
    // ResBlock skip_connection fused into conv2 (mdx_gemm_desc.skip_w): extra 1x1 K tiles over the block's raw input
    const f16* skip_a;
    const f16* skip_a2;
    const f16* skip_w;
    int skip_c1, skip_c2, skip_kt, skip_kt_per_split;
    unsigned skip_a_bytes, skip_a2_bytes, skip_w_bytes;
    // conv8p (conv8p.hip): the first c8_full tiles run whole, each of the remaining c8_rem tiles is split c8_split ways along the
    // 64-channel chunks (c8_cps chunks per split) so that the last, partly filled round of 256-CU work still covers the chip
    int c8_full, c8_rem, c8_split, c8_cps;
};

}  // namespace mdx_int
using mdx_int::GemmParams;

// 8-phase 256-pixel conv core (conv8p.hip): eligibility + launch, called from mdx_gemm_f16 / mdx_gemm_query (gemm.hip)
bool mdx_conv8p_eligible(const GemmParams& p);
int mdx_conv8p_pick_bn(const GemmParams& p, int bn_hint);
int mdx_conv8p_tiles(const GemmParams& p);
// plans the tile / tail-split geometry into p (c8_*, tiles_m, tiles_n) given the caller's workspace; returns the workspace bytes
// the planned launch needs (0 = none).  query_only: the ideal plan (sizing); otherwise a missing / too small workspace makes the
// plan fall back to unsplit tiles.
size_t mdx_conv8p_plan(GemmParams& p, int bn, size_t workspace_bytes, bool have_workspace, bool query_only);
int mdx_conv8p_launch(const GemmParams& p, int bn, hipStream_t st);

namespace {


__device__ __forceinline__ f16x4 cvt4(float a, float b, float c, float d) {
    f16x4 v;
    v[0] = (f16)a;
    v[1] = (f16)b;
    v[2] = (f16)c;
    v[3] = (f16)d;
    return v;
}

// Fused epilogue for 8 consecutive output columns of one row (fp32 in, fp16 out), in two halves so that callers can
// issue the global loads (time-embedding row, residual) EARLY -- before the LDS staging barrier / the slab loads --
// and pay their latency once, overlapped, instead of once per dependent step.
struct Row8Extras {
    float4 r0, r1;   // per-sample row bias
    f16x8 res;       // residual
};

__device__ __forceinline__ Row8Extras epilogue_prefetch_row8(const GemmParams& p, int m, int n) {
    Row8Extras x;
    if (p.rowbias) {
        const int b = m / p.HoWo;
        const float4* rb = reinterpret_cast<const float4*>(p.rowbias + (size_t)b * p.rowbias_ld + n);
        x.r0 = rb[0];
        x.r1 = rb[1];
    }
    if (p.residual) x.res = *reinterpret_cast<const f16x8*>(p.residual + (size_t)m * p.residual_ld + n);
    return x;
}

__device__ __forceinline__ f16x8 epilogue_apply_row8(const GemmParams& p, float (&f)[8], int m, int n, const Row8Extras& x) {
    if (p.rowbias) {
        f[0] += x.r0.x; f[1] += x.r0.y; f[2] += x.r0.z; f[3] += x.r0.w;
        f[4] += x.r1.x; f[5] += x.r1.y; f[6] += x.r1.z; f[7] += x.r1.w;
    }
    if (p.residual) {
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] += (float)x.res[e];
    }
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (f16)f[e];
    size_t row = (size_t)m * p.out_ld;
    if (p.out_bs) {
        const int b = m / p.HoWo;
        row = (size_t)b * p.out_bs + (size_t)(m - b * p.HoWo) * p.out_ld;
    }
    *reinterpret_cast<f16x8*>(p.out + row + n) = o;
    return o;
}

__device__ __forceinline__ void epilogue_store_row8(const GemmParams& p, float (&f)[8], int m, int n) {
    const Row8Extras x = epilogue_prefetch_row8(p, m, n);
    epilogue_apply_row8(p, f, m, n, x);
}

// Diagnostics (mdx_probe_gemm_trace): block `bid` records the 100 MHz realtime counter at phase `slot`.
// Compiled in only with -DMDX_GEMM_TRACE (libmdx_trace.so, `make trace`); the product library carries no trace code.
__device__ __forceinline__ void trace_mark(const GemmParams& p, int slot) {
#ifdef MDX_GEMM_TRACE
    if (p.trace && threadIdx.x == 0)
        p.trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + slot] = __builtin_amdgcn_s_memrealtime();
#endif
}

// Row maps: tile-local output row -> global output row m.
struct LinearRows {
    int m0;
    __device__ __forceinline__ int operator()(int row) const { return m0 + row; }
};
// HALO conv tiles are 8 x 16 pixel patches: row = py*16 + px
struct PatchRows {
    int base, W;   // base = (b*H + y0)*W + x0
    __device__ __forceinline__ int operator()(int row) const { return base + (row >> 4) * W + (row & 15); }
};

}  // namespace
