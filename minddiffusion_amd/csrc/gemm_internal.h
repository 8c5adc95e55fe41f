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
    // sub-pixel form of the nearest-2x + 3x3 conv (mdx_gemm_desc.w_sub): the per-parity 2 x 2 weights, logical [4 N][4 Cin]
    const f16* w_sub;
    unsigned w_sub_bytes;
    int c8_sub;      // resolved: this launch runs the sub-pixel form on the conv8p core
    int dense_issue; // dense launches issue their K tiles through the scalar offset (option gemm_dense_issue)
    int ln_prefetch; // LayerNorm-fold consumer: the tile's statistics partials and S[n] are touched before the K loop (option gemm_ln_prefetch)
    // lean dense kernel (dense.hip): tile_id / tiles_n and tile_id / tiles_m as a multiply-high (ceil(2^32 / d): exact for the < 2^16
    // tile ids a launch has)
    unsigned inv_tiles_n, inv_tiles_m;
    unsigned res_bytes;      // bytes of the residual tensor ([M][residual_ld] fp16), for its buffer descriptor
    // cross-attention over a cached context as the epilogue of the query projection (mdx_gemm_desc.xattn_k; lean dense kernel, BN = 64 = head dim)
    const f16* xa_k;         // [B][xa_cap][N]
    const f16* xa_vt;        // [B][N][xa_cap]
    int xa_len, xa_cap;
    float xa_scale_log2;     // scale * log2(e)
};

}  // namespace mdx_int
using mdx_int::GemmParams;

// 8-phase 256-pixel conv core (conv8p.hip): eligibility + launch, called from mdx_gemm_f16 / mdx_gemm_query (gemm.hip)
// lean dense kernel (dense.hip): dense row-major launches of the benchmarked tile shapes; false = this (tile, ring) has no lean
// instantiation and the caller launches the generic kernel
bool mdx_dense_launch(const GemmParams& p, int bm, int bn, int ns, dim3 grid, hipStream_t st);
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
    __device__ __forceinline__ int sample() const { return 0; }      // (never asked: EMODE 2 is for patch tiles)
};
// HALO conv tiles are 8 x 16 pixel patches: row = py*16 + px
struct PatchRows {
    int base, W;   // base = (b*H + y0)*W + x0
    int b;         // the sample every row of the patch belongs to (batched epilogue, EMODE 2)
    __device__ __forceinline__ int operator()(int row) const { return base + (row >> 4) * W + (row & 15); }
    __device__ __forceinline__ int sample() const { return b; }
};

// The row-major epilogue's bias columns of this thread, fetched BEFORE the K loop: biases are cold in HBM (1.7 GB of
// weights stream through the caches between two uses), and a 1-3 us miss at the start of the epilogue was the largest
// single item of a small GEMM's fixed cost (tools/gemm_trace.py).  [0..7] plain / GEGLU 'a' columns, [8..15] gate.
template <int BN, bool SWAP, int NW>
__device__ __forceinline__ void gemm_bias_prefetch(const GemmParams& p, const int n0, float (&bpre)[16]) {
#pragma unroll
    for (int e = 0; e < 16; ++e) bpre[e] = 0.f;
    if constexpr (SWAP) {
        if (!p.bias) return;
        const int tid = threadIdx.x;
        const bool geglu = p.epilogue == MDX_EPI_GEGLU;
        const int n = n0 + (geglu ? (tid & 7) : (tid % (BN / 8))) * 8;
        if (n >= p.N) return;
        const float4* b4 = reinterpret_cast<const float4*>(p.bias + n);
        const float4 x0 = b4[0], x1 = b4[1];
        bpre[0] = x0.x; bpre[1] = x0.y; bpre[2] = x0.z; bpre[3] = x0.w;
        bpre[4] = x1.x; bpre[5] = x1.y; bpre[6] = x1.z; bpre[7] = x1.w;
        if (geglu) {
            const float4 g0 = b4[16], g1 = b4[17];
            bpre[8] = g0.x; bpre[9] = g0.y; bpre[10] = g0.z; bpre[11] = g0.w;
            bpre[12] = g1.x; bpre[13] = g1.y; bpre[14] = g1.z; bpre[15] = g1.w;
        }
    }
}

// LayerNorm fold, consumer side: {mean, rstd} of token row m from the producer's per-64-column {sum, sumsq} partials (p.ln_stats),
// added in partial order with eight loads in flight (a load -> add loop made this a chain of K / 64 L2 round trips -- 10-20 -- at the
// head of every epilogue; with one block per CU nothing hides it).
__device__ __forceinline__ void gemm_ln_row_fold(const GemmParams& p, const int m, float (&mr)[2]) {
    float su = 0.f, sq = 0.f;
    if (m < p.M) {
        const float2* st = reinterpret_cast<const float2*>(p.ln_stats) + (size_t)m * p.ln_nt;
        int j = 0;
        for (; j + 8 <= p.ln_nt; j += 8) {
            float2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = st[j + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                su += v[u].x;
                sq += v[u].y;
            }
        }
        for (; j + 2 <= p.ln_nt; j += 2) {
            const float2 v0 = st[j], v1 = st[j + 1];
            su += v0.x; sq += v0.y; su += v1.x; sq += v1.y;
        }
        for (; j < p.ln_nt; ++j) {
            const float2 v = st[j];
            su += v.x;
            sq += v.y;
        }
    }
    const float inv = 1.0f / (float)p.K;
    const float mean = su * inv;
    float var = sq * inv - mean * mean;
    var = var < 0.f ? 0.f : var;
    mr[0] = mean;
    mr[1] = rsqrtf(var + p.ln_eps);
}

// In-kernel split-K reduce ("last block in finishes the tile").  Every (tile, split) block parks its fp32 accumulators in
// the workspace IN REGISTER LAYOUT -- [tile][split][register quad][thread] 16-byte pieces, so that the stores and the later
// loads are 1 KiB-per-wave contiguous -- then takes a ticket on the tile's arrival counter.  The block that draws the last
// ticket sums all nsplit partials in split order (its own included: the order, and so the fp32 result, does not depend on who
// arrives last) back into its accumulator registers and falls through to the ordinary epilogue: every epilogue feature
// (GEGLU, LayerNorm fold, row / column statistics, q|k|v split stores) works unchanged for split launches, and there is no
// reduce launch.  Hand-off (cdna_hip_programming.md, "In-launch split-K reduction" / Guideline 16, the sc1 form -- a
// __threadfence() per block measured +47 us on a 400-block launch here): partials are stored WRITE-THROUGH at agent scope
// (buffer_store ... sc1), every wave drains its stores (s_waitcnt vmcnt(0)), block barrier, ONE lane takes the ticket with a
// relaxed agent-scope atomic; the last arriver reads the partials with agent-scope (sc1) loads, which cannot hit a stale
// line of its CU's L1 or its XCD's L2.  No placement assumption: a tile's splits may run on any CUs of any XCDs.
// The counter is reset by the last arriver, so a tile's counter is zero whenever no launch is in flight on its workspace.
constexpr int MDX_TICKET_SLOTS = MDX_GEMM_WS_HEAD / 4;      // tiles per launch that can take tickets
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int BM, int BN, int NW>
__device__ __forceinline__ bool splitk_last_block_reduce(const GemmParams& p, f32x16 (&acc)[BM / (16 * NW)][BN / 64], char* smem,
                                                         const int tile_lin, const int split) {
    constexpr int NT = NW * 64;
    constexpr int TM = BM / (16 * NW);
    constexpr int TN = BN / 64;
    constexpr int Q = TM * TN * 4;                       // 16-byte pieces per thread per partial
    constexpr int U = (32 / Q) < 1 ? 1 : (32 / Q);       // partials whose loads are in flight together (<= 128 VGPRs)
    constexpr unsigned PART = (unsigned)Q * NT * 16u;    // bytes per partial
    const int tid = threadIdx.x;
    char* base = reinterpret_cast<char*>(p.ws + MDX_TICKET_SLOTS) + (size_t)tile_lin * p.nsplit * PART;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(base, (unsigned)p.nsplit * PART);
    const unsigned toff = (unsigned)tid * 16u;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
                v[0] = acc[i][j][4 * g]; v[1] = acc[i][j][4 * g + 1]; v[2] = acc[i][j][4 * g + 2]; v[3] = acc[i][j][4 * g + 3];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs,
                                                       (unsigned)split * PART + (unsigned)((i * TN + j) * 4 + g) * (NT * 16u) + toff,
                                                       0, /*sc1*/ 16);
            }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave: its write-through stores are complete
    __syncthreads();
    int* flag = reinterpret_cast<int*>(smem);
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(p.tickets + tile_lin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old >= (unsigned)p.nsplit) __builtin_trap();     // counters corrupted (two streams sharing one workspace, mdx.h)
        *flag = old == (unsigned)p.nsplit - 1u;
    }
    __syncthreads();
    const bool last = *flag != 0;
    __syncthreads();                                     // the epilogue reuses smem
    if (!last) return false;
    if (tid == 0) __hip_atomic_store(p.tickets + tile_lin, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto add = [&](const u32x4 (&v)[Q]) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const f32x4 f = __builtin_bit_cast(f32x4, v[q]);
            f32x16& a = acc[q / (TN * 4)][(q / 4) % TN];
            const int g = q & 3;
            a[4 * g] += f[0]; a[4 * g + 1] += f[1]; a[4 * g + 2] += f[2]; a[4 * g + 3] += f[3];
        }
    };
    // U partials' loads in flight per step; added strictly in split order
    int z = 0;
    for (; z + U <= p.nsplit; z += U) {
        u32x4 v[U][Q];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int q = 0; q < Q; ++q)
                v[u][q] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(z + u) * PART + (unsigned)q * (NT * 16u) + toff, 0, 16);
#pragma unroll
        for (int u = 0; u < U; ++u) add(v[u]);
    }
    for (; z < p.nsplit; ++z) {
        u32x4 v[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q)
            v[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)z * PART + (unsigned)q * (NT * 16u) + toff, 0, 16);
        add(v);
    }
    return true;
}

// Fused epilogue shared by the GEMM kernels.  SWAP: accumulators hold C^T (col = lane&31 -> m), staged through LDS
// and stored row-major with bias / rowbias / residual / GEGLU / GELU; !SWAP: split-K partial slab or transposed store.
// Optional prefetches of the lean dense kernel (dense.hip), all issued BEFORE the K loop so that the epilogue of a lone block opens
// with no global round trip: ln_pre = {mean, rstd} of this thread's row (tid < BM), lns_pre = S[n0 + tid] (tid < BN), xpre = the
// residual (/ time-embedding) rows of this thread's first NXPRE store passes.
// LEAN (the lean dense kernel, dense.hip; round 6): the launch has no per-sample row bias and no out_bs (lean_dense_eligible), and
// the store loops are written in BATCHES -- every staged row / residual row of the thread's passes requested first, then the
// arithmetic, then the stores back to back, statistics last -- instead of one pass at a time behind the generic per-pass feature
// branches (row bias and out_bs cost an integer division per pass; ~450 instructions per pass in the ISA of round 5, one pass
// ~0.4 us for a wave that has its SIMD to itself).  Same operations on the same values in the same order: bit-identical output.
// EMODE: 0 = the generic per-pass loops; 1 = LEAN as above; 3 = LEAN, but STOP behind the staging barrier (the fp16 tile stays in LDS for
// the caller: cross-attention epilogue of dense.hip); 2 = the same batched loops for HALO patch tiles (conv3x3_halo_kernel,
// PW = 16: out_bs is excluded by halo_eligible, and every row of a patch lies in ONE sample, so the per-sample row bias -- the
// time-embedding term of a ResBlock's first conv, openaimodel.py:188-190 -- is one row of 8 floats per thread for the whole tile
// instead of a division and two loads per pass).
template <int BM, int BN, bool SWAP, int NW, class RowMap, int NXPRE = 0, int EMODE = 0>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x16 (&acc)[BM / (16 * NW)][BN / 64], char* smem,
                                              const RowMap rm, const int n0, const int split, const float (&bpre)[16],
                                              const int row_block = 0, const int tile_lin = 0, const float* ln_pre = nullptr,
                                              const float* lns_pre = nullptr, const Row8Extras* xpre = nullptr,
                                              const bool ln_pre_valid = true, const bool lns_pre_valid = true) {
    constexpr bool LEAN = EMODE != 0;
    constexpr bool ROWB = EMODE == 2;
    constexpr int NT = NW * 64;           // threads per block
    constexpr int WROWS = BM / (NW / 2);  // rows of the block tile owned by one wave row (waves are (NW/2) x 2)
    constexpr int TM = WROWS / 32;
    constexpr int TN = BN / 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int hi = lane >> 5;
    const int l31 = lane & 31;
    if constexpr (!SWAP) {
        if (p.nsplit > 1) {
            // split-K partial: C layout (col = lane&31 -> n, rows -> m), 128-B row segments per store
            float* wsz = p.ws + (size_t)split * p.M * p.N;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n0 + wn * (BN / 2) + j * 32 + l31;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = rm(wm * WROWS + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi);
                        if (m < p.M && n < p.N) wsz[(size_t)m * p.N + n] = acc[i][j][r];
                    }
                }
            return;
        }
        // transposed store (V^T): stage [n][m]
        constexpr int TLD = BM + 8;
        f16* stg = reinterpret_cast<f16*>(smem);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n_l = wn * (BN / 2) + j * 32 + l31;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int m_l = wm * WROWS + i * 32 + 8 * g + 4 * hi;
                    *reinterpret_cast<f16x4*>(&stg[n_l * TLD + m_l]) =
                        cvt4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                }
            }
        __syncthreads();
        constexpr int CPT = BM / 8;        // 16-B chunks per staged n-row
        constexpr int RPT = NT / CPT;      // n-rows per pass
        const int chunk = tid % CPT, r0 = tid / CPT;
        const int m = rm(chunk * 8);
#pragma unroll
        for (int pass = 0; pass < BN / RPT; ++pass) {
            const int nrow = r0 + pass * RPT;
            const int n = n0 + nrow;
            if (n < p.N && m < p.M) {
                f16x8 v = *reinterpret_cast<const f16x8*>(&stg[nrow * TLD + chunk * 8]);
                if (p.bias) {
                    const float bb = p.bias[n];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (f16)((float)v[e] + bb);
                }
                if ((p.HoWo & 7) == 0) {
                    const int b = m / p.HoWo;
                    const int tok = m - b * p.HoWo;
                    *reinterpret_cast<f16x8*>(p.out + ((size_t)b * p.N + n) * p.out_ld + tok) = v;
                } else {
                    // tokens per sample not a multiple of 8 (5 x 5, 6 x 6, 7 x 7 ... images at the deepest level of a 320 / 384 / 448-
                    // pixel run): the 8 rows of a chunk may straddle two samples or run past M -- element stores (tiny tensors)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int mm = m + e;
                        if (mm < p.M) {
                            const int b = mm / p.HoWo;
                            p.out[((size_t)b * p.N + n) * p.out_ld + (mm - b * p.HoWo)] = v[e];
                        }
                    }
                }
            }
        }
    } else {
        // row-major store: C^T layout (col = lane&31 -> m, 4 consecutive n per register group)
        constexpr int SLD = BN + 8;
        f16* stg = reinterpret_cast<f16*>(smem);
        if (p.tickets) {      // split-K: only the block that completes the tile goes on (block-uniform)
            if (!splitk_last_block_reduce<BM, BN, NW>(p, acc, smem, tile_lin, split)) return;
        }
        if (p.ln_stats) {
            // LayerNorm fold: acc holds raw_tokens x (gamma (.) W)^T.  Per row mean / rstd from the producer's partials and
            // S[n] of this tile go through LDS (behind the staging area), then every accumulator becomes
            // rstd_m * (acc - mean_m * S_n) in fp32 -- BEFORE the fp16 staging, so the cancellation costs no precision.
            float* lnrow = reinterpret_cast<float*>(smem + (size_t)BM * SLD * 2);      // [BM][2] mean, rstd
            float* lns = lnrow + 2 * BM;                                               // [BN]
            if (ln_pre && ln_pre_valid) {      // this thread's row (tid < BM) was folded from partials fetched before the K loop (dense.hip)
                if (tid < BM) {
                    lnrow[2 * tid] = ln_pre[0];
                    lnrow[2 * tid + 1] = ln_pre[1];
                }
            } else {
                for (int r = tid; r < BM; r += NT) {
                    float mr[2];
                    gemm_ln_row_fold(p, rm(r), mr);
                    lnrow[2 * r] = mr[0];
                    lnrow[2 * r + 1] = mr[1];
                }
            }
            if (lns_pre && lns_pre_valid) {
                if (tid < BN) lns[tid] = *lns_pre;
            } else {
                for (int c = tid; c < BN; c += NT) lns[c] = (n0 + c < p.N) ? p.ln_s[n0 + c] : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m_l = wm * WROWS + i * 32 + l31;
                const float mean = lnrow[2 * m_l], rstd = lnrow[2 * m_l + 1];
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int n_l = wn * (BN / 2) + j * 32 + 8 * (r >> 2) + 4 * hi + (r & 3);
                        acc[i][j][r] = rstd * (acc[i][j][r] - mean * lns[n_l]);
                    }
            }
        }
        // plain (non-GEGLU) store: thread -> (row r0 + pass * RPP, 8 columns at n).  Its global loads (bias, and the
        // first pass's time-embedding row / residual) are issued BEFORE the staging barrier so that their latency
        // overlaps the accumulator -> LDS pass; later passes prefetch one pass ahead.
        constexpr int CPR = BN / 8;
        constexpr int RPP = NT / CPR;
        const int chunk = tid % CPR, r0 = tid / CPR;
        const int n = n0 + chunk * 8;
        const bool plain = p.epilogue != MDX_EPI_GEGLU;
        float bb[8];
        Row8Extras xa;
        [[maybe_unused]] f16x8 lres[(LEAN && EMODE != 3) ? BM / RPP : 1];      // LEAN: the residual rows of ALL the thread's passes
        [[maybe_unused]] u32x4 lrb[2] = {};                     // EMODE 2: the patch's row-bias row, this thread's 8 columns
        if (plain) {
#pragma unroll
            for (int e = 0; e < 8; ++e) bb[e] = bpre[e];   // fetched before the K loop (gemm_bias_prefetch)
            const int m = rm(r0);
            if constexpr (LEAN && EMODE != 3) {
                if constexpr (ROWB) {      // the tile's ONE row-bias row (zeros from the descriptor when there is none)
                    const __amdgpu_buffer_rsrc_t rs_rb = make_rsrc(p.rowbias, p.rowbias ? (unsigned)p.B * (unsigned)p.rowbias_ld * 4u : 0u);
                    const unsigned off = n < p.N ? ((unsigned)rm.sample() * (unsigned)p.rowbias_ld + (unsigned)n) * 4u : MDX_OOB;
                    lrb[0] = __builtin_amdgcn_raw_buffer_load_b128(rs_rb, off, 0, 0);
                    lrb[1] = __builtin_amdgcn_raw_buffer_load_b128(rs_rb, off, 16, 0);
                }
                // the rows the kernel did not fetch before its K loop: all requested now, in front of the staging pass, through a
                // descriptor that answers zero when there is no residual (unconditional loads: nothing waits behind a branch)
                const __amdgpu_buffer_rsrc_t rs_res = make_rsrc(p.residual, p.residual ? p.res_bytes : 0u);
#pragma unroll
                for (int pass = 0; pass < BM / RPP; ++pass) {
                    if (pass < NXPRE) {
                        lres[pass] = xpre[pass < NXPRE ? pass : 0].res;
                    } else {
                        const int mp = rm(r0 + pass * RPP);
                        const unsigned off = (mp < p.M && n < p.N) ? ((unsigned)mp * (unsigned)p.residual_ld + (unsigned)n) * 2u : MDX_OOB;
                        lres[pass] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_res, off, 0, 0));
                    }
                }
            } else if constexpr (NXPRE > 0) {
                xa = xpre[0];
            } else {
                if (m < p.M && n < p.N) xa = epilogue_prefetch_row8(p, m, n);
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int m_l = wm * WROWS + i * 32 + l31;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n_l = wn * (BN / 2) + j * 32 + 8 * g + 4 * hi;
                    *reinterpret_cast<f16x4*>(&stg[m_l * SLD + n_l]) =
                        cvt4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                }
            }
        __syncthreads();
        trace_mark(p, 5);
        if constexpr (EMODE == 3) {
            // stage only: the caller (dense.hip, cross-attention epilogue) consumes the fp16 tile in LDS -- as the values the store loop
            // would have written: fp16(float(staged) + bias) (the LayerNorm fold's W beta term arrives as a bias)
            constexpr int CPR3 = BN / 8, RPP3 = NT / CPR3;
            const int chunk3 = tid % CPR3, r03 = tid / CPR3;
            if (p.bias) {
#pragma unroll
                for (int pass = 0; pass < BM / RPP3; ++pass) {
                    f16x8* q = reinterpret_cast<f16x8*>(&stg[(r03 + pass * RPP3) * SLD + chunk3 * 8]);
                    f16x8 v = *q;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (f16)((float)v[e] + bpre[e]);
                    *q = v;
                }
            }
            return;
        }
        if (p.n_split && n0 >= p.n_split) {
            // a V tile of the merged q|k|v projection: out2[(b * Nv + n - n_split) * out2_ld + tok], 8 consecutive tokens
            // per 16-B store, gathered column-wise from the staged [m][n] tile (2-byte LDS reads; small next to a launch)
            constexpr int CPT = BM / 8;
            constexpr int RPT = NT / CPT;
            const int mchunk = tid % CPT, nr0 = tid / CPT;
            const int m = rm(mchunk * 8);
            const int nv = p.N - p.n_split;
#pragma unroll
            for (int pass = 0; pass < BN / RPT; ++pass) {
                const int nrow = nr0 + pass * RPT;
                const int nn = n0 + nrow;
                if (nn < p.N && m < p.M) {
                    const float bbv = p.bias ? p.bias[nn] : 0.f;
                    f16x8 v;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (f16)((float)stg[(mchunk * 8 + e) * SLD + nrow] + bbv);
                    if ((p.HoWo & 7) == 0) {
                        const int b = m / p.HoWo;
                        const int tok = m - b * p.HoWo;
                        *reinterpret_cast<f16x8*>(p.out2 + ((size_t)b * nv + (nn - p.n_split)) * p.out2_ld + tok) = v;
                    } else {      // ragged token count: element stores (see the transposed store above)
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int mm = m + e;
                            if (mm < p.M) {
                                const int b = mm / p.HoWo;
                                p.out2[((size_t)b * nv + (nn - p.n_split)) * p.out2_ld + (mm - b * p.HoWo)] = v[e];
                            }
                        }
                    }
                }
            }
        } else if (p.epilogue == MDX_EPI_GEGLU) {
            if constexpr (BN == 128) {
                // tile = 64 'a' columns | 64 'gate' columns -> 64 outputs at column n0/2
                const int chunk = tid & 7, r0 = tid >> 3;
                const int pn = n0 + chunk * 8;       // packed column of the 'a' part
                const int on = (n0 >> 1) + chunk * 8;  // output column
                float ba[8], bg[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    ba[e] = bpre[e];
                    bg[e] = bpre[8 + e];
                }
                if constexpr (LEAN) {
                    constexpr int GP = BM / (NT / 8);      // passes: 32 rows each
                    const bool okn = pn < p.N;
                    const bool has_res = p.residual != nullptr;      // (uniform; GEGLU with a residual is not a shape of the models)
                    f16x8 va[GP], vg[GP];
#pragma unroll
                    for (int pass = 0; pass < GP; ++pass) {
                        const int row = r0 + pass * (NT / 8);
                        va[pass] = *reinterpret_cast<const f16x8*>(&stg[row * SLD + chunk * 8]);
                        vg[pass] = *reinterpret_cast<const f16x8*>(&stg[row * SLD + 64 + chunk * 8]);
                    }
#pragma unroll
                    for (int pass = 0; pass < GP; ++pass) {
                        const int m = rm(r0 + pass * (NT / 8));
                        if (m < p.M && okn) {
                            float f[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                f[e] = ((float)va[pass][e] + ba[e]) * gelu_tanh_f((float)vg[pass][e] + bg[e]);
                            if (has_res) {
                                const f16x8 rr = *reinterpret_cast<const f16x8*>(p.residual + (size_t)m * p.residual_ld + on);
#pragma unroll
                                for (int e = 0; e < 8; ++e) f[e] += (float)rr[e];
                            }
                            f16x8 o;
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] = (f16)f[e];
                            *reinterpret_cast<f16x8*>(p.out + (size_t)m * p.out_ld + on) = o;
                        }
                    }
                    return;
                }
#pragma unroll
                for (int pass = 0; pass < BM / (NT / 8); ++pass) {
                    const int row = r0 + pass * (NT / 8);
                    const int m = rm(row);
                    if (m < p.M && pn < p.N) {
                        const f16x8 va = *reinterpret_cast<const f16x8*>(&stg[row * SLD + chunk * 8]);
                        const f16x8 vg = *reinterpret_cast<const f16x8*>(&stg[row * SLD + 64 + chunk * 8]);
                        float f[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            f[e] = ((float)va[e] + ba[e]) * gelu_tanh_f((float)vg[e] + bg[e]);
                        epilogue_store_row8(p, f, m, on);
                    }
                }
            }
        } else {
            // one copy of the store loop per activation, selected by a (uniform) branch: left as a per-element `if` the
            // compiler evaluates BOTH GELUs for every output and selects -- measured 1.5 us of every GEMM's epilogue
            // GroupNorm statistics for the consumer (mdx_gemm_desc.colstats_out): every thread owns 8 columns and BM / RPP
            // rows; its per-column {sum, sumsq} of the fp16 values it stores are folded over the RPP row lanes through LDS
            const bool colstats = p.colstats_out != nullptr;
            float cs[8], cq[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) cs[e] = cq[e] = 0.f;
            if constexpr (LEAN) {
                // (epilogue is MDX_EPI_NONE here: the lean kernel takes no GELU / QuickGELU launch)
                constexpr int PASSES = BM / RPP;
                const bool okn = n < p.N;
                const bool has_res = p.residual != nullptr;      // uniform
                [[maybe_unused]] const bool has_rb = p.rowbias != nullptr;
                // (ii) the staged rows
                f16x8 sv[PASSES];
#pragma unroll
                for (int pass = 0; pass < PASSES; ++pass)
                    sv[pass] = *reinterpret_cast<const f16x8*>(&stg[(r0 + pass * RPP) * SLD + chunk * 8]);
                // (iii) arithmetic + stores
                f16x8 ov[PASSES];
#pragma unroll
                for (int pass = 0; pass < PASSES; ++pass) {
                    const int m = rm(r0 + pass * RPP);
                    float f[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = (float)sv[pass][e] + bb[e];
                    if constexpr (ROWB) {
                        if (has_rb) {      // (uniform; the generic loop's order: row bias, then residual)
                            const f32x4 q0 = __builtin_bit_cast(f32x4, lrb[0]), q1 = __builtin_bit_cast(f32x4, lrb[1]);
                            f[0] += q0[0]; f[1] += q0[1]; f[2] += q0[2]; f[3] += q0[3];
                            f[4] += q1[0]; f[5] += q1[1]; f[6] += q1[2]; f[7] += q1[3];
                        }
                    }
                    if (has_res) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] += (float)lres[pass][e];
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) ov[pass][e] = (f16)f[e];
                    if (m < p.M && okn) *reinterpret_cast<f16x8*>(p.out + (size_t)m * p.out_ld + n) = ov[pass];
                }
                // (iv) statistics of the fp16 values stored, in the generic loop's order
                if (p.stats_out) {
#pragma unroll
                    for (int pass = 0; pass < PASSES; ++pass) {
                        const int m = rm(r0 + pass * RPP);
                        const bool ok = m < p.M && okn;
                        float su = 0.f, sq = 0.f;
                        if (ok) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float t = (float)ov[pass][e];
                                su += t;
                                sq += t * t;
                            }
                        }
#pragma unroll
                        for (int off = 4; off >= 1; off >>= 1) {
                            su += __shfl_xor(su, off, 64);
                            sq += __shfl_xor(sq, off, 64);
                        }
                        if ((chunk & 7) == 0 && ok)
                            reinterpret_cast<float2*>(p.stats_out)[(size_t)m * (p.N >> 6) + (n >> 6)] = make_float2(su, sq);
                    }
                }
                if (colstats) {
#pragma unroll
                    for (int pass = 0; pass < PASSES; ++pass) {
                        const int m = rm(r0 + pass * RPP);
                        if (m < p.M && okn) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float t = (float)ov[pass][e];
                                cs[e] += t;
                                cq[e] += t * t;
                            }
                        }
                    }
                }
            }
            auto store_rows = [&](auto act) {
#pragma unroll
                for (int pass = 0; pass < BM / RPP; ++pass) {
                    const int row = r0 + pass * RPP;
                    const int m = rm(row);
                    Row8Extras xn;
                    if (pass + 1 < BM / RPP) {
                        if (pass + 1 < NXPRE) {
                            xn = xpre[pass + 1 < NXPRE ? pass + 1 : 0];
                        } else {
                            const int m2 = rm(row + RPP);
                            if (m2 < p.M && n < p.N) xn = epilogue_prefetch_row8(p, m2, n);
                        }
                    }
                    float su = 0.f, sq = 0.f;
                    if (m < p.M && n < p.N) {
                        const f16x8 v = *reinterpret_cast<const f16x8*>(&stg[row * SLD + chunk * 8]);
                        float f[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = act((float)v[e] + bb[e]);
                        const f16x8 o = epilogue_apply_row8(p, f, m, n, xa);
                        if (colstats) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float t = (float)o[e];
                                cs[e] += t;
                                cq[e] += t * t;
                            }
                        }
                        if (p.stats_out) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float t = (float)o[e];    // statistics of the fp16 values actually stored
                                su += t;
                                sq += t * t;
                            }
                        }
                    }
                    if (p.stats_out) {   // 8 aligned lanes hold one 64-column slice of a row: fixed-order xor tree
#pragma unroll
                        for (int off = 4; off >= 1; off >>= 1) {
                            su += __shfl_xor(su, off, 64);
                            sq += __shfl_xor(sq, off, 64);
                        }
                        if ((chunk & 7) == 0 && m < p.M && n < p.N)
                            reinterpret_cast<float2*>(p.stats_out)[(size_t)m * (p.N >> 6) + (n >> 6)] = make_float2(su, sq);
                    }
                    xa = xn;
                }
            };
            if constexpr (!LEAN) {
                const int epi = __builtin_amdgcn_readfirstlane(p.epilogue);
                if (epi == MDX_EPI_NONE)
                    store_rows([](float x) { return x; });
                else if (epi == MDX_EPI_GELU)
                    store_rows([](float x) { return gelu_tanh_f(x); });
                else
                    store_rows([](float x) { return quick_gelu_f(x); });
            }
            if (colstats) {   // (block-uniform) fold the RPP row lanes of every column in a fixed order: deterministic
                __syncthreads();                                  // every thread is done reading the staged tile
                float* part = reinterpret_cast<float*>(smem);     // [RPP][BN][2]
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    part[((size_t)r0 * BN + chunk * 8 + e) * 2] = cs[e];
                    part[((size_t)r0 * BN + chunk * 8 + e) * 2 + 1] = cq[e];
                }
                __syncthreads();
                for (int i = tid; i < BN * 2; i += NT) {
                    const int c = i >> 1;
                    float a = 0.f;
                    for (int r = 0; r < RPP; ++r) a += part[((size_t)r * BN + c) * 2 + (i & 1)];
                    if (n0 + c < p.N) p.colstats_out[((size_t)row_block * p.N + n0 + c) * 2 + (i & 1)] = a;
                }
            }
        }
    }
}

}  // namespace
