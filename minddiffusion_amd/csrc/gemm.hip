// Implicit-GEMM conv3x3 / conv1x1 / Dense for gfx950 (MI355X).
//
//   out[m][n] = sum_k A[m][k] * W[n][k]      m = (b,yo,xo), k = (tap,cin), NHWC fp16, fp32 accumulate
//
// Replaces the reference's nn.Conv2d / nn.Dense call sites (include/mdx.h cites them).
//
// Design (MI355X-first):
//  * gemm_kernel: 256 threads = 4 wave64 (2x2), block tile {128,64} x {128,64}, BK = 64, MFMA 32x32x16 f16.
//    conv3x3_halo_kernel: stride-1 3x3 convs on 8x16-pixel patches whose 10x18 halo is DMA'd once per 64-channel
//    chunk and serves all nine taps (6x fewer activation bytes through the DMA path).
//  * Both operand tiles go HBM -> LDS with `buffer_load_dwordx4 ... lds` (16 B per lane, no VGPR
//    round trip).  The im2col gather, zero padding, stride-2, nearest-2x upsample and the
//    two-source channel concat are all folded into the per-lane SOURCE offset; out-of-range
//    offsets rely on the buffer descriptor's bounds check returning zeros.
//  * LDS image is lane-linear per DMA (8 rows x 128 B); the bank-conflict XOR swizzle
//    (chunk ^= (row>>1)&7; halo: keyed on the halo column) is applied on the source side and again on the
//    ds_read_b128 side.  Weights are stored tile-major and pre-swizzled so each DMA reads 1 KiB contiguous.
//  * 2-6 stage LDS ring, counted `s_waitcnt vmcnt(N)` + raw `s_barrier` (DMAs stay in flight across the
//    barrier); per K tile: ring-refill DMA issue, then the tile's fragment reads pinned with sched_barriers, then the MFMAs
//    (round 5); dense launches issue through the DMA's scalar offset; XCD-aware tile order; kernel arguments warmed at entry.
//  * Epilogue is staged through LDS so that global stores are full 16-B / 128-B-row coalesced;
//    bias (fetched before the K loop), per-sample time-embedding bias, residual add, GEGLU / GELU / QuickGELU, the
//    transposed (V^T) store and the split q|k row-major + V^T output are fused there.
//  * choose_tiling picks the tile height and the split-K factor per shape from a measured cost model;
//    split-K = fp32 slabs + a fused reduce/epilogue kernel (fixed slab order: deterministic).
#include "mdx_common.h"
#include "gemm_internal.h"

// timing ablations of the HALO tap loop (tools/exp/r05_halo_ablate.sh; WRONG RESULTS, never in the product build): 1 = no MFMAs,
// 2 = no fragment reads, 3 = no DMA issue inside the loop, 4 = no barrier
#ifndef MDX_HALO_ABLATE
#define MDX_HALO_ABLATE 0
#endif
// 1 = the HALO patch tiles run the batched store loops of gemm_epilogue<..., EMODE 2> (round 6); 0 = the generic per-pass loops (A/B builds)
#ifndef MDX_HALO_LEAN_EPI
#define MDX_HALO_LEAN_EPI 1
#endif
constexpr int HALO_EMODE = MDX_HALO_LEAN_EPI ? 2 : 0;

#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

namespace {


// GNA (dense launches, ksize 1): nn.GroupNorm(32) of the INPUT -- no activation: SpatialTransformer.norm -> proj_in (attention.py:
// 243-247), GLIDE's AttentionBlock.norm -> qkv -- applied to the A fragments on their way from LDS to the matrix pipe.  GroupNorm
// without an activation is affine per (sample, channel): x' = a[b][k] x + s[b][k], a = gamma rstd, s = beta - mean a.  The block
// folds its sample's column partials (the producer's colstats_out, exactly as the HALO conv does below) into two fp16 tables in LDS
// and every A fragment becomes ONE packed fma per two halves (v_pk_fma_f16, single rounding) right after its ds_read: the
// normalised tensor is never written or read, and the GroupNorm launch disappears.  An M tile lies inside one sample
// (tokens per sample %% BM == 0, checked on the host).
template <int BM, int BN, int BK, int NS, bool SWAP, bool FASTK, int NW, bool GNA = false>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void gemm_kernel(const GemmParams p) {
    // NW = 4 (2 x 2 waves) or 8 (4 x 2 waves, BM = 128 only).  The 8-wave form is for grids of at most one block per
    // CU: a wave's K-step is a serial chain (wait -> barrier -> DMA issue -> ds_read -> MFMA), so a lone 4-wave block
    // leaves each SIMD idle for most of it; eight waves halve every wave's share of the DMA issue and MFMAs and give
    // each SIMD a second wave to overlap with.
    mdx_kernarg_touch<sizeof(GemmParams)>();
    constexpr int WROWS = BM / (NW / 2);    // rows per wave row
    constexpr int TM = WROWS / 32;          // 32-row MFMA tiles per wave along m
    constexpr int TN = BN / 64;             // 32-wide MFMA tiles per wave along n
    static_assert(TM >= 1, "tile too short for this many waves");
    constexpr int ROWB = BK * 2;            // bytes per LDS row (128 | 64)
    constexpr int CPRW = BK / 8;            // 16-B chunks per row (8 | 4)
    constexpr int RPI = 64 / CPRW;          // rows covered by one wave-wide DMA instruction (8 | 16)
    constexpr int RP256 = 256 / ROWB;       // rows per 256-B LDS bank row (2 | 4): swizzle key = (row / RP256) % CPRW
    constexpr int KS = BK / 16;             // MFMA k-steps per K tile
    constexpr int A_BYTES = BM * ROWB;
    constexpr int B_BYTES = BN * ROWB;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int AJ = BM / RPI / NW;       // A-tile DMA instructions per wave
    constexpr int BJ = BN / RPI / NW;       // B-tile DMA instructions per wave
    static_assert(AJ >= 1 && BJ >= 1, "tile too small for this many loader waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int hi = lane >> 5;
    const int l31 = lane & 31;

    // XCD-aware tile order: the dispatcher places block b on XCD b % 8 (per-XCD L2s are private), so give every
    // XCD a CONTIGUOUS run of tile ids; ids run fastest along the dimension that shares the bigger operand.
    // SPREAD order (one row of M tiles, i.e. no two blocks share a weight tile): consecutive block ids -- consecutive
    // XCDs -- take consecutive (tile, split) items, so the weight stream is pulled through all 8 XCDs' fabric links
    // even when there are only 10 N tiles (the contiguous order would park them on 5 XCDs and leave 3 idle).
    const int ntiles = p.tiles_m * p.tiles_n;
    const int tile_id = p.spread ? (int)(blockIdx.x % (unsigned)ntiles) : (blockIdx.x & 7) * p.tiles_per_xcd + (blockIdx.x >> 3);
    if (tile_id >= ntiles) return;
    trace_mark(p, 0);
    int tile_m, tile_n;
    if (p.n_fastest) {
        tile_m = tile_id / p.tiles_n;
        tile_n = tile_id - tile_m * p.tiles_n;
    } else {
        tile_n = tile_id / p.tiles_m;
        tile_m = tile_id - tile_n * p.tiles_m;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int split = p.spread ? (int)(blockIdx.x / (unsigned)ntiles) : (int)blockIdx.y;
    const int kt_begin = split * p.ktiles_per_split;
    const int kt_end = min(p.ktiles, kt_begin + p.ktiles_per_split);

    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(p.a, p.a_bytes);
    const __amdgpu_buffer_rsrc_t rs_a2 = make_rsrc(p.a2 ? p.a2 : p.a, p.a2 ? p.a2_bytes : p.a_bytes);
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, p.w_bytes);

    // ---- loader coordinates (fixed per thread across K tiles)
    // DMA instruction j of a wave covers rows (wave*AJ + j)*RPI .. +RPI-1; lane -> row + lane/CPRW, physical chunk lane%CPRW
    const int lrow = lane / CPRW, lchk = lane % CPRW;
    int a_pix[AJ];            // pixel index of tap (0,0): (b*H + y0)*W + x0 (may lie outside the image)
    int a_y[AJ], a_x[AJ];     // y0, x0 (source coordinates of tap (0,0); upsampled coordinates when p.upsample)
    int a_b[AJ];              // b*H*W
    unsigned a_cb[AJ];        // byte offset of this lane's logical chunk inside a row
    unsigned a_mask[AJ];      // bit t set <=> tap t is inside the image for this row (and the row exists)
    const int Hs = p.upsample ? 2 * p.H : p.H;  // extent the taps are clipped against
    const int Ws = p.upsample ? 2 * p.W : p.W;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int row = (wave * AJ + j) * RPI + lrow;
        const int m = m0 + row;
        const bool okm = m < p.M;
        a_cb[j] = (unsigned)((lchk ^ ((row / RP256) % CPRW)) * 16);
        if (p.ksize == 1 && p.stride == 1 && !p.upsample) {
            // Dense / 1x1 conv (most launches): the source pixel IS the output row -- skip the (b, y, x) divisions
            a_y[j] = a_x[j] = 0;
            a_pix[j] = a_b[j] = okm ? m : 0;
            a_mask[j] = okm ? 1u : 0u;
            continue;
        }
        const int mm = okm ? m : 0;
        const int b = mm / p.HoWo;
        const int rem = mm - b * p.HoWo;
        const int yo = rem / p.Wo;
        const int xo = rem - yo * p.Wo;
        a_b[j] = b * p.H * p.W;
        a_y[j] = yo * p.stride - p.pad;
        a_x[j] = xo * p.stride - p.pad;
        a_pix[j] = a_b[j] + a_y[j] * p.W + a_x[j];
        unsigned mk = 0;
        for (int t = 0; t < p.ksize * p.ksize; ++t) {
            const int ky = (p.ksize == 3) ? t / 3 : 0, kx = (p.ksize == 3) ? t - ky * 3 : 0;
            const int yi = a_y[j] + ky, xi = a_x[j] + kx;
            if (okm && yi >= 0 && yi < Hs && xi >= 0 && xi < Ws) mk |= 1u << t;
        }
        a_mask[j] = mk;
    }
    // B (weights): stored tile-major and PRE-SWIZZLED on the host -- [N/64 panels][K/64 tiles][64 rows][8 chunks][8]
    // with chunk position q holding logical chunk q ^ ((row>>1)&7) -- so every DMA instruction reads 1 KiB of
    // CONTIGUOUS memory (no L2 channel camping on the K*2-byte row stride, sequential HBM streaming along K).
    static_assert(BK == 64, "weight storage is tiled for BK = 64");
    unsigned b_off[BJ];  // byte offset of this lane's 16 B inside K tile 0
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int row = (wave * BJ + j) * RPI + lrow;   // 0 .. BN-1
        const int panel = (n0 >> 6) + (row >> 6);
        b_off[j] = (unsigned)(((size_t)panel * p.kt64) * 8192 + ((row & 63) * 8 + lchk) * 16);
    }

    // Dense launches (ksize 1, stride 1, no upsample, one source: every token GEMM and 1x1 conv -- two thirds of the launches of a
    // UNet evaluation): the source offset of a K tile is the lane's row offset + kt * 128 bytes.  The lane part is fixed for the
    // whole launch and the K part is UNIFORM, so it rides in the DMA instruction's scalar offset: issuing a K tile is AJ + BJ
    // instructions and their m0 moves, instead of the tap / source decode (~45 scalar instructions: a division by the tap count)
    // plus ~10 instructions per DMA.  A block that has a SIMD to itself issues one instruction per ~4 clocks, and its K step
    // (~190 instructions at 64 x 64) was issue-bound as much as DMA-bound (round 5).
    const bool dense = FASTK && p.ksize == 1 && p.stride == 1 && !p.upsample && p.c2 == 0 && p.dense_issue;
    unsigned a_voff[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j)
        a_voff[j] = (a_mask[j] & 1u) ? (unsigned)(a_pix[j] * (p.c1 * 2)) + a_cb[j] : MDX_OOB;

    auto stage_tile = [&](int kt, int buf) {
        char* sbase = smem + buf * STAGE;
        const int k0 = kt * BK;
        if (dense) {
#pragma unroll
            for (int j = 0; j < AJ; ++j) dma16s(rs_a, sbase + (wave * AJ + j) * 1024, a_voff[j], (unsigned)k0 * 2u);
#pragma unroll
            for (int j = 0; j < BJ; ++j) dma16s(rs_w, sbase + A_BYTES + (wave * BJ + j) * 1024, b_off[j], (unsigned)kt * 8192u);
            return;
        }
        if constexpr (FASTK) {
            // K is ordered [cin chunk of 64][tap][64] (weights packed to match; Cin % 64 == 0, c1 % 64 == 0), so a K
            // tile lies in one tap and one source, and consecutive K tiles re-read the SAME pixels' cache lines
            // shifted by one tap: the 9x im2col re-read of the activations is served by L2, not by the fabric.
            const int taps = p.ksize * p.ksize;
            const int g = k0 >> 6;
            const int chunk = g / taps;
            const int tap = g - chunk * taps;
            int ci0 = chunk * 64 + (k0 & 63);
            const int ky = (p.ksize == 3) ? tap / 3 : 0;
            const int kx = (p.ksize == 3) ? tap - ky * 3 : 0;
            const bool second = ci0 >= p.c1;
            const int cs2 = (second ? p.c2 : p.c1) * 2;
            if (second) ci0 -= p.c1;
            if (!p.upsample) {
                const int dpix = ky * p.W + kx;
#pragma unroll
                for (int j = 0; j < AJ; ++j) {
                    const unsigned off = ((a_mask[j] >> tap) & 1u)
                                             ? (unsigned)((a_pix[j] + dpix) * cs2) + (unsigned)(ci0 * 2) + a_cb[j]
                                             : MDX_OOB;
                    void* dst = sbase + (wave * AJ + j) * 1024;
                    if (second)
                        dma16(rs_a2, dst, off);
                    else
                        dma16(rs_a, dst, off);
                }
            } else {
#pragma unroll
                for (int j = 0; j < AJ; ++j) {
                    const int yi = (a_y[j] + ky) >> 1, xi = (a_x[j] + kx) >> 1;
                    const unsigned off = ((a_mask[j] >> tap) & 1u)
                                             ? (unsigned)((a_b[j] + yi * p.W + xi) * cs2) + (unsigned)(ci0 * 2) + a_cb[j]
                                             : MDX_OOB;
                    void* dst = sbase + (wave * AJ + j) * 1024;
                    if (second)
                        dma16(rs_a2, dst, off);
                    else
                        dma16(rs_a, dst, off);
                }
            }
        } else {
            // generic: tap-major K, per-lane tap decode (single source only; conv_in with Cin = 8)
#pragma unroll
            for (int j = 0; j < AJ; ++j) {
                const int kk = k0 + (int)(a_cb[j] >> 1);
                const int tap = kk / p.cin;
                const int ci = kk - tap * p.cin;
                const int ky = (p.ksize == 3) ? tap / 3 : 0;
                const int kx = (p.ksize == 3) ? tap - ky * 3 : 0;
                int yi = a_y[j] + ky, xi = a_x[j] + kx;
                const bool ok = kk < p.K && ((a_mask[j] >> tap) & 1u);
                if (p.upsample) {
                    yi >>= 1;
                    xi >>= 1;
                }
                const unsigned off = ok ? (unsigned)(((size_t)(a_b[j] + yi * p.W + xi) * p.cin + ci) * 2) : MDX_OOB;
                dma16(rs_a, sbase + (wave * AJ + j) * 1024, off);
            }
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) dma16s(rs_w, sbase + A_BYTES + (wave * BJ + j) * 1024, b_off[j], (unsigned)kt * 8192u);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read offsets: row*ROWB + ((2s+hi) ^ key(row))*16; key(row) depends only on the lane for all our row bases
    const int swz = (l31 / RP256) % CPRW;
    const int a_row_off = (wm * WROWS + l31) * ROWB;
    const int b_row_off = A_BYTES + (wn * (BN / 2) + l31) * ROWB;

    // ---- main loop: NS-stage LDS ring.  DMAs of the next NS-1 K tiles stay in flight ACROSS the per-tile barrier
    // (counted s_waitcnt vmcnt + raw s_barrier: __syncthreads() would drain them), fragments for k-step s+1 are
    // fetched from LDS while the MFMAs of k-step s run.
    //   iteration t:  wait(tile t landed for this wave) -> s_barrier (=> landed for all waves, and every wave is
    //                 done reading stage (t-1)%NS) -> issue tile t+NS-1 into stage (t-1)%NS -> compute tile t.
    constexpr int LPT = AJ + BJ;  // DMA instructions per wave per K tile
    const int nt = kt_end - kt_begin;
    float bpre[16];
    gemm_bias_prefetch<BN, SWAP, NW>(p, n0, bpre);
    // LayerNorm-fold consumer (round 5): the epilogue opens with the tile's row statistics (BM x K / 64 {sum, sumsq} pairs, written by
    // the previous launch on other XCDs) and S[n] (a weight-like vector, cold in HBM): 2-3 us of misses at the head of the epilogue of
    // every q|k|v, cross-attention q and GEGLU ff1 launch, which one block per CU cannot hide.  Each thread TOUCHES one 128-byte line
    // of them here, ahead of the prologue DMAs (older in the vmcnt order: the counted waits of the main loop cover them), so that
    // the epilogue's own loads hit this XCD's L2.  Two VGPRs, kept live to the epilogue so that the loads are not sunk; the values
    // are not used -- the arithmetic is the epilogue's, unchanged.
    [[maybe_unused]] float touch_ln = 0.f, touch_s = 0.f;
    if constexpr (SWAP) {
        if (p.ln_stats != nullptr && p.ln_prefetch) {
            const int rows = min(BM, p.M - m0);
            const unsigned bytes = (unsigned)rows * (unsigned)p.ln_nt * 8u;
            const char* base = reinterpret_cast<const char*>(p.ln_stats) + (size_t)m0 * p.ln_nt * 8;
            // (one line per thread covers BM x K / 64 pairs up to K = 2048; beyond that the rest stays cold.  No loop, no add: a
            // dependent instruction here would make the compiler wait for the miss in front of the prologue)
            if ((unsigned)tid * 128u < bytes) touch_ln = *reinterpret_cast<const float*>(base + (unsigned)tid * 128u);
            if (tid < BN / 32 && n0 + tid * 32 < p.N) touch_s = p.ln_s[n0 + tid * 32];
        }
    }
#pragma unroll
    for (int i = 0; i < NS - 1; ++i)
        if (i < nt) stage_tile(kt_begin + i, i);
    // GNA tables live behind everything the ring and the epilogue staging use: [cin] a, [cin] s (fp16), [cin] {sum, sumsq} scratch
    constexpr size_t GNA_RING = (size_t)NS * (BM + BN) * BK * 2;
    constexpr size_t GNA_EPI = (size_t)(BM > BN ? BM : BN) * ((BM > BN ? BN : BM) + 8) * 2 + 4096;
    constexpr size_t GNA_BASE = GNA_RING > GNA_EPI ? GNA_RING : GNA_EPI;
    [[maybe_unused]] const f16* gta = reinterpret_cast<const f16*>(smem + GNA_BASE);
    [[maybe_unused]] const f16* gts = gta + p.cin;
    if constexpr (GNA) {
        // (the prologue DMAs above are already in flight: the fold's global round trip overlaps them)
        f16* wa = reinterpret_cast<f16*>(smem + GNA_BASE);
        f16* wsft = wa + p.cin;
        float2* csum = reinterpret_cast<float2*>(wsft + p.cin);
        const int pb = m0 / p.HoWo;
        const int cpg = p.cin >> 5;
        const float2* src = reinterpret_cast<const float2*>(p.gn_cs) + (size_t)pb * p.gn_nrb * p.cin;
        // gamma / beta of this thread's channels first: they are cold in HBM (weights stream through the caches between two uses),
        // and fetched here their miss overlaps the partials' round trip instead of following it
        constexpr int GNA_CPT = 2560 / (NW * 64);      // channels per thread at the largest Cin
        float gg[GNA_CPT], gb[GNA_CPT];
#pragma unroll
        for (int i = 0; i < GNA_CPT; ++i) {
            const int c = tid + i * NW * 64;
            gg[i] = c < p.cin ? p.gn_gamma[c] : 0.f;
            gb[i] = c < p.cin ? p.gn_beta[c] : 0.f;
        }
        for (int c = tid; c < p.cin; c += NW * 64) {
            float su = 0.f, sq = 0.f;
            int k = 0;
            for (; k + 8 <= p.gn_nrb; k += 8) {
                float2 v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = src[(size_t)(k + e) * p.cin + c];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    su += v[e].x;
                    sq += v[e].y;
                }
            }
            for (; k < p.gn_nrb; ++k) {
                const float2 v = src[(size_t)k * p.cin + c];
                su += v.x;
                sq += v.y;
            }
            csum[c] = make_float2(su, sq);
        }
        __syncthreads();
        const float inv = 1.0f / ((float)cpg * (float)p.HoWo);
#pragma unroll
        for (int i = 0; i < GNA_CPT; ++i) {
            const int c = tid + i * NW * 64;
            if (c < p.cin) {
                const int g = c / cpg;
                float su = 0.f, sq = 0.f;
                for (int e = 0; e < cpg; ++e) {
                    const float2 v = csum[g * cpg + e];
                    su += v.x;
                    sq += v.y;
                }
                const float mean = su * inv;
                float var = sq * inv - mean * mean;
                var = var < 0.f ? 0.f : var;
                const float a = gg[i] * rsqrtf(var + p.gn_eps);
                wa[c] = (f16)a;
                wsft[c] = (f16)(gb[i] - mean * a);
            }
        }
        __syncthreads();
    }
    int rd = 0;            // stage holding tile t
    int wr = NS - 1;       // stage that tile t+NS-1 goes to
    trace_mark(p, 1);
    for (int t = 0; t < nt; ++t) {
        // tiles t .. min(t+NS-2, nt-1) are outstanding; allow all but the oldest to stay in flight
        const int ahead = min(NS - 2, nt - 1 - t);
        if (NS >= 6 && ahead >= 4)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * LPT) : "memory");
        else if (NS >= 5 && ahead == 3)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPT) : "memory");
        else if (NS >= 4 && ahead == 2)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");
        else if (NS >= 3 && ahead == 1)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t == 0) trace_mark(p, 2);
        // Order inside a K tile (round 5).  A block that has its SIMDs to itself -- nearly every launch at UNet batch 2 -- runs the
        // chain wait -> barrier -> DMA issue -> ds_read -> MFMA serially, one instruction per ~3 ns, and the ISA of round 4 showed
        // the machine scheduler sinking every fragment read to its use: each MFMA waited for a ds_read issued right before it, KS
        // LDS round trips per K tile with nothing to hide them.  Now: (i) the ring-refill DMAs are issued FIRST (the operand
        // stream is latency x ring-depth bound: with the reads in front of the issue the convs lost 10-35 %), (ii) then the
        // fragment reads of the tile -- of ALL KS k-steps on tiles with at most two MFMAs per k-step, which have nothing to hide a
        // read behind; a double buffer on 128 x 128 -- pinned where they are written by sched_barriers, (iii) then the MFMAs.
        const char* sb = smem + rd * STAGE;
        constexpr bool ALLK = (TM * TN <= 2) && !GNA;
        constexpr int FD = ALLK ? KS : 2;
        static_assert(KS == 4, "64-deep K tiles");
        f16x8 af[FD][TM], bf[FD][TN];
        [[maybe_unused]] f16x8 gna[2], gns[2];      // GNA: this lane's 8 channels' {a, s} of k-step s (channel = K index: dense launch)
        [[maybe_unused]] const int gk = (kt_begin + t) * BK + hi * 8;
        auto rdfrag = [&](auto slot_c, const int ks) {
            constexpr int slot = decltype(slot_c)::value;
            const int coff = (((2 * ks + hi) ^ swz) << 4);
#pragma unroll
            for (int i = 0; i < TM; ++i) af[slot][i] = *reinterpret_cast<const f16x8*>(sb + a_row_off + i * 32 * ROWB + coff);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[slot][j] = *reinterpret_cast<const f16x8*>(sb + b_row_off + j * 32 * ROWB + coff);
            if constexpr (GNA) {
                gna[slot & 1] = *reinterpret_cast<const f16x8*>(gta + gk + 16 * ks);
                gns[slot & 1] = *reinterpret_cast<const f16x8*>(gts + gk + 16 * ks);
            }
        };
        auto mfmas = [&](auto slot_c) {
            constexpr int slot = decltype(slot_c)::value;
            if constexpr (GNA) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[slot][i] = af[slot][i] * gna[slot & 1] + gns[slot & 1];      // v_pk_fma_f16 x 4
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (SWAP)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[slot][j], af[slot][i], acc[i][j], 0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[slot][i], bf[slot][j], acc[i][j], 0, 0, 0);
                }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        const bool refill = t + NS - 1 < nt;
        if (refill) stage_tile(kt_begin + t + NS - 1, wr);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ALLK) {
            using I2 = std::integral_constant<int, 2>;
            using I3 = std::integral_constant<int, 3>;
            rdfrag(I0{}, 0); rdfrag(I1{}, 1); rdfrag(I2{}, 2); rdfrag(I3{}, 3);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(I0{}); mfmas(I1{}); mfmas(I2{}); mfmas(I3{});
        } else {
            rdfrag(I0{}, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS; ks += 2) {
                rdfrag(I1{}, ks + 1);
                __builtin_amdgcn_sched_barrier(0);
                mfmas(I0{});
                __builtin_amdgcn_sched_barrier(0);
                if (ks + 2 < KS) {
                    rdfrag(I0{}, ks + 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
                mfmas(I1{});
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        rd = (rd + 1 == NS) ? 0 : rd + 1;
        wr = (wr + 1 == NS) ? 0 : wr + 1;
    }

    __syncthreads();  // all waves done with the ring before the epilogue reuses it
    trace_mark(p, 3);
    if constexpr (SWAP) asm volatile("" ::"v"(touch_ln), "v"(touch_s));      // (the touches above stay loads issued up there)
    gemm_epilogue<BM, BN, SWAP, NW>(p, acc, smem, LinearRows{m0}, n0, split, bpre, tile_m, tile_id);
    trace_mark(p, 4);
}

// ---------------------------------------------------------------------------------------------------------------
// HALO conv kernel: 3x3, stride 1, pad 1, single source, Cin % 64 == 0, H % 8 == 0, W % 16 == 0.
//
// The generic kernel DMAs a fresh 128 x 64 activation tile for each of the 9 taps (L2 serves the re-read, but the
// per-CU DMA rate -- the measured limiter, profiles/r01_dma_probe.txt -- still pays 9x).  Here the M tile is an
// 8 x 16 PIXEL PATCH: for each 64-channel chunk the 10 x 18 halo (180 rows x 128 B, zero padding supplied by the
// buffer bounds check) is brought into LDS ONCE and the A fragments of all 9 taps are read from it at
// halo row (py+ky)*18 + (px+kx).  Activation DMA bytes drop 9 x 16 KiB -> 24 KiB per chunk (6x); with the weight
// tiles (unchanged, one per tap) total DMA bytes per chunk drop 288 -> 168 KiB (BN = 128), 216 -> 96 KiB (BN = 64).
//
// LDS: halo[2][192 rows][128 B] (double buffered across chunks; same XOR swizzle keyed by the halo row) followed by an
// NSB-stage ring of weight tiles.  The next chunk's halo is fetched one DMA instruction per wave per tap during
// taps 0..5 of the current chunk, so the DMA stream stays even.
//
// BM = 128: 8 x 16 patch, 4 waves (2 blocks per CU).  BM = 256: 16 x 16 patch, 8 waves (one block per CU): the halo grows
// 180 -> 324 rows for twice the pixels and every weight tile is shared by twice as many rows, so the DMA bytes per MAC --
// what bounds this kernel (about 60 GB/s per CU, 6 TB/s chip-wide through buffer_load..lds) -- drop by another 45 %.
//
// PW = 8 (BM = 128 only): the 8 x 8-pixel images of the deepest UNet level (64 x 64 latents: 16 of the 49 stride-1 convs,
// K = 11520 / 23040, M = 64 per sample).  An M tile is TWO WHOLE SAMPLES (2 x 64 pixels); each sample's 10 x 10 halo
// (zero padding from the buffer bounds check) sits in LDS once per 64-channel chunk: 200 halo rows instead of nine
// 128-row im2col tiles -- the activation share of the per-CU DMA stream (what bounds a lone block per CU,
// profiles/r01_dma_probe.txt) drops from 144 to 25 KiB per chunk, 288 -> 169 KiB in all at BN = 128.
//
// BDIR (weight-streaming form, for M <= 512 where every launch is bound by how fast it pulls cold weights): the weights are
// packed in MFMA-fragment order ([N / 32][K / 16][64 lanes][8], ops.pack_frag_weight) and go HBM/L2 -> VGPR directly, PFB
// pieces of 1 KiB in flight per wave and column tile -- no weight tiles in LDS, no weight DMA, and ONE barrier per 64-channel
// chunk (the halo swap) instead of one per tap.  With the LDS ring a block keeps two 8 KiB weight tiles in flight (32 KiB
// per CU: ~2 TB/s chip-wide at the HBM round trip); here 4 waves x 12 KiB x 2 blocks = 96 KiB per CU.  Same k order per
// output, same epilogue: bit-identical results.
// W4 (BDIR, 128 x 128 tiles, slab output only): the four waves sit side by side along N -- each owns 32 columns and ALL 128 rows --
// instead of 2 x 2.  In the 2 x 2 arrangement the two waves of a column pair fetch the same weight pieces, so half of the bytes
// a CU has in flight are duplicates; side by side every piece is fetched once (the A fragments, which come from LDS, are read
// by all four waves instead).
template <int BM, int BN, int NSB, bool SWAP, int PW, bool BDIR = false, bool W4 = false>
__global__ __launch_bounds__(BM * 2, BM == 128 ? 2 : 1) void conv3x3_halo_kernel(const GemmParams p) {
    static_assert(PW == 16 || (PW == 8 && BM == 128), "patch width 16, or 8 with two-sample tiles");
    static_assert(!W4 || (BDIR && BM == 128 && BN == 128 && !SWAP), "W4: weight-streaming 128 x 128 tiles with slab output");
    constexpr int NW = BM / 32;              // waves: (NW/2) x 2, each 64 patch pixels x BN/2 channels
    constexpr int PH = BM / 16;              // PW = 16: patch rows (8 | 16)
    constexpr int HWD = PW + 2;              // halo width in pixels (18 | 10)
    constexpr int WROWS = W4 ? BM : BM / (NW / 2);     // patch pixels per wave row (64; W4: all 128)
    constexpr int TM = WROWS / 32;
    constexpr int TN = W4 ? 1 : BN / 64;
    constexpr int HROWS = PW == 16 ? (PH + 2) * 18 : (BM / 64) * 100;   // halo pixels (180 | 324 | 200)
    constexpr int HINST = (HROWS + 7) / 8;   // halo DMA instructions per chunk (23 | 41)
    constexpr int HALO_BYTES = HINST * 1024;
    constexpr int HJ = (HINST + NW - 1) / NW;   // ... per wave (6 | 6)
    constexpr int B_BYTES = BN * 128;
    constexpr int BJ = BN / 8 / NW;
    static_assert(BJ >= 1, "weight tile too small for this many loader waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    mdx_kernarg_touch<sizeof(GemmParams)>();

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = W4 ? 0 : wave >> 1, wn = W4 ? wave : wave & 1;
    const int hi = lane >> 5;
    const int l31 = lane & 31;

    // SPREAD order (see gemm_kernel): a single row of M tiles deals its (tile, split) items round-robin to the XCDs
    const int ntiles = p.tiles_m * p.tiles_n;
    const int tile_id = p.spread ? (int)(blockIdx.x % (unsigned)ntiles) : (blockIdx.x & 7) * p.tiles_per_xcd + (blockIdx.x >> 3);
    if (tile_id >= ntiles) return;
    int tile_m, tile_n;
    if (p.n_fastest) {
        tile_m = tile_id / p.tiles_n;
        tile_n = tile_id - tile_m * p.tiles_n;
    } else {
        tile_n = tile_id / p.tiles_m;
        tile_m = tile_id - tile_n * p.tiles_m;
    }
    trace_mark(p, 0);
    const int n0 = tile_n * BN;
    // PW = 16: tile = one PH x 16 patch of sample pb.  PW = 8: tile = samples 2 * tile_m and 2 * tile_m + 1 (8 x 8 pixels each).
    const int pw = PW == 16 ? p.W >> 4 : 1, ph = PW == 16 ? p.H / PH : 1;
    const int pb = PW == 16 ? tile_m / (ph * pw) : tile_m * (BM / 64);
    const int prem = PW == 16 ? tile_m - pb * (ph * pw) : 0;
    const int py0 = (prem / pw) * PH, px0 = (prem % pw) * 16;
    const int split = p.spread ? (int)(blockIdx.x / (unsigned)ntiles) : (int)blockIdx.y;
    const int kt_begin = split * p.ktiles_per_split;            // multiples of 9 (host guarantees chunk-aligned splits)
    const int kt_end = min(p.ktiles, kt_begin + p.ktiles_per_split);
    const int nt = kt_end - kt_begin;
    const int c_begin = kt_begin / 9, c_end = kt_end / 9;

    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(p.a, p.a_bytes);
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, p.w_bytes);

    // halo loader: DMA instruction q of this wave covers halo rows (wave*HJ + q)*8 .. +7, lane -> row + lane/8.
    // Swizzle: physical chunk = logical chunk ^ ((halo column >> 1) & 7).  A ds_read_b128 lane group reads 16 pixels
    // that span two patch rows (halo rows 18 apart) but 16 CONSECUTIVE columns, so keying on the column (not on the
    // linear halo row) puts them on 16 distinct 16-B slots of the 256-B bank row: conflict-free.
    const int lrow = lane >> 3, lchk = lane & 7;
    unsigned hal_off[HJ];
    const unsigned row_bytes = (unsigned)p.cin * 2;
#pragma unroll
    for (int q = 0; q < HJ; ++q) {
        const int hp = (wave * HJ + q) * 8 + lrow;
        if constexpr (PW == 16) {
            const int hr = hp / 18, hc = hp - hr * 18;
            const int y = py0 - 1 + hr, x = px0 - 1 + hc;
            const bool ok = hp < HROWS && y >= 0 && y < p.H && x >= 0 && x < p.W;
            hal_off[q] = ok ? (unsigned)((pb * p.H + y) * p.W + x) * row_bytes + (unsigned)((lchk ^ ((hc >> 1) & 7)) * 16)
                            : MDX_OOB;
        } else {
            // two 10 x 10 halos back to back.  Swizzle key = (halo column / 2) % 4 | (halo row % 2) << 2: a ds_read_b128
            // lane group covers four 4-pixel runs on four consecutive patch rows (8-wide rows), and this key sends its
            // eight even and eight odd halo pixels to eight distinct 16-B slots each (conflict-free, as for PW = 16)
            const int sm = hp / 100, r = hp - sm * 100;
            const int hr = r / 10, hc = r - hr * 10;
            const int y = hr - 1, x = hc - 1;
            const bool ok = hp < HROWS && pb + sm < p.B && y >= 0 && y < 8 && x >= 0 && x < 8;
            const int key = ((hc >> 1) & 3) | ((hr & 1) << 2);
            hal_off[q] = ok ? (unsigned)(((pb + sm) * 8 + y) * 8 + x) * row_bytes + (unsigned)((lchk ^ key) * 16) : MDX_OOB;
        }
    }
    unsigned b_off[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int row = (wave * BJ + j) * 8 + lrow;
        const int panel = (n0 >> 6) + (row >> 6);
        b_off[j] = (unsigned)(((size_t)panel * p.kt64) * 8192 + ((row & 63) * 8 + lchk) * 16);
    }
    auto dma_halo = [&](int q, int chunk, int hb) {
        if (wave * HJ + q >= HINST) return;   // (wave-uniform) the last wave's spare slots lie beyond the halo image
        const unsigned off = hal_off[q] == MDX_OOB ? MDX_OOB : hal_off[q] + (unsigned)chunk * 128u;
        dma16(rs_a, smem + hb * HALO_BYTES + (wave * HJ + q) * 1024, off);
    };
    auto dma_b = [&](int kt, int stage) {
        char* sb = smem + 2 * HALO_BYTES + stage * B_BYTES;
#pragma unroll
        for (int j = 0; j < BJ; ++j) dma16s(rs_w, sb + (wave * BJ + j) * 1024, b_off[j], (unsigned)kt * 8192u);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addressing: A rows are patch pixels q = wm*64 + i*32 + l31 -> halo row of tap (0,0) = (q>>4)*18 + (q&15)
    int hp0[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int q = wm * WROWS + i * 32 + l31;
        hp0[i] = PW == 16 ? (q >> 4) * 18 + (q & 15) : (q >> 6) * 100 + ((q >> 3) & 7) * 10 + (q & 7);
    }
    const int swz_b = (l31 >> 1) & 7;
    const int b_row_off = 2 * HALO_BYTES + (wn * (BN / 2) + l31) * 128;

    float bpre[16];
    gemm_bias_prefetch<BN, SWAP, NW>(p, n0, bpre);

    if constexpr (BDIR) {
        constexpr int PFB = 12;                                  // divides the 36 k-steps of a chunk: static ring positions
        const unsigned kpieces = (unsigned)p.kt64 * 4u;          // 1 KiB pieces (16-wide k-steps) per 32-column tile
        unsigned pbase[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
            pbase[j] = (unsigned)((n0 + (W4 ? wn * 32 : wn * (BN / 2) + j * 32)) >> 5) * kpieces + (unsigned)kt_begin * 4u;
        const unsigned voffw = (unsigned)lane * 16u;
        // halo first, then the ring: the halo DMAs are then older than the PFB * TN youngest loads at every chunk boundary
#pragma unroll
        for (int q = 0; q < HJ; ++q) dma_halo(q, c_begin, 0);
        u32x4 ring[TN][PFB];
#pragma unroll
        for (int q = 0; q < PFB; ++q)
#pragma unroll
            for (int j = 0; j < TN; ++j) ring[j][q] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, voffw, (pbase[j] + q) * 1024u, 0);
        trace_mark(p, 1);
        unsigned step = 0;
        for (int c = c_begin; c < c_end; ++c) {
            const int hb = (c - c_begin) & 1;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PFB * TN) : "memory");     // this chunk's halo has landed (own DMAs)
            __builtin_amdgcn_s_barrier();                                      // ... everyone's; buffer hb ^ 1 is free
            if (c + 1 < c_end) {
#pragma unroll
                for (int q = 0; q < HJ; ++q) dma_halo(q, c + 1, hb ^ 1);
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - ky * 3;
                const int dq = ky * HWD + kx;
                int a_row[TM], a_key[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    a_row[i] = hb * HALO_BYTES + (hp0[i] + dq) * 128;
                    if constexpr (PW == 16)
                        a_key[i] = ((((l31 & 15) + kx) >> 1) & 7) << 4;
                    else
                        a_key[i] = (((((l31 & 7) + kx) >> 1) & 3) | ((((l31 >> 3) + ky) & 1) << 2)) << 4;
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    f16x8 af[TM];
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        af[i] = *reinterpret_cast<const f16x8*>(smem + a_row[i] + (((2 * s + hi) << 4) ^ a_key[i]));
                    constexpr int dummy = 0;
                    (void)dummy;
                    const int rp = (tap * 4 + s) % PFB;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const f16x8 b = __builtin_bit_cast(f16x8, ring[j][rp]);
                        ring[j][rp] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, voffw, (pbase[j] + step + (unsigned)(tap * 4 + s) + PFB) * 1024u, 0);
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            if constexpr (SWAP)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, af[i], acc[i][j], 0, 0, 0);
                            else
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], b, acc[i][j], 0, 0, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);      // keep every refill where it is written (else they sink to their use)
                }
            }
            step += 36;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // ring refills past the end of the range are still in flight
        __syncthreads();
        trace_mark(p, 3);
        if constexpr (W4) {
            // split-K partial slab [split][M][N], C layout (lane -> column, registers -> rows): 128-byte row segments per store
            float* wsz = p.ws + (size_t)split * p.M * p.N;
            const int n = n0 + wn * 32 + l31;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    int m;
                    if constexpr (PW == 16)
                        m = (pb * p.H + py0) * p.W + px0 + (row >> 4) * p.W + (row & 15);
                    else
                        m = tile_m * BM + row;
                    if (m < p.M && n < p.N) wsz[(size_t)m * p.N + n] = acc[i][0][r];
                }
            trace_mark(p, 4);
            return;
        } else {
            if constexpr (PW == 16)
                gemm_epilogue<BM, BN, SWAP, NW, PatchRows, 0, HALO_EMODE>(p, acc, smem, PatchRows{(pb * p.H + py0) * p.W + px0, p.W, pb}, n0,
                                                                          split, bpre, tile_m, tile_id);
            else
                gemm_epilogue<BM, BN, SWAP, NW>(p, acc, smem, LinearRows{tile_m * BM}, n0, split, bpre, tile_m, tile_id);
            trace_mark(p, 4);
            return;
        }
    }
    if constexpr (!BDIR) {
    // ---- GroupNorm(32) [+ SiLU] of the INPUT inside the conv (nn.GroupNorm -> SiLU -> Conv2d of a ResBlock, openaimodel.py:
    // 136-138, 159-163): the producer of the input emitted per-row-block column partials (colstats); this block folds its
    // sample's partials into a per-channel {scale, shift} table in LDS and normalises every halo slice IN PLACE after it has
    // landed (each thread the 16-byte pieces its own DMAs brought, so no extra barrier): the normalised tensor is never
    // written or read, and the GroupNorm launch disappears.  Zero padding stays zero (pieces outside the image are skipped).
    float2* gn_tab = reinterpret_cast<float2*>(smem + 2 * HALO_BYTES + NSB * B_BYTES);     // [Cin] {a, shift}
    if constexpr (PW == 16) {
        if (p.gn_cs) {
            float2* csum = reinterpret_cast<float2*>(smem);      // [Cin] channel {sum, sumsq}: the halo area is still unused
            const int cpg = p.cin >> 5;
            const float2* src = reinterpret_cast<const float2*>(p.gn_cs) + (size_t)pb * p.gn_nrb * p.cin;
            for (int c = tid; c < p.cin; c += NW * 64) {
                float su = 0.f, sq = 0.f;
                int k = 0;
                for (; k + 8 <= p.gn_nrb; k += 8) {
                    float2 v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = src[(size_t)(k + e) * p.cin + c];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        su += v[e].x;
                        sq += v[e].y;
                    }
                }
                for (; k < p.gn_nrb; ++k) {
                    const float2 v = src[(size_t)k * p.cin + c];
                    su += v.x;
                    sq += v.y;
                }
                csum[c] = make_float2(su, sq);
            }
            __syncthreads();
            for (int c = tid; c < p.cin; c += NW * 64) {
                const int g = c / cpg;
                float su = 0.f, sq = 0.f;
                for (int e = 0; e < cpg; ++e) {
                    const float2 v = csum[g * cpg + e];
                    su += v.x;
                    sq += v.y;
                }
                const float inv = 1.0f / ((float)cpg * (float)(p.H * p.W));
                const float mean = su * inv;
                float var = sq * inv - mean * mean;
                var = var < 0.f ? 0.f : var;
                const float a = p.gn_gamma[c] * rsqrtf(var + p.gn_eps);
                gn_tab[c] = make_float2(a, p.gn_beta[c] - mean * a);
            }
            __syncthreads();      // table complete; the halo area may be overwritten by the prologue DMAs
        }
    }
    auto gn_transform = [&](int chunk, int hb) {      // normalise this thread's own pieces of halo buffer hb (chunk `chunk`)
#pragma unroll
        for (int q = 0; q < HJ; ++q) {
            if (wave * HJ + q >= HINST || hal_off[q] == MDX_OOB) continue;
            char* pc = smem + hb * HALO_BYTES + (wave * HJ + q) * 1024 + lane * 16;
            const int ch0 = chunk * 64 + (int)((hal_off[q] >> 4) & 7u) * 8;       // logical chunk of this lane's piece
            const f16x8 x = *reinterpret_cast<const f16x8*>(pc);
            f16x8 y;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float2 t = gn_tab[ch0 + e];
                const float v = (float)x[e] * t.x + t.y;
                y[e] = (f16)(p.gn_silu ? silu_f(v) : v);
            }
            *reinterpret_cast<f16x8*>(pc) = y;
        }
    };
    // prologue: whole halo of the first chunk + the first NSB-1 weight tiles
#pragma unroll
    for (int q = 0; q < HJ; ++q) dma_halo(q, c_begin, 0);
#pragma unroll
    for (int i = 0; i < NSB - 1; ++i)
        if (i < nt) dma_b(kt_begin + i, i);

    int t = 0;
    int rd = 0, wr = NSB - 1;
    trace_mark(p, 1);
    for (int c = c_begin; c < c_end; ++c) {
        const int hb = (c - c_begin) & 1;
        const bool more = c + 1 < c_end;
#pragma unroll      // (the pinned form is not unrolled on the compiler's own judgement; the tap decode is nine constants only when it is:
                    // rolled, the convs of UNet batch 2 were 10-35 % slower -- profiles/r05_kloop_ab.txt)
        for (int tap = 0; tap < 9; ++tap, ++t) {
            // weight tiles t .. t+NSB-2 (and at most one halo slice per tap, each older than the weight tile issued with it) are
            // outstanding: all but the youngest (tiles ahead) x BJ instructions must have landed -- that covers tile t and, with a
            // four-stage ring, one instruction of tile t+1 when a halo slice sits between the two (conservative, never short)
            if (NSB >= 4 && t + 2 < nt)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * BJ) : "memory");
            else if (NSB >= 3 && t + 1 < nt)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BJ) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (PW == 16) {
                if (tap == 0 && p.gn_cs) {     // own slices have landed (they are older than the newest weight tile)
                    gn_transform(c, hb);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // ... and are rewritten before anyone passes the barrier
                }
            }
#if MDX_HALO_ABLATE != 4
            __builtin_amdgcn_s_barrier();
#endif
            if (t == 0) trace_mark(p, 2);
            // (round 5: DMA issue first -- the weight stream is latency x ring-depth bound, every instruction in front of the issue is
            // added to the chain: measured +10...35 % with the fragment reads in front -- then the fragment reads, pinned by
            // sched_barriers: see gemm_kernel's main loop)
#if MDX_HALO_ABLATE != 3
            if (more && tap < HJ) dma_halo(tap, c + 1, hb ^ 1);
            if (t + NSB - 1 < nt) dma_b(kt_begin + t + NSB - 1, wr);
#endif
            __builtin_amdgcn_sched_barrier(0);
            const int ky = tap / 3, kx = tap - ky * 3;
            const int dq = ky * HWD + kx;
            int a_row[TM], a_key[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int hp = hp0[i] + dq;
                a_row[i] = hb * HALO_BYTES + hp * 128;
                if constexpr (PW == 16)
                    a_key[i] = ((((l31 & 15) + kx) >> 1) & 7) << 4;   // swizzle key = halo COLUMN / 2 (see the loader)
                else   // halo column (l31 & 7) + kx, halo row parity ((l31 >> 3) + ky) & 1 (i * 4 rows keep the parity)
                    a_key[i] = (((((l31 & 7) + kx) >> 1) & 3) | ((((l31 >> 3) + ky) & 1) << 2)) << 4;
            }
            const char* sb = smem + rd * B_BYTES;
            constexpr bool ALLK = TM * TN <= 2;      // at most two MFMAs per k-step: nothing to hide a fragment read behind
            constexpr int FD = ALLK ? 4 : 2;
            f16x8 af[FD][TM], bf[FD][TN];
            auto rdfrag = [&](auto slot_c, const int ks) {
                constexpr int slot = decltype(slot_c)::value;
#if MDX_HALO_ABLATE == 2
#pragma unroll
                for (int i = 0; i < TM; ++i) { f16x8 tt; asm volatile("" : "=v"(tt)); af[slot][i] = tt; }
#pragma unroll
                for (int j = 0; j < TN; ++j) { f16x8 tt; asm volatile("" : "=v"(tt)); bf[slot][j] = tt; }
#else
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[slot][i] = *reinterpret_cast<const f16x8*>(smem + a_row[i] + (((2 * ks + hi) << 4) ^ a_key[i]));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bf[slot][j] = *reinterpret_cast<const f16x8*>(sb + b_row_off + j * 32 * 128 + (((2 * ks + hi) ^ swz_b) << 4));
#endif
            };
            auto mfmas = [&](auto slot_c) {
                constexpr int slot = decltype(slot_c)::value;
#if MDX_HALO_ABLATE == 1
#pragma unroll
                for (int i = 0; i < TM; ++i) { const f16x8 tt = af[slot][i]; asm volatile("" ::"v"(tt)); }
#pragma unroll
                for (int j = 0; j < TN; ++j) { const f16x8 tt = bf[slot][j]; asm volatile("" ::"v"(tt)); }
#else
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if constexpr (SWAP)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[slot][j], af[slot][i], acc[i][j], 0, 0, 0);
                        else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[slot][i], bf[slot][j], acc[i][j], 0, 0, 0);
                    }
#endif
            };
            using I0 = std::integral_constant<int, 0>;
            using I1 = std::integral_constant<int, 1>;
            if constexpr (ALLK) {
                using I2 = std::integral_constant<int, 2>;
                using I3 = std::integral_constant<int, 3>;
                rdfrag(I0{}, 0); rdfrag(I1{}, 1); rdfrag(I2{}, 2); rdfrag(I3{}, 3);
                __builtin_amdgcn_sched_barrier(0);
                mfmas(I0{}); mfmas(I1{}); mfmas(I2{}); mfmas(I3{});
            } else {
                rdfrag(I0{}, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < 4; ks += 2) {
                    rdfrag(I1{}, ks + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    mfmas(I0{});
                    __builtin_amdgcn_sched_barrier(0);
                    if (ks + 2 < 4) {
                        rdfrag(I0{}, ks + 2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    mfmas(I1{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            rd = (rd + 1 == NSB) ? 0 : rd + 1;
            wr = (wr + 1 == NSB) ? 0 : wr + 1;
        }
    }

    if (p.skip_w) {
        // ---- ResBlock skip_connection (openaimodel.py:174, 201-205: conv1x1 over the block's RAW input, the virtual concat of
        // x and the UNet skip tensor on the up path) as extra K tiles of this launch: out = conv3x3(h) + conv1x1(x), one
        // accumulator, one epilogue, no second launch and no fp16 round trip of the skip tensor.  Generic 128-byte-row A tiles
        // of this block's patch pixels (rows = patch pixels, 64 channels per tile) in the halo area, weight tiles in the ring
        // area, two stages, one barrier per tile; this split's share of the skip K tiles.
        constexpr int SA_BYTES = BM * 128;
        static_assert(2 * SA_BYTES <= 2 * HALO_BYTES && NSB >= 2, "skip tiles reuse the halo / ring areas");
        const __amdgpu_buffer_rsrc_t rs_s1 = make_rsrc(p.skip_a, p.skip_a_bytes);
        const __amdgpu_buffer_rsrc_t rs_s2 = make_rsrc(p.skip_a2 ? p.skip_a2 : p.skip_a, p.skip_a2 ? p.skip_a2_bytes : p.skip_a_bytes);
        const __amdgpu_buffer_rsrc_t rs_sw = make_rsrc(p.skip_w, p.skip_w_bytes);
        const int sk_begin = split * p.skip_kt_per_split, sk_end = min(p.skip_kt, sk_begin + p.skip_kt_per_split);
        constexpr int SAJ = BM / 8 / NW;              // A DMA instructions per wave per tile (8 rows each)
        int s_pix[SAJ];                               // source pixel of this lane's row, or -1
        unsigned s_cb[SAJ];
#pragma unroll
        for (int j = 0; j < SAJ; ++j) {
            const int row = (wave * SAJ + j) * 8 + lrow;
            int pix;
            if constexpr (PW == 16) {
                pix = (pb * p.H + py0 + (row >> 4)) * p.W + px0 + (row & 15);
            } else {
                pix = tile_m * BM + row;
                if (pix >= p.M) pix = -1;
            }
            s_pix[j] = pix;
            s_cb[j] = (unsigned)((lchk ^ ((row >> 1) & 7)) * 16);
        }
        unsigned sb_off[BJ];
        const int skt64 = p.skip_kt;
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int row = (wave * BJ + j) * 8 + lrow;
            const int panel = (n0 >> 6) + (row >> 6);
            sb_off[j] = (unsigned)(((size_t)panel * skt64) * 8192 + ((row & 63) * 8 + lchk) * 16);
        }
        auto stage_skip = [&](int kt, int st) {
            int ci0 = kt * 64;
            const bool second = ci0 >= p.skip_c1;
            const int cs2 = (second ? p.skip_c2 : p.skip_c1) * 2;
            if (second) ci0 -= p.skip_c1;
#pragma unroll
            for (int j = 0; j < SAJ; ++j) {
                const unsigned off = s_pix[j] >= 0 ? (unsigned)(s_pix[j] * cs2) + (unsigned)(ci0 * 2) + s_cb[j] : MDX_OOB;
                void* dst = smem + st * SA_BYTES + (wave * SAJ + j) * 1024;
                if (second)
                    dma16(rs_s2, dst, off);
                else
                    dma16(rs_s1, dst, off);
            }
#pragma unroll
            for (int j = 0; j < BJ; ++j)
                dma16(rs_sw, smem + 2 * HALO_BYTES + st * B_BYTES + (wave * BJ + j) * 1024, sb_off[j] + (unsigned)kt * 8192);
        };
        __syncthreads();            // every wave is done with the halos and the weight ring
        const int sa_row_off = (wm * WROWS + l31) * 128;
        const int s_swz = (l31 >> 1) & 7;
        if (sk_begin < sk_end) stage_skip(sk_begin, 0);
        for (int kt = sk_begin; kt < sk_end; ++kt) {
            const int st = (kt - sk_begin) & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();      // tile kt landed for everyone; stage st ^ 1 (tile kt - 1) is free
            if (kt + 1 < sk_end) stage_skip(kt + 1, st ^ 1);
            const char* sa = smem + st * SA_BYTES;
            const char* sbk = smem + 2 * HALO_BYTES + st * B_BYTES;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                f16x8 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[i] = *reinterpret_cast<const f16x8*>(sa + sa_row_off + i * 32 * 128 + (((2 * s4 + hi) ^ s_swz) << 4));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bf[j] = *reinterpret_cast<const f16x8*>(sbk + (b_row_off - 2 * HALO_BYTES) + j * 32 * 128 + (((2 * s4 + hi) ^ swz_b) << 4));
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if constexpr (SWAP)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
                        else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
                    }
            }
        }
    }

    __syncthreads();
    trace_mark(p, 3);
    if constexpr (PW == 16)
        gemm_epilogue<BM, BN, SWAP, NW, PatchRows, 0, HALO_EMODE>(p, acc, smem, PatchRows{(pb * p.H + py0) * p.W + px0, p.W, pb}, n0, split, bpre,
                                                                  tile_m, tile_id);
    else   // two whole 64-pixel samples: tile rows are consecutive output rows
        gemm_epilogue<BM, BN, SWAP, NW>(p, acc, smem, LinearRows{tile_m * BM}, n0, split, bpre, tile_m, tile_id);
    trace_mark(p, 4);
    }
}

// 8 consecutive fp32 of row m from every split-K slab, summed in slab order (deterministic).  The loads of up to four
// slabs are issued before the first add so that the HBM/L2 latency is paid once per batch, not once per slab.
__device__ __forceinline__ void splitk_sum8(const GemmParams& p, const float* base, const size_t slab, float (&f)[8]) {
    int z = 0;
    for (; z + 4 <= p.nsplit; z += 4) {
        float4 v[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4* s = reinterpret_cast<const float4*>(base + (size_t)(z + u) * slab);
            v[u][0] = s[0];
            v[u][1] = s[1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f[0] += v[u][0].x; f[1] += v[u][0].y; f[2] += v[u][0].z; f[3] += v[u][0].w;
            f[4] += v[u][1].x; f[5] += v[u][1].y; f[6] += v[u][1].z; f[7] += v[u][1].w;
        }
    }
    for (; z < p.nsplit; ++z) {
        const float4* s = reinterpret_cast<const float4*>(base + (size_t)z * slab);
        const float4 s0 = s[0], s1 = s[1];
        f[0] += s0.x; f[1] += s0.y; f[2] += s0.z; f[3] += s0.w; f[4] += s1.x; f[5] += s1.y; f[6] += s1.z; f[7] += s1.w;
    }
}

// split-K reduce + fused epilogue: one thread per (m, 8 output columns)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmParams p) {
    mdx_kernarg_touch<sizeof(GemmParams)>();
    const bool geglu = p.epilogue == MDX_EPI_GEGLU;
    const int ncols = geglu ? p.N / 2 : p.N;
    const int cpr = ncols / 8;
    const size_t total = (size_t)p.M * cpr;
    const size_t slab = (size_t)p.M * p.N;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int m = (int)(idx / cpr);
        const int oc = (int)(idx - (size_t)m * cpr) * 8;
        // time-embedding row / residual first: their latency overlaps the slab loads instead of following them
        Row8Extras xtra;
        if (p.out_mode != MDX_OUT_TRANSPOSED) xtra = epilogue_prefetch_row8(p, m, oc);
        // LayerNorm fold (consumer side): the 8 lanes of a 64-column slice share the row, so each adds every 8th of the
        // producer's partials and an xor tree finishes {sum, sumsq}; N % 64 == 0 keeps the groups whole.
        const bool ln = p.ln_stats != nullptr;
        float mean = 0.f, rstd = 1.f;
        if (ln) {
            const float2* stp = reinterpret_cast<const float2*>(p.ln_stats) + (size_t)m * p.ln_nt;
            float su = 0.f, sq = 0.f;
            for (int j = threadIdx.x & 7; j < p.ln_nt; j += 8) {
                const float2 v = stp[j];
                su += v.x;
                sq += v.y;
            }
#pragma unroll
            for (int off = 4; off >= 1; off >>= 1) {
                su += __shfl_xor(su, off, 64);
                sq += __shfl_xor(sq, off, 64);
            }
            const float inv = 1.0f / (float)p.K;
            mean = su * inv;
            float var = sq * inv - mean * mean;
            var = var < 0.f ? 0.f : var;
            rstd = rsqrtf(var + p.ln_eps);
        }
        float f[8];
        if (geglu) {
            const int pa = (oc >> 6) * 128 + (oc & 63);
            float a[8], g[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                a[e] = (p.bias && !ln) ? p.bias[pa + e] : 0.f;
                g[e] = (p.bias && !ln) ? p.bias[pa + 64 + e] : 0.f;
            }
            splitk_sum8(p, p.ws + (size_t)m * p.N + pa, slab, a);
            splitk_sum8(p, p.ws + (size_t)m * p.N + pa + 64, slab, g);
            if (ln) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    a[e] = rstd * (a[e] - mean * p.ln_s[pa + e]) + (p.bias ? p.bias[pa + e] : 0.f);
                    g[e] = rstd * (g[e] - mean * p.ln_s[pa + 64 + e]) + (p.bias ? p.bias[pa + 64 + e] : 0.f);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = a[e] * gelu_tanh_f(g[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (p.bias && !ln) ? p.bias[oc + e] : 0.f;
            splitk_sum8(p, p.ws + (size_t)m * p.N + oc, slab, f);
            if (ln) {
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = rstd * (f[e] - mean * p.ln_s[oc + e]) + (p.bias ? p.bias[oc + e] : 0.f);
            }
            if (p.epilogue == MDX_EPI_GELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = gelu_tanh_f(f[e]);
            } else if (p.epilogue == MDX_EPI_QUICKGELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = quick_gelu_f(f[e]);
            }
        }
        if (p.out_mode == MDX_OUT_TRANSPOSED) {
            const int b = m / p.HoWo;
            const int tok = m - b * p.HoWo;
#pragma unroll
            for (int e = 0; e < 8; ++e) p.out[((size_t)b * p.N + oc + e) * p.out_ld + tok] = (f16)f[e];
        } else if (p.n_split && oc >= p.n_split) {
            const int b = m / p.HoWo;
            const int tok = m - b * p.HoWo;
            const int nv = p.N - p.n_split;
#pragma unroll
            for (int e = 0; e < 8; ++e) p.out2[((size_t)b * nv + (oc - p.n_split + e)) * p.out2_ld + tok] = (f16)f[e];
        } else {
            const f16x8 o = epilogue_apply_row8(p, f, m, oc, xtra);
            if (p.stats_out) {   // producer side of the LayerNorm fold, as in gemm_epilogue
                float su = 0.f, sq = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float t = (float)o[e];
                    su += t;
                    sq += t * t;
                }
#pragma unroll
                for (int off = 4; off >= 1; off >>= 1) {
                    su += __shfl_xor(su, off, 64);
                    sq += __shfl_xor(sq, off, 64);
                }
                if ((threadIdx.x & 7) == 0)
                    reinterpret_cast<float2*>(p.stats_out)[(size_t)m * (p.N >> 6) + (oc >> 6)] = make_float2(su, sq);
            }
        }
    }
}

// split-K reduce for GroupNorm producers (mdx_gemm_desc.colstats_out): 64-row x 64-column tiles, thread = (row lane,
// 8 columns) with two rows each, so that the per-column {sum, sumsq} of a 64-row block fold inside the block (fixed order).
// Plain row-major epilogue only: bias, per-sample time-embedding row, residual.
constexpr int CS_ROWS = 64;     // rows per colstats row block of a split-K launch (mdx_gemm_query reports it)

__global__ __launch_bounds__(256) void splitk_reduce_colstats_kernel(const GemmParams p) {
    mdx_kernarg_touch<sizeof(GemmParams)>();
    __shared__ float part[32][64][2];
    const int tid = threadIdx.x;
    const int chunk = tid & 7, rl = tid >> 3;
    const int rb = blockIdx.y, n = blockIdx.x * 64 + chunk * 8;
    const size_t slab = (size_t)p.M * p.N;
    float cs[8], cq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[e] = cq[e] = 0.f;
    if (n < p.N) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = rb * CS_ROWS + rl + 32 * h;
            if (m >= p.M) continue;
            const Row8Extras xtra = epilogue_prefetch_row8(p, m, n);
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = p.bias ? p.bias[n + e] : 0.f;
            splitk_sum8(p, p.ws + (size_t)m * p.N + n, slab, f);
            const f16x8 o = epilogue_apply_row8(p, f, m, n, xtra);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = (float)o[e];
                cs[e] += t;
                cq[e] += t * t;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        part[rl][chunk * 8 + e][0] = cs[e];
        part[rl][chunk * 8 + e][1] = cq[e];
    }
    __syncthreads();
    if (tid < 128) {
        const int c = tid >> 1;
        float a = 0.f;
        for (int r = 0; r < 32; ++r) a += part[r][c][tid & 1];
        const int nn = blockIdx.x * 64 + c;
        if (nn < p.N) p.colstats_out[((size_t)rb * p.N + nn) * 2 + (tid & 1)] = a;
    }
}

int fill_params(const mdx_gemm_desc* d, GemmParams& p) {
    MDX_REQUIRE(d && d->a && d->w && d->out, "mdx_gemm_f16: null pointer");
    MDX_REQUIRE(d->ksize == 1 || d->ksize == 3, "mdx_gemm_f16: ksize must be 1 or 3 (got %d)", d->ksize);
    MDX_REQUIRE(d->stride == 1 || d->stride == 2, "mdx_gemm_f16: stride must be 1 or 2");
    MDX_REQUIRE(!(d->upsample && d->stride != 1), "mdx_gemm_f16: upsample requires stride 1");
    MDX_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->N > 0, "mdx_gemm_f16: bad extents");
    MDX_REQUIRE(d->c1 > 0 && d->c1 % 8 == 0 && d->c2 >= 0 && d->c2 % 8 == 0, "mdx_gemm_f16: c1/c2 must be multiples of 8");
    MDX_REQUIRE((d->c2 == 0) == (d->a2 == nullptr), "mdx_gemm_f16: a2/c2 mismatch");
    MDX_REQUIRE(d->N % 8 == 0, "mdx_gemm_f16: N must be a multiple of 8 (got %d)", d->N);
    p.a = (const f16*)d->a;
    p.a2 = (const f16*)d->a2;
    p.w = (const f16*)d->w;
    p.bias = d->bias;
    p.rowbias = d->rowbias;
    p.residual = (const f16*)d->residual;
    p.out = (f16*)d->out;
    p.out2 = (f16*)d->out2;
    p.stats_out = d->stats_out;
    p.colstats_out = d->colstats_out;
    p.ln_stats = d->ln_stats;
    p.ln_s = d->ln_s;
    p.ln_prefetch = mdx_opt(MDX_OPT_GEMM_LN_PREFETCH) ? 1 : 0;
    p.dense_issue = mdx_opt(MDX_OPT_GEMM_DENSE_ISSUE) ? 1 : 0;
    p.ln_nt = d->ln_nt;
    p.ln_eps = d->ln_eps;
    p.bn_hint = d->tile_n;
    p.st_hint = d->stages;
    p.out2_ld = d->out2_ld;
    p.n_split = d->n_split;
    p.ws = (float*)d->workspace;
    p.c1 = d->c1;
    p.c2 = d->c2;
    p.cin = d->c1 + d->c2;
    p.rowbias_ld = d->rowbias_ld;
    p.residual_ld = d->residual_ld;
    p.out_ld = d->out_ld;
    p.out_bs = d->out_bs;
    p.B = d->B;
    p.H = d->H;
    p.W = d->W;
    p.ksize = d->ksize;
    p.stride = d->stride;
    p.upsample = d->upsample ? 1 : 0;
    // asym_pad: zero padding on the bottom / right only (VAE Encoder Downsample: nn.Pad((0,1),(0,1)) + valid 3x3 stride 2,
    // ldm/modules/diffusionmodules/model.py:55-78) -- the taps start AT the output pixel instead of one before it
    MDX_REQUIRE(!d->asym_pad || (d->ksize == 3 && d->stride == 2 && !d->upsample),
                "mdx_gemm_f16: asym_pad applies to the 3x3 stride-2 conv only");
    p.pad = d->ksize == 3 ? (d->asym_pad ? 0 : 1) : 0;
    const int pad_hi = d->ksize == 3 ? 1 : 0;
    const int Hs = p.upsample ? 2 * d->H : d->H, Ws = p.upsample ? 2 * d->W : d->W;
    p.Ho = (Hs + p.pad + pad_hi - d->ksize) / d->stride + 1;
    p.Wo = (Ws + p.pad + pad_hi - d->ksize) / d->stride + 1;
    p.HoWo = p.Ho * p.Wo;
    p.M = d->B * p.HoWo;
    p.N = d->N;
    p.K = d->ksize * d->ksize * p.cin;
    p.epilogue = d->epilogue;
    p.out_mode = d->out_mode;
    MDX_REQUIRE(p.epilogue == MDX_EPI_NONE || p.epilogue == MDX_EPI_GEGLU || p.epilogue == MDX_EPI_GELU ||
                    p.epilogue == MDX_EPI_QUICKGELU, "mdx_gemm_f16: bad epilogue");
    MDX_REQUIRE(p.out_mode == MDX_OUT_ROWMAJOR || p.out_mode == MDX_OUT_TRANSPOSED, "mdx_gemm_f16: bad out_mode");
    if (p.epilogue == MDX_EPI_GEGLU) {
        MDX_REQUIRE(p.N % 128 == 0, "mdx_gemm_f16: GEGLU needs N %% 128 == 0");
        MDX_REQUIRE(p.out_mode == MDX_OUT_ROWMAJOR, "mdx_gemm_f16: GEGLU is row-major only");
    }
    if (p.out_mode == MDX_OUT_TRANSPOSED) {
        // (tokens per sample need not be a multiple of 8: the epilogue then stores element-wise; out_ld is the padded row length)
        MDX_REQUIRE(p.out_ld % 8 == 0 && p.out_ld >= p.HoWo, "mdx_gemm_f16: transposed store needs out_ld %% 8 == 0 and out_ld >= tokens");
        MDX_REQUIRE(!p.rowbias && !p.residual, "mdx_gemm_f16: transposed store takes bias only");
    }
    if (p.n_split) {
        MDX_REQUIRE(p.out2 && p.n_split > 0 && p.n_split < p.N && p.n_split % 128 == 0,
                    "mdx_gemm_f16: n_split must be a multiple of 128 inside (0, N) with out2 set");
        // (out_bs is allowed since round 6: the row-major part may land in a token sub-range of a larger [B][tokens][C] buffer --
        // Taichu-GLIDE's q | k of the image tokens behind the text keys, unet.py:289-297; both store paths go through
        // epilogue_apply_row8, which knows it)
        MDX_REQUIRE(p.out_mode == MDX_OUT_ROWMAJOR && p.epilogue == MDX_EPI_NONE && !p.rowbias && !p.residual,
                    "mdx_gemm_f16: the split row-major | transposed output takes bias only");
        MDX_REQUIRE(p.out2_ld % 8 == 0 && p.out2_ld >= p.HoWo,
                    "mdx_gemm_f16: transposed part needs out2_ld %% 8 == 0 and out2_ld >= tokens");
    }
    if (p.ln_stats) {
        MDX_REQUIRE(p.ln_s && p.ln_nt * 64 == p.K && p.ksize == 1 && p.c2 == 0 && p.stride == 1 && !p.upsample &&
                        p.out_mode == MDX_OUT_ROWMAJOR && p.N % (p.epilogue == MDX_EPI_GEGLU ? 128 : 64) == 0,
                    "mdx_gemm_f16: LayerNorm fold needs ln_s, ln_nt == K / 64, N %% 64 == 0 and a dense row-major GEMM");
    }
    if (p.stats_out)
        MDX_REQUIRE(p.out_mode == MDX_OUT_ROWMAJOR && p.epilogue != MDX_EPI_GEGLU && !p.n_split && p.N % 64 == 0,
                    "mdx_gemm_f16: row statistics are produced by plain row-major stores with N %% 64 == 0 only");
    if (p.colstats_out)
        MDX_REQUIRE(p.out_mode == MDX_OUT_ROWMAJOR && p.epilogue == MDX_EPI_NONE && !p.n_split && !p.ln_stats && !p.stats_out &&
                        !p.out_bs && p.N % 8 == 0,
                    "mdx_gemm_f16: column statistics come from plain row-major launches only");
    p.gn_cs = d->gn_colstats;
    p.gn_gamma = d->gn_gamma;
    p.gn_beta = d->gn_beta;
    p.gn_nrb = d->gn_nrb;
    p.gn_silu = d->gn_silu ? 1 : 0;
    p.gn_eps = d->gn_eps;
    if (p.gn_cs) {
        MDX_REQUIRE(p.gn_gamma && p.gn_beta && p.gn_nrb > 0, "mdx_gemm_f16: gn_colstats needs gn_gamma, gn_beta and gn_nrb > 0");
        if (p.ksize == 1) {     // GroupNorm (no activation) -> Dense / 1x1 conv: applied to the A fragments (gemm_kernel GNA)
            MDX_REQUIRE(p.stride == 1 && !p.upsample && p.c2 == 0 && p.cin % 64 == 0 && p.cin <= 2560 && !p.gn_silu &&
                            p.HoWo % 64 == 0,
                        "mdx_gemm_f16: the fused input GroupNorm of a dense launch needs a single source, Cin %% 64 == 0, "
                        "Cin <= 2560, gn_silu = 0 and tokens per sample %% 64 == 0");
        } else {
            MDX_REQUIRE(p.ksize == 3 && p.stride == 1 && !p.upsample && p.c2 == 0 && p.cin % 64 == 0 && p.cin % 32 == 0 &&
                            p.cin <= 640 && !d->w_frag && !(d->H == 8 && d->W == 8),
                        "mdx_gemm_f16: the fused input GroupNorm rides on a single-source 3x3 stride-1 conv with Cin %% 64 == 0, "
                        "Cin <= 640, images larger than 8 x 8 and tile-major weights");
        }
    }
    p.xa_k = (const f16*)d->xattn_k;
    p.xa_vt = (const f16*)d->xattn_vt;
    p.xa_len = d->xattn_len;
    p.xa_cap = d->xattn_cap;
    p.xa_scale_log2 = d->xattn_scale * 1.4426950408889634f;
    if (p.xa_k) {
        const int howo = d->H * d->W;
        MDX_REQUIRE(p.xa_vt && d->xattn_len > 0 && d->xattn_len <= 128 && d->xattn_cap >= d->xattn_len && d->xattn_cap % 8 == 0,
                    "mdx_gemm_f16: cross-attention epilogue needs xattn_vt, 0 < xattn_len <= 128 <= ... xattn_cap (a multiple of 8)");
        MDX_REQUIRE(d->ksize == 1 && d->stride == 1 && !d->upsample && d->c2 == 0 && d->N % 64 == 0 && d->c1 % 64 == 0 &&
                        d->epilogue == MDX_EPI_NONE && d->out_mode == MDX_OUT_ROWMAJOR && !d->residual && !d->rowbias && !d->stats_out &&
                        !d->colstats_out && !d->n_split && !d->out_bs && !d->gn_colstats && !d->defer_reduce,
                    "mdx_gemm_f16: the cross-attention epilogue rides on a plain dense row-major projection (N %% 64 == 0, Cin %% 64 == 0)");
        MDX_REQUIRE(d->tile_n == 64 && d->splitk == 1 && (howo % 128 == 0 || (howo % 64 == 0 && d->tile_m == 64)) &&
                        (d->tile_m == 0 || d->tile_m == 64 || d->tile_m == 128),
                    "mdx_gemm_f16: the cross-attention epilogue needs tile_n = 64 (one head per tile), splitk = 1 and an M tile inside one sample "
                    "(tokens per sample %% 128 == 0, or %% 64 == 0 with tile_m = 64)");
        MDX_REQUIRE((size_t)d->xattn_cap * d->N * 2 <= 0x80000000ull, "mdx_gemm_f16: cross-attention context larger than 2 GiB per sample");
    }
    p.w_sub = (const f16*)d->w_sub;
    p.w_sub_bytes = 0;
    p.c8_sub = 0;
    if (p.w_sub) {
        MDX_REQUIRE(d->upsample && d->ksize == 3 && d->stride == 1, "mdx_gemm_f16: w_sub belongs to the nearest-2x + 3x3 conv (upsample = 1)");
        const size_t sb = (size_t)((4 * (size_t)d->N + 63) / 64) * ((4 * (size_t)(d->c1 + d->c2) + 63) / 64) * 8192;
        MDX_REQUIRE(sb <= 0x80000000ull, "mdx_gemm_f16: sub-pixel weights larger than 2 GiB");
        p.w_sub_bytes = (unsigned)sb;
    }
    p.skip_a = (const f16*)d->skip_a;
    p.skip_a2 = (const f16*)d->skip_a2;
    p.skip_w = (const f16*)d->skip_w;
    p.skip_c1 = d->skip_c1;
    p.skip_c2 = d->skip_c2;
    if (p.skip_w) {
        MDX_REQUIRE(p.skip_a && d->skip_c1 > 0 && d->skip_c1 % 64 == 0 && d->skip_c2 >= 0 && d->skip_c2 % 64 == 0 &&
                        (d->skip_c2 == 0) == (d->skip_a2 == nullptr),
                    "mdx_gemm_f16: fused skip needs skip_a, skip_c1 %% 64 == 0, skip_c2 %% 64 == 0 and skip_a2 iff skip_c2");
        MDX_REQUIRE(p.ksize == 3 && p.stride == 1 && !p.upsample && p.c2 == 0 && p.out_mode == MDX_OUT_ROWMAJOR && !d->w_frag,
                    "mdx_gemm_f16: the fused skip rides on a single-source 3x3 stride-1 row-major conv (tile-major weights)");
        p.skip_kt = (d->skip_c1 + d->skip_c2) / 64;
        const size_t s1 = (size_t)d->B * d->H * d->W * d->skip_c1 * 2, s2 = (size_t)d->B * d->H * d->W * d->skip_c2 * 2;
        const size_t sw = (size_t)((p.N + 63) / 64) * p.skip_kt * 8192;
        MDX_REQUIRE(s1 <= 0x80000000ull && s2 <= 0x80000000ull && sw <= 0x80000000ull, "mdx_gemm_f16: skip operand larger than 2 GiB");
        p.skip_a_bytes = (unsigned)s1;
        p.skip_a2_bytes = (unsigned)s2;
        p.skip_w_bytes = (unsigned)sw;
    }
    if (p.rowbias) MDX_REQUIRE(p.rowbias_ld % 4 == 0, "mdx_gemm_f16: rowbias_ld must be a multiple of 4");
    if (p.residual) MDX_REQUIRE(p.residual_ld % 8 == 0, "mdx_gemm_f16: residual_ld must be a multiple of 8");
    MDX_REQUIRE(p.out_ld % 8 == 0 && p.out_bs % 8 == 0 && p.out_bs >= 0, "mdx_gemm_f16: out_ld / out_bs must be multiples of 8");
    MDX_REQUIRE(!(p.out_bs && p.out_mode == MDX_OUT_TRANSPOSED), "mdx_gemm_f16: out_bs applies to row-major output only");
    const size_t ab = (size_t)d->B * d->H * d->W * d->c1 * 2, a2b = (size_t)d->B * d->H * d->W * d->c2 * 2;
    p.kt64 = (p.K + 63) / 64;
    const size_t wb = (size_t)((p.N + 63) / 64) * p.kt64 * 8192;   // padded, tile-major storage
    MDX_REQUIRE(ab <= 0x80000000ull && a2b <= 0x80000000ull && wb <= 0x80000000ull,
                "mdx_gemm_f16: operand larger than 2 GiB is not addressable by one buffer descriptor");
    p.a_bytes = (unsigned)ab;
    p.a2_bytes = (unsigned)a2b;
    p.w_bytes = (unsigned)wb;
    return MDX_OK;
}

// Measured (tile_m, tile_n, splitk) per UNet shape: tools/tune_gemm.py times every candidate on the device with cold
// weights and writes gemm_tuned.inc.  Shapes that are not in the table fall through to pick_bn / the cost model.
// A row is keyed by shape AND launch variant: the same (M, N, K) occurs with different epilogues / operand forms in one UNet
// (proj_in, attention out + residual + row statistics, LayerNorm-fold consumers ...), and a split or tile that was measured
// for one of them says nothing about the others.  var1 = variant + 1; 0 (rows written before the key existed) = any variant,
// consulted only when no exact row matches.
struct TunedEntry {
    int M, N, K, ksize, bm, bn, ns;   // bn 0 = pick_bn's default
    int var1;
    int st;                           // LDS ring depth (0 = the occupancy rule in mdx_gemm_f16)
};
static const TunedEntry g_tuned[] = {
#include "gemm_tuned.inc"
    {0, 0, 0, 0, 0, 0, 0, 0, 0}};

// Launch variant of a descriptor (tools/tune_gemm.py computes the same number from the mdx_gemm_desc fields).
int tuned_variant(const GemmParams& p) {
    return (p.c2 > 0 ? 1 : 0) | (p.epilogue << 1) | (p.n_split ? 8 : 0) | (p.ln_stats ? 16 : 0) | (p.stats_out ? 32 : 0) |
           (p.out_mode == MDX_OUT_TRANSPOSED ? 64 : 0) | (p.colstats_out ? 128 : 0) | (p.residual ? 256 : 0) |
           (p.rowbias ? 512 : 0);
}

bool halo_eligible(const GemmParams& p, int bm);

const TunedEntry* lookup_tuned(const GemmParams& p) {
    const bool use_table = mdx_opt(MDX_OPT_GEMM_TUNED) && !mdx_opt(MDX_OPT_GEMM_BM) && !mdx_opt(MDX_OPT_GEMM_BN);
    if (!use_table || p.bn_hint || p.st_hint || p.stride != 1 || p.upsample) return nullptr;
    const int var1 = tuned_variant(p) + 1;
    const TunedEntry* any = nullptr;
    for (const TunedEntry* e = g_tuned; e->M; ++e)
        if (e->M == p.M && e->N == p.N && e->K == p.K && e->ksize == p.ksize) {
            if (e->var1 == var1) {
                any = e;
                break;
            }
            if (e->var1 == 0 && !any) any = e;
        }
    if (!any) return nullptr;
    // The key is (M, N, K, ksize), not the image geometry: a row measured at one (B, H, W) also matches other factorizations of M.
    // A 256-row entry means the 16 x 16-patch HALO kernel; where that kernel does not apply (8 x 8 images at UNet batch 8 share
    // M = 512 with the 16 x 16 level at batch 2) the row does not describe this launch -- round 5: it used to be taken, the generic
    // kernel then ran its 128-row tiles on a grid sized for 256-row ones and left the second half of every tile pair unwritten
    if (any->bm == 256 && !halo_eligible(p, 256)) return nullptr;
    return (any->bm >= 128 || !halo_eligible(p, 128)) ? any : nullptr;
}

int pick_bn(const GemmParams& p) {
    if (p.epilogue == MDX_EPI_GEGLU) return 128;
    if (p.bn_hint == 64 || p.bn_hint == 128) return p.bn_hint;
    if (const TunedEntry* e = lookup_tuned(p))
        if (e->bn) return e->bn;
    const int optbn = mdx_opt(MDX_OPT_GEMM_BN);
    if (optbn == 64 || optbn == 128) return optbn;
    if (p.N % 128 == 0) return 128;
    if (p.N % 64 == 0 || p.N < 128) return 64;
    return (p.N % 128 > 64) ? 128 : 64;
}

struct GemmCfg {
    int bm, bn, bk, ns;
};

// Tile configuration.  Experiments: mdx_set_option("gemm_ring", 2..5) forces the LDS ring depth.
GemmCfg pick_cfg(const GemmParams& p) {
    GemmCfg c;
    c.bm = 128;   // finalised by choose_tiling
    c.bn = pick_bn(p);
    c.bk = 64;
    c.ns = 2;   // ring depth is finalised in mdx_gemm_f16 once the grid size is known
    const int ring = mdx_opt(MDX_OPT_GEMM_RING);
    if (ring >= 2 && ring <= 5) c.ns = ring;
    return c;
}

bool halo_eligible(const GemmParams& p, int bm);

// Tile height and split-K factor from a cost model fitted to the B=2 micro-benchmarks
// (profiles/r01_gemm_auto_tiling.txt, tools/gemm_trace.py), in microseconds:
//   main loop   = K tiles per split x tau(kernel, tile) x occupancy    tau = 0.65 us for the generic 128x128 tile:
//                 a K-tile step of ONE block is bound by its own DMA-issue + MFMA + barrier chain, so small grids
//                 finish sooner with more, smaller blocks -- until every CU holds one (occupancy = 1 up to 256
//                 blocks, 1.15 x rounds of 512 beyond); never less than streaming the cold operands once from HBM;
//   split-K     = 3.5 (the extra reduce launch) + 0.3 x splits x slab MB (slab write + re-read);
// fixed per-launch costs are the same for every candidate and drop out.  Splits keep >= 4 K tiles (HALO convs: whole
// 64-channel chunks).  The 64-row tile is only a candidate for the generic kernel, the 256-row tile only for HALO.
struct Tiling {
    int bm, ns;
};

Tiling choose_tiling(const GemmParams& p, int bn, int forced_ns, int forced_bm) {
    const int envbm = mdx_opt(MDX_OPT_GEMM_BM);
    if (forced_ns <= 0 && forced_bm <= 0)
        if (const TunedEntry* e = lookup_tuned(p)) return Tiling{e->bm, e->ns};
    const int kt = (p.K + 63) / 64;
    const int chunks = p.cin / 64;
    const double slab_mb = (double)p.M * p.N * 4.0 / 1048576.0;
    const double unique_mb = ((double)p.N * p.K + (double)p.M * p.cin) * 2.0 / 1048576.0;
    Tiling best{128, 1};
    double best_cost = 1e30;
    static const int order[3] = {128, 256, 64};   // increasing launch complexity (see the hysteresis below)
    for (int oi = 0; oi < 3; ++oi) {
        const int bm = order[oi];
        if (envbm && envbm != bm) continue;
        if (forced_bm > 0 && forced_bm != bm) continue;
        const bool halo = bm >= 128 && halo_eligible(p, bm);
        // 256-row tiles exist for the HALO kernel only and are opt-in (MDX_GEMM_BM=256): measured on MI355X they move
        // 45 % fewer DMA bytes per MAC yet run no faster than two co-resident 128-row blocks (profiles/
        // r01_halo256_ab.txt) -- the K-step's barrier/issue structure, not the DMA rate, is what bounds this kernel
        if (bm == 256 && (!halo || !(envbm || forced_bm == 256))) continue;
        if (bm == 64 && halo_eligible(p, 128) && !envbm && forced_bm <= 0) continue;  // HALO beats the generic kernel on every conv
        const int tiles = ((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn);
        // us per K-tile step of one block running alone on its CU (tools/gemm_trace.py): fewer DMA instructions and
        // MFMAs per step for the smaller tiles and for the HALO kernel (one activation DMA per 9 taps)
        const double tau = (halo ? (bm == 256 ? 0.65 : 0.60) : 0.65) * (bm == 64 ? 0.7 : 1.0) * (bn == 64 ? 0.7 : 1.0);
        const int max_ns = forced_ns > 0 ? forced_ns : 16;
        for (int ns = forced_ns > 0 ? forced_ns : 1; ns <= max_ns; ++ns) {
            int kps, eff;
            if (halo) {
                if (ns > chunks && forced_ns <= 0) break;
                const int cps = (chunks + std::min(ns, chunks) - 1) / std::min(ns, chunks);
                kps = cps * 9;
                eff = (chunks + cps - 1) / cps;
            } else {
                if (forced_ns <= 0 && ns > 1 && kt / ns < 4) break;
                kps = (kt + std::min(ns, kt) - 1) / std::min(ns, kt);
                eff = (kt + kps - 1) / kps;
            }
            if (forced_ns <= 0 && eff != ns) continue;   // same launch as a smaller ns
            // up to 256 blocks run one per CU.  128/64-row tiles: beyond that two share a CU (their stalls overlap:
            // only ~1.15x slower each) and the grid runs in rounds of 512; 256-row tiles own a CU: rounds of 256.
            // The last, partly filled round costs as much as a full one.
            const int blocks = tiles * eff;
            const double occ = blocks <= 256 ? 1.0 : (bm == 256 ? (blocks + 255) / 256 : 1.15 * ((blocks + 511) / 512));
            const double main_us = std::max(kps * tau * occ, unique_mb / 3.5);   // cold operands stream at ~3.5 TB/s
            const double cost = main_us + (eff > 1 ? 3.5 + 0.3 * eff * slab_mb : 0.0);
            // the model is only good to ~10-20 %, so a more complex candidate must promise a clear win
            if (cost < 0.9 * best_cost) {
                best_cost = cost;
                best = Tiling{bm, ns};
            }
        }
    }
    return best;
}

template <int BM, int BN, int BK, int NS, bool SWAP, bool FASTK, int NW>
void launch_one(const GemmParams& p, dim3 grid, hipStream_t st) {
    constexpr size_t ring = (size_t)NS * (BM + BN) * BK * 2;
    constexpr size_t epi = (size_t)(BM > BN ? BM : BN) * ((BM > BN ? BN : BM) + 8) * 2 + 4096;  // staged C tile
    constexpr size_t lds = ring > epi ? ring : epi;
    static MdxPerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<BM, BN, BK, NS, SWAP, FASTK, NW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, NS, SWAP, FASTK, NW>), grid, dim3(NW * 64), lds, st, p);
}

// GNA form (GroupNorm of the input on the A fragments): four waves, ring depth 2 | 3, K-contiguous (dense) launches only
template <int BM, int BN, int NS, bool SWAP>
void launch_gna_one(const GemmParams& p, dim3 grid, hipStream_t st) {
    constexpr size_t ring = (size_t)NS * (BM + BN) * 64 * 2;
    constexpr size_t epi = (size_t)(BM > BN ? BM : BN) * ((BM > BN ? BN : BM) + 8) * 2 + 4096;
    constexpr size_t lds0 = ring > epi ? ring : epi;
    const size_t lds = lds0 + (size_t)p.cin * 12;       // a, s (fp16) + {sum, sumsq} scratch (fp32 pairs)
    static MdxPerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<BM, BN, 64, NS, SWAP, true, 4, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds0 + 2560 * 12));
    }
    hipLaunchKernelGGL((gemm_kernel<BM, BN, 64, NS, SWAP, true, 4, true>), grid, dim3(256), lds, st, p);
}

template <int BM, int BN>
void launch_gna(const GemmParams& p, int ns, bool swap, dim3 grid, hipStream_t st) {
    if (ns >= 3) {
        if (swap) launch_gna_one<BM, BN, 3, true>(p, grid, st); else launch_gna_one<BM, BN, 3, false>(p, grid, st);
    } else {
        if (swap) launch_gna_one<BM, BN, 2, true>(p, grid, st); else launch_gna_one<BM, BN, 2, false>(p, grid, st);
    }
}

template <int BM, int BN, int BK, int NS, int NW>
void launch_cfg(const GemmParams& p, bool swap, bool fastk, dim3 grid, hipStream_t st) {
    if (swap) {
        if (fastk)
            launch_one<BM, BN, BK, NS, true, true, NW>(p, grid, st);
        else
            launch_one<BM, BN, BK, NS, true, false, NW>(p, grid, st);
    } else {
        if (fastk)
            launch_one<BM, BN, BK, NS, false, true, NW>(p, grid, st);
        else
            launch_one<BM, BN, BK, NS, false, false, NW>(p, grid, st);
    }
}

template <int BM, int BN>
bool launch_bn(const GemmCfg& c, const GemmParams& p, bool swap, bool fastk, dim3 grid, hipStream_t st, bool nw8 = false) {
    if constexpr (BM == 128) {
        // eight waves (4 x 2) on the same 128-row tile: each wave owns half the rows of the 4-wave form, so a block that has a
        // CU to itself keeps two waves per SIMD busy (tile-table choice, mdx_gemm_desc.stages 10 | 11)
        if (nw8 && c.bk == 64 && (c.ns == 2 || c.ns == 3)) {
            if (c.ns == 2) launch_cfg<128, BN, 64, 2, 8>(p, swap, fastk, grid, st); else launch_cfg<128, BN, 64, 3, 8>(p, swap, fastk, grid, st);
            return true;
        }
    }
    if (c.bk == 64 && c.ns == 2) launch_cfg<BM, BN, 64, 2, 4>(p, swap, fastk, grid, st);
    else if (c.bk == 64 && c.ns == 3) launch_cfg<BM, BN, 64, 3, 4>(p, swap, fastk, grid, st);
    else if (BM == 128 && c.bk == 64 && c.ns == 4) launch_cfg<128, BN, 64, 4, 4>(p, swap, fastk, grid, st);
    else if (BM == 128 && c.bk == 64 && c.ns == 5) launch_cfg<128, BN, 64, 5, 4>(p, swap, fastk, grid, st);
    // 64-row tiles stage only 16 / 24 KB per K tile: deeper rings are cheap, and a launch with one or two blocks per CU is
    // bound by how many tiles it keeps in flight (tile-table choice)
    else if (BM == 64 && c.bk == 64 && c.ns == 4) launch_cfg<64, BN, 64, 4, 4>(p, swap, fastk, grid, st);
    else if (BM == 64 && c.bk == 64 && c.ns == 5) launch_cfg<64, BN, 64, 5, 4>(p, swap, fastk, grid, st);
    else if (BM == 64 && c.bk == 64 && c.ns == 6) launch_cfg<64, BN, 64, 6, 4>(p, swap, fastk, grid, st);
    else return false;
    return true;
}

template <int BM, int BN, int NSB, bool SWAP, int PW = 16, bool BDIR = false, bool W4 = false>
void launch_halo(const GemmParams& p, dim3 grid, hipStream_t st) {
    constexpr size_t hinst = ((PW == 16 ? (BM / 16 + 2) * 18 : (BM / 64) * 100) + 7) / 8;
    constexpr size_t ring = 2 * hinst * 1024 + (BDIR ? 0 : (size_t)NSB * BN * 128);
    constexpr size_t epi = (size_t)BM * (BN + 8) * 2 + 4096;
    constexpr size_t lds0 = ring > epi ? ring : epi;
    constexpr size_t gn_tab_max = (BDIR || PW != 16) ? 0 : 640 * 8;      // {scale, shift} table of the fused input GroupNorm
    const size_t lds = lds0 + (p.gn_cs ? (size_t)p.cin * 8 : 0);
    static MdxPerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<BM, BN, NSB, SWAP, PW, BDIR, W4>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds0 + gn_tab_max));
    }
    hipLaunchKernelGGL((conv3x3_halo_kernel<BM, BN, NSB, SWAP, PW, BDIR, W4>), grid, dim3(BM * 2), lds, st, p);
}

// weight-streaming form (fragment-major weights): 128-row tiles only, no weight ring in LDS
template <int BN, int PW>
void launch_halo_bdir(const GemmParams& p, bool swap, dim3 grid, hipStream_t st) {
    if (swap) launch_halo<128, BN, 2, true, PW, true>(p, grid, st); else launch_halo<128, BN, 2, false, PW, true>(p, grid, st);
}
template <int PW>
void launch_halo_bdir_w4(const GemmParams& p, dim3 grid, hipStream_t st) {
    launch_halo<128, 128, 2, false, PW, true, true>(p, grid, st);
}

template <int BM, int BN, int PW = 16>
void launch_halo_cfg(const GemmParams& p, int nsb, bool swap, dim3 grid, hipStream_t st) {
    if (nsb == 4) {      // (round 5) the weight stream of a lone block is latency x tiles-in-flight bound: a fourth stage
        if (swap) launch_halo<BM, BN, 4, true, PW>(p, grid, st); else launch_halo<BM, BN, 4, false, PW>(p, grid, st);
    } else if (nsb == 3) {
        if (swap) launch_halo<BM, BN, 3, true, PW>(p, grid, st); else launch_halo<BM, BN, 3, false, PW>(p, grid, st);
    } else {
        if (swap) launch_halo<BM, BN, 2, true, PW>(p, grid, st); else launch_halo<BM, BN, 2, false, PW>(p, grid, st);
    }
}

// 8 x 8-pixel images (the deepest UNet level at a 64 x 64 latent): two whole samples per 128-row tile (PW = 8).
bool halo8_eligible(const GemmParams& p) {
    if (!mdx_opt(MDX_OPT_GEMM_HALO8)) return false;
    return p.H == 8 && p.W == 8;
}

// The HALO kernel applies to 3x3 / stride 1 / single-source convs whose image tiles into 8 x 16 (16 x 16) patches, or
// whose images are 8 x 8 (bm = 128 only).
bool halo_eligible(const GemmParams& p, int bm) {
    if (!mdx_opt(MDX_OPT_GEMM_HALO)) return false;
    if (!(p.ksize == 3 && p.stride == 1 && !p.upsample && p.c2 == 0 && p.cin % 64 == 0 && p.out_mode == MDX_OUT_ROWMAJOR))
        return false;
    if (p.out_bs) return false;      // (no conv writes a strided-sample output; the patch epilogue does not carry the form)
    if (p.residual && (size_t)p.M * (size_t)p.residual_ld * 2 >= 0x80000000ull) return false;      // (its residual rows go through a 32-bit-offset descriptor)
    if (bm == 128 && halo8_eligible(p)) return true;
    return p.H % (bm / 16) == 0 && p.W % 16 == 0;
}

}  // namespace

static unsigned long long* g_gemm_trace = nullptr;
static size_t g_gemm_trace_slots = 0;

// Diagnostics: while a buffer is registered, every block of the next mdx_gemm_f16 launches writes 8 x u64 phase
// timestamps (s_memrealtime, 100 MHz): 0 start, 1 prologue issued, 2 first tile landed, 3 main loop done, 4 epilogue
// done.  NULL unregisters.  Not thread safe; not for production use.
extern "C" int mdx_probe_gemm_trace(void* buf, size_t bytes) {
    g_gemm_trace = (unsigned long long*)buf;
    g_gemm_trace_slots = buf ? bytes / 64 : 0;
    return MDX_OK;
}

// Split-K launches of at most `gemm_splitk_fixup_max` (mdx_set_option; default 4) splits reduce in the kernel (splitk_last_block_reduce)
// when the output is row-major and the tiles fit the ticket area.  The last arriver reads nsplit partials serially at the
// ~65 GB/s one block can pull, so the in-kernel form only beats the reduce launch it replaces for few splits (measured at UNet
// batch 2, profiles/r02_l_splitk_fixup.txt: 4 splits -1.7 us, 3 splits -1.4 us, 5 splits 0 ... +1.7 us, 10 splits +7 us,
// 20 splits +8 us per launch); deeper splits, transposed outputs and deferred reduces keep the [split][M][N] slabs + reduce kernel.
static bool conv8p_wanted(const mdx_gemm_desc* d, const GemmParams& p);

static bool fixup_eligible(const mdx_gemm_desc* d, const GemmParams& p, int bm, int bn, int ns) {
    const int max_ns = mdx_opt(MDX_OPT_GEMM_SPLITK_FIXUP_MAX);
    if (ns > max_ns) return false;
    if (p.out_mode != MDX_OUT_ROWMAJOR || d->defer_reduce) return false;
    const long tiles = (long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn);
    return tiles <= MDX_TICKET_SLOTS;
}

// bytes of workspace one split of this launch occupies (+ `head` bytes once)
static size_t split_bytes(const mdx_gemm_desc* d, const GemmParams& p, int bm, int bn, int ns, size_t* head) {
    if (fixup_eligible(d, p, bm, bn, ns)) {
        *head = (size_t)MDX_TICKET_SLOTS * sizeof(unsigned);
        return (size_t)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * bm * bn * sizeof(float);
    }
    *head = (size_t)MDX_TICKET_SLOTS * sizeof(unsigned);     // never used for slabs: other launches keep their tickets there
    return (size_t)p.M * p.N * sizeof(float);
}

extern "C" size_t mdx_gemm_workspace_bytes(const mdx_gemm_desc* d) {
    GemmParams p{};
    if (fill_params(d, p) != MDX_OK) return 0;
    if (conv8p_wanted(d, p)) return mdx_conv8p_plan(p, mdx_conv8p_pick_bn(p, d->tile_n), 0, false, true);
    const GemmCfg c = pick_cfg(p);
    const Tiling tl = choose_tiling(p, c.bn, d->splitk, d->tile_m);
    if (tl.ns <= 1) return 0;
    size_t head;
    // sized for the larger of the two layouts (tile-padded partials of the in-kernel form >= [M][N] slabs): a launch that a
    // small workspace clamps to fewer splits may switch form
    const size_t per = std::max(split_bytes(d, p, tl.bm, c.bn, 2, &head), split_bytes(d, p, tl.bm, c.bn, 1 << 30, &head));
    return head + (size_t)tl.ns * per;
}

static int mdx_internal_resolve_check(const mdx_gemm_desc* d, GemmParams& p);

extern "C" int mdx_gemm_check(const mdx_gemm_desc* d) {
    GemmParams p{};
    int rc = fill_params(d, p);
    if (rc != MDX_OK) return rc;
    const bool fastk = (p.cin % 64 == 0) && (p.c2 == 0 || p.c1 % 64 == 0);
    MDX_REQUIRE(fastk || p.c2 == 0, "mdx_gemm_f16: two-source input needs c1 %% 64 == 0 and Cin %% 64 == 0");
    // ... and the launch form the descriptor resolves to must exist for it (round 6): a forced tile / a tile-table row whose kernel
    // does not apply to this geometry is an error HERE, at plan time, not a launch that silently takes another form
    return mdx_internal_resolve_check(d, p);
}

// Everything mdx_gemm_f16 decides before it launches: tile shape, split-K factor (clamped to the caller's workspace), kernel.
struct Resolved {
    GemmCfg c;
    int bn, ns;
    int stages;      // LDS ring depth forced by the descriptor or the tile table (0 = the occupancy rule)
    bool halo, tuned;
    bool fixup;      // split-K reduced by the last block of each tile (no reduce launch)
    bool c8;         // the 256-pixel eight-wave conv core (conv8p.hip); implies halo, tile_m 256, no split
};

// Arrival counters of the in-kernel split-K reduce.  They used to sit in the first MDX_GEMM_WS_HEAD bytes of the caller's
// workspace, which made "hand the workspace over zeroed" part of the contract: a fresh hipMalloc'd buffer trapped, and zeroing on
// the first sight of an ADDRESS broke as soon as an allocator handed the same address out twice (round 3: the stress test after
// another test's workspace).  Now the library owns them: MDX_TICKET_SLOTS counters per (device, workspace address), carved from
// 1 MiB chunks that are zeroed once when they are allocated -- the ONE exception to "the library never allocates device memory"
// (include/mdx.h).  Every launch leaves its counters zero, so a slot is valid for whatever buffer an address names later; launches
// that may run concurrently have distinct workspaces (their partials) and therefore distinct counters.  The chunk is allocated with
// the thread's stream-capture mode relaxed and zeroed on a private stream, so a first use inside a capture works too.
struct TicketPools {
    std::mutex mu;
    std::map<std::pair<int, const void*>, unsigned*> slot;
    std::map<std::pair<int, const void*>, bool> caller_owned;      // slots bound by mdx_gemm_bind_counters: never recycled / freed here
    struct Dev { char* next = nullptr; int left = 0; hipStream_t zero_stream = nullptr; std::vector<unsigned*> free_slots; };
    std::map<int, Dev> dev;
    std::vector<std::pair<int, void*>> chunks;
};
static TicketPools g_tickets;
constexpr int TICKET_CHUNK_SLOTS = 64;

// The device a workspace lives on comes from the POINTER (hipPointerGetAttributes), not from the calling thread's current device: a
// C caller that drives several GPUs from one thread gets the right pool either way.
static int ticket_device_of(const void* ws) {
    hipPointerAttribute_t at;
    if (ws && hipPointerGetAttributes(&at, ws) == hipSuccess) return at.device;
    (void)hipGetLastError();
    int dv = 0;
    (void)hipGetDevice(&dv);
    return dv;
}

static int ticket_slot(const void* ws, unsigned** out) {
    const int dv = ticket_device_of(ws);
    std::lock_guard<std::mutex> lk(g_tickets.mu);
    const auto key = std::make_pair(dv, ws);
    const auto it = g_tickets.slot.find(key);
    if (it != g_tickets.slot.end()) {
        *out = it->second;
        return MDX_OK;
    }
    TicketPools::Dev& d = g_tickets.dev[dv];
    if (!d.free_slots.empty()) {        // a slot handed back by mdx_gemm_release_workspace: every launch left its counters zero
        unsigned* s = d.free_slots.back();
        d.free_slots.pop_back();
        g_tickets.slot.emplace(key, s);
        *out = s;
        return MDX_OK;
    }
    if (d.left == 0) {
        const size_t bytes = (size_t)TICKET_CHUNK_SLOTS * MDX_GEMM_WS_HEAD;
        int cur = 0;
        (void)hipGetDevice(&cur);
        if (cur != dv) (void)hipSetDevice(dv);
        hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
        (void)hipThreadExchangeStreamCaptureMode(&mode);
        void* mem = nullptr;
        hipError_t e = hipMalloc(&mem, bytes);
        if (e == hipSuccess && !d.zero_stream) e = hipStreamCreateWithFlags(&d.zero_stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMemsetAsync(mem, 0, bytes, d.zero_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(d.zero_stream);
        (void)hipThreadExchangeStreamCaptureMode(&mode);
        if (cur != dv) (void)hipSetDevice(cur);
        if (e != hipSuccess) {
            if (mem) (void)hipFree(mem);
            mdx_set_error("mdx_gemm_f16: allocating the split-K arrival counters failed: %s", hipGetErrorString(e));
            return MDX_E_HIP;
        }
        g_tickets.chunks.emplace_back(dv, mem);
        d.next = static_cast<char*>(mem);
        d.left = TICKET_CHUNK_SLOTS;
    }
    unsigned* s = reinterpret_cast<unsigned*>(d.next);
    d.next += MDX_GEMM_WS_HEAD;
    d.left -= 1;
    g_tickets.slot.emplace(key, s);
    *out = s;
    return MDX_OK;
}

// Caller-owned arrival counters (include/mdx.h): `counters` = MDX_GEMM_WS_HEAD bytes of ZEROED device memory on the workspace's
// device, bound to the workspace ADDRESS until mdx_gemm_release_workspace(workspace).  Launches on a bound workspace take their
// tickets there and the library makes no device allocation for them: a host that owns every byte (a graph-capturing caller on its
// own allocator) binds one set per workspace and the "one exception" of the ownership rule never fires.
extern "C" int mdx_gemm_bind_counters(const void* workspace, void* counters) {
    MDX_REQUIRE(workspace && counters && ((uintptr_t)counters % 16) == 0,
                "mdx_gemm_bind_counters: workspace and a 16-byte aligned counters buffer of MDX_GEMM_WS_HEAD zeroed bytes are required");
    const int dv = ticket_device_of(workspace);
    MDX_REQUIRE(ticket_device_of(counters) == dv, "mdx_gemm_bind_counters: the counters must live on the workspace's device");
    std::lock_guard<std::mutex> lk(g_tickets.mu);
    const auto key = std::make_pair(dv, workspace);
    const auto it = g_tickets.slot.find(key);
    if (it != g_tickets.slot.end()) {
        if (it->second == counters) return MDX_OK;
        // a library-owned set was handed to this address by an earlier launch: give it back, the caller's replaces it
        if (!g_tickets.caller_owned.count(key)) g_tickets.dev[dv].free_slots.push_back(it->second);
        g_tickets.slot.erase(it);
    }
    g_tickets.slot.emplace(key, static_cast<unsigned*>(counters));
    g_tickets.caller_owned[key] = true;
    return MDX_OK;
}

// Hands the arrival counters of ONE workspace back for reuse (include/mdx.h): call it when the workspace is freed.  Nothing may be
// in flight on it and every hipGraph captured with it must have been destroyed (a captured launch holds the counters' address).
// A caller-owned set (mdx_gemm_bind_counters) is only unbound: its memory is the caller's.
extern "C" int mdx_gemm_release_workspace(const void* workspace) {
    std::lock_guard<std::mutex> lk(g_tickets.mu);
    int n = 0;
    for (auto it = g_tickets.slot.begin(); it != g_tickets.slot.end();) {
        if (it->first.second == workspace) {
            const auto own = g_tickets.caller_owned.find(it->first);
            if (own != g_tickets.caller_owned.end())
                g_tickets.caller_owned.erase(own);
            else
                g_tickets.dev[it->first.first].free_slots.push_back(it->second);
            it = g_tickets.slot.erase(it);
            ++n;
        } else {
            ++it;
        }
    }
    return n;
}

// Frees the arrival counters (nothing may be in flight).  Later launches allocate again.
extern "C" int mdx_gemm_release_counters(void) {
    std::lock_guard<std::mutex> lk(g_tickets.mu);
    int dv = 0;
    (void)hipGetDevice(&dv);
    for (auto& c : g_tickets.chunks) {
        (void)hipSetDevice(c.first);
        (void)hipFree(c.second);
    }
    for (auto& d : g_tickets.dev)
        if (d.second.zero_stream) {
            (void)hipSetDevice(d.first);
            (void)hipStreamDestroy(d.second.zero_stream);
        }
    (void)hipSetDevice(dv);
    g_tickets.chunks.clear();
    g_tickets.slot.clear();
    g_tickets.caller_owned.clear();
    g_tickets.dev.clear();
    return MDX_OK;
}

// The eight-wave 256-pixel conv core takes a launch when the descriptor forces it (tile_m = 256 with stages = 8) or, by default,
// when the shape is eligible, has at least gemm_conv8p_min_m output pixels and at least 128 tiles (UNet batch >= 8 down to the
// 16 x 16 level): below that even a 4-way tail split cannot fill 256 CUs and the 128-row HALO tiles with their own split-K win
// (tools/conv8p_bench.py at UNet batch 2: 462 vs 603 TF/s).
static bool conv8p_wanted(const mdx_gemm_desc* d, const GemmParams& p) {
    if (d->w_frag || d->defer_reduce || d->splitk > 1 || d->asym_pad) return false;
    if (mdx_opt(MDX_OPT_GEMM_BM) || !mdx_opt(MDX_OPT_GEMM_HALO)) return false;
    if (!mdx_conv8p_eligible(p)) return false;
    if (d->tile_m == 256 && (d->stages == 8 || d->stages == 9)) return true;      // forced (9: one phase per 32-deep k-step)
    if (!mdx_opt(MDX_OPT_GEMM_CONV8P)) return false;
    if (d->tile_m != 0 || d->stages != 0) return false;
    if (d->tile_n != 0 && d->tile_n != 64 && d->tile_n != 96 && d->tile_n != 128 && d->tile_n != 160 && d->tile_n != 192) return false;
    if (p.upsample) return mdx_conv8p_tiles(p) >= mdx_opt(MDX_OPT_GEMM_SUBPIXEL_MIN_TILES);      // 2.25x fewer FLOPs: pays from far fewer tiles
    return p.M >= mdx_opt(MDX_OPT_GEMM_CONV8P_MIN_M) && mdx_conv8p_tiles(p) >= 128;
}

static int resolve_launch(const mdx_gemm_desc* d, GemmParams& p, Resolved& r) {
    r.c8 = false;
    if (conv8p_wanted(d, p)) {
        r.c = GemmCfg{256, mdx_conv8p_pick_bn(p, d->tile_n), 64, 3};
        r.bn = r.c.bn;
        r.ns = 1;
        r.stages = 8;
        r.halo = true;
        r.tuned = false;
        r.fixup = false;
        r.c8 = true;
        p.bk = 64;
        p.ktiles = (p.K + 63) / 64;
        p.ktiles_per_split = p.ktiles;
        p.skip_kt_per_split = p.skip_w ? p.skip_kt : 0;
        p.tickets = nullptr;
        (void)mdx_conv8p_plan(p, r.bn, d->workspace_bytes, d->workspace != nullptr, false);
        return MDX_OK;
    }
    r.c = pick_cfg(p);
    const int bn = r.bn = r.c.bn;
    p.bk = r.c.bk;
    p.ktiles = (p.K + r.c.bk - 1) / r.c.bk;
    r.tuned = d->splitk <= 0 && d->tile_m <= 0 && lookup_tuned(p) != nullptr;
    r.stages = p.st_hint;
    if (r.tuned) r.stages = lookup_tuned(p)->st;
    const Tiling tl = choose_tiling(p, bn, d->splitk, d->tile_m);
    r.c.bm = tl.bm;
    int ns = tl.ns;
    if (ns > p.ktiles) ns = p.ktiles;
    if (ns > 1) {
        // shrink to what the caller's workspace can hold
        size_t head;
        // counted in the larger of the two layouts (tile-padded partials of the in-kernel form >= [M][N] slabs): clamping can
        // move a launch from one form to the other, and the split count that fits must fit the form it ends up in
        const size_t slab = std::max(split_bytes(d, p, tl.bm, bn, 2, &head), split_bytes(d, p, tl.bm, bn, 1 << 30, &head));
        const size_t cap = (d->workspace && d->workspace_bytes > head) ? (d->workspace_bytes - head) / slab : 0;
        if ((size_t)ns > cap) ns = (int)cap;
        if (ns < 1) ns = 1;
        if (d->splitk > 1 && ns != d->splitk) {
            mdx_set_error("mdx_gemm_f16: workspace too small for splitk=%d (need %zu bytes)", d->splitk,
                          head + (size_t)d->splitk * slab);
            return MDX_E_WORKSPACE;
        }
    }
    r.halo = r.c.bm >= 128 && halo_eligible(p, r.c.bm);
    if (r.c.bm == 256 && !r.halo) {      // 256-row tiles exist for the HALO conv only (forced tile_m = 256 on another launch)
        mdx_set_error("mdx_gemm_f16: tile_m = 256 needs a launch the 16 x 16-patch HALO conv applies to (3x3, stride 1, H %% 16 == 0, "
                      "W %% 16 == 0, Cin %% 64 == 0)");
        return MDX_E_INVALID;
    }
    if (p.gn_cs && p.ksize == 1) {
        if (p.HoWo % r.c.bm != 0) {
            mdx_set_error("mdx_gemm_f16: the fused input GroupNorm of a dense launch needs tokens per sample (%d) %% tile_m (%d) == 0",
                          p.HoWo, r.c.bm);
            return MDX_E_INVALID;
        }
    } else if (p.gn_cs && !(r.halo && r.bn == 64 && !halo8_eligible(p))) {
        mdx_set_error("mdx_gemm_f16: the fused input GroupNorm needs a launch that resolves to the HALO 3x3 kernel with 64-column "
                      "tiles (ask mdx_gemm_query first)");
        return MDX_E_INVALID;
    }
    if (p.skip_w && !r.halo) {
        mdx_set_error("mdx_gemm_f16: the fused skip needs a launch that resolves to the HALO 3x3 kernel (ask mdx_gemm_query first)");
        return MDX_E_INVALID;
    }
    p.nsplit = ns;
    if (r.halo) {
        // chunk-aligned splits: a split owns whole 64-channel chunks (9 K tiles each)
        const int chunks = p.cin / 64;
        if (ns > chunks) ns = chunks;
        const int cps = (chunks + ns - 1) / ns;
        p.ktiles_per_split = cps * 9;
    } else {
        p.ktiles_per_split = (p.ktiles + ns - 1) / ns;
    }
    p.nsplit = (p.ktiles + p.ktiles_per_split - 1) / p.ktiles_per_split;  // no empty splits
    r.ns = p.nsplit;
    p.skip_kt_per_split = p.skip_w ? (p.skip_kt + p.nsplit - 1) / p.nsplit : 0;      // every split takes its share of the skip tiles
    r.fixup = r.ns > 1 && fixup_eligible(d, p, r.c.bm, bn, r.ns);
    p.tickets = r.fixup ? reinterpret_cast<unsigned*>(p.ws) : nullptr;
    if (r.ns > 1 && !r.fixup) p.ws += MDX_TICKET_SLOTS;     // [split][M][N] slabs of the reduce-kernel path start behind the head
    return MDX_OK;
}

static int mdx_internal_resolve_check(const mdx_gemm_desc* d, GemmParams& p) {
    Resolved r;
    return resolve_launch(d, p, r);
}

// Rows per colstats_out row block this launch would produce (0 = it cannot): the M tile for a single-pass launch (a HALO
// patch is one row block), CS_ROWS for a split-K launch; a row block never straddles two samples.
static int colstats_rows(const GemmParams& p, const Resolved& r) {
    if (p.out_mode != MDX_OUT_ROWMAJOR || p.epilogue != MDX_EPI_NONE || p.n_split || p.ln_stats || p.stats_out || p.out_bs)
        return 0;
    if (r.ns > 1 && !r.fixup) return p.HoWo % CS_ROWS == 0 ? CS_ROWS : 0;
    if (r.halo) return halo8_eligible(p) && r.c.bm == 128 ? 0 : r.c.bm;
    return p.HoWo % r.c.bm == 0 ? r.c.bm : 0;
}

// Launches the lean dense kernel (dense.hip) covers: dense (ksize 1, stride 1, one source, K tiles of whole 64-channel chunks),
// row-major through the staged epilogue (unsplit or in-kernel reduce), four waves, plain / GEGLU epilogue, no per-sample row bias,
// no out_bs, not the GroupNorm-on-A form; tile ids must fit the multiply-high decode.
static bool lean_dense_eligible(const GemmParams& p, bool swap, bool fastk, bool nw8, int ntiles) {
    if (!mdx_opt(MDX_OPT_GEMM_LEAN_DENSE) || !p.dense_issue) return false;
    if (!(p.ksize == 1 && p.stride == 1 && !p.upsample && p.c2 == 0 && fastk && swap && !nw8)) return false;
    if (p.gn_cs || p.rowbias || p.out_bs || p.skip_w) return false;
    if (p.epilogue != MDX_EPI_NONE && p.epilogue != MDX_EPI_GEGLU) return false;
    // buffer descriptors of the epilogue prefetches: 32-bit offsets
    if ((size_t)p.M * (size_t)(p.residual ? p.residual_ld : 0) * 2 >= 0x80000000ull) return false;
    if (p.ln_stats && (size_t)p.M * (size_t)p.ln_nt * 8 >= 0x80000000ull) return false;
    return ntiles < 65536 && p.tiles_m < 65536 && p.tiles_n < 65536;
}

// Tile grid, ring depth and kernel form of a resolved launch of the generic / lean dense / HALO kernels (not conv8p): shared by
// mdx_gemm_f16 and mdx_gemm_query, which reports the form.
struct LaunchGeom {
    dim3 grid;
    int ntiles, ring, st_req;
    bool fastk, nw8, swap, lean;
};

static bool lean_dense_has(int bm, int bn, int ring) {      // the instantiations dense.hip carries (launch_dense_ns)
    if (ring < 2) return false;
    if (bm == 64) return ring <= 6;
    if (bm == 128 && bn == 64) return ring <= 4;
    return bm == 128 && bn == 128 && ring <= 3;
}

static int launch_geometry(GemmParams& p, const Resolved& rs, LaunchGeom& g) {
    const GemmCfg& c = rs.c;
    const int bn = rs.bn, ns = rs.ns;
    p.tiles_m = (p.M + c.bm - 1) / c.bm;
    p.tiles_n = (p.N + bn - 1) / bn;
    g.fastk = (p.cin % 64 == 0) && (p.c2 == 0 || p.c1 % 64 == 0);
    MDX_REQUIRE(g.fastk || p.c2 == 0, "mdx_gemm_f16: two-source input needs c1 %% 64 == 0 and Cin %% 64 == 0");
    g.ntiles = p.tiles_m * p.tiles_n;
    p.tiles_per_xcd = (g.ntiles + 7) / 8;
    p.inv_tiles_n = p.tiles_n > 1 ? (unsigned)((0x100000000ull + (unsigned)p.tiles_n - 1) / (unsigned)p.tiles_n) : 0u;
    p.inv_tiles_m = p.tiles_m > 1 ? (unsigned)((0x100000000ull + (unsigned)p.tiles_m - 1) / (unsigned)p.tiles_m) : 0u;
    p.res_bytes = p.residual ? (unsigned)std::min<size_t>((size_t)p.M * (size_t)p.residual_ld * 2, 0x7fffffffull) : 0u;
    // share the bigger operand inside an XCD: unique activation bytes vs weight bytes
    p.n_fastest = ((size_t)p.M * p.cin >= (size_t)p.N * p.K) ? 1 : 0;
    g.grid = dim3(8 * p.tiles_per_xcd, ns);
    p.spread = (p.tiles_m == 1 && mdx_opt(MDX_OPT_GEMM_SPREAD)) ? 1 : 0;
    if (p.spread) g.grid = dim3(g.ntiles * ns, 1);
    g.ring = c.ns;
    g.nw8 = rs.stages >= 10;                           // eight waves per block (generic kernel, 128-row tiles)
    g.st_req = g.nw8 ? rs.stages - 8 : rs.stages;      // requested ring depth (0 = the rule below)
    if (!(mdx_opt(MDX_OPT_GEMM_RING) >= 2 && mdx_opt(MDX_OPT_GEMM_RING) <= 5)) {
        // ring depth: three stages wherever they still leave two blocks per CU (every tile but 128 x 128: 3 x 24 KB), and for
        // 128 x 128 tiles when the grid has at most one block per CU anyway; otherwise two.  tools/tune_gemm.py measures both
        // depths per shape (round 2: 143 of 153 retuned rows chose three) and the table overrides this rule.
        g.ring = (g.ntiles * ns <= 256 || c.bm + bn <= 192) ? 3 : 2;
        if (g.st_req >= 2 && g.st_req <= 6) g.ring = g.st_req;
    }
    g.swap = (ns == 1 || rs.fixup) && (p.out_mode == MDX_OUT_ROWMAJOR);
    g.lean = !rs.halo && lean_dense_eligible(p, g.swap, g.fastk, g.nw8, g.ntiles) && lean_dense_has(c.bm, bn, g.ring);
    return MDX_OK;
}

// What mdx_gemm_f16 would launch for this descriptor (no launch): out7 = {tile_m, tile_n, splitk, kernel (0 generic implicit
// GEMM, 1 HALO conv, 2 lean dense kernel of dense.hip), from_tuned_table, colstats rows per block, in-kernel split-K reduce}.  Parity tests use it to assert that the measured tile table
// (gemm_tuned.inc) is actually hit at the benchmarked shapes; the UNet plan asks it where GroupNorm statistics can come from.
extern "C" int mdx_gemm_query(const mdx_gemm_desc* d, int* out7) {
    GemmParams p{};
    int rc = fill_params(d, p);
    if (rc != MDX_OK) return rc;
    MDX_REQUIRE(out7, "mdx_gemm_query: null output");
    Resolved r;
    rc = resolve_launch(d, p, r);
    if (rc != MDX_OK) return rc;
    out7[0] = r.c.bm;
    out7[1] = r.bn;
    out7[2] = r.ns;
    int form = r.halo ? 1 : 0;
    if (!r.c8 && !r.halo) {
        LaunchGeom lg;
        rc = launch_geometry(p, r, lg);
        if (rc != MDX_OK) return rc;
        if (lg.lean) form = 2;
    }
    out7[3] = form;
    out7[4] = r.tuned ? 1 : 0;
    out7[5] = colstats_rows(p, r);
    out7[6] = r.fixup ? 1 : 0;
    return MDX_OK;
}

// Producer side of mdx_groupnorm_from_splitk_f16: the launch geometry of `d` as mdx_gemm_f16 resolves it.
int mdx_internal_split_info(const mdx_gemm_desc* d, MdxSplitInfo* info) {
    GemmParams p{};
    int rc = fill_params(d, p);
    if (rc != MDX_OK) return rc;
    Resolved r;
    rc = resolve_launch(d, p, r);
    if (rc != MDX_OK) return rc;
    MDX_REQUIRE(p.out_mode == MDX_OUT_ROWMAJOR && p.epilogue == MDX_EPI_NONE && !p.n_split && !p.ln_stats && !p.stats_out &&
                    !p.out_bs && !p.colstats_out && p.out_ld == p.N && p.N % 8 == 0,
                "deferred split-K reduce: plain dense row-major producers only");
    info->ws = p.ws;
    info->nsplit = r.ns;
    info->M = p.M;
    info->N = p.N;
    info->HoWo = p.HoWo;
    info->B = p.B;
    info->bias = p.bias;
    info->rowbias = p.rowbias;
    info->rowbias_ld = p.rowbias_ld;
    info->residual = p.residual;
    info->residual_ld = p.residual_ld;
    info->out = p.out;
    return MDX_OK;
}

extern "C" int mdx_gemm_f16(const mdx_gemm_desc* d, mdx_stream_t s) {
    GemmParams p{};
    int rc = fill_params(d, p);
    if (rc != MDX_OK) return rc;
    hipStream_t st = (hipStream_t)s;
    Resolved rs;
    rc = resolve_launch(d, p, rs);
    if (rc != MDX_OK) return rc;
    GemmCfg c = rs.c;
    const int bn = rs.bn;
    int ns = rs.ns;
    const bool halo = rs.halo;
    if (p.colstats_out) {
        const int rows = colstats_rows(p, rs);
        MDX_REQUIRE(rows > 0, "mdx_gemm_f16: this launch cannot produce column statistics (ask mdx_gemm_query first)");
        MDX_REQUIRE((p.M + rows - 1) / rows <= d->colstats_cap,
                    "mdx_gemm_f16: colstats_out holds %d row blocks, this launch writes %d (%d rows each)", d->colstats_cap,
                    (p.M + rows - 1) / rows, rows);
    }
    if (rs.c8) {
        if (p.c8_split > 1) {       // tail tiles split along K: partials in the workspace, library-owned arrival counters
            MDX_REQUIRE((uintptr_t)p.ws % 16 == 0, "mdx_gemm_f16: workspace must be 16-byte aligned");
            const int trc = ticket_slot(d->workspace, &p.tickets);
            if (trc != MDX_OK) return trc;
        }
        rc = mdx_conv8p_launch(p, bn, st);
        if (rc != MDX_OK) return rc;
        MDX_LAUNCH_CHECK("mdx_gemm_f16(conv8p)");
        return MDX_OK;
    }
    MDX_REQUIRE(!d->defer_reduce || ns > 1, "mdx_gemm_f16: defer_reduce set but the launch does not split K");
    MDX_REQUIRE(!rs.fixup || ((uintptr_t)p.ws % 16 == 0), "mdx_gemm_f16: workspace must be 16-byte aligned");
    if (rs.fixup) {
        const int trc = ticket_slot(d->workspace, &p.tickets);
        if (trc != MDX_OK) return trc;
    }
    LaunchGeom lg;
    rc = launch_geometry(p, rs, lg);
    if (rc != MDX_OK) return rc;
    const bool fastk = lg.fastk, nw8 = lg.nw8, swap = lg.swap;
    const int ntiles = lg.ntiles;
    MDX_REQUIRE(!p.xa_k || (lg.lean && !halo && bn == 64 && ns == 1 && (p.HoWo % c.bm) == 0),
                "mdx_gemm_f16: the cross-attention epilogue needs the lean dense kernel on %d x 64 tiles inside one sample (got tile %d x %d, %d splits)",
                c.bm, c.bm, bn, ns);
    dim3 grid = lg.grid;
    p.trace = (g_gemm_trace && (size_t)grid.x * grid.y <= g_gemm_trace_slots) ? g_gemm_trace : nullptr;
    GemmCfg cc = c;
    cc.ns = lg.ring;
    const int st_req = lg.st_req;
    MDX_REQUIRE(!d->w_frag || halo, "mdx_gemm_f16: fragment-major weights (w_frag) are read by the HALO 3x3 conv only");
    bool ok;
    if (halo) {
        // weight ring depth: three stages where two blocks per CU still fit (64-column tiles: 46 KB halos + 3 x 8 KB; measured
        // -5...-25 % against two stages at the UNet shapes, batch 2 and 16) and for 256-pixel patches, which own the CU; 128 x 128
        // tiles keep two (a third stage would evict the second block)
        int nsb = (c.bm == 256 || bn == 64) ? 3 : 2;
        if (st_req >= 2 && st_req <= 4) nsb = st_req;
        if (mdx_opt(MDX_OPT_HALO_NSB) >= 2 && mdx_opt(MDX_OPT_HALO_NSB) <= 4) nsb = mdx_opt(MDX_OPT_HALO_NSB);
        if (d->w_frag) {
            MDX_REQUIRE(c.bm == 128, "mdx_gemm_f16: fragment-major weights run on 128-row HALO tiles only (got tile_m %d)", c.bm);
            // 128-column tiles whose split-K partials go to slabs (no in-kernel reduce): waves side by side along N
            const bool w4 = bn == 128 && ns > 1 && !rs.fixup && !swap;
            if (halo8_eligible(p)) {
                if (w4) launch_halo_bdir_w4<8>(p, grid, st);
                else if (bn == 128) launch_halo_bdir<128, 8>(p, swap, grid, st); else launch_halo_bdir<64, 8>(p, swap, grid, st);
            } else {
                if (w4) launch_halo_bdir_w4<16>(p, grid, st);
                else if (bn == 128) launch_halo_bdir<128, 16>(p, swap, grid, st); else launch_halo_bdir<64, 16>(p, swap, grid, st);
            }
        } else if (c.bm == 256) {
            if (bn == 128) launch_halo_cfg<256, 128>(p, nsb, swap, grid, st); else launch_halo_cfg<256, 64>(p, nsb, swap, grid, st);
        } else if (halo8_eligible(p)) {
            if (bn == 128) launch_halo_cfg<128, 128, 8>(p, nsb, swap, grid, st); else launch_halo_cfg<128, 64, 8>(p, nsb, swap, grid, st);
        } else {
            if (bn == 128) launch_halo_cfg<128, 128>(p, nsb, swap, grid, st); else launch_halo_cfg<128, 64>(p, nsb, swap, grid, st);
        }
        ok = true;
    } else if (p.gn_cs) {       // (ksize 1, checked in fill_params / resolve_launch) GroupNorm of the input on the A fragments
        MDX_REQUIRE(fastk, "mdx_gemm_f16: the fused input GroupNorm needs Cin %% 64 == 0");
        // (ring depth as the occupancy rule / tile table chose it, capped at the three stages this form is built with)
        const int gns = cc.ns >= 3 ? 3 : 2;
        if (cc.bm == 64) {
            if (bn == 128) launch_gna<64, 128>(p, gns, swap, grid, st); else launch_gna<64, 64>(p, gns, swap, grid, st);
        } else {
            if (bn == 128) launch_gna<128, 128>(p, gns, swap, grid, st); else launch_gna<128, 64>(p, gns, swap, grid, st);
        }
        ok = true;
    } else if (lg.lean && mdx_dense_launch(p, cc.bm, bn, cc.ns, grid, st)) {
        ok = true;      // dense.hip: the same tile program with a division-free prologue and the epilogue's reads prefetched
    } else if (cc.bm == 64)
        ok = (bn == 128) ? launch_bn<64, 128>(cc, p, swap, fastk, grid, st) : launch_bn<64, 64>(cc, p, swap, fastk, grid, st);
    else
        ok = (bn == 128) ? launch_bn<128, 128>(cc, p, swap, fastk, grid, st, nw8) : launch_bn<128, 64>(cc, p, swap, fastk, grid, st, nw8);
    MDX_REQUIRE(ok, "mdx_gemm_f16: unsupported tile configuration bk=%d ns=%d", c.bk, c.ns);
    MDX_LAUNCH_CHECK("mdx_gemm_f16");
    if (ns > 1 && !d->defer_reduce && !rs.fixup) {
        const int ncols = p.epilogue == MDX_EPI_GEGLU ? p.N / 2 : p.N;
        const size_t total = (size_t)p.M * (ncols / 8);
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        if (p.colstats_out)
            hipLaunchKernelGGL(splitk_reduce_colstats_kernel, dim3((p.N + 63) / 64, (p.M + CS_ROWS - 1) / CS_ROWS), dim3(256), 0,
                               st, p);
        else
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p);
        MDX_LAUNCH_CHECK("mdx_gemm_f16(splitk reduce)");
    }
    return MDX_OK;
}

// First-use tuner (include/mdx.h).  The tile table (gemm_tuned.inc) covers the shapes of the benchmarked configurations; any
// other resolution / batch resolves through the cost model, which is 10-20 % off on some shapes.  This entry measures the
// launch forms the library has for ONE descriptor on the caller's stream -- tile_m x tile_n x split-K, every distinct form
// mdx_gemm_query resolves them to -- and returns the fastest as values for the descriptor's tile_m / tile_n / splitk / stages
// override fields (all zero = the library's own choice was the fastest, or within 2 % of it).  The caller keeps the answer (the
// cache is on the caller's side: minddiffusion_amd/ops.py tune_cache); the library keeps nothing.  It is the one entry that
// SYNCHRONISES (event waits on `s`) and so cannot be captured; the descriptor's output buffer is overwritten by every trial.
extern "C" int mdx_gemm_tune(const mdx_gemm_desc* d, mdx_stream_t s, void* flush, size_t flush_bytes, int reps, int* best4,
                             float* us2) {
    MDX_REQUIRE(d != nullptr && best4 != nullptr, "mdx_gemm_tune: null argument");
    MDX_REQUIRE(!d->defer_reduce && !d->colstats_out && !d->w_frag,
                "mdx_gemm_tune: descriptors whose consumer depends on the launch form (colstats_out, defer_reduce) or whose weights "
                "are packed for one form (w_frag) keep the library's choice");
    hipStream_t st = reinterpret_cast<hipStream_t>(s);
    if (reps < 1) reps = 5;
    if (reps > 31) reps = 31;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        if (e0) (void)hipEventDestroy(e0);
        mdx_set_error("mdx_gemm_tune: hipEventCreate failed");
        return MDX_E_HIP;
    }
    auto measure = [&](const mdx_gemm_desc& c, float* us) -> int {
        int rc = mdx_gemm_f16(&c, st);       // warm-up; also the validity check of this form
        if (rc != MDX_OK) return rc;
        float t[32];
        for (int r = 0; r < reps; ++r) {
            if (flush && flush_bytes) (void)hipMemsetAsync(flush, r & 1, flush_bytes, st);      // evict L2 / MALL: cold weights
            (void)hipEventRecord(e0, st);
            rc = mdx_gemm_f16(&c, st);
            (void)hipEventRecord(e1, st);
            if (rc != MDX_OK) return rc;
            if (hipEventSynchronize(e1) != hipSuccess) {
                mdx_set_error("mdx_gemm_tune: a trial launch failed on the device");
                return MDX_E_HIP;
            }
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            t[r] = ms * 1e3f;
        }
        std::sort(t, t + reps);
        *us = t[reps / 2];
        return MDX_OK;
    };
    mdx_gemm_desc base = *d;
    base.tile_m = base.tile_n = base.splitk = base.stages = 0;
    int q0[7];
    int rc = mdx_gemm_query(&base, q0);
    float t_auto = 0.f;
    if (rc == MDX_OK) rc = measure(base, &t_auto);
    if (rc != MDX_OK) {
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        return rc;
    }
    float t_best = t_auto;
    int best[4] = {0, 0, 0, 0};
    std::vector<long> seen;
    auto form = [](const int* q) { return (long)q[0] | ((long)q[1] << 10) | ((long)q[2] << 20) | ((long)q[3] << 30) | ((long)q[6] << 31); };
    seen.push_back(form(q0));
    static const int BMS[] = {64, 128, 256}, BNS[] = {64, 128}, NSS[] = {1, 2, 3, 4, 6, 8, 12, 16, 20};
    for (int bm : BMS)
        for (int bn : BNS)
            for (int ns : NSS) {
                mdx_gemm_desc c = base;
                c.tile_m = bm;
                c.tile_n = bn;
                c.splitk = ns;
                int q[7];
                if (mdx_gemm_query(&c, q) != MDX_OK) continue;
                if (q[0] != bm || q[1] != bn || q[2] != ns) continue;      // clamped to something another trial covers
                const long f = form(q);
                if (std::find(seen.begin(), seen.end(), f) != seen.end()) continue;
                seen.push_back(f);
                float t = 0.f;
                if (measure(c, &t) != MDX_OK) continue;
                if (t < t_best) {
                    t_best = t;
                    best[0] = bm, best[1] = bn, best[2] = ns, best[3] = 0;
                }
            }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (!(t_best < t_auto * 0.98f)) {
        best[0] = best[1] = best[2] = best[3] = 0;
        t_best = t_auto;
    }
    for (int i = 0; i < 4; ++i) best4[i] = best[i];
    if (us2) us2[0] = t_auto, us2[1] = t_best;
    return MDX_OK;
}
