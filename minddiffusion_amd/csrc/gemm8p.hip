// gemm8p: the 256-row, eight-wave, phase-staggered dense GEMM core for gfx950 (MI355X) -- round 4.
//
//   out[m][n] = sum_k A[m][k] * W[n][k]   (+ the row-major epilogues of gemm.hip: bias, residual, GEGLU / GELU, LayerNorm fold,
//                                            row / column statistics, q|k row-major + V^T split output)
//
// Replaces nn.Dense / 1x1 nn.Conv2d over token rows (attention.py:44, 66, 108-112, 212, 231; openaimodel.py:174) wherever the launch
// has M >= 4096 rows and at least 128 tiles of 256 x 128: the transformer blocks of the 32 x 32 and 16 x 16 levels at UNet batch >= 8
// (BASELINE configs 2-4).  Same synchronisation structure as conv8p.hip (read there first): two groups of four waves half a phase
// apart, a phase = one K tile of 64 (16 `ds_read_b128` -> vmcnt(0) -> barrier -> 16 MFMAs 32x32x16 with the six DMA instructions of
// K tile t + 2 interleaved between them -> barrier), three LDS stages of A (256 x 128 B) + B (128 x 128 B), `s_setprio 1` around the
// MFMA burst.  Wave layout 4 (M) x 2 (N), wave tile 64 x 64, accumulators in the layout gemm_epilogue expects (32x32 tiles, C^T),
// so every row-major epilogue of the 4-wave kernels works unchanged.
// Per K tile a CU moves 48 KB through the LDS DMA for 2 x 256 x 128 x 64 FLOP (47 B / clk at the MFMA peak -- 2.4x the conv core's
// figure: this kernel is expected to sit nearer the L2 -> LDS rate than the matrix pipe).
#include "gemm_internal.h"

namespace {

constexpr int G8_NT = 512;
constexpr int G8_BM = 256;

template <int BN>
__global__ __launch_bounds__(G8_NT) void gemm8p_kernel(const GemmParams p) {
    constexpr int BM = G8_BM, NW = 8;
    constexpr int TM = 2, TN = BN / 64;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int AJ = BM / 8 / NW;          // 4 DMA instructions per wave per A tile (8 rows of 128 B each)
    constexpr int BJ = BN / 8 / NW;          // 2 ... per B tile
    static_assert(BN == 128, "three stages of 256 x 64 + BN x 64 halves must fit 160 KB; the GEGLU epilogue wants 128");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;       // gemm_epilogue's wave map (4 x 2)
    const int grp = wave >> 2;                     // stagger group: waves w and w + 4 share a SIMD
    const int hi = lane >> 5, l31 = lane & 31;

    const int ntiles = p.tiles_m * p.tiles_n;
    const int tile_id = (blockIdx.x & 7) * p.tiles_per_xcd + (blockIdx.x >> 3);
    if (tile_id >= ntiles) return;
    int tile_m, tile_n;
    if (p.n_fastest) {
        tile_m = tile_id / p.tiles_n;
        tile_n = tile_id - tile_m * p.tiles_n;
    } else {
        tile_n = tile_id / p.tiles_m;
        tile_m = tile_id - tile_n * p.tiles_m;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(p.a, p.a_bytes);
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, p.w_bytes);

    // loaders: DMA instruction j of a wave covers tile rows (wave * J + j) * 8 .. + 7 (lane / 8), 16-byte position lane % 8 holds
    // logical chunk (lane % 8) ^ ((row >> 1) & 7); the weights are stored pre-swizzled and tile-major (1 KiB contiguous per piece)
    const int lrow = lane >> 3, lchk = lane & 7;
    unsigned a_off[AJ], b_off[BJ];
    const unsigned row_bytes = (unsigned)p.cin * 2u;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int row = (wave * AJ + j) * 8 + lrow;
        const int m = m0 + row;
        a_off[j] = m < p.M ? (unsigned)m * row_bytes + (unsigned)((lchk ^ ((row >> 1) & 7)) * 16) : MDX_OOB;
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int n = n0 + (wave * BJ + j) * 8 + lrow;
        b_off[j] = (unsigned)(((size_t)(n >> 6) * p.kt64) * 8192 + ((n & 63) * 8 + lchk) * 16);
    }
    auto dma_a = [&](int j, int kt, int st) {
        dma16(rs_a, smem + st * STAGE + (wave * AJ + j) * 1024, a_off[j] == MDX_OOB ? MDX_OOB : a_off[j] + (unsigned)kt * 128u);
    };
    auto dma_b = [&](int j, int kt, int st) {
        dma16(rs_w, smem + st * STAGE + A_BYTES + (wave * BJ + j) * 1024, b_off[j] + (unsigned)kt * 8192u);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int swz = (l31 >> 1) & 7;
    const int a_row_off = (wm * 64 + l31) * 128;
    const int b_row_off = A_BYTES + (wn * (BN / 2) + l31) * 128;

    float bpre[16];
    gemm_bias_prefetch<BN, true, NW>(p, n0, bpre);

    const int nt = p.ktiles;
    // prologue: K tiles 0 and 1
#pragma unroll
    for (int i = 0; i < 2; ++i)
        if (i < nt) {
#pragma unroll
            for (int j = 0; j < AJ; ++j) dma_a(j, i, i);
#pragma unroll
            for (int j = 0; j < BJ; ++j) dma_b(j, i, i);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();       // the second group runs one barrier (half a phase) behind

    int rd = 0, wr = 2;
    for (int t = 0; t < nt; ++t) {
        const char* sb = smem + rd * STAGE;
        f16x8 af[4][TM], bf[4][TN];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int coff = ((2 * s + hi) ^ swz) << 4;
#pragma unroll
            for (int i = 0; i < TM; ++i) af[s][i] = *reinterpret_cast<const f16x8*>(sb + a_row_off + i * 32 * 128 + coff);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[s][j] = *reinterpret_cast<const f16x8*>(sb + b_row_off + j * 32 * 128 + coff);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // K tile t + 1 (issued one phase ago; this wave's share)
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const bool more = t + 2 < nt;
        __builtin_amdgcn_s_setprio(1);
        // 16 MFMAs with the 6 DMA instructions of K tile t + 2 between them (one per MFMA pair: an LDS-DMA issue hides in an MFMA gap)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[s][j], af[s][i], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const int g = s * TM + i;          // 0 .. 7
                if (more) {
                    if (g < AJ) dma_a(g, t + 2, wr);
                    else if (g < AJ + BJ) dma_b(g - AJ, t + 2, wr);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        rd = rd == 2 ? 0 : rd + 1;
        wr = wr == 2 ? 0 : wr + 1;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
    __syncthreads();
    gemm_epilogue<BM, BN, true, NW>(p, acc, smem, LinearRows{m0}, n0, 0, bpre, tile_m, tile_id);
}

template <int BN>
void g8_launch(const GemmParams& p, dim3 grid, hipStream_t st) {
    constexpr size_t ring = 3u * (G8_BM + BN) * 128u;
    constexpr size_t epi = (size_t)G8_BM * (BN + 8) * 2 + 4096;
    constexpr size_t lds = ring > epi ? ring : epi;
    static MdxPerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm8p_kernel<BN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL((gemm8p_kernel<BN>), grid, dim3(G8_NT), lds, st, p);
}


// ---------------------------------------------------------------------------------------------------------------------------------
// gemm8q: the 256 x 256 tile on the same eight-wave, phase-staggered skeleton (round 4; OPT-IN, option gemm_dense8q: measured equal).
//
// Why it was built.  The 256 x 128 tile above and the 128 x 128 four-wave tile move 48 / 32 KB through the LDS DMA per 64-wide K tile
// for 2 x 256 x 128 x 64 / 2 x 128 x 128 x 64 FLOP (87 / 64 FLOP per DMA byte) and measure 600-820 TF/s on the token GEMMs of UNet
// batch >= 8, while the 3x3 conv core (conv8p.hip), 2.5x lighter per FLOP through its halo, runs 1150-1400 on the same skeleton.  A
// 256 x 256 tile needs 64 KB per 64-wide K tile for twice the FLOP: 128 FLOP per byte, the guide's 256^2 template's intensity.
// What the timing ablations say (tools/exp/r04e_ablate.py, profiles/r04_gemm8q_ablate.txt; M = 16384, N = K = 5120): whole kernel
// 855 us (991 TF/s); WITHOUT the in-loop DMA 511 us -- the MFMA + ds_read stream alone runs at 1.68 PF; DMA alone (no MFMA, no reads)
// 634 us: the operand stream, 6.7 GB = 10.6 TB/s chip-wide for this 220 MB working set, IS the bound, and the two streams overlap
// only partly (855 vs max(634, 511)).  Fetching full 128-byte rows (two 64 KB stages, a K tile in two phases) instead of 64-byte row
// segments changed nothing (DMA alone 710 us: the path is priced per byte), a persistent tile loop cost 3 %.  At the K = 640 / 1280 of
// the real launches 40 % of the time is outside the K loop (a 256 x 256 GEGLU epilogue is ~10 us of VALU work that nothing overlaps
// with one 512-thread block per CU; two co-resident four-wave blocks overlap it with each other's K loop), so the form is equal to
// the four-wave tiles inside an evaluation (tools/eval_ab.py: Wukong +0.5 %, 768^2 -0.3 %).
// Geometry.  Three 64-wide stages of 256 + 256 rows would need 192 KB, so the ring holds K slabs of 32: slab = A[256][32] + B[256][32]
// fp16 = 32 KB, FOUR slabs (128 KB), a phase = one slab: 12 `ds_read_b128` -> counted `vmcnt` -> barrier -> 16 MFMAs 32x32x16 ->
// barrier.  Wave layout 4 (M) x 2 (N) as gemm_epilogue expects, wave tile 64 x 128 (128 accumulator + 48 fragment registers), the
// two wave groups half a phase apart.
// LDS image: 64-byte rows, a DMA instruction covers 16 rows; physical 16-byte slot j of row r holds logical chunk j ^ ((r >> 2) & 3)
// (swizzle on the SOURCE offset and again on the read): the 16 lanes of a `ds_read_b128` group (16 consecutive rows, one logical
// chunk) land on 16 distinct slots of the 256-byte bank row.  The weights are the packed tiles of ops.pack_gemm_weight AS IS: logical
// chunk q of row n of K tile kt sits at ((n & 63) * 8 + (q ^ ((n >> 1) & 7))) * 16 of its 8 KiB block; slab ks is the half
// q = 4 (ks & 1) + 0 .. 3 of K tile ks >> 1 -- every lane computes its own source offset, so nothing is repacked.
// DMA issue, two forms:
//   MFMA-burst form (default): the 4 DMA instructions of slab t + 3 go between the MFMAs of phase t.
//     WAR  slot (t - 1) & 3 was last read in phase t - 1: those reads have returned when their wave passed lgkmcnt(0) behind barrier
//          2t-2 (2t-1); every wave arrives at barrier 2t after that; the refill is issued behind barrier 2t (2t+1).
//     RAW  the batch issued in phase t is waited for in the read burst of phase t + 2 -- `vmcnt(4)`: only the batch of phase t + 1
//          may still be in flight (in-order completion) -- before its wave ARRIVES at barrier 2t+4 (2t+5); first read in phase t + 3.
//   read-burst form (VAR bit 5; the guide's "load-issuing vs MFMA-entering wave split", T5): the wave group that is READING issues
//     the DMA (slab t + 2, behind its ds_reads), so that a full VMEM queue stalls a wave with slack instead of the one feeding the
//     matrix pipe.  WAR: slot (t + 2) & 3 was last read in phase t - 2, retired behind barrier 2t-3 at the latest; the issuing wave
//     has passed barrier 2t-1.  RAW: waited for in the read burst of phase t + 1, `vmcnt(4)` = the batch issued just before the wait
//     may fly on; first read in phase t + 2.
constexpr int Q8_BM = 256, Q8_BN = 256;
constexpr int Q8_HALF = 256 * 64;          // one operand's slab: 256 rows x 64 B
constexpr int Q8_SLAB = 2 * Q8_HALF;       // 32 KB
constexpr int Q8_RING = 4;

// VAR (option gemm_dense8q_var): bits 0-2 are timing ablations whose results are WRONG: bit 0 = no DMA in the loop, bit 1 = no ds_read
// in the loop, bit 2 = no MFMA; bit 3 = no half-phase stagger, bit 4 = no s_setprio, bit 5 = DMA issue in the read burst.
template <int VAR>
__global__ __launch_bounds__(G8_NT) void gemm8q_kernel(const GemmParams p) {
    constexpr int BM = Q8_BM, BN = Q8_BN, NW = 8;
    constexpr int TM = 2, TN = 4;
    constexpr bool RB = (VAR & 32) != 0;
    constexpr int DIST = RB ? 2 : 3;           // slabs the DMA runs ahead of the phase that issues it
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;       // gemm_epilogue's wave map (4 x 2)
    const int grp = wave >> 2;                     // stagger group: waves w and w + 4 share a SIMD
    const int hi = lane >> 5, l31 = lane & 31;

    const int ntiles = p.tiles_m * p.tiles_n;
    const int tile_id = (blockIdx.x & 7) * p.tiles_per_xcd + (blockIdx.x >> 3);
    if (tile_id >= ntiles) return;
    int tile_m, tile_n;
    if (p.n_fastest) {
        tile_m = tile_id / p.tiles_n;
        tile_n = tile_id - tile_m * p.tiles_n;
    } else {
        tile_n = tile_id / p.tiles_m;
        tile_m = tile_id - tile_n * p.tiles_m;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(p.a, p.a_bytes);
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, p.w_bytes);

    // loaders: DMA instruction j of a wave covers slab rows (2 wave + j) * 16 .. + 15 (lane >> 2), 16-byte slot lane & 3
    const int lrow = lane >> 2, lslot = lane & 3;
    unsigned a_off[2], b_off[2], b_key[2];
    const unsigned row_bytes = (unsigned)p.cin * 2u;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave * 2 + j) * 16 + lrow;
        const int lchunk = lslot ^ ((row >> 2) & 3);            // logical chunk of the slab this lane fetches
        const int m = m0 + row;
        a_off[j] = m < p.M ? (unsigned)m * row_bytes + (unsigned)(lchunk * 16) : MDX_OOB;
        const int n = n0 + row;
        b_off[j] = (unsigned)(((size_t)(n >> 6) * p.kt64) * 8192 + (size_t)(n & 63) * 128);
        b_key[j] = (unsigned)(lchunk | ((((n & 63) >> 1) & 7) << 4));     // logical chunk | the packed tile's row key
    }
    auto dma_a = [&](int j, int ks, int slot) {
        dma16(rs_a, smem + slot * Q8_SLAB + (wave * 2 + j) * 1024, a_off[j] == MDX_OOB ? MDX_OOB : a_off[j] + (unsigned)ks * 64u);
    };
    auto dma_b = [&](int j, int ks, int slot) {
        const unsigned q = (unsigned)((ks & 1) * 4) + (b_key[j] & 15u);
        dma16(rs_w, smem + slot * Q8_SLAB + Q8_HALF + (wave * 2 + j) * 1024,
              b_off[j] + (unsigned)(ks >> 1) * 8192u + ((q ^ (b_key[j] >> 4)) << 4));
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int swz = (l31 >> 2) & 3;
    const int a_row_off = (wm * 64 + l31) * 64;
    const int b_row_off = Q8_HALF + (wn * (BN / 2) + l31) * 64;

    float bpre[16];
    gemm_bias_prefetch<BN, true, NW>(p, n0, bpre);
    // LayerNorm-fold consumers: this thread's token row folded here, its loads in flight under the prologue DMA (one block per CU:
    // nothing else would hide the round trips at the head of the epilogue)
    float ln_pre[2] = {0.f, 0.f};
    if (p.ln_stats && tid < BM) gemm_ln_row_fold(p, m0 + tid, ln_pre);

    const int nt = p.K >> 5;                 // 32-wide slabs (K is a multiple of 64)
    // prologue: slabs 0 .. DIST - 1
#pragma unroll
    for (int i = 0; i < DIST; ++i)
        if (i < nt) {
            dma_a(0, i, i); dma_a(1, i, i); dma_b(0, i, i); dma_b(1, i, i);
        }
    if (nt >= DIST) {      // slab 0 (and the bias rows, issued before it)
        if (RB) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (!(VAR & 8) && grp == 1) __builtin_amdgcn_s_barrier();       // the second group runs one barrier (half a phase) behind

    f16x8 af[2][TM], bf[2][TN];
    if (VAR & 2) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[s][i] = *reinterpret_cast<const f16x8*>(smem + a_row_off + i * 32 * 64 + s * 32);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[s][j] = *reinterpret_cast<const f16x8*>(smem + b_row_off + j * 32 * 64 + s * 32);
        }
    }
    for (int t = 0; t < nt; ++t) {
        const char* sb = smem + (t & 3) * Q8_SLAB;
        if (!(VAR & 2)) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int coff = ((2 * s + hi) ^ swz) << 4;
#pragma unroll
                for (int i = 0; i < TM; ++i) af[s][i] = *reinterpret_cast<const f16x8*>(sb + a_row_off + i * 32 * 64 + coff);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[s][j] = *reinterpret_cast<const f16x8*>(sb + b_row_off + j * 32 * 64 + coff);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const bool more = !(VAR & 1) && t + DIST < nt;
        const int wr = (t + DIST) & 3;
        if (RB) {
            if (more) {
                dma_a(0, t + DIST, wr); dma_a(1, t + DIST, wr); dma_b(0, t + DIST, wr); dma_b(1, t + DIST, wr);
            }
            __builtin_amdgcn_sched_barrier(0);
            // slab t + 1 (issued in the previous read burst) must have landed; the batch issued just above may fly on
            if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            // slab t + 1 must have landed before anyone reads it in the next phase; the batch of slab t + 2 (issued in phase t - 1) may fly on
            if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (!(VAR & 16)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (!(VAR & 4)) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[s][j], af[s][i], acc[i][j], 0, 0, 0);
                }
                if (!RB) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) {
                        const int g = s * TM + i;          // 0 .. 3: one DMA instruction behind every four MFMAs
                        if (g < 2) dma_a(g, t + DIST, wr);
                        else dma_b(g - 2, t + DIST, wr);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (!(VAR & 16)) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    }
    if (!(VAR & 8) && grp == 0) __builtin_amdgcn_s_barrier();
    __syncthreads();
    gemm_epilogue<BM, BN, true, NW>(p, acc, smem, LinearRows{m0}, n0, 0, bpre, tile_m, tile_id, p.ln_stats ? ln_pre : nullptr);
}

template <int VAR>
void q8_launch(const GemmParams& p, dim3 grid, hipStream_t st) {
    constexpr size_t ring = (size_t)Q8_RING * Q8_SLAB;
    constexpr size_t epi = (size_t)Q8_BM * (Q8_BN + 8) * 2 + 4096;
    constexpr size_t lds = ring > epi ? ring : epi;
    static_assert(lds <= 160 * 1024, "gemm8q: LDS");
    static MdxPerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm8q_kernel<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL(gemm8q_kernel<VAR>, grid, dim3(G8_NT), lds, st, p);
}

}  // namespace

// Dense single-source row-major launches with whole 64-wide K tiles.
bool mdx_gemm8p_eligible(const GemmParams& p) {
    if (!(p.ksize == 1 && p.stride == 1 && !p.upsample && p.c2 == 0)) return false;
    if (p.cin % 64 != 0 || p.cin < 128) return false;
    if (p.out_mode != MDX_OUT_ROWMAJOR || p.gn_cs || p.skip_w) return false;
    if (p.N % 64 != 0 || p.N < 128) return false;
    if (p.epilogue == MDX_EPI_GEGLU && p.N % 128 != 0) return false;
    return true;
}

int mdx_gemm8p_tiles(const GemmParams& p) { return ((p.M + G8_BM - 1) / G8_BM) * ((p.N + 127) / 128); }

int mdx_gemm8p_launch(GemmParams& p, hipStream_t st) {
    p.tiles_m = (p.M + G8_BM - 1) / G8_BM;
    p.tiles_n = (p.N + 127) / 128;
    const int ntiles = p.tiles_m * p.tiles_n;
    p.tiles_per_xcd = (ntiles + 7) / 8;
    p.n_fastest = ((size_t)p.M * p.cin >= (size_t)p.N * p.K) ? 1 : 0;
    p.nsplit = 1;
    p.tickets = nullptr;
    p.spread = 0;
    g8_launch<128>(p, dim3(8 * p.tiles_per_xcd), st);
    return MDX_OK;
}

// ---- the 256 x 256 form (gemm8q_kernel)
bool mdx_gemm8q_eligible(const GemmParams& p) {
    if (!mdx_gemm8p_eligible(p)) return false;
    if (p.n_split && p.n_split % Q8_BN != 0) return false;        // a tile is wholly q|k or wholly V
    if (p.epilogue == MDX_EPI_GEGLU && p.N % 128 != 0) return false;
    return true;
}

int mdx_gemm8q_tiles(const GemmParams& p) { return ((p.M + Q8_BM - 1) / Q8_BM) * ((p.N + Q8_BN - 1) / Q8_BN); }

int mdx_gemm8q_launch(GemmParams& p, hipStream_t st) {
    p.tiles_m = (p.M + Q8_BM - 1) / Q8_BM;
    p.tiles_n = (p.N + Q8_BN - 1) / Q8_BN;
    const int ntiles = p.tiles_m * p.tiles_n;
    p.tiles_per_xcd = (ntiles + 7) / 8;
    p.n_fastest = ((size_t)p.M * p.cin >= (size_t)p.N * p.K) ? 1 : 0;
    p.nsplit = 1;
    p.tickets = nullptr;
    p.spread = 0;
    const dim3 grid(8 * p.tiles_per_xcd);
    switch (mdx_opt(MDX_OPT_GEMM_DENSE8Q_VAR)) {       // experiment forms / timing ablations (see gemm8q_kernel); 0 = the default form
        case 1: q8_launch<1>(p, grid, st); break;
        case 2: q8_launch<2>(p, grid, st); break;
        case 4: q8_launch<4>(p, grid, st); break;
        case 5: q8_launch<5>(p, grid, st); break;
        case 6: q8_launch<6>(p, grid, st); break;
        case 8: q8_launch<8>(p, grid, st); break;
        case 16: q8_launch<16>(p, grid, st); break;
        case 32: q8_launch<32>(p, grid, st); break;
        case 33: q8_launch<33>(p, grid, st); break;
        case 36: q8_launch<36>(p, grid, st); break;
        case 48: q8_launch<48>(p, grid, st); break;
        default: q8_launch<0>(p, grid, st); break;
    }
    return MDX_OK;
}
