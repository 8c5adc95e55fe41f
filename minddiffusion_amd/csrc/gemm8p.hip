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
