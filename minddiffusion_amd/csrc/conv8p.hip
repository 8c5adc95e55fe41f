// conv8p: the 256-pixel, eight-wave, phase-staggered 3x3 conv core for gfx950 (MI355X) -- round 4.
//
//   out[m][n] = sum_{tap, c} A[pixel(m) + tap][c] * W[n][tap, c]   (+ conv1x1 skip tiles) + bias + time-embedding row + residual
//
// Replaces nn.Conv2d 3x3 / stride 1 / pad 1 of the ResBlocks (openaimodel.py:136-138, 159-163, 174, 201-205) wherever the launch
// has M >= 8192 output pixels and the image tiles into 16 x 16 patches -- UNet batch >= 8 at the 64 x 64 / 32 x 32 (96 x 96 /
// 48 x 48) levels, i.e. BASELINE configs 2-4.  The 128-row HALO kernel of rounds 1-3 (gemm.hip) keeps every other launch.
//
// Why another kernel.  The 128 x {64,128} tiles run four waves in lockstep: wait -> barrier -> DMA issue -> ds_read -> MFMA, one
// barrier pair per K step, and measure 37-40 % of the MFMA peak.  This core follows the guide's 8-phase
// recipe (cdna_hip_programming.md section 5, T2-T5) re-derived for a HALO conv:
//   * M tile = one 16 x 16 pixel patch.  Per 64-channel chunk its 18 x 18 halo (324 rows x 128 B, zero padding from the buffer
//     bounds check) is DMA'd into LDS ONCE and serves all nine taps; weights stream per tap (BN x 128 B) through a 3-stage ring.
//     Per tap a CU moves BN x 128 B of weights + 4.6 KB of halo for 2 x 256 x BN x 64 FLOP: 19 B / clk at BN = 160.
//   * EIGHT waves = 4 (M) x 2 (N), wave tile 64 pixels x BN/2 channels on `v_mfma_f32_16x16x32_f16` (BN = 160: 4 x 5 tiles, 80
//     accumulator registers, 0.45 KB of LDS reads per MFMA).  The two waves that share a SIMD belong to different N halves and
//     run HALF A PHASE APART (the N = 1 half takes one extra barrier up front): while one half issues its MFMAs, the other
//     issues its `ds_read_b128`s -- the matrix pipe of a SIMD always has one wave feeding it and the LDS always has one half
//     reading.  `s_setprio 1` around the MFMA burst lets the arbiter prefer the computing wave (T5: only pays with such a
//     role split).
//   * A phase = one tap (K = 64): 2 x (4 + BN/32) `ds_read_b128` -> `s_waitcnt vmcnt(0)` -> barrier -> DMA issue (weights of tap
//     t + 2, one halo slice of the next chunk) -> 8 x BN/32 MFMAs -> barrier.  Every DMA batch gets a whole phase of flight time
//     (issued at the start of an MFMA burst, waited for at the end of the NEXT read burst), and at that wait it is the ONLY
//     batch outstanding, so the count is exact without per-wave bookkeeping.
//   * Hazards under the half-phase stagger (barriers numbered globally; N = 0 half: reads of phase p between barriers 2p-1 and
//     2p, MFMAs between 2p and 2p+1; N = 1 half one barrier later):
//       RAW  a wave waits for its own DMAs before ARRIVING at barrier X; data may be read after barrier X.  Batch issued in
//            phase p is waited for in the read burst of phase p+1 (before barriers 2p+2 / 2p+3), first read in phase p+2
//            (after barriers 2p+3 / 2p+4).
//       WAR  the last `ds_read` of a stage in phase p has returned when its wave passes `lgkmcnt(0)` behind barrier 2p (2p+1);
//            every wave arrives at barrier 2p+2 after that, and the refill is issued behind barrier 2p+2 (2p+3).
//   * LDS image and swizzle as in gemm.hip (lane-linear DMA destination, XOR on the SOURCE offset and again on the read):
//     halo pieces keyed on the halo column, weight tiles pre-swizzled on the host -- the packed weights of
//     ops.pack_conv_weight are used AS IS.  A 16 x 16 x 32 operand fragment is 16 consecutive rows x 4 k-quarters: conflict-free
//     under the (row >> 1) & 7 key (2-way on 2 of 16 lanes for the kx = 1 taps).
//   * ResBlock skip_connection (mdx_gemm_desc.skip_w) as extra dense K tiles after the taps, plain two-stage loop.
//   * Epilogue: accumulators (C^T: lane = pixel, 4 consecutive channels per register group) -> fp16 staging tile in LDS ->
//     16-byte coalesced stores with bias + per-sample time-embedding row + residual, and the GroupNorm column statistics of the
//     patch (mdx_gemm_desc.colstats_out; rows per block = 256).
#include "gemm_internal.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
constexpr int C8_TICKET_SLOTS = MDX_GEMM_WS_HEAD / 4;     // floats (= ticket slots) reserved at the head of the workspace

#ifndef MDX_C8_BATCHED_EPI
#define MDX_C8_BATCHED_EPI 1
#endif
constexpr int C8_NT = 512;
constexpr int C8_HINST = 41;                      // 324 halo rows / 8 rows per DMA instruction, rounded up
constexpr int C8_HALO_BYTES = C8_HINST * 1024;    // 41 984
constexpr int C8_RING = 2 * C8_HALO_BYTES;

template <int BN>
constexpr size_t c8_lds_bytes() {
    const size_t main_loop = (size_t)C8_RING + 3u * BN * 128u;
    const size_t epi = 256u * (BN + 8u) * 2u;
    return main_loop > epi ? main_loop : epi;
}

// TAPS = 9: the 3x3 conv.  TAPS = 4: the SUB-PIXEL form of Upsample's nearest-2x + 3x3 conv (openaimodel.py:57-60): for each of the
// four output parities (dy, dx) the conv is a 2 x 2 conv of the LOW-resolution tensor with pre-summed taps (ops.pack_subpixel_conv_weight):
// rows {y - 1 + dy, y + dy}, columns {x - 1 + dx, x + dx} -- 4 Cin instead of 9 Cin multiply-adds per output and a gather on the
// un-upsampled tensor.  An M tile is (low-resolution 16 x 16 patch, parity): the same halo, taps (dy + a, dx + b), weight rows
// parity * N + n, and an epilogue row map that interleaves the parities into the 2H x 2W output (pixel shuffle).
// VAR: bit 0 = the DMA instructions of a batch are issued BETWEEN the MFMAs of the burst (one per 16 x BN/2 row of MFMAs: an LDS-DMA
// issue hides in an MFMA gap) instead of in front of it -- the product form; bit 1 = no s_setprio; bit 2 = no stagger (both halves in
// step).  Measured on MI355X (tools/conv8p_bench.py --vars, profiles/r04_conv8p_variants.txt; 256 x 160 tiles, K = 2880 ... 11520):
// DMA in front 1183 / 1313 / 1369 TF/s -> between the MFMAs 1208 / 1342 / 1409 (+2-3 %); s_setprio neutral (+-0.3 %); WITHOUT the
// half-phase stagger 1044 / 1167 / 1213 (-12 %): the stagger is what the structure buys.
template <int BN, int PH, int TAPS, int VAR = 1>
__global__ __launch_bounds__(C8_NT) void conv8p_kernel(const GemmParams p) {
    mdx_kernarg_touch<sizeof(GemmParams)>();
    constexpr int NJ = BN / 32;              // 16-column MFMA tiles per wave (a wave owns BN / 2 columns)
    constexpr int B_BYTES = BN * 128;
    constexpr int BINST = BN / 8;            // DMA instructions per weight tile (8 rows of 128 B each)
    constexpr int BJ = (BINST + 7) / 8;      // ... per wave (the last round may be partial: wave-uniform guard)
    static_assert(BN % 32 == 0 && BN >= 64 && BN <= 192, "BN: 64 .. 192 in steps of 32 (LDS: two halos + three weight tiles)");
    constexpr int Q = 4 * NJ;                // 16-byte accumulator pieces per thread (tail-split partials)
    constexpr unsigned PART = (unsigned)Q * C8_NT * 16u;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3;                 // patch rows 4 wm .. 4 wm + 3
    const int wn = wave >> 2;                // column half; ALSO the stagger group (waves w and w + 4 share a SIMD)
    const int l15 = lane & 15, q = lane >> 4;

    // XCD-aware tile order (gemm.hip): block b runs on XCD b % 8; every XCD takes a contiguous run of tile ids, ids run fastest
    // along M, so the blocks of one XCD share their weight tiles in that XCD's L2
    // The first c8_full tiles (a multiple of 256) run whole; each of the remaining c8_rem tiles is split c8_split ways along the
    // 64-channel chunks so that the last, partly filled round still puts a block on every CU (the splits of one tile sit on one
    // XCD: their partials meet in that XCD's L2).
    int tile_id, split = 0, nsplit = 1;
    {
        const int bid = blockIdx.x;
        if (bid < p.c8_full) {
            tile_id = (bid & 7) * (p.c8_full >> 3) + (bid >> 3);
        } else {
            const int b2 = bid - p.c8_full;
            const int jx = b2 >> 3;
            const int r = (b2 & 7) + 8 * (jx / p.c8_split);
            if (r >= p.c8_rem) return;
            split = jx - (jx / p.c8_split) * p.c8_split;
            nsplit = p.c8_split;
            tile_id = p.c8_full + r;
        }
    }
    const int tile_n = tile_id / p.tiles_m;
    const int tile_m = tile_id - tile_n * p.tiles_m;
    const int n0 = tile_n * BN;
    const int pw = p.W >> 4, ph = p.H >> 4;
    const int patch = TAPS == 4 ? tile_m >> 2 : tile_m;
    const int par_dy = TAPS == 4 ? (tile_m >> 1) & 1 : 0, par_dx = TAPS == 4 ? tile_m & 1 : 0;
    const int pb = patch / (ph * pw);
    const int prem = patch - pb * (ph * pw);
    const int py0 = (prem / pw) * 16, px0 = (prem - (prem / pw) * pw) * 16;

    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(p.a, p.a_bytes);
    const __amdgpu_buffer_rsrc_t rs_a2 = make_rsrc(p.a2 ? p.a2 : p.a, p.a2 ? p.a2_bytes : p.a_bytes);
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, p.w_bytes);

    // ---- halo loader: piece id = qh * 8 + wave covers halo rows 8 id .. 8 id + 7 (lane / 8), 16-byte position lane % 8 holds
    // logical chunk (lane % 8) ^ ((halo column >> 1) & 7)
    const int lrow = lane >> 3, lchk = lane & 7;
    int hal_pk[6];           // (source pixel index << 3) | physical->logical chunk of this lane, or -1 outside the image
#pragma unroll
    for (int qh = 0; qh < 6; ++qh) {
        const int hp = (qh * 8 + wave) * 8 + lrow;
        const int hr = hp / 18, hc = hp - hr * 18;
        const int y = py0 - 1 + hr, x = px0 - 1 + hc;
        const bool ok = hp < 324 && y >= 0 && y < p.H && x >= 0 && x < p.W;
        hal_pk[qh] = ok ? ((((pb * p.H + y) * p.W + x) << 3) | (lchk ^ ((hc >> 1) & 7))) : -1;
    }
    auto dma_halo = [&](int qh, int chunk, int hb) {
        if (qh * 8 + wave >= C8_HINST) return;                 // wave-uniform
        int ci0 = chunk * 64;
        const bool second = ci0 >= p.c1;
        const unsigned cs2 = (unsigned)(second ? p.c2 : p.c1) * 2u;
        if (second) ci0 -= p.c1;
        const unsigned off = hal_pk[qh] >= 0 ? (unsigned)(hal_pk[qh] >> 3) * cs2 + (unsigned)(ci0 * 2) + (unsigned)((hal_pk[qh] & 7) << 4)
                                             : MDX_OOB;
        void* dst = smem + hb * C8_HALO_BYTES + (qh * 8 + wave) * 1024;
        if (second)
            dma16(rs_a2, dst, off);
        else
            dma16(rs_a, dst, off);
    };
    // ---- weight loader: piece idx = jb * 8 + wave covers tile rows 8 idx .. 8 idx + 7; the storage is tile-major and
    // pre-swizzled ([N / 64 panels][K / 64 tiles][64 rows][8 chunks][8]): every piece is 1 KiB of contiguous memory
    unsigned b_off[BJ];
#pragma unroll
    for (int jb = 0; jb < BJ; ++jb) {
        const int n = (TAPS == 4 ? (par_dy * 2 + par_dx) * p.N : 0) + n0 + (jb * 8 + wave) * 8 + lrow;
        b_off[jb] = (unsigned)(((size_t)(n >> 6) * p.kt64) * 8192 + ((n & 63) * 8 + lchk) * 16);
    }
    auto dma_b1 = [&](int jb, int kt, int stage) {
        if (jb * 8 + wave >= BINST) return;                    // wave-uniform
        dma16(rs_w, smem + C8_RING + stage * B_BYTES + (jb * 8 + wave) * 1024, b_off[jb] + (unsigned)kt * 8192u);
    };
    auto dma_b = [&](int kt, int stage) {
#pragma unroll
        for (int jb = 0; jb < BJ; ++jb) dma_b1(jb, kt, stage);
    };

    f32x4v acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};

    // ---- fragment addresses.  A: lane (l15, q) of row tile i reads halo pixel (4 wm + i + ky) * 18 + l15 + kx, logical chunk
    // 4 s + q.  B: row wn * BN/2 + 16 j + l15 of the weight tile, same chunk; its key (row >> 1) & 7 = (l15 >> 1) & 7.
    const int a_lane = ((wm * 4) * 18 + l15) * 128;
    const int b_lane = C8_RING + (wn * (BN / 2) + l15) * 128;
    int axor[3][2], bxor[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        bxor[s] = ((s * 4 + q) ^ ((l15 >> 1) & 7)) << 4;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) axor[kx][s] = ((s * 4 + q) ^ (((l15 + kx + (TAPS == 4 ? par_dx : 0)) >> 1) & 7)) << 4;
    }
    // (TAPS = 4: axor[b][s] is keyed on the halo column offset dx + b, b = 0 | 1; entry [2] is unused)

    const int nchunks_all = p.cin >> 6;
    const int c_begin = nsplit > 1 ? split * p.c8_cps : 0;
    const int c_end = nsplit > 1 ? min(nchunks_all, c_begin + p.c8_cps) : nchunks_all;
    const int nt = c_end * TAPS;             // one past the last K tile of this block

    // ---- prologue: the first chunk's halo, the weights of its taps 0 and 1
#pragma unroll
    for (int qh = 0; qh < 6; ++qh) dma_halo(qh, c_begin, 0);
    dma_b(c_begin * TAPS, 0);
    dma_b(c_begin * TAPS + 1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wn == 1 && !(VAR & 4)) __builtin_amdgcn_s_barrier();          // the N = 1 half runs one barrier (half a phase) behind

    int t = c_begin * TAPS;
    int rd = 0;                              // ring stage of K tile t (TAPS = 9: tap % 3, static; TAPS = 4: rotates across chunks)
    const int par_off = (par_dy * 18 + par_dx) * 128;
    for (int c = c_begin; c < c_end; ++c) {
        const int hb = (c - c_begin) & 1;
        const bool more = c + 1 < c_end;
        const char* abase = smem + hb * C8_HALO_BYTES + a_lane + (TAPS == 4 ? par_off : 0);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap, ++t) {
            const int ky = TAPS == 9 ? tap / 3 : tap >> 1, kx = TAPS == 9 ? tap - ky * 3 : tap & 1;
            const char* ap = abase + (ky * 18 + kx) * 128;
            const int st_r = TAPS == 9 ? tap % 3 : rd;
            const int st_w = TAPS == 9 ? (tap + 2) % 3 : (rd == 0 ? 2 : rd - 1);
            const char* bp = smem + b_lane + st_r * B_BYTES;
            // PH = 2: one phase per tap (both 32-deep k-steps read before the barrier); PH = 1: one phase per k-step (half the
            // fragment registers, twice the barriers).  The DMA batch of the tap is issued / waited for in its first phase.
#pragma unroll
            for (int sub = 0; sub < 2 / PH; ++sub) {
                f16x8 af[PH][4], bf[PH][NJ];
#pragma unroll
                for (int u = 0; u < PH; ++u) {
                    const int s = sub * PH + u;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) bf[u][j] = *reinterpret_cast<const f16x8*>(bp + j * 2048 + bxor[s]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) af[u][i] = *reinterpret_cast<const f16x8*>(ap + i * (18 * 128) + axor[kx][s]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (sub == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the batch issued one tap ago (this wave's share)
                __builtin_amdgcn_s_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                // the k-th DMA instruction of this tap's batch: weights of tap t + 2, then the halo slice(s) of the next chunk
                auto issue = [&](int k) {
                    if (k < BJ) {
                        if (t + 2 < nt) dma_b1(k, t + 2, st_w);
                    } else if (TAPS == 9) {
                        if (k == BJ && more && tap < 6) dma_halo(tap, c + 1, hb ^ 1);
                    } else if (k < BJ + 2) {
                        if (more && tap < 3) dma_halo(2 * tap + (k - BJ), c + 1, hb ^ 1);
                    }
                };
                if (sub == 0 && !(VAR & 1)) {
                    if (t + 2 < nt) dma_b(t + 2, st_w);
                    if constexpr (TAPS == 9) {
                        if (more && tap < 6) dma_halo(tap, c + 1, hb ^ 1);
                    } else {
                        if (more && tap < 3) {      // six halo slices per wave over the first three of the four taps
                            dma_halo(2 * tap, c + 1, hb ^ 1);
                            dma_halo(2 * tap + 1, c + 1, hb ^ 1);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (!(VAR & 2)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int u = 0; u < PH; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[u][j], af[u][i], acc[i][j], 0, 0, 0);
                        if constexpr ((VAR & 1) != 0) {
                            if (sub == 0 && u * 4 + i < BJ + 2) {
                                __builtin_amdgcn_sched_barrier(0);
                                issue(u * 4 + i);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
                if constexpr ((VAR & 1) != 0 && PH * 4 < BJ + 2) {      // more DMA instructions than MFMA rows in a phase: the rest behind the burst
                    if (sub == 0) {
#pragma unroll
                        for (int k = PH * 4; k < BJ + 2; ++k) issue(k);
                    }
                }
                if (!(VAR & 2)) __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
            }
            if constexpr (TAPS != 9) rd = rd == 2 ? 0 : rd + 1;
        }
    }
    if (wn == 0 && !(VAR & 4)) __builtin_amdgcn_s_barrier();          // both halves level again; every wave is done with the halos and the ring

    if (TAPS == 9 && p.skip_w && split == nsplit - 1) {      // (block-uniform) the last split -- the one with the fewest chunks -- takes the skip tiles
        // ---- ResBlock skip_connection (openaimodel.py:174, 201-205): conv1x1 over the block's RAW input as extra dense K tiles
        // of the same accumulators.  A tiles = the patch's 256 pixels x 64 channels (32 KB, in the halo area), weight tiles in
        // the ring area; two stages, one barrier per tile, all waves in step.
        const __amdgpu_buffer_rsrc_t rs_s1 = make_rsrc(p.skip_a, p.skip_a_bytes);
        const __amdgpu_buffer_rsrc_t rs_s2 = make_rsrc(p.skip_a2 ? p.skip_a2 : p.skip_a, p.skip_a2 ? p.skip_a2_bytes : p.skip_a_bytes);
        const __amdgpu_buffer_rsrc_t rs_sw = make_rsrc(p.skip_w, p.skip_w_bytes);
        int s_pix[4];
        unsigned s_cb[4];
#pragma unroll
        for (int qa = 0; qa < 4; ++qa) {
            const int r = (qa * 8 + wave) * 8 + lrow;
            s_pix[qa] = (pb * p.H + py0 + (r >> 4)) * p.W + px0 + (r & 15);
            s_cb[qa] = (unsigned)((lchk ^ ((r >> 1) & 7)) * 16);
        }
        unsigned sb_off[BJ];
#pragma unroll
        for (int jb = 0; jb < BJ; ++jb) {
            const int n = n0 + (jb * 8 + wave) * 8 + lrow;
            sb_off[jb] = (unsigned)(((size_t)(n >> 6) * p.skip_kt) * 8192 + ((n & 63) * 8 + lchk) * 16);
        }
        auto stage_skip = [&](int kt, int st) {
            int ci0 = kt * 64;
            const bool second = ci0 >= p.skip_c1;
            const unsigned cs2 = (unsigned)(second ? p.skip_c2 : p.skip_c1) * 2u;
            if (second) ci0 -= p.skip_c1;
#pragma unroll
            for (int qa = 0; qa < 4; ++qa) {
                const unsigned off = (unsigned)s_pix[qa] * cs2 + (unsigned)(ci0 * 2) + s_cb[qa];
                void* dst = smem + st * C8_HALO_BYTES + (qa * 8 + wave) * 1024;
                if (second)
                    dma16(rs_s2, dst, off);
                else
                    dma16(rs_s1, dst, off);
            }
#pragma unroll
            for (int jb = 0; jb < BJ; ++jb) {
                if (jb * 8 + wave >= BINST) continue;
                dma16(rs_sw, smem + C8_RING + st * B_BYTES + (jb * 8 + wave) * 1024, sb_off[jb] + (unsigned)kt * 8192u);
            }
        };
        const int sa_lane = (wm * 64 + l15) * 128;
        stage_skip(0, 0);
        for (int kt = 0; kt < p.skip_kt; ++kt) {
            const int st = kt & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();        // tile kt has landed for everyone; everyone is done reading stage st ^ 1
            if (kt + 1 < p.skip_kt) stage_skip(kt + 1, st ^ 1);
            const char* ap = smem + st * C8_HALO_BYTES + sa_lane;
            const char* bp = smem + b_lane + st * B_BYTES;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f16x8 af[4], bf[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) bf[j] = *reinterpret_cast<const f16x8*>(bp + j * 2048 + bxor[s]);
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const f16x8*>(ap + i * 2048 + bxor[s]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
            }
        }
    }
    __syncthreads();     // (nothing is in flight: plain barrier) the staging tile overwrites the halos / ring

    if (nsplit > 1) {
        // ---- tail split: park the fp32 accumulators in register layout ([tile][split][piece][thread] 16-byte pieces), take a
        // ticket; the block that draws the last one sums all partials in split order (its own included: deterministic) and goes on
        // to the epilogue.  Hand-off as in gemm.hip splitk_last_block_reduce (write-through stores, drained by every wave, one
        // relaxed agent-scope ticket, sc1 loads on the reading side; no placement assumption).
        const int tl = tile_id - p.c8_full;
        char* base = reinterpret_cast<char*>(p.ws + C8_TICKET_SLOTS) + (size_t)tl * nsplit * PART;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(base, (unsigned)nsplit * PART);
        const unsigned toff = (unsigned)tid * 16u;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, acc[i][j]), rs,
                                                       (unsigned)split * PART + (unsigned)(i * NJ + j) * (C8_NT * 16u) + toff, 0, /*sc1*/ 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* flag = reinterpret_cast<int*>(smem);
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(p.tickets + tl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old >= (unsigned)nsplit) __builtin_trap();
            *flag = old == (unsigned)nsplit - 1u;
        }
        __syncthreads();
        const bool last = *flag != 0;
        __syncthreads();
        if (!last) return;
        if (tid == 0) __hip_atomic_store(p.tickets + tl, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
        constexpr int QH = Q / 2;            // two batches of loads per partial: Q / 2 x 4 registers in flight beside the accumulators
        for (int z = 0; z < nsplit; ++z) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                u32x4v v[QH];
#pragma unroll
                for (int qq = 0; qq < QH; ++qq)
                    v[qq] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)z * PART + (unsigned)(h * QH + qq) * (C8_NT * 16u) + toff, 0, 16);
#pragma unroll
                for (int qq = 0; qq < QH; ++qq) acc[(h * QH + qq) / NJ][(h * QH + qq) % NJ] += __builtin_bit_cast(f32x4v, v[qq]);
            }
        }
    }

    // bias + this sample's time-embedding row: one value per column for the whole patch (epilogue thread -> 8 columns at
    // n0 + (tid % CPR) * 8); loaded here, used after the staging barrier
    constexpr int CPR = BN / 8;               // 16-byte chunks per staged row
    constexpr int RPP = C8_NT / CPR;          // rows per store pass (threads RPP * CPR .. 511 idle in the store loop)
    const int e_chunk = tid % CPR, e_r0 = tid / CPR;
    const int e_n = n0 + e_chunk * 8;
    const bool e_act = e_r0 < RPP && e_n < p.N;
    float bb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bb[e] = 0.f;
    if (e_act) {
        if (p.bias) {
            const float4* b4 = reinterpret_cast<const float4*>(p.bias + e_n);
            const float4 x0 = b4[0], x1 = b4[1];
            bb[0] = x0.x; bb[1] = x0.y; bb[2] = x0.z; bb[3] = x0.w; bb[4] = x1.x; bb[5] = x1.y; bb[6] = x1.z; bb[7] = x1.w;
        }
        if (p.rowbias) {
            const float4* r4 = reinterpret_cast<const float4*>(p.rowbias + (size_t)pb * p.rowbias_ld + e_n);
            const float4 x0 = r4[0], x1 = r4[1];
            bb[0] += x0.x; bb[1] += x0.y; bb[2] += x0.z; bb[3] += x0.w; bb[4] += x1.x; bb[5] += x1.y; bb[6] += x1.z; bb[7] += x1.w;
        }
    }

    // ---- epilogue.  C^T accumulators: lane (l15, q) of tile (i, j) holds pixel row 64 wm + 16 i + l15, channels
    // wn BN/2 + 16 j + 4 q .. + 3.
    constexpr int SLD = BN + 8;
    f16* stg = reinterpret_cast<f16*>(smem);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int m_l = wm * 64 + i * 16 + l15;
            const int n_l = wn * (BN / 2) + j * 16 + 4 * q;
            *reinterpret_cast<f16x4*>(&stg[m_l * SLD + n_l]) = cvt4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
    __syncthreads();
    const bool colstats = p.colstats_out != nullptr;
    float cs[8], cq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[e] = cq[e] = 0.f;
    // output row of patch pixel (py, px): plain = the pixel itself; sub-pixel = pixel (2 (py0 + py) + dy, 2 (px0 + px) + dx) of the 2H x 2W image
    const int rw = TAPS == 4 ? 2 * p.W : p.W;                  // output row pitch in pixels
    const int rsx = TAPS == 4 ? 2 : 1;                         // pixel step along x
    const int mbase = TAPS == 4 ? (pb * 2 * p.H + 2 * py0 + par_dy) * rw + 2 * px0 + par_dx : (pb * p.H + py0) * p.W + px0;
    const int rsy = TAPS == 4 ? 2 * rw : rw;                   // pixel step along y
    constexpr int NPASS = (256 + RPP - 1) / RPP;
#if MDX_C8_BATCHED_EPI
    // (round 6) the store loop in BATCHES, as in the lean dense kernel (gemm_epilogue EMODE 1): the residual rows of ALL passes are
    // requested first, then the staged rows, then the arithmetic, then the stores back to back -- a pass no longer waits for its own
    // LDS read and residual load one after the other (a block of this core has its CU to itself: nothing else hides them).  Same
    // operations on the same values: same bits.
    if (e_act) {
        f16x8 rv[NPASS], sv[NPASS];
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int row = e_r0 + pass * RPP;
            rv[pass] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (p.residual && row < 256)
                rv[pass] = *reinterpret_cast<const f16x8*>(p.residual + (size_t)(mbase + (row >> 4) * rsy + (row & 15) * rsx) * p.residual_ld + e_n);
        }
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int row = e_r0 + pass * RPP;
            sv[pass] = row < 256 ? *reinterpret_cast<const f16x8*>(&stg[row * SLD + e_chunk * 8]) : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int row = e_r0 + pass * RPP;
            if (row < 256) {
                const int m = mbase + (row >> 4) * rsy + (row & 15) * rsx;
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float f = (float)sv[pass][e] + bb[e];
                    if (p.residual) f += (float)rv[pass][e];
                    o[e] = (f16)f;
                }
                *reinterpret_cast<f16x8*>(p.out + (size_t)m * p.out_ld + e_n) = o;
                if (colstats) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float tv = (float)o[e];    // statistics of the fp16 values actually stored
                        cs[e] += tv;
                        cq[e] += tv * tv;
                    }
                }
            }
        }
    }
#else
    if (e_act) {
        f16x8 res_n = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (p.residual && e_r0 < 256)
            res_n = *reinterpret_cast<const f16x8*>(p.residual + (size_t)(mbase + (e_r0 >> 4) * rsy + (e_r0 & 15) * rsx) * p.residual_ld + e_n);
#pragma unroll 4
        for (int pass = 0; pass < NPASS; ++pass) {
            const int row = e_r0 + pass * RPP;
            if (row >= 256) break;
            const int m = mbase + (row >> 4) * rsy + (row & 15) * rsx;
            const f16x8 res = res_n;
            const int row2 = row + RPP;
            if (p.residual && row2 < 256)
                res_n = *reinterpret_cast<const f16x8*>(p.residual + (size_t)(mbase + (row2 >> 4) * rsy + (row2 & 15) * rsx) * p.residual_ld + e_n);
            const f16x8 v = *reinterpret_cast<const f16x8*>(&stg[row * SLD + e_chunk * 8]);
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = (float)v[e] + bb[e];
                if (p.residual) f += (float)res[e];
                o[e] = (f16)f;
            }
            *reinterpret_cast<f16x8*>(p.out + (size_t)m * p.out_ld + e_n) = o;
            if (colstats) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float tv = (float)o[e];    // statistics of the fp16 values actually stored
                    cs[e] += tv;
                    cq[e] += tv * tv;
                }
            }
        }
    }
#endif
    if (colstats) {      // (block-uniform) fold the RPP row lanes of every column in a fixed order: deterministic
        __syncthreads();                                  // every thread is done reading the staged tile
        float* part = reinterpret_cast<float*>(smem);     // [RPP][BN][2]
        if (e_r0 < RPP) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                part[((size_t)e_r0 * BN + e_chunk * 8 + e) * 2] = cs[e];
                part[((size_t)e_r0 * BN + e_chunk * 8 + e) * 2 + 1] = cq[e];
            }
        }
        __syncthreads();
        if (tid < BN * 2) {
            const int cc = tid >> 1;
            float a = 0.f;
            for (int r = 0; r < RPP; ++r) a += part[((size_t)r * BN + cc) * 2 + (tid & 1)];
            if (n0 + cc < p.N) p.colstats_out[((size_t)tile_m * p.N + n0 + cc) * 2 + (tid & 1)] = a;
        }
    }
}

template <int BN, int PH, int TAPS = 9, int VAR = 1>
void c8_launch(const GemmParams& p, dim3 grid, hipStream_t st) {
    constexpr size_t lds = c8_lds_bytes<BN>();
    static MdxPerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv8p_kernel<BN, PH, TAPS, VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL((conv8p_kernel<BN, PH, TAPS, VAR>), grid, dim3(C8_NT), lds, st, p);
}

}  // namespace

// Shapes the core handles: 3x3 / stride 1 / pad 1, channel counts in whole 64-channel chunks (two-source concat: the first
// source too), images tiling into 16 x 16 patches, plain row-major output (bias / time-embedding row / residual / column statistics /
// fused skip 1x1), tile-major weights.
bool mdx_conv8p_eligible(const GemmParams& p) {
    if (!(p.ksize == 3 && p.stride == 1 && p.pad == 1)) return false;
    if (p.upsample && (!p.w_sub || p.c2 > 0 || p.skip_w || p.N % 64 != 0)) return false;      // nearest-2x + conv: the sub-pixel form only
    if (p.cin % 64 != 0 || (p.c2 > 0 && p.c1 % 64 != 0) || p.cin < 64) return false;
    if (p.H % 16 != 0 || p.W % 16 != 0) return false;
    if (p.out_mode != MDX_OUT_ROWMAJOR || p.epilogue != MDX_EPI_NONE || p.n_split || p.ln_stats || p.stats_out || p.out_bs ||
        p.gn_cs)
        return false;
    if (p.N % 8 != 0 || p.N < 64) return false;
    return true;
}

// How many ways the tiles of a partly filled round are split: as many as still puts <= 256 blocks on the chip, at most 4 (the last
// arriver reads every partial: ~3.5 us per 160 KB), no empty split, and nothing when the round is >= 3/4 full anyway.
static int c8_tail_split(int rem, int chunks) {
    if (rem == 0 || rem >= 192 || chunks < 2) return 1;
    int s = 256 / rem;
    if (s > 4) s = 4;
    if (s > chunks) s = chunks;
    while (s > 1 && (chunks + s - 1) / s * (s - 1) >= chunks) --s;      // every split owns >= 1 chunk
    return s;
}

// Time model of a launch with N tile `bn`, microseconds (fitted to tools/conv8p_bench.py on MI355X, profiles/r04_conv8p_bench.txt):
// a tap of a 256 x 160 tile costs 1.06 us with every CU busy; 128- and 192-column tiles run at the same rate per FLOP, 96- / 64-column
// tiles at 0.88 / 0.72 of it (halo DMA and LDS reads amortised over fewer MFMAs); rounds of 256 tiles; the partly filled last round
// is split along K and costs its share of the K loop plus 3.5 us per partial the last arriver reads.
static double c8_time_us(const GemmParams& p, int bn) {
    const int taps = p.upsample ? 4 : 9;
    const int patches = p.B * (p.H >> 4) * (p.W >> 4) * (p.upsample ? 4 : 1);
    const int tn = (p.N + bn - 1) / bn;
    const long tiles = (long)patches * tn;
    const int chunks = p.cin >> 6;
    const int skip_kt = p.skip_w ? p.skip_kt : 0;
    const double eff = bn >= 128 ? 1.0 : (bn == 96 ? 0.88 : 0.72);
    const double tap = 1.06 * ((double)bn / 160.0) / eff;
    const double epi = 3.0;
    const double whole = (chunks * taps + skip_kt) * tap + epi;
    const long full = tiles / 256;
    const int rem = (int)(tiles % 256);
    double t = (double)full * whole;
    if (rem) {
        const int s = c8_tail_split(rem, chunks);
        if (s == 1)
            t += whole;
        else
            t += (((chunks + s - 1) / s) * taps + skip_kt) * tap + epi + 3.5 * (s - 1) * ((double)bn / 160.0);
    }
    return t;
}

int mdx_conv8p_pick_bn(const GemmParams& p, int bn_hint) {
    if (bn_hint == 64 || bn_hint == 96 || bn_hint == 128 || bn_hint == 160 || bn_hint == 192) return bn_hint;
    static const int cand[5] = {160, 192, 128, 96, 64};
    int best = 160;
    double best_t = 1e30;
    for (int bn : cand) {
        if (bn == 64 && p.N > 64) continue;
        if (p.upsample && p.N % bn != 0) continue;       // sub-pixel: an N tile stays inside one parity's weight rows
        const double t = c8_time_us(p, bn);
        if (t < best_t * 0.985) {        // later candidates must win clearly
            best_t = t;
            best = bn;
        }
    }
    return best;
}

// Tiles the launch would have with its best N tile (the automatic route wants >= 128: below that even a 4-way tail split leaves
// CUs idle and the 128-row tiles with their own split-K win -- measured at UNet batch 2).
int mdx_conv8p_tiles(const GemmParams& p) {
    int bn = mdx_conv8p_pick_bn(p, 0);
    if (bn < 128 && !p.upsample) bn = 128;       // counted in tiles of at least 128 columns: 128 narrow tiles are half the work of 128 wide ones
    return p.B * (p.H >> 4) * (p.W >> 4) * (p.upsample ? 4 : 1) * ((p.N + bn - 1) / bn);
}

size_t mdx_conv8p_plan(GemmParams& p, int bn, size_t workspace_bytes, bool have_workspace, bool query_only) {
    p.c8_sub = p.upsample ? 1 : 0;
    p.tiles_m = p.B * (p.H >> 4) * (p.W >> 4) * (p.c8_sub ? 4 : 1);
    p.tiles_n = (p.N + bn - 1) / bn;
    const int ntiles = p.tiles_m * p.tiles_n;
    p.nsplit = 1;
    p.c8_full = ntiles;
    p.c8_rem = 0;
    p.c8_split = 1;
    p.c8_cps = p.cin >> 6;
    p.tiles_per_xcd = (ntiles + 7) / 8;
    const int chunks = p.cin >> 6;
    const int rem = ntiles % 256;
    int s = c8_tail_split(rem, chunks);
    const size_t part = (size_t)(4 * (bn / 32)) * C8_NT * 16;
    const size_t need = s > 1 ? (size_t)MDX_GEMM_WS_HEAD + (size_t)rem * s * part : 0;
    if (s > 1 && !query_only && (!have_workspace || workspace_bytes < need)) s = 1;      // no room for the partials: run the tail unsplit
    if (s == 1) {
        if (ntiles % 8) {                 // the XCD-contiguous order of the whole tiles wants a multiple of 8: the odd tiles go to the tail, unsplit
            p.c8_full = ntiles & ~7;
            p.c8_rem = ntiles - p.c8_full;
        }
        return 0;
    }
    p.c8_full = ntiles - rem;
    p.c8_rem = rem;
    p.c8_split = s;
    p.c8_cps = (chunks + s - 1) / s;
    return need;
}

int mdx_conv8p_launch(const GemmParams& pin, int bn, hipStream_t st) {
    GemmParams p = pin;
    const dim3 grid(p.c8_full + ((p.c8_rem + 7) / 8) * 8 * p.c8_split);
    if (p.c8_sub) {       // sub-pixel form: the per-parity weights, four taps per chunk
        p.w = p.w_sub;
        p.w_bytes = p.w_sub_bytes;
        p.kt64 = (p.cin >> 6) * 4;
        switch (bn) {
            case 64: c8_launch<64, 2, 4>(p, grid, st); break;
            case 96: c8_launch<96, 2, 4>(p, grid, st); break;
            case 128: c8_launch<128, 2, 4>(p, grid, st); break;
            case 160: c8_launch<160, 2, 4>(p, grid, st); break;
            case 192: c8_launch<192, 1, 4>(p, grid, st); break;
            default:
                mdx_set_error("mdx_gemm_f16: the sub-pixel conv8p form has no %d-column tile", bn);
                return MDX_E_INVALID;
        }
        return MDX_OK;
    }
    switch (bn) {
        case 64: c8_launch<64, 2>(p, grid, st); break;
        case 96: c8_launch<96, 2>(p, grid, st); break;
        case 128: c8_launch<128, 2>(p, grid, st); break;
        case 160:
            if (p.st_hint == 9) c8_launch<160, 1>(p, grid, st);
            else switch (mdx_opt(MDX_OPT_GEMM_CONV8P_VAR)) {      // experiment forms (VAR above); 0 = the product
                case 1: c8_launch<160, 2, 9, 0>(p, grid, st); break;      // DMA issue in front of the MFMA burst
                case 2: c8_launch<160, 2, 9, 3>(p, grid, st); break;      // no s_setprio
                case 4: c8_launch<160, 2, 9, 5>(p, grid, st); break;      // no stagger
                default: c8_launch<160, 2>(p, grid, st); break;
            }
            break;
        case 192: c8_launch<192, 1>(p, grid, st); break;      // 96 accumulator + 80 fragment registers do not fit two waves per SIMD
        default:
            mdx_set_error("mdx_gemm_f16: conv8p has no %d-column tile", bn);
            return MDX_E_INVALID;
    }
    return MDX_OK;
}
