// Lean instantiation of the implicit-GEMM kernel for DENSE row-major launches (round 6).
//
//   out[m][n] = sum_k A[m][k] * W[n][k]      nn.Dense / 1x1 conv over tokens (attention.py:44, 66, 108-112, 212, 231; unet.py:267-310)
//
// Two thirds of the launches of a UNet evaluation are dense (ksize 1, stride 1, one source), and at UNet batch 2 every one of them
// runs one block per CU for 8-15 us -- of which, in the generic gemm_kernel (gemm.hip), 1.0-1.9 us went to a prologue written for
// convolutions (three integer divisions for the tile decode, tap masks, the (b, y, x) split of every loader row: ~700 scalar / vector
// instructions in front of the first DMA, issued by a wave that has its SIMD to itself at one instruction per ~4 clocks) and 1.1-2.6 us
// to an epilogue whose global loads (residual rows, the LayerNorm-fold row statistics and S[n]) each start a dependent round trip
// that nothing hides (profiles/r05_gemm_trace_kloop.txt, DESIGN.md section 8 items 1-2 of round 5).  This kernel is the same tile
// program -- same LDS image, same K order per output, same MFMA operand order, same epilogue arithmetic (gemm_epilogue of
// gemm_internal.h): results are BIT-IDENTICAL to the generic kernel's (tests/test_kernels_gpu.py) -- with
//   * a division-free prologue (2-D grid for the SPREAD order, multiply-high for the tile decode) that issues the first NS-1 K tiles'
//     DMAs ~80 instructions after the kernel-argument wait;
//   * everything the epilogue reads from global memory requested right behind those DMAs -- bias, the residual rows of the
//     thread's store passes, the LayerNorm-fold partials of the thread's row (up to 24 of them, held in registers: the adds run
//     after the K loop in the partials' order, so the statistics keep their bits) and S[n] -- so that the epilogue of a lone block is
//     staging + arithmetic + stores.
// Launch forms covered: row-major output, unsplit or in-kernel split-K reduce (tickets), bias / residual / GEGLU / row statistics /
// column statistics / LayerNorm-fold consumer / the q|k row-major + V^T split store.  Everything else (per-sample row bias, out_bs,
// GELU activations, transposed output, slab split-K, the GroupNorm-on-A form, eight waves) stays on the generic kernel.
#include "mdx_common.h"
#include "gemm_internal.h"

namespace {

// 1 = the batched store loops of gemm_epilogue<..., LEAN> (round 6); 0 = the generic per-pass loops (A/B builds: make variant)
#ifndef MDX_LEAN_EPI
#define MDX_LEAN_EPI 1
#endif
constexpr bool LEAN_EPI = MDX_LEAN_EPI != 0;
constexpr int LNR_MAX = 24;      // LayerNorm-fold partials per row a thread holds across the K loop (K <= 1536); more: the epilogue folds
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// Prefetch level PF -- how much of what the epilogue reads is requested before the K loop (measured in situ at UNet batch 2,
// profiles/r06_lean_dense_ab.txt; every level leaves the bits alone):
//   0  bias
//   1  + one dword of every line of the LayerNorm-fold consumer's row statistics and of S[n] (gemm_kernel's round-5 "touch")
//   2  + the residual rows of the thread's first store passes and S[n] itself, in registers
//   3  + the LayerNorm partials of the thread's row in registers (up to 24: ~50 more VGPRs, two blocks per CU) -- the full form;
//      a VMEM instruction costs a lone wave ~50 clocks to issue, and 29 of them in front of the first barrier cost the launch more
//      than the round trips they hide
// Cross-attention over a short cached context as the epilogue of the query projection (mdx_gemm_desc.xattn_k; BN = 64 = head dim): the
// tile's fp16 q rows are in LDS at `stg` ([BM][72]), the head's K / V^T tiles (at most two of 64 keys; attn_kernel's LDS images) at `xs`.
// Every wave takes 32 query rows through attn_kernel's tile program -- S^T = K Q^T, lazy-reference online softmax, O^T += V^T P^T, the
// masked form on a ragged last tile -- normalises, stages its 32 x 64 output rows over its own q rows and stores them: the same
// operations on the same values in the same order as mdx_attention_f16 on the stored q (bit-identical).
template <int BM>
__device__ __forceinline__ void xattn_tile_epilogue(const GemmParams& p, char* smem, const char* xs, const int m0, const int n0) {
    constexpr int SLD = 72;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (wave * 32 >= BM) return;
    const int hi = lane >> 5, l31 = lane & 31;
    f16* stg = reinterpret_cast<f16*>(smem);
    f16x8 qf[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) qf[s4] = *reinterpret_cast<const f16x8*>(&stg[(wave * 32 + l31) * SLD + 16 * s4 + 8 * hi]);
    f32x16 acc_o[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const int vswz = (lane >> 1) & 7;
    const int krow_l = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);      // attn_kernel's key permutation
    const int ntile = (p.xa_len + 63) >> 6;
    auto tile = [&](const int t, auto mask_c) {
        constexpr bool MASK = decltype(mask_c)::value;
        const char* sk = xs + t * 16384;
        const char* sv = sk + 8192;
        f32x16 acc_s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_s[kt][r] = 0.f;
            const int krow = kt * 32 + krow_l;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const f16x8 kf = *reinterpret_cast<const f16x8*>(sk + krow * 128 + (((2 * s4 + hi) ^ ((krow >> 1) & 7)) << 4));
                acc_s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s4], acc_s[kt], 0, 0, 0);
            }
        }
        if constexpr (MASK) {
            const int key0 = t * 64, kmax = p.xa_len - 1;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + kt * 32 + (r & 3) + 4 * ((r >> 2) & 1) + 8 * hi + 16 * (r >> 3);
                    if (key > kmax) acc_s[kt][r] = -INFINITY;
                }
        }
        float mx = acc_s[0][0];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc_s[kt][r]);
        {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        const float m_cand = fmaxf(m_run, mx);
        if (__builtin_amdgcn_ballot_w64((m_cand - m_run) * p.xa_scale_log2 > 8.0f)) {
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_cand) * p.xa_scale_log2);
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_o[d][r] *= alpha;
            m_run = m_cand;
        }
        const float mb = m_run * p.xa_scale_log2;
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 sc2 = {p.xa_scale_log2, p.xa_scale_log2}, nmb2 = {-mb, -mb};
        f32x2 psum2 = {0.f, 0.f};
        f16x8 pf[4];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 x2 = {acc_s[kt][r], acc_s[kt][r + 1]};
                const f32x2 a2 = __builtin_elementwise_fma(x2, sc2, nmb2);
                const f32x2 pv2 = {__builtin_amdgcn_exp2f(a2.x), __builtin_amdgcn_exp2f(a2.y)};
                psum2 += pv2;
                pf[kt * 2 + (r >> 3)][r & 7] = (f16)pv2.x;
                pf[kt * 2 + (r >> 3)][(r & 7) + 1] = (f16)pv2.y;
            }
        l_run += psum2.x + psum2.y;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const f16x8 vf = *reinterpret_cast<const f16x8*>(sv + (d * 32 + l31) * 128 + (((2 * c + hi) ^ vswz) << 4));
                acc_o[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[c], acc_o[d], 0, 0, 0);
            }
    };
    for (int t = 0; t < ntile; ++t) {
        if ((t + 1) * 64 > p.xa_len) tile(t, std::true_type{}); else tile(t, std::false_type{});
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    f16* og = stg + wave * 32 * SLD;      // this wave's own q rows: its fragments are in registers
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (f16)(acc_o[d][4 * g + e] * inv);
            *reinterpret_cast<f16x4*>(&og[l31 * SLD + d * 32 + 8 * g + 4 * hi]) = v;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (wave-local hand-over through LDS)
    for (int idx = lane; idx < 32 * 8; idx += 64) {
        const int row = idx >> 3, chunk = idx & 7;
        const int m = m0 + wave * 32 + row;
        if (m < p.M) {
            const f16x8 v = *reinterpret_cast<const f16x8*>(&og[row * SLD + chunk * 8]);
            *reinterpret_cast<f16x8*>(p.out + (size_t)m * p.out_ld + n0 + chunk * 8) = v;
        }
    }
}

template <int BM, int BN, int NS, int PF, bool XA = false>
__global__ __launch_bounds__(256, 2) void dense_kernel(const GemmParams p) {
    mdx_kernarg_touch<sizeof(GemmParams)>();
    constexpr int NW = 4;
    constexpr int WROWS = BM / 2;            // rows per wave row (waves are 2 x 2)
    constexpr int TM = WROWS / 32;
    constexpr int TN = BN / 64;
    constexpr int ROWB = 128;                // bytes per LDS row
    constexpr int KS = 4;                    // MFMA k-steps per K tile
    constexpr int A_BYTES = BM * ROWB;
    constexpr int B_BYTES = BN * ROWB;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int AJ = BM / 8 / NW;          // A-tile DMA instructions per wave (8 rows each)
    constexpr int BJ = BN / 8 / NW;
    constexpr int LPT = AJ + BJ;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int hi = lane >> 5;
    const int l31 = lane & 31;

    // ---- tile decode without a division.  SPREAD order (one row of M tiles): grid (tiles, splits) -- x runs fastest, so consecutive
    // workgroups (consecutive XCDs) take consecutive tiles exactly as gemm_kernel's 1-D grid does with its % and /.  Otherwise the
    // XCD-contiguous order: every XCD a contiguous run of tile ids.
    const int ntiles = p.tiles_m * p.tiles_n;
    const int tile_id = p.spread ? (int)blockIdx.x : (int)((blockIdx.x & 7) * p.tiles_per_xcd + (blockIdx.x >> 3));
    if (tile_id >= ntiles) return;
    trace_mark(p, 0);
    int tile_m, tile_n;      // (inv_* = ceil(2^32 / d), 0 for d == 1: the quotient is then the id itself)
    if (p.n_fastest) {
        tile_m = p.inv_tiles_n ? (int)__umulhi((unsigned)tile_id, p.inv_tiles_n) : tile_id;
        tile_n = tile_id - tile_m * p.tiles_n;
    } else {
        tile_n = p.inv_tiles_m ? (int)__umulhi((unsigned)tile_id, p.inv_tiles_m) : tile_id;
        tile_m = tile_id - tile_n * p.tiles_m;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int split = (int)blockIdx.y;
    const int kt_begin = split * p.ktiles_per_split;
    const int kt_end = min(p.ktiles, kt_begin + p.ktiles_per_split);
    const int nt = kt_end - kt_begin;

    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(p.a, p.a_bytes);
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, p.w_bytes);

    // ---- loader offsets: fixed per lane for the whole launch, the K offset rides in the DMA's scalar operand (gemm_kernel's dense issue)
    const int lrow = lane >> 3, lchk = lane & 7;
    const unsigned row_bytes = (unsigned)p.c1 * 2u;
    unsigned a_voff[AJ], b_off[BJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int row = (wave * AJ + j) * 8 + lrow;
        const int m = m0 + row;
        a_voff[j] = m < p.M ? (unsigned)m * row_bytes + (unsigned)((lchk ^ ((row >> 1) & 7)) * 16) : MDX_OOB;
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int row = (wave * BJ + j) * 8 + lrow;
        const int panel = (n0 >> 6) + (row >> 6);
        b_off[j] = (unsigned)panel * (unsigned)p.kt64 * 8192u + (unsigned)(((row & 63) * 8 + lchk) * 16);
    }
    auto stage_tile = [&](int kt, int buf) {
        char* sbase = smem + buf * STAGE;
#pragma unroll
        for (int j = 0; j < AJ; ++j) dma16s(rs_a, sbase + (wave * AJ + j) * 1024, a_voff[j], (unsigned)kt * 128u);
#pragma unroll
        for (int j = 0; j < BJ; ++j) dma16s(rs_w, sbase + A_BYTES + (wave * BJ + j) * 1024, b_off[j], (unsigned)kt * 8192u);
    };
#pragma unroll
    for (int i = 0; i < NS - 1; ++i)
        if (i < nt) stage_tile(kt_begin + i, i);
    __builtin_amdgcn_sched_barrier(0);       // (nothing below may be scheduled in front of the first tiles' issue)
    trace_mark(p, 1);

    // ---- everything the epilogue reads from global memory, requested now: younger than the prologue DMAs, older than every refill.
    // All of it goes through buffer descriptors whose bounds check answers zero for what does not exist (no bias, no residual, no
    // LayerNorm fold, columns / rows past the edge): UNCONDITIONAL loads into registers nothing touches until the K loop is over --
    // a conditional load merged with a default value makes the compiler wait for it on the spot (seen in the ISA of the first form
    // of this kernel: s_waitcnt vmcnt(0) right behind the bias load, a full miss in front of the loop) -- and the SAME number of
    // loads in every wave, so that the first wait of the loop can count them (E below).
    constexpr int CPR = BN / 8, RPP = 256 / CPR, PASSES = BM / RPP;
    constexpr int NX = PF >= 2 ? (PASSES < 4 ? PASSES : 4) : 0;
    constexpr int LNR = PF >= 3 ? LNR_MAX : 0;
    constexpr int NT = PF == 1 || PF == 2 ? 2 : 0;       // touches (LayerNorm statistics lines, S[n] lines)
    constexpr int NB = BN == 128 ? 4 : 2;      // bias loads: 8 columns, + the 8 GEGLU gate columns on 128-column tiles
    constexpr int E = NB + NX + LNR + (PF >= 2 ? 1 : 0) + NT;       // bias + residual rows + LayerNorm partials + S[n] + touches: EXACTLY the loads issued below (every
                                               // one is kept alive behind the loop; the first wait of the loop counts them)
    const bool geglu = p.epilogue == MDX_EPI_GEGLU;
    const __amdgpu_buffer_rsrc_t rs_bias = make_rsrc(p.bias, p.bias ? (unsigned)p.N * 4u : 0u);
    const __amdgpu_buffer_rsrc_t rs_res = make_rsrc(p.residual, (p.residual && !geglu) ? p.res_bytes : 0u);
    const bool ln_regs = PF >= 3 && p.ln_stats != nullptr && p.ln_nt <= LNR;
    const bool lns_regs = PF >= 2 && p.ln_stats != nullptr;
    const __amdgpu_buffer_rsrc_t rs_ln = make_rsrc(p.ln_stats, p.ln_stats ? (unsigned)p.M * (unsigned)p.ln_nt * 8u : 0u);
    const __amdgpu_buffer_rsrc_t rs_lns = make_rsrc(p.ln_s, p.ln_stats ? (unsigned)p.N * 4u : 0u);
    u32x4 braw[4] = {};
    {
        const int n = n0 + (geglu ? (tid & 7) : (tid % CPR)) * 8;
        const unsigned off = n < p.N ? (unsigned)n * 4u : MDX_OOB;
        braw[0] = __builtin_amdgcn_raw_buffer_load_b128(rs_bias, off, 0, 0);
        braw[1] = __builtin_amdgcn_raw_buffer_load_b128(rs_bias, off, 16, 0);
        if constexpr (NB == 4) {
            const unsigned goff = geglu ? off : MDX_OOB;       // GEGLU: the gate columns sit 64 further (gemm_bias_prefetch)
            braw[2] = __builtin_amdgcn_raw_buffer_load_b128(rs_bias, goff, 256, 0);
            braw[3] = __builtin_amdgcn_raw_buffer_load_b128(rs_bias, goff, 272, 0);
        }
    }
    u32x4 xraw[NX > 0 ? NX : 1];
    if constexpr (NX > 0) {
        const int n = n0 + (tid % CPR) * 8;
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            const int m = m0 + tid / CPR + q * RPP;
            const unsigned off = (m < p.M && n < p.N) ? ((unsigned)m * (unsigned)p.residual_ld + (unsigned)n) * 2u : MDX_OOB;
            xraw[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, off, 0, 0);
        }
    }
    u32x2 lnraw[LNR > 0 ? LNR : 1];
    if constexpr (LNR > 0) {
        const unsigned base = (tid < BM && m0 + tid < p.M) ? (unsigned)(m0 + tid) * (unsigned)p.ln_nt * 8u : MDX_OOB;
#pragma unroll
        for (int j = 0; j < LNR; ++j)
            lnraw[j] = __builtin_amdgcn_raw_buffer_load_b64(rs_ln, (base != MDX_OOB && j < p.ln_nt) ? base : MDX_OOB, j * 8, 0);
    }
    unsigned lns_raw = 0u;
    if constexpr (PF >= 2)
        lns_raw = __builtin_amdgcn_raw_buffer_load_b32(rs_lns, (tid < BN && n0 + tid < p.N) ? (unsigned)(n0 + tid) * 4u : MDX_OOB, 0, 0);
    unsigned touch[2] = {0u, 0u};
    if constexpr (NT > 0) {      // rows whose partials stay in memory: one dword of every 128-byte line, the epilogue's fold then hits L2
        const int rows = min(BM, p.M - m0);
        const unsigned bytes = (unsigned)rows * (unsigned)p.ln_nt * 8u;
        const unsigned base = (unsigned)m0 * (unsigned)p.ln_nt * 8u;
        touch[0] = __builtin_amdgcn_raw_buffer_load_b32(rs_ln, (unsigned)tid * 128u < bytes ? base + (unsigned)tid * 128u : MDX_OOB, 0, 0);
        touch[1] = __builtin_amdgcn_raw_buffer_load_b32(rs_lns, (PF < 2 && tid < BN / 32 && n0 + tid * 32 < p.N) ? (unsigned)(n0 + tid * 32) * 4u : MDX_OOB, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int swz = (l31 >> 1) & 7;
    const int a_row_off = (wm * WROWS + l31) * ROWB;
    const int b_row_off = A_BYTES + (wn * (BN / 2) + l31) * ROWB;

    // ---- main loop: gemm_kernel's (NS-stage ring, counted vmcnt + raw barrier, refill issue first, fragment reads pinned, MFMAs)
    int rd = 0, wr = NS - 1;
    for (int t = 0; t < nt; ++t) {
        const int ahead = min(NS - 2, nt - 1 - t);
        if (t == 0) {      // the E epilogue prefetches are younger than every prologue tile: tile 0 has landed when at most they and the
                           // other prologue tiles are outstanding; from t = 1 on they count as landed (they had a K tile's time)
            if (NS >= 6 && ahead >= 4)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * LPT + E) : "memory");
            else if (NS >= 5 && ahead == 3)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPT + E) : "memory");
            else if (NS >= 4 && ahead == 2)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT + E) : "memory");
            else if (NS >= 3 && ahead == 1)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT + E) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(E) : "memory");
        } else if (NS >= 6 && ahead >= 4)
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
        const char* sb = smem + rd * STAGE;
        constexpr bool ALLK = TM * TN <= 2;
        constexpr int FD = ALLK ? KS : 2;
        f16x8 af[FD][TM], bf[FD][TN];
        auto rdfrag = [&](auto slot_c, const int ks) {
            constexpr int slot = decltype(slot_c)::value;
            const int coff = (((2 * ks + hi) ^ swz) << 4);
#pragma unroll
            for (int i = 0; i < TM; ++i) af[slot][i] = *reinterpret_cast<const f16x8*>(sb + a_row_off + i * 32 * ROWB + coff);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[slot][j] = *reinterpret_cast<const f16x8*>(sb + b_row_off + j * 32 * ROWB + coff);
        };
        auto mfmas = [&](auto slot_c) {
            constexpr int slot = decltype(slot_c)::value;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[slot][j], af[slot][i], acc[i][j], 0, 0, 0);
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        if (t + NS - 1 < nt) stage_tile(kt_begin + t + NS - 1, wr);
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

    __syncthreads();      // all waves done with the ring before the epilogue reuses it
    trace_mark(p, 3);
    // unpack the prefetches (the loop's last wait was vmcnt(0): everything has landed).  Every prefetched register is named here
    // once, unconditionally: none of the E loads can be removed as dead, whatever the epilogue instantiation reads
#pragma unroll
    for (int q = 0; q < NB; ++q) asm volatile("" ::"v"(braw[q]));
#pragma unroll
    for (int q = 0; q < NX; ++q) asm volatile("" ::"v"(xraw[q]));
#pragma unroll
    for (int j = 0; j < LNR; ++j) asm volatile("" ::"v"(lnraw[j]));
    if constexpr (PF >= 2) asm volatile("" ::"v"(lns_raw));
#pragma unroll
    for (int q = 0; q < NT; ++q) asm volatile("" ::"v"(touch[q]));
    float bpre[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 f = __builtin_bit_cast(f32x4, braw[q]);
        bpre[4 * q] = f[0]; bpre[4 * q + 1] = f[1]; bpre[4 * q + 2] = f[2]; bpre[4 * q + 3] = f[3];
    }
    Row8Extras xpre[NX > 0 ? NX : 1];
#pragma unroll
    for (int q = 0; q < NX; ++q) xpre[q].res = __builtin_bit_cast(f16x8, xraw[q]);
    // LayerNorm fold: this thread's row statistics from the partials it holds, added in partial order (the order gemm_ln_row_fold
    // uses; the zeros of the slots past ln_nt change nothing: same bits)
    float ln_pre[2];
    {
        float su = 0.f, sq = 0.f;
#pragma unroll
        for (int j = 0; j < LNR; ++j) {
            const float2 v = __builtin_bit_cast(float2, lnraw[j]);
            su += v.x;
            sq += v.y;
        }
        const float inv = 1.0f / (float)p.K;
        const float mean = su * inv;
        float var = sq * inv - mean * mean;
        var = var < 0.f ? 0.f : var;
        ln_pre[0] = mean;
        ln_pre[1] = rsqrtf(var + p.ln_eps);
    }
    const float lns_pre = __builtin_bit_cast(float, lns_raw);
    // (xpre is read only where the descriptor has a residual; the GEGLU store path fetches nothing)
    if constexpr (XA) {
        static_assert(BN == 64, "cross-attention epilogue: one 64-wide head per tile");
        // the head's K / V^T tiles -> LDS behind the staging area, in flight while the accumulators are corrected and staged
        constexpr int XOFF = ((BM * 72 * 2 + 2 * BM * 4 + BN * 4 + 1023) / 1024) * 1024;
        char* xs = smem + XOFF;
        {
            const int bsamp = m0 / p.HoWo, head = n0 >> 6;
            const f16* kb = p.xa_k + (size_t)bsamp * p.xa_cap * p.N + head * 64;
            const f16* vb = p.xa_vt + ((size_t)bsamp * p.N + head * 64) * p.xa_cap;
            const __amdgpu_buffer_rsrc_t rs_k = make_rsrc(kb, (unsigned)(((size_t)(p.xa_len - 1) * p.N + 64) * 2));
            const __amdgpu_buffer_rsrc_t rs_v = make_rsrc(vb, (unsigned)((size_t)64 * p.xa_cap * 2));
            const int ntile = (p.xa_len + 63) >> 6;
            for (int t = 0; t < ntile; ++t) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int inst = wave * 2 + j;
                    const int krow = inst * 8 + (lane >> 3);
                    const int kchunk = (lane & 7) ^ ((krow >> 1) & 7);
                    const int key = t * 64 + krow;
                    dma16(rs_k, xs + t * 16384 + inst * 1024, key < p.xa_len ? (unsigned)(((size_t)key * p.N + kchunk * 8) * 2) : MDX_OOB);
                    const int vrow = inst * 8 + (lane >> 3);
                    const unsigned vchunk = (unsigned)((lane & 7) ^ ((vrow >> 1) & 7));
                    const int kc = t * 64 + (int)vchunk * 8;
                    dma16(rs_v, xs + t * 16384 + 8192 + inst * 1024, kc < p.xa_cap ? (unsigned)(((size_t)vrow * p.xa_cap + kc) * 2) : MDX_OOB);
                }
            }
        }
        gemm_epilogue<BM, BN, true, NW, LinearRows, NX, 3>(p, acc, smem, LinearRows{m0}, n0, split, bpre, tile_m, tile_id, ln_pre, &lns_pre, xpre,
                                                           ln_regs, lns_regs);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();      // K / V^T of every wave's DMAs landed (the staged q tile was ordered by the epilogue's own barrier)
        xattn_tile_epilogue<BM>(p, smem, xs, m0, n0);
    } else {
        gemm_epilogue<BM, BN, true, NW, LinearRows, NX, LEAN_EPI ? 1 : 0>(p, acc, smem, LinearRows{m0}, n0, split, bpre, tile_m, tile_id, ln_pre,
                                                                  &lns_pre, xpre, ln_regs, lns_regs);
    }
    trace_mark(p, 4);
}

template <int BM, int BN, int NS, int PF>
void launch_dense_one(const GemmParams& p, dim3 grid, hipStream_t st) {
    constexpr size_t ring = (size_t)NS * (BM + BN) * 64 * 2;
    constexpr size_t epi = (size_t)(BM > BN ? BM : BN) * ((BM > BN ? BN : BM) + 8) * 2 + 4096;
    constexpr size_t lds = ring > epi ? ring : epi;
    static MdxPerDeviceOnce attr_once;
    if (attr_once.first())
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_kernel<BM, BN, NS, PF>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((dense_kernel<BM, BN, NS, PF>), grid, dim3(256), lds, st, p);
}

// the cross-attention-epilogue instantiations (BN = 64, prefetch level 1): LDS = ring | staged q + LayerNorm extras + two K / V^T tile pairs
template <int BM, int NS>
void launch_dense_xa(const GemmParams& p, dim3 grid, hipStream_t st) {
    constexpr size_t ring = (size_t)NS * (BM + 64) * 64 * 2;
    constexpr size_t xa = (size_t)((BM * 72 * 2 + 2 * BM * 4 + 64 * 4 + 1023) / 1024) * 1024 + 2 * 16384;
    constexpr size_t lds = ring > xa ? ring : xa;
    static MdxPerDeviceOnce attr_once;
    if (attr_once.first())
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_kernel<BM, 64, NS, 1, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((dense_kernel<BM, 64, NS, 1, true>), grid, dim3(256), lds, st, p);
}

template <int BM, int BN, int NS>
void launch_dense_pf(const GemmParams& p, int pf, dim3 grid, hipStream_t st) {
    if constexpr (BM * BN < 128 * 128) {      // (128 x 128 tiles: the prefetch registers do not fit beside 64 accumulators)
        if (pf >= 3) return launch_dense_one<BM, BN, NS, 3>(p, grid, st);
        if (pf == 2) return launch_dense_one<BM, BN, NS, 2>(p, grid, st);
    }
    if (pf >= 1) return launch_dense_one<BM, BN, NS, 1>(p, grid, st);
    launch_dense_one<BM, BN, NS, 0>(p, grid, st);
}

template <int BM, int BN>
bool launch_dense_ns(const GemmParams& p, int ns, int pf, dim3 grid, hipStream_t st) {
    switch (ns) {
        case 2: launch_dense_pf<BM, BN, 2>(p, pf, grid, st); return true;
        case 3: launch_dense_pf<BM, BN, 3>(p, pf, grid, st); return true;
        case 4: if constexpr (BM == 64 || BN == 64) { launch_dense_pf<BM, BN, 4>(p, pf, grid, st); return true; } return false;
        case 5: if constexpr (BM == 64) { launch_dense_pf<BM, BN, 5>(p, pf, grid, st); return true; } return false;
        case 6: if constexpr (BM == 64) { launch_dense_pf<BM, BN, 6>(p, pf, grid, st); return true; } return false;
        default: return false;
    }
}

}  // namespace

// `grid` is what mdx_gemm_f16 computed for the generic kernel; the SPREAD order becomes a 2-D grid (tiles, splits) here.
// Option gemm_lean_dense: 1 = prefetch level 1 (the product), 2 = level 0, 3 / 4 = levels 2 / 3 on launches of at most two blocks
// per CU (beyond that the register-hungry levels would take a block per CU away: level 1).
bool mdx_dense_launch(const GemmParams& p, int bm, int bn, int ns, dim3 grid, hipStream_t st) {
    dim3 g = p.spread ? dim3((unsigned)(p.tiles_m * p.tiles_n), (unsigned)p.nsplit) : grid;
    const int opt = mdx_opt(MDX_OPT_GEMM_LEAN_DENSE);
    const bool small = (long)g.x * g.y <= 512;
    const int pf = opt == 2 ? 0 : (opt == 3 && small ? 2 : (opt == 4 && small ? 3 : 1));
    if (p.xa_k) {      // cross-attention epilogue (mdx_gemm_desc.xattn_k): 64-column tiles, ring depth 2 .. 4
        if (bn != 64 || ns < 2 || ns > 4) return false;
        if (bm == 64) { if (ns == 2) launch_dense_xa<64, 2>(p, g, st); else if (ns == 3) launch_dense_xa<64, 3>(p, g, st); else launch_dense_xa<64, 4>(p, g, st); return true; }
        if (bm == 128) { if (ns == 2) launch_dense_xa<128, 2>(p, g, st); else if (ns == 3) launch_dense_xa<128, 3>(p, g, st); else launch_dense_xa<128, 4>(p, g, st); return true; }
        return false;
    }
    if (bm == 64 && bn == 64) return launch_dense_ns<64, 64>(p, ns, pf, g, st);
    if (bm == 64 && bn == 128) return launch_dense_ns<64, 128>(p, ns, pf, g, st);
    if (bm == 128 && bn == 64) return launch_dense_ns<128, 64>(p, ns, pf, g, st);
    if (bm == 128 && bn == 128) return launch_dense_ns<128, 128>(p, ns, pf, g, st);
    return false;
}
