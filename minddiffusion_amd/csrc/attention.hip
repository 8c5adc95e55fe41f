// Fused flash-style attention for gfx950: O = softmax(Q K^T * scale) V without materialising scores.
// Replaces ops.matmul + ops.Softmax + ops.matmul of CrossAttention.construct
// (vision/stablediffusionv2/ldm/modules/attention.py:138-152), self- and cross-attention.
//
// Block = 4 wave64, 128 queries (32 per wave); K/V^T tiles of 64 keys are DMA'd HBM->LDS
// (buffer_load ... lds, XOR-swizzled on the source side) and double-buffered.
// Per wave and KV tile:
//   S^T[key][q] = K Q^T     : MFMA 32x32x16 f16, A = K rows (ds_read_b128), B = Q (registers)
//   online softmax           : every lane owns ONE query (col = lane&31) and 32 of the 64 keys,
//                              so row max / row sum need a single lane^32 exchange
//   O^T[d][q] += V^T P^T    : A = V^T rows (one ds_read_b128 per fragment), B = P straight from the S^T
//                              accumulator registers: the C layout of S^T IS the B layout of PV up to a
//                              permutation of the keys inside a 32-key tile, and that permutation is undone
//                              for free on the other side -- lane l31 of the QK^T MFMA multiplies K row
//                              pi(l31) (bits 2 and 3 swapped), so accumulator register r of lane (q, hi) holds
//                              key 16(r>>3) + 8hi + (r&7) and k index 8hi + j of PV chunk c is key 16c + 8hi + j:
//                              contiguous in V^T -- no cross-lane moves, no register shuffles.
// Head dims 40 / 64 / 80 / 160 (SDv2: 64; Wukong-Huahua: 8 heads => 40 / 80 / 160; GLIDE: 64).
// V must be supplied TRANSPOSED ([b][h*D+d][key]); the projection GEMM writes it that way
// (MDX_OUT_TRANSPOSED), which keeps every LDS read of this kernel wide and conflict-light.
#include "mdx_common.h"

#include <type_traits>
#include <utility>

namespace {

struct AttnParams {
    const f16* q;
    const f16* k;
    const f16* vt;
    f16* o;
    long q_bs, k_bs, vt_bs, o_bs;
    int q_ld, k_ld, vt_ld, o_ld;
    int B, heads, Nq, Nk;
    float scale_log2;  // scale * log2(e)
    unsigned k_bytes, vt_bytes;  // per-batch extents for the buffer descriptors
    int causal;                  // 1: key j is visible to query i only if j <= i (text encoder, text_encoder.py:136-139)
    // split-KV form (mdx_attention_splitkv_f16): the key tiles of a (batch, head, 128-query block) item are dealt to nsplit blocks;
    // each parks its normalised partial output + (reference, row sum) in ws_part, the last arriver of the item combines them
    int nsplit, qblocks;
    char* ws_part;               // [item][split]{ fp16 O~[128][D], float2 {m * scale_log2, l}[128] }
    unsigned* tickets;           // [item] arrival counters: zero on entry, left zero
    int fast_stage;              // full KV tiles are issued with fixed per-lane offsets + a scalar tile offset (option attn_fast_stage)
};

constexpr int BQ = 128;
constexpr int BKV = 64;
constexpr int ATTN_MAX_SPLITS_K = 8;      // most KV splits of one item (mdx_attention_splitkv_f16)

// The end of a block's work, shared by attn_kernel and attn_pipe_kernel: O /= l, staged [q][d] per wave in LDS, full-row stores -- or,
// for a split-KV launch, the partial hand-off (normalised partial + {reference, row sum} to the workspace, ticket, last arriver combines).
template <int D>
__device__ __forceinline__ void attn_finish(const AttnParams& p, f32x16 (&acc_o)[(D + 31) / 32], const float m_run, const float l_run,
                                            char* smem, const int b, const int h, const int qblk, const int q0, const int split) {
    constexpr int DT = (D + 31) / 32;
    constexpr bool LROW = (D % 32) != 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    // ---- finalize: O /= l ; stage [q][d] per wave in LDS, then full-row stores
    float l_tot;
    if constexpr (LROW) {      // O^T row D: d tile D / 32, local row D % 32 = (r & 3) + 8 (r >> 2) + 4 hi
        constexpr int LR = D % 32, LREG = (LR & 3) + 4 * (LR >> 3), LHI = (LR >> 2) & 1;
        const float mine = acc_o[D / 32][LREG];
        const float other = __shfl_xor(mine, 32, 64);
        l_tot = (hi == LHI) ? mine : other;
    } else {
        l_tot = l_run + __shfl_xor(l_run, 32, 64);
    }
    const float inv = 1.0f / l_tot;
    constexpr int OLD = DT * 32 + 8;
    f16* og = reinterpret_cast<f16*>(smem) + wave * 32 * OLD;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (f16)(acc_o[d][4 * g + e] * inv);
            *reinterpret_cast<f16x4*>(&og[l31 * OLD + d * 32 + 8 * g + 4 * hi]) = v;
        }
    __syncthreads();
    constexpr int CPR = D / 8;   // 16-B chunks per output row
    if (p.nsplit > 1) {
        // ---- split-KV hand-off (the scheme of splitk_last_block_reduce, gemm_internal.h): this block's NORMALISED partial
        // O~ = O / l in fp16 (the rounding the final output gets anyway) and per query {m * scale_log2, l} go to the workspace with
        // write-through (sc1) stores; drain; block barrier; one lane takes a ticket on the item's counter; the last arriver reads
        // all nsplit partials with sc1 loads and stores  sum_s w_s O~_s / sum_s w_s,  w_s = l_s 2^(m_s - max m)  -- summed in split
        // order whoever arrives last, so the result does not depend on the schedule.
        typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
        typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
        constexpr unsigned O_BYTES = (unsigned)BQ * D * 2u;
        constexpr unsigned PART = O_BYTES + (unsigned)BQ * 8u;
        const size_t item = ((size_t)b * p.heads + h) * p.qblocks + qblk;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.ws_part + item * p.nsplit * PART, (unsigned)p.nsplit * PART);
        for (int idx = lane; idx < 32 * CPR; idx += 64) {
            const int row = idx / CPR, chunk = idx - row * CPR;
            const f16x8 v = *reinterpret_cast<const f16x8*>(&og[row * OLD + chunk * 8]);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), rs,
                                                   (unsigned)split * PART + (unsigned)(((wave * 32 + row) * D + chunk * 8) * 2), 0, /*sc1*/ 16);
        }
        if (hi == 0) {
            u32x2v ml;
            ml[0] = __float_as_uint(m_run * p.scale_log2);
            ml[1] = __float_as_uint(l_tot);
            __builtin_amdgcn_raw_buffer_store_b64(ml, rs, (unsigned)split * PART + O_BYTES + (unsigned)(wave * 32 + l31) * 8u, 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* flag = reinterpret_cast<int*>(smem + 4 * 32 * OLD * 2);      // behind the four waves' staging rows
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(p.tickets + item, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old >= (unsigned)p.nsplit) __builtin_trap();     // counters not zero on entry (two streams on one workspace, mdx.h)
            *flag = old == (unsigned)p.nsplit - 1u;
        }
        __syncthreads();
        if (*flag == 0) return;
        if (tid == 0) __hip_atomic_store(p.tickets + item, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int idx = tid; idx < BQ * CPR; idx += 256) {
            const int row = idx / CPR, chunk = idx - row * CPR;
            const int qi = qblk * BQ + row;
            if (qi >= p.Nq) continue;
            // every partial's {reference, row sum} and O~ chunk in flight together (one fabric round trip, not 2 * nsplit)
            u32x2v ml[ATTN_MAX_SPLITS_K];
            u32x4v raw[ATTN_MAX_SPLITS_K];
#pragma unroll
            for (int z = 0; z < ATTN_MAX_SPLITS_K; ++z)
                if (z < p.nsplit) {
                    ml[z] = __builtin_amdgcn_raw_buffer_load_b64(rs, (unsigned)z * PART + O_BYTES + (unsigned)row * 8u, 0, 16);
                    raw[z] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)z * PART + (unsigned)((row * D + chunk * 8) * 2), 0, 16);
                }
            float mmax = -INFINITY;
#pragma unroll
            for (int z = 0; z < ATTN_MAX_SPLITS_K; ++z)
                if (z < p.nsplit) mmax = fmaxf(mmax, __uint_as_float(ml[z][0]));
            float acc[8], wsum = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
            for (int z = 0; z < ATTN_MAX_SPLITS_K; ++z)
                if (z < p.nsplit) {
                    const f16x8 v = __builtin_bit_cast(f16x8, raw[z]);
                    const float w = __uint_as_float(ml[z][1]) * __builtin_amdgcn_exp2f(__uint_as_float(ml[z][0]) - mmax);
                    wsum += w;
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += w * (float)v[e];
                }
            const float winv = 1.0f / wsum;
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (f16)(acc[e] * winv);
            *reinterpret_cast<f16x8*>(p.o + (size_t)b * p.o_bs + (size_t)qi * p.o_ld + h * D + chunk * 8) = o;
        }
        return;
    }
    for (int idx = lane; idx < 32 * CPR; idx += 64) {
        const int row = idx / CPR, chunk = idx - row * CPR;
        const int qi = q0 + row;
        if (qi < p.Nq) {
            const f16x8 v = *reinterpret_cast<const f16x8*>(&og[row * OLD + chunk * 8]);
            *reinterpret_cast<f16x8*>(p.o + (size_t)b * p.o_bs + (size_t)qi * p.o_ld + h * D + chunk * 8) = v;
        }
    }
}

// OCC = blocks the register budget admits per CU (a block is one wave per SIMD): 2 is the form of rounds 1-3 (<= 256 VGPRs);
// 3 (<= 168 VGPRs, D <= 64 only: the D = 64 kernel needs 182 unconstrained and fits 168 with eight spills OUTSIDE the tile loop)
// puts a third independent wave on every SIMD -- a lone wave spends ~2000 cycles on a KV tile whose VALU + MFMA issue is ~1000,
// two co-resident waves ~2200 per pair (tools/attn_bench.py), so the SIMD still has issue slots to give.
template <int D, int OCC>
__global__ __launch_bounds__(256, OCC) void attn_kernel(const AttnParams p) {
    mdx_kernarg_touch<sizeof(AttnParams)>();
    static_assert(D % 8 == 0 && D <= 160, "head dim must be a multiple of 8, <= 160");
    constexpr int KS = (D + 15) / 16;          // k-steps of QK^T (contraction over d, zero-padded to 16)
    constexpr int DT = (D + 31) / 32;          // 32-row d tiles of O^T
    constexpr int KCH = (2 * KS <= 8) ? 8 : (2 * KS <= 16 ? 16 : 32);   // 16-B chunks per K row in LDS
    constexpr int K_ROWB = KCH * 16;           // 128 | 256 | 512 bytes
    constexpr int K_RPI = 64 / KCH;            // K rows covered by one DMA instruction (8 | 4 | 2)
    constexpr int K_DMA = BKV / K_RPI / 4;     // K DMA instructions per wave per tile
    constexpr int K_BYTES = BKV * K_ROWB;      // [key][d]
    constexpr int V_ROWB = 128;                // [d][64 keys]
    constexpr int V_ROWS = DT * 32;
    constexpr int V_DMA = (V_ROWS / 8 + 3) / 4;   // V DMA instructions per wave per tile (8 rows each)
    constexpr int V_BYTES = V_ROWS * V_ROWB;
    constexpr int STAGE = K_BYTES + V_BYTES;
    // Row sums through the matrix pipe where the V^T tile has a spare row (D = 40, 80: the 32-row d tiles are padded): row D of the
    // LDS tile is all ones (written once; the DMA never touches rows >= D), so O^T[D][q] accumulates sum_key P[q][key] in the PV
    // MFMAs that run anyway -- of the fp16 P the MFMA multiplies, rescaled with O for free -- and the softmax loses its 24 VALU
    // additions per lane and tile (of ~140; the kernel is VALU-bound).  D = 64 / 160 have no spare row and keep the VALU sum.
    constexpr bool LROW = (D % 32) != 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z, h = blockIdx.y;
    // split-KV launches: blockIdx.x = query block * nsplit + split, so the blocks of one item are dispatched back to back
    const int qblk = p.nsplit > 1 ? (int)blockIdx.x / p.nsplit : (int)blockIdx.x;
    const int split = p.nsplit > 1 ? (int)blockIdx.x - qblk * p.nsplit : 0;
    const int q0 = qblk * BQ + wave * 32;

    const f16* kb = p.k + (size_t)b * p.k_bs + h * D;
    const f16* vb = p.vt + (size_t)b * p.vt_bs + (size_t)h * D * p.vt_ld;
    const __amdgpu_buffer_rsrc_t rs_k = make_rsrc(kb, p.k_bytes);
    const __amdgpu_buffer_rsrc_t rs_v = make_rsrc(vb, p.vt_bytes);

    // ---- Q fragments (B operand): lane (q = l31, hi) holds Q[q][16s + 8hi .. +7] (zero beyond D / Nq)
    f16x8 qf[KS];
    {
        const int qi = q0 + l31;
        const f16* qp = p.q + (size_t)b * p.q_bs + (size_t)qi * p.q_ld + h * D + hi * 8;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (qi < p.Nq && s * 16 + hi * 8 < D)
                qf[s] = *reinterpret_cast<const f16x8*>(qp + s * 16);
            else
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[s][e] = (f16)0.f;
        }
    }

    // K rows: physical chunk q of row r holds logical chunk q ^ key(r); key chosen so that the 16 rows of a
    // ds_read_b128 lane group land on 16 distinct 16-B slots of the 256-B bank row.
    auto kkey = [](int row) { return KCH == 8 ? ((row >> 1) & 7) : (row & 15); };
    auto stage_tile = [&](int t, int buf) {
        char* sb = smem + buf * STAGE;
        const int key0 = t * BKV;
#pragma unroll
        for (int j = 0; j < K_DMA; ++j) {
            const int row = (wave * K_DMA + j) * K_RPI + lane / KCH;   // key row within the tile
            const int chunk = (lane % KCH) ^ kkey(row);               // logical chunk this lane fetches
            const int key = key0 + row;
            const unsigned off = (key < p.Nk && chunk * 8 < D) ? (unsigned)(((size_t)key * p.k_ld + chunk * 8) * 2) : MDX_OOB;
            dma16(rs_k, sb + (wave * K_DMA + j) * 1024, off);
        }
#pragma unroll
        for (int j = 0; j < V_DMA; ++j) {
            const int inst = wave * V_DMA + j;
            if (inst * 8 < (LROW ? D : V_ROWS)) {                     // LROW: rows >= D hold the ones row / zeros, never refilled
                const int row = inst * 8 + (lane >> 3);               // d row
                const unsigned chunk = (unsigned)((lane & 7) ^ ((row >> 1) & 7));
                const int kc = key0 + (int)chunk * 8;                  // first key of this 16-B chunk
                const unsigned off = (row < D && kc < p.vt_ld) ? (unsigned)(((size_t)row * p.vt_ld + kc) * 2) : MDX_OOB;
                dma16(rs_v, sb + K_BYTES + inst * 1024, off);
            }
        }
    };
    // FULL tiles (every key of the tile exists: all but a ragged last one) -- round 5: the per-lane source offsets are the same in
    // every tile up to a UNIFORM tile offset, so they are computed once and the tile offset rides in the DMA instruction's scalar
    // operand.  stage_tile above spends ~13 VALU instructions per DMA (two 64-bit multiplies among them) -- 4 DMAs per wave and
    // tile at D = 64, a sixth of the VALU work of a kernel that is VALU-bound.
    // measured (tools/attn_bench.py, profiles/r05_attn_fast_stage.txt): D = 40 515 -> 499 us, D = 64 1013 -> 1027 us (the three-blocks-per-CU
    // register budget of D = 64 turns the extra offsets into spills) -- so only where the budget has room
    constexpr bool FASTST = D == 40 || D == 80;
    unsigned k_voff[FASTST ? K_DMA : 1], v_voff[FASTST ? V_DMA : 1];
    if constexpr (FASTST) {
#pragma unroll
    for (int j = 0; j < K_DMA; ++j) {
        const int row = (wave * K_DMA + j) * K_RPI + lane / KCH;
        const int chunk = (lane % KCH) ^ kkey(row);
        k_voff[j] = (chunk * 8 < D) ? (unsigned)(((size_t)row * p.k_ld + chunk * 8) * 2) : MDX_OOB;
    }
#pragma unroll
    for (int j = 0; j < V_DMA; ++j) {
        const int row = (wave * V_DMA + j) * 8 + (lane >> 3);
        const unsigned chunk = (unsigned)((lane & 7) ^ ((row >> 1) & 7));
        v_voff[j] = (row < D) ? (unsigned)(((size_t)row * p.vt_ld + chunk * 8) * 2) : MDX_OOB;
    }
    }
    const unsigned k_tile_bytes = (unsigned)BKV * (unsigned)p.k_ld * 2u;
    auto stage_full = [&](int t, int buf) {
        if constexpr (!FASTST) return;
        char* sb = smem + buf * STAGE;
#pragma unroll
        for (int j = 0; j < K_DMA; ++j) dma16s(rs_k, sb + (wave * K_DMA + j) * 1024, k_voff[j], (unsigned)t * k_tile_bytes);
#pragma unroll
        for (int j = 0; j < V_DMA; ++j) {
            const int inst = wave * V_DMA + j;
            if (inst * 8 < (LROW ? D : V_ROWS)) dma16s(rs_v, sb + K_BYTES + inst * 1024, v_voff[j], (unsigned)t * (BKV * 2u));
        }
    };
    const int nk_full = p.Nk / BKV;      // tiles t < nk_full are full
    auto stage = [&](int t, int buf) {
        if (FASTST && p.fast_stage && t < nk_full) stage_full(t, buf); else stage_tile(t, buf);
    };

    f32x16 acc_o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[d][r] = 0.f;
    float m_run = -INFINITY;  // running max of raw scores (same value in lanes l and l^32)
    float l_run = 0.f;        // lane-partial running sum

    const int vswz = (lane >> 1) & 7;
    const int krow_l = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);   // pi(l31): swap bits 2 and 3
    const int ntiles = (p.Nk + BKV - 1) / BKV;

    // One KV tile.  BUF and MASK are compile-time so that every LDS address is (loop-invariant register + immediate)
    // and the tail-key masking costs nothing in the full tiles.  The softmax is the VALU-bound part of this kernel at
    // D = 64 (32 exp per lane per tile vs 16 MFMAs): raw v_exp_f32 (__builtin_amdgcn_exp2f: no denormal range fix-up,
    // arguments below -126 flush to 0, which is exactly what a softmax weight of 2^-126 should be).
    auto tile = [&](auto buf_c, auto mask_c, const int t) {
        constexpr int BUF = decltype(buf_c)::value;
        constexpr bool MASK = decltype(mask_c)::value;
        const char* sk = smem + BUF * STAGE;
        const char* sv = sk + K_BYTES;

        // ---- S^T = K Q^T : two 32-key tiles.  Lane l31 multiplies K row pi(l31) (bits 2 and 3 of the row swapped): S^T's
        // accumulator rows then hold the keys in the order in which the PV MFMA wants them as its B operand -- k index
        // 8*hi + j of chunk c <-> key 16c + 8*hi + j -- so a V^T fragment is ONE contiguous 16-B LDS read.
        f32x16 acc_s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_s[kt][r] = 0.f;
            const int krow = kt * 32 + krow_l;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const f16x8 kf = *reinterpret_cast<const f16x8*>(sk + krow * K_ROWB + (((2 * s + hi) ^ kkey(krow)) << 4));
                acc_s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], acc_s[kt], 0, 0, 0);
            }
        }
        if constexpr (MASK) {   // keys beyond Nk (last tile) and, for causal attention, keys after this lane's query
            const int key0 = t * BKV;
            const int kmax = p.causal ? min(p.Nk - 1, q0 + l31) : p.Nk - 1;   // last visible key of this lane's query
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + kt * 32 + (r & 3) + 4 * ((r >> 2) & 1) + 8 * hi + 16 * (r >> 3);   // pi(row)
                    if (key > kmax) acc_s[kt][r] = -INFINITY;
                }
        }
        // ---- online softmax (one query per lane, keys split between lane and lane^32)
        float mx = acc_s[0][0];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc_s[kt][r]);
        {   // lane <-> lane^32 exchange as ONE VALU instruction (gfx950 v_permlane32_swap) instead of an LDS round trip
            // (ds_bpermute + lgkmcnt(0) sat on the critical path of every tile): after the swap the two results hold
            // {own, partner} in some order in every lane, and max is symmetric
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        // Reference maximum: move it only when some query of this wave outgrew the current one by more than 2^8 (in the log2
        // domain the exponentials are taken in).  The softmax is invariant to the reference; a stale one just means weights
        // up to 2^8 instead of <= 1 -- nowhere near the fp16 range of the P fragments or the fp32 range of l and O -- and it
        // saves the alpha exponential and the 16 * DT accumulator multiplies on almost every tile (with the exact rule
        // "any of the wave's 32 queries saw a new maximum" 55-85 % of the tiles still rescaled: new maxima keep arriving at
        // rate ~32 / t).  Tile 0: m_run = -inf, the difference is +inf, the branch is taken and alpha = exp2(-inf) = 0.
        const float m_cand = fmaxf(m_run, mx);          // finite: tile 0 always has a visible key (key 0)
        if (__builtin_amdgcn_ballot_w64((m_cand - m_run) * p.scale_log2 > 8.0f)) {
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_cand) * p.scale_log2);
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_o[d][r] *= alpha;
            m_run = m_cand;
        }
        const float mb = m_run * p.scale_log2;
        // two scores per instruction where the ISA has packed fp32 (v_pk_fma_f32 for x * scale - m, v_pk_add_f32 for the
        // row sum): the softmax VALU work, not the MFMAs, bounds this kernel at D = 64
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 sc2 = {p.scale_log2, p.scale_log2}, nmb2 = {-mb, -mb};
        f32x2 psum2 = {0.f, 0.f};
        f16x8 pf[4];  // B fragments for the 4 16-key chunks
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 x2 = {acc_s[kt][r], acc_s[kt][r + 1]};
                const f32x2 a2 = __builtin_elementwise_fma(x2, sc2, nmb2);
                const f32x2 pv2 = {__builtin_amdgcn_exp2f(a2.x), __builtin_amdgcn_exp2f(a2.y)};
                if constexpr (!LROW) psum2 += pv2;
                pf[kt * 2 + (r >> 3)][r & 7] = (f16)pv2.x;
                pf[kt * 2 + (r >> 3)][(r & 7) + 1] = (f16)pv2.y;
            }
        if constexpr (!LROW) l_run += psum2.x + psum2.y;

        // ---- O^T += V^T P^T
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const char* vr = sv + (d * 32 + l31) * V_ROWB;
                const f16x8 vf = *reinterpret_cast<const f16x8*>(vr + (((2 * c + hi) ^ vswz) << 4));   // keys 16c + 8hi .. +7
                acc_o[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[c], acc_o[d], 0, 0, 0);
            }
        }
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    const bool ragged = (p.Nk % BKV) != 0;
    int ntl = ntiles;
    if (p.causal) ntl = min(ntiles, (qblk * BQ + BQ - 1) / BKV + 1);   // tiles wholly in the future are skipped
    const int nfull = p.causal ? 0 : (ragged ? ntl - 1 : ntl);                 // leading tiles that need no masking
    // this block's tile range [t_begin, t_end): everything, or the split's share (even boundaries: tile t lives in buffer t & 1)
    int t_begin = 0, t_end = ntl;
    if (p.nsplit > 1) {
        t_begin = ((split * ntiles) / p.nsplit) & ~1;
        t_end = split + 1 == p.nsplit ? ntiles : ((((split + 1) * ntiles) / p.nsplit) & ~1);
    }
    const int nfull_l = min(nfull, t_end);

    if constexpr (LROW) {      // rows D .. V_ROWS - 1 of both V^T buffers: row D = ones, the rest zero (16-byte pieces, 8 per row)
        for (int i = tid; i < 2 * (V_ROWS - D) * 8; i += 256) {
            const int buf = i / ((V_ROWS - D) * 8), rem = i - buf * (V_ROWS - D) * 8;
            const int row = D + (rem >> 3);
            f16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = row == D ? (f16)1.0f : (f16)0.f;
            *reinterpret_cast<f16x8*>(smem + buf * STAGE + K_BYTES + row * V_ROWB + (rem & 7) * 16) = v;
        }
    }
    // tile t lives in LDS buffer t & 1; tile t + 1 is staged while tile t is computed
    stage(t_begin, 0);
    __syncthreads();
    int t = t_begin;
    for (; t + 2 <= nfull_l; t += 2) {
        stage(t + 1, 1);
        tile(B0{}, std::false_type{}, t);
        __syncthreads();
        if (t + 2 < t_end) stage(t + 2, 0);
        tile(B1{}, std::false_type{}, t + 1);
        __syncthreads();
    }
    for (; t < t_end; ++t) {      // the (masked) tail: one tile for a ragged Nk, every tile for causal attention
        if (t + 1 < t_end) stage(t + 1, (t + 1) & 1);
        const bool masked = t >= nfull;
        if (t & 1) {
            if (masked) tile(B1{}, std::true_type{}, t); else tile(B1{}, std::false_type{}, t);
        } else {
            if (masked) tile(B0{}, std::true_type{}, t); else tile(B0{}, std::false_type{}, t);
        }
        __syncthreads();
    }

    attn_finish<D>(p, acc_o, m_run, l_run, smem, b, h, qblk, q0, split);
}


// ---------------------------------------------------------------------------------------------------------------------------------
// attn8 (round 4, MEASURED SLOWER, opt-in through option attn8 = 2): the same flash attention with EIGHT waves per block in two
// groups half a tile apart -- the structure of conv8p.hip with the softmax in the place of the LDS reads.
//
// Hypothesis: PMC (profiles/r02_attention_pmc.md) shows VALU issue on 58 % and the matrix pipe on 38 % of the SIMD cycles, together
// ~96 % -- as if the softmax of one wave and the MFMAs of the other never overlapped.  So build the overlap: one global phase = one
// block barrier; group 0 runs its matrix phase M(i) = PV(i-1) + QK^T(i) in even phases and its softmax phase V(i) in odd ones,
// group 1 the other way round; both groups read the SAME K / V^T tiles (256 queries per block); the PV product of a tile moves into
// the NEXT matrix phase (its P fragments wait in registers) so that a phase is purely matrix or purely VALU.
//   phase 2i    : everyone issues the DMA of K(i+1) and V^T(i);  group 0: M(i);  group 1: V(i-1)
//   phase 2i+1  :                                                 group 0: V(i);  group 1: M(i);   wait for the batch, barrier
// Result (tools/attn_bench.py, same process, MI355X): 9216 tokens 827 -> 694 TF/s, 4096 tokens at batch 2 594 -> 474, Wukong's
// 40-wide heads 630 -> 461.  A phase takes ~1500 cycles however it is filled: the ~160 VALU instructions of a tile's softmax ARE the
// tile time (dependent-issue latency with one issuing wave per SIMD), the matrix phase hides under them in the four-wave kernel
// already (two independent blocks per CU overlap dynamically, with no barrier coupling their phases), and lock-stepping the two
// waves of a SIMD only adds the barrier and the max(M, V) of every phase.  The four-wave kernel stays the product; this one is
// kept, parity-tested, as the record of that measurement.
template <int D>
__global__ __launch_bounds__(512) void attn8_kernel(const AttnParams p) {
    mdx_kernarg_touch<sizeof(AttnParams)>();
    static_assert(D % 8 == 0 && D <= 160, "head dim must be a multiple of 8, <= 160");
    constexpr int KS = (D + 15) / 16;
    constexpr int DT = (D + 31) / 32;
    constexpr int KCH = (2 * KS <= 8) ? 8 : (2 * KS <= 16 ? 16 : 32);
    constexpr int K_ROWB = KCH * 16;
    constexpr int K_RPI = 64 / KCH;
    constexpr int K_INST = BKV / K_RPI;            // K DMA instructions per tile (8 | 16 | 32)
    constexpr int K_DMA = (K_INST + 7) / 8;        // ... per wave
    constexpr int K_BYTES = BKV * K_ROWB;
    constexpr int V_ROWB = 128;
    constexpr int V_ROWS = DT * 32;
    constexpr int V_INST = V_ROWS / 8;
    constexpr int V_DMA = (V_INST + 7) / 8;
    constexpr int V_BYTES = V_ROWS * V_ROWB;
    extern __shared__ __attribute__((aligned(16))) char smem[];      // K[2] | V^T[2]
    char* const kbuf = smem;
    char* const vbuf = smem + 2 * K_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int hi = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = blockIdx.x * 256 + wave * 32;

    const f16* kb = p.k + (size_t)b * p.k_bs + h * D;
    const f16* vb = p.vt + (size_t)b * p.vt_bs + (size_t)h * D * p.vt_ld;
    const __amdgpu_buffer_rsrc_t rs_k = make_rsrc(kb, p.k_bytes);
    const __amdgpu_buffer_rsrc_t rs_v = make_rsrc(vb, p.vt_bytes);

    f16x8 qf[KS];
    {
        const int qi = q0 + l31;
        const f16* qp = p.q + (size_t)b * p.q_bs + (size_t)qi * p.q_ld + h * D + hi * 8;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (qi < p.Nq && s * 16 + hi * 8 < D)
                qf[s] = *reinterpret_cast<const f16x8*>(qp + s * 16);
            else
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[s][e] = (f16)0.f;
        }
    }
    auto kkey = [](int row) { return KCH == 8 ? ((row >> 1) & 7) : (row & 15); };
    auto dma_k = [&](int t) {
        char* sb = kbuf + (t & 1) * K_BYTES;
        const int key0 = t * BKV;
#pragma unroll
        for (int j = 0; j < K_DMA; ++j) {
            const int inst = j * 8 + wave;
            if (inst >= K_INST) continue;
            const int row = inst * K_RPI + lane / KCH;
            const int chunk = (lane % KCH) ^ kkey(row);
            const int key = key0 + row;
            const unsigned off = (key < p.Nk && chunk * 8 < D) ? (unsigned)(((size_t)key * p.k_ld + chunk * 8) * 2) : MDX_OOB;
            dma16(rs_k, sb + inst * 1024, off);
        }
    };
    auto dma_v = [&](int t) {
        char* sb = vbuf + (t & 1) * V_BYTES;
        const int key0 = t * BKV;
#pragma unroll
        for (int j = 0; j < V_DMA; ++j) {
            const int inst = j * 8 + wave;
            if (inst >= V_INST) continue;
            const int row = inst * 8 + (lane >> 3);
            const unsigned chunk = (unsigned)((lane & 7) ^ ((row >> 1) & 7));
            const int kc = key0 + (int)chunk * 8;
            const unsigned off = (row < D && kc < p.vt_ld) ? (unsigned)(((size_t)row * p.vt_ld + kc) * 2) : MDX_OOB;
            dma16(rs_v, sb + inst * 1024, off);
        }
    };

    f32x16 acc_o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[d][r] = 0.f;
    f32x16 acc_s[2];
    f16x8 pf[4];
    float m_run = -INFINITY, l_run = 0.f;
    const int vswz = (lane >> 1) & 7;
    const int krow_l = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    const int T = p.Nk / BKV;

    // matrix phase of tile i: PV(i - 1) with the P fragments of the previous softmax phase, then S^T(i) = K(i) Q^T
    auto mphase = [&](int i) {
        if (i >= 1) {
            const char* sv = vbuf + ((i - 1) & 1) * V_BYTES;
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    const f16x8 vf = *reinterpret_cast<const f16x8*>(sv + (d * 32 + l31) * V_ROWB + (((2 * c + hi) ^ vswz) << 4));
                    acc_o[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[c], acc_o[d], 0, 0, 0);
                }
        }
        if (i < T) {
            const char* sk = kbuf + (i & 1) * K_BYTES;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_s[kt][r] = 0.f;
                const int krow = kt * 32 + krow_l;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const f16x8 kf = *reinterpret_cast<const f16x8*>(sk + krow * K_ROWB + (((2 * s + hi) ^ kkey(krow)) << 4));
                    acc_s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], acc_s[kt], 0, 0, 0);
                }
            }
        }
    };
    // softmax phase of tile i (the arithmetic of attn_kernel above, unchanged): reference maximum (moved only past 2^8), P in fp16
    auto vphase = [&](int i) {
        if (i < 0 || i >= T) return;
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
        if (__builtin_amdgcn_ballot_w64((m_cand - m_run) * p.scale_log2 > 8.0f)) {
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_cand) * p.scale_log2);
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_o[d][r] *= alpha;
            m_run = m_cand;
        }
        const float mb = m_run * p.scale_log2;
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 sc2 = {p.scale_log2, p.scale_log2}, nmb2 = {-mb, -mb};
        f32x2 psum2 = {0.f, 0.f};
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
    };

    dma_k(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = 0; i <= T; ++i) {
        // ---- phase 2i
        if (i + 1 < T) dma_k(i + 1);
        if (i < T) dma_v(i);
        __builtin_amdgcn_sched_barrier(0);
        if (grp == 0) {
            __builtin_amdgcn_s_setprio(1);
            mphase(i);
            __builtin_amdgcn_s_setprio(0);
        } else {
            vphase(i - 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ---- phase 2i + 1
        if (grp == 0) {
            vphase(i);
        } else {
            __builtin_amdgcn_s_setprio(1);
            mphase(i);
            __builtin_amdgcn_s_setprio(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the batch of phase 2i (this wave's share) has landed
        __builtin_amdgcn_s_barrier();
    }

    // ---- finalize (as attn_kernel): O /= l, per-wave staging in LDS, full-row stores
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    constexpr int OLD = DT * 32 + 8;
    f16* og = reinterpret_cast<f16*>(smem) + wave * 32 * OLD;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (f16)(acc_o[d][4 * g + e] * inv);
            *reinterpret_cast<f16x4*>(&og[l31 * OLD + d * 32 + 8 * g + 4 * hi]) = v;
        }
    __syncthreads();
    constexpr int CPR = D / 8;
    for (int idx = lane; idx < 32 * CPR; idx += 64) {
        const int row = idx / CPR, chunk = idx - row * CPR;
        const int qi = q0 + row;
        if (qi < p.Nq) {
            const f16x8 v = *reinterpret_cast<const f16x8*>(&og[row * OLD + chunk * 8]);
            *reinterpret_cast<f16x8*>(p.o + (size_t)b * p.o_bs + (size_t)qi * p.o_ld + h * D + chunk * 8) = v;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// attn_pipe (round 6): the four-wave flash attention with the tile loop SOFTWARE-PIPELINED inside every wave.
//
// attn_kernel runs a key tile as three dependent bursts -- 2 KS MFMAs of S^T = K Q^T, ~130 VALU instructions of softmax, 4 DT MFMAs
// of O^T += V^T P^T -- and leaves it to the other waves of the SIMD to fill the pipe that a burst does not use.  They do not: a wave
// issues one VALU instruction per ~7 clocks behind its own dependencies, the matrix pipe idles through every softmax and the VALU
// through every MFMA burst (PMC, profiles/r02_attention_pmc.md: VALU issue 58 % + matrix pipe 38 % of the SIMD cycles), and the
// eight-wave ping-pong (attn8 above) showed that coupling two waves with barriers does not buy the overlap either.  Here the overlap
// is built INSIDE the wave's instruction stream: the iteration of key tile t issues
//     QK^T of tile t + 1   (its scores wait in a second accumulator pair)
//     softmax of tile t    (scores computed one iteration earlier; its row maximum was taken at the END of that iteration)
//     PV of tile t
// as 2 KS + 4 DT SLOTS of { one MFMA, ~6 independent VALU instructions } fenced with sched_barriers -- every MFMA has VALU work of an
// older tile to run in its shadow and every exponential has an MFMA in flight -- plus the LDS fragment reads one or two slots ahead
// of their MFMA.  K is staged one tile ahead of V (K(t + 2) and V(t + 1) are issued in iteration t): the same two-deep LDS ring.
// Same arithmetic in the same order per output element as attn_kernel (the lazy reference maximum included): BIT-IDENTICAL results.
// Full key tiles only (Nk % 64 == 0), no causal mask; unsplit or split-KV: the self-attention launches of the UNets (4096 / 9216 / 1024 /
// 256 tokens; GLIDE's text | image keys); everything else stays on attn_kernel.
#ifndef MDX_ATTN_PIPE_PKASM
#define MDX_ATTN_PIPE_PKASM 0
#endif
template <class F, int... I>
__device__ __forceinline__ void attn_static_for(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}

template <int D>
__global__ __launch_bounds__(256, 2) void attn_pipe_kernel(const AttnParams p) {
    mdx_kernarg_touch<sizeof(AttnParams)>();
    constexpr int KS = (D + 15) / 16;
    constexpr int DT = (D + 31) / 32;
    constexpr int KCH = (2 * KS <= 8) ? 8 : (2 * KS <= 16 ? 16 : 32);
    constexpr int K_ROWB = KCH * 16;
    constexpr int K_RPI = 64 / KCH;
    constexpr int K_DMA = BKV / K_RPI / 4;
    constexpr int K_BYTES = BKV * K_ROWB;
    constexpr int V_ROWB = 128;
    constexpr int V_ROWS = DT * 32;
    constexpr int V_DMA = (V_ROWS / 8 + 3) / 4;
    constexpr int V_BYTES = V_ROWS * V_ROWB;
    constexpr bool LROW = (D % 32) != 0;       // row sums through the ones row of the V^T tile (attn_kernel)
    constexpr int NQ = 2 * KS;                 // QK^T MFMAs of a tile
    constexpr int NP = 4 * DT;                 // PV MFMAs of a tile
    extern __shared__ __attribute__((aligned(16))) char smem[];      // K[2] | V^T[2]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z, h = blockIdx.y;
    // split-KV launches: blockIdx.x = query block * nsplit + split (attn_kernel's order); this block's tiles [t_begin, t_end), even boundaries
    const int qblk = p.nsplit > 1 ? (int)blockIdx.x / p.nsplit : (int)blockIdx.x;
    const int split = p.nsplit > 1 ? (int)blockIdx.x - qblk * p.nsplit : 0;
    const int q0 = qblk * BQ + wave * 32;
    const int ntiles = p.Nk / BKV;
    int t_begin = 0, t_end = ntiles;
    if (p.nsplit > 1) {
        t_begin = ((split * ntiles) / p.nsplit) & ~1;
        t_end = split + 1 == p.nsplit ? ntiles : ((((split + 1) * ntiles) / p.nsplit) & ~1);
    }
    const int nt = t_end - t_begin;      // >= 2 (the host splits no finer than two tiles each)

    const f16* kb = p.k + (size_t)b * p.k_bs + h * D;
    const f16* vb = p.vt + (size_t)b * p.vt_bs + (size_t)h * D * p.vt_ld;
    const __amdgpu_buffer_rsrc_t rs_k = make_rsrc(kb, p.k_bytes);
    const __amdgpu_buffer_rsrc_t rs_v = make_rsrc(vb, p.vt_bytes);

    f16x8 qf[KS];
    {
        const int qi = q0 + l31;
        const f16* qp = p.q + (size_t)b * p.q_bs + (size_t)qi * p.q_ld + h * D + hi * 8;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (qi < p.Nq && s * 16 + hi * 8 < D)
                qf[s] = *reinterpret_cast<const f16x8*>(qp + s * 16);
            else
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[s][e] = (f16)0.f;
        }
    }
    auto kkey = [](int row) { return KCH == 8 ? ((row >> 1) & 7) : (row & 15); };
    // every tile is full: per-lane source offsets once, the tile offset in the DMA's scalar operand (attn_kernel's stage_full)
    unsigned k_voff[K_DMA], v_voff[V_DMA];
#pragma unroll
    for (int j = 0; j < K_DMA; ++j) {
        const int row = (wave * K_DMA + j) * K_RPI + lane / KCH;
        const int chunk = (lane % KCH) ^ kkey(row);
        k_voff[j] = (chunk * 8 < D) ? (unsigned)(((size_t)row * p.k_ld + chunk * 8) * 2) : MDX_OOB;
    }
#pragma unroll
    for (int j = 0; j < V_DMA; ++j) {
        const int row = (wave * V_DMA + j) * 8 + (lane >> 3);
        const unsigned chunk = (unsigned)((lane & 7) ^ ((row >> 1) & 7));
        v_voff[j] = (row < D) ? (unsigned)(((size_t)row * p.vt_ld + chunk * 8) * 2) : MDX_OOB;
    }
    const unsigned k_tile_bytes = (unsigned)BKV * (unsigned)p.k_ld * 2u;
    auto stage_k = [&](int t, int buf) {
#pragma unroll
        for (int j = 0; j < K_DMA; ++j)
            dma16s(rs_k, smem + buf * K_BYTES + (wave * K_DMA + j) * 1024, k_voff[j], (unsigned)(t_begin + t) * k_tile_bytes);
    };
    auto stage_v = [&](int t, int buf) {
#pragma unroll
        for (int j = 0; j < V_DMA; ++j) {
            const int inst = wave * V_DMA + j;
            if (inst * 8 < (LROW ? D : V_ROWS))
                dma16s(rs_v, smem + 2 * K_BYTES + buf * V_BYTES + inst * 1024, v_voff[j], (unsigned)(t_begin + t) * (BKV * 2u));
        }
    };

    f32x16 acc_o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const int vswz = (lane >> 1) & 7;
    const int krow_l = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4a __attribute__((ext_vector_type(4)));
    f32x2 sc2v = {p.scale_log2, p.scale_log2};      // in VECTOR registers: with the scale in SGPRs the compiler splits every v_pk_fma_f32 in two v_fma_f32
    asm volatile("" : "+v"(sc2v));

    // K fragment s of 32-key tile kt (LDS buffer kb_i) / V^T fragment of chunk c, d tile d (buffer vb_i)
    auto kfrag = [&](const int kb_i, const int kt, const int s) {
        const int krow = kt * 32 + krow_l;
        return *reinterpret_cast<const f16x8*>(smem + kb_i * K_BYTES + krow * K_ROWB + (((2 * s + hi) ^ kkey(krow)) << 4));
    };
    auto vfrag = [&](const int vb_i, const int c, const int d) {
        return *reinterpret_cast<const f16x8*>(smem + 2 * K_BYTES + vb_i * V_BYTES + (d * 32 + l31) * V_ROWB + (((2 * c + hi) ^ vswz) << 4));
    };
    // row maximum of a score pair (this lane's 32 keys, then the lane ^ 32 exchange) -- attn_kernel's order
    auto row_max = [&](const f32x16 (&sc)[2]) {
        float mx = sc[0][0];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[kt][r]);
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    };

    if constexpr (LROW) {
        for (int i = tid; i < 2 * (V_ROWS - D) * 8; i += 256) {
            const int buf = i / ((V_ROWS - D) * 8), rem = i - buf * (V_ROWS - D) * 8;
            const int row = D + (rem >> 3);
            f16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = row == D ? (f16)1.0f : (f16)0.f;
            *reinterpret_cast<f16x8*>(smem + 2 * K_BYTES + buf * V_BYTES + row * V_ROWB + (rem & 7) * 16) = v;
        }
    }
    // ---- prologue: K(0), V(0); then K(1) in flight under S(0) = K(0) Q^T and its row maximum
    stage_k(0, 0);
    stage_v(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (nt > 1) stage_k(1, 1);
    f32x16 sA[2], sB[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sA[kt][r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) sA[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfrag(0, kt, s), qf[s], sA[kt], 0, 0, 0);
    }
    float mx_cur = row_max(sA);

    // One iteration: softmax + PV of the tile whose scores are `sc` (V^T in buffer VB), QK^T of the next tile (K in buffer KBN) into
    // `sn` when NEXT.  mx_cur = row maximum of sc on entry, of sn on exit.
    auto body = [&](auto kbn_c, auto vb_c, auto next_c, f32x16 (&sc)[2], f32x16 (&sn)[2]) {
        constexpr int KBN = decltype(kbn_c)::value;
        constexpr int VB = decltype(vb_c)::value;
        constexpr bool NEXT = decltype(next_c)::value;
        // reference maximum (attn_kernel's lazy rule), outside the slots: the rescale is a rare branch
        const float m_cand = fmaxf(m_run, mx_cur);
        if (__builtin_amdgcn_ballot_w64((m_cand - m_run) * p.scale_log2 > 8.0f)) {
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_cand) * p.scale_log2);
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_o[d][r] *= alpha;
            m_run = m_cand;
        }
        const float mb = m_run * p.scale_log2;
        const f32x2 nmb2 = {-mb, -mb};
        f32x2 psum2 = {0.f, 0.f};
        unsigned pw[16];       // P as packed fp16 pairs: pw[4 c .. 4 c + 3] = the B fragment of 16-key chunk c
        // pair j (0 .. 15) of the tile: scores sc[j >> 3][2 (j & 7)], [.. + 1] -> two weights of P chunk j >> 2.  The empty asm
        // statements pin every result INSIDE the slot it is written in: without them the row-sum adds (no consumer before the end
        // of the iteration) and the exponentials (consumed slots later) are sunk to their uses, back into a serial burst
        auto pair = [&](const int j) {
            const int kt = j >> 3, r = 2 * (j & 7);
            const f32x2 x2 = {sc[kt][r], sc[kt][r + 1]};
            f32x2 a2;      // (the compiler splits the builtin in two v_fma_f32; v_pk_fma_f32 through inline asm measured SLOWER -- 9216 tokens 951 -> 1014 us,
                           // profiles/r06_attn_pipe.txt: the asm is opaque to the hazard / latency model the slots are scheduled with)
#if MDX_ATTN_PIPE_PKASM
            asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a2) : "v"(x2), "v"(sc2v), "v"(nmb2));
#else
            a2 = __builtin_elementwise_fma(x2, sc2v, nmb2);
#endif
            const f32x2 pv2 = {__builtin_amdgcn_exp2f(a2.x), __builtin_amdgcn_exp2f(a2.y)};
            if constexpr (!LROW) {
                psum2 += pv2;
                asm volatile("" : "+v"(psum2));
            }
            typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
            const h16x2 hv = {(f16)pv2.x, (f16)pv2.y};
            pw[j] = __builtin_bit_cast(unsigned, hv);
            asm volatile("" : "+v"(pw[j]));
        };
        auto pfrag = [&](const int c) {
            u32x4a v;
            v[0] = pw[4 * c]; v[1] = pw[4 * c + 1]; v[2] = pw[4 * c + 2]; v[3] = pw[4 * c + 3];
            return __builtin_bit_cast(f16x8, v);
        };
        // Slots.  0 .. NQ - 1: QK^T MFMA i (k-step i >> 1 of key half i & 1: the two accumulator chains alternate) over pairs 0 .. 7.
        // NQ .. NQ + NP - 1: PV MFMA ip (chunk ip / DT, d tile ip % DT); pairs 8 .. 15 over the first NP - DT of them (chunk 2 is
        // complete before slot NQ + 2 DT, chunk 3 before NQ + 3 DT) and the next tile's row maximum, 32 / NP scores per slot.
        // LDS fragments are read TWO slots ahead of their MFMA (a read issued one slot ahead is still in flight when it is needed).
        float mx_part = -INFINITY;
        f16x8 fr[3] = {};        // fragment of slot i sits in fr[i % 3]
        auto frag_of = [&](const int i) {      // the LDS fragment slot i multiplies (K of the NEXT tile, or V^T of this one)
            if (i < NQ) return kfrag(KBN, i & 1, i >> 1);
            return vfrag(VB, (i - NQ) / DT, (i - NQ) % DT);
        };
        constexpr int I0 = NEXT ? 0 : NQ;       // first slot that exists
        if constexpr (!NEXT) {                  // last tile: no QK^T slots -- their pairs first
#pragma unroll
            for (int j = 0; j < 8; ++j) pair(j);
        }
        fr[I0 % 3] = frag_of(I0);
        fr[(I0 + 1) % 3] = frag_of(I0 + 1);
        attn_static_for([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i >= I0) {
                if constexpr (i + 2 < NQ + NP) fr[(i + 2) % 3] = frag_of(i + 2);
                if constexpr (i < NQ) {
                    if constexpr (i < 2)      // first MFMA of a chain: C = 0 (an inline constant, no accumulator initialisation)
                        sn[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[i % 3], qf[i >> 1], f32x16{}, 0, 0, 0);
                    else
                        sn[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[i % 3], qf[i >> 1], sn[i & 1], 0, 0, 0);
#pragma unroll
                    for (int j = (i * 8) / NQ; j < ((i + 1) * 8) / NQ; ++j) pair(j);
                } else {
                    constexpr int ip = i - NQ;
                    acc_o[ip % DT] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[i % 3], pfrag(ip / DT), acc_o[ip % DT], 0, 0, 0);
#pragma unroll
                    for (int j = 8; j < 16; ++j)
                        if (((j - 8) * (NP - DT)) / 8 == ip) pair(j);
                    if constexpr (NEXT) {
#pragma unroll
                        for (int e = (ip * 32) / NP; e < ((ip + 1) * 32) / NP; ++e) mx_part = fmaxf(mx_part, sn[e >> 4][e & 15]);
                        asm volatile("" : "+v"(mx_part));
                    }
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // the slot's MFMA first ...
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // ... the LDS read of the fragment two slots on ...
                __builtin_amdgcn_sched_group_barrier(0x002, 64, 0);     // ... then its VALU work, in the MFMA's shadow
                __builtin_amdgcn_sched_barrier(0);
            }
        }, std::make_integer_sequence<int, NQ + NP>{});
        if constexpr (!LROW) l_run += psum2.x + psum2.y;
        if constexpr (NEXT) {      // lane <-> lane ^ 32 exchange of the partial maximum (attn_kernel's one-instruction swap)
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx_part), __float_as_uint(mx_part), false, false);
            mx_cur = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    // iteration t: K(t + 1) [buffer (t + 1) & 1] and V(t) [buffer t & 1] landed; K(t + 2) -> buffer t & 1, V(t + 1) -> buffer (t + 1) & 1
    int t = 0;
    for (; t + 2 < nt; t += 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stage_k(t + 2, 0);
        stage_v(t + 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        body(C1{}, C0{}, std::true_type{}, sA, sB);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 3 < nt) stage_k(t + 3, 1);
        stage_v(t + 2, 0);
        __builtin_amdgcn_sched_barrier(0);
        body(C0{}, C1{}, std::true_type{}, sB, sA);
    }
    // the last one or two tiles (scores of tile t in sA)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 1 < nt) {
        stage_v(t + 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        body(C1{}, C0{}, std::true_type{}, sA, sB);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        body(C0{}, C1{}, std::false_type{}, sB, sA);
    } else {
        body(C1{}, C0{}, std::false_type{}, sA, sB);
    }
    __syncthreads();

    attn_finish<D>(p, acc_o, m_run, l_run, smem, b, h, qblk, q0, split);
}

template <int D>
void launch_attn_pipe(const AttnParams& p, dim3 grid, hipStream_t st) {
    constexpr int KS = (D + 15) / 16, DT = (D + 31) / 32;
    constexpr int KCH = (2 * KS <= 8) ? 8 : (2 * KS <= 16 ? 16 : 32);
    constexpr size_t ring = 2 * ((size_t)BKV * KCH * 16 + (size_t)DT * 32 * 128);
    constexpr size_t ostage = (size_t)4 * 32 * (DT * 32 + 8) * 2 + 16;      // + the split-KV "last arriver" flag
    constexpr size_t lds = ring > ostage ? ring : ostage;
    static MdxPerDeviceOnce attr_once;
    if (attr_once.first())
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_pipe_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(attn_pipe_kernel<D>, grid, dim3(256), lds, st, p);
}

template <int D>
void launch_attn8(const AttnParams& p, dim3 grid, hipStream_t st) {
    constexpr int KS = (D + 15) / 16, DT = (D + 31) / 32;
    constexpr int KCH = (2 * KS <= 8) ? 8 : (2 * KS <= 16 ? 16 : 32);
    constexpr size_t ring = 2 * ((size_t)BKV * KCH * 16 + (size_t)DT * 32 * 128);
    constexpr size_t ostage = (size_t)8 * 32 * (DT * 32 + 8) * 2;
    constexpr size_t lds = ring > ostage ? ring : ostage;
    static MdxPerDeviceOnce attr_once;
    if (attr_once.first())
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn8_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(attn8_kernel<D>, grid, dim3(512), lds, st, p);
}

template <int D, int OCC>
void launch_attn(const AttnParams& p, dim3 grid, hipStream_t st) {
    constexpr int KS = (D + 15) / 16, DT = (D + 31) / 32;
    constexpr int KCH = (2 * KS <= 8) ? 8 : (2 * KS <= 16 ? 16 : 32);
    constexpr size_t stage = (size_t)BKV * KCH * 16 + (size_t)DT * 32 * 128;
    constexpr size_t ostage = (size_t)4 * 32 * (DT * 32 + 8) * 2 + 16;      // + the split-KV "last arriver" flag
    constexpr size_t lds = (2 * stage > ostage ? 2 * stage : ostage);
    static MdxPerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kernel<D, OCC>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL((attn_kernel<D, OCC>), grid, dim3(256), lds, st, p);
}

constexpr int ATTN_MAX_SPLITS = ATTN_MAX_SPLITS_K;

// CUs of the current device (256 on MI355X; read once per device so that the fill policy below follows the part it runs on)
int attn_cus() {
    static int cus[64] = {};
    int dv = 0;
    if (hipGetDevice(&dv) != hipSuccess || dv < 0 || dv >= 64) return 256;
    if (!cus[dv]) {
        int n = 0;
        cus[dv] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dv) == hipSuccess && n > 0) ? n : 256;
    }
    return cus[dv];
}

// blocks the register file admits per CU for head dim D (see attn_kernel's OCC)
int attn_blocks_per_cu(int D) { return D <= 64 ? (mdx_opt(MDX_OPT_ATTN_OCC3) ? 3 : 2) : (D <= 80 ? 2 : 1); }

// Split-KV policy.  A launch whose (batch, head, query block) items do not fill the chip's block slots even once leaves SIMDs idle
// while others carry two waves: SDv2's 64 x 64 self-attention at UNet batch 2 is 320 blocks of 64 tiles on 256 CUs -- 64 CUs run
// two blocks (82 us), 192 CUs one (done at ~50 us).  Dealing every item's key tiles to S blocks (S = slots / items, each split at
// least 24 tiles) fills the slots with equal pieces; the partial hand-off costs one 128 x D fp16 tile per block.
int attn_auto_splits(int B, int heads, int D, int Nq, int Nk) {
    const int mode = mdx_opt(MDX_OPT_ATTN_KV_SPLIT);
    const int ntiles = (Nk + BKV - 1) / BKV;
    // the hand-off is a chain of fabric round trips (drained write-through stores -> ticket -> partial loads: ~6-8 us on MI355X), so a
    // split must be worth >= 24 key tiles (~25 us): 4096 keys split two ways 72 -> 66 us, 1024 keys split four ways 15.5 -> 24 us
    // (tools/attn_bench.py, profiles/r04_attn_bench.txt)
    int cap = mode >= 2 ? ntiles / 2 : ntiles / 24;
    if (cap > ATTN_MAX_SPLITS) cap = ATTN_MAX_SPLITS;
    if (mode == 0 || cap < 2) return 1;
    if (mode >= 2) return mode < cap ? mode : cap;
    const long items = (long)((Nq + BQ - 1) / BQ) * heads * B;
    const long slots = (long)attn_cus() * attn_blocks_per_cu(D);
    long s = slots / items;
    if (s > cap) s = cap;
    return s < 2 ? 1 : (int)s;
}

// The arrival counters live in a FIXED region at the head of the workspace: launches of different shapes share one workspace in turn,
// and a region sized per launch would let a small launch's partials land on a bigger launch's counters.
constexpr long ATTN_TICKET_ITEMS = 16384;
constexpr size_t ATTN_TICKET_BYTES = (size_t)ATTN_TICKET_ITEMS * 4;
size_t attn_part_bytes(int D) { return (size_t)BQ * D * 2 + (size_t)BQ * 8; }

}  // namespace

static int attention_impl(const void* q, long q_bs, int q_ld, const void* k, long k_bs, int k_ld, const void* vt,
                          long vt_bs, int vt_ld, void* o, long o_bs, int o_ld, int B, int heads, int D, int Nq, int Nk,
                          float scale, int causal, int kv_splits, void* ws, size_t ws_bytes, mdx_stream_t s) {
    MDX_REQUIRE(q && k && vt && o, "mdx_attention_f16: null pointer");
    MDX_REQUIRE(D == 40 || D == 64 || D == 80 || D == 160, "mdx_attention_f16: head dim %d not supported (40/64/80/160)", D);
    MDX_REQUIRE(B > 0 && heads > 0 && Nq > 0 && Nk > 0, "mdx_attention_f16: bad extents");
    MDX_REQUIRE(q_ld % 8 == 0 && k_ld % 8 == 0 && vt_ld % 8 == 0 && o_ld % 8 == 0, "mdx_attention_f16: strides must be multiples of 8");
    MDX_REQUIRE(vt_ld >= Nk, "mdx_attention_f16: vt_ld < Nk");
    AttnParams p{};
    p.q = (const f16*)q;
    p.k = (const f16*)k;
    p.vt = (const f16*)vt;
    p.o = (f16*)o;
    p.q_bs = q_bs; p.k_bs = k_bs; p.vt_bs = vt_bs; p.o_bs = o_bs;
    p.q_ld = q_ld; p.k_ld = k_ld; p.vt_ld = vt_ld; p.o_ld = o_ld;
    p.B = B; p.heads = heads; p.Nq = Nq; p.Nk = Nk;
    p.scale_log2 = scale * 1.4426950408889634f;
    p.causal = causal;
    const size_t kbytes = ((size_t)(Nk - 1) * k_ld + D) * 2;
    const size_t vbytes = ((size_t)(D - 1) * vt_ld + vt_ld) * 2;
    MDX_REQUIRE(kbytes <= 0x80000000ull && vbytes <= 0x80000000ull, "mdx_attention_f16: K/V extent too large");
    p.k_bytes = (unsigned)kbytes;
    p.vt_bytes = (unsigned)vbytes;
    const int qblocks = (Nq + BQ - 1) / BQ;
    p.nsplit = 1;
    p.qblocks = qblocks;
    if (ws && !causal) {
        const int ntiles = (Nk + BKV - 1) / BKV;
        int S = kv_splits > 0 ? kv_splits : attn_auto_splits(B, heads, D, Nq, Nk);
        if (kv_splits > 0) {
            MDX_REQUIRE(S <= ATTN_MAX_SPLITS && 2 * S <= ntiles, "mdx_attention_splitkv_f16: %d splits of %d key tiles (at most %d, two tiles each)",
                        S, ntiles, ATTN_MAX_SPLITS);
        }
        const long items = (long)qblocks * heads * B;
        if (kv_splits <= 0 && S > 1) {
            // auto mode: the split is a fill policy, not a request -- it depends on options (attn_occ3, attn_kv_split) that may have
            // changed since the caller sized its workspace with mdx_attention_ws_bytes, so take what the workspace and the ticket
            // region allow and fall back to the unsplit launch (always correct) instead of failing every evaluation
            if (items > ATTN_TICKET_ITEMS || ws_bytes < ATTN_TICKET_BYTES) {
                S = 1;
            } else {
                const size_t fit = (ws_bytes - ATTN_TICKET_BYTES) / ((size_t)items * attn_part_bytes(D));
                if ((size_t)S > fit) S = fit < 2 ? 1 : (int)fit;
            }
        }
        if (S > 1) {
            MDX_REQUIRE(items <= ATTN_TICKET_ITEMS, "mdx_attention_splitkv_f16: %ld (batch, head, query block) items, at most %ld can be split", items, ATTN_TICKET_ITEMS);
            const size_t tb = ATTN_TICKET_BYTES, need = tb + (size_t)items * S * attn_part_bytes(D);
            MDX_REQUIRE(ws_bytes >= need, "mdx_attention_splitkv_f16: workspace of %zu bytes, %zu needed (mdx_attention_ws_bytes)", ws_bytes, need);
            MDX_REQUIRE((size_t)S * attn_part_bytes(D) <= 0x80000000ull, "mdx_attention_splitkv_f16: partial extent too large");
            p.nsplit = S;
            p.tickets = (unsigned*)ws;
            p.ws_part = (char*)ws + tb;
        }
    }
    p.fast_stage = mdx_opt(MDX_OPT_ATTN_FAST_STAGE) ? 1 : 0;
    dim3 grid(qblocks * p.nsplit, heads, B);
    hipStream_t st = (hipStream_t)s;
    // eight-wave form (attn8_kernel above; measured slower, opt-in): self-attention shapes, D <= 80 (D = 160 spills at two waves per SIMD)
    const int a8 = mdx_opt(MDX_OPT_ATTN8);
    if (a8 && p.nsplit == 1 && !causal && D <= 80 && Nq % 256 == 0 && Nk % BKV == 0 &&
        (a8 == 2 || (long)(Nq / 256) * heads * B >= mdx_opt(MDX_OPT_ATTN8_MIN_BLOCKS))) {
        const dim3 g8(Nq / 256, heads, B);
        switch (D) {
            case 40: launch_attn8<40>(p, g8, st); break;
            case 64: launch_attn8<64>(p, g8, st); break;
            default: launch_attn8<80>(p, g8, st); break;
        }
        MDX_LAUNCH_CHECK("mdx_attention_f16(attn8)");
        return MDX_OK;
    }
    // software-pipelined form (attn_pipe_kernel): full key tiles, no mask, unsplit, at least two key tiles
    if (mdx_opt(MDX_OPT_ATTN_PIPE) && !causal && D <= 80 && Nk % BKV == 0 && Nk >= 2 * BKV) {      // (split launches: >= 2 tiles per split, checked above)
        switch (D) {
            case 40: launch_attn_pipe<40>(p, grid, st); break;
            case 64: launch_attn_pipe<64>(p, grid, st); break;
            default: launch_attn_pipe<80>(p, grid, st); break;
        }
        MDX_LAUNCH_CHECK("mdx_attention_f16(pipe)");
        return MDX_OK;
    }
    const bool occ3 = mdx_opt(MDX_OPT_ATTN_OCC3) != 0;
    switch (D) {
        case 40: if (occ3) launch_attn<40, 3>(p, grid, st); else launch_attn<40, 2>(p, grid, st); break;
        case 64: if (occ3) launch_attn<64, 3>(p, grid, st); else launch_attn<64, 2>(p, grid, st); break;
        case 80: launch_attn<80, 2>(p, grid, st); break;
        default: launch_attn<160, 1>(p, grid, st); break;
    }
    MDX_LAUNCH_CHECK("mdx_attention_f16");
    return MDX_OK;
}

/* Workspace the split-KV form needs for this shape under the current options (0: the launch would not be split). */
extern "C" size_t mdx_attention_ws_bytes(int B, int heads, int D, int Nq, int Nk) {
    if (B <= 0 || heads <= 0 || Nq <= 0 || Nk <= 0 || !(D == 40 || D == 64 || D == 80 || D == 160)) return 0;
    const int S = attn_auto_splits(B, heads, D, Nq, Nk);
    if (S < 2) return 0;
    const long items = (long)((Nq + BQ - 1) / BQ) * heads * B;
    return ATTN_TICKET_BYTES + (size_t)items * S * attn_part_bytes(D);
}

extern "C" int mdx_attention_splitkv_f16(const void* q, long q_bs, int q_ld, const void* k, long k_bs, int k_ld,
                                         const void* vt, long vt_bs, int vt_ld, void* o, long o_bs, int o_ld, int B, int heads,
                                         int D, int Nq, int Nk, float scale, int kv_splits, void* ws, size_t ws_bytes,
                                         mdx_stream_t s) {
    MDX_REQUIRE(kv_splits >= 0, "mdx_attention_splitkv_f16: kv_splits < 0");
    MDX_REQUIRE(ws || kv_splits <= 1, "mdx_attention_splitkv_f16: %d splits need a workspace", kv_splits);
    return attention_impl(q, q_bs, q_ld, k, k_bs, k_ld, vt, vt_bs, vt_ld, o, o_bs, o_ld, B, heads, D, Nq, Nk, scale, 0,
                          kv_splits == 1 ? 0 : kv_splits, kv_splits == 1 ? nullptr : ws, ws_bytes, s);
}

extern "C" int mdx_attention_f16(const void* q, long q_bs, int q_ld, const void* k, long k_bs, int k_ld,
                                 const void* vt, long vt_bs, int vt_ld, void* o, long o_bs, int o_ld, int B, int heads,
                                 int D, int Nq, int Nk, float scale, mdx_stream_t s) {
    return attention_impl(q, q_bs, q_ld, k, k_bs, k_ld, vt, vt_bs, vt_ld, o, o_bs, o_ld, B, heads, D, Nq, Nk, scale, 0, 0, nullptr, 0, s);
}

extern "C" int mdx_attention_causal_f16(const void* q, long q_bs, int q_ld, const void* k, long k_bs, int k_ld,
                                        const void* vt, long vt_bs, int vt_ld, void* o, long o_bs, int o_ld, int B,
                                        int heads, int D, int Nq, int Nk, float scale, mdx_stream_t s) {
    MDX_REQUIRE(Nq == Nk, "mdx_attention_causal_f16: causal self-attention needs Nq == Nk (got %d, %d)", Nq, Nk);
    return attention_impl(q, q_bs, q_ld, k, k_bs, k_ld, vt, vt_bs, vt_ld, o, o_bs, o_ld, B, heads, D, Nq, Nk, scale, 1, 0, nullptr, 0, s);
}
