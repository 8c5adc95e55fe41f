// Row-local fused tail of a SpatialTransformer block for gfx950 (MI355X): everything BasicTransformerBlock does after
// the self-attention core, plus proj_out, as ONE launch.
//
//   t1  = attn1_out  Wo1^T + bo1 + tok                 (attention.py:151-152 to_out, :182 residual)
//   q2  = LN2(t1)    Wq2^T                             (:177 norm2, :108 to_q)
//   o2  = softmax(q2 K_ctx^T * scale) V_ctx            (:138-150 over the <= 80 cached context keys, per head)
//   t2  = o2         Wo2^T + bo2 + t1                  (:183)
//   h   = GEGLU(LN3(t2) W1^T + b1)                     (:41-51, :178)
//   t3  = h          W2^T + b2 + t2                    (:66, :184)
//   out = t3         Wpo^T + bpo + x_in                (:231 proj_out, :256 residual)
//
// Every stage is row-local, so a block owns BM = 32 * TM token rows for the whole chain:
//  * the A operand of every GEMM lives in LDS ([BM][C] fp16, row stride C + 8 halves: 16 consecutive rows cover all
//    64 banks exactly once for ds_read_b128 / ds_write_b64) -- no activation goes through HBM between the stages: the
//    unfused chain wrote and re-read q2, o2, t1, t2, the [M][4C] GEGLU output and t3;
//  * wave w of the C / 32 waves owns output columns [32w, 32w + 32) of every GEMM, so no two waves share a weight
//    element and weights go HBM/L2 -> VGPR directly in MFMA fragment order (1 KiB per wave-instruction, fully
//    coalesced; ops.pack_st_tail packs each wave's whole chain as ONE contiguous stream), PF pieces in flight per wave
//    across stage boundaries and barriers (raw s_barrier: a __syncthreads() would drain them);
//  * accumulators hold C^T (lane = token row, 4 consecutive columns per register group): bias / residual / LayerNorm /
//    GEGLU run on registers, LayerNorm statistics cross the waves through 8 bytes of LDS per (wave, row);
//  * the FFN streams the hidden dimension in chunks of C columns: GEGLU(chunk) -> LDS -> partial ff2 accumulate, the
//    [M][4C] intermediate never exists;
//  * cross-attention: one (head, 32-query tile) per wave, K / V^T fragments straight from the cached context
//    projections (L2-resident, 77 keys), S^T = K Q^T so a lane owns one query (softmax = one lane^32 exchange), P feeds
//    PV from the S^T accumulator registers (key permutation trick of attention.hip).
// Built for C = 320 (heads x head dim 5 x 64 or 8 x 40): the 64 x 64 level of SDv2 / Wukong (96 x 96 at 768^2), where M / BM fills
// the chip; at C = 640 / 1280 the per-block weight stream (13 / 52 MB) costs more than the launches it saves.
#include "mdx_common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct StTailParams {
    const f16* attn_o;
    const f16* tok;
    const f16* x_in;
    f16* out;
    const f16* kc;
    const f16* vtc;
    const char* wstream;
    const float* vec;
    float* colstats_out;
    f16* dbg;
    int M, HW, TC, ctx_len, heads, nshare, lead;
    float scale_log2, eps;
    int stop_after;
    unsigned wstream_bytes, kc_bytes, vtc_bytes;
};

constexpr int PF = 10;   // weight pieces (1 KiB per wave) in flight per wave

__device__ __forceinline__ void lds_barrier() {
    // own LDS writes complete, then the block barrier; VMEM (the weight ring) stays in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Every block barrier of the compute waves goes through this macro: it also counts them (one SALU add), so that the barrier schedules
// the warmer wave walks (make_sched / make_head_sched, hand-mirrored) can be CHECKED against what the compute waves execute:
// debug_stage = MDX_ST_DEBUG_COUNT_BARRIERS runs the whole chain without the warmer and writes the count to debug_out
// (tests/test_stchain_gpu.py::test_warmer_schedules_match_the_compute_waves).  A drift would deadlock the product launch.
#define LDS_BARRIER() do { lds_barrier(); ++nbar; } while (0)
constexpr int ST_COUNT_BARRIERS = 100;      // = MDX_ST_DEBUG_COUNT_BARRIERS (mdx.h)

__device__ __forceinline__ u32x4 wload(const __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned piece) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, voff, piece * 1024u, 0);
}

// One "unit": T column tiles x TM row tiles x (C / 16) k-steps.  A from LDS, weights from the ring.
template <int C, int TM, int T>
__device__ __forceinline__ void unit(f32x16 (&acc)[T][TM], const char* abuf_lane, u32x4 (&ring)[PF],
                                     const __amdgpu_buffer_rsrc_t rs, const unsigned voff, unsigned& piece) {
    constexpr int KS = C / 16;
    constexpr int LDB = (C + 8) * 2;
    static_assert((KS * T) % PF == 0, "ring position must be 0 at every unit boundary");
    // The schedule is pinned with sched_barrier(0) per k-step: left alone, the machine scheduler SINKS every ring refill down to
    // just before its use PF steps later (shorter live range) -- two loads in flight instead of PF, and the whole chain runs
    // at the L2 / HBM round-trip time per k-step (measured: 93 -> us per block).  A fragments are read AD steps ahead by hand.
    constexpr int AD = 2;
    f16x8 af[AD + 1][TM];
    f32x16 odd;
#pragma unroll
    for (int r = 0; r < 16; ++r) odd[r] = 0.f;
#pragma unroll
    for (int s = 0; s < AD; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i) af[s][i] = *reinterpret_cast<const f16x8*>(abuf_lane + i * 32 * LDB + s * 32);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if (s + AD < KS) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[(s + AD) % (AD + 1)][i] = *reinterpret_cast<const f16x8*>(abuf_lane + i * 32 * LDB + (s + AD) * 32);
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int j = s * T + t;
            const f16x8 b = __builtin_bit_cast(f16x8, ring[j % PF]);
            ring[j % PF] = wload(rs, voff, piece + j + PF);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (T * TM == 1) {
                    // a lone accumulator makes every MFMA wait for the one before it (SQ_WAIT_INST_ANY 33 % of the wave cycles
                    // in the 32-row kernel): odd k-steps go to a second accumulator, summed at the end
                    if (s & 1)
                        odd = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, af[s % (AD + 1)][i], odd, 0, 0, 0);
                    else
                        acc[t][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, af[s % (AD + 1)][i], acc[t][i], 0, 0, 0);
                } else {
                    acc[t][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, af[s % (AD + 1)][i], acc[t][i], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (T * TM == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] += odd[r];
    }
    piece += KS * T;
}

template <int T, int TM>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[T][TM]) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][i][r] = 0.f;
}

// vec layout (fp32): [bo1 | g2 | be2 | bo2 | g3 | be3 | b1 (8C: a | gate) | b2 | bpo]
template <int C>
struct VecOff {
    static constexpr int bo1 = 0, g2 = C, be2 = 2 * C, bo2 = 3 * C, g3 = 4 * C, be3 = 5 * C, b1 = 6 * C, b2 = 14 * C,
                         bpo = 15 * C, total = 16 * C;
};

// The block barriers of the compute path, in order, with the number of weight pieces (position in a wave's stream) the
// compute waves have consumed when they arrive at each: the warmer wave executes exactly this sequence.  KEEP IN STEP with the
// lds_barrier() calls of st_tail_kernel (the debug taps add barriers, and run without the warmer).
struct BarSched {
    int n;
    int pos[24];
};
template <int KS, int NH>
constexpr BarSched make_sched() {
    BarSched s{};
    int n = 0, c = 0;
    s.pos[n++] = c;                              // S0: inputs in LDS
    c += KS; s.pos[n++] = c; s.pos[n++] = c;     // S1 GEMM -> LayerNorm partials | normalised rows written
    c += KS; s.pos[n++] = c;                     // S2: q2 in H
    s.pos[n++] = c;                              // S3: cross-attention output in A
    c += KS; s.pos[n++] = c; s.pos[n++] = c;     // S4 + LayerNorm
    for (int ch = 0; ch < 4; ++ch) {
        c += 2 * KS; s.pos[n++] = c;             // GEGLU chunk in H
        c += KS;
        if (NH == 1) s.pos[n++] = c;             // single H buffer: readers done
    }
    s.pos[n++] = c;                              // t3 in A
    c += KS; s.pos[n++] = c; s.pos[n++] = c;     // S6 GEMM (x_in in X) | output rows in H
    s.n = n;
    return s;
}

// L2 warmer wave (one extra wave per block of the fused-chain kernels).  The weights are cold (1.7 GB of other weights stream
// through the caches between two uses) and every block reads the SAME stream in lockstep, so without help each piece is a
// first touch for its XCD's L2 and the ring runs at the HBM round trip (PF pieces per ~2 us per wave: the tail took 61 us per
// launch against 43 us with L2-resident weights).  The warmer waves of an XCD's blocks (block b runs on XCD b % 8: an
// observation used for speed only) touch the stream between them -- one dword per 64 bytes, 4 KiB per instruction -- `lead`
// pieces ahead of the compute waves.  It walks the SAME barrier sequence as the compute waves (a wave that skips s_barrier
// holds the whole block at its first barrier until it exits -- measured, +19 us), which is also what paces it; nothing waits
// for a touch until the very end (a wave must not exit with loads in flight: its registers are handed to the next wave while
// the data is still coming -- measured, a memory fault).
template <int NW, unsigned PIECES>
__device__ __forceinline__ void warmer_wave(const char* wstream, const BarSched& S, int nshare, unsigned lead, int lane) {
    constexpr unsigned CHUNKS = PIECES / 4;                 // 4 KiB chunks per wave stream
    unsigned ch = (unsigned)(((int)blockIdx.x >> 3) % nshare);
    unsigned tmp = 0;
    for (int bi = 0; bi < S.n; ++bi) {
        const unsigned limit = (unsigned)S.pos[bi] + lead;
        while (ch < CHUNKS * NW) {
            const unsigned pos = ch / NW, w = ch - pos * NW;    // consumption order: position-major across the waves' streams
            if (pos * 4u >= limit) break;
            const char* ptr = wstream + ((size_t)(w * PIECES + pos * 4u) * 1024u + (size_t)lane * 64u);
            asm volatile("global_load_dword %0, %1, off" : "+v"(tmp) : "v"(ptr) : "memory");
            ch += (unsigned)nshare;
        }
        asm volatile("s_barrier" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" ::"v"(tmp));     // the destination register stays live until here
}

template <int C, int TM, int D>
__global__ __launch_bounds__((C / 32 + 1) * 64) void st_tail_kernel(const StTailParams p) {
    mdx_kernarg_touch<sizeof(StTailParams)>();
    constexpr int NW = C / 32;
    constexpr int NT = NW * 64;
    constexpr int BM = 32 * TM;
    constexpr int LDW = C + 8;           // halves per LDS row
    constexpr int LDB = LDW * 2;
    constexpr int BUF = BM * LDB;
    constexpr int CPR = C / 8;           // 16-byte chunks per row
    // the FFN's hidden chunks alternate between two buffers where LDS has room for both (32-row blocks): ONE barrier per chunk
    constexpr int NH = (TM == 1) ? 2 : 1;
    using V = VecOff<C>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Xb = smem;
    char* Ab = smem + BUF;
    char* Hb = smem + 2 * BUF;
    float* vec = reinterpret_cast<float*>(smem + (2 + NH) * BUF);
    float2* part = reinterpret_cast<float2*>(vec + V::total);   // [NW][BM]

    int nbar = 0;       // block barriers executed by this (compute) wave: LDS_BARRIER
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.x * BM;
    const int n0 = wave * 32;

    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.wstream, p.wstream_bytes);
    constexpr unsigned PIECES = 3 * (C / 16) + 4 * 3 * (C / 16) + (C / 16);   // per wave
    if (wave == NW) {      // L2 warmer wave: same barrier sequence as the compute waves (warmer_wave above)
        if (p.nshare <= 0) return;
        constexpr BarSched S = make_sched<C / 16, NH>();
        warmer_wave<NW, PIECES>(p.wstream, S, p.nshare, (unsigned)p.lead, lane);
        return;
    }
    unsigned piece = (unsigned)wave * PIECES;
    const unsigned voff = (unsigned)lane * 16u;
    u32x4 ring[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) ring[j] = wload(rs_w, voff, piece + j);

    // ---- S0: rows of attn_o -> A, tok -> X, vectors -> LDS
    {
        const f16* ga = p.attn_o + (size_t)m0 * C;
        const f16* gx = p.tok + (size_t)m0 * C;
        for (int idx = tid; idx < BM * CPR; idx += NT) {
            const int row = idx / CPR, ch = idx - row * CPR;
            const f16x8 va = *reinterpret_cast<const f16x8*>(ga + (size_t)row * C + ch * 8);
            const f16x8 vx = *reinterpret_cast<const f16x8*>(gx + (size_t)row * C + ch * 8);
            *reinterpret_cast<f16x8*>(Ab + row * LDB + ch * 16) = va;
            *reinterpret_cast<f16x8*>(Xb + row * LDB + ch * 16) = vx;
        }
        for (int idx = tid; idx < V::total / 4; idx += NT)
            reinterpret_cast<float4*>(vec)[idx] = reinterpret_cast<const float4*>(p.vec)[idx];
    }
    LDS_BARRIER();

    const int lane_row_off = l31 * LDB + hi * 16;           // A-fragment base of this lane inside a row buffer
    const int epi_off = l31 * LDB + (n0 + 4 * hi) * 2;      // this lane's first 4-column group of row l31 (tile 0, g = 0)

    auto dump = [&](const char* buf) {    // debug: LDS row buffer -> dbg rows (block-uniform call)
        LDS_BARRIER();
        for (int idx = tid; idx < BM * CPR; idx += NT) {
            const int row = idx / CPR, ch = idx - row * CPR;
            *reinterpret_cast<f16x8*>(p.dbg + (size_t)(m0 + row) * C + ch * 8) =
                *reinterpret_cast<const f16x8*>(buf + row * LDB + ch * 16);
        }
    };

    // t = acc + bias + X, rounded to fp16; LayerNorm(t) -> A, t -> X  (shared by S1 and S4)
    auto residual_ln = [&](f32x16 (&acc)[1][TM], const float* bias, const float* gamma, const float* beta) {
        f16x4 th[TM][4];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float su = 0.f, sq = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bb = *reinterpret_cast<const float4*>(bias + n0 + 8 * g + 4 * hi);
                const f16x4 xr = *reinterpret_cast<const f16x4*>(Xb + epi_off + i * 32 * LDB + g * 16);
                const float b4[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f16 h = (f16)(acc[0][i][4 * g + e] + b4[e] + (float)xr[e]);
                    th[i][g][e] = h;
                    const float f = (float)h;     // statistics of the fp16 values the unfused chain stores
                    su += f;
                    sq += f * f;
                }
            }
            su += __shfl_xor(su, 32, 64);
            sq += __shfl_xor(sq, 32, 64);
            if (hi == 0) part[wave * BM + i * 32 + l31] = make_float2(su, sq);
        }
        LDS_BARRIER();      // partials visible; every wave is done reading A
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float su = 0.f, sq = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const float2 v = part[w * BM + i * 32 + l31];
                su += v.x;
                sq += v.y;
            }
            const float mean = su * (1.0f / C);
            float var = sq * (1.0f / C) - mean * mean;
            var = var < 0.f ? 0.f : var;
            const float rstd = rsqrtf(var + p.eps);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 ga = *reinterpret_cast<const float4*>(gamma + n0 + 8 * g + 4 * hi);
                const float4 be = *reinterpret_cast<const float4*>(beta + n0 + 8 * g + 4 * hi);
                const float g4[4] = {ga.x, ga.y, ga.z, ga.w}, b4[4] = {be.x, be.y, be.z, be.w};
                f16x4 xn;
#pragma unroll
                for (int e = 0; e < 4; ++e) xn[e] = (f16)(((float)th[i][g][e] - mean) * rstd * g4[e] + b4[e]);
                *reinterpret_cast<f16x4*>(Ab + epi_off + i * 32 * LDB + g * 16) = xn;
                *reinterpret_cast<f16x4*>(Xb + epi_off + i * 32 * LDB + g * 16) = th[i][g];
            }
        }
        LDS_BARRIER();
    };

    f32x16 acc1[1][TM];

    // ---- S1: t1 = attn_o Wo1^T + bo1 + tok ; X <- t1 ; A <- LN2(t1)
    zero_acc(acc1);
    unit<C, TM, 1>(acc1, Ab + lane_row_off, ring, rs_w, voff, piece);
    residual_ln(acc1, vec + V::bo1, vec + V::g2, vec + V::be2);
    if (p.stop_after == 1) { dump(Xb); return; }
    if (p.stop_after == 2) { dump(Ab); return; }

    // ---- S2: q2 = A Wq2^T -> H
    zero_acc(acc1);
    unit<C, TM, 1>(acc1, Ab + lane_row_off, ring, rs_w, voff, piece);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f16x4*>(Hb + epi_off + i * 32 * LDB + g * 16) =
                f16x4{(f16)acc1[0][i][4 * g], (f16)acc1[0][i][4 * g + 1], (f16)acc1[0][i][4 * g + 2], (f16)acc1[0][i][4 * g + 3]};
    LDS_BARRIER();
    if (p.stop_after == 3) { dump(Hb); return; }

    // ---- S3: cross-attention over the cached context keys: A <- softmax(q2 K^T scale) V, per (head, 32-row tile)
    {
        constexpr int KS = (D + 15) / 16, DT = (D + 31) / 32, KT = 3;
        const int xb = m0 / p.HW;
        const __amdgpu_buffer_rsrc_t rs_k = make_rsrc(p.kc + (size_t)xb * p.TC * C, p.kc_bytes);
        const __amdgpu_buffer_rsrc_t rs_v = make_rsrc(p.vtc + (size_t)xb * C * p.TC, p.vtc_bytes);
        const int krow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);   // pi(l31): bits 2 and 3 swapped (attention.hip)
        const u32x4 zero4 = {0u, 0u, 0u, 0u};
        for (int item = wave; item < p.heads * TM; item += NW) {
            const int h = item / TM, rt = item - h * TM;
            // register budget (168 with three waves per SIMD, 40 of them the weight ring): K fragments die into S^T before
            // the V^T fragments of d tiles >= 1 are fetched; d tile 0 is fetched under the S^T MFMAs and the softmax.
            // (Requesting the K fragments before the q2 GEMM was tried: its 12 strided loads sit in front of that GEMM's ring
            // refills -- loads return in order -- and S2..S4 got 5 us slower for the ~1 us round trip it hid.)
            u32x4 kf[KT][KS], vf[DT][2 * KT];
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const bool ok = s * 16 + hi * 8 < D;
                    const unsigned off = (unsigned)(((t * 32 + krow) * C + h * D + s * 16 + hi * 8) * 2);
                    kf[t][s] = ok ? __builtin_amdgcn_raw_buffer_load_b128(rs_k, off, 0, 0) : zero4;
                }
            auto load_v = [&](int j) {
#pragma unroll
                for (int c = 0; c < 2 * KT; ++c) {
                    const unsigned off = (unsigned)(((h * D + j * 32 + l31) * p.TC + c * 16 + hi * 8) * 2);
                    vf[j][c] = __builtin_amdgcn_raw_buffer_load_b128(rs_v, off, 0, 0);
                }
            };
            load_v(0);
            f16x8 qf[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s * 16 + hi * 8 < D)
                    qf[s] = *reinterpret_cast<const f16x8*>(Hb + (rt * 32 + l31) * LDB + (h * D + s * 16 + hi * 8) * 2);
                else
                    qf[s] = __builtin_bit_cast(f16x8, zero4);
            }
            f32x16 acc_s[KT];
#pragma unroll
            for (int t = 0; t < KT; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_s[t][r] = 0.f;
#pragma unroll
                for (int s = 0; s < KS; ++s)
                    acc_s[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, kf[t][s]), qf[s], acc_s[t], 0, 0, 0);
            }
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * 32 + (r & 3) + 4 * ((r >> 2) & 1) + 8 * hi + 16 * (r >> 3);
                    if (key >= p.ctx_len) acc_s[t][r] = -INFINITY;
                    mx = fmaxf(mx, acc_s[t][r]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mb = mx * p.scale_log2;
            float psum = 0.f;
            f16x8 pf[2 * KT];
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(acc_s[t][r] * p.scale_log2 - mb);
                    psum += pv;
                    pf[t * 2 + (r >> 3)][r & 7] = (f16)pv;
                }
            psum += __shfl_xor(psum, 32, 64);
            const float inv = 1.0f / psum;
#pragma unroll
            for (int j = 1; j < DT; ++j) load_v(j);
            f32x16 acc_o[DT];
#pragma unroll
            for (int j = 0; j < DT; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_o[j][r] = 0.f;
#pragma unroll
                for (int c = 0; c < 2 * KT; ++c)
                    acc_o[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, vf[j][c]), pf[c], acc_o[j], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < DT; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int dd = j * 32 + 8 * g + 4 * hi;
                    if (dd < D)
                        *reinterpret_cast<f16x4*>(Ab + (rt * 32 + l31) * LDB + (h * D + dd) * 2) =
                            f16x4{(f16)(acc_o[j][4 * g] * inv), (f16)(acc_o[j][4 * g + 1] * inv), (f16)(acc_o[j][4 * g + 2] * inv),
                                  (f16)(acc_o[j][4 * g + 3] * inv)};
                }
        }
    }
    LDS_BARRIER();
    if (p.stop_after == 4) { dump(Ab); return; }

    // ---- S4: t2 = o2 Wo2^T + bo2 + t1 ; X <- t2 ; A <- LN3(t2)
    zero_acc(acc1);
    unit<C, TM, 1>(acc1, Ab + lane_row_off, ring, rs_w, voff, piece);
    residual_ln(acc1, vec + V::bo2, vec + V::g3, vec + V::be3);
    if (p.stop_after == 5) { dump(Xb); return; }
    if (p.stop_after == 6) { dump(Ab); return; }

    // ---- S5: FFN, hidden dimension streamed in 4 chunks of C columns
    f32x16 acc2[1][TM];
    zero_acc(acc2);
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
        f32x16 accg[2][TM];
        zero_acc(accg);
        unit<C, TM, 2>(accg, Ab + lane_row_off, ring, rs_w, voff, piece);
        const float* ba = vec + V::b1 + c * C + n0 + 4 * hi;
        const float* bg = ba + 4 * C;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 a4 = *reinterpret_cast<const float4*>(ba + 8 * g);
                const float4 g4 = *reinterpret_cast<const float4*>(bg + 8 * g);
                const float aa[4] = {a4.x, a4.y, a4.z, a4.w}, gg[4] = {g4.x, g4.y, g4.z, g4.w};
                f16x4 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) hv[e] = (f16)((accg[0][i][4 * g + e] + aa[e]) * gelu_tanh_f(accg[1][i][4 * g + e] + gg[e]));
                *reinterpret_cast<f16x4*>(Hb + (c % NH) * BUF + epi_off + i * 32 * LDB + g * 16) = hv;
            }
        LDS_BARRIER();
        unit<C, TM, 1>(acc2, Hb + (c % NH) * BUF + lane_row_off, ring, rs_w, voff, piece);
        // one buffer: every wave must be done reading this chunk before the next one is written.  Two buffers: chunk c + 1 goes
        // to the other one, whose last readers (ff2 of chunk c - 1) all passed the barrier above before any wave got here
        if (NH == 1) LDS_BARRIER();
    }
    // t3 = acc2 + b2 + t2 -> A ; X <- x_in rows (residual of proj_out)
    {
        f16x8 xin[(BM * CPR + NT - 1) / NT];
        const f16* gx = p.x_in + (size_t)m0 * C;
#pragma unroll
        for (int k = 0; k < (BM * CPR + NT - 1) / NT; ++k) {
            const int idx = tid + k * NT;
            if (idx < BM * CPR) {
                const int row = idx / CPR, ch = idx - row * CPR;
                xin[k] = *reinterpret_cast<const f16x8*>(gx + (size_t)row * C + ch * 8);
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bb = *reinterpret_cast<const float4*>(vec + V::b2 + n0 + 8 * g + 4 * hi);
                const f16x4 xr = *reinterpret_cast<const f16x4*>(Xb + epi_off + i * 32 * LDB + g * 16);
                const float b4[4] = {bb.x, bb.y, bb.z, bb.w};
                f16x4 tv;
#pragma unroll
                for (int e = 0; e < 4; ++e) tv[e] = (f16)(acc2[0][i][4 * g + e] + b4[e] + (float)xr[e]);
                *reinterpret_cast<f16x4*>(Ab + epi_off + i * 32 * LDB + g * 16) = tv;
            }
        LDS_BARRIER();      // t3 complete in A; every lane has read its t2 elements of X
#pragma unroll
        for (int k = 0; k < (BM * CPR + NT - 1) / NT; ++k) {
            const int idx = tid + k * NT;
            if (idx < BM * CPR) {
                const int row = idx / CPR, ch = idx - row * CPR;
                *reinterpret_cast<f16x8*>(Xb + row * LDB + ch * 16) = xin[k];
            }
        }
    }
    if (p.stop_after == 7) { dump(Ab); return; }

    // ---- S6: out = t3 Wpo^T + bpo + x_in -> H -> global (+ per-column {sum, sumsq} of this row block for the next GroupNorm)
    zero_acc(acc1);
    unit<C, TM, 1>(acc1, Ab + lane_row_off, ring, rs_w, voff, piece);
    LDS_BARRIER();          // X (x_in rows) written by every thread
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bb = *reinterpret_cast<const float4*>(vec + V::bpo + n0 + 8 * g + 4 * hi);
            const f16x4 xr = *reinterpret_cast<const f16x4*>(Xb + epi_off + i * 32 * LDB + g * 16);
            const float b4[4] = {bb.x, bb.y, bb.z, bb.w};
            f16x4 ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[e] = (f16)(acc1[0][i][4 * g + e] + b4[e] + (float)xr[e]);
            *reinterpret_cast<f16x4*>(Hb + epi_off + i * 32 * LDB + g * 16) = ov;
        }
    LDS_BARRIER();
    for (int idx = tid; idx < BM * CPR; idx += NT) {
        const int row = idx / CPR, ch = idx - row * CPR;
        *reinterpret_cast<f16x8*>(p.out + (size_t)(m0 + row) * C + ch * 8) = *reinterpret_cast<const f16x8*>(Hb + row * LDB + ch * 16);
    }
    if (p.colstats_out) {
        for (int i = tid; i < 2 * C; i += NT) {
            const int c = i >> 1;
            float a = 0.f;
            if (i & 1) {
#pragma unroll 8
                for (int r = 0; r < BM; ++r) {
                    const float v = (float)*reinterpret_cast<const f16*>(Hb + r * LDB + c * 2);
                    a += v * v;
                }
            } else {
#pragma unroll 8
                for (int r = 0; r < BM; ++r) a += (float)*reinterpret_cast<const f16*>(Hb + r * LDB + c * 2);
            }
            p.colstats_out[((size_t)blockIdx.x * C + c) * 2 + (i & 1)] = a;
        }
    }
    if (p.stop_after == ST_COUNT_BARRIERS && blockIdx.x == 0 && tid == 0) *reinterpret_cast<int*>(p.dbg) = nbar;
}

template <int C, int TM, int D>
void launch_tail(const StTailParams& p, hipStream_t st) {
    constexpr int NW = C / 32, BM = 32 * TM;
    constexpr size_t lds = (size_t)(TM == 1 ? 4 : 3) * BM * (C + 8) * 2 + (size_t)VecOff<C>::total * 4 + (size_t)NW * BM * 8;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static MdxPerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&st_tail_kernel<C, TM, D>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL((st_tail_kernel<C, TM, D>), dim3(p.M / BM), dim3((NW + 1) * 64), lds, st, p);
}


// ---------------------------------------------------------------------------------------------------------------
// Row-local fused HEAD of a SpatialTransformer block: SpatialTransformer.norm (GroupNorm, statistics folded from the
// producer's per-row-block column partials) -> proj_in -> BasicTransformerBlock.norm1 -> attn1.to_q | to_k | to_v, one
// launch (attention.py:83-84, 212, 241-247, 176, 108-112).  Writes the token stream `tok` (residual of attn1), q | k
// row-major [M][2C] and V^T [B][C][tokens] -- what mdx_attention_f16 reads.
struct StHeadParams {
    const f16* x;
    const float* cs;         // [B * nrb][C][2] column partials of x from its producer
    const char* wstream;
    const float* vec;        // [gn gamma | gn beta | b_proj_in | ln1 gamma | ln1 beta]
    f16* tok;
    f16* qk;
    f16* vt;
    f16* dbg;
    int M, HW, nrb, vt_ld, nshare, lead, stop_after;
    float gn_eps, ln_eps;
};

template <int KS>
constexpr BarSched make_head_sched() {
    BarSched s{};
    int n = 0, c = 0;
    s.pos[n++] = c; s.pos[n++] = c; s.pos[n++] = c; s.pos[n++] = c;   // channel sums | read | scale / shift | GroupNorm rows in A
    c += KS; s.pos[n++] = c; s.pos[n++] = c;             // proj_in GEMM -> LayerNorm partials | LN rows in A, tok in X
    c += KS; s.pos[n++] = c;                             // q tile staged
    c += KS; s.pos[n++] = c;                             // k tile staged
    c += KS; s.pos[n++] = c;                             // v tile staged
    s.n = n;
    return s;
}

template <int C, int TM>
__global__ __launch_bounds__((C / 32 + 1) * 64) void st_head_kernel(const StHeadParams p) {
    mdx_kernarg_touch<sizeof(StHeadParams)>();
    constexpr int NW = C / 32;
    constexpr int NT = NW * 64;
    constexpr int BM = 32 * TM;
    constexpr int LDW = C + 8;
    constexpr int LDB = LDW * 2;
    constexpr int BUF = BM * LDB;
    constexpr int CPR = C / 8;
    constexpr int CPG = C / 32;          // channels per GroupNorm group (32 groups)
    constexpr unsigned PIECES = 4 * (C / 16);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Xb = smem;
    char* Ab = smem + BUF;
    char* Hb = smem + 2 * BUF;
    float* vec = reinterpret_cast<float*>(smem + 3 * BUF);      // [5 C]
    float2* csum = reinterpret_cast<float2*>(vec + 5 * C);      // [C] channel {sum, sumsq}, then {scale, shift}
    float2* part = csum + C;                                    // [NW][BM]

    int nbar = 0;       // block barriers executed by this (compute) wave: LDS_BARRIER
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.x * BM;
    const int n0 = wave * 32;
    if (wave == NW) {
        if (p.nshare <= 0) return;
        constexpr BarSched S = make_head_sched<C / 16>();
        warmer_wave<NW, PIECES>(p.wstream, S, p.nshare, (unsigned)p.lead, lane);
        return;
    }
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.wstream, (unsigned)(NW * PIECES * 1024u));
    unsigned piece = (unsigned)wave * PIECES;
    const unsigned voff = (unsigned)lane * 16u;
    u32x4 ring[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) ring[j] = wload(rs_w, voff, piece + j);

    // ---- rows of x (raw) into registers, vectors into LDS, GroupNorm statistics of this sample from the column partials
    constexpr int XL = (BM * CPR + NT - 1) / NT;
    f16x8 xr[XL];
    {
        const f16* gx = p.x + (size_t)m0 * C;
#pragma unroll
        for (int k = 0; k < XL; ++k) {
            const int idx = tid + k * NT;
            if (idx < BM * CPR) xr[k] = *reinterpret_cast<const f16x8*>(gx + (size_t)(idx / CPR) * C + (idx % CPR) * 8);
        }
        for (int idx = tid; idx < 5 * C / 4; idx += NT)
            reinterpret_cast<float4*>(vec)[idx] = reinterpret_cast<const float4*>(p.vec)[idx];
        const int b = m0 / p.HW;
        const int c = tid >> 1, half = tid & 1;       // NT == 2 C: two threads per channel, half of the row blocks each
        const float2* src = reinterpret_cast<const float2*>(p.cs) + (size_t)b * p.nrb * C + c;
        const int k0 = half * ((p.nrb + 1) >> 1), k1 = half ? p.nrb : ((p.nrb + 1) >> 1);
        float su = 0.f, sq = 0.f;
        int k = k0;
        for (; k + 8 <= k1; k += 8) {       // eight loads in flight per thread (a one-at-a-time loop is 16 L2 round trips)
            float2 v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = src[(size_t)(k + e) * C];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                su += v[e].x;
                sq += v[e].y;
            }
        }
        for (; k < k1; ++k) {
            const float2 v = src[(size_t)k * C];
            su += v.x;
            sq += v.y;
        }
        su += __shfl_xor(su, 1, 64);
        sq += __shfl_xor(sq, 1, 64);
        if (half == 0) csum[c] = make_float2(su, sq);
    }
    LDS_BARRIER();
    {
        const int c = tid >> 1;
        const int g = c / CPG;
        float su = 0.f, sq = 0.f;
#pragma unroll
        for (int e = 0; e < CPG; ++e) {
            const float2 v = csum[g * CPG + e];
            su += v.x;
            sq += v.y;
        }
        const float inv = 1.0f / ((float)CPG * (float)p.HW);
        const float mean = su * inv;
        float var = sq * inv - mean * mean;
        var = var < 0.f ? 0.f : var;
        const float a = vec[c] * rsqrtf(var + p.gn_eps);
        const float sh = vec[C + c] - mean * a;
        LDS_BARRIER();          // every thread has read the channel sums
        if ((tid & 1) == 0) csum[c] = make_float2(a, sh);
    }
    LDS_BARRIER();
#pragma unroll
    for (int k = 0; k < XL; ++k) {
        const int idx = tid + k * NT;
        if (idx < BM * CPR) {
            const int row = idx / CPR, ch = idx - row * CPR;
            f16x8 y;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float2 ss = csum[ch * 8 + e];
                y[e] = (f16)((float)xr[k][e] * ss.x + ss.y);
            }
            *reinterpret_cast<f16x8*>(Ab + row * LDB + ch * 16) = y;
        }
    }
    LDS_BARRIER();

    const int lane_row_off = l31 * LDB + hi * 16;
    const int epi_off = l31 * LDB + (n0 + 4 * hi) * 2;
    auto dump = [&](const char* buf) {
        LDS_BARRIER();
        for (int idx = tid; idx < BM * CPR; idx += NT) {
            const int row = idx / CPR, ch = idx - row * CPR;
            *reinterpret_cast<f16x8*>(p.dbg + (size_t)(m0 + row) * C + ch * 8) =
                *reinterpret_cast<const f16x8*>(buf + row * LDB + ch * 16);
        }
    };
    if (p.stop_after == 1) { dump(Ab); return; }

    // ---- proj_in: tok = GN(x) Wpi^T + b ; X <- tok ; A <- LN1(tok)
    f32x16 acc[1][TM];
    zero_acc(acc);
    unit<C, TM, 1>(acc, Ab + lane_row_off, ring, rs_w, voff, piece);
    {
        const float* bias = vec + 2 * C;
        const float* gamma = vec + 3 * C;
        const float* beta = vec + 4 * C;
        f16x4 th[TM][4];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float su = 0.f, sq = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bb = *reinterpret_cast<const float4*>(bias + n0 + 8 * g + 4 * hi);
                const float b4[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f16 h = (f16)(acc[0][i][4 * g + e] + b4[e]);
                    th[i][g][e] = h;
                    const float f = (float)h;
                    su += f;
                    sq += f * f;
                }
            }
            su += __shfl_xor(su, 32, 64);
            sq += __shfl_xor(sq, 32, 64);
            if (hi == 0) part[wave * BM + i * 32 + l31] = make_float2(su, sq);
        }
        LDS_BARRIER();
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float su = 0.f, sq = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const float2 v = part[w * BM + i * 32 + l31];
                su += v.x;
                sq += v.y;
            }
            const float mean = su * (1.0f / C);
            float var = sq * (1.0f / C) - mean * mean;
            var = var < 0.f ? 0.f : var;
            const float rstd = rsqrtf(var + p.ln_eps);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 ga = *reinterpret_cast<const float4*>(gamma + n0 + 8 * g + 4 * hi);
                const float4 be = *reinterpret_cast<const float4*>(beta + n0 + 8 * g + 4 * hi);
                const float g4[4] = {ga.x, ga.y, ga.z, ga.w}, b4[4] = {be.x, be.y, be.z, be.w};
                f16x4 xn;
#pragma unroll
                for (int e = 0; e < 4; ++e) xn[e] = (f16)(((float)th[i][g][e] - mean) * rstd * g4[e] + b4[e]);
                *reinterpret_cast<f16x4*>(Ab + epi_off + i * 32 * LDB + g * 16) = xn;
                *reinterpret_cast<f16x4*>(Xb + epi_off + i * 32 * LDB + g * 16) = th[i][g];
            }
        }
        LDS_BARRIER();
    }
    if (p.stop_after == 2) { dump(Xb); return; }
    if (p.stop_after == 3) { dump(Ab); return; }
    // token rows -> global (the residual stream the tail reads)
    for (int idx = tid; idx < BM * CPR; idx += NT) {
        const int row = idx / CPR, ch = idx - row * CPR;
        *reinterpret_cast<f16x8*>(p.tok + (size_t)(m0 + row) * C + ch * 8) = *reinterpret_cast<const f16x8*>(Xb + row * LDB + ch * 16);
    }

    // ---- q | k | v tiles: GEMM -> staging buffer (H, X, H) -> coalesced stores; V goes out transposed
    auto stage_tile = [&](char* buf) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f16x4*>(buf + epi_off + i * 32 * LDB + g * 16) =
                    f16x4{(f16)acc[0][i][4 * g], (f16)acc[0][i][4 * g + 1], (f16)acc[0][i][4 * g + 2], (f16)acc[0][i][4 * g + 3]};
        LDS_BARRIER();
    };
    auto store_rows = [&](const char* buf, int col0) {
        for (int idx = tid; idx < BM * CPR; idx += NT) {
            const int row = idx / CPR, ch = idx - row * CPR;
            *reinterpret_cast<f16x8*>(p.qk + (size_t)(m0 + row) * (2 * C) + col0 + ch * 8) =
                *reinterpret_cast<const f16x8*>(buf + row * LDB + ch * 16);
        }
    };
    zero_acc(acc);
    unit<C, TM, 1>(acc, Ab + lane_row_off, ring, rs_w, voff, piece);
    stage_tile(Hb);
    store_rows(Hb, 0);
    zero_acc(acc);
    unit<C, TM, 1>(acc, Ab + lane_row_off, ring, rs_w, voff, piece);
    stage_tile(Xb);          // (every thread finished copying tok out of X before the barrier inside the q stage)
    store_rows(Xb, C);
    zero_acc(acc);
    unit<C, TM, 1>(acc, Ab + lane_row_off, ring, rs_w, voff, piece);
    stage_tile(Hb);          // (the q rows were copied out of H before the barrier inside the k stage)
    {
        const int b = m0 / p.HW, tok0 = m0 - b * p.HW;
        constexpr int JC = BM / 8;       // 8-token chunks per channel
        for (int idx = tid; idx < C * JC; idx += NT) {
            const int c = idx / JC, j = idx - c * JC;
            f16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = *reinterpret_cast<const f16*>(Hb + (j * 8 + e) * LDB + c * 2);
            *reinterpret_cast<f16x8*>(p.vt + ((size_t)b * C + c) * p.vt_ld + tok0 + j * 8) = v;
        }
    }
    if (p.stop_after == ST_COUNT_BARRIERS && blockIdx.x == 0 && tid == 0) *reinterpret_cast<int*>(p.dbg) = nbar;
}

template <int C, int TM>
void launch_head(const StHeadParams& p, hipStream_t st) {
    constexpr int NW = C / 32, BM = 32 * TM;
    constexpr size_t lds = (size_t)3 * BM * (C + 8) * 2 + (size_t)5 * C * 4 + (size_t)C * 8 + (size_t)NW * BM * 8;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static MdxPerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&st_head_kernel<C, TM>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL((st_head_kernel<C, TM>), dim3(p.M / BM), dim3((NW + 1) * 64), lds, st, p);
}
}  // namespace

extern "C" size_t mdx_st_tail_stream_bytes(int C) { return (size_t)(C / 32) * (16 * (C / 16)) * 1024; }

extern "C" int mdx_st_tail_supported(int C, int heads, int dim_head, int tokens_per_sample, int tile_rows) {
    return C == 320 && heads * dim_head == C && (dim_head == 64 || dim_head == 40) && (tile_rows == 32 || tile_rows == 64) &&
           tokens_per_sample > 0 && tokens_per_sample % tile_rows == 0;
}

extern "C" int mdx_st_tail_f16(const mdx_st_tail_desc* d, mdx_stream_t s) {
    MDX_REQUIRE(d && d->attn_out && d->tok && d->x_in && d->out && d->ctx_k && d->ctx_vt && d->wstream && d->vec,
                "mdx_st_tail_f16: null pointer");
    MDX_REQUIRE(mdx_st_tail_supported(d->C, d->heads, d->dim_head, d->tokens, d->tile_rows),
                "mdx_st_tail_f16: unsupported shape C=%d heads=%d d=%d tokens=%d tile_rows=%d", d->C, d->heads, d->dim_head,
                d->tokens, d->tile_rows);
    MDX_REQUIRE(d->B > 0 && d->ctx_len > 0 && d->ctx_len <= d->ctx_cap && d->ctx_cap % 8 == 0 && d->ctx_cap <= 96,
                "mdx_st_tail_f16: context length %d / capacity %d (capacity: multiple of 8, <= 96)", d->ctx_len, d->ctx_cap);
    MDX_REQUIRE(d->out != d->attn_out && d->out != d->tok && d->out != d->x_in, "mdx_st_tail_f16: out must not alias an input");
    StTailParams p{};
    p.attn_o = (const f16*)d->attn_out;
    p.tok = (const f16*)d->tok;
    p.x_in = (const f16*)d->x_in;
    p.out = (f16*)d->out;
    p.kc = (const f16*)d->ctx_k;
    p.vtc = (const f16*)d->ctx_vt;
    p.wstream = (const char*)d->wstream;
    p.vec = d->vec;
    p.colstats_out = d->colstats_out;
    p.dbg = (f16*)d->debug_out;
    p.M = d->B * d->tokens;
    p.HW = d->tokens;
    p.TC = d->ctx_cap;
    p.ctx_len = d->ctx_len;
    p.heads = d->heads;
    {
        // L2 warmer waves: the blocks of an XCD share the touching (<= 32 ways); off for debug taps and tiny launches
        const int per_xcd = (p.M / d->tile_rows) / 8;
        p.nshare = (d->debug_out || d->warm == 0 || per_xcd < 4) ? 0 : (per_xcd > 32 ? 32 : per_xcd);
        p.lead = 80;
        if (d->warm > 1) {      // tuning: warm = lead * 100 + share ways
            p.lead = d->warm / 100;
            if (p.nshare > 0 && d->warm % 100 > 0 && d->warm % 100 < p.nshare) p.nshare = d->warm % 100;
        }
    }
    p.scale_log2 = d->scale * 1.4426950408889634f;
    p.eps = d->ln_eps;
    p.stop_after = d->debug_out ? d->debug_stage : 0;
    p.wstream_bytes = (unsigned)mdx_st_tail_stream_bytes(d->C);
    p.kc_bytes = (unsigned)((size_t)d->ctx_cap * d->C * 2);
    p.vtc_bytes = (unsigned)((size_t)d->C * d->ctx_cap * 2);
    hipStream_t st = (hipStream_t)s;
    if (d->dim_head == 64) {
        if (d->tile_rows == 64) launch_tail<320, 2, 64>(p, st); else launch_tail<320, 1, 64>(p, st);
    } else {
        if (d->tile_rows == 64) launch_tail<320, 2, 40>(p, st); else launch_tail<320, 1, 40>(p, st);
    }
    MDX_LAUNCH_CHECK("mdx_st_tail_f16");
    return MDX_OK;
}

// Barriers in the warmer wave's schedule for this launch form (tests compare it with what the compute waves execute).
extern "C" int mdx_st_tail_sched_barriers(int C, int tile_rows) {
    if (C != 320) return -1;
    return tile_rows == 64 ? make_sched<320 / 16, 1>().n : make_sched<320 / 16, 2>().n;
}
extern "C" int mdx_st_head_sched_barriers(int C, int tile_rows) {
    (void)tile_rows;
    if (C != 320) return -1;
    return make_head_sched<320 / 16>().n;
}

extern "C" size_t mdx_st_head_stream_bytes(int C) { return (size_t)(C / 32) * (4 * (C / 16)) * 1024; }

extern "C" int mdx_st_head_supported(int C, int tokens_per_sample, int tile_rows) {
    return C == 320 && (tile_rows == 32 || tile_rows == 64) && tokens_per_sample > 0 && tokens_per_sample % tile_rows == 0 &&
           tokens_per_sample % 8 == 0;
}

extern "C" int mdx_st_head_f16(const mdx_st_head_desc* d, mdx_stream_t s) {
    MDX_REQUIRE(d && d->x && d->colstats && d->wstream && d->vec && d->tok && d->qk && d->vt, "mdx_st_head_f16: null pointer");
    MDX_REQUIRE(mdx_st_head_supported(d->C, d->tokens, d->tile_rows), "mdx_st_head_f16: unsupported shape C=%d tokens=%d tile_rows=%d",
                d->C, d->tokens, d->tile_rows);
    MDX_REQUIRE(d->B > 0 && d->nrb > 0 && d->vt_ld >= d->tokens && d->vt_ld % 8 == 0, "mdx_st_head_f16: bad extents");
    StHeadParams p{};
    p.x = (const f16*)d->x;
    p.cs = d->colstats;
    p.wstream = (const char*)d->wstream;
    p.vec = d->vec;
    p.tok = (f16*)d->tok;
    p.qk = (f16*)d->qk;
    p.vt = (f16*)d->vt;
    p.dbg = (f16*)d->debug_out;
    p.M = d->B * d->tokens;
    p.HW = d->tokens;
    p.nrb = d->nrb;
    p.vt_ld = d->vt_ld;
    p.stop_after = d->debug_out ? d->debug_stage : 0;
    p.gn_eps = d->gn_eps;
    p.ln_eps = d->ln_eps;
    {
        const int per_xcd = (p.M / d->tile_rows) / 8;
        p.nshare = (d->debug_out || d->warm == 0 || per_xcd < 4) ? 0 : (per_xcd > 32 ? 32 : per_xcd);
        p.lead = 80;
    }
    hipStream_t st = (hipStream_t)s;
    if (d->tile_rows == 64) launch_head<320, 2>(p, st); else launch_head<320, 1>(p, st);
    MDX_LAUNCH_CHECK("mdx_st_head_f16");
    return MDX_OK;
}
