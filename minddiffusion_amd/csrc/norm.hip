// GroupNorm(32)(+SiLU) and LayerNorm on NHWC / token-major fp16 for gfx950.  HBM-bound kernels:
// 16-byte coalesced accesses, fp32 statistics, wavefront (64-lane) shuffle reductions.
//
// GroupNorm is two launches for tensors of >= 1024 pixels per sample:
//   gn_stats : every block reduces a slab of pixels to per-(sample, group) partial {sum, sumsq}
//              (deterministic: partials are written, not atomically added)
//   gn_apply : every block first folds the partials into per-channel scale/shift in LDS, then
//              streams y = silu(x * scale_c + shift_c)
// and ONE launch (gn_fused_kernel: a block owns a whole-group column block of a sample) for the small
// tensors of the deep UNet levels, where the second launch costs more than the work.
// The input may be the channel concat of two tensors (UNet skip connections): the concat is
// never materialised.
#include "mdx_common.h"

#include <stdlib.h>

#include <type_traits>

namespace {

constexpr int GN_MAX_C = 8192;
constexpr int GN_MAX_NBLK = 256;     // pixel slabs per sample (bounds the partial-sum fold in gn_apply; 512x512 VAE tensors need > 64)
constexpr int GN_TARGET_BLOCKS = 1024;

struct GnParams {
    const f16* x1;
    const f16* x2;
    const float* gamma;
    const float* beta;
    f16* y;
    float* ws;  // [B][nblk][groups][2] partial {sum, sumsq}
    int C1, C2, C, CC1, CC;  // CCx = 16-byte chunks (8 channels) per pixel
    int B, HW, groups, cpg;
    int nblk, pix;           // pixel slabs per sample, pixels per slab
    int cw, ncb;             // chunk columns per block (whole groups), column blocks
    float eps;
    int silu;
    const float* scale;   // optional FiLM modulation rows [B][mod_ld] (GLIDE ResBlock): y = GN(x)*(1+scale)+shift
    const float* shift;
    int mod_ld;
    // statistics from the producers' column partials (mdx_gemm_desc.colstats_out) instead of gn_stats: [B * nrb][Cx][2]
    const float* cs1;
    const float* cs2;
    int nrb1, nrb2;
    // latency diet (round 5; library option gn_prefetch): the affine parameters (gamma / beta / FiLM rows: cold in HBM, 1.7 GB of
    // weights stream through the caches between two uses) and the first batch of pixels are fetched at the TOP of the kernel,
    // so that their misses overlap the statistics' round trips instead of following them.  Same arithmetic, same bits.
    int pre;
};

__device__ __forceinline__ f16x8 gn_load(const GnParams& p, int b, int pix, int col) {
    if (col < p.CC1) return *reinterpret_cast<const f16x8*>(p.x1 + ((size_t)b * p.HW + pix) * p.C1 + col * 8);
    return *reinterpret_cast<const f16x8*>(p.x2 + ((size_t)b * p.HW + pix) * p.C2 + (col - p.CC1) * 8);
}

// grid (nblk, ncb, B); block 256 = tcols (= columns of this block, <= 64) x trows.
// Every thread owns ONE chunk column (8 channels) and strides over the slab's pixels, so the per-channel
// accumulators stay in registers; a fixed-order LDS fold then produces per-group partials
// (deterministic: no atomics anywhere).
__global__ __launch_bounds__(256) void gn_stats_kernel(const GnParams p) {
    mdx_kernarg_touch<sizeof(GnParams)>();
    extern __shared__ __attribute__((aligned(16))) float red[];  // [trows][cols*8][2]
    const int b = blockIdx.z, cb = blockIdx.y;
    const int p0 = blockIdx.x * p.pix;
    const int p1 = min(p.HW, p0 + p.pix);
    const int col0 = cb * p.cw;
    const int cols = min(p.cw, p.CC - col0);
    const int tid = threadIdx.x;
    const int trows = 256 / cols;
    const int tc = tid % cols, tr = tid / cols;
    const int chs = cols * 8;
    if (tr < trows) {
        float s[8], q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
        // four independent loads in flight per thread (the slab is short: latency, not bandwidth, is the cost)
        int pix = p0 + tr;
        for (; pix + 3 * trows < p1; pix += 4 * trows) {
            f16x8 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = gn_load(p, b, pix + u * trows, col0 + tc);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = (float)v[u][e];
                    s[e] += f;
                    q[e] += f * f;
                }
        }
        for (; pix < p1; pix += trows) {
            const f16x8 v = gn_load(p, b, pix, col0 + tc);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)v[e];
                s[e] += f;
                q[e] += f * f;
            }
        }
        float* dst = red + ((size_t)tr * chs + tc * 8) * 2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            dst[e * 2] = s[e];
            dst[e * 2 + 1] = q[e];
        }
    }
    __syncthreads();
    // fold the trows partial rows per channel (all threads), then each group's cpg channels with 8 lanes per group
    // and a fixed-order shuffle tree: deterministic, and no thread walks more than cpg/8 + trows values serially.
    float* csum = red + (size_t)256 * 8 * 2;   // [chs][2]
    for (int c = tid; c < chs; c += 256) {
        float s = 0.f, q = 0.f;
        for (int r = 0; r < trows; ++r) {
            s += red[((size_t)r * chs + c) * 2];
            q += red[((size_t)r * chs + c) * 2 + 1];
        }
        csum[c * 2] = s;
        csum[c * 2 + 1] = q;
    }
    __syncthreads();
    const int ng = chs / p.cpg;          // whole groups in this column block
    const int g0 = col0 * 8 / p.cpg;
    {
        const int g = tid >> 3, j = tid & 7;
        float s = 0.f, q = 0.f;
        if (g < ng) {
            for (int c = g * p.cpg + j; c < (g + 1) * p.cpg; c += 8) {
                s += csum[c * 2];
                q += csum[c * 2 + 1];
            }
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            s += __shfl_xor(s, o, 64);
            q += __shfl_xor(q, o, 64);
        }
        if (g < ng && j == 0) {
            float* o = p.ws + (((size_t)b * p.nblk + blockIdx.x) * p.groups + g0 + g) * 2;
            o[0] = s;
            o[1] = q;
        }
    }
}

// grid (nblk, ncb, B); block 256.  Prologue folds the <= 64 slab partials of this block's groups with
// 8 lanes per group (fixed shuffle order), builds per-channel scale/shift in LDS, then streams the slab.
__global__ __launch_bounds__(256) void gn_apply_kernel(const GnParams p) {
    extern __shared__ __attribute__((aligned(16))) float ss[];  // [cols*8][2] scale/shift, then [32][2] group stats
    mdx_kernarg_touch<sizeof(GnParams)>();
    const int b = blockIdx.z, cb = blockIdx.y;
    const int col0 = cb * p.cw;
    const int cols = min(p.cw, p.CC - col0);
    const int chs = cols * 8;
    float* gstat = ss + chs * 2;
    const int tid = threadIdx.x;
    const int ng = chs / p.cpg;
    const int g0 = col0 * 8 / p.cpg;
    const int p0 = blockIdx.x * p.pix;
    const int p1 = min(p.HW, p0 + p.pix);
    const int trows = 256 / cols;
    const int tc = tid % cols, tr = tid / cols;
    // (p.pre) this thread's channels' affine parameters (chs <= 512: two per thread) and its first four pixels, in flight
    // while the statistics are folded below
    float pg[2] = {0.f, 0.f}, pb[2] = {0.f, 0.f}, pm[2] = {0.f, 0.f}, pf[2] = {0.f, 0.f};
    f16x8 v0[4];
    if (p.pre) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + i * 256;
            if (c < chs) {
                pg[i] = p.gamma[col0 * 8 + c];
                pb[i] = p.beta[col0 * 8 + c];
                if (p.scale) {
                    pm[i] = p.scale[(size_t)b * p.mod_ld + col0 * 8 + c];
                    pf[i] = p.shift[(size_t)b * p.mod_ld + col0 * 8 + c];
                }
            }
        }
        if (tr < trows) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (p0 + tr + u * trows < p1) v0[u] = gn_load(p, b, p0 + tr + u * trows, col0 + tc);
        }
    }
    if (p.cs1) {
        // statistics = the producers' per-column {sum, sumsq} of every row block of this sample, folded per channel (fixed
        // order), then per group: no pass over the tensor.  The channel concat keeps its two sources' partial arrays apart.
        float* csum = gstat + 64;                    // [chs][2]
        for (int c = tid; c < chs; c += 256) {
            const int ch = col0 * 8 + c;
            const bool first = ch < p.C1;
            const float2* base = reinterpret_cast<const float2*>(first ? p.cs1 : p.cs2);
            const int cx = first ? p.C1 : p.C2, cc = first ? ch : ch - p.C1, nrb = first ? p.nrb1 : p.nrb2;
            const float2* src = base + (size_t)b * nrb * cx + cc;
            float su = 0.f, sq = 0.f;
            int k = 0;
            // Up to thirty-two partials in flight per thread: with minimal column blocks only chs (40 at C = 320) of the 256 threads fold,
            // and four loads at a time made this prologue a chain of nrb / 4 L2 round trips (8 at the 64 x 64 level) in front of
            // a kernel that streams its slab in a few microseconds.  Same order of additions as before: same bits.
            for (; k + 32 <= nrb; k += 32) {
                float2 v[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) v[u] = src[(size_t)(k + u) * cx];
#pragma unroll
                for (int u = 0; u < 32; ++u) {
                    su += v[u].x;
                    sq += v[u].y;
                }
            }
            for (; k + 16 <= nrb; k += 16) {
                float2 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = src[(size_t)(k + u) * cx];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    su += v[u].x;
                    sq += v[u].y;
                }
            }
            for (; k + 4 <= nrb; k += 4) {
                const float2 v0 = src[(size_t)k * cx], v1 = src[(size_t)(k + 1) * cx], v2 = src[(size_t)(k + 2) * cx],
                             v3 = src[(size_t)(k + 3) * cx];
                su += v0.x; sq += v0.y; su += v1.x; sq += v1.y; su += v2.x; sq += v2.y; su += v3.x; sq += v3.y;
            }
            for (; k < nrb; ++k) {
                const float2 v = src[(size_t)k * cx];
                su += v.x;
                sq += v.y;
            }
            csum[c * 2] = su;
            csum[c * 2 + 1] = sq;
        }
        __syncthreads();
    }
    {
        const int g = tid >> 3, j = tid & 7;
        float s = 0.f, q = 0.f;
        if (g < ng && p.cs1) {
            const float* csum = gstat + 64;
            for (int c = g * p.cpg + j; c < (g + 1) * p.cpg; c += 8) {
                s += csum[c * 2];
                q += csum[c * 2 + 1];
            }
        } else if (g < ng) {
            const float* w = p.ws + ((size_t)b * p.nblk * p.groups + g0 + g) * 2;
            for (int k = j; k < p.nblk; k += 8) {
                s += w[(size_t)k * p.groups * 2];
                q += w[(size_t)k * p.groups * 2 + 1];
            }
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            s += __shfl_xor(s, o, 64);
            q += __shfl_xor(q, o, 64);
        }
        if (g < ng && j == 0) {
            const float inv = 1.0f / ((float)p.cpg * (float)p.HW);
            const float mean = s * inv;
            float var = q * inv - mean * mean;
            var = var < 0.f ? 0.f : var;
            gstat[g * 2] = mean;
            gstat[g * 2 + 1] = rsqrtf(var + p.eps);
        }
    }
    __syncthreads();
    if (p.pre) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + i * 256;
            if (c < chs) {
                const int g = c / p.cpg;
                float a = pg[i] * gstat[g * 2 + 1];
                float sh = pb[i] - gstat[g * 2] * a;
                if (p.scale) {   // (x_hat*gamma + beta) * (1 + scale) + shift
                    const float m1 = 1.0f + pm[i];
                    a *= m1;
                    sh = sh * m1 + pf[i];
                }
                ss[c * 2] = a;
                ss[c * 2 + 1] = sh;
            }
        }
    } else {
        for (int c = tid; c < chs; c += 256) {
            const int g = c / p.cpg;
            float a = p.gamma[col0 * 8 + c] * gstat[g * 2 + 1];
            float sh = p.beta[col0 * 8 + c] - gstat[g * 2] * a;
            if (p.scale) {   // (x_hat*gamma + beta) * (1 + scale) + shift
                const float m1 = 1.0f + p.scale[(size_t)b * p.mod_ld + col0 * 8 + c];
                a *= m1;
                sh = sh * m1 + p.shift[(size_t)b * p.mod_ld + col0 * 8 + c];
            }
            ss[c * 2] = a;
            ss[c * 2 + 1] = sh;
        }
    }
    __syncthreads();
    if (tr >= trows) return;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e] = ss[(tc * 8 + e) * 2];
        sh[e] = ss[(tc * 8 + e) * 2 + 1];
    }
    int pix = p0 + tr;
    if (p.pre) {      // the batch fetched at the top
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (pix + u * trows < p1) {
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float f = (float)v0[u][e] * sc[e] + sh[e];
                    if (p.silu) f = silu_f(f);
                    o[e] = (f16)f;
                }
                *reinterpret_cast<f16x8*>(p.y + ((size_t)b * p.HW + pix + u * trows) * p.C + (col0 + tc) * 8) = o;
            }
        }
        pix += 4 * trows;
    }
    for (; pix + 3 * trows < p1; pix += 4 * trows) {
        f16x8 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = gn_load(p, b, pix + u * trows, col0 + tc);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = (float)v[u][e] * sc[e] + sh[e];
                if (p.silu) f = silu_f(f);
                o[e] = (f16)f;
            }
            *reinterpret_cast<f16x8*>(p.y + ((size_t)b * p.HW + pix + u * trows) * p.C + (col0 + tc) * 8) = o;
        }
    }
    for (; pix < p1; pix += trows) {
        const f16x8 v = gn_load(p, b, pix, col0 + tc);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = (float)v[e] * sc[e] + sh[e];
            if (p.silu) f = silu_f(f);
            o[e] = (f16)f;
        }
        *reinterpret_cast<f16x8*>(p.y + ((size_t)b * p.HW + pix) * p.C + (col0 + tc) * 8) = o;
    }
}

// Single-launch GroupNorm for the small tensors of the deep UNet levels (HW <= 256): grid (column blocks, B), 1024 threads.
// A block owns the MINIMAL column block of whole groups and whole 16-B chunks (L chunk columns) for ALL pixels of one
// sample, so statistics never leave the block: pass 1 accumulates {sum, sumsq} with eight independent 16-B loads in
// flight per thread (the tensor is short -- latency, not bandwidth, is the cost), fixed-order LDS folds give mean /
// rstd per group (deterministic), pass 2 re-reads the (L2-resident) slab, applies scale/shift (+FiLM) (+SiLU) and
// stores.  One launch instead of gn_stats + gn_apply for ~30 of the 61 GroupNorms of a UNet evaluation.
constexpr int GNF_THREADS = 1024;
constexpr int GNF_FOLD = 16;
constexpr int GNF_KEEP = 6;      // SLAB variant: pixels a thread keeps in registers between the two passes

// SLAB = true: the input is not a tensor but the fp32 split-K slabs of the conv / Dense that produces it
// (mdx_groupnorm_from_splitk_f16): element (m, n) = fp16( bias[n] + sum_z ws[z][m][n] + rowbias[b][n] + residual[m][n] ),
// the additions in exactly the order of splitk_reduce_kernel, so the result is bit-identical to reduce-then-GroupNorm.  The
// producer's fp16 output is stored on the way (later consumers read it), the values stay in registers for the second pass.
// Round 6: every load of an element's chain -- bias, time-embedding row, residual and the first eight slabs -- is requested before the
// first add.  The small operands go through buffer descriptors whose bounds check answers zero for an operand the producer does not
// have (so the loads are unconditional: a conditional load merged with a default made the compiler wait for it on the spot, one
// exposed L2 / HBM round trip per operand in a kernel that is nothing but round trips -- 64 blocks of 320 busy threads at the 8 x 8
// level); the adds keep the order of splitk_reduce_kernel: same bits.
struct GnSlabRsrc {
    __amdgpu_buffer_rsrc_t bias, rowbias, residual;
};

__device__ __forceinline__ GnSlabRsrc gn_slab_rsrc(const MdxSplitInfo& sp) {
    GnSlabRsrc r;
    r.bias = make_rsrc(sp.bias, sp.bias ? (unsigned)sp.N * 4u : 0u);
    r.rowbias = make_rsrc(sp.rowbias, sp.rowbias ? (unsigned)sp.B * (unsigned)sp.rowbias_ld * 4u : 0u);
    const size_t rbytes = (size_t)sp.M * (size_t)sp.residual_ld * 2;
    r.residual = make_rsrc(sp.residual, sp.residual ? (unsigned)(rbytes < 0x7fffffffull ? rbytes : 0x7fffffffull) : 0u);
    return r;
}

typedef unsigned gn_u32x4 __attribute__((ext_vector_type(4)));

template <int UMAX>
__device__ __forceinline__ f16x8 gn_slab_load(const MdxSplitInfo& sp, const GnSlabRsrc& rs, int b, int m, int n) {
    const gn_u32x4 b0 = __builtin_amdgcn_raw_buffer_load_b128(rs.bias, (unsigned)n * 4u, 0, 0);
    const gn_u32x4 b1 = __builtin_amdgcn_raw_buffer_load_b128(rs.bias, (unsigned)n * 4u, 16, 0);
    const unsigned rbo = ((unsigned)b * (unsigned)sp.rowbias_ld + (unsigned)n) * 4u;
    const gn_u32x4 r0 = __builtin_amdgcn_raw_buffer_load_b128(rs.rowbias, rbo, 0, 0);
    const gn_u32x4 r1 = __builtin_amdgcn_raw_buffer_load_b128(rs.rowbias, rbo, 16, 0);
    const gn_u32x4 rr = __builtin_amdgcn_raw_buffer_load_b128(rs.residual, ((unsigned)m * (unsigned)sp.residual_ld + (unsigned)n) * 2u, 0, 0);
    float f[8];
    {
        const f32x4 x0 = __builtin_bit_cast(f32x4, b0), x1 = __builtin_bit_cast(f32x4, b1);
        f[0] = x0[0]; f[1] = x0[1]; f[2] = x0[2]; f[3] = x0[3]; f[4] = x1[0]; f[5] = x1[1]; f[6] = x1[2]; f[7] = x1[3];
    }
    const size_t slab = (size_t)sp.M * sp.N;
    const float* base = sp.ws + (size_t)m * sp.N + n;
    // up to eight slabs' loads in flight per thread, then four / two / one; the additions stay in slab order, so the sum keeps its bits
    int z = 0;
    auto batch = [&](auto uc) {
        constexpr int U = decltype(uc)::value;
        float4 a[U], c[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a[u] = *reinterpret_cast<const float4*>(base + (size_t)(z + u) * slab);
            c[u] = *reinterpret_cast<const float4*>(base + (size_t)(z + u) * slab + 4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f[0] += a[u].x; f[1] += a[u].y; f[2] += a[u].z; f[3] += a[u].w;
            f[4] += c[u].x; f[5] += c[u].y; f[6] += c[u].z; f[7] += c[u].w;
        }
        z += U;
    };
    // (UMAX = 8 where a thread keeps at most two elements between the passes -- the 8 x 8 and 16 x 16 levels, where the launch is
    // nothing but this chain; 4 with six kept elements: eight in flight spill under the 128-register cap of a 1024-thread block)
    if constexpr (UMAX >= 8) {
        while (z + 8 <= sp.nsplit) batch(std::integral_constant<int, 8>{});
        if (z + 4 <= sp.nsplit) batch(std::integral_constant<int, 4>{});
    } else {
        while (z + 4 <= sp.nsplit) batch(std::integral_constant<int, 4>{});
    }
    if (z + 2 <= sp.nsplit) batch(std::integral_constant<int, 2>{});
    if (z < sp.nsplit) batch(std::integral_constant<int, 1>{});
    if (sp.rowbias) {
        const f32x4 x0 = __builtin_bit_cast(f32x4, r0), x1 = __builtin_bit_cast(f32x4, r1);
        f[0] += x0[0]; f[1] += x0[1]; f[2] += x0[2]; f[3] += x0[3]; f[4] += x1[0]; f[5] += x1[1]; f[6] += x1[2]; f[7] += x1[3];
    }
    if (sp.residual) {
        const f16x8 r = __builtin_bit_cast(f16x8, rr);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] += (float)r[e];
    }
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (f16)f[e];
    *reinterpret_cast<f16x8*>(sp.out + (size_t)m * sp.N + n) = o;
    return o;
}

// NT = threads per block: 1024, or 256 for launches whose (column block, sample) grid fills the chip on its own (UNet batch >= 8:
// 512+ blocks) -- a quarter of the threads per barrier, four times the blocks per CU.
// KEEPN > 0 (tensor input, round 5): every thread owns at most KEEPN pixels (checked on the host), loads them ONCE with all loads
// in flight together and keeps them in registers for the second pass, like the SLAB form: the launch is one round trip of
// loads instead of two.  The additions run in the same pixel order as the two-pass loops: same bits.
constexpr int GNF_KEEPN = 6;      // (gn_fused applies up to HW * L * 16 <= 64 KiB: at most 5 pixels per thread)
template <bool SLAB, int NT = GNF_THREADS, int KEEPN = 0>
__global__ __launch_bounds__(NT) void gn_fused_kernel(const GnParams p, const MdxSplitInfo sp) {
    extern __shared__ __attribute__((aligned(16))) float red[];   // [trows][chs][2] partials, then the folds
    mdx_kernarg_touch<sizeof(GnParams) + sizeof(MdxSplitInfo)>();
    const int b = blockIdx.y, cb = blockIdx.x;
    const int col0 = cb * p.cw;
    const int cols = min(p.cw, p.CC - col0);
    const int chs = cols * 8;
    const int tid = threadIdx.x;
    const int trows = NT / cols;
    const int tc = tid % cols, tr = tid / cols;
    const bool active = tr < trows;
    constexpr int KN = SLAB ? (KEEPN > 0 ? KEEPN : GNF_KEEP) : (KEEPN > 0 ? KEEPN : 1);      // SLAB: KEEPN = elements per thread (host-checked)
    f16x8 keep[KN];
    // (p.pre) this thread's channel's affine parameters (chs <= min(512, NT): one per thread), fetched first: see GnParams.pre
    float pg = 0.f, pb = 0.f, pm = 0.f, pf = 0.f;
    if (p.pre && tid < chs) {
        pg = p.gamma[col0 * 8 + tid];
        pb = p.beta[col0 * 8 + tid];
        if (p.scale) {
            pm = p.scale[(size_t)b * p.mod_ld + col0 * 8 + tid];
            pf = p.shift[(size_t)b * p.mod_ld + col0 * 8 + tid];
        }
    }
    if (active) {
        float s[8], q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
        int pix = tr;
        if constexpr (!SLAB && KEEPN > 0) {
#pragma unroll
            for (int k = 0; k < KEEPN; ++k)
                if (tr + k * trows < p.HW) keep[k] = gn_load(p, b, tr + k * trows, col0 + tc);
#pragma unroll
            for (int k = 0; k < KEEPN; ++k)
                if (tr + k * trows < p.HW) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float f = (float)keep[k][e];
                        s[e] += f;
                        q[e] += f * f;
                    }
                }
            pix = p.HW;      // nothing left for the loops below
        }
        if constexpr (SLAB) {
            const GnSlabRsrc srs = gn_slab_rsrc(sp);
#pragma unroll
            for (int k = 0; k < KN; ++k) {
                const int px = tr + k * trows;
                if (px < p.HW) {
                    keep[k] = gn_slab_load<(KN <= 2 ? 8 : 4)>(sp, srs, b, b * p.HW + px, (col0 + tc) * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float f = (float)keep[k][e];
                        s[e] += f;
                        q[e] += f * f;
                    }
                }
            }
            pix = p.HW;      // nothing left for the tensor loops below
        }
        for (; pix + 7 * trows < p.HW; pix += 8 * trows) {
            f16x8 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = gn_load(p, b, pix + u * trows, col0 + tc);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = (float)v[u][e];
                    s[e] += f;
                    q[e] += f * f;
                }
        }
        for (; pix + 3 * trows < p.HW; pix += 4 * trows) {      // (same order of additions as one pixel at a time: same bits)
            f16x8 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = gn_load(p, b, pix + u * trows, col0 + tc);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = (float)v[u][e];
                    s[e] += f;
                    q[e] += f * f;
                }
        }
        for (; pix < p.HW; pix += trows) {
            const f16x8 v = gn_load(p, b, pix, col0 + tc);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)v[e];
                s[e] += f;
                q[e] += f * f;
            }
        }
        float* dst = red + ((size_t)tr * chs + tc * 8) * 2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            dst[e * 2] = s[e];
            dst[e * 2 + 1] = q[e];
        }
    }
    __syncthreads();
    // fold the trows partial rows per channel in two fixed-order levels (trows -> GNF_FOLD -> 1)
    float* f16p = red + (size_t)NT * 8 * 2;      // [GNF_FOLD][chs][2]
    float* csum = f16p + (size_t)GNF_FOLD * 512 * 2;      // [chs][2]
    for (int i = tid; i < GNF_FOLD * chs; i += NT) {
        const int g16 = i / chs, c = i - g16 * chs;
        float s = 0.f, q = 0.f;
        for (int r = g16; r < trows; r += GNF_FOLD) {
            s += red[((size_t)r * chs + c) * 2];
            q += red[((size_t)r * chs + c) * 2 + 1];
        }
        f16p[(g16 * chs + c) * 2] = s;
        f16p[(g16 * chs + c) * 2 + 1] = q;
    }
    __syncthreads();
    for (int c = tid; c < chs; c += NT) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int g16 = 0; g16 < GNF_FOLD; ++g16) {
            s += f16p[(g16 * chs + c) * 2];
            q += f16p[(g16 * chs + c) * 2 + 1];
        }
        csum[c * 2] = s;
        csum[c * 2 + 1] = q;
    }
    __syncthreads();
    // per-group mean / rstd (8 lanes per group, fixed shuffle order), then per-channel scale / shift
    float* gstat = csum + 512 * 2;                         // [groups in block][2]
    float* ss = gstat + 64 * 2;                            // [chs][2]
    const int ng = chs / p.cpg;
    {
        const int g = tid >> 3, j = tid & 7;
        float s = 0.f, q = 0.f;
        if (g < ng) {
            for (int c = g * p.cpg + j; c < (g + 1) * p.cpg; c += 8) {
                s += csum[c * 2];
                q += csum[c * 2 + 1];
            }
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            s += __shfl_xor(s, o, 64);
            q += __shfl_xor(q, o, 64);
        }
        if (g < ng && j == 0) {
            const float inv = 1.0f / ((float)p.cpg * (float)p.HW);
            const float mean = s * inv;
            float var = q * inv - mean * mean;
            var = var < 0.f ? 0.f : var;
            gstat[g * 2] = mean;
            gstat[g * 2 + 1] = rsqrtf(var + p.eps);
        }
    }
    __syncthreads();
    if (p.pre) {
        if (tid < chs) {
            const int g = tid / p.cpg;
            float a = pg * gstat[g * 2 + 1];
            float sh = pb - gstat[g * 2] * a;
            if (p.scale) {   // (x_hat*gamma + beta) * (1 + scale) + shift
                const float m1 = 1.0f + pm;
                a *= m1;
                sh = sh * m1 + pf;
            }
            ss[tid * 2] = a;
            ss[tid * 2 + 1] = sh;
        }
    } else {
        for (int c = tid; c < chs; c += NT) {
            const int g = c / p.cpg;
            float a = p.gamma[col0 * 8 + c] * gstat[g * 2 + 1];
            float sh = p.beta[col0 * 8 + c] - gstat[g * 2] * a;
            if (p.scale) {   // (x_hat*gamma + beta) * (1 + scale) + shift
                const float m1 = 1.0f + p.scale[(size_t)b * p.mod_ld + col0 * 8 + c];
                a *= m1;
                sh = sh * m1 + p.shift[(size_t)b * p.mod_ld + col0 * 8 + c];
            }
            ss[c * 2] = a;
            ss[c * 2 + 1] = sh;
        }
    }
    __syncthreads();
    if (!active) return;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e] = ss[(tc * 8 + e) * 2];
        sh[e] = ss[(tc * 8 + e) * 2 + 1];
    }
    auto apply = [&](const f16x8& v, int pix) {
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = (float)v[e] * sc[e] + sh[e];
            if (p.silu) f = silu_f(f);
            o[e] = (f16)f;
        }
        *reinterpret_cast<f16x8*>(p.y + ((size_t)b * p.HW + pix) * p.C + (col0 + tc) * 8) = o;
    };
    if constexpr (SLAB || KEEPN > 0) {
#pragma unroll
        for (int k = 0; k < KN; ++k)
            if (tr + k * trows < p.HW) apply(keep[k], tr + k * trows);
        return;
    }
    int pix = tr;
    for (; pix + 7 * trows < p.HW; pix += 8 * trows) {
        f16x8 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = gn_load(p, b, pix + u * trows, col0 + tc);
#pragma unroll
        for (int u = 0; u < 8; ++u) apply(v[u], pix + u * trows);
    }
    for (; pix + 3 * trows < p.HW; pix += 4 * trows) {
        f16x8 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = gn_load(p, b, pix + u * trows, col0 + tc);
#pragma unroll
        for (int u = 0; u < 4; ++u) apply(v[u], pix + u * trows);
    }
    for (; pix < p.HW; pix += trows) apply(gn_load(p, b, pix, col0 + tc), pix);
}

int gcd_i(int a, int b) { return b ? gcd_i(b, a % b) : a; }

// Fill the launch geometry: column blocks of whole groups (<= 64 chunk columns), <= 64 pixel slabs per sample,
// aiming at ~1024 blocks so that even the 8x8-latent levels fill the 256 CUs.
void gn_geometry(GnParams& p) {
    const int lcm = p.cpg / gcd_i(p.cpg, 8) * 8;   // channels spanned by a whole number of groups AND of chunks
    const int L = lcm / 8;                          // ... in chunk columns
    int cw = (64 / L) * L;
    if (cw == 0) cw = L;                            // a single group wider than 64 chunks (cpg > 512): one group/block
    if (cw > p.CC) cw = p.CC;
    // every gn_apply block re-folds all slab partials of its groups: keep the fold short for the UNet's small tensors
    // (measured: 256 slabs cost the 64x64-latent UNet +17 % GroupNorm time) and only widen for image-sized ones
    // (VAE decoder at 256^2 / 512^2, GLIDE super-resolution), which otherwise leave 3/4 of the CUs idle
    const int cap = p.HW >= 16384 ? GN_MAX_NBLK : 64;   // keep in step with mdx_groupnorm_ws_floats
    const int max_by_pix = (p.HW + 3) / 4;          // at least ~4 pixels per slab
    // Wide column blocks give the longest contiguous runs per pixel, but at UNet batch 2 the slab cap then leaves a
    // 64x64x320 tensor with 128 blocks for 256 CUs: narrow the column blocks (whole groups at a time) until the grid
    // has >= 512 blocks (measured -1.4 % on the whole UNet evaluation at batch 2; batch 16 already has 1024 and
    // loses 1 % with narrow columns, so it keeps the wide ones)
    const int min_blocks = mdx_opt(MDX_OPT_GN_MIN_BLOCKS);
    const int slabs = cap < max_by_pix ? cap : max_by_pix;
    while (cw > L && ((p.CC + cw - 1) / cw) * p.B * slabs < min_blocks) cw -= L;
    p.cw = cw;
    p.ncb = (p.CC + cw - 1) / cw;
    // (four times the slabs for tensors of tens of MB, as the column-statistics path does, measured -0.8 % on GLIDE's 128^2 / 256^2
    // tensors -- the statistics pass folds more partials: not taken here)
    int nblk = (GN_TARGET_BLOCKS + p.ncb * p.B - 1) / (p.ncb * p.B);
    if (nblk > cap) nblk = cap;
    if (nblk > max_by_pix) nblk = max_by_pix;
    if (nblk < 1) nblk = 1;
    p.pix = (p.HW + nblk - 1) / nblk;
    p.nblk = (p.HW + p.pix - 1) / p.pix;
}

// LayerNorm: one wave per row, up to 8 chunks (C <= 4096) held in registers; exact two-pass statistics.
template <int MAXCH>
__global__ __launch_bounds__(256) void ln_kernel(const f16* __restrict__ x, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, f16* __restrict__ y, int rows,
                                                 int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int cc = C / 8;
    const f16* xr = x + (size_t)row * C;
    f16x8 v[MAXCH];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MAXCH; ++k) {
        const int col = lane + k * 64;
        if (col < cc) {
            v[k] = *reinterpret_cast<const f16x8*>(xr + col * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += (float)v[k][e];
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < MAXCH; ++k) {
        const int col = lane + k * 64;
        if (col < cc) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = (float)v[k][e] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    f16* yr = y + (size_t)row * C;
#pragma unroll
    for (int k = 0; k < MAXCH; ++k) {
        const int col = lane + k * 64;
        if (col < cc) {
            const float4 g0 = *reinterpret_cast<const float4*>(gamma + col * 8);
            const float4 g1 = *reinterpret_cast<const float4*>(gamma + col * 8 + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(beta + col * 8);
            const float4 b1 = *reinterpret_cast<const float4*>(beta + col * 8 + 4);
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (f16)(((float)v[k][e] - mean) * rstd * gg[e] + bb[e]);
            *reinterpret_cast<f16x8*>(yr + col * 8) = o;
        }
    }
}

}  // namespace

extern "C" size_t mdx_groupnorm_ws_floats(int B, int HW, int C, int groups) {
    (void)C;
    return (size_t)B * (HW >= 16384 ? GN_MAX_NBLK : 64) * groups * 2;   // slab cap of gn_geometry
}

static int groupnorm_impl(const void* x1, int C1, const void* x2, int C2, const float* gamma, const float* beta,
                          const float* scale, const float* shift, int mod_ld, void* y, int B, int HW, int groups,
                          float eps, int silu, float* ws, mdx_stream_t s, const float* cs1 = nullptr, int nrb1 = 0,
                          const float* cs2 = nullptr, int nrb2 = 0) {
    MDX_REQUIRE(x1 && gamma && beta && y && (ws || cs1), "mdx_groupnorm_f16: null pointer");
    MDX_REQUIRE((C2 == 0) == (x2 == nullptr), "mdx_groupnorm_f16: x2/C2 mismatch");
    MDX_REQUIRE((scale == nullptr) == (shift == nullptr), "mdx_groupnorm_scaleshift_f16: scale/shift mismatch");
    const int C = C1 + C2;
    MDX_REQUIRE(C1 > 0 && C1 % 8 == 0 && C2 % 8 == 0, "mdx_groupnorm_f16: channels must be multiples of 8");
    MDX_REQUIRE(groups > 0 && groups <= 32 && C % groups == 0, "mdx_groupnorm_f16: C=%d not divisible by groups=%d (<= 32)", C, groups);
    MDX_REQUIRE(C <= GN_MAX_C, "mdx_groupnorm_f16: C=%d exceeds %d", C, GN_MAX_C);
    MDX_REQUIRE(B > 0 && HW > 0 && B <= 65535, "mdx_groupnorm_f16: bad extents");
    MDX_REQUIRE(!scale || mod_ld >= C, "mdx_groupnorm_scaleshift_f16: mod_ld < C");
    GnParams p{};
    p.x1 = (const f16*)x1;
    p.x2 = (const f16*)x2;
    p.gamma = gamma;
    p.beta = beta;
    p.y = (f16*)y;
    p.ws = ws;
    p.C1 = C1;
    p.C2 = C2;
    p.C = C;
    p.CC1 = C1 / 8;
    p.CC = C / 8;
    p.B = B;
    p.HW = HW;
    p.groups = groups;
    p.cpg = C / groups;
    p.eps = eps;
    p.silu = silu;
    p.scale = scale;
    p.shift = shift;
    p.mod_ld = mod_ld;
    p.cs1 = cs1;
    p.cs2 = cs2;
    p.nrb1 = nrb1;
    p.nrb2 = nrb2;
    p.pre = mdx_opt(MDX_OPT_GN_PREFETCH) ? 1 : 0;
    hipStream_t st = (hipStream_t)s;
    if (cs1) {
        // one launch: gn_apply with its statistics folded from the producers' column partials.  Minimal column blocks
        // (whole groups AND whole 16-byte chunks) keep the fold short: chs * nrb float2 loads per block.
        MDX_REQUIRE(nrb1 > 0 && (C2 == 0 || (cs2 && nrb2 > 0)), "mdx_groupnorm_colstats_f16: missing column statistics");
        const int lcm = p.cpg / gcd_i(p.cpg, 8) * 8;
        const int L = lcm / 8;
        MDX_REQUIRE(L <= 64, "mdx_groupnorm_colstats_f16: %d channels per group is not supported", p.cpg);
        int minc = mdx_opt(MDX_OPT_GN_COL_CHUNKS);
        if (minc < 1) minc = 1;
        if (minc > 64) minc = 64;
        int cw = L * ((minc + L - 1) / L);   // >= 16 * gn_col_chunks bytes per pixel row (default 64)
        if (cw > 64) cw = L * (64 / L);      // (<= 64 chunk columns per block: one per thread row)
        if (cw > p.CC) cw = p.CC;
        // Big tensors (UNet batch >= 8 at the 64 x 64 / 32 x 32 levels: >= gn_wide_rows pixel rows in all) are bandwidth-, not
        // latency-bound, and 80-byte column blocks read every 128-byte line of a pixel row from two or three blocks (1.3 TB/s on a
        // 84 MB GroupNorm): the widest column block of whole groups, line-aligned where the group width allows, and more slabs.
        // The fold then costs chs * nrb loads per block, so the planner hands such launches pre-folded statistics (nrb = 1).
        int target = GN_TARGET_BLOCKS;
        const int wide_rows = mdx_opt(MDX_OPT_GN_WIDE_ROWS);
        // 1024 blocks keep too few bytes in flight for a tensor of tens of MB (2.8 TB/s on 84-168 MB at UNet batch 16): four times
        // the slabs from gn_boost_mb on (tools/exp/r04k_gn_bench.py, profiles/r04_gn_bench.txt: 64 x 64 x 640 at batch 16 60.8 -> 41.8 us,
        // 32 x 32 x 1280 36.2 -> 25.8, 64 x 64 x 320 28.4 -> 25.9; the batch-2 tensors, 10 MB, lose 3 us and stay below the threshold)
        const int boost_mb = mdx_opt(MDX_OPT_GN_BOOST_MB);
        if (boost_mb > 0 && (size_t)B * HW * C * 2 >= ((size_t)boost_mb << 20)) target = 4 * GN_TARGET_BLOCKS;
        if (wide_rows > 0 && (long)B * HW >= wide_rows) {
            const int cap = p.CC < 64 ? p.CC : 64;
            const int l8 = L / gcd_i(L, 8) * 8;          // whole groups AND whole 128-byte lines
            cw = l8 <= cap ? (cap / l8) * l8 : (cap / L) * L;
            if (cw < L) cw = L > p.CC ? p.CC : L;
            target = 4 * GN_TARGET_BLOCKS;
        }
        p.cw = cw;
        p.ncb = (p.CC + cw - 1) / cw;
        int nblk = (target + p.ncb * B - 1) / (p.ncb * B);
        const int max_by_pix = (HW + 3) / 4;
        if (nblk > 256) nblk = 256;
        if (nblk > max_by_pix) nblk = max_by_pix;
        if (nblk < 1) nblk = 1;
        p.pix = (HW + nblk - 1) / nblk;
        p.nblk = (HW + p.pix - 1) / p.pix;
        dim3 grid(p.nblk, p.ncb, B);
        hipLaunchKernelGGL(gn_apply_kernel, grid, dim3(256), ((size_t)cw * 8 * 2 + 64 + (size_t)cw * 8 * 2) * sizeof(float), st, p);
        MDX_LAUNCH_CHECK("mdx_groupnorm_colstats_f16");
        return MDX_OK;
    }
    {
        // fused single-launch path when one block can walk all pixels of its column block quickly: <= 64 KiB per block,
        // i.e. the 16x16 and 8x8 latent levels (measured per shape: 8.7 vs 12.4 us at HW = 256, 7.5 vs 10.7 at HW = 64;
        // at HW >= 1024 the few, long blocks lose to the two-launch slab scheme)
        const int lcm = p.cpg / gcd_i(p.cpg, 8) * 8;
        const int L = lcm / 8;
        if (mdx_opt(MDX_OPT_GN_FUSED) && L <= 64 && (size_t)HW * L * 16 <= (64u << 10)) {
            p.cw = L > p.CC ? p.CC : L;
            p.ncb = (p.CC + p.cw - 1) / p.cw;
            p.nblk = 1;
            p.pix = HW;
            constexpr size_t lds = ((size_t)GNF_THREADS * 8 * 2 + (size_t)GNF_FOLD * 512 * 2 + 512 * 2 + 64 * 2 + 512 * 2) * sizeof(float);
            if (mdx_opt(MDX_OPT_GN_FUSED_SMALL) && p.ncb * B >= 512 && p.cw <= 32) {     // the grid fills the chip: 256-thread blocks
                constexpr size_t lds4 = ((size_t)256 * 8 * 2 + (size_t)GNF_FOLD * 512 * 2 + 512 * 2 + 64 * 2 + 512 * 2) * sizeof(float);
                static MdxPerDeviceOnce attr4_once;
                if (attr4_once.first()) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_fused_kernel<false, 256>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
                }
                hipLaunchKernelGGL((gn_fused_kernel<false, 256>), dim3(p.ncb, B), dim3(256), lds4, st, p, MdxSplitInfo{});
                MDX_LAUNCH_CHECK("mdx_groupnorm_f16(fused, 256 threads)");
                return MDX_OK;
            }
            // every thread's pixels fit its registers (HW <= 8 rows of threads: the 8 x 8 ... 32 x 32 levels): one pass of loads
            if (p.pre && (HW + GNF_THREADS / p.cw - 1) / (GNF_THREADS / p.cw) <= GNF_KEEPN) {
                static MdxPerDeviceOnce attrk_once;
                if (attrk_once.first()) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_fused_kernel<false, GNF_THREADS, GNF_KEEPN>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                }
                hipLaunchKernelGGL((gn_fused_kernel<false, GNF_THREADS, GNF_KEEPN>), dim3(p.ncb, B), dim3(GNF_THREADS), lds, st, p,
                                   MdxSplitInfo{});
                MDX_LAUNCH_CHECK("mdx_groupnorm_f16(fused, one pass)");
                return MDX_OK;
            }
            static MdxPerDeviceOnce attr_once;
            if (attr_once.first()) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_fused_kernel<false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            }
            hipLaunchKernelGGL(gn_fused_kernel<false>, dim3(p.ncb, B), dim3(GNF_THREADS), lds, st, p, MdxSplitInfo{});
            MDX_LAUNCH_CHECK("mdx_groupnorm_f16(fused)");
            return MDX_OK;
        }
    }
    gn_geometry(p);
    MDX_REQUIRE((p.cw * 8) % p.cpg == 0 || p.ncb == 1, "mdx_groupnorm_f16: internal geometry error");
    dim3 grid(p.nblk, p.ncb, B);
    MDX_REQUIRE(p.cw <= 64, "mdx_groupnorm_f16: %d channels per group is not supported", p.cpg);
    const int cols = p.cw;
    // stats LDS: [trows][cols*8][2] floats with trows*cols <= 256 for every (possibly narrower, last) column block
    hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(256), ((size_t)256 * 8 * 2 + 512 * 2) * sizeof(float), st, p);
    MDX_LAUNCH_CHECK("mdx_groupnorm_f16(stats)");
    hipLaunchKernelGGL(gn_apply_kernel, grid, dim3(256), ((size_t)cols * 8 * 2 + 64) * sizeof(float), st, p);
    MDX_LAUNCH_CHECK("mdx_groupnorm_f16(apply)");
    return MDX_OK;
}

extern "C" int mdx_groupnorm_f16(const void* x1, int C1, const void* x2, int C2, const float* gamma,
                                 const float* beta, void* y, int B, int HW, int groups, float eps, int silu,
                                 float* ws, mdx_stream_t s) {
    return groupnorm_impl(x1, C1, x2, C2, gamma, beta, nullptr, nullptr, 0, y, B, HW, groups, eps, silu, ws, s);
}

// Geometry of the one-block-per-(sample, column block) scheme for a producer output [B][HW][C]: 0 if it does not apply.
static int gn_from_splitk_geometry(const MdxSplitInfo& sp, int groups, GnParams& p) {
    if (groups <= 0 || groups > 32 || sp.N % groups || sp.N % 8 || sp.nsplit < 2) return 0;
    p.C1 = p.C = sp.N;
    p.C2 = 0;
    p.CC1 = p.CC = sp.N / 8;
    p.B = sp.B;
    p.HW = sp.HoWo;
    p.groups = groups;
    p.cpg = sp.N / groups;
    const int lcm = p.cpg / gcd_i(p.cpg, 8) * 8;
    const int L = lcm / 8;
    if (L > 64 || (size_t)p.HW * L * 16 > (64u << 10)) return 0;
    p.cw = L > p.CC ? p.CC : L;
    p.ncb = (p.CC + p.cw - 1) / p.cw;
    p.nblk = 1;
    p.pix = p.HW;
    const int trows = GNF_THREADS / p.cw;
    return (p.HW + trows - 1) / trows <= GNF_KEEP;
}

extern "C" int mdx_groupnorm_from_splitk_ok(const mdx_gemm_desc* prod, int groups) {
    MdxSplitInfo sp{};
    GnParams p{};
    return prod && mdx_internal_split_info(prod, &sp) == MDX_OK && gn_from_splitk_geometry(sp, groups, p) ? 1 : 0;
}

extern "C" int mdx_groupnorm_from_splitk_f16(const mdx_gemm_desc* prod, const float* gamma, const float* beta, void* y,
                                             int groups, float eps, int silu, mdx_stream_t s) {
    MDX_REQUIRE(prod && gamma && beta && y, "mdx_groupnorm_from_splitk_f16: null pointer");
    MdxSplitInfo sp{};
    int rc = mdx_internal_split_info(prod, &sp);
    if (rc != MDX_OK) return rc;
    GnParams p{};
    MDX_REQUIRE(sp.ws && gn_from_splitk_geometry(sp, groups, p),
                "mdx_groupnorm_from_splitk_f16: producer does not split K or its output does not fit one block per column block");
    p.x1 = sp.out;
    p.gamma = gamma;
    p.beta = beta;
    p.y = (f16*)y;
    p.eps = eps;
    p.silu = silu;
    p.pre = mdx_opt(MDX_OPT_GN_PREFETCH) ? 1 : 0;
    constexpr size_t lds = ((size_t)GNF_THREADS * 8 * 2 + (size_t)GNF_FOLD * 512 * 2 + 512 * 2 + 64 * 2 + 512 * 2) * sizeof(float);
    // elements a thread keeps between the passes: 1 (8 x 8 level), 2 (16 x 16), else up to GNF_KEEP -- fewer kept elements leave
    // the registers for eight slabs' loads in flight (gn_slab_load)
    const int trows = GNF_THREADS / p.cw;
    const int per_thread = (p.HW + trows - 1) / trows;
    auto launch = [&](auto kernel) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kernel, dim3(p.ncb, sp.B), dim3(GNF_THREADS), lds, (hipStream_t)s, p, sp);
    };
    if (per_thread <= 1)
        launch(&gn_fused_kernel<true, GNF_THREADS, 1>);
    else if (per_thread <= 2)
        launch(&gn_fused_kernel<true, GNF_THREADS, 2>);
    else
        launch(&gn_fused_kernel<true, GNF_THREADS, 0>);
    MDX_LAUNCH_CHECK("mdx_groupnorm_from_splitk_f16");
    return MDX_OK;
}

// Two-level fold of a producer's column partials: dst[b][j][c] = sum over the `f` consecutive row blocks j*f .. j*f+f-1 of
// src[b][.][c] (fixed order: deterministic).  Tensors with hundreds of row blocks per sample (GLIDE's 128 x 128 / 256 x 256
// levels: 512 HALO patches) are folded ONCE here to <= 64 blocks instead of in every gn_apply block.
__global__ __launch_bounds__(256) void colstats_fold_kernel(const float2* __restrict__ src, float2* __restrict__ dst, int nrb,
                                                            int nrb2, int f, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int j = blockIdx.y, b = blockIdx.z;
    if (c >= C) return;
    const float2* s = src + ((size_t)b * nrb + (size_t)j * f) * C + c;
    const int n = min(f, nrb - j * f);
    float su = 0.f, sq = 0.f;
    int k = 0;
    for (; k + 4 <= n; k += 4) {
        const float2 v0 = s[(size_t)k * C], v1 = s[(size_t)(k + 1) * C], v2 = s[(size_t)(k + 2) * C], v3 = s[(size_t)(k + 3) * C];
        su += v0.x; sq += v0.y; su += v1.x; sq += v1.y; su += v2.x; sq += v2.y; su += v3.x; sq += v3.y;
    }
    for (; k < n; ++k) {
        const float2 v = s[(size_t)k * C];
        su += v.x;
        sq += v.y;
    }
    dst[((size_t)b * nrb2 + j) * C + c] = make_float2(su, sq);
}

extern "C" int mdx_colstats_fold_f32(const float* src, int nrb, float* dst, int nrb2, int B, int C, mdx_stream_t s) {
    MDX_REQUIRE(src && dst && nrb > 0 && nrb2 > 0 && nrb2 <= nrb && B > 0 && C > 0, "mdx_colstats_fold_f32: bad arguments");
    const int f = (nrb + nrb2 - 1) / nrb2;
    MDX_REQUIRE((nrb + f - 1) / f == nrb2, "mdx_colstats_fold_f32: nrb2 must be ceil(nrb / f) for an integer fold factor f");
    hipLaunchKernelGGL(colstats_fold_kernel, dim3((C + 255) / 256, nrb2, B), dim3(256), 0, (hipStream_t)s,
                       reinterpret_cast<const float2*>(src), reinterpret_cast<float2*>(dst), nrb, nrb2, f, C);
    MDX_LAUNCH_CHECK("mdx_colstats_fold_f32");
    return MDX_OK;
}

extern "C" int mdx_groupnorm_colstats_f16(const void* x1, int C1, const float* cs1, int nrb1, const void* x2, int C2,
                                          const float* cs2, int nrb2, const float* gamma, const float* beta,
                                          const float* scale, const float* shift, int mod_ld, void* y, int B, int HW,
                                          int groups, float eps, int silu, mdx_stream_t s) {
    MDX_REQUIRE(cs1, "mdx_groupnorm_colstats_f16: null column statistics");
    return groupnorm_impl(x1, C1, x2, C2, gamma, beta, scale, shift, mod_ld, y, B, HW, groups, eps, silu, nullptr, s, cs1,
                          nrb1, cs2, nrb2);
}

extern "C" int mdx_groupnorm_scaleshift_f16(const void* x1, int C1, const void* x2, int C2, const float* gamma,
                                            const float* beta, const float* scale, const float* shift, int mod_ld,
                                            void* y, int B, int HW, int groups, float eps, int silu, float* ws,
                                            mdx_stream_t s) {
    MDX_REQUIRE(scale && shift, "mdx_groupnorm_scaleshift_f16: null modulation pointer");
    return groupnorm_impl(x1, C1, x2, C2, gamma, beta, scale, shift, mod_ld, y, B, HW, groups, eps, silu, ws, s);
}

extern "C" int mdx_layernorm_f16(const void* x, const float* gamma, const float* beta, void* y, int rows, int C,
                                 float eps, mdx_stream_t s) {
    MDX_REQUIRE(x && gamma && beta && y, "mdx_layernorm_f16: null pointer");
    MDX_REQUIRE(C > 0 && C % 8 == 0 && C <= 4096, "mdx_layernorm_f16: C=%d must be a multiple of 8 and <= 4096", C);
    MDX_REQUIRE(rows > 0, "mdx_layernorm_f16: rows must be positive");
    hipStream_t st = (hipStream_t)s;
    dim3 grid((rows + 3) / 4);
    const f16* xi = (const f16*)x;
    f16* yo = (f16*)y;
    if (C <= 1024)
        hipLaunchKernelGGL(ln_kernel<2>, grid, dim3(256), 0, st, xi, gamma, beta, yo, rows, C, eps);
    else if (C <= 2048)
        hipLaunchKernelGGL(ln_kernel<4>, grid, dim3(256), 0, st, xi, gamma, beta, yo, rows, C, eps);
    else
        hipLaunchKernelGGL(ln_kernel<8>, grid, dim3(256), 0, st, xi, gamma, beta, yo, rows, C, eps);
    MDX_LAUNCH_CHECK("mdx_layernorm_f16");
    return MDX_OK;
}
