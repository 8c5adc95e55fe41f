// GroupNorm(32)(+SiLU) and LayerNorm on NHWC / token-major fp16 for gfx950.  HBM-bound kernels:
// 16-byte coalesced accesses, fp32 statistics, wavefront (64-lane) shuffle reductions.
//
// GroupNorm is two launches:
//   gn_stats : every block reduces a slab of pixels to per-(sample, group) partial {sum, sumsq}
//              (deterministic: partials are written, not atomically added)
//   gn_apply : every block first folds the partials into per-channel scale/shift in LDS, then
//              streams y = silu(x * scale_c + shift_c).
// The input may be the channel concat of two tensors (UNet skip connections): the concat is
// never materialised.
#include "mdx_common.h"

namespace {

constexpr int GN_MAX_C = 4096;
constexpr int GN_PIX_PER_BLOCK = 64;

struct GnParams {
    const f16* x1;
    const f16* x2;
    const float* gamma;
    const float* beta;
    f16* y;
    float* ws;  // [B][nblk][groups][2]
    int C1, C2, C, CC1, CC;  // CCx = chunks (8 channels) per pixel
    int B, HW, groups, cpg, nblk;
    float eps;
    int silu;
};

__device__ __forceinline__ f16x8 gn_load(const GnParams& p, int b, int pix, int col) {
    if (col < p.CC1) return *reinterpret_cast<const f16x8*>(p.x1 + ((size_t)b * p.HW + pix) * p.C1 + col * 8);
    return *reinterpret_cast<const f16x8*>(p.x2 + ((size_t)b * p.HW + pix) * p.C2 + (col - p.CC1) * 8);
}

// grid (nblk, B); block 256.  LDS: per-(thread-row, channel) {sum, sumsq} partials [trows][C][2] fp32.
// Deterministic: no atomics, every reduction runs in a fixed order.
__global__ __launch_bounds__(256) void gn_stats_kernel(const GnParams p) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [trows][C][2]
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * GN_PIX_PER_BLOCK;
    const int p1 = min(p.HW, p0 + GN_PIX_PER_BLOCK);
    const int tid = threadIdx.x;
    // threads tile the (pixel, chunk-column) plane with the column FIXED per thread per pass, so
    // the per-channel accumulators live in registers: tcols = min(CC,256) columns x trows rows.
    const int tcols = p.CC < 256 ? p.CC : 256;
    const int trows = 256 / tcols;
    const int tc = tid % tcols, tr = tid / tcols;
    if (tr < trows) {
        for (int cb = 0; cb < p.CC; cb += tcols) {
            const int col = cb + tc;
            if (col >= p.CC) continue;
            float s[8], q[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
            for (int pix = p0 + tr; pix < p1; pix += trows) {
                const f16x8 v = gn_load(p, b, pix, col);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = (float)v[e];
                    s[e] += f;
                    q[e] += f * f;
                }
            }
            float* dst = red + ((size_t)tr * p.C + col * 8) * 2;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                dst[e * 2] = s[e];
                dst[e * 2 + 1] = q[e];
            }
        }
    }
    __syncthreads();
    // per-group partials: one thread per group sums its cpg channels over the trows partial rows
    if (tid < p.groups) {
        float s = 0.f, q = 0.f;
        for (int r = 0; r < trows; ++r)
            for (int c = tid * p.cpg; c < (tid + 1) * p.cpg; ++c) {
                s += red[((size_t)r * p.C + c) * 2];
                q += red[((size_t)r * p.C + c) * 2 + 1];
            }
        float* o = p.ws + (((size_t)b * p.nblk + blockIdx.x) * p.groups + tid) * 2;
        o[0] = s;
        o[1] = q;
    }
}

// grid (nblk, B); block 256.  LDS: scale/shift per channel [C][2].
__global__ __launch_bounds__(256) void gn_apply_kernel(const GnParams p) {
    extern __shared__ __attribute__((aligned(16))) float ss[];  // [C*2] then [groups*2]
    float* gstat = ss + p.C * 2;
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    if (tid < p.groups) {
        float s = 0.f, q = 0.f;
        const float* w = p.ws + ((size_t)b * p.nblk * p.groups + tid) * 2;
        for (int k = 0; k < p.nblk; ++k) {
            s += w[(size_t)k * p.groups * 2];
            q += w[(size_t)k * p.groups * 2 + 1];
        }
        const float inv = 1.0f / ((float)p.cpg * (float)p.HW);
        const float mean = s * inv;
        float var = q * inv - mean * mean;
        var = var < 0.f ? 0.f : var;
        gstat[tid * 2] = mean;
        gstat[tid * 2 + 1] = rsqrtf(var + p.eps);
    }
    __syncthreads();
    for (int c = tid; c < p.C; c += 256) {
        const int g = c / p.cpg;
        const float a = p.gamma[c] * gstat[g * 2 + 1];
        ss[c * 2] = a;
        ss[c * 2 + 1] = p.beta[c] - gstat[g * 2] * a;
    }
    __syncthreads();
    const int p0 = blockIdx.x * GN_PIX_PER_BLOCK;
    const int p1 = min(p.HW, p0 + GN_PIX_PER_BLOCK);
    const int n = (p1 - p0) * p.CC;
    for (int i = tid; i < n; i += 256) {
        const int pix = p0 + i / p.CC;
        const int col = i - (i / p.CC) * p.CC;
        const f16x8 v = gn_load(p, b, pix, col);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = (float)v[e] * ss[(col * 8 + e) * 2] + ss[(col * 8 + e) * 2 + 1];
            if (p.silu) f = silu_f(f);
            o[e] = (f16)f;
        }
        *reinterpret_cast<f16x8*>(p.y + ((size_t)b * p.HW + pix) * p.C + col * 8) = o;
    }
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
    const int nblk = (HW + GN_PIX_PER_BLOCK - 1) / GN_PIX_PER_BLOCK;
    return (size_t)B * nblk * groups * 2;
}

extern "C" int mdx_groupnorm_f16(const void* x1, int C1, const void* x2, int C2, const float* gamma,
                                 const float* beta, void* y, int B, int HW, int groups, float eps, int silu,
                                 float* ws, mdx_stream_t s) {
    MDX_REQUIRE(x1 && gamma && beta && y && ws, "mdx_groupnorm_f16: null pointer");
    MDX_REQUIRE((C2 == 0) == (x2 == nullptr), "mdx_groupnorm_f16: x2/C2 mismatch");
    const int C = C1 + C2;
    MDX_REQUIRE(C1 > 0 && C1 % 8 == 0 && C2 % 8 == 0, "mdx_groupnorm_f16: channels must be multiples of 8");
    MDX_REQUIRE(groups > 0 && groups <= 256 && C % groups == 0, "mdx_groupnorm_f16: C=%d not divisible by groups=%d", C, groups);
    MDX_REQUIRE(C <= GN_MAX_C, "mdx_groupnorm_f16: C=%d exceeds %d", C, GN_MAX_C);
    MDX_REQUIRE(B > 0 && HW > 0, "mdx_groupnorm_f16: bad extents");
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
    p.nblk = (HW + GN_PIX_PER_BLOCK - 1) / GN_PIX_PER_BLOCK;
    p.eps = eps;
    p.silu = silu;
    hipStream_t st = (hipStream_t)s;
    dim3 grid(p.nblk, B);
    const int tcols = p.CC < 256 ? p.CC : 256;
    const int trows = 256 / tcols;
    hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(256), (size_t)trows * C * 2 * sizeof(float), st, p);
    MDX_LAUNCH_CHECK("mdx_groupnorm_f16(stats)");
    hipLaunchKernelGGL(gn_apply_kernel, grid, dim3(256), ((size_t)C * 2 + groups * 2) * sizeof(float), st, p);
    MDX_LAUNCH_CHECK("mdx_groupnorm_f16(apply)");
    return MDX_OK;
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
