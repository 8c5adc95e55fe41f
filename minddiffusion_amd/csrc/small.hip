// Small / elementwise kernels of the path: layout boundary, timestep embedding, small-M Dense,
// fused sampler update, hardware-layout probes, and the library's error plumbing.
#include "mdx_common.h"

#include <string.h>

// ------------------------------------------------------------------ error plumbing
static thread_local char g_err[512] = "";

void mdx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* mdx_last_error(void) { return g_err; }

// ------------------------------------------------------------------ library options
static const char* const g_opt_names[MDX_OPT_COUNT] = {"gemm_tuned", "gemm_bm", "gemm_bn", "gemm_ring", "gemm_halo", "gemm_halo8",
                                                        "gemm_splitk_fixup_max", "gemm_spread", "halo_nsb", "gn_min_blocks",
                                                        "gn_fused", "gn_col_chunks", "gemm_conv8p", "gemm_conv8p_min_m", "gemm_subpixel_min_tiles", "gemm_conv8p_var", "attn8", "attn8_min_blocks",
                                                        "gn_wide_rows", "gn_fused_small", "gn_boost_mb", "attn_occ3", "attn_kv_split", "attn_fast_stage",
                                                        "gn_prefetch", "gemm_dense_issue", "gemm_ln_prefetch", "gemm_lean_dense", "attn_pipe"};
static int g_opt[MDX_OPT_COUNT] = {1, 0, 0, 0, 1, 1, 4, 1, 0, 512, 1, 4, 1, 4096, 32, 0, 0, 192, 0, 0, 40, 1, 1, 1, 1, 1, 1, 1, 1};

int mdx_opt(int id) { return g_opt[id]; }

extern "C" int mdx_set_option(const char* name, int value) {
    MDX_REQUIRE(name, "mdx_set_option: null name");
    for (int i = 0; i < MDX_OPT_COUNT; ++i)
        if (!strcmp(name, g_opt_names[i])) {
            g_opt[i] = value;
            return MDX_OK;
        }
    mdx_set_error("mdx_set_option: unknown option '%s'", name);
    return MDX_E_INVALID;
}

extern "C" int mdx_get_option(const char* name, int* value) {
    MDX_REQUIRE(name && value, "mdx_get_option: null argument");
    for (int i = 0; i < MDX_OPT_COUNT; ++i)
        if (!strcmp(name, g_opt_names[i])) {
            *value = g_opt[i];
            return MDX_OK;
        }
    mdx_set_error("mdx_get_option: unknown option '%s'", name);
    return MDX_E_INVALID;
}
extern "C" int mdx_version(void) { return 1; }

namespace {

// ------------------------------------------------------------------ layout boundary
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, f16* __restrict__ y, int B,
                                                           int C, int HW, int Cpad) {
    const size_t total = (size_t)B * HW * Cpad;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % Cpad);
        const size_t bp = i / Cpad;
        const int pix = (int)(bp % HW);
        const int b = (int)(bp / HW);
        y[i] = c < C ? (f16)x[((size_t)b * C + c) * HW + pix] : (f16)0.f;
    }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const f16* __restrict__ x, float* __restrict__ y, int B,
                                                           int C, int HW, int Cs) {
    const size_t total = (size_t)B * C * HW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int pix = (int)(i % HW);
        const size_t bc = i / HW;
        const int c = (int)(bc % C);
        const int b = (int)(bc / C);
        y[i] = (float)x[((size_t)b * HW + pix) * Cs + c];
    }
}

// ------------------------------------------------------------------ timestep embedding (util.py:111-131)
__global__ __launch_bounds__(256) void timestep_embedding_kernel(const float* __restrict__ t, float* __restrict__ out,
                                                                 int M, int dim, float neg_log_period) {
    const int half = dim / 2;
    const int total = M * half;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int m = i / half, k = i - m * half;
        const float freq = expf(neg_log_period * (float)k / (float)half);
        const float arg = t[m] * freq;
        out[(size_t)m * dim + k] = cosf(arg);
        out[(size_t)m * dim + half + k] = sinf(arg);
        if ((dim & 1) && k == 0) out[(size_t)m * dim + dim - 1] = 0.f;
    }
}

// ------------------------------------------------------------------ small-M Dense
// One wave per output column n, MB rows of x per pass.  W row is streamed once per pass
// (16 B per lane); x is tiny and L2-resident.
template <int MB>
__global__ __launch_bounds__(256) void dense_small_kernel(const float* __restrict__ x, int x_ld,
                                                          const f16* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ out, int out_ld, int M, int N, int K,
                                                          int act_in, int act_out) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const f16* wr = w + (size_t)n * K;
    for (int m0 = 0; m0 < M; m0 += MB) {
        float acc[MB];
#pragma unroll
        for (int i = 0; i < MB; ++i) acc[i] = 0.f;
        for (int k = lane * 8; k < K; k += 64 * 8) {
            const f16x8 wv = *reinterpret_cast<const f16x8*>(wr + k);
            float wf[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) wf[e] = (float)wv[e];
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                if (m0 + i < M) {
                    const float4* xp = reinterpret_cast<const float4*>(x + (size_t)(m0 + i) * x_ld + k);
                    const float4 a = xp[0], b = xp[1];
                    float xv[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float v = act_in ? silu_f(xv[e]) : xv[e];
                        acc[i] += v * wf[e];
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const float s = wave_sum(acc[i]);
            if (lane == 0 && m0 + i < M) {
                float v = s + (bias ? bias[n] : 0.f);
                if (act_out) v = silu_f(v);
                out[(size_t)(m0 + i) * out_ld + n] = v;
            }
        }
    }
}

// ------------------------------------------------------------------ fused sampler update (plms.py:188-244)
struct StepParams {
    const float* x;
    const f16* eps_u;
    const f16* eps_c;
    const float* old1;
    const float* old2;
    const float* old3;
    const float* noise;
    float* e_t_out;
    float* x_prev;
    float* pred_x0;
    int eps_ld, B, C, HW;
    float cfg_scale, c0, c1, c2, c3;
    float sqrt_at, sqrt_one_minus_at, sqrt_a_prev, dir_coef, sigma;
};

__global__ __launch_bounds__(256) void sampler_step_kernel(const StepParams p) {
    mdx_kernarg_touch<sizeof(StepParams)>();
    const size_t total = (size_t)p.B * p.C * p.HW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int pix = (int)(i % p.HW);
        const size_t bc = i / p.HW;
        const int c = (int)(bc % p.C);
        const int b = (int)(bc / p.C);
        const size_t ei = ((size_t)b * p.HW + pix) * p.eps_ld + c;
        float e_t = (float)p.eps_c[ei];
        if (p.eps_u) {
            const float eu = (float)p.eps_u[ei];
            e_t = eu + p.cfg_scale * (e_t - eu);
        }
        if (p.e_t_out) p.e_t_out[i] = e_t;
        float ep = p.c0 * e_t;
        if (p.old1) ep += p.c1 * p.old1[i];
        if (p.old2) ep += p.c2 * p.old2[i];
        if (p.old3) ep += p.c3 * p.old3[i];
        const float xv = p.x[i];
        const float px0 = (xv - p.sqrt_one_minus_at * ep) / p.sqrt_at;
        float xp = p.sqrt_a_prev * px0 + p.dir_coef * ep;
        if (p.noise) xp += p.sigma * p.noise[i];
        if (p.pred_x0) p.pred_x0[i] = px0;
        p.x_prev[i] = xp;
    }
}

// ------------------------------------------------------------------ MFMA layout probe
__global__ void probe_mfma_kernel(const f16* a, const f16* b, float* c) {
    const int lane = threadIdx.x;
    const f16x8 av = *reinterpret_cast<const f16x8*>(a + lane * 8);
    const f16x8 bv = *reinterpret_cast<const f16x8*>(b + lane * 8);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) c[lane * 16 + r] = acc[r];
}

__global__ void probe_mfma16_kernel(const f16* a, const f16* b, float* c) {
    const int lane = threadIdx.x;
    const f16x8 av = *reinterpret_cast<const f16x8*>(a + lane * 8);
    const f16x8 bv = *reinterpret_cast<const f16x8*>(b + lane * 8);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) c[lane * 4 + r] = acc[r];
}

inline int grid_for(size_t total) {
    size_t b = (total + 255) / 256;
    return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int mdx_nchw_to_nhwc_f16(const float* x, void* y, int B, int C, int H, int W, int Cpad, mdx_stream_t s) {
    MDX_REQUIRE(x && y && B > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C, "mdx_nchw_to_nhwc_f16: bad arguments");
    const size_t total = (size_t)B * H * W * Cpad;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, x, (f16*)y, B, C,
                       H * W, Cpad);
    MDX_LAUNCH_CHECK("mdx_nchw_to_nhwc_f16");
    return MDX_OK;
}

extern "C" int mdx_nhwc_to_nchw_f32(const void* x, float* y, int B, int C, int H, int W, int Cstride,
                                    mdx_stream_t s) {
    MDX_REQUIRE(x && y && B > 0 && C > 0 && H > 0 && W > 0 && Cstride >= C, "mdx_nhwc_to_nchw_f32: bad arguments");
    const size_t total = (size_t)B * C * H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, (const f16*)x, y, B,
                       C, H * W, Cstride);
    MDX_LAUNCH_CHECK("mdx_nhwc_to_nchw_f32");
    return MDX_OK;
}

extern "C" int mdx_timestep_embedding_f32(const float* t, float* out, int M, int dim, float max_period,
                                          mdx_stream_t s) {
    MDX_REQUIRE(t && out && M > 0 && dim >= 2 && max_period > 0.f, "mdx_timestep_embedding_f32: bad arguments");
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3(grid_for((size_t)M * (dim / 2))), dim3(256), 0, (hipStream_t)s,
                       t, out, M, dim, -logf(max_period));
    MDX_LAUNCH_CHECK("mdx_timestep_embedding_f32");
    return MDX_OK;
}

extern "C" int mdx_dense_small_f32(const float* x, int x_ld, const void* w, const float* b, float* out, int out_ld,
                                   int M, int N, int K, int act_in, int act_out, mdx_stream_t s) {
    MDX_REQUIRE(x && w && out, "mdx_dense_small_f32: null pointer");
    MDX_REQUIRE(M > 0 && N > 0 && K > 0 && K % 8 == 0 && x_ld % 4 == 0 && x_ld >= K && out_ld >= N,
                "mdx_dense_small_f32: bad extents (M=%d N=%d K=%d)", M, N, K);
    dim3 grid((N + 3) / 4);
    hipStream_t st = (hipStream_t)s;
    if (M <= 2)
        hipLaunchKernelGGL(dense_small_kernel<2>, grid, dim3(256), 0, st, x, x_ld, (const f16*)w, b, out, out_ld, M, N, K, act_in, act_out);
    else
        hipLaunchKernelGGL(dense_small_kernel<8>, grid, dim3(256), 0, st, x, x_ld, (const f16*)w, b, out, out_ld, M, N, K, act_in, act_out);
    MDX_LAUNCH_CHECK("mdx_dense_small_f32");
    return MDX_OK;
}

extern "C" int mdx_sampler_step_f32(const float* x, const void* eps_u, const void* eps_c, int eps_ld, float cfg_scale,
                                    const float* old1, const float* old2, const float* old3, const float* coef4,
                                    float sqrt_at, float sqrt_one_minus_at, float sqrt_a_prev, float dir_coef,
                                    float sigma, const float* noise, float* e_t_out, float* x_prev, float* pred_x0,
                                    int B, int C, int H, int W, mdx_stream_t s) {
    MDX_REQUIRE(x && eps_c && coef4 && x_prev, "mdx_sampler_step_f32: null pointer");
    MDX_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && eps_ld >= C, "mdx_sampler_step_f32: bad extents");
    MDX_REQUIRE(sigma == 0.f || noise, "mdx_sampler_step_f32: sigma != 0 needs a noise tensor");
    MDX_REQUIRE((coef4[1] == 0.f || old1) && (coef4[2] == 0.f || old2) && (coef4[3] == 0.f || old3),
                "mdx_sampler_step_f32: non-zero multistep coefficient without its eps history");
    StepParams p{};
    p.x = x;
    p.eps_u = (const f16*)eps_u;
    p.eps_c = (const f16*)eps_c;
    p.old1 = coef4[1] != 0.f ? old1 : nullptr;
    p.old2 = coef4[2] != 0.f ? old2 : nullptr;
    p.old3 = coef4[3] != 0.f ? old3 : nullptr;
    p.noise = sigma != 0.f ? noise : nullptr;
    p.e_t_out = e_t_out;
    p.x_prev = x_prev;
    p.pred_x0 = pred_x0;
    p.eps_ld = eps_ld;
    p.B = B;
    p.C = C;
    p.HW = H * W;
    p.cfg_scale = cfg_scale;
    p.c0 = coef4[0];
    p.c1 = coef4[1];
    p.c2 = coef4[2];
    p.c3 = coef4[3];
    p.sqrt_at = sqrt_at;
    p.sqrt_one_minus_at = sqrt_one_minus_at;
    p.sqrt_a_prev = sqrt_a_prev;
    p.dir_coef = dir_coef;
    p.sigma = sigma;
    hipLaunchKernelGGL(sampler_step_kernel, dim3(grid_for((size_t)B * C * H * W)), dim3(256), 0, (hipStream_t)s, p);
    MDX_LAUNCH_CHECK("mdx_sampler_step_f32");
    return MDX_OK;
}

extern "C" int mdx_probe_mfma_32x32x16_f16(const void* a, const void* b, float* c, mdx_stream_t s) {
    MDX_REQUIRE(a && b && c, "mdx_probe_mfma_32x32x16_f16: null pointer");
    hipLaunchKernelGGL(probe_mfma_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, (const f16*)a, (const f16*)b, c);
    MDX_LAUNCH_CHECK("mdx_probe_mfma_32x32x16_f16");
    return MDX_OK;
}

extern "C" int mdx_probe_mfma_16x16x32_f16(const void* a, const void* b, float* c, mdx_stream_t s) {
    MDX_REQUIRE(a && b && c, "mdx_probe_mfma_16x16x32_f16: null pointer");
    hipLaunchKernelGGL(probe_mfma16_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, (const f16*)a, (const f16*)b, c);
    MDX_LAUNCH_CHECK("mdx_probe_mfma_16x16x32_f16");
    return MDX_OK;
}

// ------------------------------------------------------------------ DMA / load streaming probe (tools/dma_probe.py)
// Measures what ONE workgroup per CU can pull through `buffer_load ... lds` (mode 0) or through plain
// global_load_dwordx4 into registers (mode 1): tiles of 1 KiB per wave-instruction, `per` instructions per wave per
// tile, NS tiles in flight.  Not on the hot path.
namespace {
template <int NS>
__global__ __launch_bounds__(512) void dma_probe_kernel(const char* src, size_t bytes_per_block, int per, int mode,
                                                        int stride_tiles, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const size_t tile_bytes = (size_t)nw * per * 1024;
    const int nt = (int)(bytes_per_block / tile_bytes);
    const bool shared = (mode & 4) != 0;        // every block streams the SAME region (L2-resident operands shared by the tiles of a launch)
    mode &= 3;
    const char* base = src + (shared ? 0 : (size_t)blockIdx.x * bytes_per_block);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(base, (unsigned)bytes_per_block);
    float acc = 0.f;
    if (mode == 2) {
        // BOTH paths at once (round 4): per tile every wave issues `per` LDS-DMA instructions AND `per` plain 16-byte loads into
        // registers (of the next region: 2 x tile_bytes per step), to see whether the two operand paths add up on a CU
        typedef unsigned pu4 __attribute__((ext_vector_type(4)));
        const int nt2 = nt / 2;
        auto issue = [&](int t, int stage) {
            for (int j = 0; j < per; ++j) {
                const unsigned off = (unsigned)((size_t)(2 * t) * tile_bytes + ((wave * per + j) * 64 + lane) * 16);
                dma16(rs, smem + (size_t)stage * tile_bytes + (wave * per + j) * 1024, off);
            }
        };
        pu4 v[8];
        unsigned x = 0;
        for (int i = 0; i < NS - 1 && i < nt2; ++i) issue(i, i);
        int wr = NS - 1;
        for (int t = 0; t < nt2; ++t) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < per) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(((wave * per + j) * 64 + lane) * 16), (unsigned)((size_t)(2 * t + 1) * tile_bytes), 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (t + NS - 1 < nt2) issue(t + NS - 1, wr);
            wr = (wr + 1 == NS) ? 0 : wr + 1;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < per) x ^= v[j][0] ^ v[j][3];
        }
        acc = ((float*)smem)[threadIdx.x] + (float)(x & 1);
    } else if (mode == 0) {
        auto issue = [&](int t, int stage) {
            for (int j = 0; j < per; ++j) {
                const unsigned off = (unsigned)((size_t)(t * stride_tiles % nt) * tile_bytes + ((wave * per + j) * 64 + lane) * 16);
                dma16(rs, smem + (size_t)stage * tile_bytes + (wave * per + j) * 1024, off);
            }
        };
        for (int i = 0; i < NS - 1 && i < nt; ++i) issue(i, i);
        int wr = NS - 1;
        for (int t = 0; t < nt; ++t) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // conservative: drain own DMAs (depth comes from other waves/tiles)
            __builtin_amdgcn_s_barrier();
            if (t + NS - 1 < nt) issue(t + NS - 1, wr);
            wr = (wr + 1 == NS) ? 0 : wr + 1;
        }
        acc = ((float*)smem)[threadIdx.x];
    } else {
        const f32x4* g = reinterpret_cast<const f32x4*>(base);
        f32x4 v[8];
        for (int t = 0; t < nt; ++t) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < per) v[j] = g[((size_t)(t * stride_tiles % nt) * tile_bytes + ((wave * per + j) * 64 + lane) * 16) / 16];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < per) acc += v[j][0] + v[j][3];
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}
}  // namespace

extern "C" int mdx_probe_dma_stream(const void* src, size_t bytes_per_block, int nblocks, int waves, int per, int ns,
                                    int mode, int stride_tiles, float* sink, mdx_stream_t s) {
    MDX_REQUIRE(src && sink && nblocks > 0 && waves >= 1 && waves <= 8 && per >= 1 && per <= 8 && ns >= 2 && ns <= 4,
                "mdx_probe_dma_stream: bad arguments");
    const size_t lds = (size_t)ns * waves * per * 1024;
    MDX_REQUIRE(lds <= 160 * 1024, "mdx_probe_dma_stream: ring too large");
    hipStream_t st = (hipStream_t)s;
#define MDX_PROBE_LAUNCH(NSV)                                                                                        \
    {                                                                                                                \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_probe_kernel<NSV>),                             \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                           \
        hipLaunchKernelGGL(dma_probe_kernel<NSV>, dim3(nblocks), dim3(waves * 64), lds, st, (const char*)src,        \
                           bytes_per_block, per, mode, stride_tiles, sink);                                          \
    }
    if (ns == 2) MDX_PROBE_LAUNCH(2)
    else if (ns == 3) MDX_PROBE_LAUNCH(3)
    else MDX_PROBE_LAUNCH(4)
#undef MDX_PROBE_LAUNCH
    MDX_LAUNCH_CHECK("mdx_probe_dma_stream");
    return MDX_OK;
}

// ------------------------------------------------------------------ L2 -> VGPR streaming probe (tools/l2_probe.py)
// What the fused-chain kernels (stchain.hip) do with their weights: every wave streams its own contiguous region with one
// coalesced 1 KiB buffer_load_dwordx4 per piece, PF pieces in flight (schedule pinned), optionally the SAME region in every
// block (shared = 1: all CUs hit the same L2 lines, as all row blocks of a fused launch read the same weights).
namespace {
typedef unsigned pu32x4 __attribute__((ext_vector_type(4)));
template <int PFD>
__global__ __launch_bounds__(1024) void l2_probe_kernel(const char* src, unsigned bytes_per_wave, int shared, unsigned total_bytes,
                                                        float* sink) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(src, total_bytes);
    const unsigned base = ((shared ? 0u : (unsigned)blockIdx.x * nw) + wave) * bytes_per_wave;
    const unsigned voff = lane * 16u;
    const int pieces = bytes_per_wave / 1024;
    pu32x4 ring[PFD];
#pragma unroll
    for (int j = 0; j < PFD; ++j) ring[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, base + j * 1024u, 0);
    unsigned acc = 0;
    for (int p0 = 0; p0 < pieces; p0 += PFD) {
#pragma unroll
        for (int j = 0; j < PFD; ++j) {
            acc ^= ring[j][0] ^ ring[j][3];
            ring[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, base + (unsigned)(p0 + j + PFD) * 1024u, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (acc == 0x12345678u) sink[0] = 1.f;
}
}  // namespace

extern "C" int mdx_probe_l2_stream(const void* src, size_t total_bytes, unsigned bytes_per_wave, int nblocks, int waves, int pf,
                                   int shared, float* sink, mdx_stream_t s) {
    MDX_REQUIRE(src && sink && nblocks > 0 && waves >= 1 && waves <= 16 && bytes_per_wave % (32 * 1024) == 0 &&
                    total_bytes <= 0xffffffffull, "mdx_probe_l2_stream: bad arguments");
    hipStream_t st = (hipStream_t)s;
    if (pf == 8) hipLaunchKernelGGL(l2_probe_kernel<8>, dim3(nblocks), dim3(waves * 64), 0, st, (const char*)src, bytes_per_wave, shared, (unsigned)total_bytes, sink);
    else if (pf == 16) hipLaunchKernelGGL(l2_probe_kernel<16>, dim3(nblocks), dim3(waves * 64), 0, st, (const char*)src, bytes_per_wave, shared, (unsigned)total_bytes, sink);
    else if (pf == 32) hipLaunchKernelGGL(l2_probe_kernel<32>, dim3(nblocks), dim3(waves * 64), 0, st, (const char*)src, bytes_per_wave, shared, (unsigned)total_bytes, sink);
    else MDX_REQUIRE(false, "mdx_probe_l2_stream: pf must be 8, 16 or 32");
    MDX_LAUNCH_CHECK("mdx_probe_l2_stream");
    return MDX_OK;
}

// ------------------------------------------------------------------ VALU issue-rate probe (tools/exp/r04n_valu_probe.py)
namespace {
// kind 0: v_fma_f32, 1: v_exp_f32, 2: v_pk_fma_f32 (two results), 3: v_cvt_pk_f16_f32, 4: v_max3_f32, 5: v_exp_f16, 6: v_pk_fma_f16 (two
// results), 7: v_pk_max_f16 (two results) -- eight independent chains per lane
template <int KIND>
__global__ __launch_bounds__(256) void valu_probe_kernel(int iters, float* sink) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    float x[8];
    f32x2 y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        x[i] = 0.001f * (float)(threadIdx.x + i) - 0.5f;
        y[i] = f32x2{x[i], -x[i]};
    }
    const f32x2 c2 = {0.999f, 1.001f}, d2 = {1e-6f, -1e-6f};
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    h2v hx[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) hx[i] = h2v{(_Float16)x[i], (_Float16)(-x[i])};
    const h2v hc = {(_Float16)0.999f, (_Float16)1.001f}, hd = {(_Float16)1e-3f, (_Float16)-1e-3f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) x[i] = __builtin_fmaf(x[i], 0.999f, 1e-6f);
            else if (KIND == 1) x[i] = __builtin_amdgcn_exp2f(x[i]);
            else if (KIND == 2) y[i] = __builtin_elementwise_fma(y[i], c2, d2);
            else if (KIND == 3) {
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                const h2 h = {(_Float16)x[i], (_Float16)x[(i + 1) & 7]};
                x[i] = (float)h[0] + (float)h[1];
            } else if (KIND == 4) x[i] = __builtin_fmaxf(__builtin_fmaxf(x[i], x[(i + 1) & 7]), x[(i + 2) & 7]) * 0.5f;
            else if (KIND == 5) {      // v_exp_f16: is a half-precision exponential cheaper than v_exp_f32?  (round 5, attention softmax)
                _Float16 r;
                asm volatile("v_exp_f16 %0, %1" : "=v"(r) : "v"(hx[i][0]));
                hx[i][0] = r;
            } else if (KIND == 6) hx[i] = __builtin_elementwise_fma(hx[i], hc, hd);      // v_pk_fma_f16 (two results)
            else {                     // 7: v_pk_max_f16 (two results) -- the row-maximum fold on packed halves
                hx[i] = __builtin_elementwise_max(hx[i], hx[(i + 1) & 7]);
            }
        }
    }
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) a += x[i] + y[i].x + y[i].y + (float)hx[i][0] + (float)hx[i][1];
    if (a == 123.456f) sink[0] = a;
}
// Does VALU work run in the shadow of an MFMA?  (round 6, attention softmax.)  Four slots per iteration; a slot is
//   kind 0: one v_mfma_f32_32x32x16_f16 (two independent accumulator chains alternate)        kind 3: two v_exp_f32
//   kind 1: the MFMA + two v_exp_f32 (independent chains)                                       kind 4: eight v_fma_f32
//   kind 2: the MFMA + eight v_fma_f32                                                          kind 6: the softmax mix alone
//   kind 5: the MFMA + the softmax mix of one score pair (2 fma, 2 exp, 1 packed add, 1 cvt_pk)
template <int KIND>
__global__ __launch_bounds__(256) void mix_probe_kernel(int iters, float* sink) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x16 acc[2];
    f16x8 a, b;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        a[e] = (f16)(0.001f * (float)((threadIdx.x + e) & 15));
        b[e] = (f16)(0.002f * (float)((threadIdx.x + 3 * e) & 7));
    }
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 0.001f * (float)(threadIdx.x + i) - 0.5f;
    f32x2 ps = {0.f, 0.f};
    unsigned pw = 0;
    constexpr bool MF = KIND == 0 || KIND == 1 || KIND == 2 || KIND == 5;
    constexpr bool MFA = KIND >= 7;       // 7 / 8 / 9: the accumulators (9: A and B too) live in AccVGPRs (inline asm, "a" constraint)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            if constexpr (MF) acc[sl & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[sl & 1], 0, 0, 0);
            if constexpr (MFA) {
                if constexpr (KIND == 9)
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[sl & 1]) : "a"(a), "a"(b));
                else
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[sl & 1]) : "v"(a), "v"(b));
            }
            if constexpr (KIND == 1 || KIND == 3) {
                x[2 * sl] = __builtin_amdgcn_exp2f(x[2 * sl]);
                x[2 * sl + 1] = __builtin_amdgcn_exp2f(x[2 * sl + 1]);
            } else if constexpr (KIND == 2 || KIND == 4 || KIND == 7) {
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], 0.999f, 1e-6f);
            } else if constexpr (KIND == 5 || KIND == 6 || KIND == 8 || KIND == 9) {
                const float u = __builtin_fmaf(x[2 * sl], 0.999f, -0.25f), v = __builtin_fmaf(x[2 * sl + 1], 0.999f, -0.25f);
                const f32x2 e2 = {__builtin_amdgcn_exp2f(u), __builtin_amdgcn_exp2f(v)};
                ps += e2;
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                const h2 hv = {(_Float16)e2.x, (_Float16)e2.y};
                pw ^= __builtin_bit_cast(unsigned, hv);
                x[2 * sl] = e2.x - 1.0f;
                x[2 * sl + 1] = e2.y - 1.0f;
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 64, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float t = ps.x + ps.y + (float)pw;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += x[i];
#pragma unroll
    for (int r = 0; r < 16; ++r) t += acc[0][r] + acc[1][r];
    if (t == 123.456f) sink[0] = t;
}
}  // namespace

extern "C" int mdx_probe_mix_rate(int kind, int iters, int nblocks, float* sink, mdx_stream_t s) {
    MDX_REQUIRE(sink && iters > 0 && nblocks > 0 && kind >= 0 && kind <= 9, "mdx_probe_mix_rate: bad arguments");
    hipStream_t st = (hipStream_t)s;
    switch (kind) {
        case 0: hipLaunchKernelGGL(mix_probe_kernel<0>, dim3(nblocks), dim3(256), 0, st, iters, sink); break;
        case 1: hipLaunchKernelGGL(mix_probe_kernel<1>, dim3(nblocks), dim3(256), 0, st, iters, sink); break;
        case 2: hipLaunchKernelGGL(mix_probe_kernel<2>, dim3(nblocks), dim3(256), 0, st, iters, sink); break;
        case 3: hipLaunchKernelGGL(mix_probe_kernel<3>, dim3(nblocks), dim3(256), 0, st, iters, sink); break;
        case 4: hipLaunchKernelGGL(mix_probe_kernel<4>, dim3(nblocks), dim3(256), 0, st, iters, sink); break;
        case 5: hipLaunchKernelGGL(mix_probe_kernel<5>, dim3(nblocks), dim3(256), 0, st, iters, sink); break;
        case 7: hipLaunchKernelGGL(mix_probe_kernel<7>, dim3(nblocks), dim3(256), 0, st, iters, sink); break;
        case 8: hipLaunchKernelGGL(mix_probe_kernel<8>, dim3(nblocks), dim3(256), 0, st, iters, sink); break;
        case 9: hipLaunchKernelGGL(mix_probe_kernel<9>, dim3(nblocks), dim3(256), 0, st, iters, sink); break;
        default: hipLaunchKernelGGL(mix_probe_kernel<6>, dim3(nblocks), dim3(256), 0, st, iters, sink); break;
    }
    MDX_LAUNCH_CHECK("mdx_probe_mix_rate");
    return MDX_OK;
}

extern "C" int mdx_probe_valu_rate(int kind, int iters, int nblocks, float* sink, mdx_stream_t s) {
    MDX_REQUIRE(sink && iters > 0 && nblocks > 0 && kind >= 0 && kind <= 7, "mdx_probe_valu_rate: bad arguments");
    hipStream_t st = (hipStream_t)s;
    switch (kind) {
        case 0: hipLaunchKernelGGL(valu_probe_kernel<0>, dim3(nblocks), dim3(256), 0, st, iters, sink); break;
        case 1: hipLaunchKernelGGL(valu_probe_kernel<1>, dim3(nblocks), dim3(256), 0, st, iters, sink); break;
        case 2: hipLaunchKernelGGL(valu_probe_kernel<2>, dim3(nblocks), dim3(256), 0, st, iters, sink); break;
        case 3: hipLaunchKernelGGL(valu_probe_kernel<3>, dim3(nblocks), dim3(256), 0, st, iters, sink); break;
        case 5: hipLaunchKernelGGL(valu_probe_kernel<5>, dim3(nblocks), dim3(256), 0, st, iters, sink); break;
        case 6: hipLaunchKernelGGL(valu_probe_kernel<6>, dim3(nblocks), dim3(256), 0, st, iters, sink); break;
        case 7: hipLaunchKernelGGL(valu_probe_kernel<7>, dim3(nblocks), dim3(256), 0, st, iters, sink); break;
        default: hipLaunchKernelGGL(valu_probe_kernel<4>, dim3(nblocks), dim3(256), 0, st, iters, sink); break;
    }
    MDX_LAUNCH_CHECK("mdx_probe_valu_rate");
    return MDX_OK;
}
