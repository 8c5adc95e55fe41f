// Taichu-GLIDE specific small kernels: resblock up/down skip paths, text embedding,
// super-res conditioning input, fused learned-variance / DDIM sampler update.  All HBM/latency-bound elementwise work.
#include "mdx_common.h"

namespace {

inline int grid_for(size_t total) {
    size_t b = (total + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

// nn.AvgPool2d(2,2) on NHWC fp16 (unet.py:74): one thread per 8-channel chunk of an output pixel
__global__ __launch_bounds__(256) void avgpool_kernel(const f16* __restrict__ x, f16* __restrict__ y, int B, int H,
                                                      int W, int C) {
    const int Ho = H / 2, Wo = W / 2, CC = C / 8;
    const size_t total = (size_t)B * Ho * Wo * CC;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int cc = (int)(i % CC);
        size_t r = i / CC;
        const int xo = (int)(r % Wo);
        r /= Wo;
        const int yo = (int)(r % Ho);
        const int b = (int)(r / Ho);
        const f16* base = x + (((size_t)b * H + 2 * yo) * W + 2 * xo) * C + cc * 8;
        const f16x8 a = *reinterpret_cast<const f16x8*>(base);
        const f16x8 bq = *reinterpret_cast<const f16x8*>(base + C);
        const f16x8 c = *reinterpret_cast<const f16x8*>(base + (size_t)W * C);
        const f16x8 d = *reinterpret_cast<const f16x8*>(base + (size_t)W * C + C);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)(0.25f * ((float)a[e] + (float)bq[e] + (float)c[e] + (float)d[e]));
        *reinterpret_cast<f16x8*>(y + i * 8) = o;
    }
}

// ops.ResizeNearestNeighbor x2 (unet.py:46-49) on NHWC fp16
__global__ __launch_bounds__(256) void upsample_kernel(const f16* __restrict__ x, f16* __restrict__ y, int B, int H,
                                                       int W, int C) {
    const int Ho = 2 * H, Wo = 2 * W, CC = C / 8;
    const size_t total = (size_t)B * Ho * Wo * CC;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int cc = (int)(i % CC);
        size_t r = i / CC;
        const int xo = (int)(r % Wo);
        r /= Wo;
        const int yo = (int)(r % Ho);
        const int b = (int)(r / Ho);
        *reinterpret_cast<f16x8*>(y + i * 8) =
            *reinterpret_cast<const f16x8*>(x + (((size_t)b * H + (yo >> 1)) * W + (xo >> 1)) * C + cc * 8);
    }
}

__global__ __launch_bounds__(256) void text_embed_kernel(const int* __restrict__ tokens, const int* __restrict__ mask,
                                                         const f16* __restrict__ tok_emb, const f16* __restrict__ pos,
                                                         const f16* __restrict__ pad, f16* __restrict__ out, int B,
                                                         int T, int width, int n_vocab) {
    const int CC = width / 8;
    const size_t total = (size_t)B * T * CC;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int cc = (int)(i % CC);
        const size_t bt = i / CC;
        const int t = (int)(bt % T);
        f16x8 o;
        if (mask[bt]) {
            int tk = tokens[bt];
            tk = tk < 0 ? 0 : (tk >= n_vocab ? n_vocab - 1 : tk);
            const f16x8 e = *reinterpret_cast<const f16x8*>(tok_emb + (size_t)tk * width + cc * 8);
            const f16x8 p = *reinterpret_cast<const f16x8*>(pos + (size_t)t * width + cc * 8);
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = (f16)((float)e[k] + (float)p[k]);
        } else {
            o = *reinterpret_cast<const f16x8*>(pad + (size_t)t * width + cc * 8);
        }
        *reinterpret_cast<f16x8*>(out + i * 8) = o;
    }
}

// [x | legacy-bilinear(quantised low-res)] -> NHWC fp16, 8-channel stride
__global__ __launch_bounds__(256) void superres_input_kernel(const float* __restrict__ x, const float* __restrict__ low,
                                                             f16* __restrict__ out, int B, int S, int sl) {
    const size_t total = (size_t)B * S * S;
    const float ratio = (float)sl / (float)S;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int xo = (int)(i % S);
        const int yo = (int)((i / S) % S);
        const int b = (int)(i / ((size_t)S * S));
        const float sy = yo * ratio, sx = xo * ratio;   // MindSpore 1.8 ResizeBilinear, align_corners=False (legacy map)
        int y0 = (int)floorf(sy), x0 = (int)floorf(sx);
        y0 = y0 > sl - 1 ? sl - 1 : y0;
        x0 = x0 > sl - 1 ? sl - 1 : x0;
        const int y1 = y0 + 1 > sl - 1 ? sl - 1 : y0 + 1, x1 = x0 + 1 > sl - 1 ? sl - 1 : x0 + 1;
        const float fy = sy - (float)y0, fx = sx - (float)x0;
        f16x8 o;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            o[c] = (f16)x[(((size_t)b * 3 + c) * S + yo) * S + xo];
            const float* lp = low + ((size_t)b * 3 + c) * sl * sl;
            auto q = [&](int yy, int xx) { return rintf((lp[yy * sl + xx] + 1.0f) * 127.5f) / 127.5f - 1.0f; };
            const float top = q(y0, x0) * (1.f - fx) + q(y0, x1) * fx;
            const float bot = q(y1, x0) * (1.f - fx) + q(y1, x1) * fx;
            o[3 + c] = (f16)(top * (1.f - fy) + bot * fy);
        }
        o[6] = (f16)0.f;
        o[7] = (f16)0.f;
        *reinterpret_cast<f16x8*>(out + i * 8) = o;
    }
}

struct GlideStep {
    const float* x;
    const f16* out_c;
    const f16* out_u;
    const float* noise;
    float* x_next;
    float* pred_x0;
    int ld, B, HW, mode;
    float scale, log_beta, post_logvar, sqrt_recip, sqrt_recipm1, coef1, coef2, sqrt_ab_prev, sqrt_1m_ab_prev, noise_scale;
};

__global__ __launch_bounds__(256) void glide_step_kernel(const GlideStep p) {
    mdx_kernarg_touch<sizeof(GlideStep)>();
    const size_t total = (size_t)p.B * 3 * p.HW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int pix = (int)(i % p.HW);
        const int c = (int)((i / p.HW) % 3);
        const int b = (int)(i / ((size_t)3 * p.HW));
        const size_t oi = ((size_t)b * p.HW + pix) * p.ld;
        float eps = (float)p.out_c[oi + c];
        if (p.out_u) {
            const float eu = (float)p.out_u[oi + c];
            eps = eu + p.scale * (eps - eu);
        }
        const float v = (float)p.out_c[oi + 3 + c];
        const float xv = p.x[i];
        float x0 = p.sqrt_recip * xv - p.sqrt_recipm1 * eps;
        x0 = fminf(1.0f, fmaxf(-1.0f, x0));
        float xn;
        if (p.mode == 0) {
            const float frac = (v + 1.0f) * 0.5f;
            const float logvar = frac * p.log_beta + (1.0f - frac) * p.post_logvar;
            xn = p.coef1 * x0 + p.coef2 * xv;
            if (p.noise) xn += p.noise_scale * expf(0.5f * logvar) * p.noise[i];
        } else {
            const float e2 = (p.sqrt_recip * xv - x0) / p.sqrt_recipm1;
            xn = p.sqrt_ab_prev * x0 + p.sqrt_1m_ab_prev * e2;
        }
        if (p.pred_x0) p.pred_x0[i] = x0;
        p.x_next[i] = xn;
    }
}

// Text-key tables -> the text slots of every AttentionBlock's key / value buffers (mdx_glide_kv_select_f16): grid (z, slot, b).
// One 2-D copy per (slot, batch row): `rows` rows of `row_bytes` (multiples of 16) from table entry e = entry0 + b * entry_per_b.
__global__ __launch_bounds__(256) void kv_select_kernel(const mdx_glide_kv_slot* __restrict__ slots, long entry0, int entry_per_b,
                                                        int b0) {
    const mdx_glide_kv_slot sl = slots[blockIdx.y];
    const int b = b0 + (int)blockIdx.z;
    const long e = entry0 + (long)blockIdx.z * entry_per_b;
    const char* src = reinterpret_cast<const char*>(sl.src) + e * sl.src_entry_bytes;
    char* dst = reinterpret_cast<char*>(sl.dst) + (long)b * sl.dst_batch_bytes;
    const int vpr = sl.row_bytes >> 4;                 // 16-byte vectors per row
    const long total = (long)sl.rows * vpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / vpr;
        const int v = (int)(i - r * vpr);
        *reinterpret_cast<uint4*>(dst + r * sl.dst_pitch + v * 16) = *reinterpret_cast<const uint4*>(src + r * sl.src_pitch + v * 16);
    }
}

}  // namespace

extern "C" int mdx_glide_kv_select_f16(const mdx_glide_kv_slot* slots_dev, int nslots, long entry0, int entry_per_b, int b0,
                                       int nb, int blocks_per_copy, mdx_stream_t s) {
    MDX_REQUIRE(slots_dev && nslots > 0 && nslots <= 65535 && nb > 0 && nb <= 65535 && b0 >= 0 && entry0 >= 0 &&
                    (entry_per_b == 0 || entry_per_b == 1) && blocks_per_copy > 0 && blocks_per_copy <= 1024,
                "mdx_glide_kv_select_f16: bad arguments");
    hipLaunchKernelGGL(kv_select_kernel, dim3(blocks_per_copy, nslots, nb), dim3(256), 0, (hipStream_t)s, slots_dev, entry0,
                       entry_per_b, b0);
    MDX_LAUNCH_CHECK("mdx_glide_kv_select_f16");
    return MDX_OK;
}

extern "C" int mdx_avgpool2x2_f16(const void* x, void* y, int B, int H, int W, int C, mdx_stream_t s) {
    MDX_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 8 == 0,
                "mdx_avgpool2x2_f16: bad arguments");
    const size_t total = (size_t)B * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(avgpool_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, (const f16*)x, (f16*)y, B, H, W, C);
    MDX_LAUNCH_CHECK("mdx_avgpool2x2_f16");
    return MDX_OK;
}

extern "C" int mdx_upsample_nearest2x_f16(const void* x, void* y, int B, int H, int W, int C, mdx_stream_t s) {
    MDX_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "mdx_upsample_nearest2x_f16: bad arguments");
    const size_t total = (size_t)B * 4 * H * W * (C / 8);
    hipLaunchKernelGGL(upsample_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, (const f16*)x, (f16*)y, B, H, W, C);
    MDX_LAUNCH_CHECK("mdx_upsample_nearest2x_f16");
    return MDX_OK;
}

extern "C" int mdx_glide_text_embed_f16(const int* tokens, const int* mask, const void* tok_emb, const void* pos,
                                        const void* pad, void* out, int B, int T, int width, int n_vocab,
                                        mdx_stream_t s) {
    MDX_REQUIRE(tokens && mask && tok_emb && pos && pad && out && B > 0 && T > 0 && width % 8 == 0 && n_vocab > 0,
                "mdx_glide_text_embed_f16: bad arguments");
    const size_t total = (size_t)B * T * (width / 8);
    hipLaunchKernelGGL(text_embed_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, tokens, mask,
                       (const f16*)tok_emb, (const f16*)pos, (const f16*)pad, (f16*)out, B, T, width, n_vocab);
    MDX_LAUNCH_CHECK("mdx_glide_text_embed_f16");
    return MDX_OK;
}

extern "C" int mdx_glide_superres_input_f16(const float* x, const float* low, void* out, int B, int S, int s_low,
                                            mdx_stream_t s) {
    MDX_REQUIRE(x && low && out && B > 0 && S > 0 && s_low > 0, "mdx_glide_superres_input_f16: bad arguments");
    hipLaunchKernelGGL(superres_input_kernel, dim3(grid_for((size_t)B * S * S)), dim3(256), 0, (hipStream_t)s, x, low,
                       (f16*)out, B, S, s_low);
    MDX_LAUNCH_CHECK("mdx_glide_superres_input_f16");
    return MDX_OK;
}

extern "C" int mdx_glide_step_f32(const float* x, const void* out_c, const void* out_u, int ld, float guidance_scale,
                                  const float* coef8, int mode, float noise_scale, const float* noise, float* x_next,
                                  float* pred_x0, int B, int H, int W, mdx_stream_t s) {
    MDX_REQUIRE(x && out_c && coef8 && x_next && B > 0 && H > 0 && W > 0 && ld >= 6 && (mode == 0 || mode == 1),
                "mdx_glide_step_f32: bad arguments");
    MDX_REQUIRE(noise_scale == 0.f || noise, "mdx_glide_step_f32: noise_scale != 0 needs a noise tensor");
    GlideStep p{};
    p.x = x;
    p.out_c = (const f16*)out_c;
    p.out_u = (const f16*)out_u;
    p.noise = noise_scale != 0.f ? noise : nullptr;
    p.x_next = x_next;
    p.pred_x0 = pred_x0;
    p.ld = ld;
    p.B = B;
    p.HW = H * W;
    p.mode = mode;
    p.scale = guidance_scale;
    p.log_beta = coef8[0];
    p.post_logvar = coef8[1];
    p.sqrt_recip = coef8[2];
    p.sqrt_recipm1 = coef8[3];
    p.coef1 = coef8[4];
    p.coef2 = coef8[5];
    p.sqrt_ab_prev = coef8[6];
    p.sqrt_1m_ab_prev = coef8[7];
    p.noise_scale = noise_scale;
    hipLaunchKernelGGL(glide_step_kernel, dim3(grid_for((size_t)B * 3 * H * W)), dim3(256), 0, (hipStream_t)s, p);
    MDX_LAUNCH_CHECK("mdx_glide_step_f32");
    return MDX_OK;
}
