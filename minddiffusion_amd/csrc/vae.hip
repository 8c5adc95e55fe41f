// Helpers for the VAE decoder's single-head, 512-wide attention (AttnBlock, ldm/modules/diffusionmodules/model.py:151-206):
// the reference materialises the [hw, hw] score matrix with two batched matmuls; with one head of d = 512 that is the
// right shape for the implicit-GEMM kernel as well (S = Q K^T and O = P V are plain 4096 x 4096 x 512 GEMMs), so the
// K and V^T ACTIVATIONS are re-laid-out into the GEMM's packed B-operand format on the device and the softmax runs in
// place on the fp16 scores.
#include "mdx_common.h"

namespace {

// One thread per 16-B chunk of the packed destination [rows/64][K/64][64][8 chunks][8]; position q of row r holds
// logical chunk q ^ ((r >> 1) & 7) (the same image ops.pack_gemm_weight builds for weights at load time).
__global__ __launch_bounds__(256) void pack_b_operand_kernel(const f16* __restrict__ src, long src_ld, int rows, int K,
                                                             f16* __restrict__ dst, int kt64, size_t nchunks) {
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < nchunks; idx += (size_t)gridDim.x * 256) {
        const int pos = (int)(idx & 7);
        const int row = (int)((idx >> 3) & 63);
        const size_t tile = idx >> 9;
        const int ktile = (int)(tile % kt64);
        const int panel = (int)(tile / kt64);
        const int chunk = pos ^ ((row >> 1) & 7);
        const int r = panel * 64 + row;
        const int k = ktile * 64 + chunk * 8;
        f16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (f16)0.f;
        if (r < rows && k < K) v = *reinterpret_cast<const f16x8*>(src + (size_t)r * src_ld + k);   // K % 8 == 0
        *reinterpret_cast<f16x8*>(dst + idx * 8) = v;
    }
}

// In-place row softmax of fp16 scores: y = softmax(scale * x) along the row; one block per row, the row held in
// registers (cols <= 256 * 8 * MAXCH), fp32 max / sum (P.Softmax(axis=2) on the bmm output, model.py:192-194).
template <int MAXCH>
__global__ __launch_bounds__(256) void softmax_rows_kernel(f16* __restrict__ x, long ld, int cols, float scale_log2) {
    __shared__ float red[8];
    f16* row = x + (size_t)blockIdx.x * ld;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float v[MAXCH][8];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
        const int col = (c * 256 + tid) * 8;
        if (col < cols) {
            const f16x8 t = *reinterpret_cast<const f16x8*>(row + col);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[c][e] = (float)t[e];
                mx = fmaxf(mx, v[c][e]);
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float mb = mx * scale_log2;
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
        const int col = (c * 256 + tid) * 8;
        if (col < cols) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[c][e] = __builtin_amdgcn_exp2f(__builtin_fmaf(v[c][e], scale_log2, -mb));
                sum += v[c][e];
            }
        }
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
        const int col = (c * 256 + tid) * 8;
        if (col < cols) {
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (f16)(v[c][e] * inv);
            *reinterpret_cast<f16x8*>(row + col) = o;
        }
    }
}

// DiagonalGaussianDistribution sample of AutoencoderKL.encode (autoencoder.py:70-78): moments NHWC fp16 [B][HW][ld]
// = [mean (zc) | logvar (zc) | pad]; z = mean + exp(0.5 * clip(logvar, -30, 20)) * noise, NCHW fp32 out (noise NULL: mean).
__global__ __launch_bounds__(256) void gaussian_sample_kernel(const f16* __restrict__ mom, const float* __restrict__ noise,
                                                              float* __restrict__ out, int B, int zc, int HW, int ld) {
    const size_t total = (size_t)B * zc * HW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int pix = (int)(i % HW);
        const size_t bc = i / HW;
        const int c = (int)(bc % zc), b = (int)(bc / zc);
        const f16* m = mom + ((size_t)b * HW + pix) * ld;
        const float mean = (float)m[c];
        float lv = (float)m[zc + c];
        lv = fminf(fmaxf(lv, -30.0f), 20.0f);
        out[i] = noise ? mean + __expf(0.5f * lv) * noise[i] : mean;
    }
}

}  // namespace

extern "C" int mdx_vae_gaussian_sample_f32(const void* moments, int ld, const float* noise, float* out, int B, int zc,
                                           int HW, mdx_stream_t s) {
    MDX_REQUIRE(moments && out && B > 0 && zc > 0 && HW > 0 && ld >= 2 * zc, "mdx_vae_gaussian_sample_f32: bad arguments");
    const size_t total = (size_t)B * zc * HW;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gaussian_sample_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const f16*)moments, noise, out, B,
                       zc, HW, ld);
    MDX_LAUNCH_CHECK("mdx_vae_gaussian_sample_f32");
    return MDX_OK;
}

extern "C" int mdx_pack_b_operand_f16(const void* src, long src_ld, int rows, int K, void* dst, mdx_stream_t s) {
    MDX_REQUIRE(src && dst, "mdx_pack_b_operand_f16: null pointer");
    MDX_REQUIRE(rows > 0 && K > 0 && K % 8 == 0 && src_ld % 8 == 0 && src_ld >= K,
                "mdx_pack_b_operand_f16: K and src_ld must be multiples of 8 (rows=%d K=%d ld=%ld)", rows, K, src_ld);
    const int kt64 = (K + 63) / 64;
    const size_t nchunks = (size_t)((rows + 63) / 64) * kt64 * 512;
    int blocks = (int)((nchunks + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_b_operand_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const f16*)src, src_ld, rows, K,
                       (f16*)dst, kt64, nchunks);
    MDX_LAUNCH_CHECK("mdx_pack_b_operand_f16");
    return MDX_OK;
}

extern "C" int mdx_softmax_rows_f16(void* x, long ld, int rows, int cols, float scale, mdx_stream_t s) {
    MDX_REQUIRE(x, "mdx_softmax_rows_f16: null pointer");
    MDX_REQUIRE(rows > 0 && cols > 0 && cols % 8 == 0 && ld % 8 == 0 && ld >= cols && cols <= 16384,
                "mdx_softmax_rows_f16: cols must be a multiple of 8 and <= 16384 (rows=%d cols=%d)", rows, cols);
    const float sl2 = scale * 1.4426950408889634f;
    hipStream_t st = (hipStream_t)s;
    if (cols <= 2048)
        hipLaunchKernelGGL(softmax_rows_kernel<1>, dim3(rows), dim3(256), 0, st, (f16*)x, ld, cols, sl2);
    else if (cols <= 4096)
        hipLaunchKernelGGL(softmax_rows_kernel<2>, dim3(rows), dim3(256), 0, st, (f16*)x, ld, cols, sl2);
    else if (cols <= 8192)
        hipLaunchKernelGGL(softmax_rows_kernel<4>, dim3(rows), dim3(256), 0, st, (f16*)x, ld, cols, sl2);
    else
        hipLaunchKernelGGL(softmax_rows_kernel<8>, dim3(rows), dim3(256), 0, st, (f16*)x, ld, cols, sl2);
    MDX_LAUNCH_CHECK("mdx_softmax_rows_f16");
    return MDX_OK;
}
