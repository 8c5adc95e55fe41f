// Implicit-GEMM conv3x3 / conv1x1 / Dense for gfx950 (MI355X).
//
//   out[m][n] = sum_k A[m][k] * W[n][k]      m = (b,yo,xo), k = (tap,cin), NHWC fp16, fp32 accumulate
//
// Replaces the reference's nn.Conv2d / nn.Dense call sites (include/mdx.h cites them).
//
// Design (MI355X-first):
//  * 256 threads = 4 wave64 (2x2), block tile 128 x BN (BN = 128 | 64), BK = 64, MFMA 32x32x16 f16.
//  * Both operand tiles go HBM -> LDS with `buffer_load_dwordx4 ... lds` (16 B per lane, no VGPR
//    round trip).  The im2col gather, zero padding, stride-2, nearest-2x upsample and the
//    two-source channel concat are all folded into the per-lane SOURCE offset; out-of-range
//    offsets rely on the buffer descriptor's bounds check returning zeros.
//  * LDS image is lane-linear per DMA (8 rows x 128 B); the bank-conflict XOR swizzle
//    (chunk ^= (row>>1)&7) is applied on the source side and again on the ds_read_b128 side.
//  * Double-buffered LDS, one barrier per K tile.
//  * Epilogue is staged through LDS so that global stores are full 16-B / 128-B-row coalesced;
//    bias, per-sample time-embedding bias, residual add, GEGLU and the transposed (V^T) store
//    are fused there.  Small-M layers use split-K (fp32 slabs + a fused reduce/epilogue kernel).
#include "mdx_common.h"

namespace {

struct GemmParams {
    const f16* a;
    const f16* a2;
    const f16* w;
    const float* bias;
    const float* rowbias;
    const f16* residual;
    f16* out;
    float* ws;
    int c1, c2, cin;
    int rowbias_ld, residual_ld, out_ld;
    int B, H, W, Ho, Wo, HoWo, M, N, K;
    int ksize, stride, upsample, pad;
    int epilogue, out_mode;
    int ktiles, ktiles_per_split, nsplit;
    int tiles_m, tiles_n;
    unsigned a_bytes, a2_bytes, w_bytes;
};

constexpr int BM = 128;
constexpr int BK = 64;
constexpr int ROWB = BK * 2;  // 128 bytes per LDS row

__device__ __forceinline__ f16x4 cvt4(float a, float b, float c, float d) {
    f16x4 v;
    v[0] = (f16)a;
    v[1] = (f16)b;
    v[2] = (f16)c;
    v[3] = (f16)d;
    return v;
}

// Shared fused epilogue for 8 consecutive output columns of one row (fp32 in, fp16 out).
__device__ __forceinline__ void epilogue_store_row8(const GemmParams& p, float (&f)[8], int m, int n) {
    if (p.rowbias) {
        const int b = m / p.HoWo;
        const float4* rb = reinterpret_cast<const float4*>(p.rowbias + (size_t)b * p.rowbias_ld + n);
        const float4 r0 = rb[0], r1 = rb[1];
        f[0] += r0.x; f[1] += r0.y; f[2] += r0.z; f[3] += r0.w;
        f[4] += r1.x; f[5] += r1.y; f[6] += r1.z; f[7] += r1.w;
    }
    if (p.residual) {
        const f16x8 r = *reinterpret_cast<const f16x8*>(p.residual + (size_t)m * p.residual_ld + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] += (float)r[e];
    }
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (f16)f[e];
    *reinterpret_cast<f16x8*>(p.out + (size_t)m * p.out_ld + n) = o;
}

template <int BN, bool SWAP, bool FASTK>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmParams p) {
    constexpr int TN = BN / 64;             // 32-wide MFMA tiles per wave along n
    constexpr int A_BYTES = BM * ROWB;      // 16 KiB
    constexpr int B_BYTES = BN * ROWB;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int BJ = BN / 32;             // B-tile DMA instructions per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int hi = lane >> 5;
    const int l31 = lane & 31;

    const int tile_n = blockIdx.x % p.tiles_n;
    const int tile_m = blockIdx.x / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int split = blockIdx.y;
    const int kt_begin = split * p.ktiles_per_split;
    const int kt_end = min(p.ktiles, kt_begin + p.ktiles_per_split);

    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(p.a, p.a_bytes);
    const __amdgpu_buffer_rsrc_t rs_a2 = make_rsrc(p.a2 ? p.a2 : p.a, p.a2 ? p.a2_bytes : p.a_bytes);
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, p.w_bytes);

    // ---- loader coordinates (fixed per thread across K tiles)
    // A: wave-instruction j covers rows wave*32 + j*8 .. +7; lane -> row +(lane>>3), physical chunk lane&7
    int a_pix[4];   // b*H*W
    int a_y[4], a_x[4];
    unsigned a_chunk[4];  // logical 16-B chunk (un-swizzled) this lane fetches
    bool a_ok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = wave * 32 + j * 8 + (lane >> 3);
        const int m = m0 + row;
        a_ok[j] = m < p.M;
        const int mm = a_ok[j] ? m : 0;
        const int b = mm / p.HoWo;
        const int rem = mm - b * p.HoWo;
        const int yo = rem / p.Wo;
        const int xo = rem - yo * p.Wo;
        a_pix[j] = b * p.H * p.W;
        a_y[j] = yo * p.stride - p.pad;
        a_x[j] = xo * p.stride - p.pad;
        a_chunk[j] = (unsigned)((lane & 7) ^ ((row >> 1) & 7));
    }
    unsigned b_off[BJ];  // byte offset of (n, chunk) at k0 = 0, or OOB
    unsigned b_chunk[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int row = wave * (BN / 4) + j * 8 + (lane >> 3);
        const int n = n0 + row;
        b_chunk[j] = (unsigned)((lane & 7) ^ ((row >> 1) & 7));
        b_off[j] = (n < p.N) ? (unsigned)(((size_t)n * p.K + b_chunk[j] * 8) * 2) : MDX_OOB;
    }
    const int Hs = p.upsample ? 2 * p.H : p.H;  // extent the taps are clipped against
    const int Ws = p.upsample ? 2 * p.W : p.W;

    auto stage_tile = [&](int kt, int buf) {
        char* sbase = smem + buf * STAGE;
        const int k0 = kt * BK;
        if constexpr (FASTK) {
            // Cin % 64 == 0 (and c1 % 64 == 0): the whole K tile lies in one tap and one source.
            const int tap = k0 / p.cin;
            int ci0 = k0 - tap * p.cin;
            const int ky = (p.ksize == 3) ? tap / 3 : 0;
            const int kx = (p.ksize == 3) ? tap - ky * 3 : 0;
            const bool second = ci0 >= p.c1;
            const int cs = second ? p.c2 : p.c1;
            if (second) ci0 -= p.c1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int yi = a_y[j] + ky, xi = a_x[j] + kx;
                const bool ok = a_ok[j] && yi >= 0 && yi < Hs && xi >= 0 && xi < Ws;
                if (p.upsample) {
                    yi >>= 1;
                    xi >>= 1;
                }
                const unsigned off =
                    ok ? (unsigned)(((size_t)(a_pix[j] + yi * p.W + xi) * cs + ci0 + a_chunk[j] * 8) * 2) : MDX_OOB;
                void* dst = sbase + (wave * 4 + j) * 1024;
                if (second)
                    dma16(rs_a2, dst, off);
                else
                    dma16(rs_a, dst, off);
            }
        } else {
            // generic: per-lane tap decode (single source only)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kk = k0 + (int)a_chunk[j] * 8;
                const int tap = kk / p.cin;
                const int ci = kk - tap * p.cin;
                const int ky = (p.ksize == 3) ? tap / 3 : 0;
                const int kx = (p.ksize == 3) ? tap - ky * 3 : 0;
                int yi = a_y[j] + ky, xi = a_x[j] + kx;
                const bool ok = a_ok[j] && kk < p.K && yi >= 0 && yi < Hs && xi >= 0 && xi < Ws;
                if (p.upsample) {
                    yi >>= 1;
                    xi >>= 1;
                }
                const unsigned off = ok ? (unsigned)(((size_t)(a_pix[j] + yi * p.W + xi) * p.cin + ci) * 2) : MDX_OOB;
                dma16(rs_a, sbase + (wave * 4 + j) * 1024, off);
            }
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const bool ok = b_off[j] != MDX_OOB && (FASTK || (k0 + (int)b_chunk[j] * 8) < p.K);
            const unsigned off = ok ? b_off[j] + (unsigned)k0 * 2 : MDX_OOB;
            dma16(rs_w, sbase + A_BYTES + (wave * BJ + j) * 1024, off);
        }
    };

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read offsets: row*128 + ((2s+hi) ^ ((row>>1)&7))*16, (row>>1)&7 == (lane>>1)&7 for all our row bases
    const int swz = (lane >> 1) & 7;
    const int a_row_off = (wm * 64 + l31) * ROWB;
    const int b_row_off = A_BYTES + (wn * (BN / 2) + l31) * ROWB;

    if (kt_begin < kt_end) {
        stage_tile(kt_begin, 0);
    }
    __syncthreads();  // (compiler inserts vmcnt(0) for the pending LDS-DMA writes)
    int buf = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        if (kt + 1 < kt_end) stage_tile(kt + 1, buf ^ 1);
        const char* sb = smem + buf * STAGE;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int coff = (((2 * s + hi) ^ swz) << 4);
            f16x8 af[2], bf[TN];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const f16x8*>(sb + a_row_off + i * 32 * ROWB + coff);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f16x8*>(sb + b_row_off + j * 32 * ROWB + coff);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (SWAP)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
        buf ^= 1;
    }

    // ------------------------------------------------------------------ epilogue
    if constexpr (!SWAP) {
        if (p.nsplit > 1) {
            // split-K partial: C layout (col = lane&31 -> n, rows -> m), 128-B row segments per store
            float* wsz = p.ws + (size_t)split * p.M * p.N;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n0 + wn * (BN / 2) + j * 32 + l31;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        if (m < p.M && n < p.N) wsz[(size_t)m * p.N + n] = acc[i][j][r];
                    }
                }
            return;
        }
        // transposed store (V^T): stage [n][m]
        constexpr int TLD = BM + 8;
        f16* stg = reinterpret_cast<f16*>(smem);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n_l = wn * (BN / 2) + j * 32 + l31;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int m_l = wm * 64 + i * 32 + 8 * g + 4 * hi;
                    *reinterpret_cast<f16x4*>(&stg[n_l * TLD + m_l]) =
                        cvt4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                }
            }
        __syncthreads();
        const int chunk = tid & 15, r0 = tid >> 4;
        const int m = m0 + chunk * 8;
#pragma unroll
        for (int pass = 0; pass < BN / 16; ++pass) {
            const int nrow = r0 + pass * 16;
            const int n = n0 + nrow;
            if (n < p.N && m < p.M) {
                f16x8 v = *reinterpret_cast<const f16x8*>(&stg[nrow * TLD + chunk * 8]);
                if (p.bias) {
                    const float bb = p.bias[n];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (f16)((float)v[e] + bb);
                }
                const int b = m / p.HoWo;
                const int tok = m - b * p.HoWo;
                *reinterpret_cast<f16x8*>(p.out + ((size_t)b * p.N + n) * p.out_ld + tok) = v;
            }
        }
    } else {
        // row-major store: C^T layout (col = lane&31 -> m, 4 consecutive n per register group)
        constexpr int SLD = BN + 8;
        f16* stg = reinterpret_cast<f16*>(smem);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int m_l = wm * 64 + i * 32 + l31;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n_l = wn * (BN / 2) + j * 32 + 8 * g + 4 * hi;
                    *reinterpret_cast<f16x4*>(&stg[m_l * SLD + n_l]) =
                        cvt4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                }
            }
        __syncthreads();
        if (p.epilogue == MDX_EPI_GEGLU) {
            if constexpr (BN == 128) {
                // tile = 64 'a' columns | 64 'gate' columns -> 64 outputs at column n0/2
                const int chunk = tid & 7, r0 = tid >> 3;
                const int pn = n0 + chunk * 8;       // packed column of the 'a' part
                const int on = (n0 >> 1) + chunk * 8;  // output column
                float ba[8], bg[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    ba[e] = (p.bias && pn < p.N) ? p.bias[pn + e] : 0.f;
                    bg[e] = (p.bias && pn < p.N) ? p.bias[pn + 64 + e] : 0.f;
                }
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    const int row = r0 + pass * 32;
                    const int m = m0 + row;
                    if (m < p.M && pn < p.N) {
                        const f16x8 va = *reinterpret_cast<const f16x8*>(&stg[row * SLD + chunk * 8]);
                        const f16x8 vg = *reinterpret_cast<const f16x8*>(&stg[row * SLD + 64 + chunk * 8]);
                        float f[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            f[e] = ((float)va[e] + ba[e]) * gelu_tanh_f((float)vg[e] + bg[e]);
                        epilogue_store_row8(p, f, m, on);
                    }
                }
            }
        } else {
            constexpr int CPR = BN / 8;
            constexpr int RPP = 256 / CPR;
            const int chunk = tid % CPR, r0 = tid / CPR;
            const int n = n0 + chunk * 8;
            float bb[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) bb[e] = (p.bias && n < p.N) ? p.bias[n + e] : 0.f;
#pragma unroll
            for (int pass = 0; pass < BM / RPP; ++pass) {
                const int row = r0 + pass * RPP;
                const int m = m0 + row;
                if (m < p.M && n < p.N) {
                    const f16x8 v = *reinterpret_cast<const f16x8*>(&stg[row * SLD + chunk * 8]);
                    float f[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = (float)v[e] + bb[e];
                    epilogue_store_row8(p, f, m, n);
                }
            }
        }
    }
}

// split-K reduce + fused epilogue: one thread per (m, 8 output columns)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmParams p) {
    const bool geglu = p.epilogue == MDX_EPI_GEGLU;
    const int ncols = geglu ? p.N / 2 : p.N;
    const int cpr = ncols / 8;
    const size_t total = (size_t)p.M * cpr;
    const size_t slab = (size_t)p.M * p.N;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int m = (int)(idx / cpr);
        const int oc = (int)(idx - (size_t)m * cpr) * 8;
        float f[8];
        if (geglu) {
            const int pa = (oc >> 6) * 128 + (oc & 63);
            float a[8], g[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                a[e] = p.bias ? p.bias[pa + e] : 0.f;
                g[e] = p.bias ? p.bias[pa + 64 + e] : 0.f;
            }
            for (int z = 0; z < p.nsplit; ++z) {
                const float4* s = reinterpret_cast<const float4*>(p.ws + z * slab + (size_t)m * p.N + pa);
                const float4* t = reinterpret_cast<const float4*>(p.ws + z * slab + (size_t)m * p.N + pa + 64);
                const float4 s0 = s[0], s1 = s[1], t0 = t[0], t1 = t[1];
                a[0] += s0.x; a[1] += s0.y; a[2] += s0.z; a[3] += s0.w; a[4] += s1.x; a[5] += s1.y; a[6] += s1.z; a[7] += s1.w;
                g[0] += t0.x; g[1] += t0.y; g[2] += t0.z; g[3] += t0.w; g[4] += t1.x; g[5] += t1.y; g[6] += t1.z; g[7] += t1.w;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = a[e] * gelu_tanh_f(g[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = p.bias ? p.bias[oc + e] : 0.f;
            for (int z = 0; z < p.nsplit; ++z) {
                const float4* s = reinterpret_cast<const float4*>(p.ws + z * slab + (size_t)m * p.N + oc);
                const float4 s0 = s[0], s1 = s[1];
                f[0] += s0.x; f[1] += s0.y; f[2] += s0.z; f[3] += s0.w; f[4] += s1.x; f[5] += s1.y; f[6] += s1.z; f[7] += s1.w;
            }
        }
        if (p.out_mode == MDX_OUT_TRANSPOSED) {
            const int b = m / p.HoWo;
            const int tok = m - b * p.HoWo;
#pragma unroll
            for (int e = 0; e < 8; ++e) p.out[((size_t)b * p.N + oc + e) * p.out_ld + tok] = (f16)f[e];
        } else {
            epilogue_store_row8(p, f, m, oc);
        }
    }
}

int fill_params(const mdx_gemm_desc* d, GemmParams& p) {
    MDX_REQUIRE(d && d->a && d->w && d->out, "mdx_gemm_f16: null pointer");
    MDX_REQUIRE(d->ksize == 1 || d->ksize == 3, "mdx_gemm_f16: ksize must be 1 or 3 (got %d)", d->ksize);
    MDX_REQUIRE(d->stride == 1 || d->stride == 2, "mdx_gemm_f16: stride must be 1 or 2");
    MDX_REQUIRE(!(d->upsample && d->stride != 1), "mdx_gemm_f16: upsample requires stride 1");
    MDX_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->N > 0, "mdx_gemm_f16: bad extents");
    MDX_REQUIRE(d->c1 > 0 && d->c1 % 8 == 0 && d->c2 >= 0 && d->c2 % 8 == 0, "mdx_gemm_f16: c1/c2 must be multiples of 8");
    MDX_REQUIRE((d->c2 == 0) == (d->a2 == nullptr), "mdx_gemm_f16: a2/c2 mismatch");
    MDX_REQUIRE(d->N % 8 == 0, "mdx_gemm_f16: N must be a multiple of 8 (got %d)", d->N);
    p.a = (const f16*)d->a;
    p.a2 = (const f16*)d->a2;
    p.w = (const f16*)d->w;
    p.bias = d->bias;
    p.rowbias = d->rowbias;
    p.residual = (const f16*)d->residual;
    p.out = (f16*)d->out;
    p.ws = (float*)d->workspace;
    p.c1 = d->c1;
    p.c2 = d->c2;
    p.cin = d->c1 + d->c2;
    p.rowbias_ld = d->rowbias_ld;
    p.residual_ld = d->residual_ld;
    p.out_ld = d->out_ld;
    p.B = d->B;
    p.H = d->H;
    p.W = d->W;
    p.ksize = d->ksize;
    p.stride = d->stride;
    p.upsample = d->upsample ? 1 : 0;
    p.pad = d->ksize == 3 ? 1 : 0;
    const int Hs = p.upsample ? 2 * d->H : d->H, Ws = p.upsample ? 2 * d->W : d->W;
    p.Ho = (Hs + 2 * p.pad - d->ksize) / d->stride + 1;
    p.Wo = (Ws + 2 * p.pad - d->ksize) / d->stride + 1;
    p.HoWo = p.Ho * p.Wo;
    p.M = d->B * p.HoWo;
    p.N = d->N;
    p.K = d->ksize * d->ksize * p.cin;
    p.epilogue = d->epilogue;
    p.out_mode = d->out_mode;
    MDX_REQUIRE(p.epilogue == MDX_EPI_NONE || p.epilogue == MDX_EPI_GEGLU, "mdx_gemm_f16: bad epilogue");
    MDX_REQUIRE(p.out_mode == MDX_OUT_ROWMAJOR || p.out_mode == MDX_OUT_TRANSPOSED, "mdx_gemm_f16: bad out_mode");
    if (p.epilogue == MDX_EPI_GEGLU) {
        MDX_REQUIRE(p.N % 128 == 0, "mdx_gemm_f16: GEGLU needs N %% 128 == 0");
        MDX_REQUIRE(p.out_mode == MDX_OUT_ROWMAJOR, "mdx_gemm_f16: GEGLU is row-major only");
    }
    if (p.out_mode == MDX_OUT_TRANSPOSED) {
        MDX_REQUIRE(p.HoWo % 8 == 0 && p.out_ld % 8 == 0, "mdx_gemm_f16: transposed store needs tokens %% 8 == 0");
        MDX_REQUIRE(!p.rowbias && !p.residual, "mdx_gemm_f16: transposed store takes bias only");
    }
    if (p.rowbias) MDX_REQUIRE(p.rowbias_ld % 4 == 0, "mdx_gemm_f16: rowbias_ld must be a multiple of 4");
    if (p.residual) MDX_REQUIRE(p.residual_ld % 8 == 0, "mdx_gemm_f16: residual_ld must be a multiple of 8");
    MDX_REQUIRE(p.out_ld % 8 == 0, "mdx_gemm_f16: out_ld must be a multiple of 8");
    const size_t ab = (size_t)d->B * d->H * d->W * d->c1 * 2, a2b = (size_t)d->B * d->H * d->W * d->c2 * 2;
    const size_t wb = (size_t)p.N * p.K * 2;
    MDX_REQUIRE(ab <= 0x80000000ull && a2b <= 0x80000000ull && wb <= 0x80000000ull,
                "mdx_gemm_f16: operand larger than 2 GiB is not addressable by one buffer descriptor");
    p.a_bytes = (unsigned)ab;
    p.a2_bytes = (unsigned)a2b;
    p.w_bytes = (unsigned)wb;
    p.ktiles = (p.K + BK - 1) / BK;
    return MDX_OK;
}

int pick_bn(const GemmParams& p) {
    if (p.epilogue == MDX_EPI_GEGLU) return 128;
    if (p.N % 128 == 0) return 128;
    if (p.N % 64 == 0 || p.N < 128) return 64;
    return (p.N % 128 > 64) ? 128 : 64;
}

int auto_split(const GemmParams& p, int bn) {
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + bn - 1) / bn);
    int ns = 256 / tiles;               // aim for >= ~1 block per CU
    const int maxk = p.ktiles / 4;      // keep >= 4 K tiles (256 k) per split
    if (ns > maxk) ns = maxk;
    if (ns > 32) ns = 32;
    return ns < 1 ? 1 : ns;
}

template <int BN, bool SWAP>
void launch_gemm(const GemmParams& p, bool fastk, dim3 grid, hipStream_t st) {
    const size_t lds = 2 * (size_t)(BM + BN) * ROWB;
    if (fastk)
        hipLaunchKernelGGL((gemm_kernel<BN, SWAP, true>), grid, dim3(256), lds, st, p);
    else
        hipLaunchKernelGGL((gemm_kernel<BN, SWAP, false>), grid, dim3(256), lds, st, p);
}

}  // namespace

extern "C" size_t mdx_gemm_workspace_bytes(const mdx_gemm_desc* d) {
    GemmParams p{};
    if (fill_params(d, p) != MDX_OK) return 0;
    const int bn = pick_bn(p);
    const int ns = d->splitk > 0 ? d->splitk : auto_split(p, bn);
    return ns > 1 ? (size_t)ns * p.M * p.N * sizeof(float) : 0;
}

extern "C" int mdx_gemm_check(const mdx_gemm_desc* d) {
    GemmParams p{};
    int rc = fill_params(d, p);
    if (rc != MDX_OK) return rc;
    const bool fastk = (p.cin % 64 == 0) && (p.c2 == 0 || p.c1 % 64 == 0);
    MDX_REQUIRE(fastk || p.c2 == 0, "mdx_gemm_f16: two-source input needs c1 %% 64 == 0 and Cin %% 64 == 0");
    return MDX_OK;
}

extern "C" int mdx_gemm_f16(const mdx_gemm_desc* d, mdx_stream_t s) {
    GemmParams p{};
    int rc = fill_params(d, p);
    if (rc != MDX_OK) return rc;
    hipStream_t st = (hipStream_t)s;
    const int bn = pick_bn(p);
    int ns = d->splitk > 0 ? d->splitk : auto_split(p, bn);
    if (ns > p.ktiles) ns = p.ktiles;
    if (ns > 1) {
        // shrink to what the caller's workspace can hold
        const size_t slab = (size_t)p.M * p.N * sizeof(float);
        const size_t cap = d->workspace ? d->workspace_bytes / slab : 0;
        if ((size_t)ns > cap) ns = (int)cap;
        if (ns < 1) ns = 1;
        if (d->splitk > 1 && ns != d->splitk) {
            mdx_set_error("mdx_gemm_f16: workspace too small for splitk=%d (need %zu bytes)", d->splitk,
                          (size_t)d->splitk * slab);
            return MDX_E_WORKSPACE;
        }
    }
    p.nsplit = ns;
    p.ktiles_per_split = (p.ktiles + ns - 1) / ns;
    p.nsplit = (p.ktiles + p.ktiles_per_split - 1) / p.ktiles_per_split;  // no empty splits
    ns = p.nsplit;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + bn - 1) / bn;
    const bool fastk = (p.cin % 64 == 0) && (p.c2 == 0 || p.c1 % 64 == 0);
    MDX_REQUIRE(fastk || p.c2 == 0, "mdx_gemm_f16: two-source input needs c1 %% 64 == 0 and Cin %% 64 == 0");
    dim3 grid(p.tiles_m * p.tiles_n, ns);
    const bool swap = (ns == 1) && (p.out_mode == MDX_OUT_ROWMAJOR);
    if (bn == 128) {
        if (swap)
            launch_gemm<128, true>(p, fastk, grid, st);
        else
            launch_gemm<128, false>(p, fastk, grid, st);
    } else {
        if (swap)
            launch_gemm<64, true>(p, fastk, grid, st);
        else
            launch_gemm<64, false>(p, fastk, grid, st);
    }
    MDX_LAUNCH_CHECK("mdx_gemm_f16");
    if (ns > 1) {
        const int ncols = p.epilogue == MDX_EPI_GEGLU ? p.N / 2 : p.N;
        const size_t total = (size_t)p.M * (ncols / 8);
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p);
        MDX_LAUNCH_CHECK("mdx_gemm_f16(splitk reduce)");
    }
    return MDX_OK;
}
