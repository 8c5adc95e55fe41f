// Shared device/host helpers for libmdx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "mdx.h"

typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MDX_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// error plumbing (host)
void mdx_set_error(const char* fmt, ...);

#define MDX_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            mdx_set_error(__VA_ARGS__);   \
            return MDX_E_INVALID;         \
        }                                 \
    } while (0)

#define MDX_LAUNCH_CHECK(name)                                                        \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) {                                                      \
            mdx_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
            return MDX_E_HIP;                                                         \
        }                                                                             \
    } while (0)

// What a consumer that sums a producer's split-K slabs itself (mdx_groupnorm_from_splitk_f16) needs to know about that
// producer: filled by gemm.hip from the producer's descriptor with the SAME decision path as mdx_gemm_f16.
struct MdxSplitInfo {
    const float* ws;          // [nsplit][M][N] fp32 slabs
    int nsplit, M, N, HoWo, B;
    const float* bias;
    const float* rowbias;
    int rowbias_ld;
    const f16* residual;
    int residual_ld;
    f16* out;                 // the producer's fp16 output [M][N] (out_ld == N)
};
int mdx_internal_split_info(const mdx_gemm_desc* d, MdxSplitInfo* info);

// Library options (mdx_set_option): the tuning / experiment switches that used to be getenv() calls spread over the library.
// Process-global ints with defaults = what the product runs; set them before the first launch they affect.
enum MdxOpt {
    MDX_OPT_GEMM_TUNED,          // 1: consult the measured tile table (gemm_tuned.inc)
    MDX_OPT_GEMM_BM,             // 0 = auto | 64 | 128 | 256: force the M tile
    MDX_OPT_GEMM_BN,             // 0 = auto | 64 | 128: force the N tile
    MDX_OPT_GEMM_RING,           // 0 = auto | 2..5: force the LDS ring depth of the generic kernel
    MDX_OPT_GEMM_HALO,           // 1: 3x3 convs may use the HALO kernel
    MDX_OPT_GEMM_HALO8,          // 1: 8x8 images may use the two-sample HALO tile
    MDX_OPT_GEMM_SPLITK_FIXUP_MAX,   // split-K launches of at most this many splits reduce in the kernel (4)
    MDX_OPT_GEMM_SPREAD,         // 1: single-M-tile launches deal (tile, split) items round-robin to the XCDs
    MDX_OPT_HALO_NSB,            // 0 = auto | 2 | 3 | 4: weight ring depth of the HALO kernel
    MDX_OPT_GN_MIN_BLOCKS,       // GroupNorm: narrow the column blocks until the grid has this many blocks (512)
    MDX_OPT_GN_FUSED,            // 1: small tensors use the one-launch GroupNorm
    MDX_OPT_GN_COL_CHUNKS,       // column-statistics GroupNorm: a column block spans at least this many 16-byte chunks of a pixel row (4)
    MDX_OPT_GEMM_CONV8P,         // 1: eligible 3x3 convs with M >= gemm_conv8p_min_m run on the 256-pixel 8-wave core (conv8p.hip)
    MDX_OPT_GEMM_CONV8P_MIN_M,   // smallest M the 8-wave conv core is chosen for automatically (4096; it also needs >= 128 tiles)
    MDX_OPT_GEMM_SUBPIXEL_MIN_TILES,   // nearest-2x + 3x3 convs with w_sub run the sub-pixel form from this many 256-pixel tiles (32)
    MDX_OPT_GEMM_CONV8P_VAR,     // experiment forms of the 160-column conv8p kernel (conv8p.hip VAR; 0 = the product)
    MDX_OPT_ATTN8,               // (default 0: measured slower, csrc/attention.hip) 1: self-attention launches with >= attn8_min_blocks 256-query blocks use the eight-wave kernel; 2: always when eligible
    MDX_OPT_ATTN8_MIN_BLOCKS,    // (192)
    MDX_OPT_GN_WIDE_ROWS,        // column-statistics GroupNorm on tensors of at least this many pixel rows (B * H * W) uses the widest line-aligned column blocks (default 0 = never: only pays with pre-folded statistics, profiles/r04_gn_bench.txt)
    MDX_OPT_GN_FUSED_SMALL,      // (default 0: -0.2 % / -0.15 % / 0 on Wukong / 768^2 / GLIDE, within noise) 1: the one-launch GroupNorm runs 256-thread blocks when its (column block, sample) grid has >= 512 blocks
    MDX_OPT_GN_BOOST_MB,         // column-statistics GroupNorm on tensors of at least this many MB launches four times the pixel slabs (40; 0 = never)
    MDX_OPT_ATTN_OCC3,           // 1: the D <= 64 attention kernels are built for three blocks per CU (<= 168 VGPRs) instead of two
    MDX_OPT_ATTN_KV_SPLIT,       // split-KV attention (mdx_attention_splitkv_f16 with a workspace): 0 never, 1 auto (fill the chip's block slots), >= 2 force that many splits
    MDX_OPT_ATTN_FAST_STAGE,     // 1: the attention kernel issues its full KV tiles with per-lane offsets computed once + a scalar tile offset (0 = per-tile address arithmetic)
    MDX_OPT_GN_PREFETCH,         // 1: the GroupNorm kernels fetch their affine parameters (and, where a thread's pixels fit its registers, the pixels) at the top of the kernel, ahead of the statistics (0 = the round-4 order: parameters after the statistics, small tensors read twice)
    MDX_OPT_GEMM_DENSE_ISSUE,    // 1: dense launches of the generic GEMM kernel (ksize 1, stride 1, one source) keep the per-lane source offset fixed and put the K offset in the DMA instructions' scalar operand (0 = the general tap / source decode per K tile)
    MDX_OPT_GEMM_LN_PREFETCH,    // 1: LayerNorm-fold consumers (mdx_gemm_desc.ln_stats) fetch their rows' statistics partials and S[n] before the K loop (0 = at the head of the epilogue)
    MDX_OPT_GEMM_LEAN_DENSE,     // 1: dense row-major launches run the lean kernel of dense.hip (division-free prologue, first DMAs ~80 instructions in, epilogue reads prefetched before the K loop); 0 = the generic gemm_kernel (same bits)
    MDX_OPT_ATTN_PIPE,           // 1: full-tile unmasked unsplit attention launches (D <= 80) run the software-pipelined kernel (attn_pipe_kernel: QK^T of tile t + 1, softmax and PV of tile t interleaved MFMA by MFMA inside every wave; same bits); 0 = attn_kernel
    MDX_OPT_COUNT
};
int mdx_opt(int id);

// hipFuncSetAttribute (the dynamic-LDS limit of a kernel) is PER DEVICE: one flag per (kernel instantiation, device), so that a C
// caller that drives several GPUs from one process raises the limit on each of them (a process-wide `static bool` set it on the
// first device only and the launch on the second failed with MDX_E_HIP).
struct MdxPerDeviceOnce {
    bool done[64] = {};
    bool first() {       // true exactly once per device (devices >= 64: always true -- the attribute call is idempotent)
        int dv = 0;
        if (hipGetDevice(&dv) != hipSuccess || dv < 0 || dv >= 64) return true;
        if (done[dv]) return false;
        done[dv] = true;
        return true;
    }
};

// ---- device helpers
// sigmoid(x) = 1 / (1 + 2^(-x log2 e)) on the raw transcendental units (v_exp_f32 + v_rcp_f32, 1 ulp each): an IEEE
// division here costs ~10 VALU instructions per element and the epilogues / GroupNorm apply it to every output.
__device__ __forceinline__ float sigmoid_fast(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_fast(x); }

// x * sigmoid(1.702 x): CLIP's QuickGELU (wukong-huahua/ldm/modules/encoders/text_encoder.py:67-74)
__device__ __forceinline__ float quick_gelu_f(float x) { return x * sigmoid_fast(1.702f * x); }

// ops.GeLU == tanh approximation: 0.5 x (1 + tanh(u)), u = sqrt(2/pi)(x + 0.044715 x^3);
// 0.5 (1 + tanh u) == sigmoid(2u) exactly, and the sigmoid form saturates cleanly for large |u|
__device__ __forceinline__ float gelu_tanh_f(float x) {
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    return x * sigmoid_fast(2.0f * u);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// 16-byte direct-to-LDS DMA through a buffer descriptor: per-lane source byte offset `voff`
// (offsets >= num_records return zeros: this is how conv zero padding and tile tails are done),
// destination = wave-uniform `lds` + lane*16.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, void* lds_wave_uniform, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, MDX_LDS_PTR(lds_wave_uniform), 16, voff, 0, 0, 0);
}

// The same with a wave-uniform byte offset in the instruction's SCALAR offset operand: `voff` can then stay fixed across the K loop
// and the issue costs no per-lane arithmetic.
__device__ __forceinline__ void dma16s(__amdgpu_buffer_rsrc_t rsrc, void* lds_wave_uniform, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, MDX_LDS_PTR(lds_wave_uniform), 16, voff, soff, 0, 0);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
}

#define MDX_OOB 0x80000000u

// Kernel-argument warm-up (round 5 experiment, -DMDX_KERNARG_TOUCH=1): a launch's parameter block (GemmParams: ~430 bytes = 7
// cache lines) is read by scalar loads wherever the compiler first needs a field, each first touch of a line a miss behind an
// s_waitcnt -- the ISA of gemm_kernel has eight such waits in its first 700 instructions.  One dword of every line is requested
// here, at the top of the kernel, all in flight together behind ONE wait; the later loads hit the scalar cache.
#ifndef MDX_KERNARG_TOUCH
#define MDX_KERNARG_TOUCH 1
#endif
template <int BYTES>
__device__ __forceinline__ void mdx_kernarg_touch() {
#if MDX_KERNARG_TOUCH
    typedef __attribute__((address_space(4))) const int* ka_ptr;
    const ka_ptr ka = (ka_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    int acc = 0;
#pragma unroll
    for (int o = 0; o < BYTES; o += 64) acc |= ka[o / 4];
    asm volatile("" ::"s"(acc));
#endif
}
