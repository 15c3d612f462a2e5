// Shared device/host helpers for the gfx950 kernels (wave64, fp16 MFMA 16x16x32 / 32x32x16, fp32 accumulation).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The 16-bit operand path exists in two instantiations of the SAME sources (build.sh compiles every 16-bit file twice):
//   default      h16 = _Float16 (fp16 operands: three more mantissa bits, per-image gradient scale for the range), namespace vl_f16
//   -DVL_BF16    h16 = __bf16   (bf16 operands: fp32's range, no gradient-range cliff; BASELINE config 5 / north_star name it), vl_bf16
// Both run on the same-rate MFMAs (v_mfma_f32_16x16x32_{f16,bf16} / 32x32x16), share every layout and differ only in this block.
#ifdef VL_BF16
#define VLNS vl_bf16
#define VL_API(name) name##_bf16
typedef __bf16 h16;
#define H16_MAX 3.3895314e38f        // largest finite bf16
#define VL_DT16 2                    // vl_debug_tensor dtype code of a 16-bit tensor
#else
#define VLNS vl_f16
#define VL_API(name) name##_f16
typedef _Float16 h16;
#define H16_MAX 65504.f
#define VL_DT16 1
#endif
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef h16 h16x4 __attribute__((ext_vector_type(4)));
typedef h16 h16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ float h2f(h16 v) { return (float)v; }
__device__ __forceinline__ h16 f2h(float v) { return (h16)v; }   // RNE (v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32); overflows to inf above H16_MAX
// saturating form for the scaled gradients of the backward elementwise kernels (HBM-bound: the clamp is free there):
// a gradient that outgrows fp16 saturates instead of turning the whole image's gradient into inf / NaN
__device__ __forceinline__ h16 f2h_sat(float v) { return (h16)__builtin_amdgcn_fmed3f(v, -H16_MAX, H16_MAX); }

__device__ __forceinline__ f32x4 mfma16(h16x8 a, h16x8 b, f32x4 c) {
#ifdef VL_BF16
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#endif
}
// acc + a.x * b.x + a.y * b.y on the packed dot unit (v_dot2c_f32_f16 / v_dot2c_f32_bf16): no conversions
__device__ __forceinline__ float dot2_acc(h16x2 a, h16x2 b, float acc) {
#ifdef VL_BF16
    return __builtin_amdgcn_fdot2_f32_bf16(a, b, acc, false);
#else
    return __builtin_amdgcn_fdot2(a, b, acc, false);
#endif
}

// Transposed LDS read: within each 16-lane group, lane 4q+p supplies the address of row q,
// columns 4p..4p+3 of a 4x16 block of 16-bit elements; lane i receives column i of the 4 rows.
// EXEC must be all ones; every address 8-byte aligned.
__device__ __forceinline__ h16x4 lds_read_tr16(const void* lds_addr) {
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds_addr));
    return __builtin_bit_cast(h16x4, t);
}

__device__ __forceinline__ h16x8 cat4(h16x4 lo, h16x4 hi) {
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// async global -> LDS, 16 B per lane; LDS destination = wave-uniform base + lane*16.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(GLB_PTR(gsrc), LDS_PTR(lds_wave_base), 16, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// exact (erf) GELU and its derivative -- hidden_act="gelu" (configuration_vit.py:54)
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// counter-based generator: splitmix64 finaliser on (seed, index)
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// LoRA dropout (lora_dropout of LoraConfig, train_loras.py:88): element `idx` of the branch input of
// module `stream` is kept with probability 1-p; the mask is a pure function of (seed, stream, idx),
// so forward, dgrad and wgrad regenerate it instead of storing it.  Returns 0 or 1/(1-p).
__device__ __forceinline__ float drop_scale(uint64_t seed, uint32_t stream, uint64_t idx, float p, float inv_keep) {
    const uint64_t r = mix64((seed * 0xD1342543DE82EF95ull) ^ ((uint64_t)stream << 48) ^ idx);
    const float u = (float)(r >> 40) * (1.0f / 16777216.0f);
    return u >= p ? inv_keep : 0.f;
}

// XCD-aware bijective remap of a linear workgroup id: workgroups that share an XCD
// (id % 8 under round-robin placement) get a contiguous range of tiles -> neighbouring
// tiles (same A row panel) hit the same L2.  Placement is a speed assumption only.
__device__ __forceinline__ int xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7, x = id & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
}

static inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }
