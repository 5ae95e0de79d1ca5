// Shared device helpers for the gfx950 kernels (wave = 64 lanes, MFMA fragments, half rounding).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/jukebox_hip.h"

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- host-side error plumbing --------------------------------------------------------------
void jb_set_error(const std::string& msg);
#define JB_REQUIRE(cond, msg)                                              \
    do {                                                                   \
        if (!(cond)) {                                                     \
            jb_set_error(std::string(__func__) + ": " + (msg));            \
            return JB_ERR_ARG;                                             \
        }                                                                  \
    } while (0)
#define JB_UNSUPPORTED(msg)                                                \
    do {                                                                   \
        jb_set_error(std::string(__func__) + ": " + (msg));                \
        return JB_ERR_UNSUPPORTED;                                         \
    } while (0)
#define JB_CHECK_LAUNCH()                                                  \
    do {                                                                   \
        hipError_t e__ = hipGetLastError();                                \
        if (e__ != hipSuccess) {                                           \
            jb_set_error(std::string(__func__) + ": " + hipGetErrorString(e__)); \
            return JB_ERR_HIP;                                             \
        }                                                                  \
    } while (0)
#define JB_HIP(call)                                                       \
    do {                                                                   \
        hipError_t e__ = (call);                                           \
        if (e__ != hipSuccess) {                                           \
            jb_set_error(std::string(__func__) + ": " #call ": " + hipGetErrorString(e__)); \
            return JB_ERR_HIP;                                             \
        }                                                                  \
    } while (0)

// ---- per-dtype MFMA fragment description ---------------------------------------------------
// One MFMA "k-tile" holds KT contraction elements; lane l owns E consecutive ones starting at
// (l >> 4) * E, for row/col (l & 15) of its operand.  f16: v_mfma_f32_16x16x32_f16 (one
// instruction per k-tile).  f32: four v_mfma_f32_16x16x4_f32, element e of the fragment feeding
// the e-th instruction -- exact fp32 (an fmaf chain), used for the parity mode and the conv stacks.
template <typename T> struct Frag;
template <> struct Frag<f16> {
    typedef f16x8 vec;
    static constexpr int E = 8, KT = 32;
};
template <> struct Frag<float> {
    typedef f32x4 vec;
    static constexpr int E = 4, KT = 16;
};

__device__ __forceinline__ f32x4 jb_mfma(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 jb_mfma(f32x4 a, f32x4 b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
    return c;
}

template <typename T> __device__ __forceinline__ typename Frag<T>::vec jb_zero_frag() {
    typename Frag<T>::vec v;
#pragma unroll
    for (int e = 0; e < Frag<T>::E; ++e) v[e] = (T)0;
    return v;
}

// Operand fragment of one activation row: E consecutive channels from k0.
template <typename T>
__device__ __forceinline__ typename Frag<T>::vec load_row_frag(const T* __restrict__ row, bool valid, int k0, int K, bool vec) {
    constexpr int E = Frag<T>::E;
    typename Frag<T>::vec v = jb_zero_frag<T>();
    if (valid) {
        if (vec && k0 + E <= K) {
            v = *reinterpret_cast<const typename Frag<T>::vec*>(row + k0);
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e)
                if (k0 + e < K) v[e] = row[k0 + e];
        }
    }
    return v;
}

// Branch-free variants for the fast paths: unconditional 16-byte load (caller clamps the address) and a
// per-element select.  A guarded load compiles to a branch plus s_waitcnt, which serialises cold loads.
template <typename T>
__device__ __forceinline__ typename Frag<T>::vec ld_frag(const T* __restrict__ p) {
    return *reinterpret_cast<const typename Frag<T>::vec*>(p);
}
template <typename T>
__device__ __forceinline__ typename Frag<T>::vec keep_frag(bool keep, typename Frag<T>::vec v) {
#pragma unroll
    for (int e = 0; e < Frag<T>::E; ++e) v[e] = keep ? v[e] : (T)0;
    return v;
}

// Round an fp32 value to the storage type and back (models "this tensor is half in the reference").
template <typename T> __device__ __forceinline__ float jb_round(float x);
template <> __device__ __forceinline__ float jb_round<float>(float x) { return x; }
template <> __device__ __forceinline__ float jb_round<f16>(float x) { return (float)(f16)x; }

// quick_gelu (jukebox/transformer/ops.py:33-35): x * sigmoid(1.702 x); each op rounds in half mode.
template <typename T> __device__ __forceinline__ float jb_quick_gelu(float x) {
    float u = jb_round<T>(1.702f * x);
    float s = jb_round<T>(1.0f / (1.0f + expf(-u)));
    return jb_round<T>(x * s);
}

template <typename T> __device__ __forceinline__ float jb_apply_act(float v, int act) {
    if (act == JB_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == JB_ACT_QUICK_GELU) return jb_quick_gelu<T>(v);
    return v;
}

// Wave-wide (64-lane) reductions without the LDS crossbar: four DPP steps reduce each 16-lane row in the VALU
// (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror), then the four row results are read with
// v_readlane and combined.  A __shfl_xor butterfly is six dependent ds_bpermute round trips (~100+ cycles each).
template <int CTRL> __device__ __forceinline__ float jb_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float jb_readlane(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float jb_wave_sum(float v) {
    v += jb_dpp<0xB1>(v);
    v += jb_dpp<0x4E>(v);
    v += jb_dpp<0x141>(v);
    v += jb_dpp<0x140>(v);
    return (jb_readlane(v, 0) + jb_readlane(v, 16)) + (jb_readlane(v, 32) + jb_readlane(v, 48));
}
__device__ __forceinline__ float jb_wave_max(float v) {
    v = fmaxf(v, jb_dpp<0xB1>(v));
    v = fmaxf(v, jb_dpp<0x4E>(v));
    v = fmaxf(v, jb_dpp<0x141>(v));
    v = fmaxf(v, jb_dpp<0x140>(v));
    return fmaxf(fmaxf(jb_readlane(v, 0), jb_readlane(v, 16)), fmaxf(jb_readlane(v, 32), jb_readlane(v, 48)));
}

__device__ __forceinline__ float jb_load_any(const void* p, int dtype, int64_t i) {
    return dtype == JB_F16 ? (float)((const f16*)p)[i] : ((const float*)p)[i];
}
