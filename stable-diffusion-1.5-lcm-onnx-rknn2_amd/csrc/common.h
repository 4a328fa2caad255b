// Shared device/host helpers for the gfx950 LCM kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define LCM_OK 0
#define LCM_EINVAL (-1)
#define LCM_ENODEV (-2)

void lcm_set_error(const char* fmt, ...);
void lcm_prof_start(const char* name, hipStream_t s);   // no-ops unless lcm_profile_begin() is active
void lcm_prof_stop(hipStream_t s);

#define LCM_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            lcm_set_error(__VA_ARGS__);   \
            return LCM_EINVAL;            \
        }                                 \
    } while (0)

#define LCM_CHECK_LAUNCH(name)                                                  \
    do {                                                                        \
        hipError_t e__ = hipGetLastError();                                     \
        if (e__ != hipSuccess) {                                                \
            lcm_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return (int)e__;                                                    \
        }                                                                       \
    } while (0)

// x * sigmoid(x) with v_exp_f32 + v_rcp_f32 (1 ulp-class; well inside fp16 output rounding)
__device__ __forceinline__ float silu_f(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
// exact-GELU 0.5 x (1 + erf(x/sqrt2)) with erf from Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far below the fp16
// output rounding): one v_rcp + one v_exp + a degree-5 Horner instead of libm erff (~4x fewer VALU in the GEGLU epilogue)
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.0f - poly * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
    const float erfv = copysignf(erf_abs, x);
    return 0.5f * x * (1.0f + erfv);
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
