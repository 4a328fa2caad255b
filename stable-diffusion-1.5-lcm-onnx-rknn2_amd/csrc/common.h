// Shared device/host helpers for the gfx950 LCM kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef _Float16 half_t;
// hipFuncSetAttribute (the dynamic-LDS limit of a kernel) holds for the CURRENT device only: a process that drives several
// GPUs (LCM_DEVICES=all: worker i on cuda:i) must set it once per device, not once per process.
#include <atomic>
#include <mutex>
// `if (auto g = once.first()) { g.check(hipFuncSetAttribute(...)); }`: the guard holds the mutex while the caller raises the
// limit, and the device is marked done only when the guard goes out of scope with every call having succeeded -- a second
// thread making its first launch on the same device waits for the attribute instead of launching ahead of it.
struct LcmDevOnce {
    std::atomic<unsigned long long> done{0};
    std::mutex mu;
    struct Guard {
        LcmDevOnce* o;
        unsigned long long bit;
        hipError_t err;
        Guard(LcmDevOnce* o_, unsigned long long b) : o(o_), bit(b), err(hipSuccess) {}
        Guard(const Guard&) = delete;
        Guard(Guard&& g) : o(g.o), bit(g.bit), err(g.err) { g.o = nullptr; }
        explicit operator bool() const { return o != nullptr; }
        void check(hipError_t e) { if (e != hipSuccess && err == hipSuccess) err = e; }
        ~Guard() {
            if (!o) return;
            if (err == hipSuccess) o->done.fetch_or(bit, std::memory_order_release);
            else fprintf(stderr, "[lcm] hipFuncSetAttribute failed: %s\n", hipGetErrorString(err));
            o->mu.unlock();
        }
    };
    Guard first() {
        int d = 0;
        (void)hipGetDevice(&d);
        const unsigned long long bit = 1ull << (d & 63);
        if (done.load(std::memory_order_acquire) & bit) return Guard(nullptr, 0);
        mu.lock();
        if (done.load(std::memory_order_relaxed) & bit) { mu.unlock(); return Guard(nullptr, 0); }
        return Guard(this, bit);
    }
};

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
// output rounding), arranged for the VALU (the GEGLU epilogue of ff.net.0 is VALU-issue-bound: ~3500 of the ~4900 issue
// cycles of a 128x128 tile are this function):  with z = |x|/sqrt2, t = 1/(1 + p z), erf(z) = 1 - poly(t) e^{-z^2},
//   gelu(x) = 0.5 x + 0.5 |x| erf(z) = max(x, 0) - |x| * [0.5 poly(t)] * 2^(-x^2 * 0.5 log2 e)
// one v_rcp + one v_exp + 5 FMAs + 5 other VALU = 14 issue slots (the textbook arrangement took 20).
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(ax, 0.3275911f * 0.70710678118654752f, 1.0f));
    float poly = __builtin_fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
    poly = __builtin_fmaf(t, poly, 0.5f * 1.421413741f);
    poly = __builtin_fmaf(t, poly, 0.5f * -0.284496736f);
    poly = __builtin_fmaf(t, poly, 0.5f * 0.254829592f);
    poly *= t;
    const float e = __builtin_amdgcn_exp2f((x * x) * -0.72134752044448170f);        // 2^(-x^2/2 * log2 e)
    return __builtin_fmaf(-ax, poly * e, fmaxf(x, 0.0f));
}

// Sums over the 16 lanes of a DPP row (lanes 16g .. 16g+15) of eight values at once, results in every lane of the row: the
// xor-butterfly 1, 2, 4, 8 as four v_add_f32_dpp per value (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror,
// row_mirror) instead of four ds_bpermute round trips through the LDS each.  Same additions in the same tree as
// `v += __shfl_xor(v, o)` for o = 1, 2, 4, 8 (after the quad steps all lanes of a quad hold one value, so the mirrored partner
// holds what the xor partner holds): bit-identical.  Inline asm: hipcc emits v_mov_b32_dpp + v_add_f32 for the builtin (twice the
// VALU).  The eight chains are interleaved, so a step reads a register eight instructions after the step before wrote it; the
// s_nop 1 in front covers the 2 wait states between the compiler's last VALU write of an operand and the first DPP read.
#define LCM_DPP_STEP8(CTRL)                                                                                              \
    asm("s_nop 1\n\t"                                                                                                    \
        "v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                  \
        "v_add_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                  \
        "v_add_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                  \
        "v_add_f32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                  \
        "v_add_f32_dpp %4, %4, %4 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                  \
        "v_add_f32_dpp %5, %5, %5 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                  \
        "v_add_f32_dpp %6, %6, %6 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                  \
        "v_add_f32_dpp %7, %7, %7 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1"                                       \
        : "+v"(a[0]), "+v"(b[0]), "+v"(a[1]), "+v"(b[1]), "+v"(a[2]), "+v"(b[2]), "+v"(a[3]), "+v"(b[3]))
__device__ __forceinline__ void row16_sum8(float (&a)[4], float (&b)[4]) {
    LCM_DPP_STEP8("quad_perm:[1,0,3,2]");
    LCM_DPP_STEP8("quad_perm:[2,3,0,1]");
    LCM_DPP_STEP8("row_half_mirror");
    LCM_DPP_STEP8("row_mirror");
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
