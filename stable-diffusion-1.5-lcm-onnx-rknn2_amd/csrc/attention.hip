// Fused attention softmax(scale * Q K^T) V for gfx950 -- flash-style, no S x S matrix in memory.
//
// Replaces diffusers Attention (attn1 self / attn2 cross) inside BasicTransformerBlock of
// UNet2DConditionModel, reached from backends/cuda_worker.py:221-229.  SD1.5 shapes: 8 heads,
// head_dim 40/80/160, Sq = H*W in {4096,1024,256,64,...}, Sk = Sq (self) or 77 (cross).
//
// Structure (one workgroup = 4 waves = 128 query rows of one (batch, head); 64-key K/V tiles in LDS):
//   * "swapped" QK^T: S^T = K Q^T with mfma_f32_32x32x16_f16 (A = K rows from LDS via ds_read_b128,
//     B = Q rows held in registers), so each lane owns ONE query column: the online-softmax running
//     max / sum / rescale are per-lane scalars (one __shfl_xor(.,32) joins the two lane halves).
//   * the S^T accumulator registers 8s..8s+7, converted to fp16, ARE the B operand of the second MFMA
//     (O^T = V^T P^T), k-slot (half h, element j) <-> key 16s + 8(j>>2) + 4h + (j&3); no LDS round trip.
//   * V stays row-major in LDS ([key][d], coalesced from HBM); the V^T A-fragments come from two
//     ds_read_b64_tr_b16 (hardware transpose) per fragment; row strides are chosen conflict-free.
//   * K/V tiles are register-staged one tile ahead (loads issued before the MFMAs of the current tile).
#include "common.h"
#include <type_traits>

struct AttnParams {
    const half_t* Q; const half_t* K; const half_t* V; half_t* O;
    int ldq, ldk, ldv, ldo;
    int B, heads, Sq, Sk;
    float sc;   // scale * log2(e)
    int causal; // 1: key index > query index is masked (CLIP text encoder); cross/self attention of the UNet: 0
};

template <int D>
struct AttnCfg {
    static constexpr int DK = (D + 15) / 16 * 16;     // k extent of QK^T (mfma K = 16)
    static constexpr int DV = (D + 31) / 32 * 32;     // row extent of O^T (mfma M = 32)
    static constexpr int KS = DK * 2 + 16;            // K row stride bytes: 16 * odd -> b128 conflict-free
    static constexpr int VS0 = DV * 2;
    static constexpr int VS = ((VS0 % 256) == 64 || (VS0 % 256) == 192) ? VS0 : VS0 + 64;  // tr_b16 conflict-free
    static constexpr int NCH = D / 8;                 // 16-byte chunks per row
    static constexpr int LDS_BYTES = 64 * KS + 64 * VS;
    // Row sums on the matrix core: when V has a padding column (DV > D) it is set to 1.0, so row D of
    // O^T = V^T P^T is sum_k P[q][k] (accumulated in fp32, rescaled together with O).  It lands in lanes 0-31,
    // register ONES_REG of d-block ONES_DB.  Needs (D % 32) % 8 < 4, true for 40 and 80.
    static constexpr bool ONES = (DV > D) && ((D % 32) % 8 < 4);
    static constexpr int ONES_DB = D / 32, ONES_REG = 4 * ((D % 32) >> 3) + ((D % 32) & 3);
};

// WAVES = 4: 128 query rows per workgroup (the form used); WAVES = 2: 64 rows, twice the workgroups (kept as a switch: it
// measured slower, see launch_attn).  A query row's arithmetic does not depend on the split: same bits.
template <int D, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void attn_kernel(AttnParams p) {
    using C = AttnCfg<D>;
    constexpr int NT = 64 * WAVES;                      // threads
    constexpr int NLD = (64 * C::NCH + NT - 1) / NT;    // staged chunks per thread per tensor
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + 64 * C::KS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int bh = blockIdx.y, b = bh / p.heads, head = bh - b * p.heads;
    const int q0 = blockIdx.x * (32 * WAVES) + wave * 32;
    const int qrow = q0 + l31;

    const half_t* Qb = p.Q + (long long)b * p.Sq * p.ldq + head * D;
    const half_t* Kb = p.K + (long long)b * p.Sk * p.ldk + head * D;
    const half_t* Vb = p.V + (long long)b * p.Sk * p.ldv + head * D;

    // zero LDS once: pad columns (d >= D) must read as 0
    for (int i = tid * 16; i < C::LDS_BYTES; i += NT * 16) *reinterpret_cast<f4*>(smem + i) = (f4){0.f, 0.f, 0.f, 0.f};
    if (C::ONES) {
        __syncthreads();
        if (tid < 64) *reinterpret_cast<half_t*>(Vs + tid * C::VS + D * 2) = (half_t)1.0f;   // never overwritten: tiles fill cols < D
    }

    // Q fragments (B operand): lane holds Q[qrow][16ks + 8hh + j]
    h8 qf[C::DK / 16];
#pragma unroll
    for (int ks = 0; ks < C::DK / 16; ++ks) {
        const int dc = 16 * ks + 8 * hh;
        h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (qrow < p.Sq && dc < D) v = *reinterpret_cast<const h8*>(Qb + (long long)qrow * p.ldq + dc);
        qf[ks] = v;
    }

    h8 rk[NLD], rv[NLD];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + NT * i;
            h8 a = {0, 0, 0, 0, 0, 0, 0, 0}, c = {0, 0, 0, 0, 0, 0, 0, 0};
            if (idx < 64 * C::NCH) {
                const int r = idx / C::NCH, ch = idx - r * C::NCH;
                const int key = t * 64 + r;
                if (key < p.Sk) {
                    a = *reinterpret_cast<const h8*>(Kb + (long long)key * p.ldk + ch * 8);
                    c = *reinterpret_cast<const h8*>(Vb + (long long)key * p.ldv + ch * 8);
                }
            }
            rk[i] = a; rv[i] = c;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + NT * i;
            if (idx < 64 * C::NCH) {
                const int r = idx / C::NCH, ch = idx - r * C::NCH;
                *reinterpret_cast<h8*>(Ks + r * C::KS + ch * 16) = rk[i];
                *reinterpret_cast<h8*>(Vs + r * C::VS + ch * 16) = rv[i];
            }
        }
    };

    f16v oacc[C::DV / 32];
#pragma unroll
    for (int i = 0; i < C::DV / 32; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m_run = -1.0e30f, l_run = 0.f;

    const int ntiles = (p.Sk + 63) / 64;
    load_tile(0);
    // per-lane LDS addresses
    const int k_addr = l31 * C::KS + 16 * hh;                                  // + kb*32*KS + ks*32
    const int v_addr = (4 * hh + ((lane & 15) >> 2)) * C::VS + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

    auto tile_body = [&](int t, auto ragged_tag) {
        constexpr bool RAGGED = decltype(ragged_tag)::value;
        __syncthreads();          // everyone done reading the previous tile (and the zero fill)
        store_tile();
        __syncthreads();
        if (t + 1 < ntiles) load_tile(t + 1);

        // ---- S^T = K Q^T ----
        f16v sacc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const f16v zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < C::DK / 16; ++ks) {
                h8 kf = *reinterpret_cast<const h8*>(Ks + k_addr + kb * 32 * C::KS + ks * 32);
                sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], ks == 0 ? zero : sacc[kb], 0, 0, 0);
            }
        }
        // ---- online softmax (per-lane query column) ----
        if (RAGGED) {                  // last, ragged tile only (separate code path: no per-tile select cost)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (key >= p.Sk || (p.causal && key > qrow)) sacc[kb][r] = -1.0e30f;
                }
        }
        float mt = -1.0e30f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mt = fmaxf(mt, sacc[kb][r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt * p.sc);
        // The running max settles after the first few tiles; when no lane's max moved, alpha == 1 exactly and the
        // O rescale (an AGPR read-modify-write of the whole accumulator) is skipped -- wave-uniform branch.
        if (!__all(m_new == m_run)) {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < C::DV / 32; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
            m_run = m_new;
        }
        float psum = 0.f;
        h8 pf[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(sacc[kb][r] * p.sc - m_run);
                if (!C::ONES) psum += pv;
                pf[kb][r >> 3][r & 7] = (half_t)pv;
            }
        if (!C::ONES) l_run += psum;

        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int db = 0; db < C::DV / 32; ++db) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const char* base = Vs + v_addr + (kb * 32 + 16 * s) * C::VS + db * 64;
                    s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(base));
                    s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (s4 __attribute__((address_space(3)))*)(base + 8 * C::VS));
                    h4 lo_h = __builtin_bit_cast(h4, lo), hi_h = __builtin_bit_cast(h4, hi);
                    h8 vf = {lo_h[0], lo_h[1], lo_h[2], lo_h[3], hi_h[0], hi_h[1], hi_h[2], hi_h[3]};
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kb][s], oacc[db], 0, 0, 0);
                }
        }
    };
    const bool ragged = (p.Sk & 63) != 0;
    for (int t = 0; t < ntiles; ++t) {
        // masked code path: the ragged last tile, and (causal) every tile that reaches past this workgroup's first query
        if ((ragged && t == ntiles - 1) || (p.causal && t * 64 + 63 > (int)blockIdx.x * (32 * WAVES))) tile_body(t, std::true_type{});
        else tile_body(t, std::false_type{});
    }

    // ---- epilogue: O[q][d] = O^T[d][q] / l ----
    float l_tot;
    if (C::ONES) l_tot = __shfl(oacc[C::ONES_DB][C::ONES_REG], l31, 64);    // row D of O^T lives in lanes 0-31
    else l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (qrow < p.Sq) {
        half_t* orow = p.O + ((long long)b * p.Sq + qrow) * p.ldo + head * D;
#pragma unroll
        for (int db = 0; db < C::DV / 32; ++db)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int d0 = db * 32 + 8 * rq + 4 * hh;
                if (d0 < D) {
                    h4 o = {(half_t)(oacc[db][4 * rq] * inv), (half_t)(oacc[db][4 * rq + 1] * inv),
                            (half_t)(oacc[db][4 * rq + 2] * inv), (half_t)(oacc[db][4 * rq + 3] * inv)};
                    *reinterpret_cast<h4*>(orow + d0) = o;
                }
            }
    }
}

static int g_attn_waves = 0;      // 0: by grid size; 2 / 4: forced (tests)
extern "C" int lcm_set_attention_waves(int waves) {
    if (waves != 0 && waves != 2 && waves != 4) { lcm_set_error("attention_waves: %d", waves); return LCM_EINVAL; }
    g_attn_waves = waves;
    return LCM_OK;
}

template <int D, int WAVES>
static int launch_attn_w(const AttnParams& p, hipStream_t s) {
    using C = AttnCfg<D>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kernel<D, WAVES>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        attr_set = true;
    }
    dim3 grid((p.Sq + 32 * WAVES - 1) / (32 * WAVES), p.B * p.heads);
    char nm[32];
    snprintf(nm, sizeof(nm), "attn_kernel<%d, %d>", D, WAVES);
    lcm_prof_start(nm, s);
    hipLaunchKernelGGL((attn_kernel<D, WAVES>), grid, dim3(64 * WAVES), C::LDS_BYTES, s, p);
    lcm_prof_stop(s);
    LCM_CHECK_LAUNCH("attention");
    return LCM_OK;
}

template <int D>
static int launch_attn(const AttnParams& p, hipStream_t s) {
    // 128-row workgroups.  Measured at batch 1, S = 4096, d = 40 (256 workgroups of 128 rows, one per CU): 64-row workgroups
    // (512, two per CU) take 104 us against 70 us -- every workgroup re-stages all 64 K/V tiles, and that staging (not the
    // MFMA or the softmax) is what a workgroup's time is made of.  (Double-buffering the K/V tiles in LDS -- one barrier per
    // tile instead of two -- was measured too: -5 % at batch 8, +20 % at batch 1; not kept.)  The 64-row form stays selectable.
    return g_attn_waves == 2 ? launch_attn_w<D, 2>(p, s) : launch_attn_w<D, 4>(p, s);
}

extern "C" int lcm_attention_f16(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* out,
                                 int ldo, int B, int heads, int Sq, int Sk, int d, float scale, int causal, void* stream) {
    LCM_REQUIRE(Q && K && V && out, "attention: null pointer");
    LCM_REQUIRE(B > 0 && heads > 0 && Sq > 0 && Sk > 0, "attention: bad shape");
    LCM_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, "attention: misaligned leading dims");
    AttnParams p = {(const half_t*)Q, (const half_t*)K, (const half_t*)V, (half_t*)out, ldq, ldk, ldv, ldo,
                    B, heads, Sq, Sk, scale * 1.4426950408889634f, causal ? 1 : 0};
    LCM_REQUIRE(!causal || Sq == Sk, "attention: causal mask needs Sq == Sk");
    hipStream_t s = (hipStream_t)stream;
    switch (d) {
        case 40: return launch_attn<40>(p, s);
        case 64: return launch_attn<64>(p, s);
        case 80: return launch_attn<80>(p, s);
        case 160: return launch_attn<160>(p, s);
        default: lcm_set_error("attention: unsupported head_dim %d (40/64/80/160)", d); return LCM_EINVAL;
    }
}
