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
#include <utility>

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
        // P and O^T += V^T P^T per 32-key half: the four MFMAs of half 0 run in the matrix pipe while the VALU computes the
        // exponentials of half 1 (with one wave per SIMD -- batch 1 -- nothing else overlaps the two).  Every accumulator
        // still sees (kb0,s0), (kb0,s1), (kb1,s0), (kb1,s1) in that order: same bits as the unsplit form.
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            h8 pf[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(sacc[kb][r] * p.sc - m_run);
                if (!C::ONES) psum += pv;
                pf[r >> 3][r & 7] = (half_t)pv;
            }
#pragma unroll
            for (int db = 0; db < C::DV / 32; ++db)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const char* base = Vs + v_addr + (kb * 32 + 16 * s) * C::VS + db * 64;
                    s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(base));
                    s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (s4 __attribute__((address_space(3)))*)(base + 8 * C::VS));
                    h4 lo_h = __builtin_bit_cast(h4, lo), hi_h = __builtin_bit_cast(h4, hi);
                    h8 vf = {lo_h[0], lo_h[1], lo_h[2], lo_h[3], hi_h[0], hi_h[1], hi_h[2], hi_h[3]};
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[s], oacc[db], 0, 0, 0);
                }
        }
        if (!C::ONES) l_run += psum;
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
    if (waves != 0 && waves != 2 && waves != 4 && waves != 8) { lcm_set_error("attention_waves: %d", waves); return LCM_EINVAL; }
    g_attn_waves = waves;
    return LCM_OK;
}

template <int D, int WAVES>
static int launch_attn_w(const AttnParams& p, hipStream_t s) {
    using C = AttnCfg<D>;
    static LcmDevOnce attr_once;
    if (auto once_guard = attr_once.first()) {
        once_guard.check(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kernel<D, WAVES>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
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


template <int N>
__device__ __forceinline__ void attn_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int... I, class F>
__device__ __forceinline__ void attn_static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void attn_static_for(F&& f) {
    attn_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}
// Transposed fragment read as inline asm: through the builtin the compiler cannot tell the read from the LDS-DMA writes of the
// next tile in flight (other buffer) and drains the DMA (s_waitcnt vmcnt(0)) in front of it in every tile.  The caller waits
// with attn2_wait_lgkm (which names the destination registers) before the first use.
template <int OFF>
__device__ __forceinline__ s4 attn2_tr(int addr) {
    s4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N>
__device__ __forceinline__ void attn2_wait_lgkm(s4 (&r)[N]) {
    static_assert(N == 8 || N == 12, "4 reads per d-block");
    if constexpr (N == 8)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                     "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]));
}
__device__ __forceinline__ int xcd_remap_attn(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}
// ------------------------------------------------------------------------------------------------
// attn2: the self-attention kernel of the long sequences (non-causal, Sk >= 128, d in {40, 64, 80}: UNet levels 0 / 1 at
// 512^2 and up, SDXL).  Same MFMA formulation as attn_kernel (swapped QK^T on 32x32x16, S^T accumulators are the B operand
// of O^T += V^T P^T, V^T fragments by ds_read_b64_tr_b16) with the work around the MFMAs rebuilt:
//   * K/V tiles (64 keys) arrive by LDS-DMA into a DOUBLE buffer: tile t+1 is in flight during all of tile t, ONE barrier per
//     tile, no staging registers, no ds_write pass.  The LDS image is dense ([key][d], 2d-byte rows: a 1 KiB DMA piece is 64
//     consecutive 16-byte chunks); bank conflicts are handled on the SOURCE side (the lane that fills chunk position p of
//     row r fetches chunk unswz(p, r)): none needed for d = 40 (80-byte rows: 16 consecutive rows hit 16 distinct 16-byte
//     slots), a GF(2) XOR for d = 64, a rotation for d = 80 (tools/lds_bank_model.py).
//   * Q is scaled by scale*log2(e) once, in registers (fp16), so P = exp2(S - m) needs no multiply.
//   * d = 40 (MFOLD): the 8 padding k-slots of the third QK^T step carry the running max: the lanes of those slots read a
//     constant K fragment (1, 1, 0, ...) and hold (hi, lo) = fp16 split of -m in their Q fragment, so the MFMA itself
//     produces S - m and the 32 subtractions per tile disappear from the VALU (the softmax VALU, not the MFMA, bounds this
//     kernel).  The running max only moves when a score exceeds it by more than 2^THR (deferred rescale): the rescale
//     branch (O *= alpha, S -= delta, new (hi, lo)) is rare after the first tiles.
//   * row sums ride in the padding row d of O^T (V^T row d = 1): lanes whose V^T rows are padding read constant cells
//     ((1,0,0,0) or zeros) instead of tile bytes.  The cells are replicated at every immediate offset the reads use, so all
//     fragment reads are `base VGPR + immediate` with loop-invariant bases (the tile loop is unrolled by the two buffers).
//   * 8-wave workgroups (256 query rows) for batched launches halve the K/V staging per query; 4-wave workgroups (128 rows)
//     when that would leave CUs idle.  A query row's arithmetic is the same either way (one wave = 32 rows).
//   * workgroup -> (image*head, query block) through the XCD remap: the query blocks of one head run on one XCD and share its
//     K/V in that XCD's L2.
// ------------------------------------------------------------------------------------------------
template <int D>
struct Attn2Cfg {
    static constexpr int NCH = D / 8, ROWB = D * 2;
    static constexpr int DK = (D + 15) / 16 * 16, NKS = DK / 16;
    static constexpr int DV = (D + 31) / 32 * 32, NDB = DV / 32;
    static constexpr bool MFOLD = (DK - D) >= 8;        // the last k-step's upper half is padding: it carries -m
    static constexpr bool ONES = DV > D;                 // padding row D of O^T accumulates the row sums
    static constexpr int TILE = 64 * ROWB, BUFB = 2 * TILE;
    static constexpr int CELLS = 2 * BUFB;               // constant cells: V cell at CELLS + buf*BUFB + 8i*ROWB (i < 8), K cell 16 B behind
    static constexpr int LDS_BYTES = (MFOLD || ONES) ? CELLS + BUFB + 56 * ROWB + 32 : CELLS;
    static constexpr int ONES_DB = D / 32, ONES_REG = 4 * ((D % 32) >> 3) + ((D % 32) & 3);
    static_assert(!ONES || ((D % 32) % 8 < 4), "row-sum row must land in lanes 0-31");
    // chunk position of chunk c in row r of the dense LDS image (source-side swizzle), and its inverse
    static __device__ __forceinline__ int pos(int c, int r) {
        if constexpr (D == 64) return c ^ (((r >> 1) & 1) | (((r >> 2) & 1) << 1) | ((((r >> 1) ^ (r >> 3)) & 1) << 2));
        else if constexpr (D == 80) { const int v = c + 3 * ((r >> 3) & 1); return v >= NCH ? v - NCH : v; }
        else return c;
    }
    // V tile only, d = 40: LDS row of key r.  The transposed fragment read (ds_read_b64_tr_b16) takes 4 consecutive keys per
    // 32-lane group, 64 bytes of each; with 80-byte rows keys k and k+3 share banks (240 + 64 wraps past 256): 22 % of the
    // LDS cycles of the kernel were bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE).  Keys 4a + b go to row
    // 16 (a / 4) + 4 b + a % 4: the 4 keys of a read sit 320 bytes = 64 mod 256 apart -- four disjoint 64-byte windows.
    static __device__ __forceinline__ int vrow(int r) {
        if constexpr (D == 40) { const int a = r >> 2, b = r & 3; return 16 * (a >> 2) + 4 * b + (a & 3); }
        else return r;
    }
    static __device__ __forceinline__ int vkey(int row) {        // inverse of vrow
        if constexpr (D == 40) { const int q = row >> 4, rem = row & 15; return 4 * (4 * q + (rem & 3)) + (rem >> 2); }
        else return row;
    }
    static __device__ __forceinline__ int unpos(int p, int r) {
        if constexpr (D == 64) return pos(p, r);
        else if constexpr (D == 80) { const int v = p - 3 * ((r >> 3) & 1); return v < 0 ? v + NCH : v; }
        else return p;
    }
};

template <int OFF>
__device__ __forceinline__ h8 attn2_read_b128(int addr) {
    return *reinterpret_cast<const __attribute__((address_space(3))) h8*>((size_t)(unsigned)(addr + OFF));
}

// KS = 2 (sequences of 1024..4096 keys -- a property of the image, so a request computes the same bits at any batch size): the
// keys of a query block are split over TWO groups of WAVES waves, tiles [0, ceil(n/2)) and [ceil(n/2), n); each group streams
// its tiles through its own double buffer and keeps its own (m, l, O^T); at the end group 1 hands its state to group 0 through
// LDS and group 0 merges (O = O1 2^(m1-m) + O2 2^(m2-m), the row sums likewise) and stores.  The dependent chain of a query
// block -- 64 tiles at S = 4096, each a barrier, two MFMA phases and an exp2 phase that nothing overlaps at one wave per SIMD
// (batch 1: 256 workgroups) -- halves, and every SIMD holds two waves whose MFMA and VALU phases interleave.
template <int D, int WAVES, int KS>
// (second launch-bound = waves per SIMD the register budget must allow.  d = 64 with the key split: 4, i.e. <= 128 VGPRs, so that
// two 8-wave workgroups share a CU -- at 142 VGPRs one workgroup per CU left the 320 workgroups of SDXL's 4096-token level in
// two rounds)
__global__ __launch_bounds__(64 * WAVES * KS, (WAVES * KS == 8) ? (D == 64 && KS == 2 ? 4 : 2) : 1) void attn2_kernel(AttnParams p, int nqb) {
    using C = Attn2Cfg<D>;
    constexpr int NP = (2 * C::NCH + WAVES - 1) / WAVES;        // DMA pieces per wave per tile
    constexpr float THR = 8.0f;                                   // deferred rescale: P <= 2^THR (fp16: no precision cost)
    extern __shared__ __attribute__((aligned(16))) char smem_all[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = KS == 1 ? 0 : wave_all / WAVES;               // key group
    const int wave = KS == 1 ? wave_all : wave_all - grp * WAVES; // query sub-block (32 rows) within the workgroup
    char* smem = smem_all + grp * C::LDS_BYTES;                   // the group's buffers and constant cells
    const int gbase = grp * C::LDS_BYTES;
    const int l31 = lane & 31, hh = lane >> 5;
    const int v = xcd_remap_attn(blockIdx.x, gridDim.x);
    const int bh = v / nqb, qb = v - bh * nqb;
    const int b = bh / p.heads, head = bh - b * p.heads;
    const int qrow = qb * (32 * WAVES) + wave * 32 + l31;

    const half_t* Qb = p.Q + (long long)b * p.Sq * p.ldq + head * D;
    const half_t* Kb = p.K + (long long)b * p.Sk * p.ldk + head * D;
    const half_t* Vb = p.V + (long long)b * p.Sk * p.ldv + head * D;

    // ---- constant cells ----
    const int gtid = wave * 64 + lane;                            // thread index within the key group
    if constexpr (C::MFOLD || C::ONES) {
        for (int i = C::CELLS + gtid * 16; i < C::LDS_BYTES; i += 64 * WAVES * 16) *reinterpret_cast<f4*>(smem + i) = (f4){0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        if (gtid < 16) {
            char* cell = smem + C::CELLS + (gtid >> 3) * C::BUFB + (gtid & 7) * 8 * C::ROWB;
            if (C::ONES) *reinterpret_cast<half_t*>(cell) = (half_t)1.0f;                         // (1,0,0,0 | 0,0,0,0)
            if (C::MFOLD && (gtid & 3) == 0) { reinterpret_cast<half_t*>(cell + 16)[0] = (half_t)1.0f;   // (1,1,0,...): kb = (gtid&7)>>2
                                               reinterpret_cast<half_t*>(cell + 16)[1] = (half_t)1.0f; }
        }
    }

    // ---- DMA pieces of this wave: piece i < NCH = chunks [64i, 64i+64) of the K tile, else of the V tile ----
    int pc_off[NP], pc_row[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int i = wave + j * WAVES;
        const int pi = i >= C::NCH ? i - C::NCH : i;
        const int pidx = pi * 64 + lane;
        const int lrow = pidx / C::NCH, ps = pidx - lrow * C::NCH;       // LDS row this lane fills
        const int row = i >= C::NCH ? C::vkey(lrow) : lrow;                 // the key that lives there
        pc_row[j] = row;
        pc_off[j] = row * (i >= C::NCH ? p.ldv : p.ldk) + C::unpos(ps, lrow) * 8;
    }
    const int ntiles = (p.Sk + 63) >> 6;
    const int nhalf = KS == 1 ? ntiles : (ntiles + 1) >> 1;       // barrier iterations of the workgroup (= group 0's tiles)
    const int t0 = grp * nhalf, t1 = KS == 1 ? ntiles : (grp == 0 ? nhalf : ntiles);      // this group's tiles
    auto issue_tile = [&](int t, int buf) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int i = wave + j * WAVES;
            if (i < 2 * C::NCH) {
                const bool isv = i >= C::NCH;
                const int ld = isv ? p.ldv : p.ldk;
                int off = pc_off[j];
                const int last = p.Sk - 1 - t * 64;                       // rows past Sk re-read the last real row (finite; masked)
                if (pc_row[j] > last) off -= (pc_row[j] - last) * ld;
                const half_t* src = (isv ? Vb : Kb) + (long long)t * 64 * ld + off;
                char* dst = smem + buf * C::BUFB + (isv ? C::TILE + (i - C::NCH) * 1024 : i * 1024);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
        }
    };
    if (t0 < t1) issue_tile(t0, 0);

    // ---- Q fragments (B operand), scaled by scale*log2(e): lane holds Q[qrow][16ks + 8hh + j] ----
    h8 qf[C::NKS];
#pragma unroll
    for (int ks = 0; ks < C::NKS; ++ks) {
        const int dc = 16 * ks + 8 * hh;
        h8 x = {0, 0, 0, 0, 0, 0, 0, 0};
        if (qrow < p.Sq && dc < D) x = *reinterpret_cast<const h8*>(Qb + (long long)qrow * p.ldq + dc);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = (half_t)((float)x[j] * p.sc);
        qf[ks] = x;
    }

    // ---- per-lane fragment addresses (loop invariant; immediates select buffer / key block) ----
    int kaddr[C::NKS];
#pragma unroll
    for (int ks = 0; ks < C::NKS; ++ks) {
        const int c = 2 * ks + hh;
        kaddr[ks] = gbase + (c < C::NCH ? l31 * C::ROWB + C::pos(c, l31) * 16 : C::CELLS + 16);
    }
    const int vq = (lane & 15) >> 2, vp = lane & 3, vg1 = (lane >> 4) & 1;
    int vaddr[C::NDB][2];
#pragma unroll
    for (int db = 0; db < C::NDB; ++db)
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) {
            const int d = db * 32 + 16 * vg1 + 4 * vp;
            const int key = 8 * hi + 4 * hh + vq;
            vaddr[db][hi] = gbase + (d < D ? C::TILE + C::vrow(key) * C::ROWB + C::pos(d >> 3, C::vrow(key)) * 16 + (d & 7) * 2
                                          : (d == D ? C::CELLS : C::CELLS + 8) + 8 * hi * C::ROWB);
        }

    f16v oacc[C::NDB];
#pragma unroll
    for (int i = 0; i < C::NDB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m_run = 0.f, l_run = 0.f;

    auto tile = [&](int it, auto buf_tag) {
        constexpr int BUF = decltype(buf_tag)::value;
        const int t = t0 + it;                // it: iteration of the workgroup; t: tile of this key group
        attn_wait_vmcnt<0>();                 // this wave's pieces of tile t have landed
        __builtin_amdgcn_s_barrier();         // everyone's have; everyone is done with the other buffer (tile t-1)
        if (KS > 1 && t >= t1) return;        // group 1 of an odd tile count: one idle iteration (barrier only)
        if (t + 1 < t1) issue_tile(t + 1, BUF ^ 1);

        // ---- S^T = K Q^T (MFOLD: minus the running max) ----
        f16v sacc[2];
        attn_static_for<2>([&](auto kb_) {
            constexpr int kb = decltype(kb_)::value;
            const f16v zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            attn_static_for<C::NKS>([&](auto ks_) {
                constexpr int ks = decltype(ks_)::value;
                const h8 kf = attn2_read_b128<BUF * C::BUFB + kb * 32 * C::ROWB>(kaddr[ks]);
                sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], ks == 0 ? zero : sacc[kb], 0, 0, 0);
            });
        });
        if (t == ntiles - 1 && (p.Sk & 63)) {          // ragged last tile
            int lim = p.Sk - t * 64 - 4 * hh;             // (opaque: keeps the 32 compare masks out of loop-invariant SGPRs)
            asm volatile("" : "+v"(lim));
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kb * 32 + (r & 3) + 8 * (r >> 2) >= lim) sacc[kb][r] = -1.0e30f;
        }
        float mt = sacc[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mt = fmaxf(mt, sacc[kb][r]);
        {   // join the two lane halves of a query column
            // inline asm: with the builtin the compiler folds the max of the two swap results into one of them (only one lane
            // half then sees the other's maximum).  s_nop 1 = the 2 wait states between a VALU write and v_permlane*_swap.
            float ma = mt, mb = mt;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ma), "+v"(mb));
            mt = fmaxf(ma, mb);
        }
        const float grow = C::MFOLD ? mt : mt - m_run;          // how far this tile's max exceeds the running one
        float sub = C::MFOLD ? 0.f : m_run;
        if (it == 0 || __any(grow > THR)) {
            const float delta = it == 0 ? grow : fmaxf(grow, 0.f);
            if (it != 0) {
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < C::NDB; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
            }
            m_run += delta;
            if constexpr (C::MFOLD) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc[kb][r] -= delta;
                const half_t mh = (half_t)(-m_run);
                const half_t ml = (half_t)(-m_run - (float)mh);
                if (hh) { qf[C::NKS - 1][0] = mh; qf[C::NKS - 1][1] = ml; }
            } else {
                sub = m_run;
            }
        }
        // ---- P and O^T += V^T P^T per 32-key half ----
        float psum = 0.f;
        attn_static_for<2>([&](auto kb_) {
            constexpr int kb = decltype(kb_)::value;
            // the V^T fragments of this half are fetched while the VALU computes its exponentials
            s4 vr[4 * C::NDB];
            attn_static_for<C::NDB>([&](auto db_) {
                constexpr int db = decltype(db_)::value;
                attn_static_for<2>([&](auto s_) {
                    constexpr int s = decltype(s_)::value;
                    constexpr int OFF = BUF * C::BUFB + (kb * 32 + 16 * s) * C::ROWB;
                    vr[4 * db + 2 * s] = attn2_tr<OFF>(vaddr[db][0]);
                    vr[4 * db + 2 * s + 1] = attn2_tr<OFF>(vaddr[db][1]);
                });
            });
            __builtin_amdgcn_sched_barrier(0);
            h8 pf[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(C::MFOLD ? sacc[kb][r] : sacc[kb][r] - sub);
                if (!C::ONES) psum += pv;
                pf[r >> 3][r & 7] = (half_t)pv;
            }
            __builtin_amdgcn_sched_barrier(0);
            attn2_wait_lgkm(vr);
            attn_static_for<C::NDB>([&](auto db_) {
                constexpr int db = decltype(db_)::value;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const h4 lo_h = __builtin_bit_cast(h4, vr[4 * db + 2 * s]), hi_h = __builtin_bit_cast(h4, vr[4 * db + 2 * s + 1]);
                    const h8 vf = {lo_h[0], lo_h[1], lo_h[2], lo_h[3], hi_h[0], hi_h[1], hi_h[2], hi_h[3]};
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[s], oacc[db], 0, 0, 0);
                }
            });
        });
        if (!C::ONES) l_run += psum;
    };
    for (int it = 0; it < nhalf; it += 2) {
        tile(it, std::integral_constant<int, 0>{});
        if (it + 1 < nhalf) tile(it + 1, std::integral_constant<int, 1>{});
    }

    if constexpr (KS > 1) {
        // ---- merge the two key groups: group 1 -> LDS ([wave][register][lane] floats, over the tile buffers) -> group 0 ----
        constexpr int NR = C::NDB * 16 + 2;
        __syncthreads();                      // every wave is past its last fragment reads
        float* xch = reinterpret_cast<float*>(smem_all) + wave * NR * 64 + lane;
        if (grp == 1) {
#pragma unroll
            for (int i = 0; i < C::NDB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) xch[(i * 16 + r) * 64] = oacc[i][r];
            xch[(C::NDB * 16) * 64] = t0 < t1 ? m_run : -1.0e30f;
            xch[(C::NDB * 16 + 1) * 64] = l_run;
        }
        __syncthreads();
        if (grp == 1) return;
        const float m2 = xch[(C::NDB * 16) * 64], l2 = xch[(C::NDB * 16 + 1) * 64];
        const float m = fmaxf(m_run, m2);
        const float a1 = __builtin_amdgcn_exp2f(m_run - m), a2 = __builtin_amdgcn_exp2f(m2 - m);
        l_run = l_run * a1 + l2 * a2;
#pragma unroll
        for (int i = 0; i < C::NDB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[i][r] = oacc[i][r] * a1 + xch[(i * 16 + r) * 64] * a2;
    }

    // ---- epilogue: O[q][d] = O^T[d][q] / l ----
    float l_tot;
    if (C::ONES) l_tot = __shfl(oacc[C::ONES_DB][C::ONES_REG], l31, 64);
    else l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (qrow < p.Sq) {
        half_t* orow = p.O + ((long long)b * p.Sq + qrow) * p.ldo + head * D;
#pragma unroll
        for (int db = 0; db < C::NDB; ++db)
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

static int g_attn2 = 1;           // 0: long self-attention on attn_kernel as before (A/B switch)
extern "C" int lcm_set_attention_impl(int impl) {
    if (impl != 0 && impl != 1) { lcm_set_error("attention_impl: %d", impl); return LCM_EINVAL; }
    g_attn2 = impl;
    return LCM_OK;
}

template <int D, int WAVES, int KS = 1>
static int launch_attn2_w(const AttnParams& p, hipStream_t s) {
    using C = Attn2Cfg<D>;
    static_assert(KS * C::LDS_BYTES <= 160 * 1024 && (KS == 1 || 4 * WAVES * (C::NDB * 16 + 2) * 64 <= KS * C::LDS_BYTES), "attn2 LDS");
    static LcmDevOnce attr_once;
    if (auto once_guard = attr_once.first()) {
        once_guard.check(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn2_kernel<D, WAVES, KS>), hipFuncAttributeMaxDynamicSharedMemorySize, KS * C::LDS_BYTES));
    }
    const int nqb = (p.Sq + 32 * WAVES - 1) / (32 * WAVES);
    char nm[40];
    if (KS == 1) snprintf(nm, sizeof(nm), "attn2_kernel<%d, %d>", D, WAVES);
    else snprintf(nm, sizeof(nm), "attn2_kernel<%d, %d, %d>", D, WAVES, KS);
    lcm_prof_start(nm, s);
    hipLaunchKernelGGL((attn2_kernel<D, WAVES, KS>), dim3(nqb * p.B * p.heads), dim3(64 * WAVES * KS), KS * C::LDS_BYTES, s, p, nqb);
    lcm_prof_stop(s);
    LCM_CHECK_LAUNCH("attention2");
    return LCM_OK;
}

static int g_attn2_ksplit = 1;    // 0: never split the keys of a query block (A/B switch; changes the bits of the >= 1024-key levels)
extern "C" int lcm_set_attention_ksplit(int on) { g_attn2_ksplit = on ? 1 : 0; return LCM_OK; }

template <int D>
static int launch_attn2(const AttnParams& p, hipStream_t s) {
    // 256-row workgroups when they still give every CU two (the K/V tiles are staged once per workgroup), else 128-row ones
    // 1024..4096 keys: the key-split form (two groups of 4 waves over 128 query rows), whatever the batch -- the split changes the
    // summation order, so it is keyed on the sequence length alone
    const long long wg8 = (long long)((p.Sq + 255) / 256) * p.B * p.heads;
    if (g_attn2_ksplit && p.Sk >= 1024 && p.Sk <= 4096) {      // beyond 4096 keys a query block's chain is long enough to amortise itself: 9216 keys measured 2.5-6 % slower split
        // d = 40 fits 128 VGPRs: batched launches take 256 query rows x 2 key groups (16 waves, K/V staged once per 256 rows)
        if constexpr (D == 40) if (g_attn_waves == 8 || (g_attn_waves == 0 && wg8 >= 512)) return launch_attn2_w<D, 8, 2>(p, s);
        return launch_attn2_w<D, 4, 2>(p, s);
    }
    const bool w8 = g_attn_waves == 8 || (g_attn_waves == 0 && wg8 >= 512);
    return w8 ? launch_attn2_w<D, 8>(p, s) : launch_attn2_w<D, 4>(p, s);
}

// ------------------------------------------------------------------------------------------------
// Wide-head flash attention (head_dim 512): the AutoencoderKL mid-block attention (one head, d = 512, S = h*w).
// The 32x32 kernel above keeps O for all of d in registers (d <= 160); here a wave owns 16 query rows and uses
// v_mfma_f32_16x16x32_f16, so O^T (d x 16 queries) is d/16 accumulator quads = 128 VGPRs and Q (16 x d) 64.
//   * S^T = K Q^T per 16-key block: A = K rows from LDS (ds_read_b128), B = Q rows held in registers; a lane owns one query
//     column (lane & 15) and keys 4fq..4fq+3 of each block, so the online-softmax state is a per-lane scalar (max joined
//     across the 4 lane groups of a query by two shuffles).
//   * P^T never leaves registers: the accumulators of two key blocks, converted to fp16, ARE the B operand (32 key slots x
//     16 queries) of O^T += V^T P^T; slot (fq, j) <-> key 4fq + j (j < 4) / 16 + 4fq + j - 4.
//   * V stays row-major in LDS ([key][d]); the V^T A-fragment with exactly that slot order is two ds_read_b64_tr_b16
//     (4 keys x 16 d each, hardware transpose).
//   * K/V tiles (32 keys: 32 KB each) arrive by LDS-DMA (global_load_lds_dwordx4: a 1 KB row = one wave instruction, so rows
//     can carry their conflict-free padding) into a DOUBLE buffer: tile t+1 is in flight during all of tile t, one barrier
//     per tile, no staging registers and no ds_write pass (ds_write_b128 moves ~79 B/clk/CU: 830 cycles per tile here).
//   * row stride 2d + 32 B for both: 16 * (2 mod 16) makes ds_read_b128 of the 16x16x32 A-fragment pattern (lane groups
//     {0-3,12-15,20-27}, ...) conflict-free, 8 * (4 mod 32) the transposed reads.
//   * with one wave per SIMD nothing hides LDS latency but the code itself: fragment reads run 8 ahead of their MFMAs.
// No S x S matrix in memory (the GEMM -> softmax -> transpose -> GEMM form needs B * S^2 * 2 bytes: 512 MB per image at
// SDXL's 128^2 latent).  One workgroup = 64 query rows of one (image, head); 132 KB of LDS.
// ------------------------------------------------------------------------------------------------
template <int OFF>
__device__ __forceinline__ s4 attn_tr_read(unsigned lds_addr) {
    s4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF));
    return v;
}

template <int D, int TK>
__global__ __launch_bounds__(256, 1) void attn_wide_kernel(AttnParams p) {
    static_assert(D == 512 && TK == 32, "one LDS-DMA instruction per 1 KB row; 8 rows of K and of V per wave per tile");
    constexpr int RS = D * 2 + 32;                      // row stride of K and V tiles
    constexpr int NKS = D / 32, NDB = D / 16, NKB = TK / 16;
    constexpr int TILE = TK * RS, BUF = 2 * TILE;       // K tile | V tile
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    const int bh = blockIdx.y, b = bh / p.heads, head = bh - b * p.heads;
    const int qrow = blockIdx.x * 64 + wave * 16 + frow;
    const half_t* Qb = p.Q + (long long)b * p.Sq * p.ldq + head * D;
    const half_t* Kb = p.K + (long long)b * p.Sk * p.ldk + head * D + lane * 8;
    const half_t* Vb = p.V + (long long)b * p.Sk * p.ldv + head * D + lane * 8;

    // rows past Sk re-read the last real row (finite values; their scores are masked and P = 0)
    auto issue_tile = [&](int t, int buf) {
        char* ks = smem + buf * BUF + wave * (TK / 4) * RS;
#pragma unroll
        for (int i = 0; i < TK / 4; ++i) {
            const int key = t * TK + wave * (TK / 4) + i;
            const int kc = key < p.Sk ? key : p.Sk - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Kb + (long long)kc * p.ldk),
                                             (__attribute__((address_space(3))) void*)(ks + i * RS), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Vb + (long long)kc * p.ldv),
                                             (__attribute__((address_space(3))) void*)(ks + TILE + i * RS), 16, 0, 0);
        }
    };
    const int ntiles = (p.Sk + TK - 1) / TK;
    issue_tile(0, 0);

    h8 qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (qrow < p.Sq) v = *reinterpret_cast<const h8*>(Qb + (long long)qrow * p.ldq + 32 * ks + 8 * fq);
        qf[ks] = v;
    }
    f4 oacc[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i) oacc[i] = (f4){0.f, 0.f, 0.f, 0.f};
    float m_run = -1.0e30f, l_run = 0.f;

    const int k_addr = frow * RS + fq * 16;                                    // + kb*16*RS + ks*64
    const int v_addr = TILE + (4 * fq + (frow >> 2)) * RS + (4 * (frow & 3)) * 2;   // + kb*16*RS + db*32
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        attn_wait_vmcnt<0>();                 // this wave's rows of tile t have landed (tile t+1 is not issued yet)
        __builtin_amdgcn_s_barrier();         // everyone's have; everyone is done reading the other buffer (tile t-1)
        if (t + 1 < ntiles) issue_tile(t + 1, buf ^ 1);
        const char* Ks = smem + buf * BUF;

        // S^T = K Q^T: the 8 K fragments of group g+1 are in flight while group g's MFMAs run
        f4 sacc[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) sacc[kb] = (f4){0.f, 0.f, 0.f, 0.f};
        {
            constexpr int G = 8 / NKB, NG = NKS / G;
            h8 kfr[2][8];
            auto rd = [&](int g, h8* dst) {
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int j = 0; j < G; ++j)
                        dst[kb * G + j] = *reinterpret_cast<const h8*>(Ks + k_addr + kb * 16 * RS + (g * G + j) * 64);
            };
            rd(0, kfr[0]);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) rd(g + 1, kfr[(g + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < G; ++j)
#pragma unroll
                    for (int kb = 0; kb < NKB; ++kb)
                        sacc[kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kfr[g & 1][kb * G + j], qf[g * G + j], sacc[kb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (t == ntiles - 1 && (p.Sk % TK)) {           // ragged last tile
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (t * TK + kb * 16 + 4 * fq + j >= p.Sk) sacc[kb][j] = -1.0e30f;
        }
        float mt = -1.0e30f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int j = 0; j < 4; ++j) mt = fmaxf(mt, sacc[kb][j]);
        mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt * p.sc);
        if (!__all(m_new == m_run)) {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < NDB; ++i) { oacc[i][0] *= alpha; oacc[i][1] *= alpha; oacc[i][2] *= alpha; oacc[i][3] *= alpha; }
            m_run = m_new;
        }
        h8 pf[NKB / 2];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float pv = __builtin_amdgcn_exp2f(sacc[kb][j] * p.sc - m_run);
                l_run += pv;
                pf[kb >> 1][(kb & 1) * 4 + j] = (half_t)pv;
            }
        // O^T += V^T P^T: 8 transposed reads (4 d-blocks) ahead of the MFMAs that consume them.  The reads and their
        // counted waits are inline asm: through the builtin the compiler cannot tell these reads from the LDS-DMA writes of
        // tile t+1 in flight (other buffer) and drains the DMA (s_waitcnt vmcnt(0)) in the middle of every tile.
        {
            static_assert(NKB == 2, "one key pair per tile");
            constexpr int GD = 4, NG = NDB / GD;
            s4 vfr[2][2 * GD];
            const unsigned vbase = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(Ks + v_addr);
            attn_static_for<GD>([&](auto j) {
                constexpr int J = decltype(j)::value;
                vfr[0][2 * J] = attn_tr_read<J * 32>(vbase);
                vfr[0][2 * J + 1] = attn_tr_read<J * 32 + 16 * RS>(vbase);
            });
            attn_static_for<NG>([&](auto g) {
                constexpr int Gi = decltype(g)::value;
                s4* cur = vfr[Gi & 1];
                if constexpr (Gi + 1 < NG) {
                    s4* nxt = vfr[(Gi + 1) & 1];
                    attn_static_for<GD>([&](auto j) {
                        constexpr int J = decltype(j)::value;
                        nxt[2 * J] = attn_tr_read<((Gi + 1) * GD + J) * 32>(vbase);
                        nxt[2 * J + 1] = attn_tr_read<((Gi + 1) * GD + J) * 32 + 16 * RS>(vbase);
                    });
                    asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]),
                                 "+v"(cur[4]), "+v"(cur[5]), "+v"(cur[6]), "+v"(cur[7]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]),
                                 "+v"(cur[4]), "+v"(cur[5]), "+v"(cur[6]), "+v"(cur[7]));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < GD; ++j) {
                    const h4 lo_h = __builtin_bit_cast(h4, cur[2 * j]), hi_h = __builtin_bit_cast(h4, cur[2 * j + 1]);
                    const h8 vf = {lo_h[0], lo_h[1], lo_h[2], lo_h[3], hi_h[0], hi_h[1], hi_h[2], hi_h[3]};
                    oacc[Gi * GD + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[0], oacc[Gi * GD + j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        }
    }
    float l_tot = l_run + __shfl_xor(l_run, 16, 64);
    l_tot += __shfl_xor(l_tot, 32, 64);
    const float inv = 1.0f / l_tot;
    if (qrow < p.Sq) {
        half_t* orow = p.O + ((long long)b * p.Sq + qrow) * p.ldo + head * D;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            h4 o = {(half_t)(oacc[db][0] * inv), (half_t)(oacc[db][1] * inv), (half_t)(oacc[db][2] * inv), (half_t)(oacc[db][3] * inv)};
            *reinterpret_cast<h4*>(orow + db * 16 + 4 * fq) = o;
        }
    }
}

template <int D, int TK>
static int launch_attn_wide(const AttnParams& p, hipStream_t s) {
    constexpr int LDS = 2 * 2 * TK * (D * 2 + 32);
    static LcmDevOnce attr_once;
    if (auto once_guard = attr_once.first()) {
        once_guard.check(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_wide_kernel<D, TK>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    }
    dim3 grid((p.Sq + 63) / 64, p.B * p.heads);
    char nm[32];
    snprintf(nm, sizeof(nm), "attn_wide_kernel<%d,%d>", D, TK);
    lcm_prof_start(nm, s);
    hipLaunchKernelGGL((attn_wide_kernel<D, TK>), grid, dim3(256), LDS, s, p);
    lcm_prof_stop(s);
    LCM_CHECK_LAUNCH("attention_wide");
    return LCM_OK;
}

extern "C" int lcm_attention_f16(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* out,
                                 int ldo, int B, int heads, int Sq, int Sk, int d, float scale, int causal, void* stream) {
    LCM_REQUIRE(Q && K && V && out, "attention: null pointer");
    LCM_REQUIRE(B > 0 && heads > 0 && Sq > 0 && Sk > 0, "attention: bad shape");
    LCM_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, "attention: misaligned leading dims");
    AttnParams p = {(const half_t*)Q, (const half_t*)K, (const half_t*)V, (half_t*)out, ldq, ldk, ldv, ldo,
                    B, heads, Sq, Sk, scale > 0.f ? scale * 1.4426950408889634f : 1.0f, causal ? 1 : 0};   // scale <= 0: Q arrives pre-scaled
    LCM_REQUIRE(!causal || Sq == Sk, "attention: causal mask needs Sq == Sk");
    hipStream_t s = (hipStream_t)stream;
    if (g_attn2 && !causal && Sk >= 128 && g_attn_waves != 2) {       // long (self-attention) sequences: the streaming kernel
        switch (d) {
            case 40: return launch_attn2<40>(p, s);
            case 64: return launch_attn2<64>(p, s);
            case 80: return launch_attn2<80>(p, s);
            default: break;
        }
    }
    switch (d) {
        case 40: return launch_attn<40>(p, s);
        case 64: return launch_attn<64>(p, s);
        case 80: return launch_attn<80>(p, s);
        case 160: return launch_attn<160>(p, s);
        case 512: if (!causal) return launch_attn_wide<512, 32>(p, s);
        default: lcm_set_error("attention: unsupported head_dim %d (40/64/80/160; 512 without causal mask)", d); return LCM_EINVAL;
    }
}
