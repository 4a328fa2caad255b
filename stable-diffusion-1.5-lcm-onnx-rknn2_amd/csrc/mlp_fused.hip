// FeedForward of a BasicTransformerBlock as ONE kernel (gfx950):
//     out = h + Linear_2( GEGLU( Linear_1( LayerNorm(h) ) ) )          diffusers FeedForward(activation_fn="geglu") under
//     norm3 + residual (reached from the reference's self.pipe(...), backends/cuda_worker.py:221-229)
// i.e. lcm_gemm_ln_f16(epilogue GEGLU) followed by lcm_gemm_f16(bias, residual) without the [M, 4C] intermediate ever
// leaving the CU: 84 MB written and read back per block at batch 8 on the 64x64 level (C = 320), with both launches
// bound by that traffic and by the VALU of the GELU.
//
// Structure.  One workgroup = 128 rows of h = 8 waves as 4 (row blocks of 32) x 2 (column halves), two waves per SIMD so that
// one wave's GELU (VALU) and fragment reads run under the other's MFMAs (a first form with 4 waves, one per SIMD and ~500
// registers each, ran every phase exposed: 145 us, no faster than the two launches).  A wave keeps its 32 rows of h as MFMA
// B-operand fragments in registers for the whole kernel (80 VGPRs at C = 320).  The 4C intermediate is walked in chunks of 64
// columns (= 128 rows of the GEGLU-interleaved W1); wave (wm, wn) per chunk:
//     5 K-steps  acc1[64 packed cols: value/gate pairs 2wn, 2wn+1] += W1 tile (LDS) x h fragments (registers)     8 MFMAs each, 32 deep
//     epilogue   LayerNorm affine, x * gelu(gate), fp16: the 8 values a lane holds ARE the B-operand fragment of k-step wn of
//                the second product ("an accumulator tile as the next MFMA's operand"); the partner wave (wm, 1-wn) holds
//                k-step 1-wn: the two swap fragments through 16 KB of LDS (lane-linear 16-byte cells, no arithmetic)
//     1 step     acc2[rows wm][columns 160 wn .. +160] += W2 tile [C rows x 64 k] (LDS) x both k-steps                 40 MFMAs
// W1 / W2 tiles stream through LDS by LDS-DMA (global_load_lds_dwordx4; swizzle on the source side) four first-product steps
// ahead (a DMA takes ~1.1 us from issue to landing under load), behind ONE raw s_barrier per step and a counted s_waitcnt
// vmcnt; 5 W1 slots of 16 KB + 1 W2 slot of 40 KB + the LayerNorm constants of all 8C packed rows (20 KB) + the fragment
// exchange (16 KB) = 156 KB of LDS, one workgroup per CU.
//
// Bit-identity with the two-launch form (tests/test_ops_gpu.py::test_fused_mlp_is_bit_identical_to_two_launches): the first
// product walks K in the same order with the same MFMA and the same row statistics (ln_accum / ln_finish on the same
// fragments), the GEGLU element math is the one function the GEMM epilogue uses (geglu_ln_quad), the second product walks
// its K = 4C in order, one k-tile per chunk, and -- what makes the in-register hand-over exact -- the two-launch form stores
// the intermediate's columns in the order this kernel holds them (igemm_epilogue, GEGLU branch; W2's columns are packed to
// match, packing.pack_ff2_cols): every MFMA of either form multiplies the same values in the same k slots.  Only layers whose
// canonical K partition of the second product has ONE part take this kernel (the host refuses the others).
#include "igemm_common.h"
#include <stdlib.h>

struct MlpParams {
    IgemmParams g;          // the second product's epilogue view: out / ldo / bias / res / ldr / M / N (splits 1, no statistics)
    const half_t* X;        // [M][ldx] rows of h (LayerNorm input; also the residual through g.res)
    int ldx;
    const half_t* W1;       // [8C][C]  gamma (*) W, GEGLU rows interleaved in blocks of 16
    const float* ln_g;      // [8C] fp32
    const float* ln_c;      // [8C] fp32
    float eps;
    const half_t* W2;       // [C][4C], columns in operand order
};

template <int N>
__device__ __forceinline__ void mlp_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ABL (diagnostic builds only, tools/mlp_fused_ab.py): 1 = no GELU (plain product of value and gate), 2 = no LDS-DMA after the
// prologue (stale tiles), 3 = no MFMA of the first product, 4 = no MFMA of the second product.  0 = the product kernel.
template <int C, int ABL = 0>
__global__ __launch_bounds__(512, 2) void mlp_geglu_kernel(MlpParams p) {
    static_assert(C % 64 == 0 && C <= 320, "register budget: C/16 accumulator tiles + C/32 x 2 operand fragments per lane");
    constexpr int KT1 = C / 64;            // K-steps of the first product
    constexpr int F = 4 * C;               // intermediate width
    constexpr int NCH = F / 64;            // chunks of 64 intermediate columns
    constexpr int TN2 = C / 32;            // accumulator tiles of the second product along n, per wave (half of the C / 16)
    constexpr int W1S = 128 * 128;         // bytes of a W1 tile: 128 packed rows x 64 k
    constexpr int W2S = C * 128;           // bytes of a W2 tile: C rows x 64 k
    constexpr int D1 = 4, R1 = D1 + 1;     // a W1 tile is issued D1 first-product steps ahead of its use; ring slots (== KT1: slot = kt)
    constexpr int W2_OFF = R1 * W1S, LN_OFF = W2_OFF + W2S, G_OFF = LN_OFF + 2 * 8 * C * 4;   // ONE W2 slot (tile j issued as step (j, 0) begins)
    constexpr int LPT1 = 2, LPT2 = C / 64; // LDS-DMA instructions per thread of a W1 / W2 tile (512 threads)
    static_assert(KT1 == 5 && D1 == 4, "the counted waits below are written out for 5 K-steps per chunk and 4 steps of lead");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int frow = lane & 15, fq = lane >> 4;
    const int m_base = blockIdx.x * 128;
    const int pos = tid & 7, row0 = tid >> 3;           // row0 0..63
    const int schunk = (pos ^ (row0 & 7)) * 8;          // source chunk (halves) of this lane: the XOR swizzle of the image

    // LDS-DMA sources as (wave-uniform pointer) + (32-bit per-lane offset): the uniform part lives in SGPRs, one VGPR per lane
    const unsigned l1 = (unsigned)(row0 * C + schunk) * 2u, l2 = (unsigned)(row0 * F + schunk) * 2u;
    auto issue_w1 = [&](int j, int kt, int slot) {       // W1 tile (chunk j, K-step kt) into ring slot `slot` (= kt: R1 == KT1)
        const char* ub = reinterpret_cast<const char*>(p.W1 + (long long)(128 * j) * C + 64 * kt);
        char* dst = smem + slot * W1S + wave * 8 * 128;
#pragma unroll
        for (int i = 0; i < LPT1; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ub + (long long)(64 * i) * C * 2 + l1),
                                             (__attribute__((address_space(3))) void*)(dst + i * 64 * 128), 16, 0, 0);
    };
    auto issue_w2 = [&](int j) {
        const char* ub = reinterpret_cast<const char*>(p.W2 + 64 * j);
        char* dst = smem + W2_OFF + wave * 8 * 128;
#pragma unroll
        for (int i = 0; i < LPT2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ub + (long long)(64 * i) * F * 2 + l2),
                                             (__attribute__((address_space(3))) void*)(dst + i * 64 * 128), 16, 0, 0);
    };

    // the first tiles fly while the rows of h are fetched (issue order = use order)
#pragma unroll
    for (int m = 0; m < D1; ++m) issue_w1(0, m, m);

    // ---- rows of h as B-operand fragments: lane (frow, fq) holds row 32 wm + 16 b + frow, columns 32 kk + 8 fq .. + 7 ----
    h8 xa[2][C / 32];
    int m_of[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int m = m_base + 32 * wm + 16 * b + frow;
        m_of[b] = m < p.g.M ? m : -1;
#pragma unroll
        for (int kk = 0; kk < C / 32; ++kk) {
            h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (m < p.g.M) v = *reinterpret_cast<const h8*>(p.X + (long long)m * p.ldx + 32 * kk + 8 * fq);
            xa[b][kk] = v;
        }
    }
    {   // LayerNorm constants of every packed row, once
        float* lg = reinterpret_cast<float*>(smem + LN_OFF);
        for (int i = tid; i < 8 * C; i += 512) { lg[i] = p.ln_g[i]; lg[8 * C + i] = p.ln_c[i]; }
    }
    // row statistics exactly as the LayerNorm-folded GEMM accumulates them: k ascending, per fragment, then the fq butterfly
    float ln_mu[2] = {0.f, 0.f}, ln_r[2] = {0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < C / 32; ++kk)
#pragma unroll
        for (int b = 0; b < 2; ++b) ln_accum(xa[b][kk], ln_mu[b], ln_r[b]);
    ln_finish<2>(ln_mu, ln_r, C, p.eps);
    __syncthreads();
    const __attribute__((address_space(3))) float* ln_lds = (const __attribute__((address_space(3))) float*)(smem + LN_OFF);
    const char* gcell = smem + G_OFF + (wm * 4) * 1024 + lane * 16;                // [wm][k-step][b][lane] 16-byte cells
    char* gmine = smem + G_OFF + ((wm * 2 + wn) * 2) * 1024 + lane * 16;           // this wave writes k-step wn

    f4 acc2[TN2][2];
#pragma unroll
    for (int a = 0; a < TN2; ++a) { acc2[a][0] = (f4){0.f, 0.f, 0.f, 0.f}; acc2[a][1] = (f4){0.f, 0.f, 0.f, 0.f}; }

    // fragment addressing: row a*16 + frow of a tile, 16-byte chunk (kk*4 + fq) ^ (row & 7); (a*16 + frow) & 7 == frow & 7
    const int foff0 = frow * 128 + (((0 + fq) ^ (frow & 7)) << 4), foff1 = frow * 128 + (((4 + fq) ^ (frow & 7)) << 4);
    auto frag = [&](const char* tile, int a, int kk) -> h8 {
        return *reinterpret_cast<const h8*>(tile + a * 2048 + (kk ? foff1 : foff0));
    };
    auto w1tile = [&](int kt) -> const char* { return smem + kt * W1S + (64 * wn) * 128; };      // R1 == KT1: tile (j, kt) sits in slot kt
    const char* w2s = smem + W2_OFF + (TN2 * 16 * wn) * 128;

    // Issue order of the LDS-DMA stream: W1 #0..#3 (above); then, as step (j, kt) begins: [kt == 0: W2(j)] W1 #(5j + kt + 4).
    // Every step's barrier publishes the tile of the NEXT step too, so the first fragments of a step are fetched under the
    // last MFMAs of the step before (no LDS latency exposed behind a barrier).  A tile has landed once all but the DMAs issued
    // AFTER it are done; at the top of step (j, kt) that is W1 #(5j + kt + 1): newer are #(+2), #(+3) = 4 per thread, plus
    // the 5 of W2(j) when it was issued in between (kt = 1, 2).  In the last chunk only step 0 still issues.
    // The instruction order inside a step is pinned (sched_group_barrier): left alone hipcc reads ONE fragment, waits
    // lgkmcnt(0) and issues its two MFMAs -- the LDS latency exposed 8 times a step.  Pinned: every MFMA pair is followed by the
    // read of a fragment needed 4 pairs later, so 4 fragments (16 VGPRs) are in flight and a step never waits for the LDS --
    // the first four fragments of a step are read during the step before (its barrier has published this step's tile).
    // sched_barrier(0) fences each step: nothing may drift across a barrier (a read sinking below the NEXT barrier would
    // race with the DMA that refills its slot).
    mlp_wait_vmcnt<2 * LPT1>();
    __builtin_amdgcn_s_barrier();
    h8 cur[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cur[i] = frag(w1tile(0), i, 0);
    for (int j = 0; j < NCH; ++j) {
        const bool last = j + 1 == NCH;
        f4 acc1[4][2];
#pragma unroll
        for (int a = 0; a < 4; ++a) { acc1[a][0] = (f4){0.f, 0.f, 0.f, 0.f}; acc1[a][1] = (f4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int kt = 0; kt < KT1; ++kt) {
            __builtin_amdgcn_sched_barrier(0);
            if (kt == 0 || kt == 3) mlp_wait_vmcnt<2 * LPT1>();
            else if (kt == 4) {}                   // nothing new is needed: W1 #(5j+4) and W2(j) landed for the steps before
            else if (last) mlp_wait_vmcnt<0>();
            else mlp_wait_vmcnt<2 * LPT1 + LPT2>();
            if (last && kt == 3) mlp_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();          // publishes tile (j, kt+1) | W2(j); every wave is done with the step before
            __builtin_amdgcn_sched_barrier(0);
            if (ABL != 2) {
            if (kt == 0) issue_w2(j);              // the one W2 slot: its last reader was step (j-1, 5)
            if (kt == 0) issue_w1(j, D1, D1);                      // W1 #(5j + kt + 4) = tile (j, 4) | (j+1, kt-1), slot (kt + 4) % 5
            else if (!last) issue_w1(j + 1, kt - 1, kt - 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            const char* ws = w1tile(kt);
            h8 nxt[4], hi[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) hi[i] = frag(ws, i, 1);                         // k 32..63 of this tile
            if (kt + 1 < KT1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) nxt[i] = frag(w1tile(kt + 1), i, 0);      // k 0..31 of the next step's tile
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (ABL == 3) { acc1[i][0][0] += (float)cur[i][0]; acc1[i][1][1] += (float)cur[i][1]; continue; }
                acc1[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[i], xa[0][2 * kt], acc1[i][0], 0, 0, 0);
                acc1[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[i], xa[1][2 * kt], acc1[i][1], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (ABL == 3) { acc1[i][0][0] += (float)hi[i][0]; acc1[i][1][1] += (float)hi[i][1]; continue; }
                acc1[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi[i], xa[0][2 * kt + 1], acc1[i][0], 0, 0, 0);
                acc1[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi[i], xa[1][2 * kt + 1], acc1[i][1], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < (kt + 1 < KT1 ? 8 : 4); ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            if (kt + 1 == KT1) __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            if (kt + 1 < KT1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) cur[i] = nxt[i];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- GEGLU: value tile 2qq, gate tile 2qq+1 (pairs 2wn + qq of the chunk) -> this wave's k-step (wn) of the second product ----
        h8 gf[2];
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const int n = 128 * j + 64 * wn + 32 * qq + 4 * fq;          // packed row of the value
            const f4 gx = *reinterpret_cast<const __attribute__((address_space(3))) f4*>(ln_lds + n);
            const f4 cx = *reinterpret_cast<const __attribute__((address_space(3))) f4*>(ln_lds + 8 * C + n);
            const f4 gg = *reinterpret_cast<const __attribute__((address_space(3))) f4*>(ln_lds + n + 16);
            const f4 cg = *reinterpret_cast<const __attribute__((address_space(3))) f4*>(ln_lds + 8 * C + n + 16);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                h4 o;
                if (ABL == 1) { for (int e = 0; e < 4; ++e) o[e] = (half_t)(acc1[2 * qq][b][e] * acc1[2 * qq + 1][b][e] + gx[e] + cg[e]); }
                else o = geglu_ln_quad(acc1[2 * qq][b], acc1[2 * qq + 1][b], ln_r[b], ln_mu[b], gx, cx, gg, cg);
#pragma unroll
                for (int e = 0; e < 4; ++e) gf[b][4 * qq + e] = o[e];
            }
        }
        *reinterpret_cast<h8*>(gmine) = gf[0];
        *reinterpret_cast<h8*>(gmine + 1024) = gf[1];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) cur[i] = frag(w2s, i, 0);        // W2(j) landed and was published steps ago
        // ---- second product: this chunk is k-tile j of it.  The barrier publishes the fragments just written and W1 #(5j+5), the
        //      tile of step (j+1, 0): newer than it are #(5j+6 .. 5j+8) = 6 per thread ----
        if (last) mlp_wait_vmcnt<0>(); else mlp_wait_vmcnt<3 * LPT1>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // both k-steps' row fragments come back from the exchange cells (the own pair too: no register held across the barrier);
        // k-step 0 first, then 1 (the order the two-launch form accumulates in); 10 weight fragments x 2 row fragments each
        {
            const h8 q0 = *reinterpret_cast<const h8*>(gcell), q1 = *reinterpret_cast<const h8*>(gcell + 1024);
            const h8 r0 = *reinterpret_cast<const h8*>(gcell + 2048), r1 = *reinterpret_cast<const h8*>(gcell + 3072);
            h8 wv[2 * TN2 + 4];
#pragma unroll
            for (int i = 0; i < 4; ++i) wv[i] = cur[i];
#pragma unroll
            for (int i = 4; i < 2 * TN2; ++i) wv[i] = frag(w2s, i % TN2, i / TN2);
            if (!last) {
#pragma unroll
                for (int i = 0; i < 4; ++i) wv[2 * TN2 + i] = frag(w1tile(0), i, 0);   // first fragments of step (j+1, 0)
            }
#pragma unroll
            for (int i = 0; i < 2 * TN2; ++i) {
                const int a2 = i % TN2;
                if (ABL == 4) { acc2[a2][0][0] += (float)wv[i][0] * (float)q0[1]; acc2[a2][1][1] += (float)wv[i][1] * (float)r1[0]; continue; }
                acc2[a2][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[i], i < TN2 ? q0 : r0, acc2[a2][0], 0, 0, 0);
                acc2[a2][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[i], i < TN2 ? q1 : r1, acc2[a2][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);          // q0, q1, r0, r1
#pragma unroll
            for (int i = 0; i < 2 * TN2; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            if (!last) {
#pragma unroll
                for (int i = 0; i < 4; ++i) cur[i] = wv[2 * TN2 + i];
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- bias + residual + fp16 store: the GEMM epilogue itself (a 64-row x C-wide "tile" whose wave slice is C / 2 columns) ----
    const int slab_of[1] = {-1};
    igemm_epilogue<64, C>(p.g, acc2, m_of, (C / 2) * wn, fq, 0, slab_of);
}

extern "C" int lcm_canonical_splits(int kind, int m_img, int N, int K, int aux, int ph);

extern "C" int lcm_mlp_geglu_f16(const void* x, int ldx, const void* W1, const void* ln_g, const void* ln_c, float eps,
                                 const void* W2, const void* b2, void* out, int ldo, int M, int C, int img_rows, void* stream) {
    LCM_REQUIRE(x && W1 && ln_g && ln_c && W2 && out, "mlp_geglu: null pointer");
    LCM_REQUIRE(M > 0, "mlp_geglu: bad M=%d", M);
    LCM_REQUIRE(C == 320, "mlp_geglu: C=%d (register budget of the kernel: C = 320 only)", C);
    LCM_REQUIRE(ldx % 8 == 0 && ldo % 4 == 0, "mlp_geglu: ldx=%d (%%8) / ldo=%d (%%4) misaligned", ldx, ldo);
    if (img_rows <= 0 || M % img_rows) img_rows = M;
    const int parts = lcm_canonical_splits(0, img_rows, C, 4 * C, 1, 0);
    LCM_REQUIRE(parts == 1, "mlp_geglu: the canonical K partition of ff.net.2 at %d rows per image has %d parts; the fused "
                "kernel accumulates one (use the two-launch form)", img_rows, parts);
    MlpParams p = {};
    p.X = (const half_t*)x; p.ldx = ldx; p.W1 = (const half_t*)W1; p.ln_g = (const float*)ln_g; p.ln_c = (const float*)ln_c;
    p.eps = eps; p.W2 = (const half_t*)W2;
    p.g.out = (half_t*)out; p.g.ldo = ldo; p.g.bias = (const half_t*)b2; p.g.res = (const half_t*)x; p.g.ldr = ldx;
    p.g.M = M; p.g.N = C; p.g.K = 4 * C; p.g.splits = 1; p.g.epi = 0; p.g.out_scale = 1.0f; p.g.rows_per_batch = 1;
    constexpr int smem = 5 * 128 * 128 + 320 * 128 + 2 * 8 * 320 * 4 + 16 * 1024;
    static LcmDevOnce attr_once;
    if (auto once_guard = attr_once.first()) {
        once_guard.check(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_geglu_kernel<320>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    }
    hipStream_t s = (hipStream_t)stream;
    lcm_prof_start("mlp_geglu_kernel<320>", s);
    static const int abl = getenv("LCM_MLP_ABLATE") ? atoi(getenv("LCM_MLP_ABLATE")) : 0;       // diagnostic builds only
    const dim3 grid((M + 127) / 128);
    if (abl) {
        auto set = [&](const void* f) { (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, smem); };
        if (abl == 1) { set((const void*)&mlp_geglu_kernel<320, 1>); hipLaunchKernelGGL((mlp_geglu_kernel<320, 1>), grid, dim3(512), smem, s, p); }
        else if (abl == 2) { set((const void*)&mlp_geglu_kernel<320, 2>); hipLaunchKernelGGL((mlp_geglu_kernel<320, 2>), grid, dim3(512), smem, s, p); }
        else if (abl == 3) { set((const void*)&mlp_geglu_kernel<320, 3>); hipLaunchKernelGGL((mlp_geglu_kernel<320, 3>), grid, dim3(512), smem, s, p); }
        else { set((const void*)&mlp_geglu_kernel<320, 4>); hipLaunchKernelGGL((mlp_geglu_kernel<320, 4>), grid, dim3(512), smem, s, p); }
    } else
    hipLaunchKernelGGL((mlp_geglu_kernel<320>), grid, dim3(512), smem, s, p);
    lcm_prof_stop(s);
    LCM_CHECK_LAUNCH("mlp_geglu");
    return LCM_OK;
}
