// Shared pieces of the MFMA contraction kernels (igemm.hip, conv_halo.hip): parameter block, XCD-aware
// tile remap, and the common epilogue (split-K slab store | bias + row add + scale + residual | GEGLU).
#pragma once
#include "common.h"

struct IgemmParams {
    const half_t* A;
    const half_t* A2;
    const half_t* W;
    const half_t* bias;
    const half_t* rowadd;
    const half_t* res;
    half_t* out;
    int M, N, K;
    int lda, lda2, K1;
    int ldo, ldr, ld_rowadd, rows_per_batch;
    int epi;
    float out_scale;
    long long strideA, strideW, strideO;
    // conv
    int Hin, Win, Cin, Hout, Wout, stride, ups;
    int mtiles, ntiles;
    int splits;          // split-K: blockIdx.y owns k-tiles [y*nk/splits, (y+1)*nk/splits); partials -> ws
    float* ws;           // fp32 [splits][M][N]
    float* stats;        // optional fused GroupNorm statistics of the OUTPUT: [slab][N][2] = per-channel (sum, sum of
                         // squares) of the fp16-rounded values each half-tile (or reduce slab) stores; null = off
    int reduce_rows;     // (unused)
    // geometry of the canonical 32-pixel statistics slabs of the OUTPUT (splitk_reduce_kernel walks them; it must visit
    // the pixels of a slab in exactly the order a tile epilogue does): rg_kind 0 = 32 consecutive rows, 1 = 2x16 pixel
    // patches, 2 = 4x8 patches of an rg_IH x rg_IW image (the input image when rg_ph, four phase slabs per patch);
    // output image rg_OH x rg_OW
    int rg_kind, rg_ph, rg_IH, rg_IW, rg_OH, rg_OW;
    int seg_parts;       // > 1: ONE workgroup walks all K, accumulating the canonical parts separately and adding them in
                         // part order (bit-identical to the split launch + reduce, no fp32 slabs): batched launches
    int img_rows;        // output rows per image (GEMM kinds; 0 = one image)
    // LayerNorm folded into the GEMM that consumes it (lcm_gemm_ln_f16): W holds gamma (*) W, the kernel accumulates the
    // row sums of A while it walks K and the epilogue applies  y = rstd * (acc - mean * ln_g[n]) + ln_c[n]
    const float* ln_g;   // [N] fp32: sum_k W'[n][k]   (null = plain GEMM)
    const float* ln_c;   // [N] fp32: sum_k beta[k] W[n][k] + bias[n]
    float ln_eps;
    int n_iters;         // igemm2: consecutive n-tiles one workgroup walks with a continuous LDS-DMA pipeline (>= 1)
    long long stats_cap; // host side: bytes the caller allocated behind `stats` (the launch is refused when the slabs do not fit)
};

// fused-statistics bounds check shared by the contraction entry points: slabs_per_image x images x N x (sum, sumsq) fp32
#define LCM_STATS_FIT(p, slabs, images, what)                                                                              \
    do {                                                                                                                   \
        const long long need__ = (long long)(slabs) * (images) * (p).N * 2 * (long long)sizeof(float);                     \
        if ((p).stats && need__ > (p).stats_cap) {                                                                         \
            lcm_set_error("%s: statistics buffer too small: %d slabs x %d images x %d channels need %lld bytes, have %lld " \
                          "(size it with lcm_stats_bytes)", what, (int)(slabs), (int)(images), (p).N, need__, (p).stats_cap); \
            return LCM_EINVAL;                                                                                             \
        }                                                                                                                  \
    } while (0)

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // contiguous chunk of tiles per XCD (blocks are dealt round-robin over the 8 XCDs)
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// LayerNorm fold: per-lane partial row sums of an A fragment (8 fp16 of row m = b*16 + frow): v_dot2_f32_f16 with (1,1)
// and with itself.  The 4 lanes (fq = 0..3) that share a row cover the 64 channels of a k-tile between them.
__device__ __forceinline__ void ln_accum(const h8& x, float& s, float& q) {
    const h2 one = {(half_t)1.0f, (half_t)1.0f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const h2 pr = {x[2 * i], x[2 * i + 1]};
        s = __builtin_amdgcn_fdot2(pr, one, s, false);
        q = __builtin_amdgcn_fdot2(pr, pr, q, false);
    }
}
// fold the 4 fq-lanes of a row (fixed butterfly order) and turn (sum, sum of squares) into (mean, rstd)
template <int TM>
__device__ __forceinline__ void ln_finish(float (&s)[TM], float (&q)[TM], int K, float eps) {
    const float inv = 1.0f / (float)K;
#pragma unroll
    for (int b = 0; b < TM; ++b) {
        float ss = s[b], qq = q[b];
        ss += __shfl_xor(ss, 16, 64); qq += __shfl_xor(qq, 16, 64);
        ss += __shfl_xor(ss, 32, 64); qq += __shfl_xor(qq, 32, 64);
        const float mean = ss * inv;
        s[b] = mean;
        q[b] = rsqrtf(fmaxf(__builtin_fmaf(-mean, mean, qq * inv), 0.f) + eps);
    }
}

// One lane's value / gate quad of the LayerNorm-folded GEGLU projection -> x * gelu(gate) in fp16.  THE element math of that
// epilogue: the GEMM epilogue below and the fused FeedForward kernel (mlp_fused.hip) both call it, so they round alike.
__device__ __forceinline__ h4 geglu_ln_quad(f4 x, f4 g, float ln_r, float ln_mu, const f4& gx, const f4& cx, const f4& gg, const f4& cg) {
#pragma clang fp contract(off)
    h4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float xv = __builtin_fmaf(ln_r, __builtin_fmaf(-ln_mu, gx[j], x[j]), cx[j]);
        const float gv = __builtin_fmaf(ln_r, __builtin_fmaf(-ln_mu, gg[j], g[j]), cg[j]);
        o[j] = (half_t)(xv * gelu_erf_f(gv));
    }
    return o;
}

// Column of the GEGLU output [M][F] that holds value channel 16 P + 4 fq + j: "operand order".  Within every group of 32 the
// columns are arranged so that 8 consecutive ones are exactly what ONE lane of the producing MFMA tiles holds (the quads of
// value/gate pair 2u and of pair 2u+1): the consumer (ff.net.2, its weight columns packed alike: packing.pack_ff2_cols) then
// multiplies the same values in the same k slots whether it reads them back from memory or takes them from the producer's
// registers (mlp_fused.hip).
__device__ __forceinline__ int geglu_store_col(int P, int fq) { return 32 * (P >> 1) + 8 * fq + 4 * (P & 1); }

// Position of 16-byte chunk c of tile pixel q in the LDS staging image of an output tile ([pixel][CH chunks], CH = BN / 8): the
// chunks of a pixel are rotated by the pixel index so that the 16 pixels a wave touches at once fall on 16 different slots
// (the rows are a multiple of 128 bytes long: unrotated, every pixel's chunk c sits on the same banks).
template <int CH>
__device__ __forceinline__ int stage_pos(int c, int q) { return (c + q) % CH; }
template <int CH>
__device__ __forceinline__ int stage_src(int cpos, int q) { return (cpos + CH - q % CH) % CH; }

// STAGED = 1: the residual tile is READ from, and the result tile WRITTEN to, an LDS image of the output tile (`stage`, BM x BN
// fp16, stage_pos layout; tile_epilogue_staged fills it by LDS-DMA beforehand and copies it out in whole rows afterwards) instead
// of global memory in 8-byte pieces per lane.  The arithmetic is this one function either way.
template <int BM, int BN, int STAGED = 0>
__device__ __forceinline__ void igemm_epilogue(const IgemmParams& p, f4 (&acc)[BN / 32][BM / 32], const int (&m_of)[BM / 32],
                                               int n_wave, int fq, int z, const int (&slab_of)[BM / 64],
                                               const float* ln_mu = nullptr, const float* ln_r = nullptr,
                                               const __attribute__((address_space(3))) float* ln_lds = nullptr, int n_tile0 = 0,
                                               char* stage = nullptr, int q_wave = 0, int nl_wave = 0) {
    // ln_lds: the BN entries of ln_g then of ln_c of this n-tile (first channel n_tile0), staged in LDS by the kernel --
    // an ordinary global load here, inside the K loop of a kernel that keeps LDS-DMA in flight, makes the compiler drain
    // the DMA ring (s_waitcnt vmcnt(0)) in front of every fragment read
    // ln_mu / ln_r: per fragment b, mean and rstd of this lane's row (LayerNorm fold), or null
    // m_of[b]: global output row of this lane in m-tile b, or -1 (outside the problem); n_wave: first channel of
    // this wave's BN/2-wide slice.  slab_of[bp]: index of the CANONICAL statistics slab that the fragment pair
    // (2bp, 2bp+1) of this wave covers (32 output pixels: 32 consecutive rows of a GEMM, a 2x16 / 4x8 pixel patch of a
    // 3x3 convolution), or -1.  Slabs are a property of the output tensor, not of the tile shape: a 128-row tile emits
    // two per wave, a 64-row tile one, with the same pixels and the same summation order inside each -- so the
    // GroupNorm statistics (and everything downstream) do not depend on the launch plan or on the batch size.
    // Every instantiation (tile shape, kernel variant, LDS- or global-resident constants) must round the same way: no
    // compiler-chosen fused multiply-adds in here, the two that are wanted are spelled out.
#pragma clang fp contract(off)
    constexpr int TM = BM / 32, TN = BN / 32;
    if (p.splits > 1) {   // split-K: raw fp32 partial slab, epilogue runs in splitk_reduce_kernel
        float* __restrict__ wsb = p.ws + (long long)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int b = 0; b < TM; ++b) {
            const int m = m_of[b];
            if (m < 0) continue;
#pragma unroll
            for (int a = 0; a < TN; ++a) {
                const int n = n_wave + a * 16 + fq * 4;
                *reinterpret_cast<f4*>(wsb + (long long)m * p.N + n) = acc[a][b];
            }
        }
        return;
    }

    // ---- epilogue: lane holds out[m = ..+frow][n = ..+fq*4 .. +3] ----
    // Loop order: n-tile outer, m-tile inner -- bias and the 8 statistics accumulators of one n-tile are the only
    // values live across the inner loop (keeps the epilogue inside the 128-VGPR budget of the single-buffer variant).
    half_t* __restrict__ outb = p.out + z * p.strideO;
    const bool do_stats = p.stats != nullptr && p.epi == 0;
    const int lane = threadIdx.x & 63;
    if (p.epi != 1) {
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            const int n = n_wave + a * 16 + fq * 4;
            f4 bias4 = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) { h4 t = *reinterpret_cast<const h4*>(p.bias + n); bias4 = (f4){(float)t[0], (float)t[1], (float)t[2], (float)t[3]}; }
            f4 lng4 = {0.f, 0.f, 0.f, 0.f}, lnc4 = {0.f, 0.f, 0.f, 0.f};
            if (ln_mu) {
                if (ln_lds) { lng4 = *reinterpret_cast<const __attribute__((address_space(3))) f4*>(ln_lds + (n - n_tile0));
                              lnc4 = *reinterpret_cast<const __attribute__((address_space(3))) f4*>(ln_lds + BN + (n - n_tile0)); }
                else { lng4 = *reinterpret_cast<const f4*>(p.ln_g + n); lnc4 = *reinterpret_cast<const f4*>(p.ln_c + n); }
            }
#pragma unroll
            for (int bp = 0; bp < TM / 2; ++bp) {
                float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int b = 2 * bp; b < 2 * bp + 2; ++b) {
                    const int m = m_of[b];
                    if (m < 0) continue;
                    f4 v = acc[a][b];
                    if (ln_mu) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = __builtin_fmaf(ln_r[b], __builtin_fmaf(-ln_mu[b], lng4[j], v[j]), lnc4[j]);
                    }
                    v[0] += bias4[0]; v[1] += bias4[1]; v[2] += bias4[2]; v[3] += bias4[3];
                    if (p.rowadd) { h4 t = *reinterpret_cast<const h4*>(p.rowadd + (long long)(m / p.rows_per_batch) * p.ld_rowadd + n);
                        v[0] += (float)t[0]; v[1] += (float)t[1]; v[2] += (float)t[2]; v[3] += (float)t[3]; }
                    v[0] *= p.out_scale; v[1] *= p.out_scale; v[2] *= p.out_scale; v[3] *= p.out_scale;
                    if (p.epi == 2) {          // quick_gelu: x * sigmoid(1.702 x)   (CLIP text encoder MLP)
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = v[j] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930157f * v[j]));
                    } else if (p.epi == 3) {   // exact gelu
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = gelu_erf_f(v[j]);
                    }
                    int soff = 0;
                    if constexpr (STAGED != 0) {
                        const int q = q_wave + b * 16 + (lane & 15), nl = nl_wave + a * 16 + fq * 4;
                        soff = q * (BN * 2) + stage_pos<BN / 8>(nl >> 3, q) * 16 + (nl & 4) * 2;
                    }
                    if (p.res) {
                        h4 t;
                        if constexpr (STAGED != 0) t = *reinterpret_cast<const h4*>(stage + soff);
                        else t = *reinterpret_cast<const h4*>(p.res + (long long)m * p.ldr + n);
                        v[0] += (float)t[0]; v[1] += (float)t[1]; v[2] += (float)t[2]; v[3] += (float)t[3]; }
                    h4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                    if constexpr (STAGED != 0) *reinterpret_cast<h4*>(stage + soff) = o;
                    else *reinterpret_cast<h4*>(outb + (long long)m * p.ldo + n) = o;
                    if (do_stats) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { const float f = (float)o[j]; ssum[j] += f; ssq[j] += f * f; }
                    }
                }
                if (do_stats && slab_of[bp] >= 0) {   // fold the 16 pixel-lanes of the channel quad; one lane writes (sum, sumsq) x 4
row16_sum8(ssum, ssq);
                    if ((lane & 15) == 0) {
                        float* dst = p.stats + ((long long)slab_of[bp] * p.N + n) * 2;
                        *reinterpret_cast<f4*>(dst) = (f4){ssum[0], ssq[0], ssum[1], ssq[1]};
                        *reinterpret_cast<f4*>(dst + 4) = (f4){ssum[2], ssq[2], ssum[3], ssq[3]};
                    }
                }
            }
        }
    } else if constexpr (TN % 2 == 0) {  // GEGLU: even n-tile = value rows, odd n-tile = gate rows (16-row interleave)
#pragma unroll
        for (int a = 0; a < TN; a += 2) {
            const int n = n_wave + a * 16 + fq * 4;      // packed row of the value
            f4 bx = {0.f, 0.f, 0.f, 0.f}, bg = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) {
                h4 tx = *reinterpret_cast<const h4*>(p.bias + n);
                h4 tg = *reinterpret_cast<const h4*>(p.bias + n + 16);
                bx = (f4){(float)tx[0], (float)tx[1], (float)tx[2], (float)tx[3]};
                bg = (f4){(float)tg[0], (float)tg[1], (float)tg[2], (float)tg[3]};
            }
            const int nout = geglu_store_col((n_wave + a * 16) >> 5, fq);      // operand order (see geglu_store_col)
            f4 gx = {0.f, 0.f, 0.f, 0.f}, cx = gx, gg = gx, cg = gx;
            if (ln_mu) {
                if (ln_lds) {
                    const __attribute__((address_space(3))) float* gl = ln_lds + (n - n_tile0);
                    gx = *reinterpret_cast<const __attribute__((address_space(3))) f4*>(gl);
                    cx = *reinterpret_cast<const __attribute__((address_space(3))) f4*>(gl + BN);
                    gg = *reinterpret_cast<const __attribute__((address_space(3))) f4*>(gl + 16);
                    cg = *reinterpret_cast<const __attribute__((address_space(3))) f4*>(gl + BN + 16);
                } else {
                    gx = *reinterpret_cast<const f4*>(p.ln_g + n); cx = *reinterpret_cast<const f4*>(p.ln_c + n);
                    gg = *reinterpret_cast<const f4*>(p.ln_g + n + 16); cg = *reinterpret_cast<const f4*>(p.ln_c + n + 16);
                }
            }
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                const int m = m_of[b];
                if (m < 0) continue;
                h4 o;
                if (ln_mu) {          // the folded LayerNorm's constants already carry the bias (lcm_gemm_ln_f16 takes none)
                    o = geglu_ln_quad(acc[a][b], acc[a + 1][b], ln_r[b], ln_mu[b], gx, cx, gg, cg);
                } else {
                    const f4 x = acc[a][b], g = acc[a + 1][b];
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (half_t)((x[j] + bx[j]) * gelu_erf_f(g[j] + bg[j]));
                }
                *reinterpret_cast<h4*>(outb + (long long)m * p.ldo + nout) = o;
            }
        }
    }
}

// The epilogue of a whole output tile with its global traffic in whole rows (plain launches: splits == 1, epi == 0, no LayerNorm
// fold).  The per-lane form above moves the residual and the result in 8-byte pieces, 16 rows x 32 bytes per wave instruction:
// 4x the cache lines per instruction of a full-row access, and twice the store instructions -- the residual of the 512x512
// AutoencoderKL conv2 launches cost +27 % of the kernel that way.  Here: (1) the residual tile arrives by
// LDS-DMA (16 bytes per lane, full 128-byte lines, no VGPRs) into `stage`, an image of the tile laid over the K loop's dead LDS
// buffers; (2) igemm_epilogue<.., 1> reads it from there and writes the fp16 results back into the same cells (each cell is
// read and written by ONE lane); (3) every thread copies 16-byte pieces out, 16 consecutive lanes covering one pixel's row.
// row_of(q) -> global output row of tile pixel q, or -1.  The arithmetic, the statistics and their order are those of the
// per-lane form: bit-identical (tests/test_ops_gpu.py::test_staged_epilogue_is_bit_identical).
static __device__ __attribute__((aligned(256))) half_t g_zero_page_epi[128];

template <int BM, int BN, class RowOf>
__device__ __forceinline__ void tile_epilogue_staged(const IgemmParams& p, f4 (&acc)[BN / 32][BM / 32], const int (&m_of)[BM / 32],
                                                     int n_base, int wm, int wn, int fq, const int (&slab_of)[BM / 64],
                                                     char* stage, RowOf row_of) {
    constexpr int CH = BN / 8, PIECES = BM * CH, NV = (PIECES + 255) / 256;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __syncthreads();                              // every wave is done with the K loop's LDS tiles
    if (p.res) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if ((wave + 4 * i) * 64 < PIECES) {   // wave-uniform: whole 1 KiB piece in range (PIECES is a multiple of 64)
                const int v = tid + 256 * i, q = v / CH, cpos = v - q * CH;
                const int m = row_of(q);
                const half_t* src = m >= 0 ? p.res + (long long)m * p.ldr + n_base + stage_src<CH>(cpos, q) * 8 : g_zero_page_epi;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(stage + (wave + 4 * i) * 1024), 16, 0, 0);
            }
        }
        __syncthreads();                          // drains the LDS-DMA (vmcnt(0)), then barrier
    }
    igemm_epilogue<BM, BN, 1>(p, acc, m_of, n_base + wn * (BN / 2), fq, 0, slab_of, nullptr, nullptr, nullptr, 0,
                              stage, wm * (BM / 2), wn * (BN / 2));
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + 256 * i;
        if (v < PIECES) {
            const int q = v / CH, cpos = v - q * CH;
            const int m = row_of(q);
            if (m >= 0) *reinterpret_cast<h8*>(p.out + (long long)m * p.ldo + n_base + stage_src<CH>(cpos, q) * 8) =
                            *reinterpret_cast<const h8*>(stage + v * 16);
        }
    }
}
