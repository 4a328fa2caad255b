// GroupNorm(+SiLU), LayerNorm and row softmax for pixel-major fp16 activations on gfx950.
//
// Replaces torch.nn.GroupNorm (+ F.silu) in ResnetBlock2D / Transformer2DModel / conv_norm_out,
// torch.nn.LayerNorm in BasicTransformerBlock, and the softmax of the AutoencoderKL mid attention,
// as reached from backends/cuda_worker.py:221-229.  All HBM-bound: 16-byte vector accesses, fp32
// statistics, wave64 shuffle reductions, and NO atomics (bit-reproducible run to run: the worker
// contract "same seed => identical PNG", tests/test_sdxl_worker.py:171-198).
//
// GroupNorm over [B][HW][C] (C = C1 (+C2 for the fused skip concat)) runs as
//   1. gn_stats:    grid (chunks, B): per-channel partial sum / sum-of-squares of a row chunk, folded to
//                   the 32 groups -> part[b][chunk][g][2]
//   2. gn_finalize: one wave per (b, g): fixed-order reduction of the chunk partials -> mean, rstd
//   3. gn_apply:    y = silu((x - mean) * rstd * gamma + beta), written as ONE concatenated tensor.
#include "common.h"

#define GN_MAXC 2560
#define GN_LDS_FLOATS 5120  // max(nrl * C, C) * 2

__device__ __forceinline__ h8 load_cat8(const half_t* x, int C1, const half_t* x2, int C2, long long row, int c) {
    // 8 channels starting at c of row `row` of the virtual concat [x | x2]; C1 % 8 == 0
    return (c < C1) ? *reinterpret_cast<const h8*>(x + row * C1 + c)
                    : *reinterpret_cast<const h8*>(x2 + row * C2 + (c - C1));
}

__global__ __launch_bounds__(256) void gn_stats_kernel(const half_t* __restrict__ x, int C1,
                                                       const half_t* __restrict__ x2, int C2,
                                                       float* __restrict__ part, int HW, int groups, int nchunk,
                                                       int GN_ROWS) {
    __shared__ float red[GN_LDS_FLOATS];
    const int C = C1 + C2, ncc = C >> 3, cpg = C / groups;
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int r0 = chunk * GN_ROWS, r1 = min(HW, r0 + GN_ROWS);
    const long long rowbase = (long long)b * HW;
    const int nrl = ncc <= 256 ? 256 / ncc : 1;        // row lanes
    if (ncc <= 256) {
        const int rl = tid / ncc, cc = tid - rl * ncc;
        if (rl < nrl) {
            float s[8], q[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
            int r = r0 + rl;
            // 4 independent 16-byte loads in flight per thread (HBM-bound: keep the memory pipe full)
            for (; r + 3 * nrl < r1; r += 4 * nrl) {
                h8 v0 = load_cat8(x, C1, x2, C2, rowbase + r, cc * 8);
                h8 v1 = load_cat8(x, C1, x2, C2, rowbase + r + nrl, cc * 8);
                h8 v2 = load_cat8(x, C1, x2, C2, rowbase + r + 2 * nrl, cc * 8);
                h8 v3 = load_cat8(x, C1, x2, C2, rowbase + r + 3 * nrl, cc * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float a = (float)v0[j], b = (float)v1[j], c = (float)v2[j], d = (float)v3[j];
                    s[j] += (a + b) + (c + d);
                    q[j] += (a * a + b * b) + (c * c + d * d);
                }
            }
            for (; r < r1; r += nrl) {
                h8 v = load_cat8(x, C1, x2, C2, rowbase + r, cc * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) { float f = (float)v[j]; s[j] += f; q[j] += f * f; }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                red[(rl * C + cc * 8 + j) * 2 + 0] = s[j];
                red[(rl * C + cc * 8 + j) * 2 + 1] = q[j];
            }
        }
    } else {
        for (int cc = tid; cc < ncc; cc += 256) {
            float s[8], q[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
            for (int r = r0; r < r1; ++r) {
                h8 v = load_cat8(x, C1, x2, C2, rowbase + r, cc * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) { float f = (float)v[j]; s[j] += f; q[j] += f * f; }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { red[(cc * 8 + j) * 2] = s[j]; red[(cc * 8 + j) * 2 + 1] = q[j]; }
        }
    }
    __syncthreads();
    if (tid < groups) {
        float s = 0.f, q = 0.f;
        for (int rl = 0; rl < nrl; ++rl)
            for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) {
                s += red[(rl * C + c) * 2];
                q += red[(rl * C + c) * 2 + 1];
            }
        float* o = part + (((long long)b * nchunk + chunk) * groups + tid) * 2;
        o[0] = s; o[1] = q;
    }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const half_t* __restrict__ x, int C1,
                                                       const half_t* __restrict__ x2, int C2,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       half_t* __restrict__ out, int HW, int silu, int rows_per_wg) {
    // y = act(x * scale[b][c] + shift[b][c]); a thread keeps one 8-channel chunk and walks rows, so its affine
    // coefficients live in registers and the loop is load / 8 fma / store.
    const int C = C1 + C2, ncc = C >> 3;
    const int b = blockIdx.y, tid = threadIdx.x;
    const int r0 = blockIdx.x * rows_per_wg, r1 = min(HW, r0 + rows_per_wg);
    const long long rowbase = (long long)b * HW;
    const int nrl = ncc <= 256 ? 256 / ncc : 1;
    for (int cc0 = 0; cc0 < ncc; cc0 += 256) {
        int rl, cc;
        if (ncc <= 256) { rl = tid / ncc; cc = tid - rl * ncc; if (rl >= nrl) return; }
        else { rl = 0; cc = cc0 + tid; if (cc >= ncc) continue; }
        const int c = cc * 8;
        float sc[8], sh[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { sc[j] = scale[(long long)b * C + c + j]; sh[j] = shift[(long long)b * C + c + j]; }
        int r = r0 + rl;
        for (; r + nrl < r1; r += 2 * nrl) {
            h8 v0 = load_cat8(x, C1, x2, C2, rowbase + r, c);
            h8 v1 = load_cat8(x, C1, x2, C2, rowbase + r + nrl, c);
            h8 o0, o1;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float f0 = __builtin_fmaf((float)v0[j], sc[j], sh[j]), f1 = __builtin_fmaf((float)v1[j], sc[j], sh[j]);
                if (silu) { f0 = silu_f(f0); f1 = silu_f(f1); }
                o0[j] = (half_t)f0; o1[j] = (half_t)f1;
            }
            *reinterpret_cast<h8*>(out + (rowbase + r) * C + c) = o0;
            *reinterpret_cast<h8*>(out + (rowbase + r + nrl) * C + c) = o1;
        }
        for (; r < r1; r += nrl) {
            h8 v = load_cat8(x, C1, x2, C2, rowbase + r, c);
            h8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float f = __builtin_fmaf((float)v[j], sc[j], sh[j]);
                if (silu) f = silu_f(f);
                o[j] = (half_t)f;
            }
            *reinterpret_cast<h8*>(out + (rowbase + r) * C + c) = o;
        }
        if (ncc <= 256) break;
    }
}

// Same reduction, but emits the GroupNorm affine folded per (image, channel):
//   scale[b][c] = rstd * gamma[c],  shift[b][c] = beta[c] - mean * rstd * gamma[c]
// consumed by the fused conv loader (conv_halo.hip) so the normalised tensor is never materialised.
__global__ __launch_bounds__(64) void gn_finalize_affine_kernel(const float* __restrict__ part, const half_t* __restrict__ gamma,
                                                                const half_t* __restrict__ beta, float* __restrict__ scale,
                                                                float* __restrict__ shift, int nchunk, int groups, int C,
                                                                float inv_count, float eps) {
    const int bg = blockIdx.x;
    const int b = bg / groups, g = bg - b * groups, lane = threadIdx.x;
    float s = 0.f, q = 0.f;
    for (int c = lane; c < nchunk; c += 64) {
        const float* p = part + (((long long)b * nchunk + c) * groups + g) * 2;
        s += p[0]; q += p[1];
    }
    s = wave_sum(s); q = wave_sum(q);
    const float mean = s * inv_count;
    const float rstd = rsqrtf(fmaxf(q * inv_count - mean * mean, 0.f) + eps);
    const int cpg = C / groups;
    for (int j = lane; j < cpg; j += 64) {
        const int c = g * cpg + j;
        const float a = rstd * (float)gamma[c];
        scale[(long long)b * C + c] = a;
        shift[(long long)b * C + c] = (float)beta[c] - mean * a;
    }
}

// GroupNorm statistics of one (image, group) from the per-channel partial statistics the producing contraction kernel
// wrote in its epilogue (csrc/igemm_common.h): stats[(b*P + p)][C][2].  The group's channels may span both sources of a
// fused concat.  ONE routine for the finalize kernel and the single-launch kernel below (which of the two runs depends on
// the tensor size, hence on the batch: they must agree bit for bit).  Fixed order: thread-sequential over slabs (4 in
// flight), wave butterfly, then the 4 wave sums in wave order.  `red` = 8 floats of LDS.  Returns (mean, rstd) to every thread.
__device__ __forceinline__ float2 gn_group_stats(const float* __restrict__ st1, int P1, int C1, const float* __restrict__ st2, int P2,
                                                 int C2, int b, int c_lo, int c_hi, float inv_count, float eps, float* red) {
    const int tid = threadIdx.x;
    float s = 0.f, q = 0.f;
    auto accumulate = [&](const float* __restrict__ st, int P, int Cs, int a, int w) {
        if (w <= 0 || st == nullptr) return;
        const int spl = 256 / w;                   // slab lanes
        const int sl = tid / w, jc = tid - sl * w;
        if (sl >= spl) return;
        const float* base = st + ((long long)b * P * Cs + a + jc) * 2;
        const long long stride = (long long)Cs * 2;
        int pp = sl;
        for (; pp + 3 * spl < P; pp += 4 * spl) {
            const float2 v0 = *reinterpret_cast<const float2*>(base + pp * stride);
            const float2 v1 = *reinterpret_cast<const float2*>(base + (pp + spl) * stride);
            const float2 v2 = *reinterpret_cast<const float2*>(base + (pp + 2 * spl) * stride);
            const float2 v3 = *reinterpret_cast<const float2*>(base + (pp + 3 * spl) * stride);
            s += (v0.x + v1.x) + (v2.x + v3.x);
            q += (v0.y + v1.y) + (v2.y + v3.y);
        }
        for (; pp < P; pp += spl) {
            const float2 v = *reinterpret_cast<const float2*>(base + pp * stride);
            s += v.x; q += v.y;
        }
    };
    accumulate(st1, P1, C1, c_lo, min(c_hi, C1) - c_lo);                       // channels of the group in source 1
    accumulate(st2, P2, C2, max(c_lo, C1) - C1, c_hi - max(c_lo, C1));         // ... and in source 2
    s = wave_sum(s); q = wave_sum(q);
    if ((tid & 63) == 0) { red[(tid >> 6) * 2] = s; red[(tid >> 6) * 2 + 1] = q; }
    __syncthreads();
    const float ts = ((red[0] + red[2]) + red[4]) + red[6];
    const float tq = ((red[1] + red[3]) + red[5]) + red[7];
    const float mean = ts * inv_count;
    return make_float2(mean, rsqrtf(fmaxf(tq * inv_count - mean * mean, 0.f) + eps));
}

__global__ __launch_bounds__(256) void gn_finalize_from_stats_kernel(const float* __restrict__ st1, int P1, int C1,
                                                                     const float* __restrict__ st2, int P2, int C2,
                                                                     const half_t* __restrict__ gamma, const half_t* __restrict__ beta,
                                                                     float* __restrict__ scale, float* __restrict__ shift,
                                                                     int groups, float inv_count, float eps) {
    __shared__ float red[8];
    const int C = C1 + C2, cpg = C / groups;
    const int bg = blockIdx.x, b = bg / groups, g = bg - b * groups, tid = threadIdx.x;
    const int c_lo = g * cpg;
    const float2 mr = gn_group_stats(st1, P1, C1, st2, P2, C2, b, c_lo, c_lo + cpg, inv_count, eps, red);
    for (int j = tid; j < cpg; j += 256) {
        const int c = c_lo + j;
        const float a = mr.y * (float)gamma[c];
        scale[(long long)b * C + c] = a;
        shift[(long long)b * C + c] = (float)beta[c] - mr.x * a;
    }
}

// One launch for SMALL tensors (batch 1-2 UNet levels, the VAE's 64^2 level), where finalize + apply are two launches
// at the ~3 us launch floor each: workgroup (pixel slice, image, group) re-derives its group's mean / rstd from the
// producer statistics -- P * cpg * 8 bytes out of L2, cheap to repeat per slice -- and applies the affine (+SiLU) to its
// slice of pixels for the group's cpg channels.  Same arithmetic as the two-launch path.  The first GN_PRE activation
// pairs of every thread are fetched BEFORE the statistics are reduced: the two dependent memory round trips of the kernel
// (statistics, then activations) overlap instead of adding up (the kernel is pure latency at these sizes).
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
#define GN_PRE 6
__global__ __launch_bounds__(256) void gn_from_stats_fused_kernel(const half_t* __restrict__ x, int C1, const half_t* __restrict__ x2, int C2,
                                                                  const float* __restrict__ st1, int P1, const float* __restrict__ st2, int P2,
                                                                  const half_t* __restrict__ gamma, const half_t* __restrict__ beta,
                                                                  half_t* __restrict__ out, int HW, int groups, float inv_count, float eps,
                                                                  int silu, int rows_per_wg) {
    __shared__ float red[8];
    __shared__ float sc_s[128], sh_s[128];
    const int C = C1 + C2, cpg = C / groups;
    const int bg = blockIdx.y, b = bg / groups, g = bg - b * groups, tid = threadIdx.x;
    const int c_lo = g * cpg;
    // apply: work item = (pixel, channel pair of the group); consecutive threads take consecutive pairs of one pixel
    const int hp = cpg >> 1;
    const int r0 = blockIdx.x * rows_per_wg, r1 = min(HW, r0 + rows_per_wg);
    const long long rowbase = (long long)b * HW;
    const int items = (r1 - r0) * hp;
    auto load_item = [&](int i) -> h2v {
        const int pr = i / hp, j = i - pr * hp;
        const int c = c_lo + 2 * j;
        const long long row = rowbase + r0 + pr;
        return (c < C1) ? *reinterpret_cast<const h2v*>(x + row * C1 + c) : *reinterpret_cast<const h2v*>(x2 + row * C2 + (c - C1));
    };
    h2v pre[GN_PRE];
#pragma unroll
    for (int k = 0; k < GN_PRE; ++k) {
        const int i = tid + 256 * k;
        pre[k] = (i < items) ? load_item(i) : (h2v){(half_t)0, (half_t)0};
    }
    const float2 mr = gn_group_stats(st1, P1, C1, st2, P2, C2, b, c_lo, c_lo + cpg, inv_count, eps, red);
    if (tid < cpg) {
        const float a = mr.y * (float)gamma[c_lo + tid];
        sc_s[tid] = a;
        sh_s[tid] = (float)beta[c_lo + tid] - mr.x * a;
    }
    __syncthreads();
    auto apply_store = [&](int i, h2v v) {
        const int pr = i / hp, j = i - pr * hp;
        const int c = c_lo + 2 * j;
        const long long row = rowbase + r0 + pr;
        float f0 = __builtin_fmaf((float)v[0], sc_s[2 * j], sh_s[2 * j]), f1 = __builtin_fmaf((float)v[1], sc_s[2 * j + 1], sh_s[2 * j + 1]);
        if (silu) { f0 = silu_f(f0); f1 = silu_f(f1); }
        h2v o = {(half_t)f0, (half_t)f1};
        *reinterpret_cast<h2v*>(out + row * C + c) = o;
    };
#pragma unroll
    for (int k = 0; k < GN_PRE; ++k) {
        const int i = tid + 256 * k;
        if (i < items) apply_store(i, pre[k]);
    }
    for (int i = tid + 256 * GN_PRE; i < items; i += 256) apply_store(i, load_item(i));
}

static long long g_gn_fused_bytes = 8ll << 20;   // tensors up to this size take the single-launch path above
extern "C" int lcm_set_gn_fused_bytes(int64_t bytes) { g_gn_fused_bytes = bytes; return LCM_OK; }

// rows per statistics chunk: a function of the image size only (never of the batch), so the partial sums -- and the
// order they are folded in -- are the same for a request alone and inside a batch
static inline int gn_rows(int B, int HW) {
    (void)B;
    long long r = ((long long)HW + 1023) / 1024;
    if (r < 16) r = 16;
    if (r > 4096) r = 4096;
    return (int)r;
}
static inline int gn_nchunk(int B, int HW) { const int r = gn_rows(B, HW); return (HW + r - 1) / r; }

extern "C" int lcm_groupnorm_affine_f16(const void* x, int C1, const void* x2, int C2, const void* gamma,
                                        const void* beta, void* scale_out, void* shift_out, int B, int HW, int groups,
                                        float eps, void* ws, void* stream);
static int launch_gn_apply(const void* x, int C1, const void* x2, int C2, const float* scale, const float* shift, void* out,
                           int B, int HW, int silu, hipStream_t s);

extern "C" int64_t lcm_groupnorm_ws_bytes(int B, int HW, int C, int groups) {
    // chunk partials + the [B][C] scale / shift tables used by lcm_groupnorm_f16
    return ((int64_t)B * gn_nchunk(B, HW) * groups * 2 + 2 * (int64_t)B * C) * 4 + 64;
}

extern "C" int lcm_groupnorm_f16(const void* x, int C1, const void* x2, int C2, const void* gamma,
                                 const void* beta, void* out, int B, int HW, int groups, float eps, int silu,
                                 void* ws, void* stream) {
    LCM_REQUIRE(x && gamma && beta && out && ws, "groupnorm: null pointer");
    if (!x2) C2 = 0;
    const int C = C1 + C2;
    float* part = (float*)ws;
    float* scale = part + (((long long)B * gn_nchunk(B, HW) * groups * 2 + 3) & ~3ll);
    float* shift = scale + (long long)B * C;
    int rc = lcm_groupnorm_affine_f16(x, C1, x2, C2, gamma, beta, scale, shift, B, HW, groups, eps, ws, stream);
    if (rc) return rc;
    return launch_gn_apply(x, C1, x2, C2, scale, shift, out, B, HW, silu, (hipStream_t)stream);
}

extern "C" int lcm_groupnorm_affine_f16(const void* x, int C1, const void* x2, int C2, const void* gamma,
                                        const void* beta, void* scale_out, void* shift_out, int B, int HW, int groups,
                                        float eps, void* ws, void* stream) {
    LCM_REQUIRE(x && gamma && beta && scale_out && shift_out && ws, "groupnorm_affine: null pointer");
    if (!x2) C2 = 0;
    const int C = C1 + C2;
    LCM_REQUIRE(B > 0 && HW > 0 && groups > 0 && groups <= 64, "groupnorm_affine: bad shape B=%d HW=%d G=%d", B, HW, groups);
    LCM_REQUIRE(C1 % 8 == 0 && C2 % 8 == 0 && C % groups == 0 && C <= GN_MAXC, "groupnorm_affine: bad channels %d+%d", C1, C2);
    hipStream_t s = (hipStream_t)stream;
    const int nchunk = gn_nchunk(B, HW);
    float* part = (float*)ws;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nchunk, B), dim3(256), 0, s, (const half_t*)x, C1, (const half_t*)x2, C2,
                       part, HW, groups, nchunk, gn_rows(B, HW));
    LCM_CHECK_LAUNCH("gn_stats");
    hipLaunchKernelGGL(gn_finalize_affine_kernel, dim3(B * groups), dim3(64), 0, s, part, (const half_t*)gamma,
                       (const half_t*)beta, (float*)scale_out, (float*)shift_out, nchunk, groups, C,
                       1.0f / ((float)HW * (float)(C / groups)), eps);
    LCM_CHECK_LAUNCH("gn_finalize_affine");
    return LCM_OK;
}

static int launch_gn_apply(const void* x, int C1, const void* x2, int C2, const float* scale, const float* shift, void* out,
                           int B, int HW, int silu, hipStream_t s) {
    const int C = C1 + C2, ncc = C >> 3;
    const int nrl = ncc <= 256 ? 256 / ncc : 1;
    long long rows = ((long long)B * HW + 1023) / 1024;
    if (rows < nrl) rows = nrl;
    if (rows > 16ll * nrl) rows = 16ll * nrl;
    const int rows_per_wg = (int)rows;
    hipLaunchKernelGGL(gn_apply_kernel, dim3((HW + rows_per_wg - 1) / rows_per_wg, B), dim3(256), 0, s,
                       (const half_t*)x, C1, (const half_t*)x2, C2, scale, shift, (half_t*)out, HW, silu, rows_per_wg);
    LCM_CHECK_LAUNCH("gn_apply");
    return LCM_OK;
}

extern "C" int lcm_groupnorm_from_stats_f16(const void* x, int C1, const void* x2, int C2, const void* stats1, int P1,
                                            const void* stats2, int P2, const void* gamma, const void* beta, void* out,
                                            int B, int HW, int groups, float eps, int silu, void* ws, void* stream) {
    LCM_REQUIRE(stats1 && gamma && beta && ws && (x || !out), "groupnorm_from_stats: null pointer");
    if (!x2 && !(out == nullptr && stats2)) C2 = 0;
    const int C = C1 + C2;
    LCM_REQUIRE(B > 0 && HW > 0 && groups > 0 && groups <= 64 && P1 > 0, "groupnorm_from_stats: bad shape");
    LCM_REQUIRE(C1 % 8 == 0 && C2 % 8 == 0 && C % groups == 0 && C <= GN_MAXC, "groupnorm_from_stats: bad channels %d+%d", C1, C2);
    LCM_REQUIRE(C2 == 0 || (stats2 && P2 > 0), "groupnorm_from_stats: second source needs its statistics");
    hipStream_t s = (hipStream_t)stream;
    const int cpg = C / groups;
    if (out && (long long)B * HW * C * 2 <= g_gn_fused_bytes && cpg % 2 == 0 && cpg <= 128) {
        int nslice = 512 / (B * groups);
        if (nslice < 1) nslice = 1;
        if (nslice > (HW + 31) / 32) nslice = (HW + 31) / 32;
        const int rows = (HW + nslice - 1) / nslice;
        hipLaunchKernelGGL(gn_from_stats_fused_kernel, dim3((HW + rows - 1) / rows, B * groups), dim3(256), 0, s,
                           (const half_t*)x, C1, (const half_t*)x2, C2, (const float*)stats1, P1, (const float*)stats2, P2,
                           (const half_t*)gamma, (const half_t*)beta, (half_t*)out, HW, groups,
                           1.0f / ((float)HW * (float)cpg), eps, silu, rows);
        LCM_CHECK_LAUNCH("gn_from_stats_fused");
        return LCM_OK;
    }
    float* scale = (float*)ws;
    float* shift = scale + (long long)B * C;
    hipLaunchKernelGGL(gn_finalize_from_stats_kernel, dim3(B * groups), dim3(256), 0, s, (const float*)stats1, P1, C1,
                       (const float*)stats2, P2, C2, (const half_t*)gamma, (const half_t*)beta, scale, shift, groups,
                       1.0f / ((float)HW * (float)(C / groups)), eps);
    LCM_CHECK_LAUNCH("gn_finalize_from_stats");
    if (!out) return LCM_OK;        // tables only: ws = scale [B][C] | shift [B][C] for lcm_conv3x3_gn_f16
    return launch_gn_apply(x, C1, x2, C2, scale, shift, out, B, HW, silu, s);
}

// ---------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row held in registers (C <= 1536), two-pass statistics.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_kernel(const half_t* __restrict__ x, const half_t* __restrict__ gamma,
                                                        const half_t* __restrict__ beta, half_t* __restrict__ out,
                                                        int M, int C, float eps) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int ncc = C >> 3;
    const half_t* xr = x + (long long)row * C;
    h8 v[3];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int cc = lane + 64 * i;
        if (cc < ncc) {
            v[i] = *reinterpret_cast<const h8*>(xr + cc * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (float)v[i][j];
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int cc = lane + 64 * i;
        if (cc < ncc) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { float d = (float)v[i][j] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    half_t* orow = out + (long long)row * C;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int cc = lane + 64 * i;
        if (cc < ncc) {
            h8 gm = *reinterpret_cast<const h8*>(gamma + cc * 8);
            h8 bt = *reinterpret_cast<const h8*>(beta + cc * 8);
            h8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (half_t)(((float)v[i][j] - mean) * rstd * (float)gm[j] + (float)bt[j]);
            *reinterpret_cast<h8*>(orow + cc * 8) = o;
        }
    }
}

extern "C" int lcm_layernorm_f16(const void* x, const void* gamma, const void* beta, void* out, int M, int C,
                                 float eps, void* stream) {
    LCM_REQUIRE(x && gamma && beta && out, "layernorm: null pointer");
    LCM_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && C <= 1536, "layernorm: bad shape M=%d C=%d", M, C);
    hipLaunchKernelGGL(layernorm_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const half_t*)x,
                       (const half_t*)gamma, (const half_t*)beta, (half_t*)out, M, C, eps);
    LCM_CHECK_LAUNCH("layernorm");
    return LCM_OK;
}

// ---------------------------------------------------------------------------------------------
// Row softmax in place: one wave per row; any n <= ld, ld % 8 == 0.  The padding columns [n, ld) of each row are set
// to zero (the AutoencoderKL attention pads S = h*w up to a multiple of 64 for the MFMA contractions around it).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(half_t* __restrict__ x, int rows, int n, int ld) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    half_t* xr = x + (long long)row * ld;
    const int ncc = (n + 7) >> 3, nld = ld >> 3;
    float m = -3.0e38f;
    for (int cc = lane; cc < ncc; cc += 64) {
        h8 v = *reinterpret_cast<const h8*>(xr + cc * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) if (cc * 8 + j < n) m = fmaxf(m, (float)v[j]);
    }
    m = wave_max(m);
    float s = 0.f;
    for (int cc = lane; cc < ncc; cc += 64) {
        h8 v = *reinterpret_cast<const h8*>(xr + cc * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) if (cc * 8 + j < n) s += __expf((float)v[j] - m);
    }
    const float inv = 1.0f / wave_sum(s);
    for (int cc = lane; cc < nld; cc += 64) {
        h8 v = *reinterpret_cast<const h8*>(xr + cc * 8);
        h8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (cc * 8 + j < n) ? (half_t)(__expf((float)v[j] - m) * inv) : (half_t)0;
        *reinterpret_cast<h8*>(xr + cc * 8) = o;
    }
}

extern "C" int lcm_softmax_rows_f16(void* x, int rows, int n, int ld, void* stream) {
    LCM_REQUIRE(x && rows > 0 && n > 0 && n <= ld && ld % 8 == 0, "softmax: bad shape rows=%d n=%d ld=%d", rows, n, ld);
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (half_t*)x, rows,
                       n, ld);
    LCM_CHECK_LAUNCH("softmax_rows");
    return LCM_OK;
}
