// Boundary / latency kernels of the LCM hot path on gfx950: latent-side convolutions, the time-embedding
// MLP, the LCM scheduler step, the RGB8 epilogue, the 8x8 latent blob, a tiled transpose.
// (UNet conv_in / conv_out, TimestepEmbedding, LCMScheduler.step, VaeImageProcessor.postprocess,
//  reached from backends/cuda_worker.py:221-229; numpy twins backends/rknnlcm.py:596-599, :211-264.)
#include "common.h"

// ---------------------------------------------------------------------------------------------
// conv3x3 from fp32 NCHW latents (Cin = 4) -> fp16 pixel-major [B,H,W,Cout], on the matrix pipe.
// K = 9 taps x 4 channels = 36.  The fp32 input (after in_scale and the optional 4x4 pre-transform) is split into
// hi = fp16(x) and lo = fp16(x - hi): K slots 0..35 carry hi, 36..71 lo against the SAME weights, 72..95 zeros -- three
// v_mfma_f32_16x16x32_f16 per 16 pixels x 16 channels with the input kept to ~22 bits (fp16 x fp16 products are exact in the
// fp32 accumulator), i.e. the accuracy of the fp32 multiply-add form this replaces (which spent ~700 VALU instructions per
// pixel and 8 channels: 74 us for the batch-8 UNet conv_in against ~8 us of output writes).
// A = weights (rows = output channels) from LDS, B = the pixel's 96 slots built in registers; D: lane = (pixel lane&15,
// channels 4*(lane>>4)..+3) -> one 8-byte store per 16x16 tile.  A wave owns 16 consecutive pixels of the flattened
// (b, y, x) order; workgroups walk the tiles persistently so that the weights are staged once.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_c4_kernel(const float* __restrict__ in, const float* __restrict__ pre_w,
                                                      const float* __restrict__ pre_b, float in_scale,
                                                      const half_t* __restrict__ W, const half_t* __restrict__ bias,
                                                      half_t* __restrict__ out, int B, int H, int Wd, int Cout) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS image: 16-byte chunk (sg, co) = the 8 K slots 8sg..8sg+7 of output channel co, at ((sg * Cout) + co) * 16:
    // the 16 lanes of a fragment read consecutive chunks
    for (int i = threadIdx.x; i < Cout * 12; i += 256) {
        const int sg = i / Cout, co = i - sg * Cout;
        h4 v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int s4 = sg * 2 + h;                       // 4-slot group: part = s4 / 9 (hi, lo, zero), tap = s4 % 9
            v[h] = (h4){0, 0, 0, 0};
            if (s4 < 18) v[h] = *reinterpret_cast<const h4*>(W + co * 36 + (s4 % 9) * 4);
        }
        h8 o = {v[0][0], v[0][1], v[0][2], v[0][3], v[1][0], v[1][1], v[1][2], v[1][3]};
        *reinterpret_cast<h8*>(smem + (sg * Cout + co) * 16) = o;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const int npix = B * H * Wd, ntile = (npix + 15) >> 4, plane = H * Wd;
    float pw[16], pb[4];
    if (pre_w) {
#pragma unroll
        for (int i = 0; i < 16; ++i) pw[i] = pre_w[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) pb[i] = pre_b[i];
    }
    for (int t = blockIdx.x * 4 + wave; t < ntile; t += gridDim.x * 4) {
        const int pix = t * 16 + n;
        const bool live = pix < npix;
        const int b = pix / plane, rem = pix - b * plane;
        const int y = rem / Wd, x = rem - y * Wd;
        h8 xf[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            half_t e[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int s4 = 8 * s + 2 * q + h;
                const int part = s4 >= 18 ? 2 : (s4 >= 9 ? 1 : 0), tap = s4 - 9 * (s4 >= 18 ? 2 : (s4 >= 9 ? 1 : 0));
                const int iy = y + tap / 3 - 1, ix = x + tap % 3 - 1;
                float z[4] = {0.f, 0.f, 0.f, 0.f};
                if (live && part < 2 && iy >= 0 && iy < H && ix >= 0 && ix < Wd) {
                    float r[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) r[c] = in[((b * 4 + c) * H + iy) * Wd + ix] * in_scale;
                    if (pre_w) {
#pragma unroll
                        for (int o = 0; o < 4; ++o)
                            z[o] = pb[o] + pw[o * 4] * r[0] + pw[o * 4 + 1] * r[1] + pw[o * 4 + 2] * r[2] + pw[o * 4 + 3] * r[3];
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) z[c] = r[c];
                    }
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const half_t hi = (half_t)z[c];
                    e[h * 4 + c] = part == 0 ? hi : (half_t)(z[c] - (float)hi);
                }
            }
            xf[s] = (h8){e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7]};
        }
        half_t* orow = out + (long long)pix * Cout + 4 * q;
        for (int ct = 0; ct < Cout; ct += 16) {
            f4 acc = {0.f, 0.f, 0.f, 0.f};
            if (bias) {
                const h4 b4 = *reinterpret_cast<const h4*>(bias + ct + 4 * q);
                acc = (f4){(float)b4[0], (float)b4[1], (float)b4[2], (float)b4[3]};
            }
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const h8 wf = *reinterpret_cast<const h8*>(smem + (((s * 4 + q) * Cout) + ct + n) * 16);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf[s], acc, 0, 0, 0);
            }
            if (live) {
                const h4 o = {(half_t)acc[0], (half_t)acc[1], (half_t)acc[2], (half_t)acc[3]};
                *reinterpret_cast<h4*>(orow + ct) = o;
            }
        }
    }
}

extern "C" int lcm_conv3x3_c4_f32in(const void* in, const void* pre_w, const void* pre_b, float in_scale,
                                    const void* W, const void* bias, void* out, int B, int H, int Wd, int Cout,
                                    void* stream) {
    LCM_REQUIRE(in && W && out, "conv_c4: null pointer");
    LCM_REQUIRE(B > 0 && H > 0 && Wd > 0 && Cout % 16 == 0 && Cout * 192 <= 160 * 1024, "conv_c4: bad shape (Cout %d)", Cout);
    LCM_REQUIRE((long long)B * 4 * H * Wd < (1ll << 31), "conv_c4: input too large");
    LCM_REQUIRE((pre_w == nullptr) == (pre_b == nullptr), "conv_c4: pre_w/pre_b must come together");
    const int smem = Cout * 192;
    static LcmDevOnce attr_once;
    if (auto once_guard = attr_once.first()) {
        once_guard.check(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_c4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    const long long ntile = ((long long)B * H * Wd + 15) / 16;
    // persistent: the weights are staged once per workgroup; at most two workgroups per CU
    const int grid = (int)((ntile + 3) / 4 < 512 ? (ntile + 3) / 4 : 512);
    hipLaunchKernelGGL(conv_c4_kernel, dim3(grid), dim3(256), smem, (hipStream_t)stream, (const float*)in,
                       (const float*)pre_w, (const float*)pre_b, in_scale, (const half_t*)W, (const half_t*)bias,
                       (half_t*)out, B, H, Wd, Cout);
    LCM_CHECK_LAUNCH("conv_c4");
    return LCM_OK;
}

// ---------------------------------------------------------------------------------------------
// conv3x3 to <= 4 output channels.  A pixel is shared by LPP = Cin/8 (<= 64, power of two or padded)
// lanes, each owning 8 input channels over the 9 taps; shuffle reduction inside the lane group.
// ---------------------------------------------------------------------------------------------
template <int LPP>
__global__ __launch_bounds__(256) void conv_smalln_kernel(const half_t* __restrict__ in, const half_t* __restrict__ W,
                                                          const half_t* __restrict__ bias, void* __restrict__ out,
                                                          float* __restrict__ out_f32, int B, int H, int Wd, int Cin,
                                                          int Cout, int mode) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* ws = reinterpret_cast<half_t*>(smem);   // [Cout][9][Cin]
    for (int i = threadIdx.x * 8; i < Cout * 9 * Cin; i += 256 * 8)
        *reinterpret_cast<h8*>(ws + i) = *reinterpret_cast<const h8*>(W + i);
    __syncthreads();
    constexpr int PPW = 256 / LPP;                  // pixels per workgroup pass
    const int sub = threadIdx.x % LPP, pl = threadIdx.x / LPP;
    const int ncc = Cin >> 3;
    const long long npix = (long long)B * H * Wd;
    for (long long pix0 = (long long)blockIdx.x * PPW; pix0 < npix; pix0 += (long long)gridDim.x * PPW) {
        const long long pix = pix0 + pl;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (pix < npix) {
            const int x = (int)(pix % Wd), y = (int)((pix / Wd) % H), b = (int)(pix / ((long long)H * Wd));
            for (int cc = sub; cc < ncc; cc += LPP) {
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int iy = y + tap / 3 - 1, ix = x + tap % 3 - 1;
                    if (iy < 0 || iy >= H || ix < 0 || ix >= Wd) continue;
                    h8 v = *reinterpret_cast<const h8*>(in + (((long long)b * H + iy) * Wd + ix) * Cin + cc * 8);
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        if (o < Cout) {
                            h8 w = *reinterpret_cast<const h8*>(ws + (o * 9 + tap) * Cin + cc * 8);
                            // v_dot2_f32_f16: two exact fp16 products + fp32 accumulate per instruction (the scalar form
                            // spends three VALU ops per product on conversions: 112 us for the 512^2 conv_out, VALU-bound)
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                acc[o] = __builtin_amdgcn_fdot2((h2){v[2 * j], v[2 * j + 1]}, (h2){w[2 * j], w[2 * j + 1]}, acc[o], false);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int off = LPP / 2; off > 0; off >>= 1) acc[o] += __shfl_xor(acc[o], off, 64);
        if (pix < npix && sub == 0) {
            for (int o = 0; o < Cout; ++o) {
                const float yv = acc[o] + (bias ? (float)bias[o] : 0.f);
                if (out_f32) out_f32[pix * Cout + o] = yv;
                if (mode == 0) {
                    reinterpret_cast<float*>(out)[pix * Cout + o] = yv;
                } else {
                    const float u = fminf(fmaxf(yv * 0.5f + 0.5f, 0.f), 1.f);
                    reinterpret_cast<unsigned char*>(out)[pix * Cout + o] = (unsigned char)rintf(u * 255.f);
                }
            }
        }
    }
}

// Row-walking form for Cin == 8 * LPP (the AutoencoderKL conv_out: 128 -> 3 at full resolution): a lane group owns a run of
// SEG pixels of one image row and slides the 3x3 window along it -- 3 new 16-byte loads per pixel instead of 9 -- with its
// 8-channel slice of the Cout x 9 weights held in registers instead of re-read from LDS per pixel (27 ds_read_b128 each).
// Same per-pixel arithmetic and order (taps 0..8, four v_dot2 each, then the lane-group reduction) as the kernel above.
template <int LPP>
__global__ __launch_bounds__(256) void conv_smalln_row_kernel(const half_t* __restrict__ in, const half_t* __restrict__ W,
                                                              const half_t* __restrict__ bias, void* __restrict__ out,
                                                              float* __restrict__ out_f32, int B, int H, int Wd, int Cout, int mode) {
    constexpr int Cin = LPP * 8, SEG = 32, GPW = 256 / LPP;
    const int sub = threadIdx.x % LPP, gl = threadIdx.x / LPP;
    h8 wr[4][9];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            wr[o][tap] = o < Cout ? *reinterpret_cast<const h8*>(W + (o * 9 + tap) * Cin + sub * 8) : z;
        }
    float bv[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) bv[o] = (bias && o < Cout) ? (float)bias[o] : 0.f;
    const int segs_x = (Wd + SEG - 1) / SEG;
    const long long nseg = (long long)B * H * segs_x;
    for (long long sg = (long long)blockIdx.x * GPW + gl; sg < nseg; sg += (long long)gridDim.x * GPW) {
        const int sx = (int)(sg % segs_x), y = (int)((sg / segs_x) % H), b = (int)(sg / ((long long)segs_x * H));
        const int x0 = sx * SEG, x1 = min(Wd, x0 + SEG);
        const half_t* base = in + ((long long)b * H * Wd) * Cin + sub * 8;
        auto load_col = [&](int x, h8 (&c)[3]) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int iy = y + dy - 1;
                h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                c[dy] = (x >= 0 && x < Wd && iy >= 0 && iy < H) ? *reinterpret_cast<const h8*>(base + ((long long)iy * Wd + x) * Cin) : z;
            }
        };
        h8 c0[3], c1[3], c2[3];
        load_col(x0 - 1, c0);
        load_col(x0, c1);
        for (int x = x0; x < x1; ++x) {
            load_col(x + 1, c2);
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3, dx = tap % 3;
                const h8 v = dx == 0 ? c0[dy] : dx == 1 ? c1[dy] : c2[dy];
#pragma unroll
                for (int o = 0; o < 4; ++o)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[o] = __builtin_amdgcn_fdot2((h2){v[2 * j], v[2 * j + 1]}, (h2){wr[o][tap][2 * j], wr[o][tap][2 * j + 1]}, acc[o], false);
            }
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int off = LPP / 2; off > 0; off >>= 1) acc[o] += __shfl_xor(acc[o], off, 64);
            if (sub == 0) {
                const long long pix = ((long long)b * H + y) * Wd + x;
                for (int o = 0; o < Cout; ++o) {
                    const float yv = acc[o] + bv[o];
                    if (out_f32) out_f32[pix * Cout + o] = yv;
                    if (mode == 0) {
                        reinterpret_cast<float*>(out)[pix * Cout + o] = yv;
                    } else {
                        const float u = fminf(fmaxf(yv * 0.5f + 0.5f, 0.f), 1.f);
                        reinterpret_cast<unsigned char*>(out)[pix * Cout + o] = (unsigned char)rintf(u * 255.f);
                    }
                }
            }
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) { c0[dy] = c1[dy]; c1[dy] = c2[dy]; }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// conv3x3 to <= 4 channels on the matrix pipe (Cin % 64 == 0): UNet conv_out (320 -> 4, fp32 eps) and AutoencoderKL
// decoder.conv_out (128 -> 3, RGB8), optionally with the GroupNorm-apply + SiLU in front of it fused into the staging pass
// (scale / shift tables per (image, channel), rounded to fp16 exactly as gn_apply_kernel rounds): the normalised 512 x 512 x 128
// tensor (0.5 GB at batch 8) is neither written nor read back.
// One workgroup = an 8 x 16 pixel patch; per 64-channel chunk the 10 x 18 halo goes to LDS (128-byte rows, XOR-swizzled as in
// conv_halo.hip) and serves all 9 taps.  v_mfma_f32_16x16x32_f16 with A = the weights (rows = output channels; rows >= Cout
// are zero registers, never read), B = 16 pixels of one patch row: the 4 outputs of a pixel land in the lanes 0..15 of the
// wave.  13 of the 16 MFMA rows are padding -- still 5x fewer issue cycles than the 144 v_dot2 per pixel and 16 lanes of the
// VALU form (483 us at batch 8 for 0.5 GB of input: 1.1 TB/s), and the kernel becomes what it should be, a read of its input.
// Workgroups walk the patches persistently; the weights ([4][9][Cin] fp16) are staged once per workgroup.
// ---------------------------------------------------------------------------------------------
static __device__ __attribute__((aligned(256))) half_t g_zero_page_m[128];

template <int XFORM, int TW>
__global__ __launch_bounds__(256) void conv_fewout_kernel(const half_t* __restrict__ in, const float* __restrict__ gn_scale,
                                                          const float* __restrict__ gn_shift, int silu,
                                                          const half_t* __restrict__ W, const half_t* __restrict__ bias,
                                                          void* __restrict__ out, float* __restrict__ out_f32,
                                                          int B, int H, int Wd, int Cin, int Cout, int mode) {
    // TW = 16: 8 x 16 patch, a wave owns two patch rows (two 16-pixel MFMA columns); TW = 8 (images up to 64 pixels wide: four
    // times the workgroups of a 64 x 64 latent): 8 x 8 patch, a wave owns one MFMA column of 2 rows x 8 pixels
    constexpr int TH = 8, HWD = TW + 2, HROWS = (TH + 2) * HWD, HROWS_PAD = (HROWS + 7) / 8 * 8;
    constexpr int NV = (HROWS_PAD * 8 + 255) / 256, NJ = TW / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xs = smem;                                   // halo of one 64-channel chunk
    half_t* wl = reinterpret_cast<half_t*>(smem + HROWS_PAD * 128);      // [4][9][Cin]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4, pos = tid & 7;
    for (int i = tid * 8; i < 36 * Cin; i += 256 * 8) {
        const int o = i / (9 * Cin);
        h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (o < Cout) v = *reinterpret_cast<const h8*>(W + i);
        *reinterpret_cast<h8*>(wl + i) = v;
    }
    float bv[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) bv[o] = (bias && o < Cout) ? (float)bias[o] : 0.f;
    const int tiles_x = (Wd + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int ntile = B * tiles_y * tiles_x, nchunk = Cin >> 6;
    const int x_lds0 = (tid >> 3) * 128 + ((pos ^ ((tid >> 3) & 7)) << 4);
    bool first = true;
    for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
        const int bimg = t / (tiles_y * tiles_x), tr = t - bimg * tiles_y * tiles_x;
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
        const int y0 = ty * TH, x0 = tx * TW;
        int h_pix[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int hr = (tid + 256 * i) >> 3;
            const int hy = hr / HWD, hx = hr - hy * HWD;
            const int ly = y0 - 1 + hy, lx = x0 - 1 + hx;
            const bool ok = hr < HROWS && ly >= 0 && ly < H && lx >= 0 && lx < Wd;
            h_pix[i] = ok ? (bimg * H + ly) * Wd + lx : -1;
        }
        f4 acc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] = (f4){0.f, 0.f, 0.f, 0.f};
        const int py0 = TW == 16 ? 2 * wave : 2 * wave + (n >> 3), px = TW == 16 ? n : (n & 7);     // this lane's pixel (n-tile 0)
        for (int c64 = 0; c64 < nchunk; ++c64) {
            if (!first) __syncthreads();               // every wave is past its reads of the previous chunk (and of the weights' staging)
            first = false;
            const int cb = c64 << 6;
            if constexpr (XFORM != 0) {
                const float* sc = gn_scale + (long long)bimg * Cin + cb + pos * 8;
                const float* sh = gn_shift + (long long)bimg * Cin + cb + pos * 8;
                float s8[8], t8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { s8[j] = sc[j]; t8[j] = sh[j]; }
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    if (tid + 256 * i < HROWS_PAD * 8) {
                        h8 o = {0, 0, 0, 0, 0, 0, 0, 0};
                        if (h_pix[i] >= 0) {
                            const h8 v = *reinterpret_cast<const h8*>(in + (long long)h_pix[i] * Cin + cb + pos * 8);
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                float f = __builtin_fmaf((float)v[j], s8[j], t8[j]);      // as gn_apply_kernel rounds
                                if (silu) f = silu_f(f);
                                o[j] = (half_t)f;
                            }
                        }
                        *reinterpret_cast<h8*>(xs + x_lds0 + i * 32 * 128) = o;
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    if ((wave + 4 * i) * 64 < HROWS_PAD * 8) {       // wave-uniform: whole 1 KiB piece in range
                        const int hr = (tid + 256 * i) >> 3;
                        const half_t* src = h_pix[i] >= 0 ? in + (long long)h_pix[i] * Cin + cb + ((pos ^ (hr & 7)) << 3) : g_zero_page_m;
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                         (__attribute__((address_space(3))) void*)(xs + (wave + 4 * i) * 1024), 16, 0, 0);
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    h8 wf = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (n < 4) wf = *reinterpret_cast<const h8*>(wl + (n * 9 + tap) * Cin + cb + kk * 32 + q * 8);
                    const int c = kk * 4 + q;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int r = (py0 + j + dy) * HWD + px + dx;
                        const h8 xf = *reinterpret_cast<const h8*>(xs + r * 128 + ((c ^ (r & 7)) << 4));
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf, acc[j], 0, 0, 0);
                    }
                }
            }
        }
        if (q == 0) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int y = y0 + py0 + j, x = x0 + px;
                if (y < H && x < Wd) {
                    const long long pix = ((long long)bimg * H + y) * Wd + x;
                    for (int o = 0; o < Cout; ++o) {
                        const float yv = acc[j][o] + bv[o];
                        if (out_f32) out_f32[pix * Cout + o] = yv;
                        if (mode == 0) {
                            reinterpret_cast<float*>(out)[pix * Cout + o] = yv;
                        } else {
                            const float u = fminf(fmaxf(yv * 0.5f + 0.5f, 0.f), 1.f);
                            reinterpret_cast<unsigned char*>(out)[pix * Cout + o] = (unsigned char)rintf(u * 255.f);
                        }
                    }
                }
            }
        }
    }
}

static int launch_conv_fewout(const void* in, const void* gn_scale, const void* gn_shift, int silu, const void* W, const void* bias,
                              void* out, void* out_f32, int B, int H, int Wd, int Cin, int Cout, int mode, hipStream_t s) {
    const int tw = Wd <= 64 ? 8 : 16;                   // patch width: a function of the image, never of the batch
    const int smem = (tw == 16 ? 184 : 104) * 128 + 36 * Cin * 2;
    LCM_REQUIRE(smem <= 160 * 1024, "conv_smalln: weights %d bytes exceed LDS", 36 * Cin * 2);
    static LcmDevOnce attr_once;
    if (auto once_guard = attr_once.first()) {
        once_guard.check(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_fewout_kernel<0, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        once_guard.check(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_fewout_kernel<1, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        once_guard.check(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_fewout_kernel<0, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        once_guard.check(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_fewout_kernel<1, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    const long long ntile = (long long)B * ((H + 7) / 8) * ((Wd + tw - 1) / tw);
    const int per_cu = 160 * 1024 / smem < 4 ? 160 * 1024 / smem : 4;
    const int grid = (int)(ntile < 256 * per_cu ? ntile : 256 * per_cu);
#define FEWOUT(X, T)                                                                                                          \
    hipLaunchKernelGGL((conv_fewout_kernel<X, T>), dim3(grid), dim3(256), smem, s, (const half_t*)in, (const float*)gn_scale, \
                       (const float*)gn_shift, silu, (const half_t*)W, (const half_t*)bias, out, (float*)out_f32, B, H, Wd, Cin, Cout, mode)
    if (gn_scale) { if (tw == 16) FEWOUT(1, 16); else FEWOUT(1, 8); }
    else { if (tw == 16) FEWOUT(0, 16); else FEWOUT(0, 8); }
#undef FEWOUT
    LCM_CHECK_LAUNCH("conv_fewout");
    return LCM_OK;
}

extern "C" int lcm_conv3x3_smalln(const void* in, const void* W, const void* bias, void* out, void* out_f32, int B,
                                  int H, int Wd, int Cin, int Cout, int mode, void* stream);

extern "C" int lcm_conv3x3_smalln_gn(const void* in, const void* gn_scale, const void* gn_shift, int silu, const void* W,
                                     const void* bias, void* out, void* out_f32, int B, int H, int Wd, int Cin, int Cout,
                                     int mode, void* stream) {
    LCM_REQUIRE(in && W && out, "conv_smalln: null pointer");
    LCM_REQUIRE(B > 0 && H > 0 && Wd > 0 && Cin % 8 == 0 && Cout >= 1 && Cout <= 4, "conv_smalln: bad shape");
    LCM_REQUIRE(mode == 0 || mode == 1, "conv_smalln: bad mode");
    LCM_REQUIRE((gn_scale == nullptr) == (gn_shift == nullptr), "conv_smalln: gn_scale / gn_shift must come together");
    LCM_REQUIRE((long long)B * H * Wd < (1ll << 31), "conv_smalln: too many pixels");
    if (Cin % 64 == 0)
        return launch_conv_fewout(in, gn_scale, gn_shift, silu, W, bias, out, out_f32, B, H, Wd, Cin, Cout, mode, (hipStream_t)stream);
    LCM_REQUIRE(!gn_scale, "conv_smalln: the GroupNorm-fused form needs Cin %% 64 == 0 (Cin %d)", Cin);
    return lcm_conv3x3_smalln(in, W, bias, out, out_f32, B, H, Wd, Cin, Cout, mode, stream);
}

extern "C" int lcm_conv3x3_smalln(const void* in, const void* W, const void* bias, void* out, void* out_f32, int B,
                                  int H, int Wd, int Cin, int Cout, int mode, void* stream) {
    LCM_REQUIRE(in && W && out, "conv_smalln: null pointer");
    LCM_REQUIRE(B > 0 && H > 0 && Wd > 0 && Cin % 8 == 0 && Cout >= 1 && Cout <= 4, "conv_smalln: bad shape");
    LCM_REQUIRE(mode == 0 || mode == 1, "conv_smalln: bad mode");
    if (Cin % 64 == 0 && (long long)B * H * Wd < (1ll << 31))     // the kernel choice is a function of Cin alone (never of the batch)
        return launch_conv_fewout(in, nullptr, nullptr, 0, W, bias, out, out_f32, B, H, Wd, Cin, Cout, mode, (hipStream_t)stream);
    const int smem = Cout * 9 * Cin * 2;
    LCM_REQUIRE(smem <= 64 * 1024, "conv_smalln: weights %d bytes exceed LDS budget", smem);
    const long long npix = (long long)B * H * Wd;
    hipStream_t s = (hipStream_t)stream;
    const int ncc = Cin / 8;
#define LAUNCH_SN(L)                                                                                              \
    do {                                                                                                          \
        const long long wg = (npix + (256 / L) - 1) / (256 / L);                                                  \
        hipLaunchKernelGGL((conv_smalln_kernel<L>), dim3((int)(wg < 8192 ? wg : 8192)), dim3(256), smem, s,       \
                           (const half_t*)in, (const half_t*)W, (const half_t*)bias, out, (float*)out_f32, B, H,  \
                           Wd, Cin, Cout, mode);                                                                  \
    } while (0)
#define LAUNCH_ROW(L)                                                                                             \
    do {                                                                                                          \
        const long long nseg = (long long)B * H * ((Wd + 31) / 32), wg = (nseg + (256 / L) - 1) / (256 / L);      \
        hipLaunchKernelGGL((conv_smalln_row_kernel<L>), dim3((int)(wg < 16384 ? wg : 16384)), dim3(256), 0, s,    \
                           (const half_t*)in, (const half_t*)W, (const half_t*)bias, out, (float*)out_f32, B, H,  \
                           Wd, Cout, mode);                                                                       \
    } while (0)
    if (ncc == 16) LAUNCH_ROW(16);
    else if (ncc == 32) LAUNCH_ROW(32);
    else if (ncc <= 16) LAUNCH_SN(16); else if (ncc <= 32) LAUNCH_SN(32); else LAUNCH_SN(64);
#undef LAUNCH_ROW
#undef LAUNCH_SN
    LCM_CHECK_LAUNCH("conv_smalln");
    return LCM_OK;
}

// ---------------------------------------------------------------------------------------------
// small-M linear: one wave per output feature n, lanes stride K in 16-byte chunks; rows in blocks of 16 (the weight row is
// re-read from L1 per block).  Row m reads x[m % x_rows] and res[m % res_rows]: the time-embedding MLP of ALL sampler steps
// runs as one launch per layer (rows = step-major (step, image)) with the per-request inputs (guidance embedding, SDXL
// added embedding) given once.  A row's fp32 summation order does not depend on M.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void linear_smallm_kernel(const half_t* __restrict__ x, int ldx, int x_rows,
                                                            const half_t* __restrict__ W, const half_t* __restrict__ bias,
                                                            const half_t* __restrict__ res, int ldr, int res_rows,
                                                            half_t* __restrict__ out, int ldo, int M, int N, int K,
                                                            int silu_in, int silu_out) {
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const half_t* wr = W + (long long)n * K;
    for (int m0 = 0; m0 < M; m0 += 16) {
        float acc[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) acc[m] = 0.f;
        for (int k = lane * 8; k < K; k += 512) {
            h8 w = *reinterpret_cast<const h8*>(wr + k);
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                if (m0 + m < M) {
                    h8 xv = *reinterpret_cast<const h8*>(x + (long long)((m0 + m) % x_rows) * ldx + k);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float f = (float)xv[j];
                        if (silu_in) f = silu_f(f);
                        acc[m] += f * (float)w[j];
                    }
                }
            }
        }
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (m0 + m < M) {
                float v = wave_sum(acc[m]);
                if (lane == 0) {
                    if (bias) v += (float)bias[n];
                    if (res) v += (float)res[(long long)((m0 + m) % res_rows) * ldr + n];
                    if (silu_out) v = silu_f(v);
                    out[(long long)(m0 + m) * ldo + n] = (half_t)v;
                }
            }
        }
    }
}

extern "C" int lcm_linear_rows_f16(const void* x, int ldx, int x_rows, const void* W, const void* bias, const void* res, int ldr,
                                   int res_rows, void* out, int ldo, int M, int N, int K, int silu_in, int silu_out, void* stream) {
    LCM_REQUIRE(x && W && out, "linear_rows: null pointer");
    LCM_REQUIRE(M >= 1 && M <= 4096 && N > 0 && K > 0 && K % 8 == 0 && ldx % 8 == 0 && x_rows >= 1 && (!res || res_rows >= 1),
                "linear_rows: bad shape M=%d N=%d K=%d x_rows=%d res_rows=%d", M, N, K, x_rows, res_rows);
    hipLaunchKernelGGL(linear_smallm_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, ldx, x_rows,
                       (const half_t*)W, (const half_t*)bias, (const half_t*)res, ldr, res ? res_rows : 1, (half_t*)out, ldo, M, N, K,
                       silu_in, silu_out);
    LCM_CHECK_LAUNCH("linear_rows");
    return LCM_OK;
}

extern "C" int lcm_linear_smallm_f16(const void* x, int ldx, const void* W, const void* bias, const void* res, int ldr,
                                     void* out, int ldo, int M, int N, int K, int silu_in, int silu_out, void* stream) {
    LCM_REQUIRE(M >= 1 && M <= 16, "linear_smallm: bad shape M=%d N=%d K=%d", M, N, K);
    return lcm_linear_rows_f16(x, ldx, M, W, bias, res, ldr, M, out, ldo, M, N, K, silu_in, silu_out, stream);
}

// ---------------------------------------------------------------------------------------------
struct StepTimes { float t[64]; };
__global__ void timestep_embedding_kernel(StepTimes ts, half_t* __restrict__ out, int nsteps, int B, int dim) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // rows step-major: (step, image)
    const int half = dim >> 1;
    if (i >= nsteps * B * dim) return;
    const int j = i % dim;
    const float t = ts.t[i / (B * dim)];
    const int k = j < half ? j : j - half;
    const float f = expf(-9.210340371976184f * (float)k / (float)half);   // ln(10000)
    const float a = t * f;
    out[i] = (half_t)(j < half ? cosf(a) : sinf(a));
}

extern "C" int lcm_timestep_embedding_steps(const float* t_host, int nsteps, void* out, int B, int dim, void* stream) {
    LCM_REQUIRE(t_host && out && B > 0 && dim > 0 && dim % 2 == 0 && nsteps >= 1 && nsteps <= 64, "timestep_embedding: bad shape (steps %d)", nsteps);
    StepTimes ts;
    for (int i = 0; i < 64; ++i) ts.t[i] = i < nsteps ? t_host[i] : 0.f;
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((nsteps * B * dim + 255) / 256), dim3(256), 0, (hipStream_t)stream, ts,
                       (half_t*)out, nsteps, B, dim);
    LCM_CHECK_LAUNCH("timestep_embedding");
    return LCM_OK;
}

extern "C" int lcm_timestep_embedding(float t, void* out, int B, int dim, void* stream) {
    return lcm_timestep_embedding_steps(&t, 1, out, B, dim, stream);
}

// ---------------------------------------------------------------------------------------------
struct StepCoef { float sa, sb, c_skip, c_out, sap, sbp; };

__global__ void scheduler_step_kernel(const float* __restrict__ eps, const float* __restrict__ eps_u, float guidance,
                                      float* __restrict__ lat, const float* __restrict__ noise, StepCoef c, int last,
                                      int B, int h, int w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // NCHW index
    const int hw = h * w, n = B * 4 * hw;
    if (i >= n) return;
    const int pix = i % hw, ch = (i / hw) & 3, b = i / (4 * hw);
    const long long e = ((long long)b * hw + pix) * 4 + ch;   // NHWC
    float ev = eps[e];
    if (eps_u) { const float u = eps_u[e]; ev = u + guidance * (ev - u); }
    const float x = lat[i];
    const float x0 = (x - c.sb * ev) / c.sa;
    const float den = c.c_out * x0 + c.c_skip * x;
    lat[i] = last ? den : c.sap * den + c.sbp * noise[i];
}

extern "C" int lcm_scheduler_step(const void* eps, const void* eps_uncond, float guidance, void* lat, const void* noise,
                                  const float* coef6, int last, int B, int h, int w, void* stream) {
    LCM_REQUIRE(eps && lat && coef6 && (last || noise), "scheduler_step: null pointer");
    LCM_REQUIRE(B > 0 && h > 0 && w > 0, "scheduler_step: bad shape");
    StepCoef c = {coef6[0], coef6[1], coef6[2], coef6[3], coef6[4], coef6[5]};
    const int n = B * 4 * h * w;
    hipLaunchKernelGGL(scheduler_step_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const float*)eps, (const float*)eps_uncond, guidance, (float*)lat, (const float*)noise, c, last,
                       B, h, w);
    LCM_CHECK_LAUNCH("scheduler_step");
    return LCM_OK;
}

// ---------------------------------------------------------------------------------------------
__global__ void latents_pool8_kernel(const float* __restrict__ lat, half_t* __restrict__ out, int B, int h, int w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // (b, c, oy, ox)
    if (i >= B * 4 * 64) return;
    const int ox = i & 7, oy = (i >> 3) & 7, bc = i >> 6;
    // adaptive_avg_pool2d bins: [floor(o*n/8), ceil((o+1)*n/8))
    const int y0 = (oy * h) / 8, y1 = ((oy + 1) * h + 7) / 8;
    const int x0 = (ox * w) / 8, x1 = ((ox + 1) * w + 7) / 8;
    float s = 0.f;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) s += lat[((long long)bc * h + y) * w + x];
    out[i] = (half_t)(s / (float)((y1 - y0) * (x1 - x0)));
}

extern "C" int lcm_latents_pool8(const void* lat, void* out_f16, int B, int h, int w, void* stream) {
    LCM_REQUIRE(lat && out_f16 && B > 0 && h > 0 && w > 0, "latents_pool8: bad args");
    hipLaunchKernelGGL(latents_pool8_kernel, dim3((B * 256 + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const float*)lat, (half_t*)out_f16, B, h, w);
    LCM_CHECK_LAUNCH("latents_pool8");
    return LCM_OK;
}

// ---------------------------------------------------------------------------------------------
// [R][C] -> [C][R] tiled transpose through LDS (64x64 tiles, padded rows).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const half_t* __restrict__ in, int ldi, half_t* __restrict__ out,
                                                        int ldo, int R, int C, long long stride_in, long long stride_out) {
    __shared__ half_t tile[64][66];
    const half_t* ib = in + blockIdx.z * stride_in;
    half_t* ob = out + blockIdx.z * stride_out;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? ib[(long long)r * ldi + c] : (half_t)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < R) ob[(long long)c * ldo + r] = tile[tx][i];
    }
}

extern "C" int lcm_transpose_f16(const void* in, int ldi, void* out, int ldo, int R, int C, int batch, int64_t stride_in,
                                 int64_t stride_out, void* stream) {
    LCM_REQUIRE(in && out && R > 0 && C > 0 && batch > 0, "transpose: bad args");
    hipLaunchKernelGGL(transpose_kernel, dim3((C + 63) / 64, (R + 63) / 64, batch), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)in, ldi, (half_t*)out, ldo, R, C, (long long)stride_in, (long long)stride_out);
    LCM_CHECK_LAUNCH("transpose");
    return LCM_OK;
}

// ---------------------------------------------------------------------------------------------
// Profiling aid: occupy the stream for `usec` microseconds (one wave polling the 100 MHz realtime counter,
// bounded by construction) so that a host-bound eager enqueue queues up behind it and per-launch HIP events
// then time back-to-back GPU execution rather than host launch latency.  Not on the product path.
// ---------------------------------------------------------------------------------------------
__global__ void spin_kernel(unsigned long long ticks, unsigned long long* sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t = t0;
    // hard bound on iterations as well as on time: every wave exits
    for (long long i = 0; i < (1ll << 34) && (t - t0) < ticks; ++i) {
        __builtin_amdgcn_s_sleep(64);
        t = __builtin_amdgcn_s_memrealtime();
    }
    if (sink && threadIdx.x == 0) *sink = t - t0;
}

extern "C" int lcm_debug_spin(int usec, void* stream) {
    LCM_REQUIRE(usec >= 0 && usec <= 2000000, "debug_spin: usec out of range");
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long)usec * 100ull,
                       (unsigned long long*)nullptr);
    LCM_CHECK_LAUNCH("debug_spin");
    return LCM_OK;
}

// ---------------------------------------------------------------------------------------------
// Measurement only (tools/seam_cost.py): what ONE grid-wide seam costs inside a launch on this chip -- the price a cooperative
// "whole transformer block in one kernel" would pay per GEMM -> GEMM dependency instead of a kernel boundary.  n_barriers
// rounds of: every workgroup writes a 128-byte record (plain stores), lane 0 releases (agent scope), arrives on ONE monotonic
// counter, polls it relaxed with s_sleep -- BOUNDED: after `spin_limit` polls the workgroup gives up, sets the timeout word and
// every later round falls through, so the grid always drains -- then acquires and every wave re-reads a record of another
// workgroup.  The grid must be co-resident (<= 256 workgroups of 256 threads: one per CU).  Never used by the product path.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grid_barrier_probe_kernel(unsigned* counter, unsigned* timeout, float* records, int n_barriers,
                                                                 unsigned spin_limit, float* sink) {
    const int wg = blockIdx.x, nwg = gridDim.x;
    float acc = 0.f;
    for (int r = 0; r < n_barriers; ++r) {
        if (threadIdx.x < 32) records[(long long)wg * 32 + threadIdx.x] = (float)(r + wg + threadIdx.x);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(r + 1) * (unsigned)nwg;
            unsigned spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (__hip_atomic_load(timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || ++spins > spin_limit) {
                    __hip_atomic_store(timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        acc += records[(long long)((wg + 1 + r) % nwg) * 32 + (threadIdx.x & 31)];
    }
    if (sink && acc == -1.f) *sink = acc;
}

extern "C" int lcm_debug_grid_barrier(int workgroups, int n_barriers, void* state, void* stream) {
    // state: >= 8 + workgroups * 128 bytes of device memory; words 0 / 1 = counter / timeout (zeroed here, every call)
    LCM_REQUIRE(workgroups >= 1 && workgroups <= 256 && n_barriers >= 0 && n_barriers <= 4096 && state, "debug_grid_barrier: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(state, 0, 16, s) != hipSuccess) { lcm_set_error("debug_grid_barrier: memset failed"); return LCM_EINVAL; }
    hipLaunchKernelGGL(grid_barrier_probe_kernel, dim3(workgroups), dim3(256), 0, s, (unsigned*)state, (unsigned*)state + 1,
                       (float*)((char*)state + 16), n_barriers, 2000000u, (float*)nullptr);
    LCM_CHECK_LAUNCH("debug_grid_barrier");
    return LCM_OK;
}

// ---------------------------------------------------------------------------------------------
// CLIPTextEmbeddings: out[b*S + s][:] = token_embedding[ids[b*S + s]][:] + position_embedding[s][:]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_tokens_kernel(const int* __restrict__ ids, const half_t* __restrict__ tok,
                                                           const half_t* __restrict__ pos, half_t* __restrict__ out,
                                                           int rows, int S, int D, int vocab) {
    const int ncc = D >> 3;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < (long long)rows * ncc; i += (long long)gridDim.x * 256) {
        const int r = (int)(i / ncc), c = (int)(i - (long long)r * ncc) * 8;
        int id = ids[r];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        h8 a = *reinterpret_cast<const h8*>(tok + (long long)id * D + c);
        h8 b = *reinterpret_cast<const h8*>(pos + (long long)(r % S) * D + c);
        h8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (half_t)((float)a[j] + (float)b[j]);
        *reinterpret_cast<h8*>(out + (long long)r * D + c) = o;
    }
}

extern "C" int lcm_embed_tokens_f16(const void* ids, const void* tok_emb, const void* pos_emb, void* out, int B, int S,
                                    int D, int vocab, void* stream) {
    LCM_REQUIRE(ids && tok_emb && pos_emb && out, "embed_tokens: null pointer");
    LCM_REQUIRE(B > 0 && S > 0 && D > 0 && D % 8 == 0 && vocab > 0, "embed_tokens: bad shape");
    const long long total = (long long)B * S * (D / 8);
    hipLaunchKernelGGL(embed_tokens_kernel, dim3((unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048)), dim3(256), 0,
                       (hipStream_t)stream, (const int*)ids, (const half_t*)tok_emb, (const half_t*)pos_emb, (half_t*)out, B * S,
                       S, D, vocab);
    LCM_CHECK_LAUNCH("embed_tokens");
    return LCM_OK;
}

// ---------------------------------------------------------------------------------------------
// out = base + alpha * delta (fp16 in/out, fp32 math): LoRA style merge W' = W + weight * (B A)
// (pipe.set_adapters / disable_lora of the reference, backends/cuda_worker.py:165-196), written in place into the
// live weight tensor so captured graphs keep their pointers.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void axpy_f16_kernel(const half_t* __restrict__ base, const half_t* __restrict__ delta,
                                                       float alpha, half_t* __restrict__ out, long long n8) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        h8 b = *reinterpret_cast<const h8*>(base + i * 8);
        h8 d = *reinterpret_cast<const h8*>(delta + i * 8);
        h8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (half_t)((float)b[j] + alpha * (float)d[j]);
        *reinterpret_cast<h8*>(out + i * 8) = o;
    }
}

extern "C" int lcm_axpy_f16(const void* base, const void* delta, float alpha, void* out, int64_t n, void* stream) {
    LCM_REQUIRE(base && delta && out && n > 0 && n % 8 == 0, "axpy: bad args (n must be a multiple of 8)");
    const long long n8 = n / 8;
    const long long wg = (n8 + 255) / 256;
    hipLaunchKernelGGL(axpy_f16_kernel, dim3((unsigned)(wg < 4096 ? wg : 4096)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)base, (const half_t*)delta, alpha, (half_t*)out, n8);
    LCM_CHECK_LAUNCH("axpy");
    return LCM_OK;
}

// ---------------------------------------------------------------------------------------------
// Constants of a LayerNorm folded into its consumer GEMM (lcm_gemm_ln_f16), refreshed after the live weight W' changed
// (style LoRA re-merge): g[n] = sum_k W'[n][k] from the fp16 values the kernel will multiply with, c[n] = c_base[n] +
// alpha * c_delta[n].  One wave per output feature.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ln_fold_refresh_kernel(const half_t* __restrict__ W, int N, int K,
                                                              const float* __restrict__ c_base, const float* __restrict__ c_delta,
                                                              float alpha, float* __restrict__ g_out, float* __restrict__ c_out) {
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const half_t* wr = W + (long long)n * K;
    float s = 0.f;
    for (int k = lane * 8; k < K; k += 512) {
        h8 w = *reinterpret_cast<const h8*>(wr + k);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += (float)w[j];
    }
    s = wave_sum(s);
    if (lane == 0) {
        g_out[n] = s;
        if (c_out) c_out[n] = c_base[n] + (c_delta ? alpha * c_delta[n] : 0.f);
    }
}

extern "C" int lcm_ln_fold_refresh(const void* W, int N, int K, const void* c_base, const void* c_delta, float alpha,
                                   void* g_out, void* c_out, void* stream) {
    LCM_REQUIRE(W && g_out && N > 0 && K > 0 && K % 8 == 0, "ln_fold_refresh: bad args");
    LCM_REQUIRE(!c_out || c_base, "ln_fold_refresh: c_out needs c_base");
    hipLaunchKernelGGL(ln_fold_refresh_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const half_t*)W, N, K,
                       (const float*)c_base, (const float*)c_delta, alpha, (float*)g_out, (float*)c_out);
    LCM_CHECK_LAUNCH("ln_fold_refresh");
    return LCM_OK;
}

// ---------------------------------------------------------------------------------------------
// AutoencoderKL.tiled_decode glue (vae.enable_tiling(), backends/cuda_worker.py:91): linear blends of
// overlapping decoded tiles (float pixel-major [B,h,w,3]) and the crop + place + RGB8 conversion of the result.
// ---------------------------------------------------------------------------------------------
__global__ void vae_blend_kernel(const float* __restrict__ a, int ah, int aw, float* __restrict__ b, int bh, int bw,
                                 int B, int extent, int vertical) {
    // vertical: b[y][x] = a[ah-extent+y][x] * (1 - y/extent) + b[y][x] * (y/extent), y < extent, x < bw (== aw)
    // horizontal: b[y][x] = a[y][aw-extent+x] * (1 - x/extent) + b[y][x] * (x/extent), x < extent, y < bh (== ah)
    const int n = vertical ? extent * bw : bh * extent;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * n * 3) return;
    const int c = i % 3, p = (i / 3) % n, bi = i / (3 * n);
    int y, x, ya, xa;
    float t;
    if (vertical) { y = p / bw; x = p - y * bw; ya = ah - extent + y; xa = x; t = (float)y / (float)extent; }
    else { y = p / extent; x = p - y * extent; ya = y; xa = aw - extent + x; t = (float)x / (float)extent; }
    const float av = a[(((long long)bi * ah + ya) * aw + xa) * 3 + c];
    float* bp = b + (((long long)bi * bh + y) * bw + x) * 3 + c;
    *bp = av * (1.0f - t) + *bp * t;
}

extern "C" int lcm_vae_blend_f32(const void* a, int ah, int aw, void* b, int bh, int bw, int B, int extent, int vertical,
                                 void* stream) {
    LCM_REQUIRE(a && b && B > 0 && extent > 0, "vae_blend: bad args");
    LCM_REQUIRE(vertical ? (aw == bw && extent <= ah && extent <= bh) : (ah == bh && extent <= aw && extent <= bw),
                "vae_blend: tile shapes %dx%d / %dx%d do not match extent %d", ah, aw, bh, bw, extent);
    const int n = B * (vertical ? extent * bw : bh * extent) * 3;
    hipLaunchKernelGGL(vae_blend_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)a, ah, aw,
                       (float*)b, bh, bw, B, extent, vertical);
    LCM_CHECK_LAUNCH("vae_blend");
    return LCM_OK;
}

__global__ void vae_place_tile_kernel(const float* __restrict__ tile, int th, int tw, unsigned char* __restrict__ out,
                                      float* __restrict__ out_f32, int H, int W, int B, int oy, int ox, int ch, int cw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * ch * cw * 3) return;
    const int c = i % 3, x = (i / 3) % cw, y = (i / (3 * cw)) % ch, bi = i / (3 * cw * ch);
    const float v = tile[(((long long)bi * th + y) * tw + x) * 3 + c];
    const long long o = (((long long)bi * H + oy + y) * W + ox + x) * 3 + c;
    if (out_f32) out_f32[o] = v;
    const float u = fminf(fmaxf(v * 0.5f + 0.5f, 0.f), 1.f);
    out[o] = (unsigned char)rintf(u * 255.f);
}

extern "C" int lcm_vae_place_tile(const void* tile, int th, int tw, void* out_u8, void* out_f32, int H, int W, int B,
                                  int oy, int ox, int ch, int cw, void* stream) {
    LCM_REQUIRE(tile && out_u8 && B > 0, "vae_place_tile: bad args");
    LCM_REQUIRE(ch <= th && cw <= tw && oy >= 0 && ox >= 0 && oy + ch <= H && ox + cw <= W, "vae_place_tile: crop outside bounds");
    const int n = B * ch * cw * 3;
    hipLaunchKernelGGL(vae_place_tile_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)tile, th,
                       tw, (unsigned char*)out_u8, (float*)out_f32, H, W, B, oy, ox, ch, cw);
    LCM_CHECK_LAUNCH("vae_place_tile");
    return LCM_OK;
}
