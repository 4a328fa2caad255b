// the VAE's 512^2 x 128 -> 128 3x3 convolution main loop with v_mfma_f32_16x16x32_f16 (the shipped shape)
// against v_mfma_f32_32x32x16_f16 AT THE SAME 64 x 64 OUTPUT TILE PER WAVE, judged by wall time on random data.
// Standalone model of conv_halo_kernel's K loop (csrc/conv_halo.hip): LDS-DMA halo per 64-channel chunk, LDS-DMA weight slice
// per tap, two barriers per tap, 3 workgroups per CU; no GroupNorm transform, trivial epilogue (the loop is what is compared).
//   shape A: 8 x 16 pixel tile, fragments of 16 pixels (one tile row), XOR swizzle (row & 7) -- as shipped;
//   shape B: 4 x 32 pixel tile, fragments of 32 pixels (one tile row), XOR swizzle ((row >> 1) & 7) so that the four
//            16-lane groups of a ds_read_b128 see 16 different (parity, slot) pairs: conflict-free like A.
// Both read 16 ds_read_b128 per wave per tap (same LDS bytes per FLOP) and issue 32 vs 16 MFMAs.  A checksum over all
// outputs must agree between the two (same products, different summation layout).
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_shape_ab.hip -o tools/mfma_shape_ab ; run: tools/mfma_shape_ab [rounds]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct P {
    const half_t* x;   // [B*H*W][C] pixel-major
    const half_t* w;   // [Cout][9][C]
    float* out;        // [B*H*W][Cout] (layout irrelevant: summed)
    int B, H, W, C, Cout, tiles_y, tiles_x;
};

static __device__ __attribute__((aligned(256))) half_t g_zero[128];

template <int SHAPE>   // 0: 16x16x32, tile 8x16; 1: 32x32x16, tile 4x32
__global__ __launch_bounds__(256, 3) void conv_loop(P p) {
    constexpr int TH = SHAPE ? 4 : 8, TW = SHAPE ? 32 : 16, BN = 128;
    constexpr int HWD = TW + 2, HROWS = (TH + 2) * HWD, HROWS_PAD = (HROWS + 7) / 8 * 8;
    constexpr int NV = (HROWS_PAD * 8 + 255) / 256;
    constexpr int XBYTES = HROWS_PAD * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xs = smem;
    char* wsm = smem + XBYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int per_img = p.tiles_y * p.tiles_x;
    const int bimg = blockIdx.x / per_img, tr = blockIdx.x - bimg * per_img;
    const int ty = tr / p.tiles_x, tx = tr - ty * p.tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;
    auto swz = [](int r) { return SHAPE ? ((r >> 1) & 7) : (r & 7); };

    int h_pix[NV];
    const int pos = tid & 7;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int hr = (tid + 256 * i) >> 3;
        const int hy = hr / HWD, hx = hr - hy * HWD;
        const int ly = y0 - 1 + hy, lx = x0 - 1 + hx;
        const bool ok = hr < HROWS && ly >= 0 && ly < p.H && lx >= 0 && lx < p.W;
        h_pix[i] = ok ? (bimg * p.H + ly) * p.W + lx : -1;
    }
    const int w_row0 = tid >> 3;
    const long long K = 9ll * p.C;

    f4 accA[4][4];
    f16v accB[2][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) accA[a][b] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 16; ++j) accB[a][b][j] = 0.f;

    bool first = true;
    for (int c64 = 0; c64 < p.C / 64; ++c64) {
        const int cb = c64 << 6;
        for (int tap = 0; tap < 9; ++tap) {
            if (!first) __syncthreads();
            first = false;
            if (tap == 0) {
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    if ((wave + 4 * i) * 64 < HROWS_PAD * 8) {
                        const int hr = (tid + 256 * i) >> 3;
                        const half_t* src = h_pix[i] >= 0 ? p.x + (long long)h_pix[i] * p.C + cb + ((pos ^ swz(hr)) << 3) : g_zero;
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                         (__attribute__((address_space(3))) void*)(xs + (wave + 4 * i) * 1024), 16, 0, 0);
                    }
                }
            }
            {
                const half_t* wsrc = p.w + (long long)w_row0 * K + (long long)tap * p.C + cb + ((pos ^ swz(w_row0)) << 3);
#pragma unroll
                for (int i = 0; i < BN / 32; ++i)      // rows w_row0 + 32 i: swz(row) == swz(w_row0) for both swizzles (32 % 16 == 0)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + (long long)(32 * i) * K),
                                                     (__attribute__((address_space(3))) void*)(wsm + (i * 32 + wave * 8) * 128), 16, 0, 0);
            }
            __syncthreads();
            const int dy = tap / 3, dx = tap - dy * 3;
            const int tapoff = dy * HWD + dx;
            const char* wsr = wsm + (wn * 64) * 128;
            if constexpr (SHAPE == 0) {
                const int frow = lane & 15, fq = lane >> 4;
                const int q0 = wm * 64 + frow;
                const int rb0 = (q0 / TW) * HWD + (q0 % TW);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    h8 xf[4], wf[4];
                    const int c = kk * 4 + fq;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int r = rb0 + b * HWD + tapoff;
                        xf[b] = *reinterpret_cast<const h8*>(xs + r * 128 + ((c ^ (r & 7)) << 4));
                    }
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const int r = a * 16 + frow;
                        wf[a] = *reinterpret_cast<const h8*>(wsr + r * 128 + ((c ^ (r & 7)) << 4));
                    }
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b)
                            accA[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[a], xf[b], accA[a][b], 0, 0, 0);
                }
            } else {
                const int l32 = lane & 31, kh = lane >> 5;
                // wave's 64 pixels = tile rows 2 wm, 2 wm + 1 (32 pixels each)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    h8 xf[2], wf[2];
                    const int c = ks * 2 + kh;
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const int r = (wm * 2 + b) * HWD + l32 + tapoff;
                        xf[b] = *reinterpret_cast<const h8*>(xs + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
                    }
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        const int r = a * 32 + l32;
                        wf[a] = *reinterpret_cast<const h8*>(wsr + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
                    }
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b)
                            accB[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[a], xf[b], accB[a][b], 0, 0, 0);
                }
            }
        }
    }
    // trivial epilogue: every accumulator value is written once (coalescing is not the subject)
    float* o = p.out + ((long long)blockIdx.x * 256 + tid) * 64;
    if constexpr (SHAPE == 0) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) *reinterpret_cast<f4*>(o + (a * 4 + b) * 4) = accA[a][b];
    } else {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<f4*>(o + ((a * 2 + b) * 4 + j) * 4) =
                        (f4){accB[a][b][4 * j], accB[a][b][4 * j + 1], accB[a][b][4 * j + 2], accB[a][b][4 * j + 3]};
    }
}

__global__ void sum_kernel(const float* x, long long n, double* out) {
    double s = 0, q = 0;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) { s += x[i]; q += (double)x[i] * x[i]; }
    atomicAdd(out, s);
    atomicAdd(out + 1, q);
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 5;
    const int B = 8, H = 512, W = 512, C = 128, Cout = 128;
    const long long M = (long long)B * H * W;
    std::vector<half_t> hx(M * C), hw((size_t)Cout * 9 * C);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
    for (auto& v : hx) v = (half_t)(rnd() + rnd() + rnd());          // roughly normal, random bits in every mantissa
    for (auto& v : hw) v = (half_t)((rnd() + rnd()) * 0.1f);
    half_t *dx, *dw;
    float* dout;
    double* dsum;
    CHECK(hipMalloc(&dx, hx.size() * 2));
    CHECK(hipMalloc(&dw, hw.size() * 2));
    CHECK(hipMalloc(&dout, M * Cout * 4));
    CHECK(hipMalloc(&dsum, 16));
    CHECK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    const double flop = 2.0 * M * Cout * 9 * C;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto launch = [&](int shape) {
        P p{dx, dw, dout, B, H, W, C, Cout, shape ? H / 4 : H / 8, shape ? W / 32 : W / 16};
        const int grid = B * p.tiles_y * p.tiles_x;
        if (shape == 0) {
            const int smem = ((10 * 18 + 7) / 8 * 8) * 128 + 128 * 128;
            hipLaunchKernelGGL(conv_loop<0>, dim3(grid), dim3(256), smem, 0, p);
        } else {
            const int smem = ((6 * 34 + 7) / 8 * 8) * 128 + 128 * 128;
            hipLaunchKernelGGL(conv_loop<1>, dim3(grid), dim3(256), smem, 0, p);
        }
    };
    double sums[2][2];
    for (int shape = 0; shape < 2; ++shape) {
        launch(shape);
        CHECK(hipMemset(dsum, 0, 16));
        hipLaunchKernelGGL(sum_kernel, dim3(1024), dim3(256), 0, 0, dout, M * Cout, dsum);
        CHECK(hipMemcpy(sums[shape], dsum, 16, hipMemcpyDeviceToHost));
    }
    printf("checksum 16x16x32: sum %.6e sumsq %.6e | 32x32x16: sum %.6e sumsq %.6e\n", sums[0][0], sums[0][1], sums[1][0], sums[1][1]);
    for (int r = 0; r < rounds; ++r) {
        for (int shape = 0; shape < 2; ++shape) {
            launch(shape);                       // warm
            CHECK(hipEventRecord(e0, 0));
            for (int i = 0; i < 5; ++i) launch(shape);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("round %d %s: %.1f us per launch, %.0f TFLOP/s\n", r, shape ? "32x32x16 (4x32 tile)" : "16x16x32 (8x16 tile)", ms / 5 * 1e3, flop / (ms / 5 * 1e-3) * 1e-12);
            fflush(stdout);
        }
    }
    CHECK(hipDeviceSynchronize());
    return 0;
}
