// Implicit-GEMM contraction kernel for gfx950 (MI355X): linear / 1x1 conv / 3x3 conv on MFMA.
//
// Replaces torch.nn.Linear / Conv2d(1x1) / Conv2d(3x3,pad 1) inside UNet2DConditionModel and
// AutoencoderKL.decode as reached from backends/cuda_worker.py:221-229.
//
// Data layout: activations pixel-major [M = B*H*W][C] fp16, weights [N][K] fp16 (K = Cin, or
// 9*Cin ordered (ky,kx,cin) for the 3x3 case).  out[m][n] = sum_k X[m][k] * W[n][k].
//
// Structure: 256 threads = 4 waves in a 2(m) x 2(n) grid over a BM x BN output tile, BK = 64.
// X and W tiles are register-staged (global_load_dwordx4 issued one K-step ahead, written to the other
// LDS buffer after the MFMAs of the current step) into XOR-swizzled 128-byte rows, read back with
// conflict-free ds_read_b128 as MFMA 16x16x32 f16 fragments.  The MFMA is issued "swapped"
// (A operand = weight rows, B operand = pixel rows) so each lane's 4 accumulator registers are 4
// consecutive output channels of ONE pixel: the epilogue (bias, time-embedding row add, residual,
// GEGLU gate, fp16 convert) runs on 8-byte vectors with no cross-lane traffic.
// For the 3x3 case the K loop walks (tap, 64-channel chunk); each A row is the 128 contiguous bytes of
// one shifted input pixel (zero-filled outside the image; optional fused nearest-2x upsample).
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
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // contiguous chunk of tiles per XCD (blocks are dealt round-robin over the 8 XCDs)
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

template <int BM, int BN, int MODE>
__global__ __launch_bounds__(256, 2) void igemm_kernel(IgemmParams p) {
    constexpr int RA = BM / 32, RW = BN / 32;   // staged rows per thread
    constexpr int TM = BM / 32, TN = BN / 32;   // 16-wide MFMA tiles per wave along m / n
    constexpr int XBYTES = BM * 128, WBYTES = BN * 128, BUF = XBYTES + WBYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int tile = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
    const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
    const int m_base = mt * BM, n_base = nt * BN;
    const int z = blockIdx.z;

    const half_t* __restrict__ Ab = p.A + z * p.strideA;
    const half_t* __restrict__ A2b = p.A2 ? p.A2 + z * p.strideA : nullptr;
    const half_t* __restrict__ Wb = p.W + z * p.strideW;

    const int chunk = tid & 7, row0 = tid >> 3;
    const int swz = (chunk ^ (row0 & 7)) << 4;

    // ---- per-thread A row descriptors ----
    bool a_ok[RA];
    long long a_off[RA];            // MODE 0: m*lda ; MODE 1: unused
    int a_b[RA], a_y[RA], a_x[RA];  // MODE 1
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = m_base + row0 + 32 * i;
        a_ok[i] = m < p.M;
        if (MODE == 0) {
            a_off[i] = (long long)m;
        } else {
            const int hw = p.Hout * p.Wout;
            const int b = m / hw, rem = m - b * hw;
            const int oy = rem / p.Wout;
            a_b[i] = b; a_y[i] = oy * p.stride - 1; a_x[i] = (rem - oy * p.Wout) * p.stride - 1;
        }
    }
    const half_t* wptr = Wb + (long long)(n_base + row0) * p.K + chunk * 8;

    h8 ra[RA], rw[RW];
    const int nk = p.K >> 6;

    auto load_tile = [&](int kt) {
        if (MODE == 0) {
            const int k0 = kt << 6;
            const bool second = (A2b != nullptr) && (k0 >= p.K1);
            const half_t* base = second ? A2b : Ab;
            const long long ld = second ? p.lda2 : p.lda;
            const int kk = (second ? k0 - p.K1 : k0) + chunk * 8;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (a_ok[i]) v = *reinterpret_cast<const h8*>(base + a_off[i] * ld + kk);
                ra[i] = v;
            }
        } else {
            const int cpt = p.Cin >> 6;
            const int tap = kt / cpt;
            const int c0 = ((kt - tap * cpt) << 6) + chunk * 8;
            const int dy = tap / 3, dx = tap - dy * 3;
            const int Hl = p.ups ? p.Hin * 2 : p.Hin, Wl = p.ups ? p.Win * 2 : p.Win;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                int iy = a_y[i] + dy, ix = a_x[i] + dx;
                if (a_ok[i] && iy >= 0 && iy < Hl && ix >= 0 && ix < Wl) {
                    if (p.ups) { iy >>= 1; ix >>= 1; }
                    const long long off = ((long long)(a_b[i] * p.Hin + iy) * p.Win + ix) * p.Cin + c0;
                    v = *reinterpret_cast<const h8*>(Ab + off);
                }
                ra[i] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < RW; ++i)
            rw[i] = *reinterpret_cast<const h8*>(wptr + (long long)(32 * i) * p.K + (kt << 6));
    };
    auto store_tile = [&](int buf) {
        char* xs = smem + buf * BUF;
        char* ws = xs + XBYTES;
#pragma unroll
        for (int i = 0; i < RA; ++i) *reinterpret_cast<h8*>(xs + (row0 + 32 * i) * 128 + swz) = ra[i];
#pragma unroll
        for (int i = 0; i < RW; ++i) *reinterpret_cast<h8*>(ws + (row0 + 32 * i) * 128 + swz) = rw[i];
    };

    f4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;

    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const char* xs = smem + cur * BUF + (wm * (BM / 2)) * 128;
        const char* ws = smem + cur * BUF + XBYTES + (wn * (BN / 2)) * 128;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            h8 xf[TM], wf[TN];
            const int c = kk * 4 + fq;
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                const int r = b * 16 + frow;
                xf[b] = *reinterpret_cast<const h8*>(xs + r * 128 + ((c ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int a = 0; a < TN; ++a) {
                const int r = a * 16 + frow;
                wf[a] = *reinterpret_cast<const h8*>(ws + r * 128 + ((c ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[a], xf[b], acc[a][b], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds out[m = ..+frow][n = ..+fq*4 .. +3] ----
    half_t* __restrict__ outb = p.out + z * p.strideO;
#pragma unroll
    for (int b = 0; b < TM; ++b) {
        const int m = m_base + wm * (BM / 2) + b * 16 + frow;
        if (m >= p.M) continue;
        const half_t* radd = p.rowadd ? p.rowadd + (long long)(m / p.rows_per_batch) * p.ld_rowadd : nullptr;
        if (p.epi == 0) {
#pragma unroll
            for (int a = 0; a < TN; ++a) {
                const int n = n_base + wn * (BN / 2) + a * 16 + fq * 4;
                f4 v = acc[a][b];
                if (p.bias) { h4 t = *reinterpret_cast<const h4*>(p.bias + n);
                    v[0] += (float)t[0]; v[1] += (float)t[1]; v[2] += (float)t[2]; v[3] += (float)t[3]; }
                if (radd) { h4 t = *reinterpret_cast<const h4*>(radd + n);
                    v[0] += (float)t[0]; v[1] += (float)t[1]; v[2] += (float)t[2]; v[3] += (float)t[3]; }
                v[0] *= p.out_scale; v[1] *= p.out_scale; v[2] *= p.out_scale; v[3] *= p.out_scale;
                if (p.res) { h4 t = *reinterpret_cast<const h4*>(p.res + (long long)m * p.ldr + n);
                    v[0] += (float)t[0]; v[1] += (float)t[1]; v[2] += (float)t[2]; v[3] += (float)t[3]; }
                h4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                *reinterpret_cast<h4*>(outb + (long long)m * p.ldo + n) = o;
            }
        } else {  // GEGLU: even n-tile = value rows, odd n-tile = gate rows (16-row interleave)
#pragma unroll
            for (int a = 0; a < TN; a += 2) {
                const int n = n_base + wn * (BN / 2) + a * 16 + fq * 4;      // packed row of the value
                f4 x = acc[a][b], g = acc[a + 1][b];
                if (p.bias) {
                    h4 tx = *reinterpret_cast<const h4*>(p.bias + n);
                    h4 tg = *reinterpret_cast<const h4*>(p.bias + n + 16);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { x[j] += (float)tx[j]; g[j] += (float)tg[j]; }
                }
                h4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (half_t)(x[j] * gelu_erf_f(g[j]));
                const int nout = ((n_base + wn * (BN / 2) + a * 16) >> 1) + fq * 4;
                *reinterpret_cast<h4*>(outb + (long long)m * p.ldo + nout) = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int MODE>
static int launch_cfg(IgemmParams& p, int batch, hipStream_t s) {
    p.mtiles = (p.M + BM - 1) / BM;
    p.ntiles = p.N / BN;
    const int smem = 2 * (BM + BN) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<BM, BN, MODE>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_set = true;
    }
    dim3 grid(p.mtiles * p.ntiles, 1, batch);
    hipLaunchKernelGGL((igemm_kernel<BM, BN, MODE>), grid, dim3(256), smem, s, p);
    LCM_CHECK_LAUNCH("igemm");
    return LCM_OK;
}

// tile selection: the largest tile that still gives >= 2 workgroups per CU; else the most tiles
static int pick_tile(int M, int N, int batch) {
    const bool n128 = (N % 128) == 0;
    auto tiles = [&](int bm, int bn) { return (long long)((M + bm - 1) / bm) * (N / bn) * batch; };
    if (n128 && M >= 128 && tiles(128, 128) >= 512) return 128 * 1000 + 128;
    if (M >= 128 && tiles(128, 64) >= 512) return 128 * 1000 + 64;
    if (n128 && M > 64 && M < 128) return 64 * 1000 + 128;
    return 64 * 1000 + 64;
}

extern "C" int lcm_gemm_tile_config(int M, int N, int batch) { return pick_tile(M, N, batch); }

template <int MODE>
static int launch_igemm(IgemmParams& p, int batch, hipStream_t s) {
    switch (pick_tile(p.M, p.N, batch)) {
        case 128128: return launch_cfg<128, 128, MODE>(p, batch, s);
        case 128064: return launch_cfg<128, 64, MODE>(p, batch, s);
        case 64128: return launch_cfg<64, 128, MODE>(p, batch, s);
        default: return launch_cfg<64, 64, MODE>(p, batch, s);
    }
}

extern "C" int lcm_gemm_f16(const void* A, int lda, const void* A2, int lda2, int K1,
                            const void* W, const void* bias, const void* rowadd, int ld_rowadd, int rows_per_batch,
                            const void* res, int ldr, void* out, int ldo,
                            int M, int N, int K, int epilogue, float out_scale,
                            int batch, int64_t strideA, int64_t strideW, int64_t strideO, void* stream) {
    LCM_REQUIRE(A && W && out, "gemm: null pointer");
    LCM_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0, "gemm: bad shape M=%d N=%d K=%d batch=%d", M, N, K, batch);
    LCM_REQUIRE(K % 64 == 0, "gemm: K=%d must be a multiple of 64", K);
    LCM_REQUIRE(N % 64 == 0, "gemm: N=%d must be a multiple of 64", N);
    LCM_REQUIRE(lda % 8 == 0 && ldo % 4 == 0, "gemm: lda=%d (%%8) / ldo=%d (%%4) misaligned", lda, ldo);
    if (A2) LCM_REQUIRE(K1 > 0 && K1 < K && K1 % 64 == 0 && lda2 % 8 == 0, "gemm: bad split K1=%d lda2=%d", K1, lda2);
    if (res) LCM_REQUIRE(ldr % 4 == 0, "gemm: ldr=%d misaligned", ldr);
    if (rowadd) LCM_REQUIRE(rows_per_batch > 0 && ld_rowadd % 4 == 0, "gemm: bad rowadd");
    LCM_REQUIRE(epilogue == 0 || epilogue == 1, "gemm: unknown epilogue %d", epilogue);
    if (epilogue == 1) LCM_REQUIRE(!rowadd && !res && !A2, "gemm: GEGLU epilogue takes bias only");
    IgemmParams p = {};
    p.A = (const half_t*)A; p.A2 = (const half_t*)A2; p.W = (const half_t*)W;
    p.bias = (const half_t*)bias; p.rowadd = (const half_t*)rowadd; p.res = (const half_t*)res;
    p.out = (half_t*)out;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.lda2 = lda2; p.K1 = A2 ? K1 : K;
    p.ldo = ldo; p.ldr = ldr; p.ld_rowadd = ld_rowadd; p.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1;
    p.epi = epilogue; p.out_scale = out_scale;
    p.strideA = strideA; p.strideW = strideW; p.strideO = strideO;
    return launch_igemm<0>(p, batch, (hipStream_t)stream);
}

extern "C" int lcm_conv3x3_f16(const void* in, const void* W, const void* bias,
                               const void* rowadd, int ld_rowadd, const void* res, void* out,
                               int B, int Hin, int Win, int Cin, int Cout, int stride, int ups, void* stream) {
    LCM_REQUIRE(in && W && out, "conv3x3: null pointer");
    LCM_REQUIRE(B > 0 && Hin > 0 && Win > 0, "conv3x3: bad shape");
    LCM_REQUIRE(Cin % 64 == 0 && Cout % 64 == 0, "conv3x3: Cin=%d Cout=%d must be multiples of 64", Cin, Cout);
    LCM_REQUIRE(stride == 1 || stride == 2, "conv3x3: stride %d", stride);
    LCM_REQUIRE(!(ups && stride != 1), "conv3x3: upsample needs stride 1");
    if (rowadd) LCM_REQUIRE(ld_rowadd % 4 == 0, "conv3x3: ld_rowadd misaligned");
    const int Hl = ups ? 2 * Hin : Hin, Wl = ups ? 2 * Win : Win;
    IgemmParams p = {};
    p.A = (const half_t*)in; p.W = (const half_t*)W; p.bias = (const half_t*)bias;
    p.rowadd = (const half_t*)rowadd; p.res = (const half_t*)res; p.out = (half_t*)out;
    p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.stride = stride; p.ups = ups;
    p.Hout = (Hl + 2 - 3) / stride + 1; p.Wout = (Wl + 2 - 3) / stride + 1;
    p.M = B * p.Hout * p.Wout; p.N = Cout; p.K = 9 * Cin;
    p.ldo = Cout; p.ldr = Cout; p.ld_rowadd = ld_rowadd; p.rows_per_batch = p.Hout * p.Wout;
    p.epi = 0; p.out_scale = 1.0f;
    return launch_igemm<1>(p, 1, (hipStream_t)stream);
}
