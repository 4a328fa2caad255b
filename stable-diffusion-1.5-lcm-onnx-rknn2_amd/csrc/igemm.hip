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
#include "igemm_common.h"

template <int BM, int BN, int MODE, int LN = 0>
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
    const int nk_all = p.K >> 6;
    const int kt0 = (int)((long long)blockIdx.y * nk_all / p.splits);
    const int kt1 = (int)((long long)(blockIdx.y + 1) * nk_all / p.splits);

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
    // LN is a compile-time switch: a run-time branch inside the K loop makes the compiler lose track of which LDS slot the
    // LDS-DMA in flight writes, and it then drains the whole ring (s_waitcnt vmcnt(0)) before every fragment read
    constexpr bool ln = LN != 0;
    float ln_s[TM], ln_q[TM];
#pragma unroll
    for (int b = 0; b < TM; ++b) { ln_s[b] = 0.f; ln_q[b] = 0.f; }

    load_tile(kt0);
    store_tile(0);
    __syncthreads();

    for (int kt = kt0; kt < kt1; ++kt) {
        const int cur = (kt - kt0) & 1;
        if (kt + 1 < kt1) load_tile(kt + 1);
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
            if constexpr (ln) {
#pragma unroll
                for (int b = 0; b < TM; ++b) ln_accum(xf[b], ln_s[b], ln_q[b]);
            }
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[a], xf[b], acc[a][b], 0, 0, 0);
        }
        if (kt + 1 < kt1) store_tile(cur ^ 1);
        __syncthreads();
    }

    int m_of[TM];
#pragma unroll
    for (int b = 0; b < TM; ++b) {
        const int m = m_base + wm * (BM / 2) + b * 16 + frow;
        m_of[b] = m < p.M ? m : -1;
    }
    int slab_of[BM / 64];
#pragma unroll
    for (int bp = 0; bp < BM / 64; ++bp) {
        const int r = m_base + wm * (BM / 2) + bp * 32;
        slab_of[bp] = r < p.M ? (r >> 5) : -1;
    }
    if constexpr (ln) {
        ln_finish<TM>(ln_s, ln_q, p.K, p.ln_eps);
        igemm_epilogue<BM, BN>(p, acc, m_of, n_base + wn * (BN / 2), fq, z, slab_of, ln_s, ln_q);
    } else {
        igemm_epilogue<BM, BN>(p, acc, m_of, n_base + wn * (BN / 2), fq, z, slab_of);
    }
}

// ------------------------------------------------------------------------------------------------
// v2: the same tile / MFMA / epilogue structure fed by LDS-DMA (global_load_lds_dwordx4): no staging VGPRs, no
// ds_write pass, S LDS stages with a COUNTED s_waitcnt vmcnt so S-2 tiles stay in flight across the single raw
// s_barrier per K-step.  The DMA destination is lane-linear (wave base + lane*16), so the XOR swizzle is applied
// to the per-lane SOURCE chunk (lane at position pos of row r fetches chunk pos ^ (r&7)); the swizzled
// ds_read_b128 addresses are unchanged.  Zero fill (image border, m >= M) = source pointer into a zero page.
// ------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(256))) half_t g_zero_page[128];

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BM, int BN, int MODE, int S, int LN = 0>
__global__ __launch_bounds__(256, S == 1 ? (BM + BN < 256 ? 4 : 3) : 2) void igemm2_kernel(IgemmParams p) {
    constexpr int RA = BM / 32, RW = BN / 32;
    constexpr int TM = BM / 32, TN = BN / 32;
    constexpr int XBYTES = BM * 128, WBYTES = BN * 128, BUF = XBYTES + WBYTES;
    constexpr int LPT = RA + RW;                 // LDS-DMA instructions per thread per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    // A workgroup walks n_iters consecutive n-tiles of one m-tile; the (n-tile, k-tile) steps are flattened so the
    // LDS-DMA ring keeps flowing across n-tile boundaries (short-K GEMMs: no pipeline ramp / drain per tile).
    const int ngroups = p.ntiles / p.n_iters;
    const int tile = xcd_remap(blockIdx.x, p.mtiles * ngroups);
    const int mt = tile / ngroups, nt0 = (tile - mt * ngroups) * p.n_iters;
    const int m_base = mt * BM;
    const int z = blockIdx.z;

    const half_t* __restrict__ Ab = p.A + z * p.strideA;
    const half_t* __restrict__ A2b = p.A2 ? p.A2 + z * p.strideA : nullptr;
    const half_t* __restrict__ Wb = p.W + z * p.strideW;

    const int pos = tid & 7, row0 = tid >> 3;
    const int schunk = (pos ^ (row0 & 7)) * 8;      // source chunk (halves) this lane fetches

    bool a_ok[RA];
    long long a_off[RA];
    int a_b[RA], a_y[RA], a_x[RA];
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
    const half_t* wptr = Wb + (long long)row0 * p.K + schunk;
    const int nk_all = p.K >> 6;
    const int kt0 = (int)((long long)blockIdx.y * nk_all / p.splits);
    const int kt1 = (int)((long long)(blockIdx.y + 1) * nk_all / p.splits);
    const int nkr = kt1 - kt0;
    const int T = nkr * p.n_iters;

    auto issue_tile = [&](int t, int buf) {          // t = flattened (n-tile, k-tile) step
        const int ni = t / nkr;
        const int kt = kt0 + (t - ni * nkr);
        const int n_base_i = (nt0 + ni) * BN;
        char* xs = smem + buf * BUF + wave * 8 * 128;
        char* ws = xs + XBYTES;
        if (MODE == 0) {
            const int k0 = kt << 6;
            const bool second = (A2b != nullptr) && (k0 >= p.K1);
            const half_t* base = second ? A2b : Ab;
            const long long ld = second ? p.lda2 : p.lda;
            const int kk = (second ? k0 - p.K1 : k0) + schunk;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const half_t* src = a_ok[i] ? base + a_off[i] * ld + kk : g_zero_page;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(xs + i * 32 * 128), 16, 0, 0);
            }
        } else {
            const int cpt = p.Cin >> 6;
            const int tap = kt / cpt;
            const int c0 = ((kt - tap * cpt) << 6) + schunk;
            const int dy = tap / 3, dx = tap - dy * 3;
            const int Hl = p.ups ? p.Hin * 2 : p.Hin, Wl = p.ups ? p.Win * 2 : p.Win;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                int iy = a_y[i] + dy, ix = a_x[i] + dx;
                const bool ok = a_ok[i] && iy >= 0 && iy < Hl && ix >= 0 && ix < Wl;
                if (p.ups) { iy >>= 1; ix >>= 1; }
                const long long off = ((long long)(a_b[i] * p.Hin + iy) * p.Win + ix) * p.Cin + c0;
                const half_t* src = ok ? Ab + off : g_zero_page;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(xs + i * 32 * 128), 16, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const half_t* src = wptr + (long long)(n_base_i + 32 * i) * p.K + (kt << 6);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(ws + i * 32 * 128), 16, 0, 0);
        }
    };

    f4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    // compile-time (see igemm_kernel).  The epilogue's ln_g / ln_c come from LDS, not from global memory: an ordinary
    // VGPR-destination global load inside the K loop of a kernel that keeps LDS-DMA in flight makes the compiler drain the
    // ring (s_waitcnt vmcnt(0)) in front of every fragment read (+50 % on the 64x64 launches, measured).
    constexpr bool ln = LN != 0;
    float ln_s[TM], ln_q[TM];
#pragma unroll
    for (int b = 0; b < TM; ++b) { ln_s[b] = 0.f; ln_q[b] = 0.f; }

    const __attribute__((address_space(3))) float* ln_lds = nullptr;
    if constexpr (ln) {       // this n-tile's ln_g | ln_c into LDS (behind the ring's stages) before any LDS-DMA is in flight
        float* t = reinterpret_cast<float*>(smem + S * BUF);
        if (tid < BN) { t[tid] = p.ln_g[nt0 * BN + tid]; t[BN + tid] = p.ln_c[nt0 * BN + tid]; }
        __syncthreads();
        ln_lds = (const __attribute__((address_space(3))) float*)t;
    }
#pragma unroll
    for (int s = 0; s < S - 1; ++s)
        if (s < T) issue_tile(s, s);

    int buf = 0, kstep = 0, ni_cur = 0;
    for (int t = 0; t < T; ++t) {
        // tile t must have landed; up to min(S-2, tiles issued after t) newer tiles may stay in flight
        const int newer = T - 1 - t;
        if (S == 1) {   // single LDS buffer: rely on 4-5 co-resident workgroups per CU to cover the load latency
            if (t > 0) __builtin_amdgcn_s_barrier();      // everyone done reading the previous tile
            issue_tile(t, 0);
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
        } else {
            if (S >= 6 && newer >= 4) wait_vmcnt<(S >= 6 ? 4 : 0) * LPT>();
            else if (S >= 5 && newer >= 3) wait_vmcnt<(S >= 5 ? 3 : 0) * LPT>();
            else if (S >= 4 && newer >= 2) wait_vmcnt<(S >= 4 ? 2 : 0) * LPT>();
            else if (S >= 3 && newer >= 1) wait_vmcnt<(S >= 3 ? 1 : 0) * LPT>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (t + S - 1 < T) issue_tile(t + S - 1, (buf + S - 1) % S);
        }

        const char* xs = smem + buf * BUF + (wm * (BM / 2)) * 128;
        const char* ws = smem + buf * BUF + XBYTES + (wn * (BN / 2)) * 128;
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
            if constexpr (ln) {            // the row statistics of A
#pragma unroll
                for (int b = 0; b < TM; ++b) ln_accum(xf[b], ln_s[b], ln_q[b]);
            }
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[a], xf[b], acc[a][b], 0, 0, 0);
        }
        buf = (buf + 1 == S) ? 0 : buf + 1;
        if (++kstep == nkr) {          // this n-tile is complete: epilogue while the next tile's loads are in flight
            kstep = 0;
            int m_of[TM];
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                const int m = m_base + wm * (BM / 2) + b * 16 + frow;
                m_of[b] = m < p.M ? m : -1;
            }
            int slab_of[BM / 64];
#pragma unroll
            for (int bp = 0; bp < BM / 64; ++bp) {
                const int r = m_base + wm * (BM / 2) + bp * 32;
                slab_of[bp] = r < p.M ? (r >> 5) : -1;
            }
            if constexpr (ln) {
                ln_finish<TM>(ln_s, ln_q, p.K, p.ln_eps);
                igemm_epilogue<BM, BN>(p, acc, m_of, (nt0 + ni_cur) * BN + wn * (BN / 2), fq, z, slab_of, ln_s, ln_q, ln_lds, nt0 * BN);
            } else {
                igemm_epilogue<BM, BN>(p, acc, m_of, (nt0 + ni_cur) * BN + wn * (BN / 2), fq, z, slab_of);
            }
            ++ni_cur;
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};
        }
    }
}

// Split-K combine: fixed-order sum of the fp32 slabs (bit-reproducible) + the epilogue of the main kernel, and
// optionally the fused GroupNorm statistics of the result.  A workgroup owns p.reduce_rows consecutive rows; a thread
// owns one 4-channel group and walks rows, so per-channel sums need only a fixed-order fold over the row lanes.
#define RED_THREADS 1024
__global__ __launch_bounds__(RED_THREADS) void splitk_reduce_kernel(IgemmParams p) {
    __shared__ float red[RED_THREADS * 4 * 2];
    const int n4 = p.N >> 2, tid = threadIdx.x;
    // workgroup = one slab of one image: rows [s * reduce_rows, (s+1) * reduce_rows) of image b, the last slab of an image
    // may be short (image sizes that are no multiple of the slab height); slabs never straddle images
    const int hw = p.img_rows > 0 ? p.img_rows : p.M;
    const int spi = (hw + p.reduce_rows - 1) / p.reduce_rows;             // slabs per image
    const int bimg = blockIdx.x / spi, sl = blockIdx.x - bimg * spi;
    const int r0 = bimg * hw + sl * p.reduce_rows, r1 = min(bimg * hw + hw, r0 + p.reduce_rows);
    const long long slab = (long long)p.M * p.N;
    const bool small = n4 <= RED_THREADS;
    const int nrl = small ? RED_THREADS / n4 : 1;        // row lanes: threads that share a channel group
    const int rl = small ? tid / n4 : 0;
    const bool do_stats = p.stats != nullptr;
    for (int cg = small ? tid - rl * n4 : tid; cg < n4; cg += RED_THREADS) {
        const int n = cg * 4;
        float ss[4] = {0.f, 0.f, 0.f, 0.f}, qq[4] = {0.f, 0.f, 0.f, 0.f};
        if (rl < nrl) {
            f4 bias4 = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) { h4 t = *reinterpret_cast<const h4*>(p.bias + n); bias4 = (f4){(float)t[0], (float)t[1], (float)t[2], (float)t[3]}; }
            for (int m = r0 + rl; m < r1; m += nrl) {
                const float* src = p.ws + (long long)m * p.N + n;
                f4 v = *reinterpret_cast<const f4*>(src);
                for (int s = 1; s < p.splits; ++s) {
                    f4 t = *reinterpret_cast<const f4*>(src + s * slab);
                    v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
                }
                v[0] += bias4[0]; v[1] += bias4[1]; v[2] += bias4[2]; v[3] += bias4[3];
                if (p.rowadd) { h4 t = *reinterpret_cast<const h4*>(p.rowadd + (long long)(m / p.rows_per_batch) * p.ld_rowadd + n);
                    v[0] += (float)t[0]; v[1] += (float)t[1]; v[2] += (float)t[2]; v[3] += (float)t[3]; }
                v[0] *= p.out_scale; v[1] *= p.out_scale; v[2] *= p.out_scale; v[3] *= p.out_scale;
                if (p.res) { h4 t = *reinterpret_cast<const h4*>(p.res + (long long)m * p.ldr + n);
                    v[0] += (float)t[0]; v[1] += (float)t[1]; v[2] += (float)t[2]; v[3] += (float)t[3]; }
                h4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                *reinterpret_cast<h4*>(p.out + (long long)m * p.ldo + n) = o;
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float f = (float)o[j]; ss[j] += f; qq[j] += f * f; }
            }
        }
        if (do_stats) {
            if (!small) {       // one thread owns the channel group for the whole slab
                float* dst = p.stats + ((long long)blockIdx.x * p.N + n) * 2;
                *reinterpret_cast<f4*>(dst) = (f4){ss[0], qq[0], ss[1], qq[1]};
                *reinterpret_cast<f4*>(dst + 4) = (f4){ss[2], qq[2], ss[3], qq[3]};
            } else if (rl < nrl) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { red[((rl * p.N) + n + j) * 2] = ss[j]; red[((rl * p.N) + n + j) * 2 + 1] = qq[j]; }
            }
        }
        if (small) break;
    }
    if (do_stats && small) {
        __syncthreads();
        for (int c = tid; c < p.N; c += RED_THREADS) {
            float s = 0.f, q = 0.f;
            for (int r = 0; r < nrl; ++r) { s += red[(r * p.N + c) * 2]; q += red[(r * p.N + c) * 2 + 1]; }
            float* dst = p.stats + ((long long)blockIdx.x * p.N + c) * 2;
            dst[0] = s; dst[1] = q;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
#include <map>
#include <mutex>
static float* g_ws[16] = {};
static long long g_ws_bytes[16] = {};
struct StreamWs { float* ptr; long long bytes; };
static std::map<hipStream_t, StreamWs> g_stream_ws;       // lanes: one workspace per launch stream (two passes in flight
static std::mutex g_ws_mu;                                // must not share slabs); falls back to the device-wide one

extern "C" int lcm_set_workspace(void* ptr, int64_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { lcm_set_error("set_workspace: no device"); return LCM_ENODEV; }
    std::lock_guard<std::mutex> lk(g_ws_mu);
    g_ws[dev] = (float*)ptr;
    g_ws_bytes[dev] = ptr ? bytes : 0;
    return LCM_OK;
}

extern "C" int lcm_set_stream_workspace(void* stream, void* ptr, int64_t bytes) {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    if (ptr) g_stream_ws[(hipStream_t)stream] = StreamWs{(float*)ptr, (long long)bytes};
    else g_stream_ws.erase((hipStream_t)stream);
    return LCM_OK;
}

float* lcm_splitk_workspace(long long* bytes, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    auto it = g_stream_ws.find(s);
    if (it != g_stream_ws.end()) { *bytes = it->second.bytes; return it->second.ptr; }
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16) { *bytes = 0; return nullptr; }
    *bytes = g_ws[dev] ? g_ws_bytes[dev] : 0;
    return g_ws[dev];
}
// rows per reduce workgroup: a power of two <= 32 chosen from `hw` = output rows PER IMAGE only (the slab structure --
// hence the order in which the fused statistics are summed -- depends on the image shape, never on how many images share
// the launch) so that an image has >= ~128 slabs where it can; ceil(hw / rows) slabs per image, the last one may be short
int lcm_reduce_rows(int hw) {
    int rs = 32;
    while (rs > 1 && (hw + rs - 1) / rs < 128) rs >>= 1;
    return rs;
}
int lcm_reduce_slabs(int hw) { const int rs = lcm_reduce_rows(hw); return (hw + rs - 1) / rs; }

void lcm_launch_splitk_reduce(IgemmParams& p, hipStream_t s) {
    const int hw = p.img_rows > 0 && p.M % p.img_rows == 0 ? p.img_rows : p.M;
    p.img_rows = hw;
    p.reduce_rows = lcm_reduce_rows(hw);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((p.M / hw) * lcm_reduce_slabs(hw))), dim3(RED_THREADS), 0, s, p);
}

// ---- per-shape launch plans (filled by the host-side autotuner; heuristics below are the fallback) ----
#include <tuple>
struct PlanVal { int bm, bn, splits, variant; };
static std::map<std::tuple<int, int, int, int, int>, PlanVal> g_plans;
static std::mutex g_plans_mu;

extern "C" int lcm_plan_set(int kind, int M, int N, int K, int aux, int bm, int bn, int splits, int variant) {
    if (!((bm == 128 || bm == 64) && (bn == 128 || bn == 64 || bn == 160) && splits >= 1 && splits <= 64 && variant >= -1 && variant <= 4)) {
        lcm_set_error("plan_set: bad plan %dx%d splits %d variant %d", bm, bn, splits, variant);
        return LCM_EINVAL;
    }
    std::lock_guard<std::mutex> lk(g_plans_mu);
    g_plans[std::make_tuple(kind, M, N, K, aux)] = PlanVal{bm, bn, splits, variant};
    return LCM_OK;
}

extern "C" int lcm_plan_clear(void) {
    std::lock_guard<std::mutex> lk(g_plans_mu);
    g_plans.clear();
    return LCM_OK;
}

bool lcm_plan_get(int kind, int M, int N, int K, int aux, int* bm, int* bn, int* splits, int* variant) {
    std::lock_guard<std::mutex> lk(g_plans_mu);
    auto it = g_plans.find(std::make_tuple(kind, M, N, K, aux));
    if (it == g_plans.end()) return false;
    *bm = it->second.bm; *bn = it->second.bn; *splits = it->second.splits; *variant = it->second.variant;
    return true;
}

// Tile + split-K selection.
//
// What decides the numbers and what does not.  The fp32 summation order of an output element is fixed by the K
// partition alone (a workgroup walks its k-tiles in order; the split-K combine adds the slabs in slab order); tile
// shape, ring depth and kernel variant are pure launch parameters, and the fused statistics are written per canonical
// 32-pixel slab whatever the tile (igemm_epilogue).  So the split factor is a function of the PER-IMAGE problem
// (kind, rows per image, N, K) -- from the plan table entry of that per-image shape when there is one, else from the
// deterministic heuristic below -- and is never chosen by timing at run time nor from the batch size: a request gets
// the same bits alone, in a batch of 8, in another process and on another box.  Tile / variant come from the plan
// entry of the TOTAL shape (tuned freely) or the occupancy heuristic.
struct TilePick { int bm, bn, splits; };
static int g_target_wgs = 384, g_max_splits = 16, g_min_wgs = 256;
static int g_split_max_rows = 1024, g_split_cap = 4;
static int g_variant = -1;     // -1: auto (1 stage when >= 4 workgroups per CU are available, else 2); 0: register-staged
                               // double buffer (v1); 1/2/3/4: LDS-DMA pipeline with that many stages

void lcm_tuning(int* target_wgs, int* max_splits, int* min_wgs) {
    *target_wgs = g_target_wgs; *max_splits = g_max_splits; *min_wgs = g_min_wgs;
}

static int g_conv_impl = 1;    // 1: LDS-halo conv (conv_halo.hip) for stride-1 3x3; 0: row-gather igemm everywhere

extern "C" int lcm_set_conv_impl(int impl) {
    if (impl != 0 && impl != 1) { lcm_set_error("conv_impl: %d", impl); return LCM_EINVAL; }
    g_conv_impl = impl;
    return LCM_OK;
}

extern "C" int lcm_set_kernel_variant(int variant) {
    if (variant < -1 || variant > 6) { lcm_set_error("kernel_variant: %d", variant); return LCM_EINVAL; }
    g_variant = variant;
    return LCM_OK;
}

extern "C" int lcm_set_tuning(int target_wgs, int max_splits, int min_wgs) {
    if (target_wgs > 0) g_target_wgs = target_wgs;
    if (max_splits > 0) g_max_splits = max_splits;
    if (min_wgs > 0) g_min_wgs = min_wgs;
    return LCM_OK;
}

extern "C" int lcm_set_split_policy(int max_rows_per_image, int max_parts) {
    if (max_rows_per_image < 0 || max_parts < 1 || max_parts > 64) { lcm_set_error("split_policy: %d rows / %d parts", max_rows_per_image, max_parts); return LCM_EINVAL; }
    g_split_max_rows = max_rows_per_image;
    g_split_cap = max_parts;
    return LCM_OK;
}

// fixed_splits < 0: choose the split factor too (the canonical-partition heuristic, called with the per-image shape);
// >= 1: the partition is given, pick the most efficient tile that fills the chip with it
static TilePick pick_tile(int M, int N, int K, int batch, int fixed_splits) {
    const int cand[4][2] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}};
    const int nk = K >> 6;
    TilePick best = {64, 64, fixed_splits > 0 ? fixed_splits : 1};
    long long best_wgs = -1;
    for (int c = 0; c < 4; ++c) {
        const int bm = cand[c][0], bn = cand[c][1];
        if (N % bn) continue;
        if (bm == 128 && M < 128) continue;
        if (bm == 64 && bn == 128 && M >= 128) continue;
        const long long tiles = (long long)((M + bm - 1) / bm) * (N / bn) * batch;
        int splits = fixed_splits > 0 ? fixed_splits : 1;
        if (fixed_splits < 0 && batch == 1 && tiles < g_min_wgs && nk >= 16) {
            splits = (int)((g_target_wgs + tiles - 1) / tiles);
            if (splits > nk / 8) splits = nk / 8;
            if (splits > g_max_splits) splits = g_max_splits;
            if (splits < 1) splits = 1;
        }
        const long long wgs = tiles * splits;
        if (wgs >= g_min_wgs) return {bm, bn, splits};
        if (wgs > best_wgs) { best_wgs = wgs; best = {bm, bn, splits}; }
    }
    return best;
}

// The partition is paid for at every batch size (a batch of 8 splits every image the way a lone image is split, and the
// fp32 slabs then cost HBM traffic the batched launch would not otherwise need), so it is kept to where a lone image
// cannot fill the chip any other way: at most 1024 output rows per image (the 32x32 latent level and below) and at most
// 4 parts (at batch 1 the measured difference between 4 and 10 parts is within noise: profiles/r01_chain_latency_conv.txt).
int lcm_split_policy(int m_img, int sp) {
    if (m_img > g_split_max_rows) return 1;
    if (sp > g_split_cap) sp = g_split_cap;
    return sp < 1 ? 1 : sp;
}

// canonical split factor of a GEMM-kind contraction (kind 0: linear / 1x1, kind 1: row-gather 3x3) with `m_img`
// output rows per image
static int canonical_splits_gemm(int kind, int m_img, int N, int K) {
    int pbm, pbn, psp, pv;
    int sp = lcm_plan_get(kind, m_img, N, K, 1, &pbm, &pbn, &psp, &pv) ? psp : pick_tile(m_img, N, K, 1, -1).splits;
    if (sp > (K >> 6)) sp = K >> 6;
    return lcm_split_policy(m_img, sp);
}

int lcm_canonical_splits_halo(int m_img, int N, int K, int IH, int IW, int W, int ph, int xform);

// The K partition the library will use for a contraction of this per-image shape (the autotuner tunes tile / variant
// around it).  kind 2 (LDS-halo 3x3): aux = (output width << 1) | gn-fused flag, ph = 1 for the phase-decomposed
// upsample convolution.
extern "C" int lcm_canonical_splits(int kind, int m_img, int N, int K, int aux, int ph) {
    if (kind == 0 || kind == 1) return canonical_splits_gemm(kind, m_img, N, K);
    const int W = aux >> 1;
    if (kind != 2 || W <= 0 || m_img % W) { lcm_set_error("canonical_splits: bad key"); return LCM_EINVAL; }
    const int H = m_img / W;
    return lcm_canonical_splits_halo(m_img, N, K, ph ? (H + 1) / 2 : H, ph ? (W + 1) / 2 : W, W, ph, aux & 1);
}

extern "C" int lcm_gemm_tile_config(int M, int N, int batch) {
    TilePick t = pick_tile(M, N, 64, batch, 1);
    return t.bm * 1000 + t.bn;
}

template <int BM, int BN, int MODE, int S, int LN = 0>
static int launch_v2(IgemmParams& p, dim3 grid, hipStream_t s) {
    constexpr int smem = S * (BM + BN) * 128 + (LN ? 2 * BN * 4 : 0);       // LN: + this n-tile's ln_g | ln_c
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm2_kernel<BM, BN, MODE, S, LN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_set = true;
    }
    char nm[64];
    snprintf(nm, sizeof(nm), "igemm2_kernel<%d, %d, %d, %d, %d>%s", BM, BN, MODE, S, LN, p.splits > 1 ? " +splitk" : "");
    lcm_prof_start(nm, s);
    hipLaunchKernelGGL((igemm2_kernel<BM, BN, MODE, S, LN>), grid, dim3(256), smem, s, p);
    lcm_prof_stop(s);
    return 0;
}

static int g_persist_n = 0;      // 1: short-K GEMM launches let a workgroup walk several n-tiles (persistent-over-N);
                                 // bit-identical, measured neutral on the UNet shapes (the limit is L2->LDS bytes per FLOP of
                                 // the 128x64 tile, not the per-tile pipeline ramp), so it stays off

extern "C" int lcm_set_persist_n(int on) { g_persist_n = on ? 1 : 0; return LCM_OK; }

template <int BM, int BN, int MODE, int LN = 0>
static int launch_cfg(IgemmParams& p, int batch, int splits, int plan_variant, hipStream_t s) {
    p.mtiles = (p.M + BM - 1) / BM;
    p.ntiles = p.N / BN;
    p.splits = splits;
    p.n_iters = 1;
    int variant = g_variant >= 0 ? g_variant : plan_variant;
    // persistent-over-N: short K (<= 20 k-tiles), several n-tiles, and more tiles than the chip can hold at once
    if (g_persist_n && MODE == 0 && !LN && splits == 1 && batch == 1 && variant != 0 && p.ntiles > 1 && (p.K >> 6) <= 20 &&
        (long long)p.mtiles * p.ntiles > 512) {
        int ni = p.ntiles;                                         // largest divisor of ntiles keeping >= 384 workgroups
        while (ni > 1 && (p.ntiles % ni != 0 || (long long)p.mtiles * (p.ntiles / ni) < 384)) --ni;
        p.n_iters = ni;
    }
    dim3 grid(p.mtiles * (p.ntiles / p.n_iters), splits, batch);
    if (variant == 5) variant = (BM + BN <= 128) ? 4 : -1;      // deep prefetch on the small tile only
    if (variant == 6) variant = (BM + BN <= 128) ? 6 : -1;
    if (variant < 0) {   // auto: enough workgroups for 4 per CU -> single buffer; small tile -> 4-stage prefetch (short-K,
                         // latency-bound GEMMs); else double buffer.  A persistent-over-N launch needs a real ring.
        const long long wgs = (long long)grid.x * grid.y * grid.z;
        variant = (wgs >= 1024 && p.n_iters == 1) ? 1 : ((BM + BN <= 128) ? 4 : 2);
    }
    if (p.n_iters > 1 && variant == 1) variant = 2;
    if (variant == 4 && BM + BN > 192) variant = 3;      // 4 x 32 KiB stages only for the small tiles
    if (variant == 0) {
        const int smem = 2 * (BM + BN) * 128;
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<BM, BN, MODE, LN>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            attr_set = true;
        }
        char nm[64];
        snprintf(nm, sizeof(nm), "igemm_kernel<%d, %d, %d, %d>%s", BM, BN, MODE, LN, p.splits > 1 ? " +splitk" : "");
        lcm_prof_start(nm, s);
        hipLaunchKernelGGL((igemm_kernel<BM, BN, MODE, LN>), grid, dim3(256), smem, s, p);
        lcm_prof_stop(s);
    } else if (variant == 1) {
        launch_v2<BM, BN, MODE, 1, LN>(p, grid, s);
    } else if (variant == 2) {
        launch_v2<BM, BN, MODE, 2, LN>(p, grid, s);
    } else if (variant == 3) {
        launch_v2<BM, BN, MODE, 3, LN>(p, grid, s);
    } else if (variant == 6) {
        if constexpr (BM + BN <= 128) launch_v2<BM, BN, MODE, 6, LN>(p, grid, s);
    } else {
        launch_v2<BM, BN, MODE, 4, LN>(p, grid, s);
    }
    LCM_CHECK_LAUNCH("igemm");
    if (splits > 1) {
        lcm_launch_splitk_reduce(p, s);
        LCM_CHECK_LAUNCH("splitk_reduce");
    }
    return LCM_OK;
}

// Fused-statistics bookkeeping: `img_rows` = output rows per image.  On return *slabs_per_image is the number of
// [N][2] partial rows the launch wrote per image (0 = statistics not produced; the caller then runs the standalone
// statistics kernel).
template <int MODE>
static int launch_igemm(IgemmParams& p, int batch, hipStream_t s, int img_rows, bool allow_split, bool want_stats,
                        int* slabs_per_image) {
    long long wsb = 0;
    p.ws = lcm_splitk_workspace(&wsb, s);
    if (img_rows <= 0 || p.M % img_rows) img_rows = p.M;
    p.img_rows = img_rows;
    int splits = 1;
    if (allow_split && p.ws && p.epi == 0 && batch == 1) {      // no workspace registered: the library never splits
        splits = canonical_splits_gemm(MODE, img_rows, p.N, p.K);
        if (splits > 1 && (long long)splits * p.M * p.N * 4 > wsb) {
            lcm_set_error("split-K workspace too small: %d x %d x %d fp32 slabs need %lld MB, have %lld MB "
                          "(lcm_set_workspace / LCM_SPLITK_WS_MB)", splits, p.M, p.N,
                          ((long long)splits * p.M * p.N * 4 + (1 << 20) - 1) >> 20, wsb >> 20);
            return LCM_EINVAL;
        }
    }
    TilePick t = pick_tile(p.M, p.N, p.K, batch, splits);
    int variant = -1, pbm, pbn, psp, pv;
    // the 160-wide tile (N = 320 / 640 / 960 ...: fewer L2->LDS bytes per FLOP than 64-wide) exists for the plain GEMM only;
    // its 5 n-fragments per wave cannot carry the GEGLU value/gate pairing
    if (lcm_plan_get(MODE, p.M, p.N, p.K, batch, &pbm, &pbn, &psp, &pv) && p.N % pbn == 0 &&
        !(pbn == 160 && (MODE != 0 || p.epi == 1)) && !(pbm == 128 && p.M < 128)) {
        t.bm = pbm; t.bn = pbn;
        variant = pv;
    }
    if (slabs_per_image) *slabs_per_image = 0;
    p.stats = want_stats ? p.stats : nullptr;
    if (p.stats) {
        const bool ok = p.epi == 0 && batch == 1 && p.N <= 2048 && img_rows % 32 == 0;
        if (p.epi == 0 && batch == 1 && p.N <= 2048 && splits > 1) {      // reduce slabs: any image size
            if (slabs_per_image) *slabs_per_image = lcm_reduce_slabs(img_rows);
        } else if (ok) {
            if (slabs_per_image) *slabs_per_image = img_rows / 32;       // canonical 32-row slabs (igemm_epilogue)
        }
        if (!ok && !(p.epi == 0 && batch == 1 && p.N <= 2048 && splits > 1)) p.stats = nullptr;
    }
    const int code = t.bm * 1000 + t.bn;
    if constexpr (MODE == 0) if (p.ln_g) {     // LayerNorm-folded GEMM: its own instantiations (compile-time switch in the K loop)
        switch (code) {
            case 128128: return launch_cfg<128, 128, 0, 1>(p, batch, splits, variant, s);
            case 128064: return launch_cfg<128, 64, 0, 1>(p, batch, splits, variant, s);
            case 64128: return launch_cfg<64, 128, 0, 1>(p, batch, splits, variant, s);
            case 128160: return launch_cfg<128, 160, 0, 1>(p, batch, splits, variant, s);
            case 64160: return launch_cfg<64, 160, 0, 1>(p, batch, splits, variant, s);
            default: return launch_cfg<64, 64, 0, 1>(p, batch, splits, variant, s);
        }
    }
    switch (code) {
        case 128128: return launch_cfg<128, 128, MODE>(p, batch, splits, variant, s);
        case 128064: return launch_cfg<128, 64, MODE>(p, batch, splits, variant, s);
        case 64128: return launch_cfg<64, 128, MODE>(p, batch, splits, variant, s);
        case 128160: if constexpr (MODE == 0) return launch_cfg<128, 160, MODE>(p, batch, splits, variant, s);
        case 64160: if constexpr (MODE == 0) return launch_cfg<64, 160, MODE>(p, batch, splits, variant, s);
        default: return launch_cfg<64, 64, MODE>(p, batch, splits, variant, s);
    }
}

extern "C" int lcm_gemm_f16(const void* A, int lda, const void* A2, int lda2, int K1,
                            const void* W, const void* bias, const void* rowadd, int ld_rowadd, int rows_per_batch,
                            const void* res, int ldr, void* out, int ldo,
                            int M, int N, int K, int epilogue, float out_scale,
                            int batch, int64_t strideA, int64_t strideW, int64_t strideO, int img_rows,
                            void* stats_out, int* slabs_per_image, void* stream) {
    LCM_REQUIRE(A && W && out, "gemm: null pointer");
    LCM_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0, "gemm: bad shape M=%d N=%d K=%d batch=%d", M, N, K, batch);
    LCM_REQUIRE(K % 64 == 0, "gemm: K=%d must be a multiple of 64", K);
    LCM_REQUIRE(N % 64 == 0, "gemm: N=%d must be a multiple of 64", N);
    LCM_REQUIRE(lda % 8 == 0 && ldo % 4 == 0, "gemm: lda=%d (%%8) / ldo=%d (%%4) misaligned", lda, ldo);
    if (A2) LCM_REQUIRE(K1 > 0 && K1 < K && K1 % 64 == 0 && lda2 % 8 == 0, "gemm: bad split K1=%d lda2=%d", K1, lda2);
    if (res) LCM_REQUIRE(ldr % 4 == 0, "gemm: ldr=%d misaligned", ldr);
    if (rowadd) LCM_REQUIRE(rows_per_batch > 0 && ld_rowadd % 4 == 0, "gemm: bad rowadd");
    LCM_REQUIRE(epilogue >= 0 && epilogue <= 3, "gemm: unknown epilogue %d", epilogue);
    if (epilogue == 1) LCM_REQUIRE(!rowadd && !res && !A2, "gemm: GEGLU epilogue takes bias only");
    IgemmParams p = {};
    p.A = (const half_t*)A; p.A2 = (const half_t*)A2; p.W = (const half_t*)W;
    p.bias = (const half_t*)bias; p.rowadd = (const half_t*)rowadd; p.res = (const half_t*)res;
    p.out = (half_t*)out;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.lda2 = lda2; p.K1 = A2 ? K1 : K;
    p.ldo = ldo; p.ldr = ldr; p.ld_rowadd = ld_rowadd; p.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1;
    p.epi = epilogue; p.out_scale = out_scale;
    p.strideA = strideA; p.strideW = strideW; p.strideO = strideO;
    p.stats = (float*)stats_out;
    // strided (batched-matrix) calls never split: whether a request runs alone or in a batch must not change its K partition
    const bool allow_split = batch == 1 && strideA == 0 && strideW == 0 && strideO == 0;
    return launch_igemm<0>(p, batch, (hipStream_t)stream, img_rows, allow_split, stats_out != nullptr, slabs_per_image);
}

// LayerNorm -> Linear as ONE contraction (BasicTransformerBlock norm1->attn1.to_q|k|v, norm2->attn2.to_q, norm3->ff.net.0):
//   LN(x) W^T + b = rstd * (x (gamma (*) W)^T - mean * g) + c,   g[n] = sum_k (gamma (*) W)[n][k],  c[n] = sum_k beta[k] W[n][k] + b[n]
// W must hold gamma (*) W (fp16), ln_g / ln_c fp32 [N].  The row statistics are accumulated from the A fragments while the
// kernel walks K (each wave sees every k of its rows), so the LayerNorm costs no launch and no pass over HBM.  Never split
// over K.  Epilogues: LCM_EPI_NONE or LCM_EPI_GEGLU (ln_g / ln_c in the packed row order of W).
extern "C" int lcm_gemm_ln_f16(const void* A, int lda, const void* W, const void* ln_g, const void* ln_c, float eps,
                               void* out, int ldo, int M, int N, int K, int epilogue, int img_rows, void* stream) {
    LCM_REQUIRE(A && W && out && ln_g && ln_c, "gemm_ln: null pointer");
    LCM_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_ln: bad shape M=%d N=%d K=%d", M, N, K);
    LCM_REQUIRE(K % 64 == 0 && N % 64 == 0, "gemm_ln: N=%d K=%d must be multiples of 64", N, K);
    LCM_REQUIRE(lda % 8 == 0 && ldo % 4 == 0, "gemm_ln: lda=%d (%%8) / ldo=%d (%%4) misaligned", lda, ldo);
    LCM_REQUIRE(epilogue == 0 || epilogue == 1, "gemm_ln: epilogue %d (0 = none, 1 = GEGLU)", epilogue);
    IgemmParams p = {};
    p.A = (const half_t*)A; p.W = (const half_t*)W; p.out = (half_t*)out;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.K1 = K; p.ldo = ldo; p.rows_per_batch = 1;
    p.epi = epilogue; p.out_scale = 1.0f;
    p.ln_g = (const float*)ln_g; p.ln_c = (const float*)ln_c; p.ln_eps = eps;
    return launch_igemm<0>(p, 1, (hipStream_t)stream, img_rows, false, false, nullptr);
}

struct HaloParams {
    IgemmParams g;
    int C1;
    const float* gn_scale;
    const float* gn_shift;
    int silu;
    int H, W;
    int tiles_y, tiles_x;
};
int lcm_conv_halo_launch(HaloParams& hp, int B, hipStream_t s, int* slabs_per_image);

extern "C" int lcm_conv3x3_gn_f16(const void* in, int C1, const void* in2, int C2, const void* gn_scale,
                                  const void* gn_shift, int silu, const void* W, const void* bias, const void* rowadd,
                                  int ld_rowadd, const void* res, void* out, int B, int Hin, int Win, int Cout, int ups,
                                  void* stats_out, int* slabs_per_image, void* stream) {
    LCM_REQUIRE(in && W && out, "conv3x3_gn: null pointer");
    if (!in2) C2 = 0;
    const int Cin = C1 + C2;
    LCM_REQUIRE(B > 0 && Hin > 0 && Win > 0, "conv3x3_gn: bad shape");
    LCM_REQUIRE(C1 % 64 == 0 && C2 % 64 == 0 && Cout % 64 == 0, "conv3x3_gn: C1=%d C2=%d Cout=%d must be multiples of 64", C1, C2, Cout);
    LCM_REQUIRE((gn_scale == nullptr) == (gn_shift == nullptr), "conv3x3_gn: scale/shift must come together");
    if (rowadd) LCM_REQUIRE(ld_rowadd % 4 == 0, "conv3x3_gn: ld_rowadd misaligned");
    HaloParams hp = {};
    IgemmParams& p = hp.g;
    p.A = (const half_t*)in; p.A2 = (const half_t*)in2; p.W = (const half_t*)W; p.bias = (const half_t*)bias;
    p.rowadd = (const half_t*)rowadd; p.res = (const half_t*)res; p.out = (half_t*)out;
    // ups flags: +4 / +8 crop the upsampled output to 2*Hin-1 rows / 2*Win-1 columns (Upsample2D called with the odd-sized
    // skip's output_size: F.interpolate(size=2h-1, mode="nearest") picks source floor(d*h/(2h-1)) == d>>1 for every d)
    const int crop_h = (ups >> 2) & 1, crop_w = (ups >> 3) & 1;
    ups &= 3;
    LCM_REQUIRE(ups >= 0 && ups <= 2 && !(ups == 2 && gn_scale), "conv3x3_gn: ups=%d (2 = phase-packed weights, no fused GroupNorm)", ups);
    // (the conv sees ZERO padding beyond the cropped upsampled image, so the border outputs use fewer taps than the
    // pre-summed phase weights of ups=2 hold: odd targets take the loader-fused form, ups=1, with the plain 3x3 weights)
    LCM_REQUIRE(ups == 1 || (!crop_h && !crop_w), "conv3x3_gn: an odd output size needs ups=1 (plain 3x3 weights), got ups=%d", ups);
    p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.stride = 1; p.ups = ups;
    hp.H = ups ? 2 * Hin - crop_h : Hin; hp.W = ups ? 2 * Win - crop_w : Win;
    p.Hout = hp.H; p.Wout = hp.W;
    p.M = B * hp.H * hp.W; p.N = Cout; p.K = (ups == 2 ? 4 : 9) * Cin;
    p.ldo = Cout; p.ldr = Cout; p.ld_rowadd = ld_rowadd; p.rows_per_batch = hp.H * hp.W;
    p.epi = 0; p.out_scale = 1.0f; p.splits = 1;
    hp.C1 = C1; hp.gn_scale = (const float*)gn_scale; hp.gn_shift = (const float*)gn_shift; hp.silu = silu;
    p.stats = (float*)stats_out;
    if (slabs_per_image) *slabs_per_image = 0;
    const int hrc = lcm_conv_halo_launch(hp, B, (hipStream_t)stream, slabs_per_image);
    if (hrc < 0) return hrc;
    if (hrc != 0) {
        lcm_set_error("conv3x3_gn: no tile configuration for B=%d %dx%d Cin=%d Cout=%d", B, hp.H, hp.W, Cin, Cout);
        return LCM_EINVAL;
    }
    LCM_CHECK_LAUNCH("conv_halo");
    return LCM_OK;
}

extern "C" int lcm_conv3x3_f16(const void* in, const void* W, const void* bias,
                               const void* rowadd, int ld_rowadd, const void* res, void* out,
                               int B, int Hin, int Win, int Cin, int Cout, int stride, int ups,
                               void* stats_out, int* slabs_per_image, void* stream) {
    LCM_REQUIRE(in && W && out, "conv3x3: null pointer");
    LCM_REQUIRE(B > 0 && Hin > 0 && Win > 0, "conv3x3: bad shape");
    LCM_REQUIRE(Cin % 64 == 0 && Cout % 64 == 0, "conv3x3: Cin=%d Cout=%d must be multiples of 64", Cin, Cout);
    LCM_REQUIRE(stride == 1 || stride == 2, "conv3x3: stride %d", stride);
    LCM_REQUIRE(!(ups && stride != 1), "conv3x3: upsample needs stride 1");
    if (rowadd) LCM_REQUIRE(ld_rowadd % 4 == 0, "conv3x3: ld_rowadd misaligned");
    if (stride == 1 && (g_conv_impl == 1 || (ups & 3) == 2 || (ups & 12)))
        return lcm_conv3x3_gn_f16(in, Cin, nullptr, 0, nullptr, nullptr, 0, W, bias, rowadd, ld_rowadd, res, out, B, Hin, Win,
                                  Cout, ups, stats_out, slabs_per_image, stream);
    const int Hl = ups ? 2 * Hin : Hin, Wl = ups ? 2 * Win : Win;
    IgemmParams p = {};
    p.A = (const half_t*)in; p.W = (const half_t*)W; p.bias = (const half_t*)bias;
    p.rowadd = (const half_t*)rowadd; p.res = (const half_t*)res; p.out = (half_t*)out;
    p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.stride = stride; p.ups = ups;
    p.Hout = (Hl + 2 - 3) / stride + 1; p.Wout = (Wl + 2 - 3) / stride + 1;
    p.M = B * p.Hout * p.Wout; p.N = Cout; p.K = 9 * Cin;
    p.ldo = Cout; p.ldr = Cout; p.ld_rowadd = ld_rowadd; p.rows_per_batch = p.Hout * p.Wout;
    p.epi = 0; p.out_scale = 1.0f;
    p.stats = (float*)stats_out;
    return launch_igemm<1>(p, 1, (hipStream_t)stream, p.Hout * p.Wout, true, stats_out != nullptr, slabs_per_image);
}
