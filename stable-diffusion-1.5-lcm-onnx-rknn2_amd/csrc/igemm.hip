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

// EPI = 1 (plain launches with a residual: MODE 0, no LayerNorm fold, no segments, one n-tile per workgroup): the residual tile
// arrives by LDS-DMA and the result leaves in whole rows through an LDS image of the tile laid over the ring (tile_epilogue_staged)
template <int BM, int BN, int MODE, int S, int LN = 0, int SEG = 0, int EPI = 0>
__global__ __launch_bounds__(256, S == 1 ? (BM + BN < 256 ? 4 : 3) : 2) void igemm2_kernel(IgemmParams p) {
    static_assert(!EPI || (MODE == 0 && !LN && !SEG && S * (BM + BN) * 128 >= BM * BN * 2), "staged epilogue: plain GEMM whose ring holds the tile");
    // SEG = 1: segmented accumulation.  The launch's canonical K partition has p.seg_parts parts but this (batched) launch
    // fills the chip without splitting: one workgroup walks all of K, keeps the running part in `acc` and adds the finished
    // parts into `tot` in part order -- the additions the split-K reduce makes, in registers, with no fp32 slabs in HBM.
    static_assert(!SEG || (S >= 2 && !LN), "segmented accumulation: ring variants of the plain kernel");
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

#pragma unroll
    for (int s = 0; s < S - 1; ++s)
        if (s < T) issue_tile(s, s);
    const __attribute__((address_space(3))) float* ln_lds = nullptr;
    if constexpr (ln) {       // this n-tile's ln_g | ln_c into LDS (behind the ring's stages), outside the K loop and under
                              // the latency of the first tiles just issued
        float* t = reinterpret_cast<float*>(smem + S * BUF);
        if (tid < BN) { t[tid] = p.ln_g[nt0 * BN + tid]; t[BN + tid] = p.ln_c[nt0 * BN + tid]; }
        __syncthreads();
        ln_lds = (const __attribute__((address_space(3))) float*)t;
    }

    f4 tot[SEG ? TN : 1][SEG ? TM : 1];
    int part = 0, nkp = nkr;               // SEG: k-tiles of the current part
    if constexpr (SEG) {
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < TM; ++b) tot[a][b] = (f4){0.f, 0.f, 0.f, 0.f};
        nkp = (int)((long long)nk_all / p.seg_parts);          // part 0 = k-tiles [0, nk_all / parts)
    }
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
        if constexpr (SEG) {
            if (++kstep == nkp) {          // a part of the canonical K partition is complete: fold it in, in part order
                kstep = 0;
#pragma unroll
                for (int a = 0; a < TN; ++a)
#pragma unroll
                    for (int b = 0; b < TM; ++b) {
                        tot[a][b][0] += acc[a][b][0]; tot[a][b][1] += acc[a][b][1];
                        tot[a][b][2] += acc[a][b][2]; tot[a][b][3] += acc[a][b][3];
                        acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};
                    }
                ++part;
                nkp = (int)((long long)(part + 1) * nk_all / p.seg_parts) - (int)((long long)part * nk_all / p.seg_parts);
            }
            continue;
        }
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
            } else if constexpr (EPI != 0) {       // (the launcher gives such a launch one n-tile per workgroup: the ring is dead here)
                tile_epilogue_staged<BM, BN>(p, acc, m_of, (nt0 + ni_cur) * BN, wm, wn, fq, slab_of, smem,
                                             [&](int q) { return m_base + q < p.M ? m_base + q : -1; });
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
    if constexpr (SEG) {
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
        igemm_epilogue<BM, BN>(p, tot, m_of, nt0 * BN + wn * (BN / 2), fq, z, slab_of);
    }
}

// Split-K combine: fixed-order sum of the fp32 slabs (bit-reproducible) + the epilogue of the main kernel, and optionally
// the fused GroupNorm statistics of the result -- written per CANONICAL 32-pixel slab, with the pixels of a slab visited
// and folded exactly as igemm_epilogue does it (two 16-pixel fragments per lane-row, then the 16-lane butterfly), so a
// layer's statistics are the same bits whether it ran split (this kernel), segmented in one workgroup, or unsplit.
// One wave = one slab x 16 channels: lane (l = lane & 15, fq = lane >> 4) owns pixel l of both fragments, channels 4fq..4fq+3.
__device__ __forceinline__ int reduce_pixel(const IgemmParams& p, int slab, int b, int l) {
    if (p.rg_kind == 0) {
        const int m = slab * 32 + b * 16 + l;
        return m < p.M ? m : -1;
    }
    const int TW = p.rg_kind == 1 ? 16 : 8, RS = 32 / TW;
    const int sx = (p.rg_IW + TW - 1) / TW, sy = (p.rg_IH + RS - 1) / RS;
    int phase = 0;
    if (p.rg_ph) { phase = slab & 3; slab >>= 2; }
    const int bimg = slab / (sy * sx), r = slab - bimg * (sy * sx);
    const int syi = r / sx, sxi = r - syi * sx;
    const int q = b * 16 + l;                          // pixel of the slab, row-major over its RS x TW patch
    const int y = syi * RS + q / TW, x = sxi * TW + q % TW;
    if (y >= p.rg_IH || x >= p.rg_IW) return -1;
    if (p.rg_ph) return (bimg * p.rg_OH + 2 * y + (phase >> 1)) * p.rg_OW + 2 * x + (phase & 1);
    return (bimg * p.rg_OH + y) * p.rg_OW + x;
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(IgemmParams p, int total_slabs) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63, ncg = p.N >> 4;
    const long long wv = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int slab = (int)(wv / ncg), cg = (int)(wv - (long long)slab * ncg);
    if (slab >= total_slabs) return;
    const int l = lane & 15, fq = lane >> 4;
    const int n = cg * 16 + fq * 4;
    const long long slab_stride = (long long)p.M * p.N;
    f4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) { h4 t = *reinterpret_cast<const h4*>(p.bias + n); bias4 = (f4){(float)t[0], (float)t[1], (float)t[2], (float)t[3]}; }
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int m = reduce_pixel(p, slab, b, l);
        if (m < 0) continue;
        const float* src = p.ws + (long long)m * p.N + n;
        f4 v = *reinterpret_cast<const f4*>(src);
        for (int s = 1; s < p.splits; ++s) {
            f4 t = *reinterpret_cast<const f4*>(src + s * slab_stride);
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
        for (int j = 0; j < 4; ++j) { const float f = (float)o[j]; ssum[j] += f; ssq[j] += f * f; }
    }
    if (p.stats) {
row16_sum8(ssum, ssq);
        if (l == 0) {
            float* dst = p.stats + ((long long)slab * p.N + n) * 2;
            *reinterpret_cast<f4*>(dst) = (f4){ssum[0], ssq[0], ssum[1], ssq[1]};
            *reinterpret_cast<f4*>(dst + 4) = (f4){ssum[2], ssq[2], ssum[3], ssq[3]};
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

// Ownership: an entry belongs to whoever registered it, identified by the workspace pointer.  Registering another pointer over
// a live entry is refused (two owners on one stream handle would silently share -- or free -- each other's slabs: torch hands
// its streams out of a 32-entry pool per device, so handles repeat); (ptr, bytes == 0) forgets the entry only if it still holds
// that pointer; (NULL, 0) forgets it unconditionally.
extern "C" int lcm_set_stream_workspace(void* stream, void* ptr, int64_t bytes) {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    auto it = g_stream_ws.find((hipStream_t)stream);
    if (ptr && bytes > 0) {
        if (it != g_stream_ws.end() && it->second.ptr != (float*)ptr) {
            lcm_set_error("set_stream_workspace: stream %p already carries the workspace %p of another owner (new %p): "
                          "every lane needs a stream of its own", stream, (void*)it->second.ptr, ptr);
            return LCM_EINVAL;
        }
        g_stream_ws[(hipStream_t)stream] = StreamWs{(float*)ptr, (long long)bytes};
    } else if (it != g_stream_ws.end() && (!ptr || it->second.ptr == (float*)ptr)) {
        g_stream_ws.erase(it);
    }
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
// number of canonical slabs of a launch's output (all images)
int lcm_total_slabs(const IgemmParams& p) {
    if (p.rg_kind == 0) return (p.M + 31) / 32;
    const int TW = p.rg_kind == 1 ? 16 : 8, RS = 32 / TW;
    const int per_img = ((p.rg_IH + RS - 1) / RS) * ((p.rg_IW + TW - 1) / TW) * (p.rg_ph ? 4 : 1);
    return (p.M / (p.rg_OH * p.rg_OW)) * per_img;
}

void lcm_launch_splitk_reduce(IgemmParams& p, hipStream_t s) {
    const int total = lcm_total_slabs(p);
    const long long waves = (long long)total * (p.N >> 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, p, total);
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
static int g_split_max_rows = 4096, g_split_cap = 8;
int g_seg_mode = 0;            // 0: segmented accumulation when the unsplit launch has >= min_wgs tiles; 1: whenever the partition
                               // has parts; 2: never (always split + reduce) -- all three give the same bits
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

extern "C" int lcm_set_seg_mode(int mode) {
    if (mode < 0 || mode > 2) { lcm_set_error("seg_mode: %d", mode); return LCM_EINVAL; }
    g_seg_mode = mode;
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

// Where the canonical partition may have parts: images of at most 4096 output rows (the 64x64 latent level and below), at
// most 8 parts.  A batched launch pays nothing for it (segmented accumulation keeps the parts in registers); what the bound
// limits is the fp32 slab traffic of the launches in between (2-4 images, too few tiles to go unsplit).
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

int g_staged_epi_gemm = 0;       // lcm_set_staged_epilogue bit 1: plain GEMMs with a residual take the staged epilogue too.  Off: measured
                                 // neutral to -3 % on the transformer shapes (tools/gemm_epi_ab.py: 22.3 / 22.3, 19.8 / 20.5, 48.2 / 49.2 us) --
                                 // their epilogue is not what bounds them; the 3x3 convolutions' is (conv_halo.hip)

template <int BM, int BN, int MODE, int S, int LN = 0, int SEG = 0>
static int launch_v2(IgemmParams& p, dim3 grid, hipStream_t s) {
    constexpr int smem = S * (BM + BN) * 128 + (LN ? 2 * BN * 4 : 0);       // LN: + this n-tile's ln_g | ln_c
    // the staged epilogue: a plain launch with a residual tile to fetch, the tile image fits the ring, strided z-batches excluded
    constexpr bool can_stage = MODE == 0 && !LN && !SEG && S * (BM + BN) * 128 >= BM * BN * 2;
    if constexpr (can_stage) if (g_staged_epi_gemm && p.res && p.splits == 1 && p.n_iters == 1 && p.epi == 0 && grid.z == 1) {
        static LcmDevOnce attr_once_e;
        if (auto once_guard = attr_once_e.first()) {
            once_guard.check(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm2_kernel<BM, BN, MODE, S, LN, SEG, 1>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        }
        char nm[64];
        snprintf(nm, sizeof(nm), "igemm2_kernel<%d, %d, %d, %d, %d, %d, 1>", BM, BN, MODE, S, LN, SEG);
        lcm_prof_start(nm, s);
        hipLaunchKernelGGL((igemm2_kernel<BM, BN, MODE, S, LN, SEG, 1>), grid, dim3(256), smem, s, p);
        lcm_prof_stop(s);
        return 0;
    }
    static LcmDevOnce attr_once;
    if (auto once_guard = attr_once.first()) {
        once_guard.check(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm2_kernel<BM, BN, MODE, S, LN, SEG>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    }
    char nm[64];
    snprintf(nm, sizeof(nm), "igemm2_kernel<%d, %d, %d, %d, %d, %d>%s", BM, BN, MODE, S, LN, SEG, p.splits > 1 ? " +splitk" : "");
    lcm_prof_start(nm, s);
    hipLaunchKernelGGL((igemm2_kernel<BM, BN, MODE, S, LN, SEG>), grid, dim3(256), smem, s, p);
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
    if constexpr (!LN) if (p.seg_parts > 1) {             // segmented accumulation: ring variants only
        if (variant < 2 || variant > 4) variant = (BM + BN <= 128) ? 4 : 2;
        if (variant == 2) launch_v2<BM, BN, MODE, 2, 0, 1>(p, grid, s);
        else if (variant == 3) launch_v2<BM, BN, MODE, 3, 0, 1>(p, grid, s);
        else if constexpr (BM + BN <= 192) launch_v2<BM, BN, MODE, 4, 0, 1>(p, grid, s);
        else launch_v2<BM, BN, MODE, 3, 0, 1>(p, grid, s);
        LCM_CHECK_LAUNCH("igemm");
        return LCM_OK;
    }
    if (variant == 0) {
        const int smem = 2 * (BM + BN) * 128;
        static LcmDevOnce attr_once;
        if (auto once_guard = attr_once.first()) {
            once_guard.check(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<BM, BN, MODE, LN>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, smem));
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
    p.rg_kind = 0;
    if (p.stats) {
        const bool ok = p.epi == 0 && batch == 1 && p.N <= 2048 && img_rows % 32 == 0;
        if (ok && slabs_per_image) *slabs_per_image = img_rows / 32;       // canonical 32-row slabs (igemm_epilogue / reduce)
        if (!ok) p.stats = nullptr;
        else LCM_STATS_FIT(p, img_rows / 32, p.M / img_rows, "gemm");
    }
    p.seg_parts = 1;
    if (splits > 1) {    // a batched launch that fills the chip unsplit keeps the canonical partition in registers instead
        const long long tiles = (long long)((p.M + t.bm - 1) / t.bm) * (p.N / t.bn) * batch;
        if (g_seg_mode == 1 || (g_seg_mode == 0 && tiles >= g_min_wgs)) { p.seg_parts = splits; splits = 1; }
    }
    if (splits > 1 && (long long)splits * p.M * p.N * 4 > wsb) {
        lcm_set_error("split-K workspace too small: %d x %d x %d fp32 slabs need %lld MB, have %lld MB "
                      "(lcm_set_workspace / LCM_SPLITK_WS_MB)", splits, p.M, p.N,
                      ((long long)splits * p.M * p.N * 4 + (1 << 20) - 1) >> 20, wsb >> 20);
        return LCM_EINVAL;
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
                            void* stats_out, int64_t stats_bytes, int* slabs_per_image, void* stream) {
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
    p.stats = (float*)stats_out; p.stats_cap = stats_bytes;
    LCM_REQUIRE(!stats_out || slabs_per_image, "gemm: stats_out needs slabs_per_image");
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

// Bytes that always suffice for the fused statistics of an [M, N] contraction output made of M / img_rows images (img_rows 0:
// one image): per image at most max(img_rows / 16, 256) + 64 slabs (canonical 32-pixel slabs incl. partial patches at the image
// border of a convolution, four phase slabs per patch of an upsampling one; a split-K reduce emits the same slabs).
extern "C" int64_t lcm_stats_bytes(int M, int N, int img_rows) {
    if (M <= 0 || N <= 0) return 0;
    if (img_rows <= 0 || M % img_rows) img_rows = M;
    const long long per_img = (img_rows / 16 > 256 ? img_rows / 16 : 256) + 64;
    return (int64_t)(per_img * (M / img_rows) * N * 2 * (long long)sizeof(float));
}

struct HaloParams {
    IgemmParams g;
    int C1;
    const float* gn_scale;
    const float* gn_shift;
    int silu;
    int H, W;
    int tiles_y, tiles_x;
    int staged_epi;
};
int lcm_conv_halo_launch(HaloParams& hp, int B, hipStream_t s, int* slabs_per_image);

extern "C" int lcm_conv3x3_gn_f16(const void* in, int C1, const void* in2, int C2, const void* gn_scale,
                                  const void* gn_shift, int silu, const void* W, const void* bias, const void* rowadd,
                                  int ld_rowadd, const void* res, void* out, int B, int Hin, int Win, int Cout, int ups,
                                  void* stats_out, int64_t stats_bytes, int* slabs_per_image, void* stream) {
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
    p.stats = (float*)stats_out; p.stats_cap = stats_bytes;
    LCM_REQUIRE(!stats_out || slabs_per_image, "conv3x3_gn: stats_out needs slabs_per_image");
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
                               void* stats_out, int64_t stats_bytes, int* slabs_per_image, void* stream) {
    LCM_REQUIRE(in && W && out, "conv3x3: null pointer");
    LCM_REQUIRE(B > 0 && Hin > 0 && Win > 0, "conv3x3: bad shape");
    LCM_REQUIRE(Cin % 64 == 0 && Cout % 64 == 0, "conv3x3: Cin=%d Cout=%d must be multiples of 64", Cin, Cout);
    LCM_REQUIRE(stride == 1 || stride == 2, "conv3x3: stride %d", stride);
    LCM_REQUIRE(!(ups && stride != 1), "conv3x3: upsample needs stride 1");
    if (rowadd) LCM_REQUIRE(ld_rowadd % 4 == 0, "conv3x3: ld_rowadd misaligned");
    if (stride == 1 && (g_conv_impl == 1 || (ups & 3) == 2 || (ups & 12)))
        return lcm_conv3x3_gn_f16(in, Cin, nullptr, 0, nullptr, nullptr, 0, W, bias, rowadd, ld_rowadd, res, out, B, Hin, Win,
                                  Cout, ups, stats_out, stats_bytes, slabs_per_image, stream);
    const int Hl = ups ? 2 * Hin : Hin, Wl = ups ? 2 * Win : Win;
    IgemmParams p = {};
    p.A = (const half_t*)in; p.W = (const half_t*)W; p.bias = (const half_t*)bias;
    p.rowadd = (const half_t*)rowadd; p.res = (const half_t*)res; p.out = (half_t*)out;
    p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.stride = stride; p.ups = ups;
    p.Hout = (Hl + 2 - 3) / stride + 1; p.Wout = (Wl + 2 - 3) / stride + 1;
    p.M = B * p.Hout * p.Wout; p.N = Cout; p.K = 9 * Cin;
    p.ldo = Cout; p.ldr = Cout; p.ld_rowadd = ld_rowadd; p.rows_per_batch = p.Hout * p.Wout;
    p.epi = 0; p.out_scale = 1.0f;
    p.stats = (float*)stats_out; p.stats_cap = stats_bytes;
    LCM_REQUIRE(!stats_out || slabs_per_image, "conv3x3: stats_out needs slabs_per_image");
    return launch_igemm<1>(p, 1, (hipStream_t)stream, p.Hout * p.Wout, true, stats_out != nullptr, slabs_per_image);
}
