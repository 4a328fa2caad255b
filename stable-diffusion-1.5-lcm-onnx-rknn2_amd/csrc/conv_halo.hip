// 3x3 convolution (stride 1, pad 1) as an implicit GEMM with an LDS-staged input halo tile, optionally with
// GroupNorm-apply + SiLU fused into the staging pass and the skip concat / nearest-2x upsample fused into the
// loader.  gfx950 (MI355X).
//
// Replaces ResnetBlock2D.{norm1->silu->conv1, norm2->silu->conv2} and Upsample2D.conv of UNet2DConditionModel /
// AutoencoderKL.decode (reached from backends/cuda_worker.py:221-229).
//
// One workgroup = a TH x TW patch of output pixels (BM = TH*TW = 128 or 64) x BN output channels.  K is walked
// channel-chunk major: for each 64-channel chunk the (TH+2) x (TW+2) input halo is staged ONCE into LDS
// (128-byte rows, XOR-swizzled; out-of-image pixels are zeros, so no masking later) and reused by all 9 taps;
// only the 64 x BN weight slice of the tap is streamed per K-step (LDS-DMA).  Against the row-gather igemm this
// cuts global->LDS traffic per K-step from (BM+BN) to (BN + 1.4*BM/9) rows and removes the per-tap address VALU.
// With XFORM the halo goes through registers: y = silu(x * scale[b][c] + shift[b][c]) (the GroupNorm affine
// folded per (image, channel)) is applied once per staged element -- 1.4x per element per n-tile instead of
// 9x -- and the normalised tensor is never written to HBM.
// PH = 1: "nearest-2x upsample -> conv3x3" evaluated as four 2x2 PHASE convolutions on the low-resolution input:
// output pixel (2y+py, 2x+px) only ever sees input rows {y-1+py, y+py} and columns {x-1+px, x+px}, with the 3x3 taps
// that land on the same input pixel pre-summed at pack time (packing.pack_conv3x3_up2) -- 16 instead of 36
// multiply-adds per output and input channel.  The workgroup then owns a TH x TW patch of INPUT pixels and one
// phase (the 4 phases of a patch are adjacent tiles, so they share the halo in L2); weights are
// [phase][Cout][2][2][Cin].
// Single-buffered LDS (<= 40 KiB): 4 workgroups per CU cover the staging latency (measured faster than a
// deeper pipeline at 2 workgroups per CU for the large grids this kernel is used on).
// Wave layout, MFMA (16x16x32 f16, weights as the A operand) and epilogue are those of igemm.hip.
#include "igemm_common.h"

struct HaloParams {
    IgemmParams g;          // A/A2 = inputs, W, bias, rowadd, res, out, M,N,K, ldo, ldr, Hin/Win (physical), Cin, ups, splits, ws
    int C1;                 // channels taken from g.A; the remaining Cin - C1 from g.A2
    const float* gn_scale;  // [B][Cin] fp32 or null
    const float* gn_shift;
    int silu;
    int H, W;               // logical (post-upsample) image size == output size
    int tiles_y, tiles_x;
    int staged_epi;         // 1: residual in / result out through an LDS image of the tile, whole rows (tile_epilogue_staged)
};

static __device__ __attribute__((aligned(256))) half_t g_zero_page_h[128];

// Canonical GroupNorm-statistics slabs of a convolution output (see igemm_epilogue): a slab is the 32-pixel patch
// (32/TW rows x TW columns, TW fixed by the image width) that one fragment pair of a wave covers, indexed by its
// position in the image -- the same pixels whether the workgroup tile is 8 or 4 rows high.  Phase-decomposed upsample
// convolutions index (input patch, phase).  Patches past the last image row carry no slab.
template <int TH, int TW, int PH>
__device__ __forceinline__ void halo_slabs(int (&slab_of)[TH * TW / 64], int wm, int bimg, int y0, int x0, int IH, int IW, int phase) {
    constexpr int RS = 32 / TW;                       // rows per slab
    const int sx = (IW + TW - 1) / TW, sy = (IH + RS - 1) / RS;
#pragma unroll
    for (int bp = 0; bp < TH * TW / 64; ++bp) {
        const int y = y0 + wm * (TH / 2) + bp * RS;
        const int idx = (bimg * sy + y / RS) * sx + x0 / TW;
        slab_of[bp] = y < IH ? (PH ? idx * 4 + phase : idx) : -1;
    }
}
static inline int halo_slabs_per_image(int IH, int IW, int TW, bool ph) {
    const int rs = 32 / TW;
    return ((IH + rs - 1) / rs) * ((IW + TW - 1) / TW) * (ph ? 4 : 1);
}

// PF = 1 (XFORM only): the raw halo of the next chunk is prefetched into registers under the taps of the current one
// EPI = 1: the residual / result tile moves through an LDS image of the tile in whole rows (tile_epilogue_staged; plain launches)
template <int TH, int TW, int BN, int XFORM, int PH, int PF = 0, int EPI = 0>
__global__ __launch_bounds__(256, PF ? 2 : 3) void conv_halo_kernel(HaloParams hp) {
    static_assert(!PF || XFORM, "halo prefetch: register-staged (GroupNorm-fused) path only");
    constexpr int NT = PH ? 4 : 9;
    constexpr int BM = TH * TW;
    constexpr int TM = BM / 32, TN = BN / 32, RW = BN / 32;
    constexpr int HWD = TW + 2, HROWS = (TH + 2) * HWD, HROWS_PAD = (HROWS + 7) / 8 * 8;
    constexpr int NV = (HROWS_PAD * 8 + 255) / 256;      // halo 16-byte vectors per thread
    constexpr int XBYTES = HROWS_PAD * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xs = smem;
    char* wsm = smem + XBYTES;

    const IgemmParams& p = hp.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int tile = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
    const int mt4 = tile / p.ntiles, nt = tile - mt4 * p.ntiles;
    const int phase = PH ? (mt4 & 3) : 0, mt = PH ? (mt4 >> 2) : mt4;
    const int py = phase >> 1, px = phase & 1;
    const int IH = PH ? p.Hin : hp.H, IW = PH ? p.Win : hp.W;      // image the halo is cut from
    const int per_img = hp.tiles_y * hp.tiles_x;
    const int bimg = mt / per_img, tr = mt - bimg * per_img;
    const int ty = tr / hp.tiles_x, tx = tr - ty * hp.tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;
    const int n_base = nt * BN;
    const int C2 = p.Cin - hp.C1;

    // ---- per-thread halo vector descriptors (constant over the K loop) ----
    int h_pix[NV];       // physical pixel index (b*Hin + py)*Win + px, or -1 (zero fill)
    const int pos = tid & 7;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + 256 * i;
        const int hr = v >> 3;
        const int hy = hr / HWD, hx = hr - hy * HWD;
        int ly = y0 - 1 + hy, lx = x0 - 1 + hx;
        const bool ok = hr < HROWS && ly >= 0 && ly < IH && lx >= 0 && lx < IW;
        if (!PH && p.ups) { ly >>= 1; lx >>= 1; }
        h_pix[i] = ok ? (bimg * p.Hin + ly) * p.Win + lx : -1;
    }
    // glds path: LDS position is lane-linear (v*16) and the SOURCE chunk is swizzled;
    // register path: the source chunk is the thread's (tid&7) and the LDS position is swizzled:
    //   row hr = (tid>>3) + 32*i, so (hr & 7) = (tid>>3) & 7 for every i.
    const int x_lds0 = (tid >> 3) * 128 + ((pos ^ ((tid >> 3) & 7)) << 4);
    const int w_row0 = tid >> 3;
    const int w_schunk = (pos ^ (w_row0 & 7)) * 8;
    const half_t* wptr = p.W + (PH ? (long long)phase * p.N * p.K : 0ll) + (long long)(n_base + w_row0) * p.K + w_schunk;

    // ---- per-lane fragment row bases ----
    const int frow = lane & 15, fq = lane >> 4;
    // halo row of this lane's pixel in m-tile b is affine in b: rb0 + b * (16/TW) * HWD
    const int q0 = wm * TM * 16 + frow;
    const int rb0 = (q0 / TW) * HWD + (q0 % TW);
    constexpr int RB_STEP = (16 / TW) * HWD;

    f4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};

    const int nchunks = p.Cin >> 6;
    const int c_begin = (int)((long long)blockIdx.y * nchunks / p.splits);
    const int c_end = (int)((long long)(blockIdx.y + 1) * nchunks / p.splits);
    bool first = true;

    // XFORM: the raw halo of chunk c+1 is fetched into registers while the 9 taps of chunk c run (issued behind the barrier
    // of tap 1, consumed at the next tap 0): the global-memory latency of the register-staged path -- the tensors that take
    // this path do not fit the Infinity Cache -- is no longer exposed once per chunk.
    h8 hv[XFORM ? NV : 1];
    auto fetch_halo = [&](int c64) {
        if constexpr (XFORM != 0) {
            const int cb = c64 << 6;
            const bool second = cb >= hp.C1;
            const half_t* src_base = second ? p.A2 : p.A;
            const int src_ld = second ? C2 : hp.C1;
            const int src_c = second ? cb - hp.C1 : cb;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (h_pix[i] >= 0) v = *reinterpret_cast<const h8*>(src_base + (long long)h_pix[i] * src_ld + src_c + pos * 8);
                hv[i] = v;
            }
        }
    };
    if (PF && c_begin < c_end) fetch_halo(c_begin);

    for (int c64 = c_begin; c64 < c_end; ++c64) {
        const int cb = c64 << 6;
        const bool second = cb >= hp.C1;
        const half_t* src_base = second ? p.A2 : p.A;
        const int src_ld = second ? C2 : hp.C1;
        const int src_c = second ? cb - hp.C1 : cb;
        for (int tap = 0; tap < NT; ++tap) {
            if (!first) __syncthreads();          // all waves finished reading the W tile (and the halo when tap == 0)
            first = false;
            if (PF && tap == 1 && c64 + 1 < c_end) fetch_halo(c64 + 1);
            if (tap == 0) {
                if (XFORM) {
                    if (!PF) fetch_halo(c64);
                    const float* sc = hp.gn_scale + (long long)bimg * p.Cin + cb + pos * 8;
                    const float* sh = hp.gn_shift + (long long)bimg * p.Cin + cb + pos * 8;
                    float s8[8], t8[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) { s8[j] = sc[j]; t8[j] = sh[j]; }
#pragma unroll
                    for (int i = 0; i < NV; ++i) {
                        if (tid + 256 * i < HROWS_PAD * 8) {
                            h8 o = {0, 0, 0, 0, 0, 0, 0, 0};
                            if (h_pix[i] >= 0) {
#pragma unroll
                                for (int j = 0; j < 8; ++j) {
                                    float f = __builtin_fmaf((float)hv[i][j], s8[j], t8[j]);      // as gn_apply_kernel rounds
                                    if (hp.silu) f = silu_f(f);
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
                            const half_t* src = h_pix[i] >= 0
                                ? src_base + (long long)h_pix[i] * src_ld + src_c + ((pos ^ (hr & 7)) << 3)
                                : g_zero_page_h;
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                             (__attribute__((address_space(3))) void*)(xs + (wave + 4 * i) * 1024),
                                                             16, 0, 0);
                        }
                    }
                }
            }
            {   // weight slice of this tap: rows n_base.., 64 channels at (tap, cb)
                const half_t* wsrc = wptr + (long long)tap * p.Cin + cb;
#pragma unroll
                for (int i = 0; i < RW; ++i)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + (long long)(32 * i) * p.K),
                                                     (__attribute__((address_space(3))) void*)(wsm + (i * 32 + wave * 8) * 128),
                                                     16, 0, 0);
            }
            __syncthreads();      // drains the LDS-DMA (vmcnt(0)) and the ds_writes, then barrier

            const int dy = PH ? (tap >> 1) + py : tap / 3, dx = PH ? (tap & 1) + px : tap - (tap / 3) * 3;
            const int tapoff = dy * HWD + dx;
            const char* wsr = wsm + (wn * (BN / 2)) * 128;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                h8 xf[TM], wf[TN];
                const int c = kk * 4 + fq;
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    const int r = rb0 + b * RB_STEP + tapoff;
                    xf[b] = *reinterpret_cast<const h8*>(xs + r * 128 + ((c ^ (r & 7)) << 4));
                }
#pragma unroll
                for (int a = 0; a < TN; ++a) {
                    const int r = a * 16 + frow;
                    wf[a] = *reinterpret_cast<const h8*>(wsr + r * 128 + ((c ^ (r & 7)) << 4));
                }
#pragma unroll
                for (int a = 0; a < TN; ++a)
#pragma unroll
                    for (int b = 0; b < TM; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[a], xf[b], acc[a][b], 0, 0, 0);
            }
        }
    }

    int m_of[TM];
#pragma unroll
    for (int b = 0; b < TM; ++b) {
        const int q = (wm * TM + b) * 16 + frow;
        const int y = y0 + q / TW, x = x0 + q % TW;
        m_of[b] = (y < IH && x < IW) ? (PH ? (bimg * hp.H + 2 * y + py) * hp.W + 2 * x + px : (bimg * hp.H + y) * hp.W + x) : -1;
    }
    int slab_of[BM / 64];
    halo_slabs<TH, TW, PH>(slab_of, wm, bimg, y0, x0, IH, IW, phase);
    static_assert(XBYTES + BN * 128 >= BM * BN * 2, "the staged epilogue lays the output tile over the halo + weight buffers");
    if constexpr (EPI != 0) {
        tile_epilogue_staged<BM, BN>(p, acc, m_of, n_base, wm, wn, fq, slab_of, smem, [&](int q) {
            const int y = y0 + q / TW, x = x0 + q % TW;
            return (y < IH && x < IW) ? (PH ? (bimg * hp.H + 2 * y + py) * hp.W + 2 * x + px : (bimg * hp.H + y) * hp.W + x) : -1;
        });
    } else {
        igemm_epilogue<BM, BN>(p, acc, m_of, n_base + wn * (BN / 2), fq, 0, slab_of);
    }
}

// ------------------------------------------------------------------------------------------------
// Pipelined variant for launches that cannot put 3-4 workgroups on every CU (batch-1 UNet levels, split-K
// slices): with ~1 workgroup per CU nothing else hides the staging latency, so the weight slices run through a
// WS-deep LDS-DMA ring with a counted vmcnt (WS-2 slices stay in flight across the single raw barrier per
// K-step) and the halo is double-buffered, the next chunk's halo being fetched while the current 9 taps run.
// ------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void halo_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// TPS = taps per K-step: 1, or one whole kernel row (3; 2 for the 2x2 phase kernels).  Measured on the batch-1 launches
// (one wave per SIMD, tools/trace_conv.py): a one-tap step costs ~900 cycles for 8-16 MFMAs per wave -- barrier, LDS-DMA
// issue and two exposed ds_read latencies -- because nothing may cross the barrier; with a row of taps per step the
// ds_reads of tap i+1 run under the MFMAs of tap i and there is one barrier per row.
// SEG = 1: segmented accumulation (see igemm2_kernel): the canonical K partition (over 64-channel chunks) of a batched launch
// is kept in registers -- finished parts are added into `tot` in part order -- instead of fp32 slabs + a reduce launch.
template <int TH, int TW, int BN, int WS, int PH, int TPS, int SEG = 0, int EPI = 0>
__global__ __launch_bounds__(256, 2) void conv_halo_pipe_kernel(HaloParams hp) {
    constexpr int NT = PH ? 4 : 9;
    constexpr int G = NT / TPS;            // K-steps per 64-channel chunk
    static_assert(TPS == 1 || TPS == (PH ? 2 : 3), "taps per step: 1 or one kernel row");
    constexpr int BM = TH * TW;
    constexpr int TM = BM / 32, TN = BN / 32, RW = BN / 32;
    constexpr int HWD = TW + 2, HROWS = (TH + 2) * HWD, HROWS_PAD = (HROWS + 7) / 8 * 8;
    constexpr int NV = (HROWS_PAD * 8 + 255) / 256;
    constexpr int XBYTES = HROWS_PAD * 128, WBYTES = BN * 128, SBYTES = TPS * WBYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xs0 = smem;                      // two halo buffers
    char* wsm0 = smem + 2 * XBYTES;        // WS weight-step buffers (TPS tap slices each)

    const IgemmParams& p = hp.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int tile = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
    const int mt4 = tile / p.ntiles, nt = tile - mt4 * p.ntiles;
    const int phase = PH ? (mt4 & 3) : 0, mt = PH ? (mt4 >> 2) : mt4;
    const int py = phase >> 1, px = phase & 1;
    const int IH = PH ? p.Hin : hp.H, IW = PH ? p.Win : hp.W;      // image the halo is cut from
    const int per_img = hp.tiles_y * hp.tiles_x;
    const int bimg = mt / per_img, tr = mt - bimg * per_img;
    const int ty = tr / hp.tiles_x, tx = tr - ty * hp.tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;
    const int n_base = nt * BN;
    const int C2 = p.Cin - hp.C1;

    int h_pix[NV];
    const int pos = tid & 7;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int hr = (tid + 256 * i) >> 3;
        const int hy = hr / HWD, hx = hr - hy * HWD;
        int ly = y0 - 1 + hy, lx = x0 - 1 + hx;
        const bool ok = hr < HROWS && ly >= 0 && ly < IH && lx >= 0 && lx < IW;
        if (!PH && p.ups) { ly >>= 1; lx >>= 1; }
        h_pix[i] = ok ? (bimg * p.Hin + ly) * p.Win + lx : -1;
    }
    const int w_row0 = tid >> 3;
    const int w_schunk = (pos ^ (w_row0 & 7)) * 8;
    const half_t* wptr = p.W + (PH ? (long long)phase * p.N * p.K : 0ll) + (long long)(n_base + w_row0) * p.K + w_schunk;

    const int frow = lane & 15, fq = lane >> 4;
    const int q0 = wm * TM * 16 + frow;
    const int rb0 = (q0 / TW) * HWD + (q0 % TW);
    constexpr int RB_STEP = (16 / TW) * HWD;

    f4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};

    const int nchunks = p.Cin >> 6;
    const int c_begin = (int)((long long)blockIdx.y * nchunks / p.splits);
    const int c_end = (int)((long long)(blockIdx.y + 1) * nchunks / p.splits);
    const int T = (c_end - c_begin) * G;

    auto issue_halo = [&](int c64, int hb) {
        const int cb = c64 << 6;
        const bool second = cb >= hp.C1;
        const half_t* src_base = second ? p.A2 : p.A;
        const int src_ld = second ? C2 : hp.C1;
        const int src_c = second ? cb - hp.C1 : cb;
        char* xs = xs0 + hb * XBYTES;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if ((wave + 4 * i) * 64 < HROWS_PAD * 8) {
                const int hr = (tid + 256 * i) >> 3;
                const half_t* src = h_pix[i] >= 0
                    ? src_base + (long long)h_pix[i] * src_ld + src_c + ((pos ^ (hr & 7)) << 3)
                    : g_zero_page_h;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(xs + (wave + 4 * i) * 1024), 16, 0, 0);
            }
        }
    };
    auto issue_w = [&](int t, int wb) {     // t = flattened (chunk, tap group) step
        const int c64 = c_begin + t / G, tg = t - (t / G) * G;
#pragma unroll
        for (int tp = 0; tp < TPS; ++tp) {
            const half_t* wsrc = wptr + (long long)(tg * TPS + tp) * p.Cin + (c64 << 6);
            char* wsm = wsm0 + wb * SBYTES + tp * WBYTES;
#pragma unroll
            for (int i = 0; i < RW; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + (long long)(32 * i) * p.K),
                                                 (__attribute__((address_space(3))) void*)(wsm + (i * 32 + wave * 8) * 128), 16, 0, 0);
        }
    };

    // prologue: halo of the first chunk, then WS-1 weight slices
    if (T > 0) issue_halo(c_begin, 0);
#pragma unroll
    for (int s = 0; s < WS - 1; ++s)
        if (s < T) issue_w(s, s);

    f4 tot[SEG ? TN : 1][SEG ? TM : 1];
    int part = 0, kstep = 0, plen = T;     // SEG: K-steps of the current part (parts = ranges of 64-channel chunks)
    if constexpr (SEG) {
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < TM; ++b) tot[a][b] = (f4){0.f, 0.f, 0.f, 0.f};
        plen = (int)((long long)nchunks / p.seg_parts) * G;
    }
    int wb = 0;
    constexpr int LPS = RW * TPS;          // LDS-DMA instructions per thread per K-step
    for (int t = 0; t < T; ++t) {
        const int chunk = t / G, tg = t - chunk * G;
        // W(t) (and, being older, this chunk's halo) must have landed; min(WS-2, T-1-t) newer steps may stay in flight
        const int newer = T - 1 - t;
        if (WS >= 4 && newer >= 2) halo_wait_vmcnt<(WS >= 4 ? 2 : 0) * LPS>();
        else if (WS >= 3 && newer >= 1) halo_wait_vmcnt<(WS >= 3 ? 1 : 0) * LPS>();
        else halo_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (t + WS - 1 < T) issue_w(t + WS - 1, (wb + WS - 1) % WS);
        if (tg == 0 && c_begin + chunk + 1 < c_end) issue_halo(c_begin + chunk + 1, (chunk + 1) & 1);

        const char* xs = xs0 + (chunk & 1) * XBYTES;
        if constexpr (TPS == 1) {
            const int tap = tg;
            const int dy = PH ? (tap >> 1) + py : tap / 3, dx = PH ? (tap & 1) + px : tap - (tap / 3) * 3;
            const int tapoff = dy * HWD + dx;
            const char* wsr = wsm0 + wb * SBYTES + (wn * (BN / 2)) * 128;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                h8 xf[TM], wf[TN];
                const int c = kk * 4 + fq;
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    const int r = rb0 + b * RB_STEP + tapoff;
                    xf[b] = *reinterpret_cast<const h8*>(xs + r * 128 + ((c ^ (r & 7)) << 4));
                }
#pragma unroll
                for (int a = 0; a < TN; ++a) {
                    const int r = a * 16 + frow;
                    wf[a] = *reinterpret_cast<const h8*>(wsr + r * 128 + ((c ^ (r & 7)) << 4));
                }
#pragma unroll
                for (int a = 0; a < TN; ++a)
#pragma unroll
                    for (int b = 0; b < TM; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[a], xf[b], acc[a][b], 0, 0, 0);
            }
        } else {
            // a row of taps per step, software-pipelined over HALF taps (32 of the 64 channels of a tap): the fragments of
            // half-step h+1 are fetched from LDS while the MFMAs of half-step h run (one wave per SIMD has nobody else to hide
            // the ds_read latency behind).  Half a tap, not a whole one: lgkmcnt is a 4-bit counter, and with two whole taps
            // of fragment reads in flight (24 for the 128x64 tile) the compiler can only wait for ALL of them
            // (s_waitcnt lgkmcnt(0)) before the first MFMA -- which serialises exactly the two phases this is meant to overlap.
            h8 xf[2][TM], wf[2][TN];
            auto load_half = [&](int h, int buf) {
                const int tp = h >> 1, kk = h & 1;
                const int tapoff = (tg + (PH ? py : 0)) * HWD + tp + (PH ? px : 0);
                const char* wsr = wsm0 + wb * SBYTES + tp * WBYTES + (wn * (BN / 2)) * 128;
                const int c = kk * 4 + fq;
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    const int r = rb0 + b * RB_STEP + tapoff;
                    xf[buf][b] = *reinterpret_cast<const h8*>(xs + r * 128 + ((c ^ (r & 7)) << 4));
                }
#pragma unroll
                for (int a = 0; a < TN; ++a) {
                    const int r = a * 16 + frow;
                    wf[buf][a] = *reinterpret_cast<const h8*>(wsr + r * 128 + ((c ^ (r & 7)) << 4));
                }
            };
            load_half(0, 0);
#pragma unroll
            for (int h = 0; h < 2 * TPS; ++h) {
                if (h + 1 < 2 * TPS) load_half(h + 1, (h + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < TN; ++a)
#pragma unroll
                    for (int b = 0; b < TM; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[h & 1][a], xf[h & 1][b], acc[a][b], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        wb = (wb + 1 == WS) ? 0 : wb + 1;
        if constexpr (SEG) {
            if (++kstep == plen) {         // a part of the canonical K partition is complete: fold it in, in part order
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
                plen = ((int)((long long)(part + 1) * nchunks / p.seg_parts) - (int)((long long)part * nchunks / p.seg_parts)) * G;
            }
        }
    }

    int m_of[TM];
#pragma unroll
    for (int b = 0; b < TM; ++b) {
        const int q = (wm * TM + b) * 16 + frow;
        const int y = y0 + q / TW, x = x0 + q % TW;
        m_of[b] = (y < IH && x < IW) ? (PH ? (bimg * hp.H + 2 * y + py) * hp.W + 2 * x + px : (bimg * hp.H + y) * hp.W + x) : -1;
    }
    int slab_of[BM / 64];
    halo_slabs<TH, TW, PH>(slab_of, wm, bimg, y0, x0, IH, IW, phase);
    if constexpr (EPI != 0) {       // staged: the tile image lies over the (dead) halo buffers and weight ring
        static_assert(2 * XBYTES + WS * SBYTES >= BM * BN * 2, "staged epilogue: the ring holds the tile");
        auto row_of = [&](int q) {
            const int y = y0 + q / TW, x = x0 + q % TW;
            return (y < IH && x < IW) ? (PH ? (bimg * hp.H + 2 * y + py) * hp.W + 2 * x + px : (bimg * hp.H + y) * hp.W + x) : -1;
        };
        if constexpr (SEG) tile_epilogue_staged<BM, BN>(p, tot, m_of, n_base, wm, wn, fq, slab_of, smem, row_of);
        else tile_epilogue_staged<BM, BN>(p, acc, m_of, n_base, wm, wn, fq, slab_of, smem, row_of);
    } else {
        if constexpr (SEG) igemm_epilogue<BM, BN>(p, tot, m_of, n_base + wn * (BN / 2), fq, 0, slab_of);
        else igemm_epilogue<BM, BN>(p, acc, m_of, n_base + wn * (BN / 2), fq, 0, slab_of);
    }
}

// split-K combine kernel lives in igemm.hip
extern void lcm_launch_splitk_reduce(IgemmParams& p, hipStream_t s);
extern float* lcm_splitk_workspace(long long* bytes, hipStream_t s);
extern void lcm_tuning(int* target_wgs, int* max_splits, int* min_wgs);
extern int lcm_split_policy(int m_img, int sp);
extern bool lcm_plan_get(int kind, int M, int N, int K, int aux, int* bm, int* bn, int* splits, int* variant);

static int g_halo_pipe_below = 768;      // workgroup count under which the pipelined (WS=3) variant is used

extern "C" int lcm_set_halo_pipe_threshold(int wgs) { g_halo_pipe_below = wgs; return LCM_OK; }
static int g_halo_prefetch = 0;          // GroupNorm-fused conv: prefetch the next chunk's raw halo (A/B switch; bit-neutral)
extern "C" int lcm_set_halo_prefetch(int on) { g_halo_prefetch = on ? 1 : 0; return LCM_OK; }

template <int TH, int TW, int BN, int XFORM, int PH>
static void launch_halo(HaloParams& hp, hipStream_t s, int force) {     // force: 1 single-buffer, 2 pipelined, 3 pipelined with a row of taps per step, else by grid size
    constexpr int HROWS_PAD = ((TH + 2) * (TW + 2) + 7) / 8 * 8;
    hp.tiles_y = ((PH ? hp.g.Hin : hp.H) + TH - 1) / TH;
    hp.tiles_x = ((PH ? hp.g.Win : hp.W) + TW - 1) / TW;
    hp.g.ntiles = hp.g.N / BN;
    dim3 grid(hp.g.mtiles * hp.g.ntiles, hp.g.splits, 1);
    if constexpr (!XFORM) if (hp.g.seg_parts > 1) {     // segmented accumulation: the pipelined kernel (2 workgroups per CU: room for `tot`)
        // (a kernel row of taps per step double-buffers its fragments in registers: with the second accumulator set that only
        // fits the 64-wide tile; wider tiles take one tap per step)
        constexpr int TPS = (BN <= 64) ? (PH ? 2 : 3) : 1;
        constexpr int WS = 3;
        constexpr int smem = 2 * HROWS_PAD * 128 + WS * TPS * BN * 128;
        static_assert(smem <= 160 * 1024, "halo ring exceeds LDS");
        static LcmDevOnce attr_once;
        if (auto once_guard = attr_once.first()) {
            once_guard.check(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_pipe_kernel<TH, TW, BN, WS, PH, TPS, 1>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        }
        char nm[80];
        if (hp.staged_epi && hp.g.res && hp.g.splits == 1) {       // segmented launch with a residual: staged epilogue
            static LcmDevOnce attr_once_e;
            if (auto once_guard = attr_once_e.first()) {
                once_guard.check(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_pipe_kernel<TH, TW, BN, WS, PH, TPS, 1, 1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            }
            snprintf(nm, sizeof(nm), "conv_halo_pipe_kernel<%d, %d, %d, %d, %d, %d, 1, 1>", TH, TW, BN, WS, PH, TPS);
            lcm_prof_start(nm, s);
            hipLaunchKernelGGL((conv_halo_pipe_kernel<TH, TW, BN, WS, PH, TPS, 1, 1>), grid, dim3(256), smem, s, hp);
            lcm_prof_stop(s);
            return;
        }
        snprintf(nm, sizeof(nm), "conv_halo_pipe_kernel<%d, %d, %d, %d, %d, %d, 1>", TH, TW, BN, WS, PH, TPS);
        lcm_prof_start(nm, s);
        hipLaunchKernelGGL((conv_halo_pipe_kernel<TH, TW, BN, WS, PH, TPS, 1>), grid, dim3(256), smem, s, hp);
        lcm_prof_stop(s);
        return;
    }
    if constexpr (!XFORM && BN <= 128) if (force == 3) {
        // one kernel row of taps per K-step (plan variant 3): batch-1 launches, one workgroup per CU
        constexpr int TPS = PH ? 2 : 3;
        constexpr int WS = BN == 64 ? 3 : 2;
        constexpr int smem = 2 * HROWS_PAD * 128 + WS * TPS * BN * 128;
        static_assert(smem <= 160 * 1024, "row-step halo ring exceeds LDS");
        static LcmDevOnce attr_once;
        if (auto once_guard = attr_once.first()) {
            once_guard.check(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_pipe_kernel<TH, TW, BN, WS, PH, TPS>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        }
        char nm[64];
        snprintf(nm, sizeof(nm), "conv_halo_pipe_kernel<%d, %d, %d, %d, %d, %d>%s", TH, TW, BN, WS, PH, TPS, hp.g.splits > 1 ? " +splitk" : "");
        lcm_prof_start(nm, s);
        hipLaunchKernelGGL((conv_halo_pipe_kernel<TH, TW, BN, WS, PH, TPS>), grid, dim3(256), smem, s, hp);
        lcm_prof_stop(s);
        return;
    }
    if (!XFORM && force != 1 && (force == 2 || force == 3 || (long long)grid.x * grid.y < g_halo_pipe_below)) {
        constexpr int WS = 3;
        constexpr int smem = 2 * HROWS_PAD * 128 + WS * BN * 128;
        static LcmDevOnce attr_once;
        if (auto once_guard = attr_once.first()) {
            once_guard.check(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_pipe_kernel<TH, TW, BN, WS, PH, 1>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        }
        char nm[64];
        snprintf(nm, sizeof(nm), "conv_halo_pipe_kernel<%d, %d, %d, %d, %d, 1>%s", TH, TW, BN, WS, PH, hp.g.splits > 1 ? " +splitk" : "");
        lcm_prof_start(nm, s);
        hipLaunchKernelGGL((conv_halo_pipe_kernel<TH, TW, BN, WS, PH, 1>), grid, dim3(256), smem, s, hp);
        lcm_prof_stop(s);
        return;
    }
    constexpr int smem = HROWS_PAD * 128 + BN * 128;
    char nm[64];
    if constexpr (XFORM != 0) if (g_halo_prefetch) {
        snprintf(nm, sizeof(nm), "conv_halo_kernel<%d, %d, %d, %d, %d, 1>%s", TH, TW, BN, XFORM, PH, hp.g.splits > 1 ? " +splitk" : "");
        lcm_prof_start(nm, s);
        hipLaunchKernelGGL((conv_halo_kernel<TH, TW, BN, XFORM, PH, 1>), grid, dim3(256), smem, s, hp);
        lcm_prof_stop(s);
        return;
    }
    // staged only where there is a residual tile to fetch (measured: 950 -> 801 us with, 698 -> 724 us without, 8 x 512^2 x 128)
    const bool staged = hp.staged_epi && hp.g.splits == 1 && hp.g.res != nullptr;      // split launches store fp32 slabs: nothing to stage
    snprintf(nm, sizeof(nm), "conv_halo_kernel<%d, %d, %d, %d, %d, 0, %d>%s", TH, TW, BN, XFORM, PH, staged ? 1 : 0, hp.g.splits > 1 ? " +splitk" : "");
    lcm_prof_start(nm, s);
    if (staged) hipLaunchKernelGGL((conv_halo_kernel<TH, TW, BN, XFORM, PH, 0, 1>), grid, dim3(256), smem, s, hp);
    else hipLaunchKernelGGL((conv_halo_kernel<TH, TW, BN, XFORM, PH>), grid, dim3(256), smem, s, hp);
    lcm_prof_stop(s);
}

// Tile + split candidates of the halo conv.  fixed_splits < 0: choose the split factor as well (the canonical
// K-partition heuristic, evaluated on ONE image); >= 1: partition given, pick the most efficient tile that fills the chip.
struct HaloPick { int bm, bn, splits; };
static HaloPick halo_pick(int B, int IH, int IW, int PHM, int TW, long long M, int N, int nchunks, int fixed_splits) {
    int target, max_splits, min_wgs;
    lcm_tuning(&target, &max_splits, &min_wgs);
    const int cand[4][2] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}};
    HaloPick best = {0, 0, fixed_splits > 0 ? fixed_splits : 1};
    long long best_wgs = -1;
    for (int c = 0; c < 4; ++c) {
        const int cbm = cand[c][0], cbn = cand[c][1];
        if (N % cbn) continue;
        if (TW == 8 && cbm == 128) continue;
        const int th = cbm / TW;
        const long long mt = (long long)B * ((IH + th - 1) / th) * ((IW + TW - 1) / TW) * PHM;
        if ((long long)mt * cbm > 2ll * M && cbm == 128) continue;      // tiles that would be mostly padding
        const long long tiles = mt * (N / cbn);
        int sp = fixed_splits > 0 ? fixed_splits : 1;
        if (fixed_splits < 0 && tiles < min_wgs && nchunks >= 2) {
            sp = (int)((target + tiles - 1) / tiles);
            if (sp > nchunks) sp = nchunks;
            if (sp > max_splits) sp = max_splits;
        }
        const long long wgs = tiles * sp;
        if (wgs >= min_wgs) return {cbm, cbn, sp};
        if (wgs > best_wgs) { best_wgs = wgs; best = {cbm, cbn, sp}; }
    }
    return best;
}

static inline int halo_tw(int IW) { return (IW % 16 == 0 || IW > 16) ? 16 : 8; }

// canonical split factor of a stride-1 3x3 convolution with this PER-IMAGE shape (see igemm.hip, "What decides the numbers")
int lcm_canonical_splits_halo(int m_img, int N, int K, int IH, int IW, int W, int ph, int xform) {      // W: output width
    const int nchunks = (K / (ph ? 4 : 9)) >> 6;
    int pbm, pbn, psp, pv;
    // one partition per layer shape: the GroupNorm-fused form of a conv (chosen from the TOTAL tensor size, i.e. from the
    // batch) must produce the bits of the plain form, so both read the plain form's entry
    (void)xform;
    int sp = lcm_plan_get(2, m_img, N, K, W << 1, &pbm, &pbn, &psp, &pv)
                 ? psp : halo_pick(1, IH, IW, ph ? 4 : 1, halo_tw(IW), m_img, N, nchunks, -1).splits;
    if (sp > nchunks) sp = nchunks;
    return lcm_split_policy(m_img, sp);
}

static int g_staged_epi = 1;     // 1: plain (unsplit) launches move residual / result through an LDS image of the tile in whole
                                 // rows; 0: per-lane 8-byte pieces.  Bit-identical (A/B switch)
extern int g_staged_epi_gemm;
extern "C" int lcm_set_staged_epilogue(int on) { g_staged_epi = (on & 1) ? 1 : 0; g_staged_epi_gemm = (on & 2) ? 1 : 0; return LCM_OK; }

// returns 0 when launched, 1 when the shape is not handled here, < 0 on error
int lcm_conv_halo_launch(HaloParams& hp, int B, hipStream_t s, int* slabs_per_image) {
    IgemmParams& p = hp.g;
    hp.staged_epi = g_staged_epi;
    const bool ph = p.ups == 2;                      // phase-decomposed upsample conv: tiles walk the INPUT image, x4 phases
    const int IH = ph ? p.Hin : hp.H, IW = ph ? p.Win : hp.W, PHM = ph ? 4 : 1;
    const int TW = halo_tw(IW);
    long long ws_bytes = 0;
    p.ws = lcm_splitk_workspace(&ws_bytes, s);
    const int nchunks = p.Cin >> 6;
    const int m_img = p.M / B;
    p.img_rows = m_img;
    // the K partition: a function of the per-image problem only
    const int splits = p.ws ? lcm_canonical_splits_halo(m_img, p.N, p.K, IH, IW, hp.W, ph ? 1 : 0, hp.gn_scale ? 1 : 0) : 1;
    // tile / variant: launch parameters (plan entry of the total shape, else the occupancy heuristic)
    HaloPick hk = halo_pick(B, IH, IW, PHM, TW, p.M, p.N, nchunks, splits);
    int bm = hk.bm, bn = hk.bn, force = 0;
    {   // autotuned plan for this shape, if any (kind 2; aux = (W << 1) | xform)
        int pbm, pbn, psp, pv;
        if (lcm_plan_get(2, p.M, p.N, p.K, (hp.W << 1) | (hp.gn_scale ? 1 : 0), &pbm, &pbn, &psp, &pv) && p.N % pbn == 0 &&
            !(TW == 8 && pbm == 128)) {
            bm = pbm; bn = pbn;
            force = (pv >= 1 && pv <= 3) ? pv : 0;
        }
    }
    if (!bm) return 1;
    const int th = bm / TW;
    p.mtiles = B * ((IH + th - 1) / th) * ((IW + TW - 1) / TW) * PHM;
    p.splits = splits;
    p.seg_parts = 1;
    if (splits > 1 && !hp.gn_scale) {   // a batched launch that fills the chip unsplit keeps the canonical partition in registers
        extern int g_seg_mode;
        int tgt, mxs, min_wgs;
        lcm_tuning(&tgt, &mxs, &min_wgs);
        if (g_seg_mode == 1 || (g_seg_mode == 0 && (long long)p.mtiles * (p.N / bn) >= min_wgs)) { p.seg_parts = splits; p.splits = 1; }
    }
    if (p.splits > 1 && (long long)p.splits * p.M * p.N * 4 > ws_bytes) {
        lcm_set_error("split-K workspace too small: %d x %d x %d fp32 slabs need %lld MB, have %lld MB "
                      "(lcm_set_workspace / LCM_SPLITK_WS_MB)", p.splits, p.M, p.N,
                      ((long long)p.splits * p.M * p.N * 4 + (1 << 20) - 1) >> 20, ws_bytes >> 20);
        return LCM_EINVAL;
    }
    p.rg_kind = TW == 16 ? 1 : 2; p.rg_ph = ph ? 1 : 0; p.rg_IH = IH; p.rg_IW = IW; p.rg_OH = hp.H; p.rg_OW = hp.W;
    if (p.stats) {   // fused GroupNorm statistics of the output: canonical 32-pixel slabs (halo_slabs / splitk_reduce_kernel)
        if (p.N > 2048) p.stats = nullptr;
        else {
            if (slabs_per_image) *slabs_per_image = halo_slabs_per_image(IH, IW, TW, ph);
            LCM_STATS_FIT(p, halo_slabs_per_image(IH, IW, TW, ph), B, "conv3x3");
        }
    }
    const bool xf = hp.gn_scale != nullptr;
    if (ph && xf) return 1;
#define HALO_CASE(TH_, TW_, BN_)                                                           \
    if (th == TH_ && TW == TW_ && bn == BN_) {                                             \
        if (ph) launch_halo<TH_, TW_, BN_, 0, 1>(hp, s, force);                            \
        else if (xf) launch_halo<TH_, TW_, BN_, 1, 0>(hp, s, force);                       \
        else launch_halo<TH_, TW_, BN_, 0, 0>(hp, s, force);                               \
    } else
    HALO_CASE(8, 16, 128) HALO_CASE(8, 16, 64) HALO_CASE(4, 16, 128) HALO_CASE(4, 16, 64)
    HALO_CASE(8, 8, 128) HALO_CASE(8, 8, 64) HALO_CASE(8, 16, 160) HALO_CASE(4, 16, 160) HALO_CASE(8, 8, 160) { return 1; }
#undef HALO_CASE
    if (p.splits > 1) lcm_launch_splitk_reduce(p, s);
    return 0;
}
