// Stride-1 4x4 convolution forward (PFA_CONV_S1: the discriminator's c4, models.py:237-241) with the input PATCH kept in LDS.
//
// pfa_gather_gemm stages, for every (tap, channel chunk) K-step, one 128-byte piece per output pixel: with stride 1 the 16
// taps re-fetch the same input pixels 16 times and the K-loop is bound by the L2->LDS instruction rate.
// Here a workgroup owns an 8 x 32 patch of output pixels of one image (wider maps are tiled in x too) and 128 output channels.  It stages the
// (8+3) x (32+3) input pixels of a 64-channel chunk ONCE - double-buffered, 7 DMA pieces per wave and chunk, one per tap
// step of the previous chunk - and walks the 16 taps inside LDS; only the weight tile (128 rows x 128 B per step, 2 pieces)
// streams through a 3-stage ring with a counted vmcnt.  DMA instructions per K-step and wave: 2.4 instead of 6.
// A patch pixel occupies 144 B (128 + one unused piece): 16 consecutive pixels then fall on 16 distinct bank groups for
// every channel chunk without an XOR swizzle, so the tap shift is a wave-uniform address offset (one VALU add per read).
// The two waves of every SIMD (w and w + 4) run in opposite phases - fragment reads + DMA issue + counted wait | 32 MFMAs -
// swapping at every barrier, the late group one barrier behind (the staggered ring of gather_gemm.hip, which pays off
// once the read phase is this light).  c4 forward, 16 x 32x32x256 -> 31x31x512: 64 us against 89 us (1.0 PF/s).
//
// Called by pfa_gather_gemm for the launches it fits (conv_s1_patch_ok); results equal the implicit GEMM's up to the order
// of the float32 sums (channel chunk outer, tap inner instead of tap outer).
#include "patch_epilogue.h"

#include <type_traits>
#include <utility>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

#define PATCH_WAIT_LGKM(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory")
template <int OFF> __device__ __forceinline__ bf16x8 lds_read16o(unsigned addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
struct PatchK {
    const char* src;     // [nb][hin][win][cin] bf16
    const char* w;       // [n][16][cin] bf16
    char* out;           // [nb][hout][wout][n] bf16 (splitk == 1)
    float* slab;         // splitk > 1: float32 [splitk][nb*hout*wout][n] partial sums for the finish pass
    float* psum;         // optional [nb * tiles_y * tiles_x][n] partial sums / sums of squares (splitk == 1)
    float* psq;
    int nb, hin, win, cin, n, hout, wout, tiles_y, tiles_x, ntiles, nwg;
    int org;             // patch origin = output position - org: 1 = forward (pad 1), 2 = input gradient (pad 2, taps reversed)
    int flip;            // input gradient: patch offset k reads weight tap 15 - k
    int splitk, cper;    // K slices over the channel chunks, chunks per slice
};

constexpr int TY = 8, TX = 32;                    // output patch of a workgroup: 256 pixels
constexpr int PH = TY + 3, PW = TX + 3;           // input patch with halo
constexpr int PSLOTS = PH * PW;                   // 385 pixels
constexpr int PSTRIDE = 144;                      // bytes per patch pixel (see the header)
constexpr int PPASS = 7;                          // 7 x 512 lanes x 16 B >= 385 x 144 B
constexpr int PATCH_BYTES = PPASS * 512 * 16;     // 57,344 (padded: every lane of every pass writes)
constexpr int W_BYTES = 128 * 128;
constexpr int LDS_BYTES = 2 * PATCH_BYTES + 3 * W_BYTES;      // 163,840: the whole LDS of a CU

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void k_conv_s1_patch(const PatchK p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;                 // 4 x 2 waves, 64 pixels x 64 channels per wave
    const int lrow = lane & 15, kk = lane >> 4;
    unsigned wid = xcd_remap(blockIdx.x, p.nwg);             // an XCD takes a contiguous run: the n-tiles of a patch together
    const int nt = wid % p.ntiles; wid /= p.ntiles;
    const int split = wid % p.splitk; wid /= p.splitk;
    const int tx = wid % p.tiles_x; wid /= p.tiles_x;
    const int ty = wid % p.tiles_y, img = wid / p.tiles_y;
    const int oy0 = ty * TY, ox0 = tx * TX, n0 = nt * 128;
    const char* zero = (const char*)g_pfa_zero_page;

    // ---- patch staging: this lane's 7 pieces (9 pieces per pixel, the ninth unused)
    unsigned poff[PPASS];
    unsigned pok = 0;
#pragma unroll
    for (int j = 0; j < PPASS; ++j) {
        const int q = j * 512 + tid, slot = q / 9, g = q - slot * 9;
        const int py = slot / PW, px = slot - py * PW;
        const int iy = oy0 - p.org + py, ix = ox0 - p.org + px;
        const bool ok = g < 8 && slot < PSLOTS && (unsigned)iy < (unsigned)p.hin && (unsigned)ix < (unsigned)p.win;
        poff[j] = ok ? (unsigned)((((img * p.hin + iy) * p.win + ix) * p.cin + g * 8) * 2) : 0u;
        pok |= ok ? (1u << j) : 0u;
    }
    const int c0 = split * p.cper;                           // first channel chunk of this K slice
    auto stage_patch_piece = [&](int chunk, int j) {         // chunk: index inside the slice (selects the LDS buffer)
        const char* gp = ((pok >> j) & 1u) ? p.src + poff[j] + (c0 + chunk) * 128 : zero;
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gp), LDS_PTR(smem + (chunk & 1) * PATCH_BYTES + (j * 512 + wave * 64) * 16), 16, 0, 0);
    };
    // ---- weight staging: 2 pieces per step (rows r0 + 64 i), chunk slot cs holds source chunk cs ^ swz(row)
    const int cs = tid & 7, r0 = tid >> 3;
    const int wg = cs ^ ((r0 >> 1) & 7);
    unsigned woff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) woff[i] = (unsigned)(((n0 + r0 + 64 * i) * 16 * p.cin + wg * 8) * 2);
    auto stage_w = [&](int step, int buf) {                 // step = chunk * 16 + tap
        const int chunk = c0 + (step >> 4), tap = p.flip ? 15 - (step & 15) : (step & 15);
        char* wb = smem + 2 * PATCH_BYTES + buf * W_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const char* gp = p.w + woff[i] + (tap * p.cin + chunk * 64) * 2;
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gp), LDS_PTR(wb + (i * 512 + wave * 64) * 16), 16, 0, 0);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // activation fragment b of this wave = 16 consecutive output pixels of patch row wm*2 + b/2
    const unsigned lds0 = (unsigned)(size_t)LDS_PTR(smem);
    unsigned abyte[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) abyte[b] = ((wm * 2 + (b >> 1)) * PW + (b & 1) * 16 + lrow) * PSTRIDE + kk * 16;
    unsigned wfb[2];
    {
        const unsigned sw = (lrow >> 1) & 7;
#pragma unroll
        for (int h = 0; h < 2; ++h) wfb[h] = lds0 + 2 * PATCH_BYTES + (wn * 64 + lrow) * 128 + (((kk + 4 * h) ^ sw) << 4);
    }

    const int nchunks = p.cper, nsteps = nchunks * 16;
#pragma unroll
    for (int j = 0; j < PPASS; ++j) stage_patch_piece(0, j);
    stage_w(0, 0);
    stage_w(1, 1);                                           // nsteps >= 16

    // Step i, read phase: fragments of stage i; DMA of [one patch piece of the next chunk] + weight stage i+2; counted
    // wait for weight stage i+1 (everything older - the patch pieces too - has then landed); every read retired ->
    // barrier -> MFMA phase -> barrier.  Barrier numbering as in gather_gemm.hip's staggered ring: a stage is waited
    // for before the first barrier of the step before its use, a buffer is re-filled after the second barrier of the
    // step of its last use, which the other group reaches only after retiring its reads.
    bf16x8 wf[2][4], af[2][4];
    const bool late = wave >= 4;
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");          // patch 0 and weight stage 0 have landed
    __builtin_amdgcn_s_barrier();
    if (late) __builtin_amdgcn_s_barrier();
    for (int i = 0; i < nsteps; ++i) {
        const int chunk = i >> 4, tap = i & 15;
        __builtin_amdgcn_sched_barrier(0);
        {
            const unsigned tb = lds0 + (chunk & 1) * PATCH_BYTES + ((tap >> 2) * PW + (tap & 3)) * PSTRIDE;
            const unsigned rb = (i % 3) * W_BYTES;
            const unsigned w0 = wfb[0] + rb, w1 = wfb[1] + rb;
            wf[0][0] = lds_read16o<0>(w0); wf[0][1] = lds_read16o<2048>(w0); wf[0][2] = lds_read16o<4096>(w0); wf[0][3] = lds_read16o<6144>(w0);
#pragma unroll
            for (int b = 0; b < 4; ++b) af[0][b] = lds_read16o<0>(abyte[b] + tb);
            wf[1][0] = lds_read16o<0>(w1); wf[1][1] = lds_read16o<2048>(w1); wf[1][2] = lds_read16o<4096>(w1); wf[1][3] = lds_read16o<6144>(w1);
#pragma unroll
            for (int b = 0; b < 4; ++b) af[1][b] = lds_read16o<64>(abyte[b] + tb);
        }
        const bool pp = tap < PPASS && chunk + 1 < nchunks;
        if (pp) stage_patch_piece(chunk + 1, tap);
        if (i + 2 < nsteps) {
            stage_w(i + 2, (i + 2) % 3);
            if (pp) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        PATCH_WAIT_LGKM(0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int a = 0; a < 4; ++a) pin(wf[h][a]);
#pragma unroll
            for (int b = 0; b < 4; ++b) pin(af[h][b]);
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[h][a], af[h][b], acc[a][b], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    }
    if (!late) __builtin_amdgcn_s_barrier();

    // ---- epilogue: lane holds D[channel = .. + 4 kk + r][pixel = .. + lrow]; pixels outside the image (the 32nd column of
    //      a 31-wide map, rows past the last one) hold sums over real data and are masked here
    bool valid[4];
    size_t opix[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int oy = oy0 + wm * 2 + (b >> 1), ox = ox0 + (b & 1) * 16 + lrow;
        valid[b] = oy < p.hout && ox < p.wout;
        opix[b] = valid[b] ? (size_t)(img * p.hout + oy) * p.wout + ox : 0;
    }
    if (p.splitk > 1) {                                    // partial sums for the finish pass (rows = output pixels)
        const size_t npix = (size_t)p.nb * p.hout * p.wout;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (!valid[b]) continue;
            float* srow = p.slab + ((size_t)split * npix + opix[b]) * p.n;
#pragma unroll
            for (int a = 0; a < 4; ++a) *(f32x4*)(srow + n0 + wn * 64 + a * 16 + 4 * kk) = acc[a][b];
        }
        return;
    }
    if (p.psum) {
        __syncthreads();                                   // every wave is done with the staging buffers
        float* red = (float*)smem;                         // [2][4 wm][128]
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int b = 0; b < 4; ++b) { const float v = valid[b] ? acc[a][b][r] : 0.f; s += v; q += v * v; }
                s = row_sum16(s);
                q = row_sum16(q);
                if (lrow == 15) {
                    const int nl = wn * 64 + a * 16 + 4 * kk + r;
                    red[wm * 128 + nl] = s;
                    red[(4 + wm) * 128 + nl] = q;
                }
            }
        __syncthreads();
        if (tid < 128) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) { s += red[i * 128 + tid]; q += red[(4 + i) * 128 + tid]; }
            const size_t trow = ((size_t)img * p.tiles_y + ty) * p.tiles_x + tx;
            p.psum[trow * p.n + n0 + tid] = s;
            p.psq[trow * p.n + n0 + tid] = q;
        }
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        if (!valid[b]) continue;
        char* orow = p.out + opix[b] * p.n * 2;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int col = n0 + wn * 64 + a * 16 + 4 * kk;
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16_t)acc[a][b][r];
            *(bf16x4*)(orow + col * 2) = o;
        }
    }
}


// ---- the same walk on MFMA 32x32x16 (round 2).  With 16x16x32 operands a ds_read_b128 lane group mixes two k-chunks of 16
// pixels, which collide on the 144-byte pixel pitch: SQ_LDS_BANK_CONFLICT was 33 % of this kernel's LDS-active cycles.  With
// 32x32x16 all 16 lanes of a group read the SAME k-chunk of 16 (of 32) consecutive pixels: 16 distinct bank slots.  A wave
// owns 64 pixels (patch rows wm*2, wm*2+1) x 64 channels = 2 x 2 tiles; a step is four k-steps (16 reads, 16 MFMAs).  The
// accumulators leave through the whole-row epilogue of patch_epilogue.h (K-slice planes, statistics, bf16 rows).
template <int N> __device__ __forceinline__ void wait_vmc() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct Patch32K {
    GatherK g;           // src1, w, out1 / slab / psum / psq, n, splitk ... as pfa_gather_gemm prepared them
    int hin, win, cin, hout, wout, tiles_y, tiles_x, ntiles, nwg;
    int org, flip, cper;
    int order;           // work-list order of k_conv_s1_inplace (0: n-tile fastest, 1: n-tile slowest)
};

// BN: 128 or 64 output channels per workgroup (64: the launches whose 128-channel tiles alone would leave half the chip idle -
// c4's input gradient - run un-sliced on twice the workgroups instead of K-sliced through float32 partial planes)
template <int BN>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void k_conv_s1_patch32(const Patch32K q) {
    constexpr int FN = BN / 64, WP = BN / 64, WB = BN * 128;      // weight tiles per wave, DMA pieces per lane, bytes per stage
    const GatherK& p = q.g;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;                 // 4 x 2 waves, 64 pixels x 64 channels per wave
    const int lr = lane & 31, lh = lane >> 5;
    unsigned wid = xcd_remap(blockIdx.x, q.nwg);
    const int nt = wid % q.ntiles; wid /= q.ntiles;
    const int split = wid % p.splitk; wid /= p.splitk;
    const int tx = wid % q.tiles_x; wid /= q.tiles_x;
    const int ty = wid % q.tiles_y, img = wid / q.tiles_y;
    const int oy0 = ty * TY, ox0 = tx * TX, n0 = nt * BN;
    const char* zero = (const char*)g_pfa_zero_page;

    unsigned poff[PPASS];
    unsigned pok = 0;
#pragma unroll
    for (int j = 0; j < PPASS; ++j) {
        const int qq = j * 512 + tid, slot = qq / 9, g = qq - slot * 9;
        const int py = slot / PW, px = slot - py * PW;
        const int iy = oy0 - q.org + py, ix = ox0 - q.org + px;
        const bool ok = g < 8 && slot < PSLOTS && (unsigned)iy < (unsigned)q.hin && (unsigned)ix < (unsigned)q.win;
        poff[j] = ok ? (unsigned)((((img * q.hin + iy) * q.win + ix) * q.cin + g * 8) * 2) : 0u;
        pok |= ok ? (1u << j) : 0u;
    }
    const int c0 = split * q.cper;
    auto stage_patch_piece = [&](int chunk, int j) {
        const char* gp = ((pok >> j) & 1u) ? p.src1 + poff[j] + (c0 + chunk) * 128 : zero;
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gp), LDS_PTR(smem + (chunk & 1) * PATCH_BYTES + (j * 512 + wave * 64) * 16), 16, 0, 0);
    };
    const int cs = tid & 7, r0 = tid >> 3;
    const int wg = cs ^ ((r0 >> 1) & 7);
    unsigned woff[WP];
#pragma unroll
    for (int i = 0; i < WP; ++i) woff[i] = (unsigned)(((n0 + r0 + 64 * i) * 16 * q.cin + wg * 8) * 2);
    auto stage_w = [&](int step, int buf) {
        const int chunk = c0 + (step >> 4), tap = q.flip ? 15 - (step & 15) : (step & 15);
        char* wb = smem + 2 * PATCH_BYTES + buf * WB;
#pragma unroll
        for (int i = 0; i < WP; ++i) {
            const char* gp = p.w + woff[i] + (tap * q.cin + chunk * 64) * 2;
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gp), LDS_PTR(wb + (i * 512 + wave * 64) * 16), 16, 0, 0);
        }
    };

    f32x16 acc[FN][2];
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    const unsigned lds0 = (unsigned)(size_t)LDS_PTR(smem);
    int tpos[2];
    unsigned abyte[2];               // activation tile b: patch row wm*2 + b, columns lr; k-chunk lh (+ 32 s per k-step)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        tpos[b] = (wm * 2 + b) * TX + lr;
        abyte[b] = lds0 + ((wm * 2 + b) * PW + lr) * PSTRIDE + lh * 16;
    }
    unsigned wadr[4];                // weight rows wn*BN/2 + lr (+ 32 a: offset 4096), swizzled chunk of k-step s
    {
        const int row = wn * (BN / 2) + lr;                   // (row + 32 has the same (row >> 1) & 7)
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) wadr[s2] = lds0 + 2 * PATCH_BYTES + row * 128 + (((2 * s2 + lh) ^ ((row >> 1) & 7)) << 4);
    }

    const int nchunks = q.cper, nsteps = nchunks * 16;
#pragma unroll
    for (int j = 0; j < PPASS; ++j) stage_patch_piece(0, j);
    stage_w(0, 0);
    stage_w(1, 1);

    bf16x8 wf[FN][4], af[2][4];      // [tile][k-step]
    const bool late = wave >= 4;
    wait_vmc<WP>();                                           // patch 0 and weight stage 0 have landed
    __builtin_amdgcn_s_barrier();
    if (late) __builtin_amdgcn_s_barrier();
    for (int i = 0; i < nsteps; ++i) {
        const int chunk = i >> 4, tap = i & 15;
        __builtin_amdgcn_sched_barrier(0);
        {
            const unsigned tb = (chunk & 1) * PATCH_BYTES + ((tap >> 2) * PW + (tap & 3)) * PSTRIDE;
            const unsigned rb = (i % 3) * WB;
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                const unsigned wa = wadr[s2] + rb;
                wf[0][s2] = lds_read16o<0>(wa);
                if (FN > 1) wf[FN - 1][s2] = lds_read16o<4096>(wa);
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const unsigned aa = abyte[b] + tb;
                af[b][0] = lds_read16o<0>(aa); af[b][1] = lds_read16o<32>(aa); af[b][2] = lds_read16o<64>(aa); af[b][3] = lds_read16o<96>(aa);
            }
        }
        const bool pp = tap < PPASS && chunk + 1 < nchunks;
        if (pp) stage_patch_piece(chunk + 1, tap);
        if (i + 2 < nsteps) {
            stage_w(i + 2, (i + 2) % 3);
            if (pp) wait_vmc<WP + 1>(); else wait_vmc<WP>();
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        PATCH_WAIT_LGKM(0);
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
#pragma unroll
            for (int t = 0; t < FN; ++t) pin(wf[t][s2]);
            pin(af[0][s2]); pin(af[1][s2]);
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
            for (int a = 0; a < FN; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[a][s2], af[b][s2], acc[a][b], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    }
    if (!late) __builtin_amdgcn_s_barrier();

    // ---- epilogue: accumulators -> fp32 [position][channel] image in LDS -> whole rows (patch_epilogue.h)
    constexpr int RS = patch_image_stride<BN>();
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        char* row = smem + tpos[b] * RS;
#pragma unroll
        for (int a = 0; a < FN; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
                *(f32x4*)(row + (wn * (BN / 2) + a * 32 + 8 * g + 4 * lh) * 4) = v;
            }
    }
    __syncthreads();
    auto pix_of = [&](int r, bool& valid, size_t& opix) {
        const int oy = oy0 + r / TX, ox = ox0 + r % TX;
        valid = oy < q.hout && ox < q.wout;
        opix = valid ? ((size_t)img * q.hout + oy) * q.wout + ox : 0;
    };
    const size_t trow = ((size_t)img * q.tiles_y + ty) * q.tiles_x + tx;
    patch_epilogue_rows<BN>(p, n0, tid, split, trow, smem, pix_of);
}

// ---- the same kernel with the K loop of conv_patch2.hip / conv_tconv4.hip: ONE barrier per tap step, one fragment set
// re-read in place a step ahead (counted lgkmcnt), the 16 tap bodies compiled once each (tap offset = immediate).  The two-phase
// loop above spends ~1750 cycles per step on this tile, the in-place loop ~1540 on the stride-2 kernels.  160 KiB hold two
// patch buffers and THREE weight stages, so the weight slot is a run-time offset (four address adds per step) and a stage
// is prefetched two steps ahead: stage i+2 is issued at the start of step i into the slot stage i-1 left (its reads retired
// during step i-1) and waited for at the end of step i, before the barrier behind which step i+1 reads it ahead.
template <int OFF> __device__ __forceinline__ void s1_rd(unsigned addr, bf16x8& v) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(v) : "v"(addr), "n"(OFF));
}
constexpr int s1_tap_off(int v) { return ((v >> 2) * PW + (v & 3)) * PSTRIDE; }

// BN: 128 output channels per workgroup, or 64 (FN = 1: the launches whose 128-channel tiles alone would leave half the chip idle -
// c4's input gradient at batch 16 - run UN-sliced on twice the workgroups: no float32 partial planes, no finish launch)
template <int BN>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void k_conv_s1_inplace(const Patch32K q) {
    constexpr int FN = BN / 64, FM = 2, KS = 4, NR = FN + FM;
    constexpr int WB = BN * 128, WP = BN / 64;               // bytes of a weight stage, its DMA pieces per wave
    const GatherK& p = q.g;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int lr = lane & 31, lh = lane >> 5;
    // work list: an XCD (private 4 MiB L2) takes a contiguous run of it.  order 0: n-tile fastest - the workgroups that share a
    // patch sit together; order 1: n-tile slowest - an XCD streams ONE n-tile's weights (1 MiB for c4: resident in its L2)
    // and every patch is fetched by as many XCDs as there are n-tiles
    unsigned wid = xcd_remap(blockIdx.x, q.nwg);
    int nt;
    if (q.order == 0) { nt = wid % q.ntiles; wid /= q.ntiles; }
    else { const unsigned per = q.nwg / q.ntiles; nt = wid / per; wid -= nt * per; }
    const int split = wid % p.splitk; wid /= p.splitk;
    const int tx = wid % q.tiles_x; wid /= q.tiles_x;
    const int ty = wid % q.tiles_y, img = wid / q.tiles_y;
    const int oy0 = ty * TY, ox0 = tx * TX, n0 = nt * BN;
    const int c0 = split * q.cper;
    const int nsteps = q.cper * 16;

    // All DMA in the buffer form of pfa_common.h (buf_dma16): per-lane offsets fixed for the whole loop, the channel chunk
    // (and the tap, for the weights) in ONE scalar per piece; PFA_DMA_OOB where the hardware is to write zeros - out-of-image
    // pixels, the padding piece, and every piece of a chunk behind the slice's last one (those are still issued: all steps
    // issue and wait alike).  Weights first: their latency runs under the patch's address arithmetic.
    const int cs = tid & 7, r0 = tid >> 3;
    const int wg = cs ^ ((r0 >> 1) & 7);
    unsigned woff[WP];
#pragma unroll
    for (int i = 0; i < WP; ++i) woff[i] = (unsigned)(((n0 + r0 + 64 * i) * 16 * q.cin + wg * 8) * 2);
    const unsigned cin2 = (unsigned)q.cin * 2;
    const unsigned tap0 = q.flip ? 15u * cin2 : 0u;              // tap of position V: flip ? 15 - V : V
    const int tapd = q.flip ? -(int)cin2 : (int)cin2;
    auto stage_w = [&](unsigned chunk_off, auto v_, unsigned slot_off) {         // tap position V of the chunk with scalar chunk_off
        constexpr int V = decltype(v_)::value;
        char* wb = smem + 2 * PATCH_BYTES + slot_off;
        const unsigned soff = chunk_off + (unsigned)((int)tap0 + V * tapd);
#pragma unroll
        for (int i = 0; i < WP; ++i) buf_dma16(p.w, wb + (i * 512 + wave * 64) * 16, woff[i], soff);
    };
    unsigned ck_cur = (unsigned)c0 * 128, ck_nxt = q.cper > 1 ? ck_cur + 128 : PFA_DMA_OOB;      // this chunk's and the next one's scalar
    stage_w(ck_cur, std::integral_constant<int, 0>{}, 0);
    stage_w(ck_cur, std::integral_constant<int, 1>{}, WB);
    unsigned poff[PPASS];
#pragma unroll
    for (int j = 0; j < PPASS; ++j) {
        const int qq = j * 512 + tid, slot = qq / 9, g = qq - slot * 9;
        const int py = slot / PW, px = slot - py * PW;
        const int iy = oy0 - q.org + py, ix = ox0 - q.org + px;
        const unsigned ok = (unsigned)(g < 8) & (unsigned)(slot < PSLOTS) & (unsigned)((unsigned)iy < (unsigned)q.hin) & (unsigned)((unsigned)ix < (unsigned)q.win);
        const unsigned keep = 0u - ok;
        poff[j] = ((unsigned)((((img * q.hin + iy) * q.win + ix) * q.cin + g * 8) * 2) & keep) | (PFA_DMA_OOB & ~keep);
    }
    auto stage_patch_piece = [&](unsigned chunk_off, int buf, int j) {
        buf_dma16(p.src1, smem + buf * PATCH_BYTES + (j * 512 + wave * 64) * 16, poff[j], chunk_off);
    };
#pragma unroll
    for (int j = 0; j < PPASS; ++j) stage_patch_piece(ck_cur, 0, j);

    f32x16 acc[FN][FM];
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    const unsigned lds0 = (unsigned)(size_t)LDS_PTR(smem);
    int tpos[FM];
    unsigned abuf[2][FM];
#pragma unroll
    for (int b = 0; b < FM; ++b) {
        tpos[b] = (wm * 2 + b) * TX + lr;
        abuf[0][b] = lds0 + ((wm * 2 + b) * PW + lr) * PSTRIDE + lh * 16;
        abuf[1][b] = abuf[0][b] + PATCH_BYTES;
    }
    unsigned wadr[KS];
    {
        const int row = wn * (BN / 2) + lr;
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) wadr[s2] = lds0 + 2 * PATCH_BYTES + row * 128 + (((2 * s2 + lh) ^ ((row >> 1) & 7)) << 4);
    }
    bf16x8 fw[FN][KS], fa[FM][KS];
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) {
#pragma unroll
        for (int a = 0; a < FN; ++a) fw[a][s2] = bf16x8{};
#pragma unroll
        for (int b = 0; b < FM; ++b) fa[b][s2] = bf16x8{};
    }
    // fragment E (0, 1: weight tiles; 2, 3: activation tiles) of k-step S2 of the step at tap position V: weight addresses wv[]
    // (slot already added), activation bases ab[]
    auto read_frag = [&](auto v_, auto s2_, auto e_, const unsigned (&ab)[FM], const unsigned (&wv)[KS]) {
        constexpr int V = decltype(v_)::value, S2 = decltype(s2_)::value, E = decltype(e_)::value;
        if constexpr (E < FN) s1_rd<E * 4096>(wv[S2], fw[E][S2]);
        else s1_rd<s1_tap_off(V) + S2 * 32>(ab[E - FN], fa[E - FN][S2]);
    };
    auto read_kstep = [&](auto v_, auto s2_, const unsigned (&ab)[FM], const unsigned (&wv)[KS]) {
        [&]<int... E>(std::integer_sequence<int, E...>) {
            (read_frag(v_, s2_, std::integral_constant<int, E>{}, ab, wv), ...);
        }(std::make_integer_sequence<int, NR>{});
    };

    wait_vmc<0>();
    __builtin_amdgcn_s_barrier();
    unsigned acur[FM] = {abuf[0][0], abuf[0][1]}, aoth[FM] = {abuf[1][0], abuf[1][1]};
    // weight slot offsets of stage i (being computed), i+1 (read ahead during step i), i+2 (filled during step i)
    unsigned w_cur = 0, w_nxt = WB, w_fil = 2 * WB;
    {
        unsigned wv[KS];
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) wv[s2] = wadr[s2] + w_cur;
        read_kstep(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, acur, wv);
        read_kstep(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, acur, wv);
        read_kstep(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}, acur, wv);
        read_kstep(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{}, acur, wv);
    }

    // step at tap position V of chunk `chunk` (patch buffer chunk & 1); more: a step follows
    auto body = [&](int chunk, auto v_, const bool more) {
        constexpr int V = decltype(v_)::value, NV = (V + 1) & 15;
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        constexpr bool p_issued = V < PPASS;
        unsigned wv[KS];                                     // the next step's weight fragment addresses
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) wv[s2] = wadr[s2] + w_nxt;
        auto kstep = [&](auto s2_) {
            constexpr int S2 = decltype(s2_)::value;
            if (more) PATCH_WAIT_LGKM((KS - 1) * NR); else PATCH_WAIT_LGKM((KS - 1 - S2) * NR);
#pragma unroll
            for (int a = 0; a < FN; ++a) pin(fw[a][S2]);
#pragma unroll
            for (int b = 0; b < FM; ++b) pin(fa[b][S2]);
            __builtin_amdgcn_sched_barrier(0);
            // the k-step's reads and DMA in the shadows of its own MFMAs (see conv_patch2.hip): M(w0,a0) [DMA] M(w1,a0) R(a0)
            // M(w0,a1) R(w0) M(w1,a1) R(w1) R(a1); fragment index: 0 .. FN-1 weight tiles, FN .. activation tiles
            auto mfma = [&](auto a_, auto b_) {
                constexpr int A = decltype(a_)::value, B = decltype(b_)::value;
                acc[A][B] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[A][S2], fa[B][S2], acc[A][B], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            auto rd = [&](auto e_) {
                if (more) {
                    if (V == 15) read_frag(std::integral_constant<int, NV>{}, s2_, e_, aoth, wv);
                    else read_frag(std::integral_constant<int, NV>{}, s2_, e_, acur, wv);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
            using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
            mfma(I0{}, I0{});
            if (S2 == 0) stage_w(V + 2 < 16 ? ck_cur : ck_nxt, std::integral_constant<int, (V + 2) & 15>{}, w_fil);       // stage i+2
            if (S2 == 2 && p_issued) stage_patch_piece(ck_nxt, (chunk + 1) & 1, V < PPASS ? V : 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (FN == 2) { mfma(I1{}, I0{}); rd(I2{}); mfma(I0{}, I1{}); rd(I0{}); mfma(I1{}, I1{}); rd(I1{}); rd(I3{}); }
            else { rd(I1{}); mfma(I0{}, I1{}); rd(I0{}); rd(I2{}); }        // (one weight tile: M(w,a0) [DMA] R(a0) M(w,a1) R(w) R(a1))
        };
        kstep(std::integral_constant<int, 0>{});
        kstep(std::integral_constant<int, 1>{});
        kstep(std::integral_constant<int, 2>{});
        kstep(std::integral_constant<int, 3>{});
        // stage i+2 (issued at k-step 0) has landed; a patch piece issued behind it may stay in flight
        wait_vmc<p_issued ? 1 : 0>();
        const unsigned t = w_cur; w_cur = w_nxt; w_nxt = w_fil; w_fil = t;
    };
    for (int chunk = 0; chunk < q.cper; ++chunk) {
        const bool next = chunk + 1 < q.cper;
        [&]<int... V>(std::integer_sequence<int, V...>) {
            (body(chunk, std::integral_constant<int, V>{}, V < 15 || next), ...);
        }(std::make_integer_sequence<int, 16>{});
#pragma unroll
        for (int b = 0; b < FM; ++b) { const unsigned t = acur[b]; acur[b] = aoth[b]; aoth[b] = t; }
        ck_cur = ck_nxt;
        ck_nxt = chunk + 2 < q.cper ? ck_cur + 128 : PFA_DMA_OOB;
    }
    wait_vmc<0>();                               // (the pieces issued behind the slice's end)

    constexpr int RS = patch_image_stride<BN>();
    __syncthreads();
#pragma unroll
    for (int b = 0; b < FM; ++b) {
        char* row = smem + tpos[b] * RS;
#pragma unroll
        for (int a = 0; a < FN; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
                *(f32x4*)(row + (wn * (BN / 2) + a * 32 + 8 * g + 4 * lh) * 4) = v;
            }
    }
    __syncthreads();
    auto pix_of = [&](int r, bool& valid, size_t& opix) {
        const int oy = oy0 + r / TX, ox = ox0 + r % TX;
        valid = oy < q.hout && ox < q.wout;
        opix = valid ? ((size_t)img * q.hout + oy) * q.wout + ox : 0;
    };
    const size_t trow = ((size_t)img * q.tiles_y + ty) * q.tiles_x + tx;
    patch_epilogue_rows<BN>(p, n0, tid, split, trow, smem, pix_of);
}

}  // namespace

// Geometry the patch kernel is built for (the caller has already checked: bf16, one source, one unpadded output, no bias /
// activation).  grad = 0: PFA_CONV_S1 forward (output hin-1 x win-1); grad = 1: PFA_CONV_S1_DGRAD (output hin+1 x win+1).
bool pfa_conv_s1_patch_ok(int grad, int nb, int hin, int win, int cin, int n, int splitk, int* stat_tiles) {
    const int hout = grad ? hin + 1 : hin - 1, wout = grad ? win + 1 : win - 1;
    if (cin % 64 || n % 128 || hout < 1 || wout < 1 || splitk < 1 || (cin / 64) % splitk) return false;
    if (stat_tiles) *stat_tiles = nb * ((hout + TY - 1) / TY) * ((wout + TX - 1) / TX);
    return true;
}

int pfa_conv_s1_patch(int grad, const void* src, const void* w, void* out, float* slab, int splitk, float* psum, float* psq,
                      int nb, int hin, int win, int cin, int n, hipStream_t st) {
    PatchK k;
    k.src = (const char*)src; k.w = (const char*)w; k.out = (char*)out; k.slab = slab; k.psum = psum; k.psq = psq;
    k.nb = nb; k.hin = hin; k.win = win; k.cin = cin; k.n = n;
    k.hout = grad ? hin + 1 : hin - 1; k.wout = grad ? win + 1 : win - 1;
    k.org = grad ? 2 : 1; k.flip = grad ? 1 : 0;
    k.splitk = splitk; k.cper = cin / 64 / splitk;
    k.tiles_y = (k.hout + TY - 1) / TY; k.tiles_x = (k.wout + TX - 1) / TX; k.ntiles = n / 128;
    k.nwg = nb * k.tiles_y * k.tiles_x * k.ntiles * splitk;
    static std::atomic<bool> attr_done{false};   // (launches may come from several host threads)
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)k_conv_s1_patch, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_done = true;
    }
    hipLaunchKernelGGL(k_conv_s1_patch, dim3(k.nwg), dim3(512), LDS_BYTES, st, k);
    PFA_LAUNCHED("k_conv_s1_patch");
    return 0;
}

// Plan of a stride-1 launch on the patch kernel: 128-channel tiles, un-sliced when they give the chip ~a workgroup per CU
// (the caller's K slices are dropped: no float32 partial planes, the epilogue of the kernel does what the finish pass
// would), else with the caller's K slices.  64-channel tiles would run c4's input gradient un-sliced on 256 workgroups, but
// a step of this kernel's two-phase loop does not get shorter with half the MFMAs: 87 us against 73 us for the K-sliced
// launch + finish pass (tools/s1_patch_ab.py) - they stay behind pfa_set_knob(3, 7).
bool pfa_conv_s1_plan(int grad, int nb, int hin, int win, int cin, int n, int splitk_caller, int* bn, int* splitk, int* stat_tiles) {
    const int hout = grad ? hin + 1 : hin - 1, wout = grad ? win + 1 : win - 1;
    if (cin % 64 || n % 64 || hout < 1 || wout < 1) return false;
    // (buffer-form LDS-DMA: every in-range offset stays below 2 GiB)
    if ((long long)nb * hin * win * cin * 2 >= (1LL << 31) || (long long)n * 16 * cin * 2 >= (1LL << 31)) return false;
    const int tiles = nb * ((hout + TY - 1) / TY) * ((wout + TX - 1) / TX);
    int b = 0, sk = 1;
    // (knob 10: the threshold, 128.  c4's input gradient then runs un-sliced on 128 workgroups with the fused BatchNorm
    // backward in its own epilogue - no float32 partial planes (33.5 MB written and re-read), no finish launch.  Inside the
    // three-stream step that is faster (round 2: 3.306 vs 3.329 ms; round 3, three same-box pairs: 3.044 vs 3.065 ms), because
    // the other streams use the idle half of the chip; alone on the GPU the launch takes 113 us instead of 60 + 12, which is
    // what a one-stream kernel trace shows.  The timed step decides: 128; 192 restores the K-sliced plan.)
    const int want = pfa_knob(10) > 0 ? pfa_knob(10) : (pfa_knob(15) ? 192 : 128);      // knob 15: the launch is alone on the GPU
    if (pfa_knob(3) == 7) b = 64;                                 // (tests: 64-channel tiles whenever the geometry fits)
    else if (n % 128 == 0 && (long long)tiles * (n / 128) >= want) b = 128;
    // 64-channel tiles on the in-place loop (round 4): where 128-channel tiles would need K slices to fill the chip - c4's input
    // gradient when it runs alone: 256 un-sliced workgroups, no float32 partial planes (2 x 33.5 MB written and read back), no
    // finish launch.  (knob 3 = 9: off, the K-sliced plan of round 3.)
    else if ((long long)tiles * (n / 64) >= 192 && pfa_knob(3) != 9) b = 64;
    else if (n % 128 == 0) {
        b = 128;
        sk = splitk_caller < 1 ? 1 : splitk_caller;
        if ((cin / 64) % sk) return false;
    } else return false;
    if (bn) *bn = b;
    if (splitk) *splitk = sk;
    if (stat_tiles) *stat_tiles = tiles;
    return true;
}

// The 32x32x16 form of the kernel (the default): `k` as pfa_gather_gemm prepared it for the launch (k.splitk = the plan's;
// K-sliced: bias / statistics already moved to the finish pass).
template <int BN>
static int launch_s1_patch32(const Patch32K& q, hipStream_t st) {
    constexpr int LOOP_BYTES = 2 * PATCH_BYTES + 3 * BN * 128, OUT_BYTES = 256 * patch_image_stride<BN>();
    constexpr int LDS = LOOP_BYTES > OUT_BYTES ? LOOP_BYTES : OUT_BYTES;
    static_assert(LDS <= 163840, "LDS");
    static std::atomic<bool> attr_done{false};   // (launches may come from several host threads)
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)k_conv_s1_patch32<BN>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_done = true;
    }
    hipLaunchKernelGGL(k_conv_s1_patch32<BN>, dim3(q.nwg), dim3(512), LDS, st, q);
    PFA_LAUNCHED("k_conv_s1_patch32");
    return 0;
}

int pfa_conv_s1_patch32(int grad, const GatherK& k, int bn, int hin, int win, int cin, hipStream_t st) {
    Patch32K q;
    q.g = k;
    q.hin = hin; q.win = win; q.cin = cin;
    q.hout = grad ? hin + 1 : hin - 1; q.wout = grad ? win + 1 : win - 1;
    q.org = grad ? 2 : 1; q.flip = grad ? 1 : 0;
    q.cper = cin / 64 / k.splitk;
    q.order = pfa_knob(5) == 1 ? 1 : 0;
    q.tiles_y = (q.hout + TY - 1) / TY; q.tiles_x = (q.wout + TX - 1) / TX; q.ntiles = k.n / bn;
    q.nwg = k.nb * q.tiles_y * q.tiles_x * q.ntiles * k.splitk;
    if (pfa_knob(3) != 8) {                                  // knob 3 = 8: the two-phase loop (A/B)
        static std::atomic<bool> attr_done{false};   // (launches may come from several host threads)
        if (!attr_done) {
            (void)hipFuncSetAttribute((const void*)k_conv_s1_inplace<128>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
            (void)hipFuncSetAttribute((const void*)k_conv_s1_inplace<64>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
            attr_done = true;
        }
        if (bn == 128) hipLaunchKernelGGL(k_conv_s1_inplace<128>, dim3(q.nwg), dim3(512), LDS_BYTES, st, q);
        else hipLaunchKernelGGL(k_conv_s1_inplace<64>, dim3(q.nwg), dim3(512), LDS_BYTES, st, q);
        PFA_LAUNCHED("k_conv_s1_inplace");
        return 0;
    }
    return bn == 128 ? launch_s1_patch32<128>(q, st) : launch_s1_patch32<64>(q, st);
}
