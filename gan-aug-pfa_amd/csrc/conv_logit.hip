// The PatchGAN logit layer: Conv2d(C, 1, kernel 4, stride 1, padding 1) + bias (models.py:243), C = 512 - forward, input
// gradient and weight / bias gradient.
//
// One output channel makes this layer HBM-shaped (0.12 GMAC against a 16 MB activation): on the MFMA kernels, which pad the
// single column to a 16-column tile and slice K sixteen ways, its seven launches per step (three forwards, three input
// gradients, two weight gradients - plus finish, reduce and column-sum passes) took ~250 us, 6 % of the step, for 0.1 % of
// its FLOPs.  Here they are three plain vector kernels, each one pass over the activation:
//   k_logit_fwd    out(n, oy, ox) = bias + sum_{tap, c} x(n, oy-1+ky, ox-1+kx, c) w(tap, c): a block per (image, output row,
//                  16 columns) pulls its 4 x 19 input pixels into LDS with one round of LDS-DMA; a lane owns 8 channels
//                  (their 16 taps of weights in registers, v_dot2 arithmetic), a wave two neighbouring outputs.
//   k_logit_dgrad  g(n, y, x, c) = sum_tap dlogit(n, y+1-ky, x+1-kx) w(tap, c), FUSED with the backward of what feeds the
//                  layer: dy = act'(scale raw + shift) g and the BatchNorm-backward partial sums (sum dy, sum dy xhat) -
//                  the separate pass over g, raw and dy disappears.  A thread owns 4 channels (their 16 taps in registers)
//                  and 8 pixels; raw / dy move as whole 1-KiB rows per wave.
//   k_logit_wgrad  dw(tap, c) += sum_pixels dlogit x(pixel + tap, c), db += sum dlogit: a thread owns 4 channels, all 16
//                  taps and 16 pixels (loaded once, all in flight); a block = 4 input rows of one image reduces in LDS and
//                  meets the other blocks in float atomics on whole rows of dw.
#include "pfa_common.h"
#include <atomic>

namespace {

template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    const int sh = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true);
    return v + __int_as_float(sh);
}
__device__ __forceinline__ float row_sum16(float v) {      // sum over a DPP row, valid in its last lane
    v = dpp_add<0x111>(v);
    v = dpp_add<0x112>(v);
    v = dpp_add<0x114>(v);
    v = dpp_add<0x118>(v);
    return v;
}

constexpr int LOGIT_WROW = 8192 + 16;   // a workgroup's weight-gradient partials [16 taps][512] + the bias gradient's, padded

struct LogitK {
    const char* x;           // [nb][h][w][c] bf16: the layer's input (forward, wgrad) / raw output of the layer below (dgrad)
    const bf16_t* wt;        // [16][c] bf16
    const float* bias;
    char* out;               // forward: logits [nb][h-1][w-1][ld]; dgrad: dy [nb][h][w][c]
    const char* dl;          // dlogits [nb][h-1][w-1][ld] bf16 (value in lane 0)
    const float *scale, *shift, *mean, *invstd;
    float* part;             // dgrad: [2][blocks][c] partial sums, or NULL
    float* dw;               // wgrad: [16][c] float32, +=
    float* db;
    float* wslab;            // wgrad: [workgroups][LOGIT_WROW] partial rows, or NULL (float atomics)
    int nb, h, w, c, ld, act;
    float slope;
    int xtiles, rgroups;
    // forward with the BCE-with-logits loss in its epilogue (pfa_logit_fwd_bce): dlogits out, one loss partial per workgroup
    char* bce_dl;
    float* bce_parts;
    float bce_target, bce_lscale, bce_gscale;
};

// ------------------------------------------------------------------ forward
// block = (image, output row, 16 output columns): its 4 x 19 input pixels - all 512 channels, 1 KiB a pixel - arrive in
// LDS by one round of LDS-DMA (one wave instruction per pixel); two blocks share a CU, so one's loads overlap the other's
// arithmetic.  Lane = 8 channels (its 16 taps of weights stay in registers), wave = two neighbouring outputs (they share
// 3 of their 4 input columns: 20 LDS reads for 32 tap products).
constexpr int FT = 16, FPW = FT + 3;

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

__device__ __forceinline__ float dot8(const bf16x8 a, const bf16x8 b, float c) {
    c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1), c, false);
    c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3), c, false);
    c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 4, 5), __builtin_shufflevector(b, b, 4, 5), c, false);
    c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 6, 7), __builtin_shufflevector(b, b, 6, 7), c, false);
    return c;
}
__device__ __forceinline__ float wave_sum(float v) {       // the same total in every lane
    const int r = __float_as_int(row_sum16(v));
    return __int_as_float(__builtin_amdgcn_readlane(r, 15)) + __int_as_float(__builtin_amdgcn_readlane(r, 31)) +
           __int_as_float(__builtin_amdgcn_readlane(r, 47)) + __int_as_float(__builtin_amdgcn_readlane(r, 63));
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4))) void k_logit_fwd(const LogitK p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [4][FPW] pixels x 1 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned b = xcd_remap(blockIdx.x, gridDim.x);
    const int xt = b % p.xtiles; b /= p.xtiles;
    const int ho = p.h - 1, wo = p.w - 1;
    const int oy = b % ho, n = b / ho;
    const int ox0 = xt * FT;
    const char* zero = (const char*)g_pfa_zero_page + (lane & 15) * 16;
    for (int s = wave; s < 4 * FPW; s += 8) {
        const int r = s / FPW, j = s - r * FPW;
        const int y = oy - 1 + r, x = ox0 - 1 + j;
        const bool ok = (unsigned)y < (unsigned)p.h && (unsigned)x < (unsigned)p.w;
        const char* gp = ok ? p.x + ((size_t)(n * p.h + y) * p.w + x) * 1024 + lane * 16 : zero;
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gp), LDS_PTR(smem + s * 1024), 16, 0, 0);
    }
    bf16x8 wr[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) wr[t] = *(const bf16x8*)(p.wt + (size_t)t * 512 + lane * 8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        bf16x8 xv[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) xv[j] = *(const bf16x8*)(smem + ((r * FPW + 2 * wave + j) * 1024 + lane * 16));
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            if (j < 4) a0 = dot8(xv[j], wr[r * 4 + j], a0);
            if (j >= 1) a1 = dot8(xv[j], wr[r * 4 + j - 1], a1);
        }
    }
    a0 = wave_sum(a0);
    a1 = wave_sum(a1);
    const int ox = ox0 + 2 * wave;
    const float bias = p.bias ? p.bias[0] : 0.f;
    // lane 0 / lane 1 write the two logits: channel 0 the value, the pad channels zeros
    float term = 0.f;
    if (lane < 2 && ox + lane < wo) {
        bf16x8 o = {};
        o[0] = (bf16_t)((lane ? a1 : a0) + bias);
        const size_t off = (((size_t)(n * ho + oy) * wo + ox + lane) * p.ld) * 2;
        char* dst = p.out + off;
        *(bf16x8*)dst = o;
        for (int k = 8; k < p.ld; k += 8) *(bf16x8*)(dst + k * 2) = bf16x8{};
        if (p.bce_dl) {
            // BCEWithLogitsLoss on the STORED logit, as k_bce_logits forms it (train_gan.py:42,58,60,67): the loss term and
            // dloss/dlogit = gscale (sigmoid(x) - t) / count, in the dlogits layout (value in channel 0)
            const float x = (float)o[0], cnt = (float)((size_t)p.nb * ho * wo);
            term = fmaxf(x, 0.f) - x * p.bce_target + log1pf(expf(-fabsf(x)));
            bf16x8 g = {};
            g[0] = (bf16_t)(p.bce_gscale * (1.f / (1.f + expf(-x)) - p.bce_target) / cnt);
            char* gd = p.bce_dl + off;
            *(bf16x8*)gd = g;
            for (int k = 8; k < p.ld; k += 8) *(bf16x8*)(gd + k * 2) = bf16x8{};
        }
    }
    if (p.bce_parts) {                           // one partial per workgroup, its 16 terms added in a fixed order
        __shared__ float terms[16];
        if (lane < 2) terms[2 * wave + lane] = term;
        __syncthreads();
        if (tid == 0) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) t += terms[i];
            p.bce_parts[blockIdx.x] = p.bce_lscale * t / (float)((size_t)p.nb * ho * wo);
        }
    }
}

// ------------------------------------------------------------------ input gradient + fused backward of the layer below
// block = (image, input row y, 32-column tile); thread = (4-channel chunk tid & 127 [+128 k], pixel group tid >> 7)
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4))) void k_logit_dgrad(const LogitK p) {
    __shared__ float dls[4][40];                 // dlogits rows y+1-ky (ky 0..3), columns x0-2 .. x0+33 (+pad)
    __shared__ float red[2][4][512];
    const int tid = threadIdx.x, c4 = tid & 127, pg = tid >> 7;
    const unsigned wid = xcd_remap(blockIdx.x, gridDim.x);
    unsigned b = wid;
    const int xt = b % p.xtiles; b /= p.xtiles;
    const int y = b % p.h, n = b / p.h;
    const int x0 = xt * 32, ho = p.h - 1, wo = p.w - 1;
    if (tid < 4 * 36) {
        const int ky = tid / 36, j = tid - ky * 36;
        const int oy = y + 1 - ky, ox = x0 - 2 + j;
        float v = 0.f;
        if ((unsigned)oy < (unsigned)ho && (unsigned)ox < (unsigned)wo)
            v = (float)*(const bf16_t*)(p.dl + (((size_t)(n * ho + oy) * wo + ox) * p.ld) * 2);
        dls[ky][j] = v;
    }
    __syncthreads();
    const size_t tile = wid;
    const float gneg = p.act == PFA_ACT_LRELU ? p.slope : p.act == PFA_ACT_RELU ? 0.f : 1.f;     // act'(y) for y <= 0
    {
        const int cc = c4 * 4;                   // (512 channels: one pass)
        // weights as bf16 pairs over neighbouring taps (kx, kx + 1): g accumulates by v_dot2 against pairs of dlogits
        // (exact: both are bf16 values), and the 16 taps of 4 channels take 32 registers - two blocks a CU need <= 128
        bf16x2 wp[8][4];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const bf16x4 wa = *(const bf16x4*)(p.wt + (size_t)(2 * t) * p.c + cc);
            const bf16x4 wb = *(const bf16x4*)(p.wt + (size_t)(2 * t + 1) * p.c + cc);
#pragma unroll
            for (int e = 0; e < 4; ++e) wp[t][e] = bf16x2{wa[e], wb[e]};
        }
        float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.scale) {
            const f32x4 a = *(const f32x4*)(p.scale + cc), s2 = *(const f32x4*)(p.shift + cc);
#pragma unroll
            for (int e = 0; e < 4; ++e) { sc[e] = a[e]; sh[e] = s2[e]; }
        }
        // sum dy and sum dy*raw here; sum dy*xhat = invstd (sum dy*raw - mean sum dy) once at the end
        float ss[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
        const size_t rowoff = (((size_t)(n * p.h + y) * p.w) * p.c + cc) * 2;
        const char* rrow = p.x + rowoff;
        char* orow = p.out + rowoff;
        const unsigned pitch = (unsigned)p.c * 2;
        bf16x4 rv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int x = x0 + pg + 4 * j;
            rv[j] = *(const bf16x4*)(rrow + (x < p.w ? (unsigned)x * pitch : 0u));
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int xl = pg + 4 * j, x = x0 + xl;
            if (x >= p.w) continue;
            // g = sum_{ky,kx} dl(y+1-ky, x+1-kx) w(ky,kx): dls[ky][xl + 2 + 1 - kx]
            float g[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 4; ++ky)
#pragma unroll
                for (int kp = 0; kp < 2; ++kp) {
                    const bf16x2 dp = {(bf16_t)dls[ky][xl + 3 - 2 * kp], (bf16_t)dls[ky][xl + 2 - 2 * kp]};
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[e] = __builtin_amdgcn_fdot2_f32_bf16(dp, wp[ky * 2 + kp][e], g[e], false);
                }
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float r = (float)rv[j][e];
                const float yv = r * sc[e] + sh[e];
                const float d = (yv > 0.f ? 1.f : gneg) * g[e];
                ss[e] += d;
                sq[e] += d * r;
                o[e] = (bf16_t)d;
            }
            *(bf16x4*)(orow + (unsigned)x * pitch) = o;
        }
        if (p.mean) {
            const f32x4 mu = *(const f32x4*)(p.mean + cc), is = *(const f32x4*)(p.invstd + cc);
#pragma unroll
            for (int e = 0; e < 4; ++e) sq[e] = is[e] * (sq[e] - mu[e] * ss[e]);
        }
        if (p.part) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { red[0][pg][(cc & 511) + e] = ss[e]; red[1][pg][(cc & 511) + e] = sq[e]; }
            __syncthreads();
            if (pg == 0) {
                const size_t nblocks = gridDim.x;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ci = (cc & 511) + e;
                    p.part[tile * p.c + cc + e] = red[0][0][ci] + red[0][1][ci] + red[0][2][ci] + red[0][3][ci];
                    p.part[(nblocks + tile) * p.c + cc + e] = red[1][0][ci] + red[1][1][ci] + red[1][2][ci] + red[1][3][ci];
                }
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------ weight and bias gradient
// block = (image, four INPUT rows), 1024 threads; thread = (4-channel chunk tid & 127, pixel group tid >> 7 = row pair x
// columns xg + 4 j) with all 16 tap accumulators of its channels: every input pixel is loaded once, 16 loads per thread in
// flight.  The eight pixel groups meet in LDS (plain adds: same-address LDS atomics from eight waves took 95 us), blocks in
// global float atomics on whole rows of dw.
__global__ __launch_bounds__(1024) void k_logit_wgrad(const LogitK p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* part = (float*)smem;                  // [4][16 taps][512]
    float* dls = part + 4 * 16 * 512;            // [7 rows][w + 4]: dlogits rows y0-2 .. y0+4, columns -2 .. w+1
    const int tid = threadIdx.x, c4 = tid & 127, pg = tid >> 7, rp = pg >> 2, xg = pg & 3;
    const int n = blockIdx.x / p.rgroups, y0 = (blockIdx.x % p.rgroups) * 4;
    const int ho = p.h - 1, wo = p.w - 1, pitch = p.w + 4;
    float bsum = 0.f;
    for (int q = tid; q < 7 * pitch; q += 1024) {
        const int r = q / pitch, j = q - r * pitch;
        const int oy = y0 - 2 + r, ox = j - 2;
        float v = 0.f;
        if ((unsigned)oy < (unsigned)ho && (unsigned)ox < (unsigned)wo)
            v = (float)*(const bf16_t*)(p.dl + (((size_t)(n * ho + oy) * wo + ox) * p.ld) * 2);
        dls[q] = v;
        if (r >= 2 && r < 6) bsum += v;          // the bias gradient counts every dlogits row once: rows y0 .. y0 + 3
    }
    __syncthreads();
    const int cc = c4 * 4;
    float acc[16][4];
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][e] = 0.f;
    for (int ry = 0; ry < 2; ++ry) {
        const int yl = rp * 2 + ry, y = y0 + yl;
        if (y >= p.h) break;
        const char* xrow = p.x + (((size_t)(n * p.h + y) * p.w) * 512 + cc) * 2;
        for (int x0 = 0; x0 < p.w; x0 += 32) {
            bf16x4 xv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int x = x0 + xg + 4 * j;
                xv[j] = *(const bf16x4*)(xrow + (size_t)(x < p.w ? x : 0) * 1024);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int x = x0 + xg + 4 * j;
                if (x >= p.w) continue;
                float xf[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) xf[e] = (float)xv[j][e];
                // input (y, x) meets output (y + 1 - ky, x + 1 - kx) under tap (ky, kx): dls row yl + 3 - ky, index x + 3 - kx
#pragma unroll
                for (int ky = 0; ky < 4; ++ky) {
                    const float* dr = dls + (yl + 3 - ky) * pitch + x;
#pragma unroll
                    for (int kx = 0; kx < 4; ++kx) {
                        const float d = dr[3 - kx];
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[ky * 4 + kx][e] += d * xf[e];
                    }
                }
            }
        }
    }
    // the eight pixel groups meet in two rounds over four [16][512] regions (every word has one owner per round)
    float* mine = part + (size_t)(pg & 3) * 8192 + cc;
    if (pg < 4) {
#pragma unroll
        for (int t = 0; t < 16; ++t) *(f32x4*)(mine + t * 512) = f32x4{acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
    }
    __syncthreads();
    if (pg >= 4) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const f32x4 v = *(const f32x4*)(mine + t * 512);
            *(f32x4*)(mine + t * 512) = f32x4{v[0] + acc[t][0], v[1] + acc[t][1], v[2] + acc[t][2], v[3] + acc[t][3]};
        }
    }
    __syncthreads();
    // slab: this workgroup's sums as one row of partials (k_logit_wgrad_reduce adds the rows in order: reproducible);
    // without a slab float atomics, whose order - and with it the last bits of dw - varies from run to run
    float* row = p.wslab ? p.wslab + (size_t)blockIdx.x * LOGIT_WROW : nullptr;
    for (int q = tid; q < 8192; q += 1024) {
        const float v = part[q] + part[8192 + q] + part[2 * 8192 + q] + part[3 * 8192 + q];
        if (row) row[q] = v; else atomicAdd(p.dw + q, v);
    }
    if (p.db || row) {
        __shared__ float bs[16];
        for (int off = 32; off > 0; off >>= 1) bsum += __shfl_xor(bsum, off);
        if ((tid & 63) == 0) bs[tid >> 6] = bsum;
        __syncthreads();
        if (tid < 16) {
            float v = bs[tid];
            v = row_sum16(v);
            if (tid == 15) { if (row) row[8192] = v; else atomicAdd(p.db, v); }
        }
    }
}

// dw[q] += sum over the workgroups' rows, in row order; element 8192 of a row is the bias gradient's partial
__global__ __launch_bounds__(256) void k_logit_wgrad_reduce(const float* __restrict__ slab, int rows, float* dw, float* db) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q > 8192 || (q == 8192 && !db)) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = 0;
    for (; r + 4 <= rows; r += 4) {
        s0 += slab[(size_t)r * LOGIT_WROW + q];
        s1 += slab[(size_t)(r + 1) * LOGIT_WROW + q];
        s2 += slab[(size_t)(r + 2) * LOGIT_WROW + q];
        s3 += slab[(size_t)(r + 3) * LOGIT_WROW + q];
    }
    for (; r < rows; ++r) s0 += slab[(size_t)r * LOGIT_WROW + q];
    const float t = (s0 + s1) + (s2 + s3);
    if (q == 8192) *db += t; else dw[q] += t;
}

}  // namespace

static constexpr size_t LOGIT_LDS_MAX = 159 * 1024;
static size_t logit_wgrad_lds(int wd) { return (size_t)7 * (wd + 4) * 4 + (size_t)4 * 16 * 512 * 4; }

extern "C" int pfa_logit_supported(int dtype, int c, int ld, int h, int w) {
    // (the thread mappings are built on 512 channels = 128 chunks of 4: ndf * 8 of models.py:232-241 at the default ndf)
    // (the width bound is the weight-gradient kernel's LDS budget, see pfa_logit_wgrad: the engine routes all three passes of
    // the layer on this predicate alone, so a map too wide for one of them must not take the others either)
    return dtype == PFA_BF16 && c == 512 && ld >= 8 && ld % 8 == 0 && h >= 2 && w >= 2 && logit_wgrad_lds(w) <= LOGIT_LDS_MAX &&
           pfa_knob(3) != 1 && pfa_knob(3) != 3;
}

static int logit_common(LogitK& k, int nb, int h, int w, int c, int ld) {
    PFA_REQUIRE(pfa_logit_supported(PFA_BF16, c, ld, h, w), "pfa_logit_*: needs bf16, 512 channels, ld a multiple of 8");
    PFA_REQUIRE(nb > 0 && (long long)nb * h * w * c * 2 < (1ll << 31) * 8, "pfa_logit_*: bad geometry");
    k.nb = nb; k.h = h; k.w = w; k.c = c; k.ld = ld;
    k.xtiles = (w + 31) / 32;
    return 0;
}

extern "C" int pfa_logit_fwd_nparts(int nb, int h, int wd) { return nb * (h - 1) * ((wd - 1 + FT - 1) / FT); }

static int logit_fwd_launch(const void* x, const void* w, const float* bias, void* logits, int nb, int h, int wd, int c, int ld,
                            void* dlogits, float* parts, float target, float lscale, float gscale, pfa_stream s);

extern "C" int pfa_logit_fwd(const void* x, const void* w, const float* bias, void* logits, int nb, int h, int wd, int c,
                             int ld, pfa_stream s) {
    return logit_fwd_launch(x, w, bias, logits, nb, h, wd, c, ld, nullptr, nullptr, 0.f, 0.f, 0.f, s);
}

extern "C" int pfa_logit_fwd_bce(const void* x, const void* w, const float* bias, void* logits, int nb, int h, int wd, int c,
                                 int ld, float target, float lscale, float gscale, float* parts, void* dlogits, pfa_stream s) {
    PFA_REQUIRE(parts && dlogits, "pfa_logit_fwd_bce: needs the partials buffer and the dlogits tensor");
    return logit_fwd_launch(x, w, bias, logits, nb, h, wd, c, ld, dlogits, parts, target, lscale, gscale, s);
}

static int logit_fwd_launch(const void* x, const void* w, const float* bias, void* logits, int nb, int h, int wd, int c, int ld,
                            void* dlogits, float* parts, float target, float lscale, float gscale, pfa_stream s) {
    PFA_REQUIRE(x && w && logits, "pfa_logit_fwd: null pointer");
    LogitK k{};
    if (int rc = logit_common(k, nb, h, wd, c, ld)) return rc;
    k.x = (const char*)x; k.wt = (const bf16_t*)w; k.bias = bias; k.out = (char*)logits;
    k.bce_dl = (char*)dlogits; k.bce_parts = parts; k.bce_target = target; k.bce_lscale = lscale; k.bce_gscale = gscale;
    k.xtiles = (wd - 1 + FT - 1) / FT;
    constexpr int LDS = 4 * FPW * 1024;
    static std::atomic<bool> attr_done{false};   // (launches may come from several host threads)
    if (!attr_done) {
        const hipError_t e = hipFuncSetAttribute((const void*)k_logit_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        PFA_REQUIRE(e == hipSuccess, "pfa_logit_fwd: %s", hipGetErrorString(e));
        attr_done = true;
    }
    hipLaunchKernelGGL(k_logit_fwd, dim3(nb * (h - 1) * k.xtiles), dim3(512), LDS, (hipStream_t)s, k);
    PFA_LAUNCHED("k_logit_fwd");
    return 0;
}

extern "C" int pfa_logit_dgrad_tiles(int nb, int h, int wd) { return nb * h * ((wd + 31) / 32); }

extern "C" int pfa_logit_dgrad(const void* dlogits, int ld, const void* w, const void* raw, const float* scale,
                               const float* shift, const float* mean, const float* invstd, int act, float slope, void* dy,
                               float* part, int nb, int h, int wd, int c, pfa_stream s) {
    PFA_REQUIRE(dlogits && w && raw && dy && (scale == nullptr) == (shift == nullptr) && (mean == nullptr) == (invstd == nullptr) &&
                (mean == nullptr) == (part == nullptr), "pfa_logit_dgrad: inconsistent arguments");
    LogitK k{};
    if (int rc = logit_common(k, nb, h, wd, c, ld)) return rc;
    k.dl = (const char*)dlogits; k.wt = (const bf16_t*)w; k.x = (const char*)raw; k.out = (char*)dy;
    k.scale = scale; k.shift = shift; k.mean = mean; k.invstd = invstd; k.part = part; k.act = act; k.slope = slope;
    hipLaunchKernelGGL(k_logit_dgrad, dim3(nb * h * k.xtiles), dim3(512), 0, (hipStream_t)s, k);
    PFA_LAUNCHED("k_logit_dgrad");
    return 0;
}

extern "C" size_t pfa_logit_wgrad_slab_floats(int nb, int h) { return (size_t)nb * ((h + 3) / 4) * LOGIT_WROW; }

extern "C" int pfa_logit_wgrad(const void* dlogits, int ld, const void* x, float* dw, float* db, int nb, int h, int wd, int c,
                               float* slab, pfa_stream s) {
    PFA_REQUIRE(dlogits && x && dw, "pfa_logit_wgrad: null pointer");
    LogitK k{};
    if (int rc = logit_common(k, nb, h, wd, c, ld)) return rc;
    k.dl = (const char*)dlogits; k.x = (const char*)x; k.dw = dw; k.db = db; k.wslab = slab;
    k.rgroups = (h + 3) / 4;
    const size_t lds = logit_wgrad_lds(wd);
    PFA_REQUIRE(lds <= LOGIT_LDS_MAX, "pfa_logit_wgrad: image too wide");
    static std::atomic<size_t> attr_lds{0};      // (the kernel also has a few static bytes: ask for what is used)
    if (attr_lds.load(std::memory_order_relaxed) < lds) {
        const hipError_t e = hipFuncSetAttribute((const void*)k_logit_wgrad, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        PFA_REQUIRE(e == hipSuccess, "pfa_logit_wgrad: %s", hipGetErrorString(e));
        attr_lds.store(lds, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(k_logit_wgrad, dim3(nb * k.rgroups), dim3(1024), lds, (hipStream_t)s, k);
    PFA_LAUNCHED("k_logit_wgrad");
    if (slab) {
        hipLaunchKernelGGL(k_logit_wgrad_reduce, dim3(33), dim3(256), 0, (hipStream_t)s, (const float*)slab, nb * k.rgroups, dw, db);
        PFA_LAUNCHED("k_logit_wgrad_reduce");
    }
    return 0;
}
