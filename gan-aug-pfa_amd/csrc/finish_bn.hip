// Whole-batch finish of a K-sliced convolution on a SMALL map: slice combine + BatchNorm (forward: statistics, running
// buffers, normalise, both activations; backward: activation' and the BatchNorm backward) in ONE launch.
//
// The layers at 16x16 positions and below (e5, e6, d7, d6, d5 of the default generator) are weight-bandwidth-bound GEMMs with
// a few hundred to a few thousand rows: their K slices meet in a finish pass, and the BatchNorm after them used to be
// two more dependent launches (k_bn_finalize + k_bn_act, k_bn_bwd_finalize + k_bn_bwd_apply) of ~5 us each - the minimum
// lifetime of a dependent launch - on 1-4 MB of data.  With at most 4096 rows one workgroup can own ALL rows of a 16-channel
// column group: it sums the slices (the sums stay in registers up to 2048 rows, are formed twice above that), reduces the
// batch statistics inside the workgroup - no cross-workgroup hand-off - and writes the finished tensors.  n / 16 workgroups
// of 1024 threads; a thread owns 8 channels (two 16-byte slab pieces, one 16-byte bf16 store) of up to 8 rows.
//
// pfa_gather_gemm(fuse_out = PFA_FUSE_DEFER) leaves the float32 partial planes [splitk][rows][n] in the slab for these.
#include "pfa_common.h"

namespace {

constexpr int FB_THREADS = 1024, FB_SLOTS = FB_THREADS / 2;      // row slots per pass
constexpr int FB_KEEP = 4, FB_MAXJ = 8;      // up to 2048 rows: slice sums stay in registers; up to 4096: summed twice

struct FinBnK {
    const float* slab;
    int splitk, rows, n;                  // planes, rows (= output pixels), columns (= slab row length)
    int serial;                           // A/B (pfa_set_knob(1, 1024)): one slice per round trip, as before round 3
    // ---- forward
    const float *gamma, *beta;
    float eps, momentum;
    int repeats;
    float *running_mean, *running_var;
    int64_t* nbt;
    char *raw, *out_l, *out_r;            // bf16 [rows][ld]; out_l / out_r optional
    int ld;
    float slope;
    float *scale, *shift, *mean, *invstd; // outputs (forward) / inputs (backward; scale & shift, mean & invstd optional pairs)
    // ---- backward
    int n1;                               // columns [0, n1) are a plain sum -> plain; [n1, n) the fused ones -> dr
    char* plain; int ld1;
    const char *fraw, *g2;                // bf16 [rows][ld2]: the node's stored output, second gradient (optional)
    int act1, act2;
    float *dgamma, *dbeta;                // += (optional)
    char* dr; int ld2;
};

__device__ __forceinline__ void load8(const float* p, float* v) {
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
}
__device__ __forceinline__ void store8(char* p, const float* v) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
    *(bf16x8*)p = o;
}

// t[e] = sum over the slices of row r, columns c0 .. c0+7 (ascending slice order: reproducible)
__device__ __forceinline__ void sum_slices(const FinBnK& p, int r, int c0, float* t) {
    const size_t plane = (size_t)p.rows * p.n;
    const float* src = p.slab + (size_t)r * p.n + c0;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = 0.f;
    // four slices' loads in flight at a time (a thread's chain of dependent round trips is what bounds these launches);
    // the additions stay in ascending slice order
    int s = 0;
    for (; !p.serial && s + 4 <= p.splitk; s += 4) {
        float u[4][8];
#pragma unroll
        for (int k = 0; k < 4; ++k) load8(src + (s + k) * plane, u[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] += u[k][e];
    }
    for (; s < p.splitk; ++s) {
        float u[8];
        load8(src + s * plane, u);
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] += u[e];
    }
}

// Two per-channel sums of every thread -> totals over the workgroup's rows: tot[stat][16 channels] (double) in LDS.
// Lanes of a wave that share the channel half (lane & 1) first, then the 16 waves.
__device__ __forceinline__ void block_totals(float (&a)[8], float (&b)[8], double (*tot)[16], float (*part)[2][2][8]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int off = 2; off < 64; off <<= 1) {
            a[e] += __shfl_xor(a[e], off);
            b[e] += __shfl_xor(b[e], off);
        }
    }
    if (lane < 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { part[wave][0][lane][e] = a[e]; part[wave][1][lane][e] = b[e]; }
    }
    __syncthreads();
    if (tid < 32) {
        const int stat = tid >> 4, ch = tid & 15;
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < FB_THREADS / 64; ++w) s += (double)part[w][stat][ch >> 3][ch & 7];
        tot[stat][ch] = s;
    }
    __syncthreads();
}

// ------------------------------------------------------------------ forward: statistics -> running buffers -> activations
template <bool KEEP>
__global__ __launch_bounds__(FB_THREADS) void k_finish_bn_fwd(const FinBnK p) {
    constexpr int NJ = KEEP ? FB_KEEP : FB_MAXJ;
    __shared__ float part[FB_THREADS / 64][2][2][8];
    __shared__ double tot[2][16];
    __shared__ float coef[2][16];
    const int tid = threadIdx.x, half = tid & 1, slot = tid >> 1;
    const int cb = blockIdx.x * 16, c0 = cb + half * 8;
    if (blockIdx.x == 0 && tid == 0 && p.nbt) *p.nbt += p.repeats;
    // (the finishing lanes fetch their coefficients before the reduction: see k_bn_finalize)
    float g32 = 1.f, b32 = 0.f, rm32 = 0.f, rv32 = 0.f;
    if (tid < 16) {
        if (p.gamma) g32 = p.gamma[cb + tid];
        if (p.beta) b32 = p.beta[cb + tid];
        if (p.running_mean) { rm32 = p.running_mean[cb + tid]; rv32 = p.running_var[cb + tid]; }
    }
    float v[KEEP ? NJ : 1][8];
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int r = slot + FB_SLOTS * j;
        if (r >= p.rows) continue;
        float t[8];
        sum_slices(p, r, c0, t);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s1[e] += t[e];
            s2[e] += t[e] * t[e];
            if (KEEP) v[j][e] = t[e];
        }
    }
    block_totals(s1, s2, tot, part);
    if (tid < 16) {
        const int ch = cb + tid;
        const double count = (double)p.rows;
        const double mean = tot[0][tid] / count;
        double var = tot[1][tid] / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const double invstd = 1.0 / sqrt(var + (double)p.eps);
        const double g = g32, b = b32;
        const float sc = (float)(g * invstd), sh = (float)(b - mean * g * invstd);
        coef[0][tid] = sc; coef[1][tid] = sh;
        p.scale[ch] = sc; p.shift[ch] = sh;
        if (p.mean) p.mean[ch] = (float)mean;
        if (p.invstd) p.invstd[ch] = (float)invstd;
        if (p.running_mean && p.repeats > 0) {
            const double unb = count > 1.0 ? var * (count / (count - 1.0)) : var;
            double rm = rm32, rv = rv32;
            for (int r = 0; r < p.repeats; ++r) {
                rm = (1.0 - p.momentum) * rm + p.momentum * mean;
                rv = (1.0 - p.momentum) * rv + p.momentum * unb;
            }
            p.running_mean[ch] = (float)rm;
            p.running_var[ch] = (float)rv;
        }
    }
    __syncthreads();
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = coef[0][half * 8 + e]; sh[e] = coef[1][half * 8 + e]; }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int r = slot + FB_SLOTS * j;
        if (r >= p.rows) continue;
        float t[8];
        if (KEEP) {
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = v[j][e];
        } else {
            sum_slices(p, r, c0, t);
        }
        const size_t off = ((size_t)r * p.ld + c0) * 2;
        float x[8], yl[8], yr[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            x[e] = (float)(bf16_t)t[e];                      // the stored conv output is what is normalised (as k_bn_act reads it)
            const float y = x[e] * sc[e] + sh[e];
            yl[e] = y > 0.f ? y : y * p.slope;
            yr[e] = y > 0.f ? y : 0.f;
        }
        store8(p.raw + off, x);
        if (p.out_l) store8(p.out_l + off, yl);
        if (p.out_r) store8(p.out_r + off, yr);
    }
}

// ------------------------------------------------------------------ backward: activation' (+ BatchNorm backward) of the node
template <bool KEEP>
__global__ __launch_bounds__(FB_THREADS) void k_finish_bn_bwd(const FinBnK p) {
    constexpr int NJ = KEEP ? FB_KEEP : FB_MAXJ;
    __shared__ float part[FB_THREADS / 64][2][2][8];
    __shared__ double tot[2][16];
    __shared__ float coef[3][16];
    const int tid = threadIdx.x, half = tid & 1, slot = tid >> 1;
    const int cb = blockIdx.x * 16, c0 = cb + half * 8;
    if (cb < p.n1) {                                         // a plain output: the sum itself (n1 is a multiple of 16)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int r = slot + FB_SLOTS * j;
            if (r >= p.rows) continue;
            float t[8];
            sum_slices(p, r, c0, t);
            store8(p.plain + ((size_t)r * p.ld1 + c0) * 2, t);
        }
        return;
    }
    const int f0 = c0 - p.n1, fb = cb - p.n1;                // column inside the fused output
    const bool bn = p.mean != nullptr;
    float g32 = 1.f, is32 = 1.f, mu32 = 0.f, dg32 = 0.f, db32 = 0.f;
    if (bn && tid < 16) {
        if (p.gamma) g32 = p.gamma[fb + tid];
        is32 = p.invstd[fb + tid];
        mu32 = p.mean[fb + tid];
        if (p.dgamma) dg32 = p.dgamma[fb + tid];
        if (p.dbeta) db32 = p.dbeta[fb + tid];
    }
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
    if (p.scale) { load8(p.scale + f0, sc); load8(p.shift + f0, sh); }
    const float neg1 = act_neg(p.act1, p.slope), neg2 = act_neg(p.act2, p.slope);
    // d = act1'(y) g + act2'(y) g2 of row r (y = scale x + shift, x = the node's stored output), and x itself
    auto row = [&](int r, float* d, float* x) {
        float t[8];
        sum_slices(p, r, c0, t);
        const size_t off = ((size_t)r * p.ld2 + f0) * 2;
        const bf16x8 xr = *(const bf16x8*)(p.fraw + off);
        bf16x8 gr = {};
        if (p.g2) gr = *(const bf16x8*)(p.g2 + off);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            x[e] = (float)xr[e];
            const float y = x[e] * sc[e] + sh[e];
            d[e] = act_grad_lin(y, neg1) * t[e];
            if (p.g2) d[e] += act_grad_lin(y, neg2) * (float)gr[e];
        }
    };
    float v[KEEP ? NJ : 1][8];
    float s1[8], s2[8];                                      // sum d, sum d x  (sum d xhat = invstd (sum d x - mean sum d))
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    if (bn || KEEP) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int r = slot + FB_SLOTS * j;
            if (r >= p.rows) continue;
            float d[8], x[8];
            row(r, d, x);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s1[e] += d[e];
                s2[e] += d[e] * x[e];
                if (KEEP) v[j][e] = d[e];
            }
        }
    }
    float k0[8], k1[8], k2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { k0[e] = 1.f; k1[e] = 0.f; k2[e] = 0.f; }
    if (bn) {
        block_totals(s1, s2, tot, part);
        if (tid < 16) {
            const int ch = fb + tid;
            const double sdy = tot[0][tid], count = (double)p.rows;
            const double sdyx = (double)is32 * (tot[1][tid] - (double)mu32 * sdy);      // sum d xhat
            if (p.dgamma) p.dgamma[ch] = dg32 + (float)sdyx;         // gradients accumulate into a zeroed arena
            if (p.dbeta) p.dbeta[ch] = db32 + (float)sdy;
            const double s = (double)g32 * is32;
            coef[0][tid] = (float)s;
            coef[1][tid] = (float)(-s * is32 * sdyx / count);
            coef[2][tid] = (float)(-s * sdy / count + s * is32 * mu32 * sdyx / count);
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) { k0[e] = coef[0][half * 8 + e]; k1[e] = coef[1][half * 8 + e]; k2[e] = coef[2][half * 8 + e]; }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int r = slot + FB_SLOTS * j;
        if (r >= p.rows) continue;
        float d[8], x[8];
        if (KEEP) {
            const bf16x8 xr = *(const bf16x8*)(p.fraw + ((size_t)r * p.ld2 + f0) * 2);
#pragma unroll
            for (int e = 0; e < 8; ++e) { d[e] = v[j][e]; x[e] = (float)xr[e]; }
        } else {
            row(r, d, x);
        }
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = k0[e] * d[e] + k1[e] * x[e] + k2[e];
        store8(p.dr + ((size_t)r * p.ld2 + f0) * 2, o);
    }
}

int check_common(const float* slab, int splitk, int rows, int n, int dtype) {
    PFA_REQUIRE(slab && splitk >= 1 && rows >= 1 && n >= 16, "pfa_finish_bn: bad arguments");
    PFA_REQUIRE(dtype == PFA_BF16, "pfa_finish_bn: built for bf16 tensors");
    PFA_REQUIRE(pfa_finish_bn_supported(dtype, rows, n), "pfa_finish_bn: needs n %% 16 == 0 and at most %d rows", FB_SLOTS * FB_MAXJ);
    return 0;
}

}  // namespace

extern "C" int pfa_finish_bn_supported(int dtype, int rows, int n) {
    return dtype == PFA_BF16 && rows >= 1 && rows <= FB_SLOTS * FB_MAXJ && n >= 16 && n % 16 == 0;
}

extern "C" int pfa_finish_bn_fwd(const float* slab, int splitk, int rows, int n, const float* gamma, const float* beta,
                                 float eps, float momentum, int repeats, float* running_mean, float* running_var,
                                 int64_t* num_batches_tracked, int dtype, void* raw, int ld, void* out_l, void* out_r,
                                 float slope, float* scale, float* shift, float* mean, float* invstd, pfa_stream s) {
    if (int rc = check_common(slab, splitk, rows, n, dtype)) return rc;
    PFA_REQUIRE(raw && scale && shift && ld >= n && ld % 8 == 0 && (running_mean == nullptr) == (running_var == nullptr),
                "pfa_finish_bn_fwd: bad arguments");
    FinBnK k{};
    k.slab = slab; k.splitk = splitk; k.rows = rows; k.n = n; k.serial = (pfa_knob(1) & 1024) != 0;
    k.gamma = gamma; k.beta = beta; k.eps = eps; k.momentum = momentum; k.repeats = repeats;
    k.running_mean = running_mean; k.running_var = running_var; k.nbt = num_batches_tracked;
    k.raw = (char*)raw; k.out_l = (char*)out_l; k.out_r = (char*)out_r; k.ld = ld; k.slope = slope;
    k.scale = scale; k.shift = shift; k.mean = mean; k.invstd = invstd;
    if (rows <= FB_SLOTS * FB_KEEP) hipLaunchKernelGGL(k_finish_bn_fwd<true>, dim3(n / 16), dim3(FB_THREADS), 0, (hipStream_t)s, k);
    else hipLaunchKernelGGL(k_finish_bn_fwd<false>, dim3(n / 16), dim3(FB_THREADS), 0, (hipStream_t)s, k);
    PFA_LAUNCHED("k_finish_bn_fwd");
    return 0;
}

extern "C" int pfa_finish_bn_bwd(const float* slab, int splitk, int rows, int n, int n1, void* plain, int ld1, const void* raw,
                                 const void* g2, int act1, int act2, float slope, const float* scale, const float* shift,
                                 const float* mean, const float* invstd, const float* gamma, float* dgamma, float* dbeta,
                                 int dtype, void* dr, int ld2, pfa_stream s) {
    if (int rc = check_common(slab, splitk, rows, n, dtype)) return rc;
    PFA_REQUIRE(n1 >= 0 && n1 < n && n1 % 16 == 0 && (n1 == 0 || (plain && ld1 >= n1)) && raw && dr && ld2 >= n - n1 && ld2 % 8 == 0 &&
                (scale == nullptr) == (shift == nullptr) && (mean == nullptr) == (invstd == nullptr),
                "pfa_finish_bn_bwd: bad arguments");
    FinBnK k{};
    k.slab = slab; k.splitk = splitk; k.rows = rows; k.n = n; k.serial = (pfa_knob(1) & 1024) != 0;
    k.n1 = n1; k.plain = (char*)plain; k.ld1 = ld1; k.fraw = (const char*)raw; k.g2 = (const char*)g2;
    k.act1 = act1; k.act2 = act2; k.slope = slope;
    k.scale = (float*)scale; k.shift = (float*)shift; k.mean = (float*)mean; k.invstd = (float*)invstd;
    k.gamma = gamma; k.dgamma = dgamma; k.dbeta = dbeta; k.dr = (char*)dr; k.ld2 = ld2;
    if (rows <= FB_SLOTS * FB_KEEP) hipLaunchKernelGGL(k_finish_bn_bwd<true>, dim3(n / 16), dim3(FB_THREADS), 0, (hipStream_t)s, k);
    else hipLaunchKernelGGL(k_finish_bn_bwd<false>, dim3(n / 16), dim3(FB_THREADS), 0, (hipStream_t)s, k);
    PFA_LAUNCHED("k_finish_bn_bwd");
    return 0;
}
