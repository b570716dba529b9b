// Finish of a K-sliced convolution on a MID-SIZE map (up to 4096 output pixels: the 8x8 and 16x16 levels of the generator at
// B = 16) in ONE launch on the whole chip: slice combine + training-mode BatchNorm (forward: statistics, running buffers,
// normalise, both activations; backward: activation' and the BatchNorm backward of the node the gradient belongs to).
//
// These layers used to take two dependent launches on the step's critical chain - k_splitk_finish_vec (slices -> conv output
// + per-block partial statistics) and k_bn_act_stats / k_bn_bwd_apply_stats (coefficients from the partials, then one more
// pass over the tensor) - 16-22 us per pair of which ~5 us is the second launch's fixed cost and the rest its re-read of what
// the first had in registers.  k_finish_bn_* (finish_bn.hip) does the same in one launch without any hand-off, but on n / 16
// workgroups: 32 CUs cannot pull a 4-17 MB slab, it wins up to 256 rows only.  Here a workgroup owns 64 channels x 32-128
// rows (up to 256 workgroups), keeps its rows' slice sums in registers, and the workgroups of one 64-channel column group
// meet ONCE inside the launch:
//   publish every workgroup stores its 2 x 64 partial sums as 8-byte {epoch, value} granules, write-through (sc1);
//   sweep   128 lanes re-read the granules of their channel from all row blocks (sc1 loads: they bypass the CU's L1, no
//           fence anywhere) until every tag is the launch's epoch - BOUNDED: a launch whose workgroups are not all resident
//           within ~1 s gives up, raises the error word and finishes with garbage instead of hanging the GPU - and add them
//           in the fixed order of the row blocks: every workgroup of a group forms the same bits;
//   leave   every workgroup adds 1 to a counter; the last one zeroes it and stores the epoch word the next launch starts from.
// (MI355X_MICROARCH.md, visibility, form R2.)
// The hand-off costs ~3 us against the ~5 us floor of a dependent launch plus the second pass's traffic.  Residency: 256-thread
// workgroups with 3 KB of LDS; only the step's main stream launches them one at a time, so every waiting workgroup's missing
// partners are queued behind kernels that end on their own.
//
// Numerics: those of k_finish_bn_fwd / k_finish_bn_bwd (float32 slice sums in ascending slice order, statistics of the float32
// sums, the bf16-ROUNDED conv output is what is normalised, coefficients in double), with the statistics summed per row block
// and then over the row blocks in ascending order.
#include "pfa_common.h"

namespace {

constexpr int FC_CG = 64;              // channels of a column group
constexpr int FC_SLOTS = 32;           // row slots per pass: 256 threads = 32 slots x 8 chunks of 8 channels
constexpr int FC_MAXP = 4;             // row passes a thread keeps in registers
constexpr int FC_MAXGX = 32;           // row blocks of a column group (the partials every workgroup reads)
constexpr unsigned FC_SPIN_MAX = 1u << 19;

typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;

struct FinCoopK {
    const float* slab;
    int splitk, rows, n, passes, gx, nmeet;      // nmeet: workgroups that take part in a meeting (the launch's last one bumps the epoch)
    float* part;                          // granules: uint64 [gx][2][n]
    unsigned* bar;                        // word 0: error flag, 1: epoch, 2: leave counter
    // ---- forward
    const float *gamma, *beta;
    float eps, momentum;
    int repeats;
    float *running_mean, *running_var;
    int64_t* nbt;
    char *raw, *out_l, *out_r;            // bf16 [rows][ld]
    int ld;
    float slope;
    float *scale, *shift, *mean, *invstd; // outputs (forward) / inputs (backward)
    // ---- backward
    int n1;                               // columns [0, n1) are a plain sum -> plain; [n1, n) the fused ones -> dr
    char* plain; int ld1;
    const char *fraw, *g2;
    int act1, act2;
    float *dgamma, *dbeta;
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

// t[ps][e] = sum over the slices of the rows of the thread's passes (ascending slice order), all passes' loads of two slices in
// flight together
template <int P>
__device__ __forceinline__ void sum_slices(const FinCoopK& p, const int* r, const bool* live, int c0, float (*t)[8]) {
    const size_t plane = (size_t)p.rows * p.n;
#pragma unroll
    for (int ps = 0; ps < P; ++ps)
#pragma unroll
        for (int e = 0; e < 8; ++e) t[ps][e] = 0.f;
    int s = 0;
    for (; s + 2 <= p.splitk; s += 2) {
        float u[2][P][8];
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int ps = 0; ps < P; ++ps)
                if (live[ps]) load8(p.slab + (size_t)(s + k) * plane + (size_t)r[ps] * p.n + c0, u[k][ps]);
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int ps = 0; ps < P; ++ps)
                if (live[ps])
#pragma unroll
                    for (int e = 0; e < 8; ++e) t[ps][e] += u[k][ps][e];
    }
    for (; s < p.splitk; ++s) {
#pragma unroll
        for (int ps = 0; ps < P; ++ps)
            if (live[ps]) {
                float u[8];
                load8(p.slab + (size_t)s * plane + (size_t)r[ps] * p.n + c0, u);
#pragma unroll
                for (int e = 0; e < 8; ++e) t[ps][e] += u[e];
            }
    }
}

// The meeting of a column group's workgroups.  a, b: this thread's two sums per channel (8 channels from chunk * 8) over its
// rows.  -> tot[stat][64]: the sums over ALL rows of the group's 64 channels (double), in LDS, behind a barrier.
// stat0: first channel of the group inside the granule rows (row length pn).
// Granules (MI355X_MICROARCH.md, visibility, form R2: the data is the flag): every partial sum travels as ONE aligned 8-byte
// write-through store {tag = this launch's epoch, value}; a reader re-reads its granules until every tag is the epoch.  No
// counter, no drain and no fence lie between a workgroup's last row and its partners' sums: one store's way to memory and
// one load's way back.  The epoch is a word of the workspace (read at the start, bumped by the launch's last workgroup to
// leave: replayed launch lists carry no per-launch argument); stale granules of earlier launches carry earlier epochs.
__device__ __forceinline__ void group_totals(const FinCoopK& p, float (&a)[8], float (&b)[8], int bx, int stat0, int pn,
                                             unsigned epoch, double (*tot)[FC_CG], float (*wpart)[2][FC_CG]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ---- the workgroup's own partial: lanes of a wave that share the chunk (lane & 7) first, then the four waves
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int off = 8; off < 64; off <<= 1) {
            a[e] += __shfl_xor(a[e], off);
            b[e] += __shfl_xor(b[e], off);
        }
    }
    if (lane < 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { wpart[wave][0][lane * 8 + e] = a[e]; wpart[wave][1][lane * 8 + e] = b[e]; }
    }
    __syncthreads();
    gu64* gran = (gu64*)p.part;
    gu32* ctl = (gu32*)p.bar;               // [0] error flag, [1] epoch of the last finished launch, [2] workgroups that have left
    if (tid < 2 * FC_CG) {
        const int stat = tid >> 6, ch = tid & 63;
        const float mine = (wpart[0][stat][ch] + wpart[1][stat][ch]) + (wpart[2][stat][ch] + wpart[3][stat][ch]);
        gu64* row0 = gran + (size_t)stat * pn + stat0 + ch;
        __hip_atomic_store(row0 + (size_t)bx * 2 * pn, ((unsigned long long)epoch << 32) | __float_as_uint(mine), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        // ---- every partial of this channel: all granules in flight, re-read until complete; summed in ascending row-block order
        float v[FC_MAXGX];
        unsigned spins = 0;
        for (;;) {
            unsigned long long g[FC_MAXGX];
#pragma unroll
            for (int k = 0; k < FC_MAXGX; ++k)
                g[k] = k < p.gx ? __hip_atomic_load(row0 + (size_t)k * 2 * pn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                : ((unsigned long long)epoch << 32);
            bool ok = true;
#pragma unroll
            for (int k = 0; k < FC_MAXGX; ++k) { ok = ok && (unsigned)(g[k] >> 32) == epoch; v[k] = __uint_as_float((unsigned)g[k]); }
            if (ok) break;
            if (++spins > FC_SPIN_MAX) {           // not all partners became resident: give up loudly instead of hanging
                __hip_atomic_store(ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < FC_MAXGX; ++k) s += (double)v[k];
        tot[stat][ch] = s;
    }
    __syncthreads();
    // ---- leave: the launch's last workgroup publishes the epoch it ran under (every partner has read its granules by then)
    if (tid == 0) {
        const unsigned left = __hip_atomic_fetch_add(ctl + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (left == (unsigned)p.nmeet - 1) {
            __hip_atomic_store(ctl + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ctl + 1, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------ forward
template <int P>
__global__ __launch_bounds__(256) void k_finish_coop_fwd(const FinCoopK p) {
    __shared__ float wpart[4][2][FC_CG];
    __shared__ double tot[2][FC_CG];
    __shared__ float coef[2][FC_CG];
    const int tid = threadIdx.x, chunk = tid & 7, slot = tid >> 3;
    const int bx = blockIdx.x, gy = blockIdx.y;
    const int cb = gy * FC_CG, c0 = cb + chunk * 8;
    if (bx == 0 && gy == 0 && tid == 0 && p.nbt) *p.nbt += p.repeats;
    const unsigned epoch = __hip_atomic_load((gu32*)p.bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    float g32 = 1.f, b32 = 0.f, rm32 = 0.f, rv32 = 0.f;       // fetched ahead of the meeting
    if (tid < FC_CG) {
        if (p.gamma) g32 = p.gamma[cb + tid];
        if (p.beta) b32 = p.beta[cb + tid];
        if (bx == 0 && p.running_mean) { rm32 = p.running_mean[cb + tid]; rv32 = p.running_var[cb + tid]; }
    }
    int r[P];
    bool live[P];
#pragma unroll
    for (int ps = 0; ps < P; ++ps) {
        r[ps] = (bx * P + ps) * FC_SLOTS + slot;
        live[ps] = r[ps] < p.rows;
    }
    float v[P][8];
    sum_slices<P>(p, r, live, c0, v);
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
#pragma unroll
    for (int ps = 0; ps < P; ++ps)
        if (live[ps])
#pragma unroll
            for (int e = 0; e < 8; ++e) { s1[e] += v[ps][e]; s2[e] += v[ps][e] * v[ps][e]; }
    group_totals(p, s1, s2, bx, cb, p.n, epoch, tot, wpart);
    if (tid < FC_CG) {
        const int ch = cb + tid;
        const double count = (double)p.rows;
        const double mean = tot[0][tid] / count;
        double var = tot[1][tid] / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const double invstd = 1.0 / sqrt(var + (double)p.eps);
        const double g = g32, b = b32;
        const float sc = (float)(g * invstd), sh = (float)(b - mean * g * invstd);
        coef[0][tid] = sc; coef[1][tid] = sh;
        if (bx == 0) {                     // one workgroup of the group publishes (all of them hold the same bits)
            p.scale[ch] = sc; p.shift[ch] = sh;
            if (p.mean) p.mean[ch] = (float)mean;
            if (p.invstd) p.invstd[ch] = (float)invstd;
            if (p.running_mean && p.repeats > 0) {
                const double unb = count > 1.0 ? var * (count / (count - 1.0)) : var;
                double rm = rm32, rv = rv32;
                for (int k = 0; k < p.repeats; ++k) {
                    rm = (1.0 - p.momentum) * rm + p.momentum * mean;
                    rv = (1.0 - p.momentum) * rv + p.momentum * unb;
                }
                p.running_mean[ch] = (float)rm;
                p.running_var[ch] = (float)rv;
            }
        }
    }
    __syncthreads();
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = coef[0][chunk * 8 + e]; sh[e] = coef[1][chunk * 8 + e]; }
#pragma unroll
    for (int ps = 0; ps < P; ++ps) {
        if (!live[ps]) continue;
        const size_t off = ((size_t)r[ps] * p.ld + c0) * 2;
        float x[8], yl[8], yr[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            x[e] = (float)(bf16_t)v[ps][e];                  // the stored conv output is what is normalised (as k_bn_act reads it)
            const float y = x[e] * sc[e] + sh[e];
            yl[e] = y >= 0.f ? y : y * p.slope;
            yr[e] = y <= 0.f ? 0.f : y;                      // (ReLU keeps a NaN, like torch.relu)
        }
        store8(p.raw + off, x);
        if (p.out_l) store8(p.out_l + off, yl);
        if (p.out_r) store8(p.out_r + off, yr);
    }
}

// ------------------------------------------------------------------ backward
template <int P>
__global__ __launch_bounds__(256) void k_finish_coop_bwd(const FinCoopK p) {
    __shared__ float wpart[4][2][FC_CG];
    __shared__ double tot[2][FC_CG];
    __shared__ float coef[3][FC_CG];
    const int tid = threadIdx.x, chunk = tid & 7, slot = tid >> 3;
    const int bx = blockIdx.x, gy = blockIdx.y;
    const int cb = gy * FC_CG, c0 = cb + chunk * 8;
    int r[P];
    bool live[P];
#pragma unroll
    for (int ps = 0; ps < P; ++ps) {
        r[ps] = (bx * P + ps) * FC_SLOTS + slot;
        live[ps] = r[ps] < p.rows;
    }
    float v[P][8];
    if (cb < p.n1) {                                         // a plain output: the sum itself (n1 is a multiple of 64); no meeting
        sum_slices<P>(p, r, live, c0, v);
#pragma unroll
        for (int ps = 0; ps < P; ++ps)
            if (live[ps]) store8(p.plain + ((size_t)r[ps] * p.ld1 + c0) * 2, v[ps]);
        return;
    }
    const int f0 = c0 - p.n1, fb = cb - p.n1;                // column inside the fused output
    const unsigned epoch = __hip_atomic_load((gu32*)p.bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    float g32 = 1.f, is32 = 1.f, mu32 = 0.f, dg32 = 0.f, db32 = 0.f;
    if (tid < FC_CG) {
        if (p.gamma) g32 = p.gamma[fb + tid];
        is32 = p.invstd[fb + tid];
        mu32 = p.mean[fb + tid];
        if (bx == 0 && p.dgamma) dg32 = p.dgamma[fb + tid];
        if (bx == 0 && p.dbeta) db32 = p.dbeta[fb + tid];
    }
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
    if (p.scale) { load8(p.scale + f0, sc); load8(p.shift + f0, sh); }
    const float neg1 = act_neg(p.act1, p.slope), neg2 = act_neg(p.act2, p.slope);
    bf16x8 xr[P], gr[P];
#pragma unroll
    for (int ps = 0; ps < P; ++ps) {                          // (in flight under the slice sums)
        xr[ps] = bf16x8{}; gr[ps] = bf16x8{};
        if (live[ps]) {
            const size_t off = ((size_t)r[ps] * p.ld2 + f0) * 2;
            xr[ps] = *(const bf16x8*)(p.fraw + off);
            if (p.g2) gr[ps] = *(const bf16x8*)(p.g2 + off);
        }
    }
    sum_slices<P>(p, r, live, c0, v);
    float s1[8], s2[8];                                      // sum d, sum d x  (sum d xhat = invstd (sum d x - mean sum d))
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
#pragma unroll
    for (int ps = 0; ps < P; ++ps) {
        if (!live[ps]) continue;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = (float)xr[ps][e];
            const float y = x * sc[e] + sh[e];
            float d = act_grad_lin(y, neg1) * v[ps][e];
            if (p.g2) d += act_grad_lin(y, neg2) * (float)gr[ps][e];
            v[ps][e] = d;
            s1[e] += d;
            s2[e] += d * x;
        }
    }
    group_totals(p, s1, s2, bx, fb, p.n - p.n1, epoch, tot, wpart);
    if (tid < FC_CG) {
        const int ch = fb + tid;
        const double sdy = tot[0][tid], count = (double)p.rows;
        const double sdyx = (double)is32 * (tot[1][tid] - (double)mu32 * sdy);      // sum d xhat
        if (bx == 0) {
            if (p.dgamma) p.dgamma[ch] = dg32 + (float)sdyx;     // gradients accumulate into a zeroed arena
            if (p.dbeta) p.dbeta[ch] = db32 + (float)sdy;
        }
        const double s = (double)g32 * is32;
        coef[0][tid] = (float)s;
        coef[1][tid] = (float)(-s * is32 * sdyx / count);
        coef[2][tid] = (float)(-s * sdy / count + s * is32 * mu32 * sdyx / count);
    }
    __syncthreads();
    float k0[8], k1[8], k2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { k0[e] = coef[0][chunk * 8 + e]; k1[e] = coef[1][chunk * 8 + e]; k2[e] = coef[2][chunk * 8 + e]; }
#pragma unroll
    for (int ps = 0; ps < P; ++ps) {
        if (!live[ps]) continue;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = k0[e] * v[ps][e] + k1[e] * (float)xr[ps][e] + k2[e];
        store8(p.dr + ((size_t)r[ps] * p.ld2 + f0) * 2, o);
    }
}

// row passes per workgroup: the fewest (most workgroups) that keep a column group within FC_MAXGX row blocks
int plan_passes(int rows) {
    for (int ps = 1; ps <= FC_MAXP; ps *= 2)
        if ((rows + FC_SLOTS * ps - 1) / (FC_SLOTS * ps) <= FC_MAXGX) return ps;
    return 0;
}

int check_common(const float* slab, int splitk, int rows, int n, int dtype, float* part, unsigned* bar) {
    PFA_REQUIRE(slab && splitk >= 1 && part && bar, "pfa_finish_coop: bad arguments");
    PFA_REQUIRE(pfa_finish_coop_supported(dtype, rows, n), "pfa_finish_coop: needs bf16, n %% 64 == 0 and at most %d rows",
                FC_SLOTS * FC_MAXP * FC_MAXGX);
    PFA_REQUIRE((((uintptr_t)part) & 7u) == 0, "pfa_finish_coop: the granule workspace must be 8-byte aligned");
    return 0;
}

}  // namespace

extern "C" int pfa_finish_coop_supported(int dtype, int rows, int n) {
    return dtype == PFA_BF16 && rows >= 1 && rows <= FC_SLOTS * FC_MAXP * FC_MAXGX && n >= FC_CG && n % FC_CG == 0;
}

extern "C" size_t pfa_finish_coop_part_floats(int rows, int n) {
    const int ps = plan_passes(rows);
    if (ps == 0 || n <= 0) return 0;
    return (size_t)((rows + FC_SLOTS * ps - 1) / (FC_SLOTS * ps)) * 2 * n * 2;      // 8-byte granules
}

extern "C" size_t pfa_finish_coop_bar_words(int n) { return n > 0 ? 32 : 0; }

extern "C" int pfa_finish_coop_fwd(const float* slab, int splitk, int rows, int n, const float* gamma, const float* beta,
                                   float eps, float momentum, int repeats, float* running_mean, float* running_var,
                                   int64_t* num_batches_tracked, int dtype, void* raw, int ld, void* out_l, void* out_r,
                                   float slope, float* scale, float* shift, float* mean, float* invstd, float* part,
                                   uint32_t* bar, pfa_stream s) {
    if (int rc = check_common(slab, splitk, rows, n, dtype, part, bar)) return rc;
    PFA_REQUIRE(raw && scale && shift && ld >= n && ld % 8 == 0 && (running_mean == nullptr) == (running_var == nullptr),
                "pfa_finish_coop_fwd: bad arguments");
    FinCoopK k{};
    k.slab = slab; k.splitk = splitk; k.rows = rows; k.n = n;
    k.passes = plan_passes(rows); k.gx = (rows + FC_SLOTS * k.passes - 1) / (FC_SLOTS * k.passes);
    k.part = part; k.bar = bar; k.nmeet = k.gx * (n / FC_CG);
    k.gamma = gamma; k.beta = beta; k.eps = eps; k.momentum = momentum; k.repeats = repeats;
    k.running_mean = running_mean; k.running_var = running_var; k.nbt = num_batches_tracked;
    k.raw = (char*)raw; k.out_l = (char*)out_l; k.out_r = (char*)out_r; k.ld = ld; k.slope = slope;
    k.scale = scale; k.shift = shift; k.mean = mean; k.invstd = invstd;
    const dim3 grid(k.gx, n / FC_CG);
    hipStream_t st = (hipStream_t)s;
    if (k.passes == 1) hipLaunchKernelGGL(k_finish_coop_fwd<1>, grid, dim3(256), 0, st, k);
    else if (k.passes == 2) hipLaunchKernelGGL(k_finish_coop_fwd<2>, grid, dim3(256), 0, st, k);
    else hipLaunchKernelGGL(k_finish_coop_fwd<4>, grid, dim3(256), 0, st, k);
    PFA_LAUNCHED("k_finish_coop_fwd");
    return 0;
}

extern "C" int pfa_finish_coop_bwd(const float* slab, int splitk, int rows, int n, int n1, void* plain, int ld1, const void* raw,
                                   const void* g2, int act1, int act2, float slope, const float* scale, const float* shift,
                                   const float* mean, const float* invstd, const float* gamma, float* dgamma, float* dbeta,
                                   int dtype, void* dr, int ld2, float* part, uint32_t* bar, pfa_stream s) {
    if (int rc = check_common(slab, splitk, rows, n, dtype, part, bar)) return rc;
    PFA_REQUIRE(n1 >= 0 && n1 < n && n1 % FC_CG == 0 && (n1 == 0 || (plain && ld1 >= n1)) && raw && dr && ld2 >= n - n1 &&
                ld2 % 8 == 0 && (scale == nullptr) == (shift == nullptr) && mean && invstd,
                "pfa_finish_coop_bwd: bad arguments (the fused node must carry a BatchNorm; n1 %% 64 == 0)");
    FinCoopK k{};
    k.slab = slab; k.splitk = splitk; k.rows = rows; k.n = n;
    k.passes = plan_passes(rows); k.gx = (rows + FC_SLOTS * k.passes - 1) / (FC_SLOTS * k.passes);
    k.part = part; k.bar = bar; k.nmeet = k.gx * ((n - n1) / FC_CG);
    k.n1 = n1; k.plain = (char*)plain; k.ld1 = ld1; k.fraw = (const char*)raw; k.g2 = (const char*)g2;
    k.act1 = act1; k.act2 = act2; k.slope = slope;
    k.scale = (float*)scale; k.shift = (float*)shift; k.mean = (float*)mean; k.invstd = (float*)invstd;
    k.gamma = gamma; k.dgamma = dgamma; k.dbeta = dbeta; k.dr = (char*)dr; k.ld2 = ld2;
    const dim3 grid(k.gx, n / FC_CG);
    hipStream_t st = (hipStream_t)s;
    if (k.passes == 1) hipLaunchKernelGGL(k_finish_coop_bwd<1>, grid, dim3(256), 0, st, k);
    else if (k.passes == 2) hipLaunchKernelGGL(k_finish_coop_bwd<2>, grid, dim3(256), 0, st, k);
    else hipLaunchKernelGGL(k_finish_coop_bwd<4>, grid, dim3(256), 0, st, k);
    PFA_LAUNCHED("k_finish_coop_bwd");
    return 0;
}
