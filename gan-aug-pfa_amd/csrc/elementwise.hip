// HBM-bound kernels of the GAN step: BatchNorm statistics / apply / backward, activations, losses,
// Adam, weight shadows and layout conversion.  All tensors are NHWC viewed as [rows][c], moved in
// 16-byte chunks per lane (coalesced: consecutive lanes touch consecutive chunks).
//
// Reference semantics restated here (file:line in /root/reference):
//   nn.BatchNorm2d training/eval        models.py:179,181,231,239 (eps 1e-5, momentum 0.1)
//   nn.LeakyReLU(0.2) / nn.ReLU / Tanh  models.py:178,180,186,223,232,240
//   BCEWithLogitsLoss / L1Loss          train_gan.py:42-43,58-61,67-69
//   optim.Adam(lr, betas=(0.5,0.999))   train_gan.py:140-141
#include "pfa_common.h"

thread_local char g_pfa_err[512] = "";
std::atomic<uint64_t> g_pfa_launches{0};

extern "C" int pfa_version(void) { return 100; }
extern "C" const char* pfa_last_error_string(void) { return g_pfa_err; }
extern "C" uint64_t pfa_launch_count(void) { return g_pfa_launches.load(); }

namespace {

inline int grid_for(int64_t work, int block = 256, int cap = 256 * 16) {
    int64_t g = (work + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// grid for the per-channel elementwise kernels: total threads a multiple of cpr (chunks per row)
inline int grid_chan(int64_t nchunks, int cpr, int cap = 256 * 8) {
    int g = grid_for(nchunks, 256, cap);
    if (cpr > 256) {                      // more than 256 chunks per row: make blocks * 256 a multiple of cpr
        const int per = cpr / 256;
        g = (g + per - 1) / per * per;
    }
    return g;
}

// ------------------------------------------------------------------ BatchNorm statistics
// Block = 8 channels x 32 tile-lanes: the per-tile partials are summed in parallel (double), then one lane
// per channel finishes.  Deterministic: fixed tile -> lane assignment and a fixed-order LDS tree.
constexpr int FIN_CH = 8, FIN_LANES = 32, FIN_UNROLL = 8;
__device__ __forceinline__ void tile_sum2(const float* __restrict__ a, const float* __restrict__ b, int ntiles, int c,
                                          int ch, bool valid, double (*sh)[FIN_LANES][FIN_CH], double& ra, double& rb) {
    const int cl = threadIdx.x % FIN_CH, tl = threadIdx.x / FIN_CH;
    // FIN_UNROLL independent loads per array in flight (the partials were written by other XCDs a moment ago: every load
    // is a trip past the L2; one dependent load per iteration made a 1024-tile finalize take 20 us) and as many partial
    // sums, combined in a fixed order
    double s[FIN_UNROLL], q[FIN_UNROLL];
#pragma unroll
    for (int u = 0; u < FIN_UNROLL; ++u) s[u] = q[u] = 0.0;
    if (valid) {
        int t = tl;
        for (; t + (FIN_UNROLL - 1) * FIN_LANES < ntiles; t += FIN_UNROLL * FIN_LANES) {
            float av[FIN_UNROLL], bv[FIN_UNROLL];
#pragma unroll
            for (int u = 0; u < FIN_UNROLL; ++u) {
                av[u] = a[(size_t)(t + u * FIN_LANES) * c + ch];
                bv[u] = b[(size_t)(t + u * FIN_LANES) * c + ch];
            }
#pragma unroll
            for (int u = 0; u < FIN_UNROLL; ++u) { s[u] += (double)av[u]; q[u] += (double)bv[u]; }
        }
#pragma unroll
        for (int u = 0; u < FIN_UNROLL - 1; ++u)
            if (t + u * FIN_LANES < ntiles) {
                s[u] += (double)a[(size_t)(t + u * FIN_LANES) * c + ch];
                q[u] += (double)b[(size_t)(t + u * FIN_LANES) * c + ch];
            }
    }
    sh[0][tl][cl] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    sh[1][tl][cl] = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
    __syncthreads();
    for (int off = FIN_LANES / 2; off > 0; off >>= 1) {
        if (tl < off) {
            sh[0][tl][cl] += sh[0][tl + off][cl];
            sh[1][tl][cl] += sh[1][tl + off][cl];
        }
        __syncthreads();
    }
    ra = sh[0][0][cl];
    rb = sh[1][0][cl];
}

__global__ __launch_bounds__(256) void k_bn_finalize(const float* __restrict__ psum, const float* __restrict__ psq,
                                                     int ntiles, int c, double count, const float* gamma,
                                                     const float* beta, float eps, float momentum, int repeats,
                                                     float* running_mean, float* running_var, int64_t* nbt,
                                                     float* scale, float* shift, float* mean_out, float* invstd_out) {
    __shared__ double sh[2][FIN_LANES][FIN_CH];
    const int ch = blockIdx.x * FIN_CH + threadIdx.x % FIN_CH;
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += repeats;
    // the finishing lanes fetch their coefficients BEFORE the reduction: issued after it, each of these loads is a
    // serial ~1 us round trip on the critical path of the forward pass
    const bool fin = threadIdx.x < FIN_CH && ch < c;
    float g32 = 1.f, b32 = 0.f, rm32 = 0.f, rv32 = 0.f;
    if (fin) {
        if (gamma) g32 = gamma[ch];
        if (beta) b32 = beta[ch];
        if (running_mean) { rm32 = running_mean[ch]; rv32 = running_var[ch]; }
    }
    double s, q;
    tile_sum2(psum, psq, ntiles, c, ch, ch < c, sh, s, q);
    if (!fin) return;
    const double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const double g = g32, b = b32;
    scale[ch] = (float)(g * invstd);
    shift[ch] = (float)(b - mean * g * invstd);
    if (mean_out) mean_out[ch] = (float)mean;
    if (invstd_out) invstd_out[ch] = (float)invstd;
    if (running_mean && repeats > 0) {
        const double unb = count > 1.0 ? var * (count / (count - 1.0)) : var;
        double rm = rm32, rv = rv32;
        for (int r = 0; r < repeats; ++r) {
            rm = (1.0 - momentum) * rm + momentum * mean;
            rv = (1.0 - momentum) * rv + momentum * unb;
        }
        running_mean[ch] = (float)rm;
        running_var[ch] = (float)rv;
    }
}

__global__ void k_bn_eval_coeffs(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                 int c, float* scale, float* shift) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    const double invstd = 1.0 / sqrt((double)rv[ch] + (double)eps);
    const double g = gamma ? (double)gamma[ch] : 1.0, b = beta ? (double)beta[ch] : 0.0;
    scale[ch] = (float)(g * invstd);
    shift[ch] = (float)(b - (double)rm[ch] * g * invstd);
}

// ------------------------------------------------------------------ normalise + dual activation
// Per-lane channel coefficients: CH consecutive floats starting at c0 (16-byte aligned), or a constant when the array
// is absent.  ONE uniform branch and vector loads: written element by element (`p ? p[c0+e] : k`) the compiler emits a
// separately branched, serialised load per element, ~1 us each - 15 us of fixed latency per launch.
template <int CH> __device__ __forceinline__ void load_coef(const float* __restrict__ p, int c0, float dflt, float* out) {
    if (p) {
#pragma unroll
        for (int e = 0; e < CH; e += 4) {
            const f32x4 t = *(const f32x4*)(p + c0 + e);
            out[e] = t[0]; out[e + 1] = t[1]; out[e + 2] = t[2]; out[e + 3] = t[3];
        }
    } else {
#pragma unroll
        for (int e = 0; e < CH; ++e) out[e] = dflt;
    }
}

// The grid is sized so that (threads in the grid) % (chunks per row) == 0: a lane then always lands on the same
// channel chunk and keeps its scale/shift in registers for the whole grid-stride loop.
template <typename T>
__global__ __launch_bounds__(256) void k_bn_act(const T* __restrict__ raw, int64_t nchunks, int cpr,
                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                float slope, T* out_l, T* out_r) {
    constexpr int CH = ElemTraits<T>::CH;
    const int64_t first = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c0 = (int)(first % cpr) * CH;
    float sc[CH], sh[CH];
    load_coef<CH>(scale, c0, 1.f, sc);
    load_coef<CH>(scale ? shift : nullptr, c0, 0.f, sh);
    for (int64_t idx = first; idx < nchunks; idx += (int64_t)gridDim.x * 256) {
        Chunk<T> x, l, r;
        x.load(raw + idx * CH);
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            const float y = x.v[e] * sc[e] + sh[e];
            l.v[e] = y >= 0.f ? y : y * slope;
            r.v[e] = y <= 0.f ? 0.f : y;                 // (ReLU keeps a NaN, like torch.relu: the compare is false for it)
        }
        if (out_l) l.store(out_l + idx * CH);
        if (out_r) r.store(out_r + idx * CH);
    }
}

// ------------------------------------------------------------------ BatchNorm backward, pass 1
// dy = act1'(y) g1 + act2'(y) g2 ; partial sums of dy and dy*xhat per channel and block.
constexpr int BWD_PASSES = 64;
template <typename T>
__global__ __launch_bounds__(256) void k_bn_bwd_reduce(const T* __restrict__ raw, int64_t rows, int c, int cpr,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const T* __restrict__ g1,
                                                       int act1, const T* __restrict__ g2, int act2, float slope,
                                                       T* dy, float* part, int ntiles, int passes) {
    constexpr int CH = ElemTraits<T>::CH;
    __shared__ float red[2][256][CH];
    const int tid = threadIdx.x;
    const int col = tid % cpr, rsub = tid / cpr, rpp = 256 / cpr;
    const int c0 = col * CH;
    float sc[CH], sh[CH], mu[CH], is[CH], s_dy[CH], s_dyx[CH];
    load_coef<CH>(scale, c0, 1.f, sc);
    load_coef<CH>(scale ? shift : nullptr, c0, 0.f, sh);
    load_coef<CH>(mean, c0, 0.f, mu);
    load_coef<CH>(mean ? invstd : nullptr, c0, 1.f, is);
#pragma unroll
    for (int e = 0; e < CH; ++e) {
        s_dy[e] = 0.f;
        s_dyx[e] = 0.f;
    }
    const int64_t row0 = (int64_t)blockIdx.x * rpp * passes;
    for (int ps = 0; ps < passes; ++ps) {
        const int64_t row = row0 + (int64_t)ps * rpp + rsub;
        if (row >= rows) break;
        const int64_t off = (row * cpr + col) * CH;
        Chunk<T> x, a, b, o;
        x.load(raw + off);
        a.load(g1 + off);
        if (g2) b.load(g2 + off);
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            const float y = x.v[e] * sc[e] + sh[e];
            float d = act_grad(y, act1, slope) * a.v[e];
            if (g2) d += act_grad(y, act2, slope) * b.v[e];
            o.v[e] = d;
            s_dy[e] += d;
            s_dyx[e] += d * ((x.v[e] - mu[e]) * is[e]);
        }
        o.store(dy + off);
    }
    if (!part) return;
#pragma unroll
    for (int e = 0; e < CH; ++e) { red[0][tid][e] = s_dy[e]; red[1][tid][e] = s_dyx[e]; }
    __syncthreads();
    for (int off = rpp >> 1; off > 0; off >>= 1) {       // fixed-shape tree over the row groups (rpp is a power of two)
        if (rsub < off) {
#pragma unroll
            for (int e = 0; e < CH; ++e) {
                red[0][tid][e] += red[0][tid + off * cpr][e];
                red[1][tid][e] += red[1][tid + off * cpr][e];
            }
        }
        __syncthreads();
    }
    if (tid < cpr) {
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            part[(size_t)blockIdx.x * c + tid * CH + e] = red[0][tid][e];
            part[((size_t)ntiles + blockIdx.x) * c + tid * CH + e] = red[1][tid][e];
        }
    }
}

__global__ __launch_bounds__(256) void k_bn_bwd_finalize(const float* __restrict__ part, int ntiles, int c,
                                                         double count, const float* gamma, const float* mean,
                                                         const float* invstd, float* dgamma, float* dbeta,
                                                         float* coef) {
    __shared__ double sh[2][FIN_LANES][FIN_CH];
    const int ch = blockIdx.x * FIN_CH + threadIdx.x % FIN_CH;
    const bool fin = threadIdx.x < FIN_CH && ch < c;
    float g32 = 1.f, is32 = 0.f, mu32 = 0.f, dg32 = 0.f, db32 = 0.f;      // fetched ahead of the reduction (see k_bn_finalize)
    if (fin) {
        if (gamma) g32 = gamma[ch];
        is32 = invstd[ch];
        mu32 = mean[ch];
        if (dgamma) dg32 = dgamma[ch];
        if (dbeta) db32 = dbeta[ch];
    }
    double sdy, sdyx;
    tile_sum2(part, part + (size_t)ntiles * c, ntiles, c, ch, ch < c, sh, sdy, sdyx);
    if (!fin) return;
    if (dgamma) dgamma[ch] = dg32 + (float)sdyx;       // gradients accumulate into a zeroed arena (zero_grad semantics)
    if (dbeta) dbeta[ch] = db32 + (float)sdy;
    const double g = g32;
    const double is = is32, mu = mu32;
    const double s = g * is;
    // dr = s*(dy - sdy/cnt - xhat*sdyx/cnt),  xhat = (raw-mu)*is
    coef[ch] = (float)s;
    coef[c + ch] = (float)(-s * is * sdyx / count);
    coef[2 * c + ch] = (float)(-s * sdy / count + s * is * mu * sdyx / count);
}

template <typename T>
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const T* dy, const T* __restrict__ raw, int64_t nchunks,
                                                      int cpr, int c, const float* __restrict__ coef, T* dr) {
    constexpr int CH = ElemTraits<T>::CH;
    const int64_t first = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c0 = (int)(first % cpr) * CH;           // fixed per lane (grid sized as for k_bn_act)
    float ka[CH], kb[CH], kc[CH];
    load_coef<CH>(coef, c0, 0.f, ka);
    load_coef<CH>(coef + c, c0, 0.f, kb);
    load_coef<CH>(coef + 2 * c, c0, 0.f, kc);
    for (int64_t idx = first; idx < nchunks; idx += (int64_t)gridDim.x * 256) {
        Chunk<T> d, x, o;
        d.load(dy + idx * CH);
        x.load(raw + idx * CH);
#pragma unroll
        for (int e = 0; e < CH; ++e) o.v[e] = ka[e] * d.v[e] + kb[e] * x.v[e] + kc[e];
        o.store(dr + idx * CH);
    }
}

// ------------------------------------------------------------------ BatchNorm: statistics finished by their consumer
// k_bn_act_stats / k_bn_bwd_apply_stats fold the coefficient launch (k_bn_finalize / k_bn_bwd_finalize: a dependent ~5 us
// launch on the critical chain of every BatchNorm layer, forward and backward) into the prologue of the pass that applies
// the coefficients.  A workgroup owns a group of CG channels (64, or all of them when the layer has fewer) x a run of rows:
// it sums the per-tile partials of ITS channels - every workgroup of a channel group in the same fixed order, hence to the
// same bits; no cross-workgroup hand-off - and the workgroups of row block 0 publish the coefficients the backward pass reads
// later (and the running statistics).  Order per channel: ST_LANES virtual lanes, lane l adds tiles l, l + ST_LANES, ... in
// ascending order (double), then a halving tree over the lanes.
constexpr int ST_LANES = 16, ST_CG = 64, ST_UNROLL = 8;

// -> sums of a[tile][ch] and b[tile][ch] over the tiles for the cg channels from ch0, left in sh[0][0][.] / sh[1][0][.]
// `prefetch` is called once, right behind the issue of the FIRST round of partial-sum loads: vector-memory results come back
// in issue order, so bulk loads issued in front of the partials would put their whole latency in front of the statistics.
template <typename F>
__device__ __forceinline__ void group_tile_sums(const float* __restrict__ a, const float* __restrict__ b, int ntiles, int c,
                                                int ch0, int cg, double (*sh)[ST_LANES][ST_CG], F prefetch) {
    const int quads = cg >> 2, tid = threadIdx.x;
    const int quad = tid % quads, tl = tid / quads;
    const bool active = tl < ST_LANES;
    bool fetched = false;
    if (active) {
        double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
        const float* pa = a + ch0 + 4 * quad;
        const float* pb = b + ch0 + 4 * quad;
        for (int t0 = tl; t0 < ntiles; t0 += ST_LANES * ST_UNROLL) {
            f32x4 av[ST_UNROLL], bv[ST_UNROLL];
#pragma unroll
            for (int u = 0; u < ST_UNROLL; ++u) {           // all loads of a round in flight: each is a trip past the L2
                const int t = t0 + u * ST_LANES;
                if (t < ntiles) {
                    av[u] = *(const f32x4*)(pa + (size_t)t * c);
                    bv[u] = *(const f32x4*)(pb + (size_t)t * c);
                } else {
                    av[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    bv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            if (!fetched) { prefetch(); fetched = true; }
#pragma unroll
            for (int u = 0; u < ST_UNROLL; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) { s[e] += (double)av[u][e]; q[e] += (double)bv[u][e]; }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { sh[0][tl][4 * quad + e] = s[e]; sh[1][tl][4 * quad + e] = q[e]; }
    }
    if (!fetched) prefetch();                               // (lanes without a share of the partial sums)
    __syncthreads();
    for (int off = ST_LANES / 2; off > 0; off >>= 1) {
        if (active && tl < off) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sh[0][tl][4 * quad + e] += sh[0][tl + off][4 * quad + e];
                sh[1][tl][4 * quad + e] += sh[1][tl + off][4 * quad + e];
            }
        }
        __syncthreads();
    }
}

// a 16-byte chunk kept packed (4 registers) while it is in flight; unpacked to floats when it is used
template <typename T> struct Raw16;
template <> struct Raw16<float> {
    f32x4 q;
    __device__ __forceinline__ void load(const float* p) { q = *(const f32x4*)p; }
    __device__ __forceinline__ void unpack(Chunk<float>& c) const { c.v[0] = q[0]; c.v[1] = q[1]; c.v[2] = q[2]; c.v[3] = q[3]; }
};
template <> struct Raw16<bf16_t> {
    bf16x8 q;
    __device__ __forceinline__ void load(const bf16_t* p) { q = *(const bf16x8*)p; }
    __device__ __forceinline__ void unpack(Chunk<bf16_t>& c) const {
#pragma unroll
        for (int i = 0; i < 8; ++i) c.v[i] = (float)q[i];
    }
};
constexpr int ST_DEPTH = 8;            // row passes a lane keeps in flight (k_bn_act_stats; half of it, two tensors, backward)

// grid (row blocks, channel groups).  Rows of a group are 16-byte chunks x cg/CH; a lane keeps one chunk column.
template <typename T>
__global__ __launch_bounds__(256) void k_bn_act_stats(const T* __restrict__ raw, int64_t rows, int c, int cg,
                                                      const float* __restrict__ psum, const float* __restrict__ psq,
                                                      int ntiles, double count, const float* gamma, const float* beta,
                                                      float eps, float momentum, int repeats, float* running_mean,
                                                      float* running_var, int64_t* nbt, float* scale, float* shift,
                                                      float* mean_out, float* invstd_out, float slope, T* out_l, T* out_r) {
    constexpr int CH = ElemTraits<T>::CH;
    __shared__ double sh[2][ST_LANES][ST_CG];
    __shared__ float cf[2][ST_CG];
    const int tid = threadIdx.x, ch0 = blockIdx.y * cg;
    const bool pub = blockIdx.x == 0;                   // this workgroup publishes its group's coefficients
    const bool fin = tid < cg;
    // the first ST_DEPTH row passes of this lane are fetched while the statistics are being finished (issued right behind the
    // first round of partial-sum loads): their latency - the tensor was written by the previous kernel: a trip to memory -
    // overlaps the reduction instead of following it
    const int cprg = cg / CH, col = tid % cprg, rsub = tid / cprg, rpp = 256 / cprg;
    const size_t coff = (size_t)ch0 + (size_t)col * CH;
    const int64_t stride = (int64_t)gridDim.x * rpp;
    int64_t row = (int64_t)blockIdx.x * rpp + rsub;
    Raw16<T> pre[ST_DEPTH];
    auto prefetch = [&]() {
#pragma unroll
        for (int k = 0; k < ST_DEPTH; ++k)
            if (row + k * stride < rows) pre[k].load(raw + (size_t)(row + k * stride) * c + coff);
    };
    if (pub && blockIdx.y == 0 && tid == 0 && nbt) *nbt += repeats;
    float g32 = 1.f, b32 = 0.f, rm32 = 0.f, rv32 = 0.f;    // fetched ahead of the reduction (see k_bn_finalize)
    if (fin) {
        if (gamma) g32 = gamma[ch0 + tid];
        if (beta) b32 = beta[ch0 + tid];
        if (pub && running_mean) { rm32 = running_mean[ch0 + tid]; rv32 = running_var[ch0 + tid]; }
    }
    group_tile_sums(psum, psq, ntiles, c, ch0, cg, sh, prefetch);
    if (fin) {
        const double s = sh[0][0][tid], q = sh[1][0][tid];
        const double mean = s / count;
        double var = q / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const double invstd = 1.0 / sqrt(var + (double)eps);
        const double g = g32, b = b32;
        const float sc = (float)(g * invstd), sf = (float)(b - mean * g * invstd);
        cf[0][tid] = sc;
        cf[1][tid] = sf;
        if (pub) {
            const int ch = ch0 + tid;
            scale[ch] = sc;
            shift[ch] = sf;
            if (mean_out) mean_out[ch] = (float)mean;
            if (invstd_out) invstd_out[ch] = (float)invstd;
            if (running_mean && repeats > 0) {
                const double unb = count > 1.0 ? var * (count / (count - 1.0)) : var;
                double rm = rm32, rv = rv32;
                for (int r = 0; r < repeats; ++r) {
                    rm = (1.0 - momentum) * rm + momentum * mean;
                    rv = (1.0 - momentum) * rv + momentum * unb;
                }
                running_mean[ch] = (float)rm;
                running_var[ch] = (float)rv;
            }
        }
    }
    __syncthreads();
    float sc[CH], sf[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) { sc[e] = cf[0][col * CH + e]; sf[e] = cf[1][col * CH + e]; }
    auto emit = [&](const Raw16<T>& xr, int64_t r) {
        Chunk<T> x, l, rr;
        xr.unpack(x);
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            const float y = x.v[e] * sc[e] + sf[e];
            l.v[e] = y >= 0.f ? y : y * slope;
            rr.v[e] = y <= 0.f ? 0.f : y;                // (ReLU keeps a NaN, like torch.relu: the compare is false for it)
        }
        if (out_l) l.store(out_l + (size_t)r * c + coff);
        if (out_r) rr.store(out_r + (size_t)r * c + coff);
    };
    // ST_DEPTH row passes in flight per lane: the grid is kept small (every workgroup re-reads the partial sums), so the
    // bandwidth has to come from the depth of each lane's queue, not from the number of lanes
    for (;;) {
#pragma unroll
        for (int k = 0; k < ST_DEPTH; ++k)
            if (row + k * stride < rows) emit(pre[k], row + k * stride);
        row += ST_DEPTH * stride;
        if (row >= rows) break;
#pragma unroll
        for (int k = 0; k < ST_DEPTH; ++k)
            if (row + k * stride < rows) pre[k].load(raw + (size_t)(row + k * stride) * c + coff);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_bn_bwd_apply_stats(const T* dy, const T* __restrict__ raw, int64_t rows, int c,
                                                            int cg, const float* __restrict__ part, int ntiles, double count,
                                                            const float* gamma, const float* mean, const float* invstd,
                                                            float* dgamma, float* dbeta, T* dr) {
    constexpr int CH = ElemTraits<T>::CH;
    __shared__ double sh[2][ST_LANES][ST_CG];
    __shared__ float cf[3][ST_CG];
    const int tid = threadIdx.x, ch0 = blockIdx.y * cg;
    const bool pub = blockIdx.x == 0, fin = tid < cg;
    const int cprg = cg / CH, col = tid % cprg, rsub = tid / cprg, rpp = 256 / cprg;
    const size_t coff = (size_t)ch0 + (size_t)col * CH;
    const int64_t stride = (int64_t)gridDim.x * rpp;
    int64_t row = (int64_t)blockIdx.x * rpp + rsub;
    constexpr int DEPTH = ST_DEPTH / 2;
    Raw16<T> pd[DEPTH], px[DEPTH];                      // first row passes fetched under the statistics (see k_bn_act_stats)
    auto prefetch = [&]() {
#pragma unroll
        for (int k = 0; k < DEPTH; ++k)
            if (row + k * stride < rows) {
                pd[k].load(dy + (size_t)(row + k * stride) * c + coff);
                px[k].load(raw + (size_t)(row + k * stride) * c + coff);
            }
    };
    float g32 = 1.f, is32 = 0.f, mu32 = 0.f, dg32 = 0.f, db32 = 0.f;
    if (fin) {
        const int ch = ch0 + tid;
        if (gamma) g32 = gamma[ch];
        is32 = invstd[ch];
        mu32 = mean[ch];
        if (pub && dgamma) dg32 = dgamma[ch];
        if (pub && dbeta) db32 = dbeta[ch];
    }
    group_tile_sums(part, part + (size_t)ntiles * c, ntiles, c, ch0, cg, sh, prefetch);
    if (fin) {
        const double sdy = sh[0][0][tid], sdyx = sh[1][0][tid];
        if (pub) {
            if (dgamma) dgamma[ch0 + tid] = dg32 + (float)sdyx;     // gradients accumulate into a zeroed arena
            if (dbeta) dbeta[ch0 + tid] = db32 + (float)sdy;
        }
        const double g = g32, is = is32, mu = mu32, s = g * is;
        cf[0][tid] = (float)s;
        cf[1][tid] = (float)(-s * is * sdyx / count);
        cf[2][tid] = (float)(-s * sdy / count + s * is * mu * sdyx / count);
    }
    __syncthreads();
    float ka[CH], kb[CH], kc[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) { ka[e] = cf[0][col * CH + e]; kb[e] = cf[1][col * CH + e]; kc[e] = cf[2][col * CH + e]; }
    auto emit = [&](const Raw16<T>& dr16, const Raw16<T>& xr16, int64_t r) {
        Chunk<T> d, x, o;
        dr16.unpack(d);
        xr16.unpack(x);
#pragma unroll
        for (int e = 0; e < CH; ++e) o.v[e] = ka[e] * d.v[e] + kb[e] * x.v[e] + kc[e];
        o.store(dr + (size_t)r * c + coff);
    };
    for (;;) {
#pragma unroll
        for (int k = 0; k < DEPTH; ++k)
            if (row + k * stride < rows) emit(pd[k], px[k], row + k * stride);
        row += DEPTH * stride;
        if (row >= rows) break;
#pragma unroll
        for (int k = 0; k < DEPTH; ++k)
            if (row + k * stride < rows) {
                pd[k].load(dy + (size_t)(row + k * stride) * c + coff);
                px[k].load(raw + (size_t)(row + k * stride) * c + coff);
            }
    }
}

// ------------------------------------------------------------------ losses
__device__ __forceinline__ float block_sum(float v, float* sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0) for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
    return t;   // valid on thread 0
}

template <typename T>
__global__ __launch_bounds__(256) void k_bce_logits(const T* __restrict__ logits, int64_t count, int ld, float target,
                                                    float lscale, float gscale, float* loss_acc, T* dlogits, int parts) {
    constexpr int CH = ElemTraits<T>::CH;
    __shared__ float sh[4];
    float local = 0.f;
    const int cpr = ld / CH;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
        const float x = to_f32(logits[i * ld]);
        local += fmaxf(x, 0.f) - x * target + log1pf(expf(-fabsf(x)));
        if (dlogits) {
            const float sg = 1.f / (1.f + expf(-x));
            Chunk<T> o;
#pragma unroll
            for (int e = 0; e < CH; ++e) o.v[e] = 0.f;
            for (int k = 1; k < cpr; ++k) o.store(dlogits + i * ld + k * CH);
            o.v[0] = gscale * (sg - target) / (float)count;
            o.store(dlogits + i * ld);
        }
    }
    const float t = block_sum(local, sh);
    // parts: one partial per workgroup (added in index order by k_loss_finish: reproducible) instead of a float atomic
    if (threadIdx.x == 0 && loss_acc) {
        if (parts) loss_acc[blockIdx.x] = lscale * t / (float)count;
        else atomicAdd(loss_acc, lscale * t / (float)count);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_l1_tanh_bwd(const T* __restrict__ fake, const T* __restrict__ real,
                                                     int64_t count, int ld, int c_real, const T* __restrict__ dgan,
                                                     int dgan_ld, int dgan_off, float lambda, float* loss_acc,
                                                     T* dpre, int parts) {
    constexpr int CH = ElemTraits<T>::CH;
    __shared__ float sh[4];
    const int cpr = ld / CH;
    const int64_t nchunks = count * cpr;
    const float inv = 1.f / ((float)count * (float)c_real);
    float local = 0.f;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < nchunks; idx += (int64_t)gridDim.x * 256) {
        const int64_t pix = idx / cpr;
        const int c0 = (int)(idx - pix * cpr) * CH;
        Chunk<T> f, r, o;
        f.load(fake + idx * CH);
        r.load(real + idx * CH);
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            float d = 0.f;
            if (c0 + e < c_real) {
                local += fabsf(f.v[e] - r.v[e]);
                d = l1_tanh_grad(f.v[e], r.v[e], dgan ? to_f32(dgan[pix * dgan_ld + dgan_off + c0 + e]) : 0.f, lambda, inv);
            }
            o.v[e] = d;
        }
        if (dpre) o.store(dpre + idx * CH);
    }
    const float t = block_sum(local, sh);
    if (threadIdx.x == 0 && loss_acc) {
        if (parts) loss_acc[blockIdx.x] = lambda * t * inv;
        else atomicAdd(loss_acc, lambda * t * inv);
    }
}

// out[0] = sum parts[b0 .. e0) + sum parts[b1 .. e1), out[1] = sum parts[b2 .. e2) + sum parts[b3 .. e3) (b2 < 0: out[0] only):
// one workgroup, every range in a fixed order (lane-strided, then the wave / block tree of block_sum)
struct LossRanges { int b[4], e[4]; };
__global__ __launch_bounds__(256) void k_loss_finish(const float* __restrict__ parts, const LossRanges r, float* out) {
    __shared__ float sh[4];
    float res[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float local = 0.f;
        for (int i = r.b[k] + (int)threadIdx.x; i < r.e[k]; i += 256) local += parts[i];
        res[k] = block_sum(local, sh);
        __syncthreads();                         // sh is reused by the next range
    }
    if (threadIdx.x == 0) {
        out[0] = res[0] + res[1];
        if (r.b[2] >= 0) out[1] = res[2] + res[3];
    }
}

// ------------------------------------------------------------------ column sums
// Pass 1: at most COLSUM_BLOCKS blocks; a lane owns one 16-byte column chunk and strides over rows, the
// per-column partials of a block are combined through LDS.  Pass 2: tile_sum over the block partials.
constexpr int COLSUM_BLOCKS = 512;
template <typename T>
__global__ __launch_bounds__(256) void k_colsum_partial(const T* __restrict__ x, int64_t rows, int c, int cpr,
                                                        float* ws) {
    constexpr int CH = ElemTraits<T>::CH;
    __shared__ float red[256][CH];
    const int tid = threadIdx.x;
    const int col = tid % cpr, rsub = tid / cpr, rpp = 256 / cpr;
    float s[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) s[e] = 0.f;
    for (int64_t row = (int64_t)blockIdx.x * rpp + rsub; row < rows; row += (int64_t)gridDim.x * rpp) {
        Chunk<T> v;
        v.load(x + (row * cpr + col) * CH);
#pragma unroll
        for (int e = 0; e < CH; ++e) s[e] += v.v[e];
    }
#pragma unroll
    for (int e = 0; e < CH; ++e) red[tid][e] = s[e];
    __syncthreads();
    // fixed-shape tree over the rpp row groups (rpp = 256/cpr is a power of two): log2(rpp) steps instead of one lane
    // walking rpp rows (27 us for an 8-channel tensor)
    for (int off = rpp >> 1; off > 0; off >>= 1) {
        if (rsub < off) {
#pragma unroll
            for (int e = 0; e < CH; ++e) red[tid][e] += red[tid + off * cpr][e];
        }
        __syncthreads();
    }
    if (tid < cpr) {
#pragma unroll
        for (int e = 0; e < CH; ++e) ws[(size_t)blockIdx.x * c + tid * CH + e] = red[tid][e];
    }
}
__global__ __launch_bounds__(256) void k_colsum_final(const float* __restrict__ ws, int nblk, int c, int c_out,
                                                      float* out) {
    __shared__ double sh[2][FIN_LANES][FIN_CH];
    const int ch = blockIdx.x * FIN_CH + threadIdx.x % FIN_CH;
    double s, unused;
    tile_sum2(ws, ws, nblk, c, ch, ch < c_out, sh, s, unused);
    if (threadIdx.x < FIN_CH && ch < c_out) out[ch] += (float)s;
}

// ------------------------------------------------------------------ Adam
__global__ void k_adam_tick(int32_t* step) { *step += 1; }

// optim.Adam(lr, betas) of one element (torch: step_size = lr / (1 - b1^t); denom = sqrt(v) / sqrt(1 - b2^t) + eps).  Three
// kernels apply it (k_adam, k_adam_shadow, k_adam_gaps) and must give the same bits: no contraction.
struct AdamK { float b1, b2, eps, step_size, bc2s, grad_scale; };
__device__ __forceinline__ AdamK adam_consts(float lr, float b1, float b2, float eps, int step, const int32_t* step_dev,
                                             float grad_scale) {
    const int t = step_dev ? *step_dev : step;
    const float bc1 = 1.f - powf(b1, (float)t);
    return AdamK{b1, b2, eps, lr / bc1, sqrtf(1.f - powf(b2, (float)t)), grad_scale};
}
__device__ __forceinline__ void adam_update(const AdamK& k, float& p, float g, float& m, float& v) {
#pragma clang fp contract(off)
    const float ge = g * k.grad_scale;
    const float m1 = k.b1 * m;
    const float m2 = (1.f - k.b1) * ge;
    m = m1 + m2;
    const float v1 = k.b2 * v;
    const float v2 = (1.f - k.b2) * ge;
    const float v3 = v2 * ge;
    v = v1 + v3;
    const float den = sqrtf(v) / k.bc2s + k.eps;
    const float upd = k.step_size * (m / den);
    p = p - upd;
}

__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g,
                                              const float* __restrict__ g2, float* __restrict__ m,
                                              float* __restrict__ v, int64_t n, float lr, float b1, float b2,
                                              float eps, int step, const int32_t* step_dev, float grad_scale, int nt) {
    const AdamK k = adam_consts(lr, b1, b2, eps, step, step_dev, grad_scale);
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        // m and v are touched once per step and nowhere else: non-temporal, so that 650 MB of moments do not sweep the
        // activations and weight copies the next kernels read out of the L2 / MALL (pfa_set_knob(1, 256) = plain, A/B)
        f32x4 pp = ((f32x4*)p)[i], gg, mm, vv;
        if (nt) {
            gg = nt & 2 ? __builtin_nontemporal_load((const f32x4*)g + i) : ((const f32x4*)g)[i];
            mm = __builtin_nontemporal_load((f32x4*)m + i); vv = __builtin_nontemporal_load((f32x4*)v + i);
        } else { gg = ((const f32x4*)g)[i]; mm = ((f32x4*)m)[i]; vv = ((f32x4*)v)[i]; }
        if (g2) gg += ((const f32x4*)g2)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float pe = pp[e], me = mm[e], ve = vv[e];
            adam_update(k, pe, gg[e], me, ve);
            pp[e] = pe; mm[e] = me; vv[e] = ve;
        }
        ((f32x4*)p)[i] = pp;
        if (nt) { __builtin_nontemporal_store(mm, (f32x4*)m + i); __builtin_nontemporal_store(vv, (f32x4*)v + i); }
        else { ((f32x4*)m)[i] = mm; ((f32x4*)v)[i] = vv; }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        float pe = p[i], me = m[i], ve = v[i];
        adam_update(k, pe, g[i] + (g2 ? g2[i] : 0.f), me, ve);
        p[i] = pe; m[i] = me; v[i] = ve;
    }
}

// Adam on the arena ranges that are NOT convolution weights (BatchNorm gamma / beta, biases: the gaps between the layers a
// k_adam_shadow launch covers): one workgroup per range, tab[i] = {first element, length}.
__global__ __launch_bounds__(256) void k_adam_gaps(const int64_t* __restrict__ tab, float* __restrict__ p,
                                                   const float* __restrict__ g, const float* __restrict__ g2,
                                                   float* __restrict__ m, float* __restrict__ v, float lr, float b1, float b2,
                                                   float eps, int step, const int32_t* step_dev, float grad_scale) {
    const AdamK k = adam_consts(lr, b1, b2, eps, step, step_dev, grad_scale);
    const int64_t o = tab[2 * blockIdx.x], n = tab[2 * blockIdx.x + 1];
    for (int64_t j = threadIdx.x; j < n; j += 256) {
        const int64_t i = o + j;
        float pe = p[i], me = m[i], ve = v[i];
        adam_update(k, pe, g[i] + (g2 ? g2[i] : 0.f), me, ve);
        p[i] = pe; m[i] = me; v[i] = ve;
    }
}

// ------------------------------------------------------------------ weight shadows
__device__ __forceinline__ int real_channel(int cp, int c1r, int c1p, int c2r) {
    if (cp < c1p) return cp < c1r ? cp : -1;
    const int c2 = cp - c1p;
    return c2 < c2r ? c1r + c2 : -1;
}

template <typename T>
__global__ __launch_bounds__(256) void k_shadow_same(const float* __restrict__ master, int rows, int cr, int c1r,
                                                     int c1p, int c2r, int cp_tot, T* same) {
    const int64_t total = (int64_t)rows * 16 * cp_tot;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int cp = (int)(idx % cp_tot);
        const int64_t rt = idx / cp_tot;     // row*16 + tap
        const int c = real_channel(cp, c1r, c1p, c2r);
        same[idx] = from_f32<T>(c >= 0 ? master[rt * cr + c] : 0.f);
    }
}

// tr[cp][tap][r] = master[r][tap][creal(cp)]: 32x32 tiles through LDS, both sides coalesced.
template <typename T>
__global__ __launch_bounds__(256) void k_shadow_tr(const float* __restrict__ master, int rows, int cr, int c1r,
                                                   int c1p, int c2r, int cp_tot, int rows_p, T* tr) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z;
    const int cp0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, cp = cp0 + tx;
        float v = 0.f;
        if (r < rows && cp < cp_tot) {
            const int c = real_channel(cp, c1r, c1p, c2r);
            if (c >= 0) v = master[((size_t)r * 16 + tap) * cr + c];
        }
        tile[ty + 8 * k][tx] = v;
    }
    __syncthreads();
    for (int k = 0; k < 4; ++k) {
        const int cp = cp0 + ty + 8 * k, r = r0 + tx;
        if (cp < cp_tot && r < rows_p) tr[((size_t)cp * 16 + tap) * rows_p + r] = from_f32<T>(tile[tx][ty + 8 * k]);
    }
}

constexpr int SHADOW_ELEMS = 256 * 8;
// ------------------------------------------------------------------ NCHW float32 <-> NHWC T
template <typename T>
__global__ __launch_bounds__(256) void k_pack_nhwc(const float* __restrict__ src, int n, int c, int64_t hw, int cp,
                                                   T* dst) {
    const int64_t total = (int64_t)n * hw;
    for (int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x; pix < total; pix += (int64_t)gridDim.x * 256) {
        const int64_t img = pix / hw, off = pix - img * hw;
        for (int ch = 0; ch < cp; ++ch)
            dst[pix * cp + ch] = from_f32<T>(ch < c ? src[(img * c + ch) * hw + off] : 0.f);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_unpack_nchw(const T* __restrict__ src, int n, int c, int64_t hw, int cp,
                                                     float* dst) {
    const int64_t total = (int64_t)n * c * hw;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t off = idx % hw, t = idx / hw;
        const int ch = (int)(t % c);
        const int64_t img = t / c;
        dst[idx] = to_f32(src[(img * hw + off) * cp + ch]);
    }
}


// Both shadows of every layer in ONE launch: a workgroup reads a 64 (rows) x 64 (padded channels) tile of one tap from the
// fp32 master once (256-byte coalesced rows), keeps it in LDS and writes `same` (rows of 64 channels) and `tr` (rows of 64
// master-rows) with 16-byte stores.  The layer of a block is found by one ballot over the descriptor table (a serial scan
// costs a load latency per layer and block).  335 MB instead of 502 MB for the generator, no per-element div/mod.
// ADAM: the same launch first applies optim.Adam to the master elements of its tile (p, g [+ g2], m, v of the flat arenas:
// a weight's elements sit at the same offsets in all of them) - one pass over the parameters instead of k_adam followed by a
// re-read of the master for the copies.
struct AdamArenas {
    float* p; const float* g; const float* g2; float* m; float* v;
    float lr, b1, b2, eps, grad_scale; int step; const int32_t* step_dev;
};
constexpr int SH_TILE = 64;
template <typename T, bool ADAM>
__global__ __launch_bounds__(256) void k_shadow_fused_batch(const pfa_shadow_desc* __restrict__ descs, int n, const AdamArenas A) {
    constexpr int CH = ElemTraits<T>::CH;
    constexpr int GROUPS = SH_TILE / CH;                  // 16-byte groups per tile row
    // 64 x 64 floats, element (row, col) at column (col + row) % 64: rows and columns both fall on distinct banks without a
    // padding column - 16,384 bytes exactly, what a CU has left beside a 144-KiB patch-kernel workgroup of the backward chain
    // this launch runs under (with 16,640 bytes the two could not share a CU)
    __shared__ float tile_[SH_TILE * SH_TILE];
    auto tile = [&](int row, int col) -> float& { return tile_[row * SH_TILE + ((col + row) & (SH_TILE - 1))]; };
    const int tid = threadIdx.x, lane = tid & 63;
    const int mine = lane < n ? descs[lane].tr_blk0 : 0x7fffffff;
    const int l = __popcll(__ballot(mine <= (int)blockIdx.x)) - 1;
    const pfa_shadow_desc d = descs[l];
    const int cr = d.c1r + d.c2r, cp_tot = d.c1p + d.c2p;
    const int tc = (cp_tot + SH_TILE - 1) / SH_TILE, trn = (d.rows_p + SH_TILE - 1) / SH_TILE;
    int lb = (int)blockIdx.x - d.tr_blk0;
    const int cp0 = (lb % tc) * SH_TILE; lb /= tc;
    const int r0 = (lb % trn) * SH_TILE;
    const int tap = lb / trn;
    {
        const int tx = tid & 63, ty = tid >> 6;
        const int cp = cp0 + tx;
        const int c = cp < cp_tot ? real_channel(cp, d.c1r, d.c1p, d.c2r) : -1;
        if (!ADAM) {
#pragma unroll
            for (int k = 0; k < SH_TILE / 4; ++k) {
                const int r = r0 + ty + 4 * k;
                tile(ty + 4 * k, tx) = (c >= 0 && r < d.rows) ? d.master[((size_t)r * 16 + tap) * cr + c] : 0.f;
            }
        } else if (d.c1r == d.c1p && d.c2r == d.c2p && (cr & 3) == 0) {
            // no padded channels: the tile's rows are 256 contiguous bytes of the arenas - 16 lanes x 16 bytes per row, 16 rows
            // per pass, every access a whole 16-byte vector as in k_adam
            const AdamK ak = adam_consts(A.lr, A.b1, A.b2, A.eps, A.step, A.step_dev, A.grad_scale);
            const size_t base = (size_t)(d.master - A.p);
            const int q = tid & 15, rr = tid >> 4, cpq = cp0 + 4 * q;
#pragma unroll
            for (int k = 0; k < SH_TILE / 16; ++k) {
                const int r = r0 + rr + 16 * k;
                f32x4 pv = {0.f, 0.f, 0.f, 0.f};
                if (cpq < cp_tot && r < d.rows) {
                    const size_t i = (base + ((size_t)r * 16 + tap) * cr + cpq) >> 2;
                    pv = ((f32x4*)A.p)[i];
                    f32x4 gv = ((const f32x4*)A.g)[i];
                    f32x4 mv = __builtin_nontemporal_load((f32x4*)A.m + i), vv = __builtin_nontemporal_load((f32x4*)A.v + i);
                    if (A.g2) gv += ((const f32x4*)A.g2)[i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float pe = pv[e], me = mv[e], ve = vv[e];
                        adam_update(ak, pe, gv[e], me, ve);
                        pv[e] = pe; mv[e] = me; vv[e] = ve;
                    }
                    ((f32x4*)A.p)[i] = pv;
                    __builtin_nontemporal_store(mv, (f32x4*)A.m + i);
                    __builtin_nontemporal_store(vv, (f32x4*)A.v + i);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) tile(rr + 16 * k, 4 * q + e) = pv[e];
            }
        } else {
            const AdamK ak = adam_consts(A.lr, A.b1, A.b2, A.eps, A.step, A.step_dev, A.grad_scale);
            const size_t base = (size_t)(d.master - A.p);           // the weight's first element in the arenas
#pragma unroll 4
            for (int k = 0; k < SH_TILE / 4; ++k) {
                const int r = r0 + ty + 4 * k;
                float pe = 0.f;
                if (c >= 0 && r < d.rows) {
                    const size_t i = base + ((size_t)r * 16 + tap) * cr + c;
                    pe = A.p[i];
                    float me = __builtin_nontemporal_load(A.m + i), ve = __builtin_nontemporal_load(A.v + i);
                    adam_update(ak, pe, A.g[i] + (A.g2 ? A.g2[i] : 0.f), me, ve);
                    A.p[i] = pe;
                    __builtin_nontemporal_store(me, A.m + i);
                    __builtin_nontemporal_store(ve, A.v + i);
                }
                tile(ty + 4 * k, tx) = pe;
            }
        }
    }
    __syncthreads();
    T* same = (T*)d.same;
    T* trp = (T*)d.tr;
#pragma unroll
    for (int it = tid; it < SH_TILE * GROUPS; it += 256) {
        const int line = it / GROUPS, g = it % GROUPS;
        if (same) {                                         // line = master row, g = channel group
            const int r = r0 + line, cpb = cp0 + g * CH;
            if (r < d.rows && cpb < cp_tot) {
                Chunk<T> o;
#pragma unroll
                for (int e = 0; e < CH; ++e) o.v[e] = tile(line, g * CH + e);
                o.store(same + ((size_t)r * 16 + tap) * cp_tot + cpb);
            }
        }
        if (trp) {                                          // line = padded channel, g = master-row group
            const int cp = cp0 + line, rb = r0 + g * CH;
            if (cp < cp_tot && rb < d.rows_p) {
                Chunk<T> o;
#pragma unroll
                for (int e = 0; e < CH; ++e) o.v[e] = tile(g * CH + e, line);
                o.store(trp + ((size_t)cp * 16 + tap) * d.rows_p + rb);
            }
        }
    }
}

// NHWC (padded) -> packed HWC uint8 image rows, u8 = byte((x*mul + add) * 255): the reference's `tensor*0.5+0.5` followed
// by TF.to_pil_image's mul(255).byte() (generate_synthetic_data.py:66-71), each product / sum rounded on its own like the
// separate torch kernels (no fused multiply-add), the final conversion truncating.
template <typename T>
__global__ __launch_bounds__(256) void k_unpack_u8(const T* __restrict__ src, int64_t pixels, int c, int cp, float mul,
                                                   float add, uint8_t* __restrict__ dst) {
    const int64_t total = pixels * c;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t pix = idx / c;
        const int ch = (int)(idx - pix * c);
        const float t = __fadd_rn(__fmul_rn(to_f32(src[pix * cp + ch]), mul), add);
        const float v = __fmul_rn(t, 255.f);
        dst[idx] = (uint8_t)(int)fminf(fmaxf(v, 0.f), 255.f);
    }
}

}  // namespace

#define DISPATCH_T(dtype, expr_bf16, expr_f32)           \
    do {                                                 \
        if ((dtype) == PFA_BF16) { expr_bf16; }          \
        else if ((dtype) == PFA_F32) { expr_f32; }       \
        else PFA_FAIL(-1, "bad dtype %d", (int)(dtype)); \
    } while (0)

extern "C" int pfa_bn_finalize(const float* psum, const float* psq, int ntiles, int c, double count,
                               const float* gamma, const float* beta, float eps, float momentum, int repeats,
                               float* running_mean, float* running_var, int64_t* nbt, float* scale, float* shift,
                               float* mean, float* invstd, pfa_stream s) {
    PFA_REQUIRE(psum && psq && scale && shift && ntiles > 0 && c > 0 && count > 0, "pfa_bn_finalize: bad arguments");
    hipLaunchKernelGGL(k_bn_finalize, dim3((c + FIN_CH - 1) / FIN_CH), dim3(256), 0, (hipStream_t)s, psum, psq, ntiles, c, count,
                       gamma, beta, eps, momentum, repeats, running_mean, running_var, nbt, scale, shift, mean, invstd);
    PFA_LAUNCHED("k_bn_finalize");
    return 0;
}

extern "C" int pfa_bn_eval_coeffs(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                  int c, float* scale, float* shift, pfa_stream s) {
    PFA_REQUIRE(rm && rv && scale && shift && c > 0, "pfa_bn_eval_coeffs: bad arguments");
    hipLaunchKernelGGL(k_bn_eval_coeffs, dim3((c + 63) / 64), dim3(64), 0, (hipStream_t)s, gamma, beta, rm, rv, eps, c,
                       scale, shift);
    PFA_LAUNCHED("k_bn_eval_coeffs");
    return 0;
}

extern "C" int pfa_bn_act(const void* raw, int64_t rows, int c, int dtype, const float* scale, const float* shift,
                          float slope, void* out_l, void* out_r, pfa_stream s) {
    PFA_REQUIRE(raw && rows > 0 && c > 0 && (out_l || out_r), "pfa_bn_act: bad arguments");
    PFA_REQUIRE(c % pfa_ch(dtype) == 0, "pfa_bn_act: c=%d not a multiple of %d", c, pfa_ch(dtype));
    const int cpr = c / pfa_ch(dtype);
    PFA_REQUIRE(cpr <= 256 ? 256 % cpr == 0 : cpr % 256 == 0, "pfa_bn_act: c/%d = %d must divide or be a multiple of 256", pfa_ch(dtype), cpr);
    const int64_t nchunks = rows * cpr;
    const int grid = grid_chan(nchunks, cpr);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(k_bn_act<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)s,
                                  (const bf16_t*)raw, nchunks, cpr, scale, shift, slope, (bf16_t*)out_l, (bf16_t*)out_r),
               hipLaunchKernelGGL(k_bn_act<float>, dim3(grid), dim3(256), 0, (hipStream_t)s,
                                  (const float*)raw, nchunks, cpr, scale, shift, slope, (float*)out_l, (float*)out_r));
    PFA_LAUNCHED("k_bn_act");
    return 0;
}

// Channel group and grid of the statistics-in-the-prologue passes; cg = 0: shape not built (callers keep the two launches).
static int stats_group(int c, int dtype) {
    const int ch = pfa_ch(dtype);
    if (c % ch) return 0;
    if (c % ST_CG == 0) return ST_CG;
    if (c < ST_CG && c % 4 == 0 && 256 % (c / ch) == 0) return c;
    return 0;
}
static dim3 stats_grid(int64_t rows, int c, int cg, int dtype, int ntiles) {
    const int rpp = 256 / (cg / pfa_ch(dtype)), groups = c / cg;
    const int64_t passes = (rows + rpp - 1) / rpp;
    // every workgroup re-reads ntiles x cg x 2 floats of partials: keep that under ~64 MB per launch
    int64_t cap = ((int64_t)64 << 20) / ((int64_t)ntiles * cg * 8);
    if (cap > 2048) cap = 2048;
    int64_t gx = cap / groups;
    if (gx > (passes + 3) / 4) gx = (passes + 3) / 4;       // at least ~4 row passes per lane
    if (gx < 1) gx = 1;
    return dim3((unsigned)gx, (unsigned)groups);
}

extern "C" int pfa_bn_stats_fused_supported(int dtype, int c, int ntiles) {
    const int lim = pfa_knob(14) > 0 ? pfa_knob(14) : 512;      // (256 -> 512: +0.27 % on the step, three same-box pairs)
    return (dtype == PFA_BF16 || dtype == PFA_F32) && pfa_knob(14) >= 0 && ntiles > 0 && ntiles <= lim && stats_group(c, dtype) > 0;
}

extern "C" int pfa_bn_act_stats(const void* raw, int64_t rows, int c, int dtype, const float* psum, const float* psq,
                                int ntiles, double count, const float* gamma, const float* beta, float eps, float momentum,
                                int repeats, float* running_mean, float* running_var, int64_t* nbt, float* scale,
                                float* shift, float* mean, float* invstd, float slope, void* out_l, void* out_r,
                                pfa_stream s) {
    PFA_REQUIRE(raw && psum && psq && scale && shift && rows > 0 && ntiles > 0 && count > 0 && (out_l || out_r),
                "pfa_bn_act_stats: bad arguments");
    const int cg = stats_group(c, dtype);
    PFA_REQUIRE(cg > 0, "pfa_bn_act_stats: channel count %d not built (pfa_bn_stats_fused_supported)", c);
    const dim3 grid = stats_grid(rows, c, cg, dtype, ntiles);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(k_bn_act_stats<bf16_t>, grid, dim3(256), 0, (hipStream_t)s, (const bf16_t*)raw, rows, c, cg,
                                  psum, psq, ntiles, count, gamma, beta, eps, momentum, repeats, running_mean, running_var,
                                  nbt, scale, shift, mean, invstd, slope, (bf16_t*)out_l, (bf16_t*)out_r),
               hipLaunchKernelGGL(k_bn_act_stats<float>, grid, dim3(256), 0, (hipStream_t)s, (const float*)raw, rows, c, cg,
                                  psum, psq, ntiles, count, gamma, beta, eps, momentum, repeats, running_mean, running_var,
                                  nbt, scale, shift, mean, invstd, slope, (float*)out_l, (float*)out_r));
    PFA_LAUNCHED("k_bn_act_stats");
    return 0;
}

extern "C" int pfa_bn_bwd_apply_stats(const void* dy, const void* raw, int64_t rows, int c, int dtype, const float* part,
                                      int ntiles, double count, const float* gamma, const float* mean, const float* invstd,
                                      float* dgamma, float* dbeta, void* dr, pfa_stream s) {
    PFA_REQUIRE(dy && raw && part && mean && invstd && dr && rows > 0 && ntiles > 0 && count > 0,
                "pfa_bn_bwd_apply_stats: bad arguments");
    const int cg = stats_group(c, dtype);
    PFA_REQUIRE(cg > 0, "pfa_bn_bwd_apply_stats: channel count %d not built (pfa_bn_stats_fused_supported)", c);
    const dim3 grid = stats_grid(rows, c, cg, dtype, ntiles);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(k_bn_bwd_apply_stats<bf16_t>, grid, dim3(256), 0, (hipStream_t)s, (const bf16_t*)dy,
                                  (const bf16_t*)raw, rows, c, cg, part, ntiles, count, gamma, mean, invstd, dgamma, dbeta,
                                  (bf16_t*)dr),
               hipLaunchKernelGGL(k_bn_bwd_apply_stats<float>, grid, dim3(256), 0, (hipStream_t)s, (const float*)dy,
                                  (const float*)raw, rows, c, cg, part, ntiles, count, gamma, mean, invstd, dgamma, dbeta,
                                  (float*)dr));
    PFA_LAUNCHED("k_bn_bwd_apply_stats");
    return 0;
}

// rows per block: fewer passes for small tensors so that even a 4x4 map spreads over the chip
static int bwd_passes(int64_t rows, int cpr) {
    const int rpp = 256 / cpr;
    int64_t p = (rows + (int64_t)rpp * 512 - 1) / ((int64_t)rpp * 512);     // <= ~512 partial tiles for the finalize
    return p < 1 ? 1 : (p > BWD_PASSES ? BWD_PASSES : (int)p);
}

extern "C" int pfa_bn_bwd_reduce(const void* raw, int64_t rows, int c, int dtype, const float* scale,
                                 const float* shift, const float* mean, const float* invstd, const void* g1, int act1,
                                 const void* g2, int act2, float slope, void* dy, float* part, int* ntiles,
                                 pfa_stream s) {
    PFA_REQUIRE(raw && g1 && dy && rows > 0 && c > 0, "pfa_bn_bwd_reduce: bad arguments");
    const int ch = pfa_ch(dtype);
    PFA_REQUIRE(c % ch == 0, "pfa_bn_bwd_reduce: c=%d not a multiple of %d", c, ch);
    const int cpr = c / ch;
    PFA_REQUIRE(cpr <= 256 && 256 % cpr == 0, "pfa_bn_bwd_reduce: c/%d = %d must divide 256", ch, cpr);
    PFA_REQUIRE((mean == nullptr) == (part == nullptr), "pfa_bn_bwd_reduce: mean and part come together");
    const int passes = bwd_passes(rows, cpr);
    const int rows_per_block = (256 / cpr) * passes;
    const int tiles = (int)((rows + rows_per_block - 1) / rows_per_block);
    if (ntiles) *ntiles = tiles;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(k_bn_bwd_reduce<bf16_t>, dim3(tiles), dim3(256), 0, (hipStream_t)s,
                                  (const bf16_t*)raw, rows, c, cpr, scale, shift, mean, invstd, (const bf16_t*)g1, act1,
                                  (const bf16_t*)g2, act2, slope, (bf16_t*)dy, part, tiles, passes),
               hipLaunchKernelGGL(k_bn_bwd_reduce<float>, dim3(tiles), dim3(256), 0, (hipStream_t)s, (const float*)raw,
                                  rows, c, cpr, scale, shift, mean, invstd, (const float*)g1, act1, (const float*)g2,
                                  act2, slope, (float*)dy, part, tiles, passes));
    PFA_LAUNCHED("k_bn_bwd_reduce");
    return 0;
}

extern "C" int pfa_bn_bwd_tiles(int64_t rows, int c, int dtype) {
    const int cpr = c / pfa_ch(dtype);
    if (cpr < 1 || cpr > 256 || 256 % cpr) return -1;
    const int rows_per_block = (256 / cpr) * bwd_passes(rows, cpr);
    return (int)((rows + rows_per_block - 1) / rows_per_block);
}

extern "C" int pfa_bn_bwd_finalize(const float* part, int ntiles, int c, double count, const float* gamma,
                                   const float* mean, const float* invstd, float* dgamma, float* dbeta, float* coef,
                                   pfa_stream s) {
    PFA_REQUIRE(part && mean && invstd && coef && ntiles > 0 && c > 0, "pfa_bn_bwd_finalize: bad arguments");
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3((c + FIN_CH - 1) / FIN_CH), dim3(256), 0, (hipStream_t)s, part, ntiles, c, count,
                       gamma, mean, invstd, dgamma, dbeta, coef);
    PFA_LAUNCHED("k_bn_bwd_finalize");
    return 0;
}

extern "C" int pfa_bn_bwd_apply(const void* dy, const void* raw, int64_t rows, int c, int dtype, const float* coef,
                                void* dr, pfa_stream s) {
    PFA_REQUIRE(dy && raw && coef && dr && rows > 0, "pfa_bn_bwd_apply: bad arguments");
    PFA_REQUIRE(c % pfa_ch(dtype) == 0, "pfa_bn_bwd_apply: c not a multiple of the chunk width");
    const int cpr = c / pfa_ch(dtype);
    PFA_REQUIRE(cpr <= 256 ? 256 % cpr == 0 : cpr % 256 == 0, "pfa_bn_bwd_apply: unsupported channel count %d", c);
    const int64_t nchunks = rows * cpr;
    const int grid = grid_chan(nchunks, cpr);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(k_bn_bwd_apply<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)s,
                                  (const bf16_t*)dy, (const bf16_t*)raw, nchunks, cpr, c, coef, (bf16_t*)dr),
               hipLaunchKernelGGL(k_bn_bwd_apply<float>, dim3(grid), dim3(256), 0, (hipStream_t)s,
                                  (const float*)dy, (const float*)raw, nchunks, cpr, c, coef, (float*)dr));
    PFA_LAUNCHED("k_bn_bwd_apply");
    return 0;
}

static int bce_logits_launch(const void* logits, int64_t count, int ld, int dtype, float target, float lscale,
                             float gscale, float* loss_acc, void* dlogits, pfa_stream s, int parts) {
    PFA_REQUIRE(logits && count > 0 && ld > 0 && ld % pfa_ch(dtype) == 0, "pfa_bce_logits: bad arguments");
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(k_bce_logits<bf16_t>, dim3(grid_for(count, 256, 64)), dim3(256), 0, (hipStream_t)s,
                                  (const bf16_t*)logits, count, ld, target, lscale, gscale, loss_acc, (bf16_t*)dlogits, parts),
               hipLaunchKernelGGL(k_bce_logits<float>, dim3(grid_for(count, 256, 64)), dim3(256), 0, (hipStream_t)s,
                                  (const float*)logits, count, ld, target, lscale, gscale, loss_acc, (float*)dlogits, parts));
    PFA_LAUNCHED("k_bce_logits");
    return 0;
}

extern "C" int pfa_bce_logits(const void* logits, int64_t count, int ld, int dtype, float target, float lscale,
                              float gscale, float* loss_acc, void* dlogits, pfa_stream s) {
    return bce_logits_launch(logits, count, ld, dtype, target, lscale, gscale, loss_acc, dlogits, s, 0);
}

extern "C" int pfa_bce_logits_parts(const void* logits, int64_t count, int ld, int dtype, float target, float lscale,
                                    float gscale, float* parts, void* dlogits, pfa_stream s) {
    PFA_REQUIRE(parts, "pfa_bce_logits_parts: no partials buffer");
    return bce_logits_launch(logits, count, ld, dtype, target, lscale, gscale, parts, dlogits, s, 1);
}

extern "C" int pfa_bce_logits_nparts(int64_t count) { return count > 0 ? grid_for(count, 256, 64) : 0; }

static int l1_tanh_bwd_launch(const void* fake, const void* real, int64_t count, int ld, int c_real, int dtype,
                              const void* dgan, int dgan_ld, int dgan_off, float lambda, float* loss_acc, void* dpre,
                              pfa_stream s, int parts) {
    PFA_REQUIRE(fake && real && count > 0 && ld > 0 && ld % pfa_ch(dtype) == 0 && c_real > 0 && c_real <= ld,
                "pfa_l1_tanh_bwd: bad arguments");
    const int64_t nchunks = count * (ld / pfa_ch(dtype));
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(k_l1_tanh_bwd<bf16_t>, dim3(grid_for(nchunks, 256, 1024)), dim3(256), 0,
                                  (hipStream_t)s, (const bf16_t*)fake, (const bf16_t*)real, count, ld, c_real,
                                  (const bf16_t*)dgan, dgan_ld, dgan_off, lambda, loss_acc, (bf16_t*)dpre, parts),
               hipLaunchKernelGGL(k_l1_tanh_bwd<float>, dim3(grid_for(nchunks, 256, 1024)), dim3(256), 0,
                                  (hipStream_t)s, (const float*)fake, (const float*)real, count, ld, c_real,
                                  (const float*)dgan, dgan_ld, dgan_off, lambda, loss_acc, (float*)dpre, parts));
    PFA_LAUNCHED("k_l1_tanh_bwd");
    return 0;
}

extern "C" int pfa_l1_tanh_bwd(const void* fake, const void* real, int64_t count, int ld, int c_real, int dtype,
                               const void* dgan, int dgan_ld, int dgan_off, float lambda, float* loss_acc, void* dpre,
                               pfa_stream s) {
    return l1_tanh_bwd_launch(fake, real, count, ld, c_real, dtype, dgan, dgan_ld, dgan_off, lambda, loss_acc, dpre, s, 0);
}

extern "C" int pfa_l1_tanh_bwd_parts(const void* fake, const void* real, int64_t count, int ld, int c_real, int dtype,
                                     const void* dgan, int dgan_ld, int dgan_off, float lambda, float* parts, void* dpre,
                                     pfa_stream s) {
    PFA_REQUIRE(parts, "pfa_l1_tanh_bwd_parts: no partials buffer");
    return l1_tanh_bwd_launch(fake, real, count, ld, c_real, dtype, dgan, dgan_ld, dgan_off, lambda, parts, dpre, s, 1);
}

extern "C" int pfa_l1_tanh_bwd_nparts(int64_t count, int ld, int dtype) {
    if (count <= 0 || ld <= 0 || pfa_ch(dtype) <= 0) return 0;
    return grid_for(count * (ld / pfa_ch(dtype)), 256, 1024);
}

extern "C" int pfa_loss_finish(const float* parts, int b0, int e0, int b1, int e1, int b2, int e2, int b3, int e3, float* out,
                               pfa_stream s) {
    PFA_REQUIRE(parts && out && b0 >= 0 && b1 >= 0 && (b2 < 0 || b3 >= 0), "pfa_loss_finish: bad arguments");
    LossRanges r;
    r.b[0] = b0; r.e[0] = e0; r.b[1] = b1; r.e[1] = e1; r.b[2] = b2; r.e[2] = e2; r.b[3] = b3; r.e[3] = e3;
    hipLaunchKernelGGL(k_loss_finish, dim3(1), dim3(256), 0, (hipStream_t)s, parts, r, out);
    PFA_LAUNCHED("k_loss_finish");
    return 0;
}

extern "C" size_t pfa_colsum_ws(int64_t rows, int c) { return (size_t)COLSUM_BLOCKS * (size_t)c; }

extern "C" int pfa_colsum(const void* x, int64_t rows, int c, int c_out, int dtype, float* out, float* ws,
                          pfa_stream s) {
    PFA_REQUIRE(x && out && ws && rows > 0 && c > 0 && c_out > 0 && c_out <= c, "pfa_colsum: bad arguments");
    const int ch = pfa_ch(dtype);
    PFA_REQUIRE(c % ch == 0, "pfa_colsum: c=%d not a multiple of %d", c, ch);
    const int cpr = c / ch;
    PFA_REQUIRE(cpr <= 256 && 256 % cpr == 0, "pfa_colsum: c/%d = %d must divide 256", ch, cpr);
    const int rpp = 256 / cpr;
    int nblk = (int)((rows + rpp - 1) / rpp);
    if (nblk > COLSUM_BLOCKS) nblk = COLSUM_BLOCKS;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(k_colsum_partial<bf16_t>, dim3(nblk), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x,
                                  rows, c, cpr, ws),
               hipLaunchKernelGGL(k_colsum_partial<float>, dim3(nblk), dim3(256), 0, (hipStream_t)s, (const float*)x,
                                  rows, c, cpr, ws));
    PFA_LAUNCHED("k_colsum_partial");
    hipLaunchKernelGGL(k_colsum_final, dim3((c_out + FIN_CH - 1) / FIN_CH), dim3(256), 0, (hipStream_t)s, ws, nblk, c,
                       c_out, out);
    PFA_LAUNCHED("k_colsum_final");
    return 0;
}

extern "C" int pfa_adam(float* p, const float* g, const float* g2, float* m, float* v, int64_t n, float lr, float beta1,
                        float beta2, float eps, int step, const int32_t* step_dev, float grad_scale, pfa_stream s) {
    PFA_REQUIRE(p && g && m && v && n > 0, "pfa_adam: bad arguments");
    PFA_REQUIRE(step_dev || step >= 1, "pfa_adam: step must be >= 1");
    PFA_REQUIRE(((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 == 0, "pfa_adam: arenas must be 16-byte aligned");
    PFA_REQUIRE(!g2 || (uintptr_t)g2 % 16 == 0, "pfa_adam: second gradient arena must be 16-byte aligned");
    hipLaunchKernelGGL(k_adam, dim3(grid_for(n >> 2, 256, pfa_knob(11) > 0 ? pfa_knob(11) : 2048)), dim3(256), 0, (hipStream_t)s, p, g, g2, m, v, n, lr,
                       beta1, beta2, eps, step, step_dev, grad_scale, (pfa_knob(1) & 256) ? 0 : ((pfa_knob(1) & 512) ? 3 : 1));
    PFA_LAUNCHED("k_adam");
    return 0;
}

extern "C" int pfa_adam_tick(int32_t* step_dev, pfa_stream s) {
    PFA_REQUIRE(step_dev, "pfa_adam_tick: null");
    hipLaunchKernelGGL(k_adam_tick, dim3(1), dim3(1), 0, (hipStream_t)s, step_dev);
    PFA_LAUNCHED("k_adam_tick");
    return 0;
}

extern "C" int pfa_weight_shadow(const float* master, int rows, int c1r, int c1p, int c2r, int c2p, int rows_p,
                                 int dtype, void* same, void* tr, pfa_stream s) {
    PFA_REQUIRE(master && rows > 0 && c1r > 0 && c1r <= c1p && c2r <= c2p && rows_p >= rows, "pfa_weight_shadow: bad arguments");
    const int cr = c1r + c2r, cp = c1p + c2p;
    if (same) {
        const int64_t total = (int64_t)rows * 16 * cp;
        DISPATCH_T(dtype,
                   hipLaunchKernelGGL(k_shadow_same<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s,
                                      master, rows, cr, c1r, c1p, c2r, cp, (bf16_t*)same),
                   hipLaunchKernelGGL(k_shadow_same<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, master,
                                      rows, cr, c1r, c1p, c2r, cp, (float*)same));
        PFA_LAUNCHED("k_shadow_same");
    }
    if (tr) {
        dim3 grid((cp + 31) / 32, (rows_p + 31) / 32, 16);
        DISPATCH_T(dtype,
                   hipLaunchKernelGGL(k_shadow_tr<bf16_t>, grid, dim3(256), 0, (hipStream_t)s, master, rows, cr, c1r,
                                      c1p, c2r, cp, rows_p, (bf16_t*)tr),
                   hipLaunchKernelGGL(k_shadow_tr<float>, grid, dim3(256), 0, (hipStream_t)s, master, rows, cr, c1r, c1p,
                                      c2r, cp, rows_p, (float*)tr));
        PFA_LAUNCHED("k_shadow_tr");
    }
    return 0;
}

extern "C" int pfa_shadow_blocks(int rows, int c1p, int c2p, int rows_p, int* same_blocks, int* tr_blocks) {
    PFA_REQUIRE(rows > 0 && c1p > 0 && c2p >= 0 && rows_p >= rows && same_blocks && tr_blocks, "pfa_shadow_blocks: bad arguments");
    const int cp = c1p + c2p;
    *same_blocks = (int)(((int64_t)rows * 16 * cp + SHADOW_ELEMS - 1) / SHADOW_ELEMS);      // (unused by the fused launch)
    *tr_blocks = ((cp + SH_TILE - 1) / SH_TILE) * ((rows_p + SH_TILE - 1) / SH_TILE) * 16;     // 64x64 tiles x 16 taps
    return 0;
}

extern "C" int pfa_weight_shadow_batch(const pfa_shadow_desc* descs_dev, int n, int same_blocks, int tr_blocks,
                                       int dtype, pfa_stream s) {
    PFA_REQUIRE(descs_dev && n > 0 && n <= 64 && tr_blocks > 0, "pfa_weight_shadow_batch: bad arguments (at most 64 layers per table)");
    (void)same_blocks;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL((k_shadow_fused_batch<bf16_t, false>), dim3(tr_blocks), dim3(256), 0, (hipStream_t)s, descs_dev, n, AdamArenas{}),
               hipLaunchKernelGGL((k_shadow_fused_batch<float, false>), dim3(tr_blocks), dim3(256), 0, (hipStream_t)s, descs_dev, n, AdamArenas{}));
    PFA_LAUNCHED("k_shadow_fused_batch");
    return 0;
}

extern "C" int pfa_adam_shadow_batch(const pfa_shadow_desc* descs_dev, int n, int tr_blocks, int dtype, const int64_t* gaps_dev,
                                     int ngaps, float* p, const float* g, const float* g2, float* m, float* v, float lr,
                                     float beta1, float beta2, float eps, int step, const int32_t* step_dev, float grad_scale,
                                     pfa_stream s) {
    PFA_REQUIRE(p && g && m && v && (n == 0 || (descs_dev && n > 0 && n <= 64 && tr_blocks > 0)) && (ngaps == 0 || gaps_dev) &&
                (n > 0 || ngaps > 0), "pfa_adam_shadow_batch: bad arguments (at most 64 layers per table)");
    PFA_REQUIRE(step_dev || step >= 1, "pfa_adam_shadow_batch: step must be >= 1");
    const AdamArenas A{p, g, g2, m, v, lr, beta1, beta2, eps, grad_scale, step, step_dev};
    if (n > 0) {
        DISPATCH_T(dtype,
                   hipLaunchKernelGGL((k_shadow_fused_batch<bf16_t, true>), dim3(tr_blocks), dim3(256), 0, (hipStream_t)s, descs_dev, n, A),
                   hipLaunchKernelGGL((k_shadow_fused_batch<float, true>), dim3(tr_blocks), dim3(256), 0, (hipStream_t)s, descs_dev, n, A));
        PFA_LAUNCHED("k_adam_shadow");
    }
    if (ngaps > 0) {
        hipLaunchKernelGGL(k_adam_gaps, dim3(ngaps), dim3(256), 0, (hipStream_t)s, gaps_dev, p, g, g2, m, v, lr, beta1, beta2, eps,
                           step, step_dev, grad_scale);
        PFA_LAUNCHED("k_adam_gaps");
    }
    return 0;
}

extern "C" int pfa_pack_nhwc(const float* src, int n, int c, int h, int w, int cp, int dtype, void* dst, pfa_stream s) {
    PFA_REQUIRE(src && dst && n > 0 && c > 0 && cp >= c && h > 0 && w > 0, "pfa_pack_nhwc: bad arguments");
    const int64_t hw = (int64_t)h * w;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(k_pack_nhwc<bf16_t>, dim3(grid_for(n * hw)), dim3(256), 0, (hipStream_t)s, src, n, c,
                                  hw, cp, (bf16_t*)dst),
               hipLaunchKernelGGL(k_pack_nhwc<float>, dim3(grid_for(n * hw)), dim3(256), 0, (hipStream_t)s, src, n, c, hw,
                                  cp, (float*)dst));
    PFA_LAUNCHED("k_pack_nhwc");
    return 0;
}

extern "C" int pfa_unpack_nchw(const void* src, int n, int c, int h, int w, int cp, int dtype, float* dst,
                               pfa_stream s) {
    PFA_REQUIRE(src && dst && n > 0 && c > 0 && cp >= c && h > 0 && w > 0, "pfa_unpack_nchw: bad arguments");
    const int64_t hw = (int64_t)h * w;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(k_unpack_nchw<bf16_t>, dim3(grid_for(n * c * hw)), dim3(256), 0, (hipStream_t)s,
                                  (const bf16_t*)src, n, c, hw, cp, dst),
               hipLaunchKernelGGL(k_unpack_nchw<float>, dim3(grid_for(n * c * hw)), dim3(256), 0, (hipStream_t)s,
                                  (const float*)src, n, c, hw, cp, dst));
    PFA_LAUNCHED("k_unpack_nchw");
    return 0;
}

extern "C" int pfa_unpack_u8(const void* src, int64_t pixels, int c, int cp, int dtype, float mul, float add,
                             uint8_t* dst, pfa_stream s) {
    PFA_REQUIRE(src && dst && pixels > 0 && c > 0 && cp >= c, "pfa_unpack_u8: bad arguments");
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(k_unpack_u8<bf16_t>, dim3(grid_for(pixels * c)), dim3(256), 0, (hipStream_t)s,
                                  (const bf16_t*)src, pixels, c, cp, mul, add, dst),
               hipLaunchKernelGGL(k_unpack_u8<float>, dim3(grid_for(pixels * c)), dim3(256), 0, (hipStream_t)s,
                                  (const float*)src, pixels, c, cp, mul, add, dst));
    PFA_LAUNCHED("k_unpack_u8");
    return 0;
}

extern "C" int pfa_fill_zero(void* p, size_t bytes, pfa_stream s) {
    PFA_REQUIRE(p || bytes == 0, "pfa_fill_zero: null");
    if (bytes == 0) return 0;
    hipError_t e = hipMemsetAsync(p, 0, bytes, (hipStream_t)s);
    if (e != hipSuccess) PFA_FAIL((int)e, "pfa_fill_zero: %s", hipGetErrorString(e));
    return 0;
}

namespace {
__global__ void k_axpy(float* __restrict__ y, const float* __restrict__ x, int64_t n, float alpha) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] += alpha * x[i];
}
}  // namespace

extern "C" int pfa_axpy(float* y, const float* x, int64_t n, float alpha, pfa_stream s) {
    PFA_REQUIRE((y && x) || n == 0, "pfa_axpy: null");
    if (n <= 0) return 0;
    const int blocks = (int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
    hipLaunchKernelGGL(k_axpy, dim3(blocks), dim3(256), 0, (hipStream_t)s, y, x, n, alpha);
    PFA_LAUNCHED("k_axpy");
    return 0;
}
