// Epilogue of the patch kernels: the workgroup's 256-position x BN-channel accumulator tile goes through LDS once and
// leaves the chip in whole NHWC rows.
//
// The MFMA accumulator layout gives a lane 4 consecutive channels of one pixel: stored (or, for the fused backward,
// loaded) straight from registers that is 8 bytes per lane and 32-byte pieces of 16 different rows per wave instruction -
// the K-loop's result left at ~2 TB/s, and the fused activation / BatchNorm-backward epilogue (two more strided operand
// streams) cost a launch 20-30 us.  Here every wave first drops its accumulators into an fp32 [position][channel] image
// in LDS (row stride BN*4+16 bytes: conflict-free 16-byte writes); after one barrier, thread t owns the 16-byte output chunk
// (8 channels) t % (BN/8) of rows t / (BN/8) + k * (512 / (BN/8)): its channels are FIXED, so per-channel coefficients are
// loaded once, every global access is a 16-byte piece of a 2*BN-byte contiguous row segment (a wave covers whole rows),
// and the BatchNorm partial sums accumulate in registers across the thread's rows.
//
// Same semantics as gather_epilogue (gather_common.h): K-slice partials, bias, fused backward (fuse_out 1|2), eval-mode
// affine map with two activation outputs (fuse_out 3), partial statistics, activation, one or two outputs.
#pragma once
#include <type_traits>

#include "gather_common.h"

namespace {

typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ void load8f(const float* p, bool vec, int limit, float* v) {     // v[e] = p[e] for e < limit
    if (vec) {
        const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
    } else {
        for (int e = 0; e < 8; ++e)
            if (e < limit) v[e] = p[e];
    }
}

// LDS row stride of the fp32 [position][channel] image a kernel writes before calling patch_epilogue_rows
template <int BN> constexpr int patch_image_stride() { return BN * 4 + 16; }

// pix_of(r, valid, opix): tile-local position r (0..255, the LDS row) -> does it exist, its output pixel index.
// The caller has written its accumulators to smem as float32 [256 positions][BN channels] with row stride
// patch_image_stride<BN>() and passed a barrier.
template <int BN, typename PixOf>
__device__ __forceinline__ void patch_epilogue_rows(const GatherK& p, const int n0, const int tid, const int split,
                                                    const size_t trow, char* smem, PixOf pix_of,
                                                    unsigned long long* stamps = nullptr) {
    constexpr int RS = patch_image_stride<BN>();
    constexpr int CPR = BN / 8;                  // 16-byte output chunks per row
    constexpr int RPP = 512 / CPR;               // rows per pass
    constexpr int NP = 256 / RPP;                // passes: 8 (BN 128) or 4 (BN 64)
    typedef bf16_t T;
    const int lane = tid & 63;

    const int cc = tid % CPR, rsub = tid / CPR;
    const int col = n0 + cc * 8;                 // first of this thread's 8 output columns
    const char* lrd = smem + rsub * RS + cc * 32;

    // ---- K-sliced launch: float32 partial rows for the finish pass
    if (p.splitk > 1) {
        if (col < p.slab_ld) {
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                bool valid; size_t opix;
                pix_of(rsub + k * RPP, valid, opix);
                if (!valid) continue;
                float* dst = p.slab + ((size_t)split * p.npix + opix) * p.slab_ld + col;
                *(f32x4*)dst = *(const f32x4*)(lrd + k * RPP * RS);
                *(f32x4*)(dst + 4) = *(const f32x4*)(lrd + k * RPP * RS + 16);
            }
        }
        return;
    }

    // ---- which output these columns go to
    const bool first = col < p.n1;
    char* obase = first ? p.out1 : p.out2;
    const int ocol = first ? col : col - p.n1;
    const int old = first ? p.ld1 : p.ld2;
    const bool live = col < p.n && ocol < old;   // (n is a multiple of 64 on this route: whole chunks)
    const bool fused = (p.fuse_out == 1 && first) || (p.fuse_out == 2 && !first);

    float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p.bias && live) load8f(p.bias + col, !p.bias_unaligned, p.n - col, bv);
    float sc[8], sh[8], mu[8], is[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; mu[e] = 0.f; is[e] = 1.f; }
    if (fused && live) {
        const bool vec = ocol + 8 <= p.fuse_nreal && !p.coef_unaligned;
        if (p.fuse_scale) { load8f(p.fuse_scale + ocol, vec, p.fuse_nreal - ocol, sc); load8f(p.fuse_shift + ocol, vec, p.fuse_nreal - ocol, sh); }
        if (p.fuse_mean) { load8f(p.fuse_mean + ocol, vec, p.fuse_nreal - ocol, mu); load8f(p.fuse_invstd + ocol, vec, p.fuse_nreal - ocol, is); }
    } else if (p.fuse_out == 3 && live) {
        const bool vec = col + 8 <= p.n && !p.coef_unaligned;
        load8f(p.fuse_scale + col, vec, p.n - col, sc);
        load8f(p.fuse_shift + col, vec, p.n - col, sh);
    }

    // ---- the rows, four at a time: operand loads of the fused backward first (in flight together), then arithmetic and
    //      stores.  (Activations here are LeakyReLU / ReLU / none: the route excludes tanh.)  The three kinds of row work
    //      are compiled as three loops: per-element decisions on launch-wide flags would be scalar branches per element.
    const float neg = act_neg(p.act, p.slope), neg1 = act_neg(p.fuse_act1, p.slope), neg2 = act_neg(p.fuse_act2, p.slope);
    const bool has_g2 = fused && p.fuse_g2 != nullptr;
    float ss[8], sq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ss[e] = 0.f; sq[e] = 0.f; }
    const bool stats = p.psum != nullptr;
    constexpr int RB = 4;                        // rows per batch
    static_assert(NP % RB == 0, "row batches");
    auto rows = [&](auto kind_, auto g2_) {      // KIND 0: bias, statistics, activation; 1: fused backward; 2: affine map, two outputs
        constexpr int KIND = decltype(kind_)::value;
        constexpr bool G2 = KIND == 1 && decltype(g2_)::value;
        constexpr bool PLAIN = KIND == 0 && decltype(g2_)::value;     // KIND 0 without bias and activation (a BatchNorm layer's conv)
#pragma unroll
        for (int k0 = 0; k0 < NP; k0 += RB) {
            bool valid[RB];
            size_t opix[RB];
#pragma unroll
            for (int k = 0; k < RB; ++k) {
                pix_of(rsub + (k0 + k) * RPP, valid[k], opix[k]);
                valid[k] = valid[k] && live;
            }
            bf16x8 xr[RB], gr[RB];
            if (KIND == 1) {
#pragma unroll
                for (int k = 0; k < RB; ++k) {
                    const size_t e = valid[k] ? opix[k] * old + ocol : 0;     // (rows without work read element 0: no branch per load)
                    xr[k] = *(const bf16x8*)(p.fuse_raw + e * sizeof(T));
                    if (G2) gr[k] = *(const bf16x8*)(p.fuse_g2 + e * sizeof(T));
                }
            }
#pragma unroll
            for (int k = 0; k < RB; ++k) {
                if (!valid[k]) continue;
                const char* src = lrd + (k0 + k) * RPP * RS;
                const f32x4 lo = *(const f32x4*)src, hi = *(const f32x4*)(src + 16);
                float v[8];
                if (KIND == 0 && PLAIN) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = lo[e] + bv[e]; v[4 + e] = hi[e] + bv[4 + e]; }
                }
                bf16x8 o;
                if (KIND == 1) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float x = (float)xr[k][e];
                        const float y = x * sc[e] + sh[e];
                        float d = act_grad_lin(y, neg1) * v[e];
                        if (G2) d += act_grad_lin(y, neg2) * (float)gr[k][e];
                        ss[e] += d;
                        sq[e] += d * ((x - mu[e]) * is[e]);
                        o[e] = (bf16_t)d;
                    }
                } else if (KIND == 2) {
                    bf16x8 o2;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float y = v[e] * sc[e] + sh[e];
                        o[e] = (bf16_t)act_lin(y, neg);
                        o2[e] = (bf16_t)act_lin(y, neg2);
                    }
                    if (p.out2 && col < p.ld2) *(bf16x8*)(p.out2 + (opix[k] * p.ld2 + col) * sizeof(T)) = o2;
                } else {
                    // (pairs: v_pk_add_f32 / v_pk_fma_f32 - this loop is VALU-bound, 8 elements x 8 rows a thread)
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const f32x2 v2 = {v[e], v[e + 1]};
                        f32x2 s2 = {ss[e], ss[e + 1]}, q2 = {sq[e], sq[e + 1]};
                        s2 += v2;
                        q2 = __builtin_elementwise_fma(v2, v2, q2);
                        ss[e] = s2[0]; ss[e + 1] = s2[1]; sq[e] = q2[0]; sq[e + 1] = q2[1];
                    }
                    if (PLAIN) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)act_lin(v[e], neg);
                    }
                }
                *(bf16x8*)(obase + (opix[k] * old + ocol) * sizeof(T)) = o;
            }
        }
    };
    if (fused) {
        if (has_g2) rows(std::integral_constant<int, 1>{}, std::true_type{});
        else rows(std::integral_constant<int, 1>{}, std::false_type{});
    } else if (p.fuse_out == 3) {
        rows(std::integral_constant<int, 2>{}, std::false_type{});
    } else if (p.bias == nullptr && neg == 1.f) {
        rows(std::integral_constant<int, 0>{}, std::true_type{});
    } else {
        rows(std::integral_constant<int, 0>{}, std::false_type{});
    }

    // ---- partial statistics of this tile.  Every thread holds the sums of its 8 channels over its rows; the 512 / CPR
    // threads that share a chunk column meet through LDS: plain 16-byte writes, one barrier, then 2 BN threads add one column
    // each (consecutive lanes, consecutive floats).  (Butterfly shuffles - 32-48 ds_bpermute with their waits - took 1.4-2 k
    // cycles here, and the two-stage reduce behind them as much again.)
    if (stamps && tid == 0) stamps[4] = __builtin_amdgcn_s_memtime();       // (developer timing: rows issued)
    if (!stats) return;
    constexpr int RG = 512 / CPR;                // row groups: threads per chunk column
    __syncthreads();                             // the fp32 image has been consumed
    float* red = (float*)smem;                   // [RG][2][BN]
    {
        float* dst = red + (size_t)rsub * 2 * BN + cc * 8;
        *(f32x4*)dst = f32x4{ss[0], ss[1], ss[2], ss[3]};
        *(f32x4*)(dst + 4) = f32x4{ss[4], ss[5], ss[6], ss[7]};
        *(f32x4*)(dst + BN) = f32x4{sq[0], sq[1], sq[2], sq[3]};
        *(f32x4*)(dst + BN + 4) = f32x4{sq[4], sq[5], sq[6], sq[7]};
    }
    if (stamps && tid == 0) stamps[5] = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (tid < 2 * BN) {
        const int stat = tid / BN, c = tid - stat * BN;
        const int scol = n0 + c - p.stat_col0;
        const float* src = red + stat * BN + c;
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
        for (int g = 0; g < RG; g += 4) {
            t0 += src[(size_t)(g + 0) * 2 * BN];
            t1 += src[(size_t)(g + 1) * 2 * BN];
            t2 += src[(size_t)(g + 2) * 2 * BN];
            t3 += src[(size_t)(g + 3) * 2 * BN];
        }
        if (scol >= 0 && scol < p.stat_n) (stat ? p.psq : p.psum)[trow * p.stat_n + scol] = (t0 + t1) + (t2 + t3);
    }
}

}  // namespace
