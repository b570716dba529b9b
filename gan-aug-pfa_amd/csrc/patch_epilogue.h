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
#include "gather_common.h"

namespace {

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
                                                    const size_t trow, char* smem, PixOf pix_of) {
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
    //      stores.  (Activations here are LeakyReLU / ReLU / none: the route excludes tanh.)
    auto act_of = [&](float t, int act) { return act == PFA_ACT_LRELU ? (t >= 0.f ? t : t * p.slope) : act == PFA_ACT_RELU ? (t > 0.f ? t : 0.f) : t; };
    const bool has_g2 = fused && p.fuse_g2 != nullptr;
    float ss[8], sq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ss[e] = 0.f; sq[e] = 0.f; }
    const bool stats = p.psum != nullptr;
    constexpr int RB = 4;                        // rows per batch
    static_assert(NP % RB == 0, "row batches");
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
        if (fused) {
#pragma unroll
            for (int k = 0; k < RB; ++k) {
                const size_t e = valid[k] ? opix[k] * old + ocol : 0;     // (rows without work read element 0: no branch per load)
                xr[k] = *(const bf16x8*)(p.fuse_raw + e * sizeof(T));
            }
            if (has_g2) {
#pragma unroll
                for (int k = 0; k < RB; ++k) {
                    const size_t e = valid[k] ? opix[k] * old + ocol : 0;
                    gr[k] = *(const bf16x8*)(p.fuse_g2 + e * sizeof(T));
                }
            }
        }
#pragma unroll
        for (int k = 0; k < RB; ++k) {
            if (!valid[k]) continue;
            const char* src = lrd + (k0 + k) * RPP * RS;
            const f32x4 lo = *(const f32x4*)src, hi = *(const f32x4*)(src + 16);
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = lo[e] + bv[e]; v[4 + e] = hi[e] + bv[4 + e]; }
            bf16x8 o, o2;
            if (fused) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float x = (float)xr[k][e];
                    const float y = x * sc[e] + sh[e];
                    float d = act_grad(y, p.fuse_act1, p.slope) * v[e];
                    if (has_g2) d += act_grad(y, p.fuse_act2, p.slope) * (float)gr[k][e];
                    ss[e] += d;
                    sq[e] += d * ((x - mu[e]) * is[e]);
                    o[e] = (bf16_t)d;
                }
            } else if (p.fuse_out == 3) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float y = v[e] * sc[e] + sh[e];
                    o[e] = (bf16_t)act_of(y, p.act);
                    o2[e] = (bf16_t)act_of(y, p.fuse_act2);
                }
                if (p.out2 && col < p.ld2) *(bf16x8*)(p.out2 + (opix[k] * p.ld2 + col) * sizeof(T)) = o2;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    ss[e] += v[e];
                    sq[e] += v[e] * v[e];
                    o[e] = (bf16_t)act_of(v[e], p.act);
                }
            }
            *(bf16x8*)(obase + (opix[k] * old + ocol) * sizeof(T)) = o;
        }
    }

    // ---- partial statistics of this tile: lanes that share a chunk column inside a wave (lane % CPR), then the 8 waves
    if (!stats) return;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int off = CPR; off < 64; off <<= 1) {
            ss[e] += __shfl_xor(ss[e], off);
            sq[e] += __shfl_xor(sq[e], off);
        }
    }
    __syncthreads();                             // the fp32 image has been consumed
    float* red = (float*)smem;                   // [8 waves][2][BN]
    const int wave = tid >> 6;
    if (lane < CPR) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[(wave * 2 + 0) * BN + lane * 8 + e] = ss[e];
            red[(wave * 2 + 1) * BN + lane * 8 + e] = sq[e];
        }
    }
    __syncthreads();
    const int scol = n0 + tid - p.stat_col0;
    if (tid < BN && scol >= 0 && scol < p.stat_n) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) { s += red[(w * 2 + 0) * BN + tid]; q += red[(w * 2 + 1) * BN + tid]; }
        p.psum[trow * p.stat_n + scol] = s;
        p.psq[trow * p.stat_n + scol] = q;
    }
}

}  // namespace
