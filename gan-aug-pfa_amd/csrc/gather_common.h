// Shared pieces of the implicit-GEMM family (gather_gemm.hip, conv_patch.hip): the kernel argument block, MFMA wrappers,
// LDS fragment reads the compiler cannot see through, and the EPILOGUE every kernel of the family ends with - K-slice
// partials, bias, the fused activation / BatchNorm backward (pfa_gather_args.fuse_out 1|2), the eval-mode affine map
// (fuse_out 3), BatchNorm partial sums and the NHWC stores (one or two outputs).
#pragma once
#include "pfa_common.h"

// Kernel argument block of the family (filled from pfa_gather_args by fill_params in gather_gemm.hip).
struct GatherK {
    const char* src1;
    const char* src2;
    const char* w;
    char* out1;
    char* out2;
    const float* bias;
    float* psum;
    float* psq;
    float* slab;
    int slab_ld;
    int mode, nb, hq, wq, hs, ws;
    int c1, c2, ctot;
    int cpt, cpt_shift, ktot, ksteps;
    int n, n1, ld1, ld2;
    int M, ho, wo, npix;
    int splitk, kper;
    int act;
    float slope;
    int mtiles, ntiles, nwg;
    int classes, order;      // parity classes (4 for PFA_TCONV_S2), work-list order (kernel comment)
    int wide, spt, sp1;      // wide path: K-steps per tap, K-steps of a tap that come from src1
    int dbg;                 // ablation bits (pfa_set_knob 1): 1 no output stores, 2 no stats, 4 no MFMA, 8 no DMA
    // fused activation / BatchNorm backward on one output (dgrad launches): dy = act1'(y)*acc + act2'(y)*g2
    int fuse_out, fuse_act1, fuse_act2;
    const char* fuse_raw;
    const char* fuse_g2;
    const float *fuse_scale, *fuse_shift, *fuse_mean, *fuse_invstd;
    int fuse_nreal;          // entries of the fuse_* coefficient arrays (the fused output's real channel count)
    int bias_unaligned;      // bias pointer not 16-byte aligned: element-wise loads
    int coef_unaligned;      // likewise for the fuse_* arrays
    int stat_col0, stat_n;   // statistics cover output columns [stat_col0, stat_col0 + stat_n), row length stat_n
    int small32;             // every operand and result tensor is below 2 GiB (32-bit byte offsets are safe)
};

// csrc/conv_patch2.hip: stride-2 convolution / transposed convolution with the input patch kept in LDS
struct Patch2Plan {
    int bn;                  // output channels per workgroup: 128 or 64
    int tile16;              // 1: 16 x 16 position tile (maps up to 16 wide), 0: 8 x 32
    int splitk;              // K slices actually used (<= the caller's)
    int tiles_y, tiles_x;    // position tiles per image
    int stat_tiles;          // rows of psum / psq an un-split launch writes
    int tconv4;              // 1: the four parity classes of a transposed conv in one workgroup (csrc/conv_tconv4.hip)
};
bool pfa_tconv4_ok(const GatherK& k, int dtype, int tiles_y, int tiles_x);
int pfa_tconv4(const GatherK& k, int tiles_y, int tiles_x, hipStream_t st, const char** kernel_name);
bool pfa_conv_patch2_plan(const GatherK& k, int dtype, Patch2Plan* plan);
int pfa_conv_patch2(const GatherK& k, const Patch2Plan& plan, hipStream_t st, const char** kernel_name);

// csrc/conv_patch.hip: the stride-1 patch kernel on MFMA 32x32x16 with the whole-row epilogue
bool pfa_conv_s1_plan(int grad, int nb, int hin, int win, int cin, int n, int splitk_caller, int* bn, int* splitk, int* stat_tiles);
int pfa_conv_s1_patch32(int grad, const GatherK& k, int bn, int hin, int win, int cin, hipStream_t st);

// csrc/conv_rows.hip: the implicit GEMM over im2col rows with the patch kernels' K loop (bf16, 64-multiple channel counts)
bool pfa_conv_rows_ok(const GatherK& k, int dtype);
int pfa_conv_rows(GatherK k, int classes, int (*order_of)(const GatherK&), hipStream_t st, const char** kernel_name);

// csrc/conv_image.hip: stride-2 convolution of the image-side layers (8-channel sources, 64 outputs)
bool pfa_conv_image_ok(const GatherK& k, int dtype, int* stat_tiles);
int pfa_conv_image(const GatherK& k, hipStream_t st, const char** kernel_name);

namespace {

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ f32x4 run(const frag& a, const frag& b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    typedef f32x4 frag;
    // 16 floats of K per (lane group, chunk): lane group kk supplies k = 4*kk + j to MFMA j for BOTH
    // operands, so products pair on the same k; the order of the k-sum is permuted, nothing else.
    static __device__ __forceinline__ f32x4 run(const frag& a, const frag& b, f32x4 c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
        return c;
    }
};

template <typename T> struct Vec4Of;
template <> struct Vec4Of<float> { typedef f32x4 type; };
template <> struct Vec4Of<bf16_t> { typedef bf16x4 type; };

template <typename T> __device__ __forceinline__ void store4(char* p, const float* v);
template <> __device__ __forceinline__ void store4<float>(char* p, const float* v) {
    f32x4 t = {v[0], v[1], v[2], v[3]};
    *(f32x4*)p = t;
}
template <> __device__ __forceinline__ void store4<bf16_t>(char* p, const float* v) {
    bf16x4 t = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    *(bf16x4*)p = t;
}


#define PFA_WAIT_LGKM(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory")

// LDS fragment read that hipcc cannot see through.  The compiler drains vmcnt before any LDS read that may alias an
// LDS-DMA in flight - with one dynamic LDS array that is every read - which would serialise the next tile's DMA behind
// this tile's reads.  Reads issued this way are retired by an explicit PFA_WAIT_LGKM followed by pin() on each
// destination (the empty asm makes every later use of the value depend on the wait).
template <typename F, int OFF> __device__ __forceinline__ F lds_read16(unsigned addr) {
    static_assert(sizeof(F) == 16 && OFF >= 0 && OFF < 65536, "fragment size / ds offset field");
    F v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// N fragments 16 tile rows (2048 bytes) apart from one address register
template <typename F, int N, int I = 0> struct ReadFrags {
    static __device__ __forceinline__ void run(F* dst, unsigned addr) {
        if constexpr (I < N) {
            dst[I] = lds_read16<F, I * 2048>(addr);
            ReadFrags<F, N, I + 1>::run(dst, addr);
        }
    }
};
template <typename F> __device__ __forceinline__ void pin(F& v) { asm volatile("" : "+v"(v)); }

// Sum of v over the 16 lanes of a DPP row, valid in the row's last lane (lane & 15 == 15): v += row_shr:1,2,4,8 with
// out-of-row lanes reading zero.  Fixed order, so the result is reproducible.
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    const int sh = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true);
    return v + __int_as_float(sh);
}
__device__ __forceinline__ float row_sum16(float v) {
    v = dpp_add<0x111>(v);      // row_shr:1
    v = dpp_add<0x112>(v);      // row_shr:2
    v = dpp_add<0x114>(v);      // row_shr:4
    v = dpp_add<0x118>(v);      // row_shr:8
    return v;
}

// EARLY == 0: K-step = fragment reads -> issue of the next tile's DMA -> MFMAs (the DMA overlaps the MFMAs only).
// EARLY == 1: K-step = issue of the next tile's DMA -> fragment reads (lds_read16) -> MFMAs: the DMA is in flight
//             for the whole step (cdna guide T3/T4 "minimum 2-phase": stage before the reads, one barrier per tile).
// EARLY == 2: as 1, with the DMA pieces issued between the MFMAs of the first half (see the loop).
template <typename T> __device__ __forceinline__ void load4(const char* p, float* v);
template <> __device__ __forceinline__ void load4<float>(const char* p, float* v) {
    const f32x4 t = *(const f32x4*)p;
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
template <> __device__ __forceinline__ void load4<bf16_t>(const char* p, float* v) {
    const bf16x4 t = *(const bf16x4*)p;
    v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
}


// Epilogue of one workgroup tile.  acc[a][b]: lane holds D[n = n0 + wn*FN*16 + a*16 + 4*kk + r][row b*16 + lrow of this wave's
// rows]; mval[b] / opix[b]: whether that row exists and its output pixel index; trow: the tile's row in psum/psq.
// Rows that do not exist must hold exact zeros in acc when statistics are requested without a fused backward (the
// implicit GEMM gets that from the zero page, the patch kernels clear them before the call).
template <typename T, int BN, int WM, int FN, int FM>
__device__ __forceinline__ void gather_epilogue(const GatherK& p, f32x4 (&acc)[FN][FM], const bool (&mval)[FM],
                                                const size_t (&opix)[FM], const int n0, const int wn, const int wm,
                                                const int tid, const int split, const size_t trow, char* smem) {
    constexpr int TN = FN * 16;
    const int lane = tid & 63, lrow = lane & 15, kk = lane >> 4;
    const float neg = act_neg(p.act, p.slope), neg1 = act_neg(p.fuse_act1, p.slope), neg2 = act_neg(p.fuse_act2, p.slope);

    if (p.splitk > 1) {
#ifdef PFA_RING_DIAG      // diagnostic build: knob 1 bit 1 drops the partial planes' stores
        if (p.dbg & 1) return;
#endif
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            const int nbase = n0 + wn * TN + a * 16 + 4 * kk;
            if (nbase < p.slab_ld) {
#pragma unroll
                for (int b = 0; b < FM; ++b)
                    if (mval[b]) *(f32x4*)(p.slab + ((size_t)split * p.npix + opix[b]) * p.slab_ld + nbase) = acc[a][b];
            }
        }
        return;
    }

    if (p.bias) {
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            const int nbase = n0 + wn * TN + a * 16 + 4 * kk;
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (nbase + 4 <= p.n && !(p.bias_unaligned)) {
                const f32x4 b4 = *(const f32x4*)(p.bias + nbase);
#pragma unroll
                for (int r = 0; r < 4; ++r) bv[r] = b4[r];
            } else {
                for (int r = 0; r < 4; ++r)
                    if (nbase + r < p.n) bv[r] = p.bias[nbase + r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int b = 0; b < FM; ++b) acc[a][b][r] += mval[b] ? bv[r] : 0.f;   // rows >= M stay exact zeros
            }
        }
    }

    float fs[FN][4], fq[FN][4];        // per-lane partial statistics (filled by the fused backward, else from acc)
    if (p.fuse_out == 1 || p.fuse_out == 2) {
        // This launch computes g = dL/d(act(y)) for the tensor `raw` (y = scale*raw + shift).  Turn it into
        // dy = act1'(y)*g (+ act2'(y)*g2) right here and collect the BatchNorm-backward sums (dy, dy*xhat) per column:
        // the separate elementwise pass over g, raw and dy disappears.
        // Phase 1: every raw / g2 vector this lane needs, loaded unconditionally (lanes or tiles without work read
        // element 0) so that all of them are in flight together; written with the loads inside `if (valid)` each one
        // became a separately branched, serialised load - the fused launches were 2x slower than un-fused ones.
        typedef typename Vec4Of<T>::type vec4_t;
        vec4_t xr[FN][FM], gr[FN][FM];
        const bool has_g2 = p.fuse_g2 != nullptr;
        const int fld = p.fuse_out == 1 ? p.ld1 : p.ld2;
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            const int nbase = n0 + wn * TN + a * 16 + 4 * kk;
            const bool mine = p.fuse_out == 1 ? nbase < p.n1 : nbase >= p.n1;
            const int col = p.fuse_out == 1 ? nbase : nbase - p.n1;
            const bool ok_a = mine && col < fld;
#pragma unroll
            for (int b = 0; b < FM; ++b) {
                const size_t e = (ok_a && mval[b]) ? opix[b] * fld + col : 0;
                xr[a][b] = *(const vec4_t*)(p.fuse_raw + ((p.dbg & 64) ? 0 : e) * sizeof(T));    // ablation bit 64: no traffic
            }
        }
        if (has_g2) {
#pragma unroll
            for (int a = 0; a < FN; ++a) {
                const int nbase = n0 + wn * TN + a * 16 + 4 * kk;
                const bool mine = p.fuse_out == 1 ? nbase < p.n1 : nbase >= p.n1;
                const int col = p.fuse_out == 1 ? nbase : nbase - p.n1;
                const bool ok_a = mine && col < fld;
#pragma unroll
                for (int b = 0; b < FM; ++b) {
                    const size_t e = (ok_a && mval[b]) ? opix[b] * fld + col : 0;
                    gr[a][b] = *(const vec4_t*)(p.fuse_g2 + ((p.dbg & 64) ? 0 : e) * sizeof(T));
                }
            }
        }
        // Phase 2: the arithmetic
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            const int nbase = n0 + wn * TN + a * 16 + 4 * kk;
            const bool mine = p.fuse_out == 1 ? nbase < p.n1 : nbase >= p.n1;
            const int col = p.fuse_out == 1 ? nbase : nbase - p.n1;
            const int ld = fld;
#pragma unroll
            for (int r = 0; r < 4; ++r) { fs[a][r] = 0.f; fq[a][r] = 0.f; }
            if (!mine || col >= ld) continue;
            // coefficients of 4 consecutive channels: uniform branches + one 16-byte load per array (element-wise
            // `ptr ? ptr[i] : k` compiles to a branched, serialised load per element).  The arrays hold the layer's real
            // channel count, which is a multiple of 4 wherever a norm layer exists; a ragged tail reads element-wise.
            float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f}, mu[4] = {0.f, 0.f, 0.f, 0.f}, is[4] = {1.f, 1.f, 1.f, 1.f};
            const bool whole = col + 4 <= p.fuse_nreal && !p.coef_unaligned;
            if (p.fuse_scale) {
                if (whole) {
                    const f32x4 a4 = *(const f32x4*)(p.fuse_scale + col), b4 = *(const f32x4*)(p.fuse_shift + col);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { sc[r] = a4[r]; sh[r] = b4[r]; }
                } else {
                    for (int r = 0; r < 4; ++r)
                        if (col + r < p.fuse_nreal) { sc[r] = p.fuse_scale[col + r]; sh[r] = p.fuse_shift[col + r]; }
                }
            }
            if (p.fuse_mean) {
                if (whole) {
                    const f32x4 a4 = *(const f32x4*)(p.fuse_mean + col), b4 = *(const f32x4*)(p.fuse_invstd + col);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { mu[r] = a4[r]; is[r] = b4[r]; }
                } else {
                    for (int r = 0; r < 4; ++r)
                        if (col + r < p.fuse_nreal) { mu[r] = p.fuse_mean[col + r]; is[r] = p.fuse_invstd[col + r]; }
                }
            }
#pragma unroll
            for (int b = 0; b < FM; ++b) {
                if (!mval[b]) continue;
                float x[4], g2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) x[r] = (float)xr[a][b][r];
                if (has_g2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) g2[r] = (float)gr[a][b][r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float y = x[r] * sc[r] + sh[r];
                    float d = act_grad_lin(y, neg1) * acc[a][b][r];
                    if (has_g2) d += act_grad_lin(y, neg2) * g2[r];
                    acc[a][b][r] = d;
                    fs[a][r] += d;
                    fq[a][r] += d * ((x[r] - mu[r]) * is[r]);
                }
            }
        }
    }

    if (p.fuse_out == 3) {
        // Eval-mode BatchNorm folded into the epilogue: y = scale[n]*acc + shift[n] (coefficients from the running
        // statistics, known before the launch); out1 = act(y) and, when given, out2 = fuse_act2(y) over the same columns.
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            const int nbase = n0 + wn * TN + a * 16 + 4 * kk;
            float sc[4] = {0.f, 0.f, 0.f, 0.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
            if (nbase + 4 <= p.n && !p.coef_unaligned) {   // one 16-byte load per array (see the note in the backward block)
                const f32x4 a4 = *(const f32x4*)(p.fuse_scale + nbase), b4 = *(const f32x4*)(p.fuse_shift + nbase);
#pragma unroll
                for (int r = 0; r < 4; ++r) { sc[r] = a4[r]; sh[r] = b4[r]; }
            } else {
                for (int r = 0; r < 4; ++r)
                    if (nbase + r < p.n) { sc[r] = p.fuse_scale[nbase + r]; sh[r] = p.fuse_shift[nbase + r]; }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int b = 0; b < FM; ++b) acc[a][b][r] = mval[b] ? acc[a][b][r] * sc[r] + sh[r] : 0.f;
            }
        }
    }

    if (p.psum && !(p.dbg & 2)) {
        // per-column sums over this tile's rows (rows >= M are exact zeros: their operands were the zero page,
        // and BatchNorm layers carry no bias)
        __syncthreads();
        float* red = (float*)smem;                // [2][WM][BN]
#pragma unroll
        for (int a = 0; a < FN; ++a) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = 0.f, q = 0.f;
                if (p.fuse_out == 1 || p.fuse_out == 2) { s = fs[a][r]; q = fq[a][r]; }
                else {
#pragma unroll
                    for (int b = 0; b < FM; ++b) { const float v = acc[a][b][r]; s += v; q += v * v; }
                }
                // sum over the 16 lanes of this DPP row (= the 16 tile rows a lane group holds): four row_shr adds, pure
                // VALU.  (__shfl_xor lowers to ds_bpermute - an LDS round trip per step, 128 per lane here: ~2 us per launch.)
                s = row_sum16(s);
                q = row_sum16(q);
                if (lrow == 15) {
                    const int nl = wn * TN + a * 16 + 4 * kk + r;
                    red[wm * BN + nl] = s;
                    red[(WM + wm) * BN + nl] = q;
                }
            }
        }
        __syncthreads();
        const int scol = n0 + tid - p.stat_col0;
        if (tid < BN && scol >= 0 && scol < p.stat_n) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int i = 0; i < WM; ++i) { s += red[i * BN + tid]; q += red[(WM + i) * BN + tid]; }
            p.psum[trow * p.stat_n + scol] = s;
            p.psq[trow * p.stat_n + scol] = q;
        }
    }

#pragma unroll
    for (int a = 0; a < FN; ++a) {
        const int nbase = n0 + wn * TN + a * 16 + 4 * kk;
        char* obase; int col, ld;
        if (nbase < p.n1) { obase = p.out1; col = nbase; ld = p.ld1; }
        else if (p.fuse_out == 3) continue;          // affine mode: out2 mirrors out1's columns, nothing lives beyond n1
        else { obase = p.out2; col = nbase - p.n1; ld = p.ld2; }
        if (col < ld && !(p.dbg & 1)) {
#pragma unroll
            for (int b = 0; b < FM; ++b) {
                if (!mval[b]) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float t = acc[a][b][r];
                    if (BN == 16 && p.act == PFA_ACT_TANH) v[r] = tanhf(t);          // only the 3-channel output layer
                    else v[r] = act_lin(t, neg);
                }
                store4<T>(obase + (opix[b] * ld + col) * sizeof(T), v);
                if (p.fuse_out == 3 && p.out2 && col < p.ld2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = act_lin(acc[a][b][r], neg2);
                    store4<T>(p.out2 + (opix[b] * p.ld2 + col) * sizeof(T), v);
                }
            }
        }
    }
}

}  // namespace
