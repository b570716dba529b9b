// Weight-gradient implicit GEMM on MFMA.
//
// Replaces the weight-gradient half of autograd's backward for nn.Conv2d (models.py:177,223,230,238,243)
// and nn.ConvTranspose2d (models.py:184,189,194) as executed by train_gan.py:62,70.
//
//   dw[i][tap][c] += sum_m direct[m][i] * gathered[pix(m) + tap][c]
//
// Conv2d:          direct = dOut  (i = Co), gathered = input (c = Ci)            -> dw[Co][16][Ci]
// ConvTranspose2d: direct = input (i = Ci, incl. virtual concat), gathered = dOut -> dw[Ci][16][Co]
// both with gathered pixel (qy*s-1+ky, qx*s-1+kx): the result lands directly in the memory layout of the
// float32 master parameter (channels-last OIHW / IOHW).
//
// GEMM view: rows i (direct channels), columns j = flattened (tap, gathered channel), reduction over the
// pixels m.  Both operands are stored pixel-major (NHWC), i.e. the contraction index is the SLOW axis of
// both tiles; the tiles are staged pixel-major into LDS by LDS-DMA and handed to the MFMA through the
// hardware transpose read ds_read_b64_tr_b16 (bf16) / plain ds_read_b32 (f32, one value per lane).
// The reduction is sliced over blockIdx.z; slices combine with float32 atomics (dw is pre-zeroed).
#include "pfa_common.h"
#include <atomic>

constexpr int WGRAD_DEFAULT_EARLY = 0;

namespace {

struct WgradK {
    const char* dir1;
    const char* dir2;
    const char* gat1;
    const char* gat2;
    float* dw;
    int stride, nb, hq, wq, hg, wg;
    int d1p, d1r, d2p, d2r, dp;
    int g1p, g1r, g2p, g2r, gp, gr;
    int M, mper;
    unsigned magic_hw, magic_wq;
    int step_img, step_row, step_col;   // KM pixels = step_img images + step_row rows + step_col columns
    int jtot;        // 16 * gp
    int jtiles, itiles, nwg;
    int dbg;         // ablation bits (pfa_set_knob 1): 1 no atomics, 4 no MFMA, 8 no DMA
    int early;       // K-loop order (pfa_set_knob 4: 1 = reads first, 2 = DMA first, 3 = 256-row ring tile where it applies; 0 = default)
    float* slab;     // optional [msplit][tiles][BI*128] partial tiles (deterministic combine instead of atomics)
    int kgroups;     // 1, or 2: K-group pairs (k_wgrad KG)
    int msplit;
    int fresh;       // dw holds zeros and this launch has one slice: store, do not add
    // row-aligned geometry (k_wgrad<..., FAST>): a reduction step covers a piece of ONE row, whole rows or whole images, so a
    // lane's place inside the step never changes and the step's position is one scalar
    int rps;         // rows of the map a step covers (1 when it covers a piece of a row)
    int xspan;       // columns of the map a step covers: min(KM, wq)
};

__device__ __forceinline__ int fastdiv(int n, int d, unsigned magic) {
    int q = (int)__umulhi((unsigned)n, magic);
    int r = n - q * d;
    if (r >= d) ++q;      // magic = floor(2^32/d): q is exact or one short for n < 2^31
    return q;
}

// Transpose read that hipcc cannot see through: the compiler drains vmcnt before any LDS read that may alias an LDS-DMA in
// flight, which would serialise the next tile's DMA behind this tile's reads (same device as gather_gemm.hip's lds_read16).
// The caller retires these with an explicit s_waitcnt lgkmcnt + pin().
__device__ __forceinline__ bf16x4 lds_read_tr(unsigned addr) {
    bf16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
template <typename F> __device__ __forceinline__ void pin(F& v) { asm volatile("" : "+v"(v)); }

// KG == 2 (bf16): two wave groups share one output tile and split every reduction step between them (KM = 128 pixels per
// step, group g takes rows 64g..64g+63); their accumulators are added through LDS at the end.  Same waves per CU as two
// independent 4-wave workgroups, half the partial tiles to write and reduce.
// FAST (bf16, row-aligned geometry: wgrad_fast_ok): both operands staged with the buffer form of the LDS-DMA (pfa_common.h) - a
// per-lane offset fixed for the whole loop, one scalar per operand and step, PFA_DMA_OOB for the lanes whose gathered pixel
// falls outside the image at the edge the step touches.  The general staging below rebuilds a 64-bit pointer per piece
// from per-lane pixel coordinates it advances with carries: 329 instructions per step for 32 MFMAs, half of them that
// arithmetic (tools/issue_mix.py).
template <typename T, int BI, int WI, int WJ, int EARLY, int KG = 1, bool FAST = false>
__global__ __launch_bounds__(64 * WI * WJ * KG) __attribute__((amdgpu_waves_per_eu(2))) void k_wgrad(const WgradK p) {
    constexpr int NTO = 64 * WI * WJ;                // threads that own the output tile
    constexpr int NT = NTO * KG;                     // 256 threads (4 waves) or 512 (8 waves: the 256-row ring tile, KG = 2)
    constexpr int CH = ElemTraits<T>::CH;
    constexpr bool BF = sizeof(T) == 2;
    constexpr int BJ = 128;
    constexpr int KM = (BF ? 64 : 32) * KG;          // pixels per reduction step
    static_assert(KG == 1 || (KG == 2 && BF && EARLY != 3), "K groups: bf16, two-stage loops");
    constexpr int TI = BI / WI, TJ = BJ / WJ, FI = TI / 16, FJ = TJ / 16;
    constexpr int DROWB = BI * (int)sizeof(T), GROWB = BJ * (int)sizeof(T);
    constexpr int RCD = DROWB / 16, RCG = GROWB / 16;   // chunks per tile row
    constexpr int D_BYTES = KM * DROWB, G_BYTES = KM * GROWB, STAGE = D_BYTES + G_BYTES;
    constexpr int DSLOTS = KM * RCD, GSLOTS = KM * RCG;
    constexpr int DIT = (DSLOTS + NT - 1) / NT, GIT = GSLOTS / NT;
    static_assert((NT == 256 || NT == 512) && GSLOTS % NT == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave / (WI * WJ), wv = wave % (WI * WJ);      // K group, wave inside the group
    const int wi = wv % WI, wj = wv / WI;
    // work list: j-tile fastest (neighbours share the direct tile and overlapping gathered pixels), then i-tile,
    // then reduction slice; each XCD takes one contiguous chunk of it
    unsigned wid = xcd_remap(blockIdx.x, p.nwg);
    const int jt = wid % p.jtiles; wid /= p.jtiles;
    const int it = wid % p.itiles; wid /= p.itiles;
    const int j0 = jt * BJ, i0 = it * BI;
    const int mbeg = (int)wid * p.mper;
    const int mend = min(p.M, mbeg + p.mper);
    const char* zero = (const char*)g_pfa_zero_page;

    // swizzle of the chunk index inside a tile row (applied on the DMA source and on the reads).  A transpose read
    // touches, per 32-lane half, rows {r..r+3, r+8..r+11} x 32 B: row bits 0,1,3 must select distinct 32-byte
    // segments of the 256-byte bank row (bit 2 only moves inside the segment).
    auto swz = [](int row, int rc) -> int {
        if (rc < 16) return 0;
        return BF ? (((row & 3) << 2) | ((row >> 2) & 3)) : ((row & 1) << 2);
    };

    // ---- direct operand: this thread's chunk column is fixed for the whole kernel
    const int d_cs = tid % RCD, d_r0 = tid / RCD;
    const int d_g = d_cs ^ swz(d_r0, RCD);
    const char* d_base = nullptr;
    int d_ld = 0;
    {
        const int ic = i0 + d_g * CH;
        if (ic < p.d1p) { d_base = p.dir1 + (size_t)ic * sizeof(T); d_ld = p.d1p; }
        else if (ic < p.dp) { d_base = p.dir2 + (size_t)(ic - p.d1p) * sizeof(T); d_ld = p.d2p; }
    }
    const bool d_active = (DSLOTS >= NT) || (tid < DSLOTS);
    // ---- gathered operand: chunk column -> (tap, channel) fixed per thread
    const int g_cs = tid % RCG, g_r0 = tid / RCG;
    const int g_g = g_cs ^ swz(g_r0, RCG);
    const char* g_base = nullptr;
    int g_ld = 0, g_ky = 0, g_kx = 0;
    {
        const int jc = j0 + g_g * CH;
        if (jc < p.jtot) {
            const int tap = jc / p.gp, c = jc - tap * p.gp;
            g_ky = tap >> 2; g_kx = tap & 3;
            if (c < p.g1p) { g_base = p.gat1 + (size_t)c * sizeof(T); g_ld = p.g1p; }
            else { g_base = p.gat2 + (size_t)(c - p.g1p) * sizeof(T); g_ld = p.g2p; }
        }
    }
    const int hw = p.hq * p.wq;
    // pixel coordinates of this thread's gathered rows, advanced incrementally by KM pixels per step
    int g_ni[GIT], g_qy[GIT], g_qx[GIT];
#pragma unroll
    for (int i = 0; i < GIT; ++i) {
        const int mg = mbeg + i * (NT / RCG) + g_r0;
        g_ni[i] = fastdiv(mg, hw, p.magic_hw);
        const int rem = mg - g_ni[i] * hw;
        g_qy[i] = fastdiv(rem, p.wq, p.magic_wq);
        g_qx[i] = rem - g_qy[i] * p.wq;
    }

    // ---- FAST: per-lane offsets and edge bits of every piece, fixed for the loop
    unsigned fd_off[FAST ? DIT : 1], fg_off[FAST ? GIT : 1], fg_bits[FAST ? GIT : 1];
    const char* fd_base = nullptr;          // (wave-uniform) the direct tile's first column
    const char* fg_base = nullptr;          // the gathered tensor, moved back by one row and one pixel: no offset is negative
    int fd_ld = 0;
    if constexpr (FAST) {
        const bool first = i0 < p.d1p;      // (the tile lies inside one source: wgrad_fast_ok)
        fd_base = first ? p.dir1 + (size_t)i0 * sizeof(T) : p.dir2 + (size_t)(i0 - p.d1p) * sizeof(T);
        fd_ld = first ? p.d1p : p.d2p;
        const bool d_ok = i0 + d_g * CH < p.dp;
#pragma unroll
        for (int i = 0; i < DIT; ++i) {
            const int pi = i * (NT / RCD) + d_r0;
            fd_off[i] = d_ok ? (unsigned)((pi * fd_ld + d_g * CH) * (int)sizeof(T)) : PFA_DMA_OOB;
        }
        fg_base = p.gat1 - (size_t)(p.wg + 1) * p.g1p * sizeof(T);
        const int jc = j0 + g_g * CH;
        const bool g_ok = jc < p.jtot;
        const int tap = g_ok ? jc / p.gp : 0, c = jc - tap * p.gp;
        const int ky = tap >> 2, kx = tap & 3;
#pragma unroll
        for (int i = 0; i < GIT; ++i) {
            const int pi = i * (NT / RCG) + g_r0;                    // pixel inside the step
            const int img = pi / hw, rem = pi - img * hw;            // (0, pi while a step stays inside one image)
            const int ry = rem / p.wq, qx = rem - ry * p.wq;
            const int pix = (img * p.hg + p.stride * ry + ky - 1) * p.wg + p.stride * qx + kx - 1 + p.wg + 1;
            fg_off[i] = g_ok ? (unsigned)((pix * p.g1p + c) * (int)sizeof(T)) : PFA_DMA_OOB;
            // the lane reads outside the image when the step touches: 1 the left edge, 2 the right, 4 the top, 8 the bottom
            fg_bits[i] = (kx == 0 && qx == 0 ? 1u : 0u) | (kx == 3 && qx == p.xspan - 1 ? 2u : 0u) |
                         (ky == 0 && ry == 0 ? 4u : 0u) | (ky == 3 && ry == p.rps - 1 ? 8u : 0u);
        }
    }
    auto stage = [&](int mstep, int buf) {
        if (p.dbg & 8) return;
        char* dbase = smem + buf * STAGE;
        char* gbase = dbase + D_BYTES;
        if constexpr (FAST) {
            // the step's place: image, row, column of its first pixel (scalars) -> one offset per operand, the edges it touches
            const int ms = __builtin_amdgcn_readfirstlane(mstep);
            const int ni = fastdiv(ms, hw, p.magic_hw), r = ms - ni * hw;
            const int qy = fastdiv(r, p.wq, p.magic_wq), qx = r - qy * p.wq;
            const unsigned edge = (qx == 0 ? 1u : 0u) | (qx + p.xspan == p.wq ? 2u : 0u) | (qy == 0 ? 4u : 0u) |
                                  (qy + p.rps >= p.hq ? 8u : 0u);
            const unsigned dso = (unsigned)ms * (unsigned)fd_ld * (unsigned)sizeof(T);
            const unsigned gso = (unsigned)(((ni * p.hg + p.stride * qy) * p.wg + p.stride * qx) * p.g1p) * (unsigned)sizeof(T);
#pragma unroll
            for (int i = 0; i < DIT; ++i) buf_dma16(fd_base, dbase + (i * NT + wave * 64) * 16, fd_off[i], dso);
#pragma unroll
            for (int i = 0; i < GIT; ++i)
                buf_dma16(fg_base, gbase + (i * NT + wave * 64) * 16, (fg_bits[i] & edge) ? PFA_DMA_OOB : fg_off[i], gso);
            return;
        }
        if (d_active) {
#pragma unroll
            for (int i = 0; i < DIT; ++i) {
                const int mg = mstep + i * (NT / RCD) + d_r0;
                const bool ok = d_base && mg < mend;
                const char* gp = ok ? d_base + (size_t)mg * d_ld * sizeof(T) : zero;
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gp), LDS_PTR(dbase + (i * NT + wave * 64) * 16), 16, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < GIT; ++i) {
            const int mg = mstep + i * (NT / RCG) + g_r0;
            const int y = g_qy[i] * p.stride - 1 + g_ky, x = g_qx[i] * p.stride - 1 + g_kx;
            const bool ok = (g_base != nullptr) & (mg < mend) & ((unsigned)y < (unsigned)p.hg) & ((unsigned)x < (unsigned)p.wg);
            const char* gp = ok ? g_base + ((size_t)(g_ni[i] * p.hg + y) * p.wg + x) * g_ld * sizeof(T) : zero;
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gp), LDS_PTR(gbase + (i * NT + wave * 64) * 16), 16, 0, 0);
            // advance by KM pixels: at most one carry per level (step_col < wq, step_row < hq)
            g_qx[i] += p.step_col;
            if (g_qx[i] >= p.wq) { g_qx[i] -= p.wq; ++g_qy[i]; }
            g_qy[i] += p.step_row;
            if (g_qy[i] >= p.hq) { g_qy[i] -= p.hq; ++g_ni[i]; }
            g_ni[i] += p.step_img;
        }
    };

    f32x4 acc[FI][FJ];
#pragma unroll
    for (int a = 0; a < FI; ++a)
#pragma unroll
        for (int b = 0; b < FJ; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int l15 = lane & 15, kk = lane >> 4;
    bf16x8 af[2][FI], bfr[2][FJ];           // bf16 path: the fragments of one reduction step
    // Phase 1 of a step: every LDS (transpose) read.  hipcc drains vmcnt before LDS reads that may alias an
    // LDS-DMA in flight, so the NEXT tile's DMA is issued only after these reads (phase 2) and then runs under
    // the MFMAs (phase 3) instead of in front of them.
    const unsigned lds0 = (unsigned)(size_t)LDS_PTR(smem);
    auto read_frags = [&](int buf, bool raw) {
        if (p.dbg & 4) return;
        const char* dbase = smem + buf * STAGE;
        const char* gbase = dbase + D_BYTES;
        const unsigned dbase_u = lds0 + buf * STAGE, gbase_u = dbase_u + D_BYTES;
#pragma unroll
        for (int k32 = 0; k32 < 2; ++k32) {
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
                // 4 pixels x 16 channels block per 16-lane group, delivered channel-major
                const int m = 64 * kg + 32 * k32 + 8 * kk + 4 * e2 + (l15 >> 2);
#pragma unroll
                for (int a = 0; a < FI; ++a) {
                    const int col = wi * TI + a * 16 + 4 * (lane & 3);
                    const int off = m * DROWB + (((col >> 3) ^ swz(m, RCD)) << 4) + ((col >> 2) & 1) * 8;
                    bf16x4 t = raw ? lds_read_tr(dbase_u + off)
                                   : __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                                         (__attribute__((address_space(3))) bf16x4*)(dbase + off));
#pragma unroll
                    for (int e = 0; e < 4; ++e) af[k32][a][4 * e2 + e] = t[e];
                }
#pragma unroll
                for (int b = 0; b < FJ; ++b) {
                    const int col = wj * TJ + b * 16 + 4 * (lane & 3);
                    const int off = m * GROWB + (((col >> 3) ^ swz(m, RCG)) << 4) + ((col >> 2) & 1) * 8;
                    bf16x4 t = raw ? lds_read_tr(gbase_u + off)
                                   : __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                                         (__attribute__((address_space(3))) bf16x4*)(gbase + off));
#pragma unroll
                    for (int e = 0; e < 4; ++e) bfr[k32][b][4 * e2 + e] = t[e];
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mfma_frags = [&]() {
        if (p.dbg & 4) return;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k32 = 0; k32 < 2; ++k32) {
#pragma unroll
            for (int a = 0; a < FI; ++a)
#pragma unroll
                for (int b = 0; b < FJ; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[k32][a], bfr[k32][b], acc[a][b], 0, 0, 0);
        }
    };
    auto compute_f32 = [&](int buf) {
        if (p.dbg & 4) return;
        const char* dbase = smem + buf * STAGE;
        const char* gbase = dbase + D_BYTES;
#pragma unroll
        for (int k4 = 0; k4 < KM / 4; ++k4) {
            const int m = 4 * k4 + kk;
            float a32[FI], b32[FJ];
#pragma unroll
            for (int a = 0; a < FI; ++a) {
                const int col = wi * TI + a * 16 + l15;
                a32[a] = *(const float*)(dbase + m * DROWB + (((col >> 2) ^ swz(m, RCD)) << 4) + (col & 3) * 4);
            }
#pragma unroll
            for (int b = 0; b < FJ; ++b) {
                const int col = wj * TJ + b * 16 + l15;
                b32[b] = *(const float*)(gbase + m * GROWB + (((col >> 2) ^ swz(m, RCG)) << 4) + (col & 3) * 4);
            }
#pragma unroll
            for (int a = 0; a < FI; ++a)
#pragma unroll
                for (int b = 0; b < FJ; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a32[a], b32[b], acc[a][b], 0, 0, 0);
        }
    };

    if constexpr (EARLY == 3) {
        // Three LDS stages, DMA queue never drained (same scheme as gather_gemm.hip's EARLY == 3): step i waits with a
        // counted vmcnt for the pieces of stage i, leaves stage i+1's in flight across the barrier and issues stage
        // i+2's.  Every wave issues exactly DIT + GIT instructions per stage (the address select never skips one).
        static_assert(BF && DSLOTS >= NT, "ring variant: bf16, every thread stages direct rows");
        constexpr int NPIECE = DIT + GIT;
        const int nsteps = (mend - mbeg + KM - 1) / KM;
        if (nsteps > 0) stage(mbeg, 0);
        if (nsteps > 1) stage(mbeg + KM, 1);
        for (int i = 0; i < nsteps; ++i) {
            const int rb = i % 3;
            if (i + 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (i + 2 < nsteps) stage(mbeg + (i + 2) * KM, (i + 2) % 3);
            __builtin_amdgcn_sched_barrier(0);
            read_frags(rb, true);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k32 = 0; k32 < 2; ++k32) {
#pragma unroll
                for (int a = 0; a < FI; ++a) pin(af[k32][a]);
#pragma unroll
                for (int b = 0; b < FJ; ++b) pin(bfr[k32][b]);
            }
            mfma_frags();
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
    if (mbeg < mend) stage(mbeg, 0);
    int buf = 0;
    for (int ms = mbeg; ms < mend; ms += KM) {
        __syncthreads();              // tile `ms` has landed; everyone is done with the other buffer
        if constexpr (BF) {
            if constexpr (EARLY == 1) {
                // DMA first: the next tile is in flight under both the transpose reads and the MFMAs
                if (ms + KM < mend) stage(ms + KM, buf ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                read_frags(buf, true);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int k32 = 0; k32 < 2; ++k32) {
#pragma unroll
                    for (int a = 0; a < FI; ++a) pin(af[k32][a]);
#pragma unroll
                    for (int b = 0; b < FJ; ++b) pin(bfr[k32][b]);
                }
                mfma_frags();
            } else {
                read_frags(buf, false);
                if (ms + KM < mend) stage(ms + KM, buf ^ 1);
                mfma_frags();
            }
        } else {
            if (ms + KM < mend) stage(ms + KM, buf ^ 1);
            compute_f32(buf);
        }
        buf ^= 1;
    }
    }   // EARLY != 3

    if constexpr (KG == 2) {
        // group 1 hands its accumulators to group 0 through LDS (BI*BJ floats = 64 KiB of the 128 KiB staging area)
        __syncthreads();
        f32x4* xch = (f32x4*)smem;
        const int t = tid & (NTO - 1);
        if (kg == 1) {
#pragma unroll
            for (int a = 0; a < FI; ++a)
#pragma unroll
                for (int b = 0; b < FJ; ++b) xch[(a * FJ + b) * NTO + t] = acc[a][b];
        }
        __syncthreads();
        if (kg == 1) return;
#pragma unroll
        for (int a = 0; a < FI; ++a)
#pragma unroll
            for (int b = 0; b < FJ; ++b) acc[a][b] += xch[(a * FJ + b) * NTO + t];
    }
    if (p.dbg & 1) return;
    if (p.slab) {
        // partial tile in accumulator order: element ((a*FJ+b)*4+r)*NTO + tid -> coalesced plain stores
        float* dst = p.slab + ((size_t)wid * (p.itiles * p.jtiles) + (size_t)it * p.jtiles + jt) * (BI * BJ) + (tid & (NTO - 1));
#pragma unroll
        for (int a = 0; a < FI; ++a)
#pragma unroll
            for (int b = 0; b < FJ; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[((a * FJ + b) * 4 + r) * NTO] = acc[a][b][r];
        return;
    }
    // ---- epilogue: lane holds D[i = .. + 4*kk + r][j = .. + l15]; scatter-add into the master layout
#pragma unroll
    for (int b = 0; b < FJ; ++b) {
        const int j = j0 + wj * TJ + b * 16 + l15;
        if (j >= p.jtot) continue;
        const int tap = j / p.gp, cp = j - tap * p.gp;
        int creal;
        if (cp < p.g1p) creal = cp < p.g1r ? cp : -1;
        else creal = (cp - p.g1p) < p.g2r ? p.g1r + cp - p.g1p : -1;
        if (creal < 0) continue;
#pragma unroll
        for (int a = 0; a < FI; ++a) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ip = i0 + wi * TI + a * 16 + 4 * kk + r;
                int ireal;
                if (ip < p.d1p) ireal = ip < p.d1r ? ip : -1;
                else ireal = (ip - p.d1p) < p.d2r ? p.d1r + ip - p.d1p : -1;
                if (ireal < 0) continue;
                float* q = p.dw + ((size_t)ireal * 16 + tap) * p.gr + creal;
                if (p.fresh && p.msplit == 1) *q = acc[a][b][r];        // one slice: this workgroup is the only writer
                else atomicAdd(q, acc[a][b][r]);
            }
        }
    }
}

// dw[i][tap][c] += sum over slices of the slab tiles.  One thread per tile element and slice group (blockIdx.y);
// within a group the slices are summed in a fixed order with 8 independent partial sums (loads stay in flight);
// groups (only used when there are many slices) combine with one float atomic each - with a single group
// (msplit <= 64) the result is bitwise reproducible.
constexpr int RED_GROUP = 64;
template <int BI, int WI, int WJ>
__global__ __launch_bounds__(256) void k_wgrad_reduce(const WgradK p) {
    constexpr int BJ = 128, TI = BI / WI, TJ = BJ / WJ, FJ = TJ / 16, NT = 64 * WI * WJ;
    const int tile = blockIdx.x / (BI * BJ / 256);
    const int e = (blockIdx.x % (BI * BJ / 256)) * 256 + threadIdx.x;
    const int tid = e % NT, q = e / NT, r = q & 3, ab = q >> 2, b = ab % FJ, a = ab / FJ;     // the producer's layout
    const int lane = tid & 63, wave = tid >> 6, wi = wave % WI, wj = wave / WI;
    const int jt = tile % p.jtiles, it = tile / p.jtiles;
    const int ip = it * BI + wi * TI + a * 16 + 4 * (lane >> 4) + r;
    const int j = jt * BJ + wj * TJ + b * 16 + (lane & 15);
    if (j >= p.jtot) return;
    const int tap = j / p.gp, cp = j - tap * p.gp;
    int creal, ireal;
    if (cp < p.g1p) creal = cp < p.g1r ? cp : -1;
    else creal = (cp - p.g1p) < p.g2r ? p.g1r + cp - p.g1p : -1;
    if (ip < p.d1p) ireal = ip < p.d1r ? ip : -1;
    else ireal = (ip - p.d1p) < p.d2r ? p.d1r + ip - p.d1p : -1;
    if (creal < 0 || ireal < 0) return;
    const size_t stride = (size_t)p.itiles * p.jtiles * (BI * BJ);
    const int k0 = blockIdx.y * RED_GROUP, k1 = min(p.msplit, k0 + RED_GROUP);
    const float* src = p.slab + (size_t)tile * (BI * BJ) + e + (size_t)k0 * stride;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int k = k0;
    for (; k + 8 <= k1; k += 8, src += 8 * stride) {
#pragma unroll
        for (int u = 0; u < 8; ++u) s[u] += src[(size_t)u * stride];
    }
    for (int u = 0; k < k1; ++k, ++u, src += stride) s[u] += *src;
    const float tot = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    float* dst = p.dw + ((size_t)ireal * 16 + tap) * p.gr + creal;
    if (gridDim.y == 1) *dst = p.fresh ? tot : *dst + tot;
    else atomicAdd(dst, tot);
}

template <typename T, int BI, int WI, int WJ, int EARLY, int KG = 1, bool FAST = false>
int launch_wgrad_v(WgradK k, int msplit, hipStream_t st) {
    constexpr int KM = (sizeof(T) == 2 ? 64 : 32) * KG;
    constexpr size_t lds = (EARLY == 3 ? 3 : 2) * (KM * BI * sizeof(T) + KM * 128 * sizeof(T));
    k.jtiles = (k.jtot + 127) / 128;
    k.itiles = (k.dp + BI - 1) / BI;
    k.nwg = k.jtiles * k.itiles * msplit;
    dim3 grid(k.nwg);
    static std::atomic<bool> attr_done{false};   // (launches may come from several host threads)
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)k_wgrad<T, BI, WI, WJ, EARLY, KG, FAST>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        attr_done = true;
    }
    k.msplit = msplit;
    hipLaunchKernelGGL((k_wgrad<T, BI, WI, WJ, EARLY, KG, FAST>), grid, dim3(64 * WI * WJ * KG), lds, st, k);
    PFA_LAUNCHED("k_wgrad");
    if (k.slab) {
        hipLaunchKernelGGL((k_wgrad_reduce<BI, WI, WJ>),
                           dim3(k.jtiles * k.itiles * (BI * 128 / 256), (msplit + RED_GROUP - 1) / RED_GROUP), dim3(256), 0, st, k);
        PFA_LAUNCHED("k_wgrad_reduce");
    }
    return 0;
}

template <typename T, int BI, int WI, int WJ>
int launch_wgrad(const WgradK& k, int msplit, hipStream_t st) {
    if (k.early && sizeof(T) == 2) return launch_wgrad_v<T, BI, WI, WJ, 1>(k, msplit, st);
    return launch_wgrad_v<T, BI, WI, WJ, 0>(k, msplit, st);
}

}  // namespace

static int wgrad_setup(const pfa_wgrad_args* a, WgradK& k, int& msplit, bool& small_i) {
    PFA_REQUIRE(a && a->struct_size == (int)sizeof(pfa_wgrad_args), "pfa_wgrad: bad struct_size");
    PFA_REQUIRE(a->dtype == PFA_F32 || a->dtype == PFA_BF16, "pfa_wgrad: bad dtype");
    const int ch = pfa_ch(a->dtype);
    PFA_REQUIRE(a->stride == 1 || a->stride == 2, "pfa_wgrad: stride must be 1 or 2");
    PFA_REQUIRE(a->nb > 0 && a->hq > 0 && a->wq > 0 && a->hg > 0 && a->wg > 0, "pfa_wgrad: empty geometry");
    PFA_REQUIRE(a->d1p > 0 && a->d1p % ch == 0 && a->d2p % ch == 0 && a->g1p > 0 && a->g1p % ch == 0 && a->g2p % ch == 0,
                "pfa_wgrad: padded channel counts must be multiples of %d", ch);
    PFA_REQUIRE(a->d1r <= a->d1p && a->d2r <= a->d2p && a->g1r <= a->g1p && a->g2r <= a->g2p && a->d1r > 0 && a->g1r > 0,
                "pfa_wgrad: real channel counts exceed padded ones");
    PFA_REQUIRE(a->dir1 && a->gat1 && a->dw && (a->d2p == 0 || a->dir2) && (a->g2p == 0 || a->gat2), "pfa_wgrad: null pointer");
    PFA_REQUIRE(a->hg == (a->hq - 1) * a->stride + 2 && a->wg == (a->wq - 1) * a->stride + 2,
                "pfa_wgrad: gathered size %dx%d inconsistent with direct %dx%d at stride %d", a->hg, a->wg, a->hq,
                a->wq, a->stride);
    k.dir1 = (const char*)a->dir1; k.dir2 = (const char*)a->dir2;
    k.gat1 = (const char*)a->gat1; k.gat2 = (const char*)a->gat2; k.dw = a->dw;
    k.slab = a->slab;
    k.fresh = a->fresh;
    k.stride = a->stride; k.nb = a->nb; k.hq = a->hq; k.wq = a->wq; k.hg = a->hg; k.wg = a->wg;
    k.d1p = a->d1p; k.d1r = a->d1r; k.d2p = a->d2p; k.d2r = a->d2r; k.dp = a->d1p + a->d2p;
    k.g1p = a->g1p; k.g1r = a->g1r; k.g2p = a->g2p; k.g2r = a->g2r; k.gp = a->g1p + a->g2p; k.gr = a->g1r + a->g2r;
    k.M = a->nb * a->hq * a->wq;
    const int hw = a->hq * a->wq;
    k.magic_hw = (unsigned)(0x100000000ull / (unsigned)hw);
    k.magic_wq = (unsigned)(0x100000000ull / (unsigned)a->wq);
    if (hw == 1) k.magic_hw = 0xffffffffu;
    if (a->wq == 1) k.magic_wq = 0xffffffffu;
    k.dbg = pfa_knob(1);
    k.early = pfa_knob(4) == 1 ? 0 : pfa_knob(4) == 2 ? 1 : WGRAD_DEFAULT_EARLY;
    k.jtot = 16 * k.gp;
    small_i = k.dp <= 16;
    msplit = a->msplit < 1 ? 1 : a->msplit;
    // K-group pairs: bf16 launches on the 128-row tile with at least two reduction slices run as half as many 8-wave
    // workgroups, each reducing two 64-pixel half-steps per step - half the partial tiles to write and reduce
    // (e3/c3 43 -> 39 us, e4 49 -> 40 us; +2.2 % pairs/s, +2.0 % when pairing from four slices on only).  knob 4 = 5
    // turns them off, 6 pairs from four slices on; the explicit loop variants (knob 4 = 1..3) keep the 4-wave tile.
    const int pair_min = pfa_knob(4) == 6 ? 4 : 2;
    k.kgroups = ((pfa_knob(4) == 0 || pfa_knob(4) == 4 || pfa_knob(4) == 6) && a->dtype == PFA_BF16 && !small_i && msplit >= pair_min) ? 2 : 1;
    if (k.kgroups == 2) msplit = (msplit + 1) / 2;
    const int km = (a->dtype == PFA_BF16 ? 64 : 32) * k.kgroups;
    k.step_img = km / hw;
    k.step_row = (km % hw) / a->wq;
    k.step_col = (km % hw) % a->wq;
    int steps = (k.M + km - 1) / km;
    if (msplit > steps) msplit = steps;
    int sper = (steps + msplit - 1) / msplit;
    msplit = (steps + sper - 1) / sper;
    k.mper = sper * km;
    return 0;
}

// Row-aligned geometry for k_wgrad<..., FAST>: bf16 on the 128-row tile, one gathered source, every 128-row tile inside one
// direct source, a step (km pixels) that is a piece of one row, whole rows or whole images, no ragged last step, tensors
// small enough for 32-bit offsets.  OPT-IN: pfa_set_knob(1, 32768).  Same bits as the general staging; per layer at B = 16
// (tools/wgrad_bench.py) e2 / c2 36.5 -> 32.2 us, d5 34.0 -> 30.3, e5 24.4 -> 23.0, the one-step launches of the 2 x 2 maps +1 us
// (the per-piece set-up divides); the step +0.5 % on one stream, -0.1 % eager / -0.6 ... -1.0 % through launch lists on three,
// with every launch-size knob re-swept - so the default keeps the general staging.
static bool wgrad_fast_ok(const pfa_wgrad_args* a, WgradK& k, int km) {
    if (!(pfa_knob(1) & 32768)) return false;
    if (a->dtype != PFA_BF16 || a->g2p != 0 || k.dp <= 16) return false;
    if (a->d2p != 0 && a->d1p % 128 != 0) return false;
    const int hw = a->hq * a->wq;
    if (k.M % km != 0) return false;
    if (!(a->wq % km == 0 || (km % a->wq == 0 && (hw % km == 0 || km % hw == 0)))) return false;
    const long long dbytes = (long long)k.M * (a->d1p > a->d2p ? a->d1p : a->d2p) * 2;
    const long long gbytes = ((long long)a->nb * a->hg * a->wg + a->wg + 2) * a->g1p * 2;
    if (dbytes >= (1ll << 31) || gbytes >= (1ll << 31)) return false;
    k.xspan = a->wq < km ? a->wq : km;
    k.rps = a->wq >= km ? 1 : (km / a->wq < a->hq ? km / a->wq : a->hq);
    return true;
}

// 256-row tile with the 3-stage ring: bf16 layers with at least 256 direct channels and a K loop long enough for a ring
static bool wgrad_ring(const pfa_wgrad_args* a, const WgradK& k) {
    // (measured 3 % slower than the 128-row two-stage tile on the step's layers, tools/wgrad_bench.py: off unless knob 4 = 3)
    return a->dtype == PFA_BF16 && k.dp >= 256 && k.mper >= 3 * 64 && pfa_knob(4) == 3;
}

extern "C" int pfa_wgrad_slices(const pfa_wgrad_args* a) {
    WgradK k; int msplit; bool small_i;
    if (int rc = wgrad_setup(a, k, msplit, small_i)) return rc < 0 ? rc : -rc;
    int ps;
    if (pfa_wgrad_image_plan(a, &ps)) return ps;
    if (pfa_wgrad_patch_plan(a, &ps, nullptr, nullptr)) return ps;
    return msplit * k.kgroups;
}

extern "C" size_t pfa_wgrad_slab_floats(const pfa_wgrad_args* a) {
    WgradK k; int msplit; bool small_i;
    if (wgrad_setup(a, k, msplit, small_i)) return 0;
    int ps;
    if (pfa_wgrad_image_plan(a, &ps)) return pfa_wgrad_image_slab_floats(a, ps);
    if (pfa_wgrad_patch_plan(a, &ps, nullptr, nullptr)) return pfa_wgrad_patch_slab_floats(a, ps);
    const int bi = small_i ? 16 : (wgrad_ring(a, k) ? 256 : 128);
    return (size_t)msplit * ((k.jtot + 127) / 128) * ((k.dp + bi - 1) / bi) * bi * 128;
}

extern "C" int pfa_wgrad(const pfa_wgrad_args* a, pfa_stream s) {
    WgradK k; int msplit; bool small_i;
    if (int rc = wgrad_setup(a, k, msplit, small_i)) return rc;
    hipStream_t st = (hipStream_t)s;

    hipEvent_t te0, te1;
    const bool timing = pfa_timing_begin(st, &te0, &te1);
    int rc;
    const char* name;
    int pslices = 0;
    if (a->slab && pfa_wgrad_image_plan(a, &pslices)) {
        rc = pfa_wgrad_image(a, st, &name);
        msplit = pslices;
    } else if (pfa_wgrad_patch_plan(a, &pslices, nullptr, nullptr)) {
        rc = pfa_wgrad_patch(a, st, &name);
        msplit = pslices;
    } else if (a->dtype == PFA_BF16) {
        if (!small_i && wgrad_ring(a, k)) { rc = launch_wgrad_v<bf16_t, 256, 4, 2, 3>(k, msplit, st); name = "k_wgrad<bf16,256,ring>"; }
        else if (k.kgroups == 2) {
            if (wgrad_fast_ok(a, k, 128)) { rc = launch_wgrad_v<bf16_t, 128, 2, 2, 1, 2, true>(k, msplit, st); name = "k_wgrad<bf16,128,kg2,rows>"; }
            else { rc = launch_wgrad_v<bf16_t, 128, 2, 2, 1, 2>(k, msplit, st); name = "k_wgrad<bf16,128,kg2>"; }
        }
        else if (small_i) { rc = launch_wgrad<bf16_t, 16, 1, 4>(k, msplit, st); name = "k_wgrad<bf16,16>"; }
        else if (!k.early && wgrad_fast_ok(a, k, 64)) { rc = launch_wgrad_v<bf16_t, 128, 2, 2, 0, 1, true>(k, msplit, st); name = "k_wgrad<bf16,128,rows>"; }
        else { rc = launch_wgrad<bf16_t, 128, 2, 2>(k, msplit, st); name = "k_wgrad<bf16,128>"; }
    } else {
        rc = small_i ? launch_wgrad<float, 16, 1, 4>(k, msplit, st) : launch_wgrad<float, 128, 2, 2>(k, msplit, st);
        name = small_i ? "k_wgrad<f32,16>" : "k_wgrad<f32,128>";
    }
    if (timing) {
        // algorithmic FLOPs: the MACs of the layer's forward on real channels; bytes: both operands once + the gradient
        const double es = a->dtype == PFA_BF16 ? 2.0 : 4.0;
        const double flop = 2.0 * (double)k.M * 16.0 * (double)(k.d1r + k.d2r) * (double)k.gr;
        const double bytes = es * ((double)k.M * k.dp + (double)a->nb * a->hg * a->wg * k.gp) + 4.0 * (double)(k.d1r + k.d2r) * 16.0 * k.gr;
        pfa_timing_end(st, te0, te1, name, flop, bytes, msplit);
    }
    return rc;
}
