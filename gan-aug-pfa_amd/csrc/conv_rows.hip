// Implicit GEMM over im2col ROWS for the launches no patch kernel takes - the generator's small maps (4 x 4, 2 x 2 and 1 x 1
// position grids: e6, e7, d7, d6 and their input gradients, M = 64 .. 256 rows at batch 16, 8-17 MB of weights each) and
// whatever else the patch planners turn down - with the K loop of the patch kernels (conv_patch2.hip / conv_tconv4.hip).
//
// Replaces, for bf16 launches with 64-multiple channel counts: the 256 x 128 ring tile of k_gather_gemm (gather_gemm.hip), i.e.
// nn.Conv2d / nn.ConvTranspose2d forward and input gradients of models.py:177,184,189,194 at the U-Net's inner levels.
// Same arguments (GatherK), same K order (tap-major, source 1's chunks before source 2's), same K slices, same work-list
// orders, same outputs (whole-row epilogue of patch_epilogue.h: partial planes, bias, fused backward, affine map, statistics).
//
// A workgroup owns 256 rows (positions) x 64 columns (output channels): 8 waves as 4 x 2, a wave 64 rows x 32 columns = two
// MFMA 32x32x16 tiles.  A K-step is 64 channels of one tap of one source: the 256 gathered activation rows (32 KiB) and the
// 64 weight rows (8 KiB), both as rows of 128 bytes with the chunk swizzle c ^ ((row >> 1) & 7) on the source side.  40 KiB per
// stage: FOUR ring slots fill the CU's 160 KiB, so a stage is issued three steps ahead of its MFMAs, waited for with a counted
// vmcnt a step later (two steps of HBM latency hidden: every weight byte of these layers is a first touch) and read in place
// one step ahead (counted lgkmcnt).  All DMA in the buffer form of pfa_common.h: a row's byte offset is fixed for the
// whole loop (one per source: their pixel pitches may differ), the tap's shift and the channel chunk move the BASE, and a
// row whose tap falls outside the image - or that does not exist - carries PFA_DMA_OOB: one select on a per-row tap mask.
// The loop runs whole turns of the ring: steps behind the slice's end stage zeros (out-of-range scalars, no traffic) and add
// nothing, so there is no tail code and every step issues and waits alike.
#include "patch_epilogue.h"

#include <type_traits>
#include <utility>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int RW_BM = 256, RW_BN = 64;
constexpr int RW_A = RW_BM * 128, RW_W = RW_BN * 128, RW_STAGE = RW_A + RW_W;     // 32 + 8 KiB
constexpr int RW_SLOTS = 4;
constexpr int RW_LDS = RW_SLOTS * RW_STAGE;                                        // 163,840: the whole LDS of a CU
static_assert(RW_LDS <= 163840 && 256 * patch_image_stride<RW_BN>() <= RW_LDS, "LDS");

template <int N> __device__ __forceinline__ void rw_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int OFF> __device__ __forceinline__ void rw_rd(unsigned addr, bf16x8& v) {      // in place: see conv_patch2.hip lds_rd
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(v) : "v"(addr), "n"(OFF));
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void k_conv_rows(const GatherK p) {
    constexpr int FM = 2, KS = 4, NR = 1 + FM;               // fragment reads of a k-step: one weight, two activation
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int lr = lane & 31, lh = lane >> 5;
    // work list: the orders of k_gather_gemm (gather_gemm.hip), chosen by the host from the bytes the eight L2s would fetch
    unsigned wid = xcd_remap(blockIdx.x, p.nwg);
    int nt, mt, split, cls;
    if (p.order == 1) {
        nt = wid % p.ntiles; wid /= p.ntiles;
        split = wid % p.splitk; wid /= p.splitk;
        cls = wid % p.classes; mt = wid / p.classes;
    } else if (p.order == 2) {
        mt = wid % p.mtiles; wid /= p.mtiles;
        split = wid % p.splitk; wid /= p.splitk;
        cls = wid % p.classes; nt = wid / p.classes;
    } else {
        nt = wid % p.ntiles; wid /= p.ntiles;
        mt = wid % p.mtiles; wid /= p.mtiles;
        split = wid % p.splitk; cls = wid / p.splitk;
    }
    const int py = cls >> 1, px = cls & 1;
    const int m0 = mt * RW_BM, n0 = nt * RW_BN;
    const bool tconv = p.mode == PFA_TCONV_S2;
    const int sgn = p.mode >= PFA_TCONV_S2 ? -1 : 1;         // taps walk backwards in the transposed forms
    const int ntaps = tconv ? 4 : 16;
    const int hw = p.hq * p.wq;
    const int pad = 3 * p.ws + 4;                            // pixels the row offsets are shifted by: no offset is negative

    // ---- the slice's K-steps and the DMA stream's position in them (three steps ahead of the MFMAs)
    const int ks0 = split * p.kper;
    const int nlive = min(p.ksteps, ks0 + p.kper) - ks0;
    const int nsteps = (nlive + RW_SLOTS - 1) / RW_SLOTS * RW_SLOTS;
    int ds = 0, d_tap = ks0 / p.spt, d_in = ks0 - d_tap * p.spt;     // step of the slice, its tap, its K-step inside the tap

    // ---- weight staging: one piece per wave (rows 8 wave .. +7 of the tile)
    const int cs = lane & 7;
    unsigned woff;
    {
        const int r = wave * 8 + (lane >> 3), n = n0 + r;
        const int g = cs ^ ((r >> 1) & 7);
        woff = n < p.n ? (unsigned)((n * 16 * p.ctot + g * 8) * 2) : PFA_DMA_OOB;
    }
    // ---- activation staging: four pieces per wave (rows 8 (8 j + wave) .. +7); per row the pixel of tap (0, 0) (+ pad) times
    // the source's pixel pitch, and a bit per tap: the tap's pixel exists
    const unsigned c1x2 = (unsigned)p.c1 * 2, c2x2 = (unsigned)p.c2 * 2;
    unsigned voff1[4], voff2[4], tmask[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (j * 8 + wave) * 8 + (lane >> 3), m = m0 + r;
        const int g = cs ^ ((r >> 1) & 7);
        const unsigned rowok = (unsigned)(m < p.M);
        const int mm = m < p.M ? m : 0;
        const int ni = mm / hw, rem = mm - ni * hw;
        const int qy = rem / p.wq, qx = rem - qy * p.wq;
        int by, bx;
        if (p.mode == PFA_CONV_S2) { by = 2 * qy - 1; bx = 2 * qx - 1; }
        else if (p.mode == PFA_CONV_S1) { by = qy - 1; bx = qx - 1; }
        else if (tconv) { by = qy + py; bx = qx + px; }
        else { by = qy + 1; bx = qx + 1; }
        // bit t = tap (ty, tx) of this row reads a pixel that exists: row bits x column bits (t = 4 ty + tx; transposed: 2 ty + tx)
        unsigned xb = 0, mk = 0;
        const int tw = tconv ? 2 : 4;
        for (int tx = 0; tx < tw; ++tx) xb |= (unsigned)((unsigned)(bx + sgn * tx) < (unsigned)p.ws) << tx;
        for (int ty = 0; ty < tw; ++ty) mk |= (0u - (unsigned)((unsigned)(by + sgn * ty) < (unsigned)p.hs)) & (xb << (tw * ty));
        mk &= 0u - rowok;
        tmask[j] = mk;
        const unsigned pix = (unsigned)((ni * p.hs + by) * p.ws + bx + pad);
        voff1[j] = pix * c1x2 + (unsigned)g * 16;
        voff2[j] = pix * c2x2 + (unsigned)g * 16;
    }
    // what a step's pieces share (computed once per step, scalar): the activation base with the tap's shift and the chunk,
    // the weights' scalar offset, the tap's bit (0 behind the slice's end: every row out of range)
    struct ST { const char* abase; unsigned wsoff, tbit; bool second; };
    auto step_state = [&]() -> ST {
        const bool live = ds < nlive;
        const int ty = tconv ? d_tap >> 1 : d_tap >> 2, tx = tconv ? d_tap & 1 : d_tap & 3;
        const int widx = tconv ? (1 - py + 2 * ty) * 4 + (1 - px + 2 * tx) : d_tap;
        const bool second = d_in >= p.sp1;
        const int cc = second ? d_in - p.sp1 : d_in;
        const long long shift = (long long)(sgn * (ty * p.ws + tx) - pad) * (long long)(second ? c2x2 : c1x2) + cc * 128;
        ST s;
        s.second = second;
        s.abase = (second ? p.src2 : p.src1) + shift;
        s.wsoff = live ? (unsigned)((widx * p.ctot + (second ? p.c1 : 0) + cc * 64) * 2) : PFA_DMA_OOB;
        s.tbit = live ? 1u << d_tap : 0u;
        return s;
    };
    auto advance = [&]() {
        ++ds;
        if (++d_in == p.spt) { d_in = 0; ++d_tap; }
    };
    auto stage = [&](const ST& s, auto slot_, int op0, int op1) {           // DMA ops [op0, op1) of a step into ring slot SL
        constexpr int SL = decltype(slot_)::value;
        char* sb = smem + SL * RW_STAGE;
#pragma unroll
        for (int k = op0; k < op1; ++k) {
            if (k < 4) {
                const unsigned v = (tmask[k] & s.tbit) ? (s.second ? voff2[k] : voff1[k]) : PFA_DMA_OOB;
                buf_dma16(s.abase, sb + ((k * 8 + wave) * 64) * 16, v, 0u);
            } else {
                buf_dma16(p.w, sb + RW_A + wave * 1024, woff, s.wsoff);
            }
        }
    };

    f32x16 acc[FM];
#pragma unroll
    for (int b = 0; b < FM; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;

    // ---- fragment addresses (32x32x16: lane (lr, lh) supplies row lr of a 32-row tile, the 16-byte chunk 2 s + lh of k-step s).
    // Two register sets: slots 0 / 1 at immediates 0 / RW_STAGE from the first, slots 2 / 3 from the second (the ds offset
    // field holds 16 bits).
    const unsigned lds0 = (unsigned)(size_t)LDS_PTR(smem);
    unsigned aadr[2][FM][KS], wadr[2][KS];
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) {
#pragma unroll
        for (int b = 0; b < FM; ++b) {
            const int row = wm * 64 + b * 32 + lr;
            aadr[0][b][s2] = lds0 + row * 128 + (((2 * s2 + lh) ^ ((row >> 1) & 7)) << 4);
            aadr[1][b][s2] = aadr[0][b][s2] + 2 * RW_STAGE;
        }
        const int rw = wn * 32 + lr;
        wadr[0][s2] = lds0 + RW_A + rw * 128 + (((2 * s2 + lh) ^ ((rw >> 1) & 7)) << 4);
        wadr[1][s2] = wadr[0][s2] + 2 * RW_STAGE;
    }
    bf16x8 fw[KS], fa[FM][KS];   // the ONE fragment set
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) {
        fw[s2] = bf16x8{};
#pragma unroll
        for (int b = 0; b < FM; ++b) fa[b][s2] = bf16x8{};
    }
    // fragment E (0: weights, 1 + b: activation tile b) of k-step S2 of the stage in ring slot SL
    auto read_frag = [&](auto slot_, auto s2_, auto e_) {
        constexpr int SL = decltype(slot_)::value, S2 = decltype(s2_)::value, E = decltype(e_)::value;
        constexpr int SET = SL >> 1, OFF = (SL & 1) * RW_STAGE;
        if constexpr (E == 0) rw_rd<OFF>(wadr[SET][S2], fw[S2]);
        else rw_rd<OFF>(aadr[SET][E - 1][S2], fa[E - 1][S2]);
    };
    auto read_kstep = [&](auto slot_, auto s2_) {
        read_frag(slot_, s2_, std::integral_constant<int, 0>{});
        read_frag(slot_, s2_, std::integral_constant<int, 1>{});
        read_frag(slot_, s2_, std::integral_constant<int, 2>{});
    };

    // ---- prologue: stages 0 .. 2 (weights and activations alike: one step's five pieces per wave each)
    { const ST s = step_state(); stage(s, std::integral_constant<int, 0>{}, 0, 5); advance(); }
    { const ST s = step_state(); stage(s, std::integral_constant<int, 1>{}, 0, 5); advance(); }
    { const ST s = step_state(); stage(s, std::integral_constant<int, 2>{}, 0, 5); advance(); }
    rw_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    read_kstep(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    read_kstep(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
    read_kstep(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
    read_kstep(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{});

    // one K-step on ring slot SL: its MFMAs, the next step's fragments read ahead (slot SL + 1), stage +3 issued (slot SL + 3)
    auto body = [&](auto slot_) {
        constexpr int SL = decltype(slot_)::value, NSL = (SL + 1) & 3, DSL = (SL + 3) & 3;
        const ST st = step_state();              // (before the barrier: scalar work, nothing waits for it)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        auto kstep = [&](auto s2_) {
            constexpr int S2 = decltype(s2_)::value;
            PFA_WAIT_LGKM((KS - 1) * NR);
            pin(fw[S2]);
#pragma unroll
            for (int b = 0; b < FM; ++b) pin(fa[b][S2]);
            __builtin_amdgcn_sched_barrier(0);
            // the k-step's reads and DMA in the shadows of its two MFMAs: M(w,a0) [DMA] R(a0) M(w,a1) R(w) R(a1)
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[S2], fa[0][S2], acc[0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            stage(st, std::integral_constant<int, DSL>{}, S2 == 0 ? 0 : S2 + 1, S2 + 2);     // pieces 0 1 | 2 | 3 | 4 (weights)
            __builtin_amdgcn_sched_barrier(0);
            read_frag(std::integral_constant<int, NSL>{}, s2_, std::integral_constant<int, 1>{});
            __builtin_amdgcn_sched_barrier(0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[S2], fa[1][S2], acc[1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            read_frag(std::integral_constant<int, NSL>{}, s2_, std::integral_constant<int, 0>{});
            read_frag(std::integral_constant<int, NSL>{}, s2_, std::integral_constant<int, 2>{});
            __builtin_amdgcn_sched_barrier(0);
        };
        kstep(std::integral_constant<int, 0>{});
        kstep(std::integral_constant<int, 1>{});
        kstep(std::integral_constant<int, 2>{});
        kstep(std::integral_constant<int, 3>{});
        advance();
        rw_wait_vm<5>();                         // everything but this step's pieces has landed
    };
    for (int i = 0; i < nsteps; i += RW_SLOTS) {
        body(std::integral_constant<int, 0>{});
        body(std::integral_constant<int, 1>{});
        body(std::integral_constant<int, 2>{});
        body(std::integral_constant<int, 3>{});
    }
    rw_wait_vm<0>();                             // (the pieces issued behind the slice's end,
    PFA_WAIT_LGKM(0);                            //  the fragments read behind its last step)
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) { pin(fw[s2]); pin(fa[0][s2]); pin(fa[1][s2]); }

    // ---- epilogue: accumulators -> fp32 [row][channel] image in LDS -> whole NHWC rows out (patch_epilogue.h).
    // 32x32 accumulator: lane holds, for register group g = 0..3, channels 8 g + 4 lh .. +3 of row lr of the tile.
    constexpr int RS = patch_image_stride<RW_BN>();
    __syncthreads();                             // every wave is done with the ring
#pragma unroll
    for (int b = 0; b < FM; ++b) {
        char* row = smem + (wm * 64 + b * 32 + lr) * RS;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = {acc[b][4 * g], acc[b][4 * g + 1], acc[b][4 * g + 2], acc[b][4 * g + 3]};
            *(f32x4*)(row + (wn * 32 + 8 * g + 4 * lh) * 4) = v;
        }
    }
    __syncthreads();
    auto pix_of = [&](int r, bool& valid, size_t& opix) {
        const int m = m0 + r;
        valid = m < p.M;
        opix = 0;
        if (valid) {
            if (tconv) {
                const int ni = m / hw, rem = m - ni * hw;
                const int qy = rem / p.wq, qx = rem - qy * p.wq;
                opix = ((size_t)ni * p.ho + 2 * qy + py) * p.wo + 2 * qx + px;
            } else opix = (size_t)m;
        }
    };
    patch_epilogue_rows<RW_BN>(p, n0, tid, split, (size_t)cls * p.mtiles + mt, smem, pix_of);
}

}  // namespace

// bf16 launches on whole 64-channel chunks (both sources, the output columns), no tanh, tensors small enough for the
// shifted 32-bit row offsets.  OPT-IN: pfa_set_knob(1, 16384).  Measured on the training step's eight small-map launches (M <=
// 256 rows, 8 K-steps per slice): 14.1 us against the ring tile's 15.9 us per launch in situ, the step +0.55 % on one stream and
// -0.9 % on three (64-column tiles fetch the activations through L2 twice as often as the ring's 128-column tiles, beside the
// weight gradients' traffic).  At 8 K-steps per workgroup the loop is a quarter of such a launch - ~5 us of launch, ~3 us until
// the first stage has landed, ~2 us of partial-plane stores are the rest - so no K loop brings these launches under ~10 us.
bool pfa_conv_rows_ok(const GatherK& k, int dtype) {
    if (!(pfa_knob(1) & 16384)) return false;
    if (dtype != PFA_BF16 || !k.wide || !k.small32) return false;
    if (k.c1 % 64 || k.c2 % 64 || k.n % 64 || k.act == PFA_ACT_TANH) return false;
    const long long cmax = k.c1 > k.c2 ? k.c1 : k.c2;
    if (((long long)k.nb * k.hs * k.ws + 6 * k.ws + 8) * cmax * 2 >= (1LL << 31)) return false;
    return true;
}

// k: as dispatch_gather would launch it (splitk / kper final; bias / statistics moved to the finish pass when K-sliced);
// order_of: the work-list order for the tile counts set here (gather_gemm.hip's choose_order).
int pfa_conv_rows(GatherK k, int classes, int (*order_of)(const GatherK&), hipStream_t st, const char** kernel_name) {
    k.mtiles = (k.M + RW_BM - 1) / RW_BM;
    k.ntiles = k.n / RW_BN;
    k.nwg = k.mtiles * k.ntiles * classes * k.splitk;
    k.classes = classes;
    k.order = order_of(k);
    static std::atomic<bool> attr_done{false};   // (launches may come from several host threads)
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)k_conv_rows, hipFuncAttributeMaxDynamicSharedMemorySize, RW_LDS);
        attr_done = true;
    }
    if (kernel_name) *kernel_name = "k_conv_rows<bf16,256,64>";
    hipLaunchKernelGGL(k_conv_rows, dim3(k.nwg), dim3(512), RW_LDS, st, k);
    PFA_LAUNCHED("k_conv_rows");
    return 0;
}
