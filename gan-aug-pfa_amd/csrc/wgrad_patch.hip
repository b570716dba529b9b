// Weight gradient of the MFMA-heavy layers with the gathered operand's PATCH kept in LDS.
//
// Replaces, for the wide layers, the weight-gradient half of autograd's backward for nn.Conv2d (models.py:177,230,238) and
// nn.ConvTranspose2d (models.py:189,194) as executed by train_gan.py:62,70 - the launches pfa_wgrad used to run on its
// 128 x 128 tile (wgrad.hip), which re-stages the direct operand for each of the 16 taps and the gathered pixels for each
// tap on its own: 8 LDS-DMA instructions per wave for 32 MFMAs, bound by the L2->LDS instruction rate.
//
//   dw[i][ky*4+kx][c] += sum_pixels direct[p][i] * gathered[s*p - 1 + (ky, kx)][c]
//
// A workgroup owns 128 direct channels x ONE tap row ky x all four kx x 64 gathered channels (8 waves: wave = (half of the
// direct channels, kx), 64 x 64 outputs each as 2 x 2 MFMA 32x32x16 tiles) and walks a slice of the pixels in steps of
// 64 = TY x TX positions.  Per step it stages the direct tile [64 px][128 ch] once and, for its tap row, the gathered
// PATCH with halo once ((TX + 3) pixels per row at stride 1; the odd and the even column plane, TX + 1 pixels each, at
// stride 2): the four kx waves read the same patch at four constant offsets.  4-5 DMA instructions per wave and step for
// 16 MFMAs of twice the work.
// Both operands are pixel-major while the contraction runs over pixels: fragments come through the hardware transpose
// read ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group).  LDS images: direct rows of 256 B with 16-byte chunk
// c at c ^ ((px & 3) << 2), patch pixels of 128 B with chunk c at c ^ (((slot >> 1) & 1) << 2): the 4 rows x 64 B a 32-lane
// half touches fall on four distinct quarters of the 256-byte bank row, at every tap offset.
// K loop as in conv_patch2.hip: one barrier per step, a ring of 4 stages with two in flight across every barrier (counted
// vmcnt), fragments read IN PLACE two k-steps (2 x 16 pixels) ahead of their MFMAs - the LDS queue holds 15 reads - and
// every LDS address is a per-lane register plus an immediate (the loop is unrolled over the 4 stages).
#include "pfa_common.h"
#include <atomic>

#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

struct WgpK {
    const char* dir1;
    const char* dir2;
    const char* gat;
    float* dw;
    float* slab;             // [msplit][tiles][128][4][64] partial tiles, or NULL: float atomics into dw
    int d1p, dp, gp;         // direct channels of source 1 / in all, gathered channels
    int nb, hq, wq, hg, wg;
    int tiles_x, tiles_y, nsteps, sper, msplit;
    int fresh;               // pfa_wgrad_args.fresh
    unsigned long long* stamps;   // developer timing (pfa_set_debug_buffer): 8 slots per workgroup, or NULL
    int itiles, ctiles, nwg;
};

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
// transpose read INTO the register pair that held the same half-fragment two k-steps ago (see conv_patch2.hip)
template <int OFF> __device__ __forceinline__ void tr_rd(unsigned addr, bf16x4& v) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "+v"(v) : "v"(addr), "n"(OFF));
}
template <typename F> __device__ __forceinline__ void pin(F& v) { asm volatile("" : "+v"(v)); }

constexpr int DIR_BYTES = 64 * 256;
constexpr int wgp_stages(int stage_bytes) { return 4 * stage_bytes <= 163840 ? 4 : 3; }

// STRIDE 1 | 2; TX 32 (2 x 32 positions per step) | 16 (4 x 16)
// ABL: ablation bits of a diagnostic instantiation (1 no MFMA, 2 no fragment reads, 4 no DMA - results are wrong)
// NKY: tap rows per workgroup.  2: the pair (ky0, ky0 + KYD) with KYD = 1 at stride 1, 2 at stride 2 - the second row's
// gathered pixels are the first's one patch row further down, so the patch grows by ONE row (TY + 1 instead of 2 TY), the
// direct tile and its fragments serve both rows, and a step carries twice the MFMAs for 6 instead of 2 x 5 DMA instructions;
// three ring stages of 48 KiB instead of four of 40.
template <int STRIDE, int TX, int ABL = 0, int NKY = 1>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void k_wgrad_patch(const WgpK p) {
    constexpr bool no_mfma = ABL & 1, no_read = ABL & 2, no_dma = ABL & 4;
    constexpr int KYG = 4 / NKY, KYD = STRIDE == 1 ? 1 : 2;               // tap-row groups; distance of a group's two rows
    constexpr int TY = 64 / TX;
    constexpr int PLW = STRIDE == 1 ? TX + 3 : TX + 1;                    // pixels of a patch row (per plane at stride 2)
    constexpr int RP = ((STRIDE == 1 ? PLW : 2 * PLW) + 3) & ~3;          // row pitch in pixel slots, a multiple of 4
    constexpr int PSLOTS = (TY + NKY - 1) * RP;
    constexpr int PP = (PSLOTS * 8 + 511) / 512;                          // DMA passes of the patch
    constexpr int PATCH_BYTES = PP * 8192;
    constexpr int STAGE = DIR_BYTES + PATCH_BYTES;
    // Ring: four stages where they fit - one barrier at the top of a step, the stage three steps ahead issued over its four
    // k-steps.  Three stages (two tap rows at stride 2 on the 2 x 32 tile: 48 KiB each): the barrier sits in the MIDDLE of
    // the step - behind it every wave has left stage i-1 (its last fragments went into the MFMAs of the step before) and
    // stage i+1 has landed (vmcnt(0) in front of the barrier) - and the second half of the step issues stage i+2 into the slot
    // of stage i-1 and reads ahead into stage i+1.
    constexpr int NSTAGE = wgp_stages(STAGE);
    constexpr int NOPS = 2 + PP;                                          // DMA instructions per wave and step (at most)
    // the last patch pass covers PSLOTS * 8 - (PP - 1) * 512 lanes: only its first LASTW waves issue it (the DMA is priced per
    // instruction: on 7 of 8 waves one of 4-5 instructions per step disappears)
    constexpr int LASTW = (PSLOTS * 8 - (PP - 1) * 512 + 63) / 64;
    static_assert(LASTW >= 1 && LASTW <= 8, "last patch pass");
    static_assert(NSTAGE * STAGE <= 163840, "LDS");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave & 1, kx = wave >> 1;
    const bool last_pass = wave < LASTW;                     // (wave-uniform) this wave issues the last patch pass
    // work list: (c tile, ky, i tile) fastest - the workgroups of one pixel slice share both operands' pixels - then slice
    unsigned wid = xcd_remap(blockIdx.x, p.nwg);
    const int ct = wid % p.ctiles; wid /= p.ctiles;
    const int kg = wid % KYG; wid /= KYG;
    const int ky = NKY == 1 ? kg : (STRIDE == 1 ? 2 * kg : kg);          // (first) tap row of this workgroup
    const int it = wid % p.itiles;
    const int split = wid / p.itiles;
    const int i0 = it * 128, c0 = ct * 64;
    auto stamp = [&](int k) {          // shader clock at points of the workgroup's life (a separate measurement, never a result)
        if (p.stamps && tid == 0) p.stamps[(size_t)blockIdx.x * 8 + k] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);
    if (p.stamps && tid == 0) p.stamps[(size_t)blockIdx.x * 8 + 4] = __builtin_amdgcn_s_memrealtime();
    const int step0 = split * p.sper;
    const int nlive = min(p.nsteps, step0 + p.sper) - step0;       // real steps of this slice
    const int nsteps = (nlive + NSTAGE - 1) / NSTAGE * NSTAGE;     // the loop runs whole turns of the ring: the extra steps of the
                                                                   // last slice stage zeros and add nothing

    // ---- DMA roles (LDS-DMA in the buffer form of pfa_common.h: a per-lane offset that is fixed for the whole loop, ONE
    // scalar per operand and step for the tile's position, PFA_DMA_OOB where the hardware is to write zeros).
    // Direct tile: 2 pieces; lane -> (pixel, chunk position).  Edge flags (4 bits per piece): 1 = beyond the right edge in the
    // last tile column, 2 = beyond the bottom edge in the last tile row, 4 = left of the image in the first column,
    // 8 = above the image in the first tile row.
    const bool d_first = i0 < p.d1p;
    const char* dbase = d_first ? p.dir1 : p.dir2;
    const int dch = d_first ? p.d1p : p.dp - p.d1p, di0 = d_first ? i0 : i0 - p.d1p;
    const int qx_last = (p.tiles_x - 1) * TX, qy_last = (p.tiles_y - 1) * TY;
    unsigned d_off[2], d_flag = 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = j * 512 + tid, px = q >> 4, cpos = q & 15;
        const int ch = (cpos ^ ((px & 3) << 2)) * 8;
        const int r = px / TX, cx = px % TX;
        d_off[j] = (unsigned)(((r * p.wq + cx) * dch + di0 + ch) * 2);
        d_flag |= (unsigned)(qx_last + cx >= p.wq) << (4 * j);
        d_flag |= ((unsigned)(qy_last + r >= p.hq) << 1) << (4 * j);
    }
    // Patch: the tile's pixels start one pixel left of / above the tile: the per-lane offset counts from there and the
    // gathered operand's base is moved back by one row and one pixel, so that no offset is negative.  A slot no fragment
    // reads (row padding, the tail of the last pass) carries PFA_DMA_OOB for good.
    unsigned g_off[PP], g_flag = 0;
#pragma unroll
    for (int j = 0; j < PP; ++j) {
        const int q = j * 512 + tid, slot = q >> 3, cpos = q & 7;
        const int ch = (cpos ^ (((slot >> 1) & 1) << 2)) * 8;
        const int r = slot / RP, jj = slot - r * RP;
        int xrel; unsigned dead;
        if (STRIDE == 1) { xrel = jj - 1; dead = (unsigned)(jj >= PLW); }
        else { const int pl = jj / PLW, idx = jj - pl * PLW; xrel = 2 * idx - 1 + pl; dead = (unsigned)(jj >= 2 * PLW); }
        const int yrel = STRIDE * r - 1 + ky;
        dead |= (unsigned)(slot >= PSLOTS);
        const unsigned f = (unsigned)(STRIDE * qx_last + xrel >= p.wg) | ((unsigned)(STRIDE * qy_last + yrel >= p.hg) << 1) |
                           ((unsigned)(xrel < 0) << 2) | ((unsigned)(yrel < 0) << 3);
        const unsigned keep = dead - 1u;         // all ones for a live slot
        g_off[j] = ((unsigned)((((yrel + 1) * p.wg + xrel + 1) * p.gp + c0 + ch) * 2) & keep) | (PFA_DMA_OOB & ~keep);
        g_flag |= f << (4 * j);
    }
    const char* const gbase = p.gat - (size_t)(p.wg + 1) * p.gp * 2;
    // position of the DMA stream (NSTAGE-1 steps ahead of the MFMAs): tile (img, ty, tx) of step `ds`
    int ds = 0, d_tx, d_ty, d_img;
    {
        const int t = step0;
        const int per = p.tiles_x * p.tiles_y;
        d_img = t / per; const int rem = t - d_img * per;
        d_ty = rem / p.tiles_x; d_tx = rem - d_ty * p.tiles_x;
    }
    // what a step's pieces share: the two scalar offsets of its tile (out of range behind the slice's last step: those
    // pieces are still issued - zeros into a free ring slot, no memory traffic - so every step issues and waits alike) and
    // the tile's edge mask
    struct TS { unsigned dsoff, gsoff, emask; };
    auto tile_state = [&]() -> TS {
        const int qy0 = d_ty * TY, qx0 = d_tx * TX;
        const bool live = ds < nlive;
        TS t;
        t.emask = (d_tx == p.tiles_x - 1 ? 1u : 0u) | (d_ty == p.tiles_y - 1 ? 2u : 0u) | (d_tx == 0 ? 4u : 0u) | (d_ty == 0 ? 8u : 0u);
        t.dsoff = live ? (unsigned)(((d_img * p.hq + qy0) * p.wq + qx0) * dch * 2) : PFA_DMA_OOB;
        t.gsoff = live ? (unsigned)(((d_img * p.hg + STRIDE * qy0) * p.wg + STRIDE * qx0) * p.gp * 2) : PFA_DMA_OOB;
        return t;
    };
    auto stage = [&](const TS& t, int slot, int op0, int op1) {           // DMA ops [op0, op1) of the tile t into ring slot `slot`
        if (no_dma) return;
        char* sbase = smem + slot * STAGE;
#pragma unroll
        for (int k = op0; k < op1; ++k) {
            if (k < 2) {
                const unsigned v = (d_flag & ((t.emask & 3u) << (4 * k))) ? PFA_DMA_OOB : d_off[k];
                buf_dma16(dbase, sbase + (k * 512 + wave * 64) * 16, v, t.dsoff);
            } else {
                const int j = k - 2;
                if (j == PP - 1 && !last_pass) continue;
                // (every lane takes part: masking the lanes of slots no fragment reads out of the instruction made the step
                // 5 % slower, the DMA is not priced per active lane)
                const unsigned v = (g_flag & (t.emask << (4 * j))) ? PFA_DMA_OOB : g_off[j];
                buf_dma16(gbase, sbase + DIR_BYTES + (j * 512 + wave * 64) * 16, v, t.gsoff);
            }
        }
    };
    auto advance = [&]() {
        ++ds;
        if (++d_tx == p.tiles_x) { d_tx = 0; if (++d_ty == p.tiles_y) { d_ty = 0; ++d_img; } }
    };

    f32x16 acc[NKY][2][2];
#pragma unroll
    for (int y = 0; y < NKY; ++y)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[y][a][b][e] = 0.f;

    // ---- fragment addresses (ds_read_b64_tr_b16: lane 4q+pp of a 16-lane group supplies row q, columns 4pp..4pp+3 of the
    // group's 4 x 16 block; 32x32x16 operands: lanes 0-15 / 16-31 = rows 0-15 / 16-31 of the tile, lanes 32-63 the second
    // 8 k of the 16)
    const unsigned lds0 = (unsigned)(size_t)LDS_PTR(smem);
    const int grp = lane >> 4, cb = 16 * (grp & 1), lh = grp >> 1, rq = (lane & 15) >> 2, pp4 = (lane & 3) * 4;
    unsigned aadr[2][NSTAGE], badr[2][NSTAGE];     // [tile][ring slot]: the ds offset field holds 16 bits
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int ch = wi * 64 + a * 32 + cb + pp4, px = 8 * lh + rq;
        aadr[a][0] = lds0 + px * 256 + ((((ch >> 3) ^ (rq << 2)) & 15) << 4) + (ch & 7) * 2;
#pragma unroll
        for (int sl = 1; sl < NSTAGE; ++sl) aadr[a][sl] = aadr[a][0] + sl * STAGE;
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int ch = b * 32 + cb + pp4;
        const int tap = STRIDE == 1 ? kx : ((kx & 1) ? PLW : 0) + (kx >> 1);       // stride 2: kx odd -> even-column plane
        const int slot = 8 * lh + rq + tap;                                          // + the (k-step, half) part below
        badr[b][0] = lds0 + DIR_BYTES + slot * 128 + ((((ch >> 3) ^ (((slot >> 1) & 1) << 2)) & 7) << 4) + (ch & 7) * 2;
#pragma unroll
        for (int sl = 1; sl < NSTAGE; ++sl) badr[b][sl] = badr[b][0] + sl * STAGE;
    }
    // immediate of (k-step s, half e): direct rows 16 s + 4 e; patch slots of pixel 16 s + 4 e of the step
    // fragments of two k-steps (parity of the k-step index)
    bf16x4 fa[2][2][2], fb[2][NKY][2][2];     // [set][tile][half], [set][tap row][tile][half]
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                fa[s][t][e] = bf16x4{};
#pragma unroll
                for (int y = 0; y < NKY; ++y) fb[s][y][t][e] = bf16x4{};
            }
    constexpr int NR = 4 + 4 * NKY;           // fragment reads of a k-step
    // fragment group G of k-step S of the stage in ring slot SL: 0 / 1 = direct tile a (two halves), 2 + y * 2 + b = gathered
    // tile b of tap row y (two halves)
    auto read_frag = [&](auto slot_, auto s_, auto g_) {
        if (no_read) return;
        constexpr int SL = decltype(slot_)::value, S = decltype(s_)::value, SET = S & 1, G = decltype(g_)::value;
        constexpr int ppix = TX == 32 ? (S >> 1) * RP + 16 * (S & 1) : S * RP;      // first pixel slot of the k-step's rows
        if constexpr (G < 2) {
            tr_rd<(16 * S) * 256>(aadr[G][SL], fa[SET][G][0]);
            tr_rd<(16 * S + 4) * 256>(aadr[G][SL], fa[SET][G][1]);
        } else {
            constexpr int Y = (G - 2) >> 1, B = (G - 2) & 1;        // (the second tap row: the same pixels one patch row further down)
            tr_rd<(ppix + Y * RP) * 128>(badr[B][SL], fb[SET][Y][B][0]);
            tr_rd<(ppix + Y * RP + 4) * 128>(badr[B][SL], fb[SET][Y][B][1]);
        }
    };
    auto read_kstep = [&](auto slot_, auto s_) {             // all NR reads of k-step S of the stage in ring slot SL
        [&]<int... G>(std::integer_sequence<int, G...>) {
            (read_frag(slot_, s_, std::integral_constant<int, G>{}), ...);
        }(std::make_integer_sequence<int, 2 + 2 * NKY>{});
    };

    // ---- prologue: stages 0..NSTAGE-2, the fragments of k-steps 0 and 1 of stage 0
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s) { stage(tile_state(), s, 0, NOPS); advance(); }
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    read_kstep(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    read_kstep(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
    stamp(1);

    // one step on ring slot SL; more: a step follows
    auto body = [&](int i, auto slot_, const bool more) {
        constexpr int SL = decltype(slot_)::value, NSL = (SL + 1) % NSTAGE, DSL = (SL + NSTAGE - 1) % NSTAGE;
        const TS ts = tile_state();              // the tile staged during this step: NSTAGE-1 ahead, into the slot stage i-1 has left
        if constexpr (NSTAGE == 4) {
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        auto kstep = [&](auto s_) {
            constexpr int S = decltype(s_)::value, SET = S & 1;
            // fragments of this k-step have landed: the reads issued after them are those of the next k-step (8), if any
            if (!no_read) { if (S < 3 || more) wait_lgkm<NR>(); else wait_lgkm<0>(); }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    pin(fa[SET][t][e]);
#pragma unroll
                    for (int y = 0; y < NKY; ++y) pin(fb[SET][y][t][e]);
                }
            __builtin_amdgcn_sched_barrier(0);
            auto mfma = [&](auto a_, auto y_, auto b_) {
                constexpr int A = decltype(a_)::value, Y = decltype(y_)::value, B = decltype(b_)::value;
                const bf16x8 av = __builtin_shufflevector(fa[SET][A][0], fa[SET][A][1], 0, 1, 2, 3, 4, 5, 6, 7);
                const bf16x8 bv = __builtin_shufflevector(fb[SET][Y][B][0], fb[SET][Y][B][1], 0, 1, 2, 3, 4, 5, 6, 7);
                if (!no_mfma) acc[Y][A][B] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[Y][A][B], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            // two k-steps ahead, into the registers just consumed
            auto rd = [&](auto g_) {
                if constexpr (S < 2) read_frag(slot_, std::integral_constant<int, S + 2>{}, g_);
                else if (more) read_frag(std::integral_constant<int, NSL>{}, std::integral_constant<int, S - 2>{}, g_);
                __builtin_amdgcn_sched_barrier(0);
            };
            // this step's DMA, spread over the k-steps (three stages: over the two behind the barrier)
            auto dma = [&]() {
                if constexpr (NSTAGE == 4) stage(ts, DSL, (S * NOPS) / 4, ((S + 1) * NOPS) / 4);
                else if constexpr (S >= 2) stage(ts, DSL, ((S - 2) * NOPS) / 2, ((S - 1) * NOPS) / 2);
                __builtin_amdgcn_sched_barrier(0);
            };
            using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
            using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
            if constexpr (NKY == 1) {
                // the k-step's other instructions in the shadows of its own MFMAs (see conv_patch2.hip): a fragment group is
                // read again as soon as the last MFMA that takes its old value has been issued
                mfma(I0{}, I0{}, I0{}); dma(); mfma(I1{}, I0{}, I0{}); rd(I2{}); mfma(I0{}, I0{}, I1{}); rd(I0{}); mfma(I1{}, I0{}, I1{}); rd(I1{}); rd(I3{});
            } else {
                mfma(I0{}, I0{}, I0{}); dma(); mfma(I1{}, I0{}, I0{}); rd(I2{}); mfma(I0{}, I0{}, I1{}); mfma(I1{}, I0{}, I1{}); rd(I3{});
                mfma(I0{}, I1{}, I0{}); mfma(I1{}, I1{}, I0{}); rd(std::integral_constant<int, 4>{});
                mfma(I0{}, I1{}, I1{}); rd(I0{}); mfma(I1{}, I1{}, I1{}); rd(I1{}); rd(std::integral_constant<int, 5>{});
            }
        };
        kstep(std::integral_constant<int, 0>{});
        kstep(std::integral_constant<int, 1>{});
        if constexpr (NSTAGE == 3) {
            wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        kstep(std::integral_constant<int, 2>{});
        kstep(std::integral_constant<int, 3>{});
        advance();
        if constexpr (NSTAGE == 4) {
            if (no_dma) wait_vm<0>();
            else if (last_pass) wait_vm<NOPS>();
            else wait_vm<NOPS - 1>();
        }
    };
    for (int i = 0; i < nsteps; i += NSTAGE) {
        body(i, std::integral_constant<int, 0>{}, true);
        body(i + 1, std::integral_constant<int, 1>{}, true);
        if constexpr (NSTAGE == 4) {
            body(i + 2, std::integral_constant<int, 2>{}, true);
            body(i + 3, std::integral_constant<int, 3>{}, i + 4 < nsteps);
        } else {
            body(i + 2, std::integral_constant<int, 2>{}, i + 3 < nsteps);
        }
    }
    wait_vm<0>();                                // (the pieces issued behind the slice's end)

    stamp(2);
    // ---- epilogue: lane holds D[i = 8 g + 4 lh' + r][c = lane & 31] of tile (a, b): rows = direct channels, columns = gathered
    const int col = lane & 31, rh = lane >> 5;
    if (p.slab) {
#pragma unroll
        for (int y = 0; y < NKY; ++y) {
            const int tile = (it * 4 + ky + y * KYD) * p.ctiles + ct;
            float* dst = p.slab + ((size_t)split * (p.itiles * 4 * p.ctiles) + tile) * (128 * 4 * 64);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int i = wi * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * rh;
                        dst[(i * 4 + kx) * 64 + b * 32 + col] = acc[y][a][b][e];
                    }
        }
        if (p.stamps) {
            __builtin_amdgcn_s_waitcnt(0);
            stamp(3);
            if (tid == 0) p.stamps[(size_t)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_memrealtime();
        }
        return;
    }
#pragma unroll
    for (int y = 0; y < NKY; ++y)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = i0 + wi * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * rh;
                    atomicAdd(p.dw + ((size_t)i * 16 + (ky + y * KYD) * 4 + kx) * p.gp + c0 + b * 32 + col, acc[y][a][b][e]);
                }
}

// dw[i][tap][c] += sum over slices of the partial tiles, slices in ascending order (bitwise reproducible).
__global__ __launch_bounds__(256) void k_wgrad_patch_reduce(const WgpK p) {
    const int ntiles = p.itiles * 4 * p.ctiles;
    const size_t per = 128 * 4 * 64;
    const size_t e4 = (size_t)blockIdx.x * 256 + threadIdx.x;          // one float4 (4 gathered channels) per thread
    if (e4 >= (size_t)ntiles * per / 4) return;
    const size_t e = e4 * 4;
    const int tile = (int)(e / per), rem = (int)(e - (size_t)tile * per);
    const int c = rem & 63, kx = (rem >> 6) & 3, i = rem >> 8;
    const int ct = tile % p.ctiles, ky = (tile / p.ctiles) & 3, it = tile / (4 * p.ctiles);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const float* src = p.slab + e;
    for (int k = 0; k < p.msplit; ++k) s += *(const f32x4*)(src + (size_t)k * ntiles * per);
    float* dst = p.dw + ((size_t)(it * 128 + i) * 16 + ky * 4 + kx) * p.gp + ct * 64 + c;
    if (p.fresh) *(f32x4*)dst = s;               // (dw holds zeros: no need to read it)
    else *(f32x4*)dst = *(const f32x4*)dst + s;
}

template <int STRIDE, int TX, int ABL = 0, int NKY = 1>
int launch_wgp(const WgpK& k, hipStream_t st) {
    constexpr int TY = 64 / TX;
    constexpr int PLW = STRIDE == 1 ? TX + 3 : TX + 1;
    constexpr int RP = ((STRIDE == 1 ? PLW : 2 * PLW) + 3) & ~3;
    constexpr int PP = ((TY + NKY - 1) * RP * 8 + 511) / 512;
    constexpr int STAGE = DIR_BYTES + PP * 8192;
    constexpr int LDS = wgp_stages(STAGE) * STAGE;
    static std::atomic<bool> attr_done{false};   // (launches may come from several host threads)
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)k_wgrad_patch<STRIDE, TX, ABL, NKY>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_done = true;
    }
    hipLaunchKernelGGL((k_wgrad_patch<STRIDE, TX, ABL, NKY>), dim3(k.nwg), dim3(512), LDS, st, k);
    PFA_LAUNCHED("k_wgrad_patch");
    return 0;
}

}  // namespace

// The launches the patch kernel takes: bf16, direct channels in whole 128-tiles per source, ONE gathered source in whole
// 64-tiles, no pad channels on either side, a direct map at least 16 wide and 64 positions large, tensors below 2 GiB.
// Reduction slices: enough for ~256 workgroups, at least 4 steps each.  pfa_set_knob(4, 7) turns the route off.
// Tap rows per workgroup: 1; 2 with pfa_set_knob(13, 2).  Two rows cut the LDS-DMA instructions per MFMA by 40 % and the
// fragment reads by 25 %, and a launch of the same workgroup count has twice the slices.  Alone, c4 goes 125 -> 106 us and
// e4 / d4 49 / 78 -> 43 / 74 us (four-stage ring); on the 2 x 32 tile at stride 2 (three stages, barrier in mid-step) the
// step's DMA and MFMAs hardly overlap (3700 cycles per double step against 2330 + 1980 alone) and nothing is gained.
// Inside the training step: 3.15 ms against 3.13 (knob 13 = 2), 3.13 against 3.14 (knob 13 = 3) - the partial tiles of
// the extra slices cost what the loop saves, so one row stays the default.
// knob 13 = 3: two rows where the ring keeps four stages (stride 1, and the 4 x 16 tile at stride 2)
static int pfa_wgrad_patch_nky(int stride, int tx) {
    const int k = pfa_knob(13);
    return k == 2 ? 2 : k == 3 ? (stride == 1 || tx == 16 ? 2 : 1) : 1;
}

bool pfa_wgrad_patch_plan(const pfa_wgrad_args* a, int* slices, int* tx_out, int* steps_out) {
    const int k4 = pfa_knob(4);
    if (k4 == 7 || k4 == 1 || k4 == 2 || k4 == 3) return false;              // explicit old-kernel variants / route off
    if (a->dtype != PFA_BF16 || a->g2p != 0) return false;
    if (a->d1p % 128 || a->d2p % 128 || a->g1p % 64) return false;
    if (a->d1r != a->d1p || a->d2r != a->d2p || a->g1r != a->g1p) return false;
    if (a->wq < 16 || (long long)a->hq * a->wq < 64) return false;
    const long long dbytes = (long long)a->nb * a->hq * a->wq * (a->d1p > a->d2p ? a->d1p : a->d2p) * 2;
    const long long gbytes = ((long long)a->nb * a->hg * a->wg + a->wg + 2) * a->g1p * 2;     // (+ the one row and pixel the per-lane offsets are shifted by)
    if (dbytes >= (1ll << 31) || gbytes >= (1ll << 31)) return false;
    const int tx = a->wq >= 24 ? 32 : 16, ty = 64 / tx;
    const int steps = a->nb * ((a->hq + ty - 1) / ty) * ((a->wq + tx - 1) / tx);
    const int tiles = ((a->d1p + a->d2p) / 128) * 4 * (a->g1p / 64);           // output tiles
    const int nky = pfa_wgrad_patch_nky(a->stride, tx);
    const int wgs = tiles / nky;                                                // workgroups of one reduction slice
    // workgroups a launch aims for (knob 8).  In-situ sweep: 512 -> 3.39 ms per step, 256 -> 3.35, 192 -> 3.30, 128 -> 3.28,
    // 96 -> 3.30, 64 -> 3.36: these launches share the GPU with the main stream, and fewer slices are fewer partial tiles
    const int want = pfa_knob(8) > 0 ? pfa_knob(8) : 128;
    int ms = (want + wgs - 1) / wgs;
    if (ms > steps / 4) ms = steps / 4;
    if (ms < 1) ms = 1;
    const int sper = (steps + ms - 1) / ms;
    ms = (steps + sper - 1) / sper;
    // too few workgroups, or so few output tiles that the launch is all partial-tile traffic (the 64 -> 128 channel layers:
    // 4 tiles x 64 slices, 38 us on wgrad.hip against 46 us here); knob 4 = 8: whenever the geometry fits
    if (k4 != 8 && ((long long)wgs * ms < 96 || tiles < 8)) return false;
    if (slices) *slices = ms;
    if (tx_out) *tx_out = tx;
    if (steps_out) *steps_out = steps;
    return true;
}

size_t pfa_wgrad_patch_slab_floats(const pfa_wgrad_args* a, int slices) {
    return (size_t)slices * ((a->d1p + a->d2p) / 128) * 4 * (a->g1p / 64) * (128 * 4 * 64);
}

int pfa_wgrad_patch(const pfa_wgrad_args* a, hipStream_t st, const char** name) {
    int ms, tx, steps;
    PFA_REQUIRE(pfa_wgrad_patch_plan(a, &ms, &tx, &steps), "pfa_wgrad_patch: geometry not supported");
    WgpK k;
    k.dir1 = (const char*)a->dir1; k.dir2 = (const char*)a->dir2; k.gat = (const char*)a->gat1;
    k.dw = a->dw; k.slab = a->slab; k.fresh = a->fresh;
    k.d1p = a->d1p; k.dp = a->d1p + a->d2p; k.gp = a->g1p;
    k.nb = a->nb; k.hq = a->hq; k.wq = a->wq; k.hg = a->hg; k.wg = a->wg;
    const int ty = 64 / tx;
    k.tiles_x = (a->wq + tx - 1) / tx; k.tiles_y = (a->hq + ty - 1) / ty;
    k.nsteps = steps; k.msplit = ms; k.sper = (steps + ms - 1) / ms;
    k.itiles = k.dp / 128; k.ctiles = k.gp / 64;
    const int nky = pfa_wgrad_patch_nky(a->stride, tx);
    k.nwg = k.itiles * (4 / nky) * k.ctiles * ms;
    k.stamps = (g_pfa_dbgbuf && g_pfa_dbgbuf_bytes >= (size_t)k.nwg * 64) ? (unsigned long long*)g_pfa_dbgbuf : nullptr;
    int rc;
    if (a->stride == 2 && tx == 32 && nky == 2 && pfa_knob(7) >= 11 && pfa_knob(7) <= 17) {   // developer ablations (results are wrong)
        switch (pfa_knob(7) - 10) {
            case 1: rc = launch_wgp<2, 32, 1, 2>(k, st); break;
            case 2: rc = launch_wgp<2, 32, 2, 2>(k, st); break;
            case 3: rc = launch_wgp<2, 32, 3, 2>(k, st); break;
            case 4: rc = launch_wgp<2, 32, 4, 2>(k, st); break;
            case 5: rc = launch_wgp<2, 32, 5, 2>(k, st); break;
            case 6: rc = launch_wgp<2, 32, 6, 2>(k, st); break;
            default: rc = launch_wgp<2, 32, 7, 2>(k, st); break;
        }
    } else
    if (a->stride == 2 && tx == 32 && pfa_knob(7) >= 11 && pfa_knob(7) <= 17) {        // developer ablations (results are wrong)
        switch (pfa_knob(7) - 10) {
            case 1: rc = launch_wgp<2, 32, 1>(k, st); break;
            case 2: rc = launch_wgp<2, 32, 2>(k, st); break;
            case 3: rc = launch_wgp<2, 32, 3>(k, st); break;
            case 4: rc = launch_wgp<2, 32, 4>(k, st); break;
            case 5: rc = launch_wgp<2, 32, 5>(k, st); break;
            case 6: rc = launch_wgp<2, 32, 6>(k, st); break;
            default: rc = launch_wgp<2, 32, 7>(k, st); break;
        }
    } else
    if (nky == 2) {
        if (a->stride == 1) rc = tx == 32 ? launch_wgp<1, 32, 0, 2>(k, st) : launch_wgp<1, 16, 0, 2>(k, st);
        else rc = tx == 32 ? launch_wgp<2, 32, 0, 2>(k, st) : launch_wgp<2, 16, 0, 2>(k, st);
    } else
    if (a->stride == 1) rc = tx == 32 ? launch_wgp<1, 32>(k, st) : launch_wgp<1, 16>(k, st);
    else rc = tx == 32 ? launch_wgp<2, 32>(k, st) : launch_wgp<2, 16>(k, st);
    if (name) *name = a->stride == 1 ? (tx == 32 ? "k_wgrad_patch<s1,2x32>" : "k_wgrad_patch<s1,4x16>")
                                     : (tx == 32 ? "k_wgrad_patch<s2,2x32>" : "k_wgrad_patch<s2,4x16>");
    if (rc) return rc;
    if (k.slab) {
        const size_t quads = (size_t)k.itiles * 4 * k.ctiles * (128 * 4 * 64) / 4;
        hipLaunchKernelGGL(k_wgrad_patch_reduce, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, k);
        PFA_LAUNCHED("k_wgrad_patch_reduce");
    }
    return 0;
}
