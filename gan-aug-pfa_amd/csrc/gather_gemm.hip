// Implicit-GEMM gather kernel on MFMA: Conv2d / ConvTranspose2d forward and input gradients.
//
// Replaces nn.Conv2d (models.py:177,223,230,238,243) and nn.ConvTranspose2d (models.py:184,189,194)
// forward, and the input-gradient halves of their autograd backward (train_gan.py:62,70).
//
// GEMM view (no im2col buffer): rows m = output positions, columns n = output channels,
// K = (tap, channel) with the channel contiguous in NHWC memory.  One K-step is 128 bytes per row
// (64 bf16 / 32 f32 channels) = 8 chunks of 16 B; chunk f of the flattened K axis is
// (tap = f / chunks_per_tap, channel chunk = f % chunks_per_tap), so layers with few channels
// (3 -> padded 8) pack several taps into one K-step instead of wasting it.
//
// Data path: every lane owns one 16-B chunk slot of the LDS tile and fetches it with
// global_load_lds_dwordx4 straight from HBM/L2 into LDS (per-lane source address = the gather;
// out-of-image taps, K tails and missing weight rows point at a zero page).  The LDS image is
// lane-linear, so the bank-conflict swizzle (chunk ^= (row>>1)&7) is applied to the SOURCE chunk
// index and again on the ds_read_b128 fragment reads.  Two LDS stages, one barrier per K-step.
// MFMA orientation is D[n][m] (weights = A operand, activations = B operand): a lane then holds
// 4 consecutive output channels of one pixel -> 8/16-byte NHWC stores.
#include "gather_common.h"

// developer knobs (pfa_set_knob): [0] gather tile config, [1] ablation bits, ... (include/pfa_hip.h).  Relaxed atomics: the
// library may be entered from several host threads / streams at once; a knob is a measurement switch, never a result.
std::atomic<int> g_knob[16];
static thread_local const char* g_last_gather_name = "";

// Optional per-launch timing (pfa_gather_timing): HIP events recorded on the launch stream right before and right after the
// MFMA kernel of a pfa_gather_gemm / pfa_wgrad call (a split-K call's finish kernel and a weight gradient's reduce kernel
// are outside the bracket).  The store is bounded: launches beyond TIMED_MAX while timing is on are not recorded.
#include <mutex>
#include <vector>
struct TimedLaunch { hipEvent_t e0, e1; const char* name; double flop, bytes; int splitk, nwg; };
static std::mutex g_timed_mu;
static std::vector<TimedLaunch> g_timed;
static std::atomic<bool> g_timing{false};
constexpr size_t TIMED_MAX = 1 << 16;

bool pfa_timing_begin(hipStream_t st, hipEvent_t* e0, hipEvent_t* e1) {
    if (!g_timing.load(std::memory_order_relaxed)) return false;
    {
        std::lock_guard<std::mutex> lk(g_timed_mu);
        if (g_timed.size() >= TIMED_MAX) return false;
    }
    if (hipEventCreate(e0) != hipSuccess) return false;
    if (hipEventCreate(e1) != hipSuccess) { (void)hipEventDestroy(*e0); return false; }
    (void)hipEventRecord(*e0, st);
    return true;
}
void pfa_timing_end(hipStream_t st, hipEvent_t e0, hipEvent_t e1, const char* name, double flop, double bytes, int splitk, int nwg) {
    (void)hipEventRecord(e1, st);
    std::lock_guard<std::mutex> lk(g_timed_mu);
    g_timed.push_back(TimedLaunch{e0, e1, name, flop, bytes, splitk, nwg});
}

namespace {

template <typename T, int BM, int BN, int WM, int WN, int EARLY>
__global__ __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu(2))) void k_gather_gemm(const GatherK p) {
    constexpr int CH = ElemTraits<T>::CH;
    constexpr int NT = 64 * WM * WN;                       // threads
    constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE = A_BYTES + W_BYTES;
    constexpr int RSTEP = NT / 8;                          // tile rows covered by one staging pass
    constexpr int AROWS = BM / RSTEP;
    constexpr int WROWS = BN >= RSTEP ? BN / RSTEP : 1;
    constexpr int WWAVES = BN >= RSTEP ? NT / 64 : BN / 8; // waves that take part in staging the weight tile
    static_assert(TM % 16 == 0 && TN % 16 == 0 && BM % RSTEP == 0 && RSTEP % 16 == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    // Work list: each XCD (private L2) takes one contiguous chunk of it, so the slowest index decides what an XCD shares.
    //   order 0: n-tile, m-tile, K slice, parity class (slowest): an XCD holds one class / slice - weights fetched once,
    //            the input once per class or slice;
    //   order 1: n-tile, K slice, class, m-tile (slowest): an XCD holds a range of pixels with ALL their classes and
    //            slices - the input is fetched once, the weights once per XCD;
    //   order 2: m-tile, K slice, class, n-tile (slowest): an XCD holds a range of output channels - the weights are
    //            fetched once, the input once per XCD.
    // The host picks the order with the fewest fetched bytes (choose_order); results do not depend on it.
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
    const int m0 = mt * BM, n0 = nt * BN;
    const char* zero = (const char*)g_pfa_zero_page;

    // ---- staging roles: this thread fills chunk slot `cs` of rows r0 + 32*i
    const int cs = tid & 7, r0 = tid >> 3;
    const int g = cs ^ ((r0 >> 1) & 7);          // source chunk (the swizzle, on the source side)
    const int hw = p.hq * p.wq;
    int a_yx[AROWS], a_img[AROWS];
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
        const int m = m0 + r0 + RSTEP * i;
        a_img[i] = -1;
        a_yx[i] = 0;
        if (m < p.M) {
            const int ni = m / hw, rem = m - ni * hw;
            const int qy = rem / p.wq, qx = rem - qy * p.wq;
            int by, bx;
            if (p.mode == PFA_CONV_S2) { by = 2 * qy - 1; bx = 2 * qx - 1; }
            else if (p.mode == PFA_CONV_S1) { by = qy - 1; bx = qx - 1; }
            else if (p.mode == PFA_TCONV_S2) { by = qy + py; bx = qx + px; }
            else { by = qy + 1; bx = qx + 1; }
            a_yx[i] = (by << 16) | (bx & 0xffff);
            a_img[i] = ni * p.hs * p.ws;
        }
    }
    const int sgn = (p.mode >= PFA_TCONV_S2) ? -1 : 1;

    // ---- generic staging: every lane derives (tap, channel chunk) from its flattened chunk index each K-step.
    //      Used when a K-step can span several taps (few channels: the 3/6-channel image layers).
    auto stage_narrow = [&](int ks, int buf) {
        char* abase = smem + buf * STAGE;
        char* wbase = abase + A_BYTES;
        const int f = ks * 8 + g;
        const bool kval = f < p.ktot;
        int tap, c8;
        if (p.cpt_shift >= 0) { tap = f >> p.cpt_shift; c8 = f & (p.cpt - 1); }
        else { tap = f / p.cpt; c8 = f - tap * p.cpt; }
        const int c = c8 * CH;
        int ty, tx, widx;
        if (p.mode == PFA_TCONV_S2) {
            ty = tap >> 1; tx = tap & 1;
            widx = (1 - py + 2 * ty) * 4 + (1 - px + 2 * tx);
        } else {
            ty = tap >> 2; tx = tap & 3;
            widx = tap;
        }
        const bool first = c < p.c1;
        const char* sbase = first ? p.src1 : p.src2;
        const int csrc = first ? p.c1 : p.c2, cc = first ? c : c - p.c1;
        const long long zoff = zero - sbase;       // select an OFFSET, not a pointer: stays branch-free
#pragma unroll
        for (int i = 0; i < AROWS; ++i) {
            const int y = (a_yx[i] >> 16) + sgn * ty;
            const int x = (int)(short)(a_yx[i] & 0xffff) + sgn * tx;
            const bool ok = kval & (a_img[i] >= 0) & ((unsigned)y < (unsigned)p.hs) & ((unsigned)x < (unsigned)p.ws);
            const long long off = ((long long)(a_img[i] + y * p.ws + x) * csrc + cc) * (long long)sizeof(T);
            const char* gp = sbase + (ok ? off : zoff);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gp), LDS_PTR(abase + (i * NT + wave * 64) * 16), 16, 0, 0);
        }
        if (wave < WWAVES) {
            const long long zoffw = zero - p.w;
#pragma unroll
            for (int i = 0; i < WROWS; ++i) {
                const int n = n0 + r0 + RSTEP * i;
                const bool ok = kval & (n < p.n);
                const long long off = ((long long)(n * 16 + widx) * p.ctot + c) * (long long)sizeof(T);
                const char* gp = p.w + (ok ? off : zoffw);
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gp), LDS_PTR(wbase + (i * NT + wave * 64) * 16), 16, 0, 0);
            }
        }
    };

    // ---- wide staging (channel counts that are multiples of one K-step: every MFMA-heavy layer).
    //      A K-step lies inside one (tap, source) segment, so the row pointers are rebuilt only when the
    //      segment changes and otherwise just advance by 128 bytes: ~2 VALU ops per row per K-step.
    // Per row: a 32-bit byte offset from the segment's base pointer (tensors are < 4 GiB) and one validity bit; invalid
    // rows (padding taps, rows >= M, weight rows >= N) read the zero page and do not advance.
    unsigned a_off[AROWS], w_off[WROWS];
    unsigned okmask = 0;                 // bit i: activation row i valid; bit 16+i: weight row i valid
    const char* seg_base = p.src1;
    int seg_tap = 0, seg_part = 0, seg_left = 0;
    auto seg_setup = [&](int kc) {      // kc: K-step offset inside the segment (non-zero only for a split-K start)
        int ty, tx, widx;
        if (p.mode == PFA_TCONV_S2) {
            ty = seg_tap >> 1; tx = seg_tap & 1;
            widx = (1 - py + 2 * ty) * 4 + (1 - px + 2 * tx);
        } else {
            ty = seg_tap >> 2; tx = seg_tap & 3;
            widx = seg_tap;
        }
        seg_base = seg_part ? p.src2 : p.src1;
        const int csrc = seg_part ? p.c2 : p.c1;
        const int cc = (kc * 8 + g) * CH;
        okmask = 0;
#pragma unroll
        for (int i = 0; i < AROWS; ++i) {
            const int y = (a_yx[i] >> 16) + sgn * ty;
            const int x = (int)(short)(a_yx[i] & 0xffff) + sgn * tx;
            const bool ok = (a_img[i] >= 0) & ((unsigned)y < (unsigned)p.hs) & ((unsigned)x < (unsigned)p.ws);
            a_off[i] = (unsigned)(((a_img[i] + y * p.ws + x) * csrc + cc) * (int)sizeof(T));
            okmask |= ok ? (1u << i) : 0u;
        }
        const int cw = (seg_part ? p.c1 : 0) + cc;
#pragma unroll
        for (int i = 0; i < WROWS; ++i) {
            const int n = n0 + r0 + RSTEP * i;
            w_off[i] = (unsigned)((((n * 16 + widx) * p.ctot) + cw) * (int)sizeof(T));
            okmask |= (n < p.n) ? (1u << (16 + i)) : 0u;
        }
        seg_left = (seg_part ? p.spt - p.sp1 : p.sp1) - kc;
    };
    // piece j of a wide stage: activation row pass j (j < AROWS) or weight row pass j - AROWS
    auto stage_wide_piece = [&](int buf, int j) {
        char* abase = smem + buf * STAGE;
        char* wbase = abase + A_BYTES;
        if (j < AROWS) {
            const int i = j;
            const bool ok = (okmask >> i) & 1u;
            const char* gp = ok ? seg_base + a_off[i] : zero;
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gp), LDS_PTR(abase + (i * NT + wave * 64) * 16), 16, 0, 0);
            a_off[i] += 128u;
        } else if (wave < WWAVES) {
            const int i = j - AROWS;
            const bool ok = (okmask >> (16 + i)) & 1u;
            const char* gp = ok ? p.w + w_off[i] : zero;
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gp), LDS_PTR(wbase + (i * NT + wave * 64) * 16), 16, 0, 0);
            w_off[i] += 128u;
        }
    };
    auto stage_wide_next = [&]() {
        if (--seg_left == 0) {          // next segment: the other source of this tap, or the next tap
            if (seg_part == 0 && p.sp1 < p.spt) seg_part = 1;
            else { seg_part = 0; ++seg_tap; }
            seg_setup(0);               // (one harmless setup past the last tap: pointers are never used)
        }
    };
    auto stage_wide = [&](int buf) {
#pragma unroll
        for (int j = 0; j < AROWS + WROWS; ++j) stage_wide_piece(buf, j);
        stage_wide_next();
    };
    auto stage = [&](int ks, int buf) {
        if (p.dbg & 8) return;
        if (p.wide) stage_wide(buf);
        else stage_narrow(ks, buf);
    };

    f32x4 acc[FN][FM];
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int lrow = lane & 15, kk = lane >> 4;
    // A K-step in three phases: (1) all 2*(FN+FM) fragment reads, (2) issue of the next tile's DMA (its address
    // arithmetic runs under the LDS latency), (3) the MFMAs, which the DMA then overlaps.
    // Fragments of one K-step: both 64-byte halves when they fit (64x64 per-wave tiles), else half by half.
    constexpr bool ALL_AT_ONCE = (FN + FM) <= 8;
    constexpr int NH = ALL_AT_ONCE ? 2 : 1;
    struct Frags {
        typename Mma<T>::frag wf[NH][FN], af[NH][FM];
    };
    auto read_half = [&](Frags& f, int buf, int h, int slot) {
        const char* abase = smem + buf * STAGE;
        const char* wbase = abase + A_BYTES;
        const int ch = kk + 4 * h;
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            const int row = wn * TN + a * 16 + lrow;
            f.wf[slot][a] = *(const typename Mma<T>::frag*)(wbase + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int b = 0; b < FM; ++b) {
            const int row = wm * TM + b * 16 + lrow;
            f.af[slot][b] = *(const typename Mma<T>::frag*)(abase + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4));
        }
    };
    auto mfma_half = [&](const Frags& f, int slot) {
#pragma unroll
        for (int a = 0; a < FN; ++a)
#pragma unroll
            for (int b = 0; b < FM; ++b) acc[a][b] = Mma<T>::run(f.wf[slot][a], f.af[slot][b], acc[a][b]);
    };
    Frags fA;
    // 2-stage path, phase 1: fragment reads; phase 2 (caller): issue of the next tile's DMA; phase 3: MFMAs under that DMA
    auto read_frags = [&](int buf) {
        if (p.dbg & 4) return;
        read_half(fA, buf, 0, 0);
        if constexpr (ALL_AT_ONCE) read_half(fA, buf, 1, 1);
        __builtin_amdgcn_sched_barrier(0);      // keep the reads ahead (the scheduler sinks them otherwise)
    };
    auto mfma_frags = [&](int buf) {
        if (p.dbg & 4) return;
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(fA, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ALL_AT_ONCE) {
            mfma_half(fA, 1);
        } else {
            read_half(fA, buf, 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_half(fA, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    const int ks0 = split * p.kper;
    const int ks1 = min(p.ksteps, ks0 + p.kper);
    if (p.wide) {
        seg_tap = ks0 / p.spt;
        const int within = ks0 - seg_tap * p.spt;
        seg_part = within >= p.sp1 ? 1 : 0;
        seg_setup(seg_part ? within - p.sp1 : within);
    }
    if constexpr (!EARLY) {
        if (ks0 < ks1) stage(ks0, 0);
        for (int ks = ks0; ks < ks1; ++ks) {
            const int buf = (ks - ks0) & 1;
            __syncthreads();          // waits vmcnt(0): stage ks has landed; everyone is done reading buf^1
            read_frags(buf);
            if (ks + 1 < ks1) stage(ks + 1, buf ^ 1);
            mfma_frags(buf);
        }
    } else {
        // per-lane byte offsets of this wave's first weight / activation fragment inside a stage, per 64-byte half; the
        // other fragments sit 16 rows = 2048 bytes further on (the swizzle term depends on row & 15 only)
        typedef typename Mma<T>::frag frag_t;
        const unsigned sw = (lrow >> 1) & 7;
        const unsigned lds0 = (unsigned)(size_t)LDS_PTR(smem);
        unsigned wfb[2], afb[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            wfb[h] = lds0 + A_BYTES + (wn * TN + lrow) * 128 + (((kk + 4 * h) ^ sw) << 4);
            afb[h] = lds0 + (wm * TM + lrow) * 128 + (((kk + 4 * h) ^ sw) << 4);
        }
        auto read_half_raw = [&](Frags& f, int buf, int h, int slot) {
            ReadFrags<frag_t, FN>::run(f.wf[slot], wfb[h] + buf * STAGE);
            ReadFrags<frag_t, FM>::run(f.af[slot], afb[h] + buf * STAGE);
        };
        auto pin_half = [&](Frags& f, int slot) {
#pragma unroll
            for (int a = 0; a < FN; ++a) pin(f.wf[slot][a]);
#pragma unroll
            for (int b = 0; b < FM; ++b) pin(f.af[slot][b]);
        };
        if constexpr (EARLY == 4) {
            // The ring of EARLY == 3, software-pipelined at half-step granularity: the barrier sits in the MIDDLE of a
            // step.  Step i: issue the reads of (i, second half) -> MFMAs of (i, first half), whose fragments were
            // read during step i-1, with the DMA pieces of stage i+2 between them -> retire every read of stage i,
            // wait (counted) for stage i+1, barrier -> issue the reads of (i+1, first half) into the other register
            // set -> MFMAs of (i, second half).  No MFMA ever waits for an LDS read issued right behind a barrier.
            // Buffer reuse: stage (i+2)%3 held step i-1, whose reads every wave retired before the barrier of step i-1.
            static_assert(ALL_AT_ONCE, "ring variant needs both K halves in registers");
            constexpr int NPIECE = AROWS + WROWS, MF0 = FN * FM;
            const int nks = ks1 - ks0;
            Frags fB;
            if (nks > 0) stage_wide(0);
            if (nks > 1) stage_wide(1);
            if (nks > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (nks > 0) read_half_raw(fA, 0, 0, 0);
            auto body = [&](int i, Frags& cur, Frags& nxt) {
                const int buf = i % 3, nbuf = (i + 2) % 3;
                const bool sp = i + 2 < nks;
                read_half_raw(cur, buf, 1, 1);
                PFA_WAIT_LGKM(FN + FM);                      // the first half (issued a half-step ago) is here
                pin_half(cur, 0);
#pragma unroll
                for (int a = 0; a < FN; ++a)
#pragma unroll
                    for (int b = 0; b < FM; ++b) {
                        acc[a][b] = Mma<T>::run(cur.wf[0][a], cur.af[0][b], acc[a][b]);
                        const int idx = a * FM + b;
                        const int lo = (idx * NPIECE) / MF0, hi = ((idx + 1) * NPIECE) / MF0;
                        if (hi > lo) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (sp) {
#pragma unroll
                                for (int j = lo; j < hi; ++j) stage_wide_piece(nbuf, j);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                if (sp) stage_wide_next();
                __builtin_amdgcn_sched_barrier(0);
                PFA_WAIT_LGKM(0);                            // every read of stage i has retired
                pin_half(cur, 1);
                if (i + 1 < nks) {
                    if (sp) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    read_half_raw(nxt, (i + 1) % 3, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                mfma_half(cur, 1);
                __builtin_amdgcn_sched_barrier(0);
            };
            int i = 0;
            for (; i + 1 < nks; i += 2) {
                body(i, fA, fB);
                body(i + 1, fB, fA);
            }
            if (i < nks) body(i, fA, fB);
        } else if constexpr (EARLY == 5 || EARLY == 6) {
            // The 3-stage ring with the two waves of every SIMD in OPPOSITE phases (the guide's staggered 8-phase idea):
            // waves 0-3 (one per SIMD) and waves 4-7 (the second wave of each SIMD) alternate between a read phase
            // (fragment reads of stage i + the DMA pieces of stage i+2 + the counted wait for stage i+1) and an MFMA
            // phase, swapping roles at every barrier; the late group runs one barrier behind.  The MFMA pipe of a SIMD
            // then always has one wave with all its fragments in registers, and LDS reads / DMA issue of the other
            // wave run under it.  EARLY == 5: one phase pair per K-step (32 MFMAs); 6: one per 32-wide half (16 MFMAs).
            // Barrier numbering: early group B1(i) = #2i+1, B2(i) = #2i+2; late group #1 extra, B1(i) = #2i+2,
            // B2(i) = #2i+3 (per phase pair).  RAW: stage i+1 is waited for (own pieces, counted vmcnt) before B1 of the
            // last phase of step i; a wave reads it after its B2 of that phase, which the other group reaches only
            // after its own wait.  WAR: stage (i+2)%3 was read in step i-1, reads retired (lgkmcnt(0)) before B1 of its
            // phase; its DMA is issued after the issuing wave's B2 of that phase, which the other group reaches only
            // after retiring.
            static_assert(ALL_AT_ONCE && NT == 512, "staggered ring: eight waves, both K halves in registers");
            constexpr int NPIECE = AROWS + WROWS;
            constexpr int NPH = EARLY == 5 ? 1 : 2;
            const int nks = ks1 - ks0;
            const bool late = wave >= 4;
            const bool no_dma = p.dbg & 8, no_mfma = p.dbg & 4, no_read = p.dbg & 128;      // ablations
            if (nks > 0 && !no_dma) stage_wide(0);
            if (nks > 1 && !no_dma) stage_wide(1);
            if (nks > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();            // stage 0 is in LDS for everyone
            if (late) __builtin_amdgcn_s_barrier();
            for (int i = 0; i < nks; ++i) {
                const int buf = i % 3, nbuf = (i + 2) % 3;
                const bool sp = i + 2 < nks && !no_dma;
#pragma unroll
                for (int h = 0; h < NPH; ++h) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (!no_read || i == 0) {
                        if constexpr (NPH == 1) {
                            read_half_raw(fA, buf, 0, 0);
                            read_half_raw(fA, buf, 1, 1);
                        } else {
                            read_half_raw(fA, buf, h, 0);
                        }
                    }
                    if (sp) {
#pragma unroll
                        for (int j = h * NPIECE / NPH; j < (h + 1) * NPIECE / NPH; ++j) stage_wide_piece(nbuf, j);
                    }
                    if (h == NPH - 1) {
                        if (sp) {
                            stage_wide_next();
                            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory");
                        } else {
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    PFA_WAIT_LGKM(0);
                    pin_half(fA, 0);
                    if constexpr (NPH == 1) pin_half(fA, 1);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_s_setprio(1);
                    if (!no_mfma) {
                        mfma_half(fA, 0);
                        if constexpr (NPH == 1) mfma_half(fA, 1);
                    }
                    __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                }
            }
            if (!late) __builtin_amdgcn_s_barrier();
        } else if constexpr (EARLY == 3) {
            // Three LDS stages, the DMA queue never drained (cdna guide T3/T4): step i waits, with a COUNTED vmcnt, for
            // the pieces of stage i only - those of stage i+1 (issued during step i-1) stay in flight across the barrier -
            // then issues the pieces of stage i+2 between its first-half MFMAs.  Stage (i+2)%3 held step i-1, whose
            // reads every wave retired before its MFMAs of step i-1, i.e. before this step's barrier.  Wide staging
            // only (every wave issues exactly NPIECE instructions per stage: the counted wait relies on it); the host
            // selects this variant for wide launches (launch_gather).
            static_assert(ALL_AT_ONCE, "ring variant needs both K halves in registers");
            constexpr int NPIECE = AROWS + WROWS, MF0 = FN * FM;
            const int nks = ks1 - ks0;
            const bool no_dma = p.dbg & 8;
            if (nks > 0 && !no_dma) stage_wide(0);
            if (nks > 1 && !no_dma) stage_wide(1);
            for (int i = 0; i < nks; ++i) {
                const int buf = i % 3;
                if (i + 1 < nks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                bool sp = i + 2 < nks && !no_dma;
                const int nbuf = (i + 2) % 3;
                if (sp && (p.dbg & 32)) {            // ablation bit 32: the whole stage in one burst, before the reads
                    stage_wide(nbuf);
                    sp = false;
                }
                __builtin_amdgcn_sched_barrier(0);
#ifdef PFA_RING_DIAG      // diagnostic build (make CXXFLAGS+=-DPFA_RING_DIAG): knob 1 bits 128 / 4 drop the ring's fragment reads / MFMAs
                if (!(p.dbg & 128)) {
                    read_half_raw(fA, buf, 0, 0);
                    read_half_raw(fA, buf, 1, 1);
                }
#else
                read_half_raw(fA, buf, 0, 0);
                read_half_raw(fA, buf, 1, 1);
#endif
                PFA_WAIT_LGKM(FN + FM);
                pin_half(fA, 0);
#pragma unroll
                for (int a = 0; a < FN; ++a)
#pragma unroll
                    for (int b = 0; b < FM; ++b) {
#ifdef PFA_RING_DIAG
                        if (!(p.dbg & 4))
#endif
                        acc[a][b] = Mma<T>::run(fA.wf[0][a], fA.af[0][b], acc[a][b]);
                        const int idx = a * FM + b;
                        const int lo = (idx * NPIECE) / MF0, hi = ((idx + 1) * NPIECE) / MF0;   // pieces due after this MFMA
                        if (hi > lo) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (sp) {
#pragma unroll
                                for (int j = lo; j < hi; ++j) stage_wide_piece(nbuf, j);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                if (sp) stage_wide_next();
                __builtin_amdgcn_sched_barrier(0);
                PFA_WAIT_LGKM(0);
                pin_half(fA, 1);
#ifdef PFA_RING_DIAG
                if (!(p.dbg & 4))
#endif
                mfma_half(fA, 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
        if (ks0 < ks1) stage(ks0, 0);
        // EARLY == 2 (wide staging, both halves in registers): the DMA pieces of the next tile are spread over the
        // MFMAs of the first half instead of being issued in one burst - a wave stalls on each global_load_lds while
        // the CU's address unit drains the queue (that stall, not the landing, is what serialised DMA and MFMA).
        constexpr int NPIECE = AROWS + WROWS, MF0 = FN * FM;
        constexpr bool CAN_SPREAD = EARLY == 2 && ALL_AT_ONCE && MF0 >= NPIECE && MF0 % NPIECE == 0;
        const bool wide_ok = CAN_SPREAD && p.wide && !(p.dbg & 8);
        for (int ks = ks0; ks < ks1; ++ks) {
            const int buf = (ks - ks0) & 1;
            // stage ks has landed (vmcnt(0)); every wave retired its reads of buf^1 (PFA_WAIT_LGKM) before its MFMAs of
            // step ks-1, i.e. before arriving here, so the DMA below may overwrite buf^1 at once
            __syncthreads();
            const bool more = ks + 1 < ks1;
            if constexpr (CAN_SPREAD) {
                const bool sp = more && wide_ok;
                if (more && !wide_ok) stage(ks + 1, buf ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                read_half_raw(fA, buf, 0, 0);
                read_half_raw(fA, buf, 1, 1);
                PFA_WAIT_LGKM(FN + FM);
                pin_half(fA, 0);
#pragma unroll
                for (int a = 0; a < FN; ++a)
#pragma unroll
                    for (int b = 0; b < FM; ++b) {
                        acc[a][b] = Mma<T>::run(fA.wf[0][a], fA.af[0][b], acc[a][b]);
                        constexpr int PER = MF0 / NPIECE;
                        if ((a * FM + b) % PER == PER - 1) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (sp) stage_wide_piece(buf ^ 1, (a * FM + b) / PER);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                if (sp) stage_wide_next();
                __builtin_amdgcn_sched_barrier(0);
                PFA_WAIT_LGKM(0);
                pin_half(fA, 1);
                mfma_half(fA, 1);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                if (more) stage(ks + 1, buf ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                if (!(p.dbg & 4)) {
                    read_half_raw(fA, buf, 0, 0);
                    if constexpr (ALL_AT_ONCE) {
                        read_half_raw(fA, buf, 1, 1);
                        PFA_WAIT_LGKM(FN + FM);          // LDS reads return in order: the first half is here
                        pin_half(fA, 0);
                        mfma_half(fA, 0);
                        __builtin_amdgcn_sched_barrier(0);   // keep these MFMAs ahead of the second wait
                        PFA_WAIT_LGKM(0);
                        pin_half(fA, 1);
                        mfma_half(fA, 1);
                    } else {
                        PFA_WAIT_LGKM(0);
                        pin_half(fA, 0);
                        mfma_half(fA, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        read_half_raw(fA, buf, 1, 0);
                        PFA_WAIT_LGKM(0);
                        pin_half(fA, 0);
                        mfma_half(fA, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        }   // two-stage variants
    }

    // ------------------------------------------------------------------ epilogue
    // lane holds D[n = .. + 4*kk + r][m = .. + lrow]
    size_t opix[FM];
    bool mval[FM];
#pragma unroll
    for (int b = 0; b < FM; ++b) {
        const int m = m0 + wm * TM + b * 16 + lrow;
        mval[b] = m < p.M;
        opix[b] = 0;
        if (mval[b]) {
            if (p.mode == PFA_TCONV_S2) {
                const int ni = m / hw, rem = m - ni * hw;
                const int qy = rem / p.wq, qx = rem - qy * p.wq;
                opix[b] = ((size_t)ni * p.ho + 2 * qy + py) * p.wo + 2 * qx + px;
            } else {
                opix[b] = (size_t)m;
            }
        }
    }

    gather_epilogue<T, BN, WM, FN, FM>(p, acc, mval, opix, n0, wn, wm, tid, split, (size_t)cls * p.mtiles + mt, smem);
}

// ---------------------------------------------------------------- split-K finish
// out[pix][n] = act(bias + sum_s slab[s][pix][n]); optional BatchNorm partial statistics per row tile.
// Fast path: one float4 column chunk per lane (consecutive lanes = consecutive chunks of a row, coalesced),
// 256/cpr rows per pass, FIN_PASSES passes per block, per-column partial sums reduced through LDS.
constexpr int FIN_PASSES = 8;
// Activation / BatchNorm backward of a K-sliced input-gradient launch (pfa_gather_args.fuse_out 1|2), applied by the
// finish pass to the summed gradient: same arithmetic as the GEMM epilogue's fused block.
struct FinishFuse {
    int out;                 // 0 = off, 1 = out1, 2 = out2
    int act1, act2;
    int nreal;               // real channels of the fused output (coefficient array length)
    int stat_col0, stat_n;   // statistics cover the fused output's columns only
    const char* raw;
    const char* g2;
    const float* scale;
    const float* shift;
    const float* mean;
    const float* invstd;
};
// v[ps] += the slices' values of pass ps, in ascending slice order; KU slices x PS passes of loads in flight at a time
// (a block with few passes would otherwise wait one round trip per slice)
template <int KU, int PS>
__device__ __forceinline__ void sum_planes(const float* src0, size_t pstride, size_t plane, int splitk, const bool* live, f32x4* v) {
    int k = 0;
    if (KU > 1) {
        for (; k + KU <= splitk; k += KU) {
            f32x4 t[KU][PS];
#pragma unroll
            for (int u = 0; u < KU; ++u)
#pragma unroll
                for (int ps = 0; ps < PS; ++ps)
                    if (live[ps]) t[u][ps] = *(const f32x4*)(src0 + ps * pstride + (k + u) * plane);
#pragma unroll
            for (int u = 0; u < KU; ++u)
#pragma unroll
                for (int ps = 0; ps < PS; ++ps)
                    if (live[ps]) v[ps] += t[u][ps];
        }
    }
    for (; k < splitk; ++k) {
#pragma unroll
        for (int ps = 0; ps < PS; ++ps)
            if (live[ps]) v[ps] += *(const f32x4*)(src0 + ps * pstride + k * plane);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_splitk_finish_vec(const float* __restrict__ slab, int splitk, int pixels,
                                                           int n, int n1, const float* __restrict__ bias, int act,
                                                           float slope, char* out1, int ld1, char* out2, int ld2,
                                                           float* psum, float* psq, int cpr, int passes,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           int act2, const FinishFuse fz) {
    // scale != nullptr: the eval-mode affine epilogue of a K-sliced launch (pfa_gather_args.fuse_out == 3):
    // y = scale*sum + shift, out1 = act(y) and, when given, out2 = act2(y) over the same columns
    __shared__ float red[2][256][4];
    const int tid = threadIdx.x;
    const int col = tid % cpr, rsub = tid / cpr, rpp = 256 / cpr;
    const int c0 = col * 4;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *(const f32x4*)(bias + c0);
    f32x4 sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
    if (scale) { sc4 = *(const f32x4*)(scale + c0); sh4 = *(const f32x4*)(shift + c0); }
    // fused backward: this lane's four columns of the fused output, their coefficients
    const bool fmine = fz.out && (fz.out == 1 ? c0 < n1 : c0 >= n1);
    const int fcol = fz.out == 1 ? c0 : c0 - n1;
    const int fld = fz.out == 1 ? ld1 : ld2;
    f32x4 mu4 = {0.f, 0.f, 0.f, 0.f}, is4 = {1.f, 1.f, 1.f, 1.f};
    if (fmine) {
        const bool whole = fcol + 4 <= fz.nreal;
        if (fz.scale) {
            if (whole) { sc4 = *(const f32x4*)(fz.scale + fcol); sh4 = *(const f32x4*)(fz.shift + fcol); }
            else for (int e = 0; e < 4; ++e) if (fcol + e < fz.nreal) { sc4[e] = fz.scale[fcol + e]; sh4[e] = fz.shift[fcol + e]; }
        }
        if (fz.mean) {
            if (whole) { mu4 = *(const f32x4*)(fz.mean + fcol); is4 = *(const f32x4*)(fz.invstd + fcol); }
            else for (int e = 0; e < 4; ++e) if (fcol + e < fz.nreal) { mu4[e] = fz.mean[fcol + e]; is4[e] = fz.invstd[fcol + e]; }
        }
    }
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    const size_t row0 = (size_t)blockIdx.x * rpp * (passes < 0 ? -passes : passes);
    const size_t plane = (size_t)pixels * n;
    // All passes of the block are loaded together (the pass loop is unrolled, so every pass's read of slice k is in
    // flight at once): a block is latency-, not bandwidth-bound.  The sum over slices stays in ascending k per element.
    const float neg = act_neg(act, slope), neg2 = act_neg(act2, slope);
    const float fneg1 = act_neg(fz.act1, slope), fneg2 = act_neg(fz.act2, slope);
    f32x4 v[FIN_PASSES];
    bool live[FIN_PASSES];
#pragma unroll
    for (int ps = 0; ps < FIN_PASSES; ++ps) {
        live[ps] = ps < (passes < 0 ? -passes : passes) && row0 + (size_t)ps * rpp + rsub < (size_t)pixels;
        v[ps] = bv;
    }
    const float* src0 = slab + (row0 + rsub) * n + c0;
    const size_t pstride = (size_t)rpp * n;
    if (passes < 0) sum_planes<1, FIN_PASSES>(src0, pstride, plane, splitk, live, v);      // (A/B: knob 1 bit 1024)
    else if (passes == 1) sum_planes<8, 1>(src0, pstride, plane, splitk, live, v);
    else if (passes == 2) sum_planes<4, 2>(src0, pstride, plane, splitk, live, v);
    else if (passes <= 4) sum_planes<2, 4>(src0, pstride, plane, splitk, live, v);
    else sum_planes<1, FIN_PASSES>(src0, pstride, plane, splitk, live, v);
#pragma unroll
    for (int ps = 0; ps < FIN_PASSES; ++ps) {
        if (!live[ps]) continue;
        const size_t row = row0 + (size_t)ps * rpp + rsub;
        float o[4];
        if (fz.out) {
            if (fmine) {
                float x[4], g2[4] = {0.f, 0.f, 0.f, 0.f};
                load4<T>(fz.raw + (row * fld + fcol) * sizeof(T), x);
                if (fz.g2) load4<T>(fz.g2 + (row * fld + fcol) * sizeof(T), g2);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float y = x[e] * sc4[e] + sh4[e];
                    float d = act_grad_lin(y, fneg1) * v[ps][e];
                    if (fz.g2) d += act_grad_lin(y, fneg2) * g2[e];
                    o[e] = d;
                    s[e] += d;
                    q[e] += d * ((x[e] - mu4[e]) * is4[e]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = v[ps][e];
            }
            if (c0 < n1) store4<T>(out1 + (row * ld1 + c0) * sizeof(T), o);
            else store4<T>(out2 + (row * ld2 + (c0 - n1)) * sizeof(T), o);
            continue;
        }
        s += v[ps];
        q += v[ps] * v[ps];
        if (scale) {
            const f32x4 y = v[ps] * sc4 + sh4;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = act_lin(y[e], neg);          // (the affine map is never followed by tanh)
            store4<T>(out1 + (row * ld1 + c0) * sizeof(T), o);
            if (out2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = act_lin(y[e], neg2);
                store4<T>(out2 + (row * ld2 + c0) * sizeof(T), o);
            }
            continue;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = act_lin(v[ps][e], neg);            // (tanh outputs take k_splitk_finish_small)
        if (c0 < n1) store4<T>(out1 + (row * ld1 + c0) * sizeof(T), o);
        else store4<T>(out2 + (row * ld2 + (c0 - n1)) * sizeof(T), o);
    }
    if (!psum) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][tid][e] = s[e]; red[1][tid][e] = q[e]; }
    __syncthreads();
    for (int off = rpp >> 1; off > 0; off >>= 1) {       // fixed-shape tree over the row groups (rpp is a power of two)
        if (rsub < off) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                red[0][tid][e] += red[0][tid + off * cpr][e];
                red[1][tid][e] += red[1][tid + off * cpr][e];
            }
        }
        __syncthreads();
    }
    if (tid < cpr) {
        if (fz.out) {       // rows of stat_n columns: the fused output's only
            const int sc0 = tid * 4 - fz.stat_col0;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (sc0 + e >= 0 && sc0 + e < fz.stat_n) {
                    psum[(size_t)blockIdx.x * fz.stat_n + sc0 + e] = red[0][tid][e];
                    psq[(size_t)blockIdx.x * fz.stat_n + sc0 + e] = red[1][tid][e];
                }
            return;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            psum[(size_t)blockIdx.x * n + tid * 4 + e] = red[0][tid][e];
            psq[(size_t)blockIdx.x * n + tid * 4 + e] = red[1][tid][e];
        }
    }
}

// Generic path (any n): block = 64 rows, thread t walks columns t, t+256, ...
template <typename T>
__global__ __launch_bounds__(256) void k_splitk_finish(const float* __restrict__ slab, int splitk, int pixels, int n,
                                                       int n1, const float* __restrict__ bias, int act, float slope,
                                                       char* out1, int ld1, char* out2, int ld2, float* psum,
                                                       float* psq, int slab_ld) {
    const int row0 = blockIdx.x * 64;
    const int rows = min(64, pixels - row0);
    // a single output also gets its pad channels [n, ld1) written (zeros), like the direct epilogue does
    const int ncols = (n1 >= n) ? max(n, ld1) : n;
    for (int col = threadIdx.x; col < ncols; col += 256) {
        const bool real = col < n;
        const float bv = (bias && real) ? bias[col] : 0.f;
        float s = 0.f, q = 0.f;
        for (int r = 0; r < rows; ++r) {
            const size_t pix = row0 + r;
            float v = bv;
            if (real)
                for (int k = 0; k < splitk; ++k) v += slab[((size_t)k * pixels + pix) * slab_ld + col];
            s += v;
            q += v * v;
            v = apply_act(v, act, slope);
            if (col < n1) ((T*)out1)[pix * ld1 + col] = from_f32<T>(v);
            else ((T*)out2)[pix * ld2 + (col - n1)] = from_f32<T>(v);
        }
        if (psum && real) {
            psum[(size_t)blockIdx.x * n + col] = s;
            psq[(size_t)blockIdx.x * n + col] = q;
        }
    }
}

// Few output channels (the 1-channel logit conv, the 3-channel image conv): one lane per (pixel, channel).
template <typename T>
__global__ __launch_bounds__(256) void k_splitk_finish_small(const float* __restrict__ slab, int splitk, int pixels,
                                                             int n, const float* __restrict__ bias, int act,
                                                             float slope, T* out, int ld, int slab_ld) {
    const size_t total = (size_t)pixels * ld;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const size_t pix = idx / ld;
        const int col = (int)(idx - pix * ld);
        float v = 0.f;
        if (col < n) {
            v = bias ? bias[col] : 0.f;
            for (int k = 0; k < splitk; ++k) v += slab[((size_t)k * pixels + pix) * slab_ld + col];
            v = apply_act(v, act, slope);
        }
        out[idx] = from_f32<T>(v);
    }
}

// rows of a [pixels][n] slab one finish block covers, and whether the vector path applies
inline bool finish_vec_ok(int n, int n1, int ld1, int act) {
    const int cpr = n / 4;
    return n % 4 == 0 && cpr >= 1 && cpr <= 256 && 256 % cpr == 0 && act != PFA_ACT_TANH && (n1 >= n ? ld1 == n : true);
}
// passes per block: fewer when there are few pixels, so that the grid still covers the chip
inline int finish_passes(int n, int pixels) {
    const int rpp = 256 / (n / 4);
    int passes = pixels / (rpp * 512);
    return passes < 1 ? 1 : (passes > FIN_PASSES ? FIN_PASSES : passes);
}
inline int finish_rows_per_block(int n, int n1, int ld1, int act, int pixels) {
    return finish_vec_ok(n, n1, ld1, act) ? (256 / (n / 4)) * finish_passes(n, pixels) : 64;
}

int fill_params(const pfa_gather_args* a, GatherK& k, int& classes) {
    PFA_REQUIRE(a && a->struct_size == (int)sizeof(pfa_gather_args), "pfa_gather_gemm: bad struct_size");
    PFA_REQUIRE(a->dtype == PFA_F32 || a->dtype == PFA_BF16, "pfa_gather_gemm: bad dtype %d", a->dtype);
    PFA_REQUIRE(a->mode >= PFA_CONV_S2 && a->mode <= PFA_CONV_S1_DGRAD, "pfa_gather_gemm: bad mode %d", a->mode);
    const int ch = pfa_ch(a->dtype);
    PFA_REQUIRE(a->nb > 0 && a->hq > 0 && a->wq > 0 && a->hs > 0 && a->ws > 0, "pfa_gather_gemm: empty geometry");
    PFA_REQUIRE(a->c1 > 0 && a->c1 % ch == 0 && a->c2 >= 0 && a->c2 % ch == 0,
                "pfa_gather_gemm: source channels (%d,%d) must be multiples of %d", a->c1, a->c2, ch);
    PFA_REQUIRE(a->n > 0 && a->n1 > 0 && a->n1 % 4 == 0 && a->ld1 % 4 == 0 && a->ld2 % 4 == 0 && a->n1 <= a->ld1,
                "pfa_gather_gemm: output split n=%d n1=%d ld1=%d ld2=%d", a->n, a->n1, a->ld1, a->ld2);
    PFA_REQUIRE(a->src1 && a->w && a->out1 && (a->c2 == 0 || a->src2) && (a->ld2 == 0 || a->out2),
                "pfa_gather_gemm: null pointer");
    PFA_REQUIRE(a->hs < 32768 && a->ws < 32768, "pfa_gather_gemm: spatial size too large");
    PFA_REQUIRE((a->psum == nullptr) == (a->psq == nullptr), "pfa_gather_gemm: psum/psq must come together");
    k.src1 = (const char*)a->src1; k.src2 = (const char*)a->src2; k.w = (const char*)a->w;
    k.out1 = (char*)a->out1; k.out2 = (char*)a->out2; k.bias = a->bias;
    k.psum = a->psum; k.psq = a->psq; k.slab = a->slab;
    k.mode = a->mode; k.nb = a->nb; k.hq = a->hq; k.wq = a->wq; k.hs = a->hs; k.ws = a->ws;
    k.c1 = a->c1; k.c2 = a->c2; k.ctot = a->c1 + a->c2;
    k.cpt = k.ctot / ch; k.cpt_shift = ilog2_exact(k.cpt);
    const int ntaps = a->mode == PFA_TCONV_S2 ? 4 : 16;
    k.ktot = ntaps * k.cpt; k.ksteps = (k.ktot + 7) / 8;
    k.n = a->n; k.n1 = a->n1; k.ld1 = a->ld1; k.ld2 = a->ld2;
    k.M = a->nb * a->hq * a->wq;
    classes = a->mode == PFA_TCONV_S2 ? 4 : 1;
    if (a->mode == PFA_TCONV_S2) { k.ho = 2 * a->hq; k.wo = 2 * a->wq; }
    else { k.ho = a->hq; k.wo = a->wq; }
    k.npix = a->nb * k.ho * k.wo;
    k.splitk = a->splitk < 1 ? 1 : a->splitk;
    if (k.splitk > k.ksteps) k.splitk = k.ksteps;
    k.kper = (k.ksteps + k.splitk - 1) / k.splitk;
    k.splitk = (k.ksteps + k.kper - 1) / k.kper;     // no empty slices
    k.act = a->act; k.slope = a->slope;
    const int kstep_ch = 8 * ch;                  // channels per K-step
    const long long es = a->dtype == PFA_BF16 ? 2 : 4;
    const bool small32 = (long long)a->nb * a->hs * a->ws * (a->c1 > a->c2 ? a->c1 : a->c2) * es < (1ll << 31) &&
                         (long long)a->n * 16 * (a->c1 + a->c2) * es < (1ll << 31);
    k.small32 = small32 && (long long)k.npix * (a->ld1 > a->ld2 ? a->ld1 : a->ld2) * es < (1ll << 31) ? 1 : 0;
    k.wide = (a->c1 % kstep_ch == 0 && a->c2 % kstep_ch == 0 && small32) ? 1 : 0;
    k.spt = k.wide ? k.ctot / kstep_ch : 1;
    k.sp1 = k.wide ? a->c1 / kstep_ch : 1;
    k.dbg = pfa_knob(1);
    k.fuse_out = a->fuse_out; k.fuse_act1 = a->fuse_act1; k.fuse_act2 = a->fuse_act2;
    k.fuse_raw = (const char*)a->fuse_raw; k.fuse_g2 = (const char*)a->fuse_g2;
    k.fuse_scale = a->fuse_scale; k.fuse_shift = a->fuse_shift; k.fuse_mean = a->fuse_mean; k.fuse_invstd = a->fuse_invstd;
    k.stat_col0 = 0; k.stat_n = a->n;
    // vector (16-byte) loads of per-channel coefficients need aligned arrays; anything else takes the element-wise path
    auto misaligned = [](const void* q) { return q && ((uintptr_t)q & 15u); };
    k.bias_unaligned = misaligned(a->bias) ? 1 : 0;
    const bool coef_bad = misaligned(a->fuse_scale) || misaligned(a->fuse_shift) || misaligned(a->fuse_mean) || misaligned(a->fuse_invstd);
    k.fuse_nreal = a->fuse_out == 1 ? (a->n < a->n1 ? a->n : a->n1) : a->fuse_out == 2 ? a->n - a->n1 : a->n;
    k.coef_unaligned = coef_bad ? 1 : 0;
    if (a->fuse_out == 3) {
        PFA_REQUIRE(a->fuse_scale && a->fuse_shift, "pfa_gather_gemm: the affine epilogue needs scale and shift");
        PFA_REQUIRE(a->n1 >= a->n && a->bias == nullptr && a->psum == nullptr && a->act != PFA_ACT_TANH &&
                    a->fuse_act2 != PFA_ACT_TANH,
                    "pfa_gather_gemm: the affine epilogue writes all columns to out1 (n1 >= n), without bias, statistics or tanh");
        PFA_REQUIRE(a->out2 == nullptr || a->ld2 == a->ld1, "pfa_gather_gemm: affine epilogue outputs must share one row length");
    } else if (a->fuse_out == PFA_FUSE_DEFER) {
        PFA_REQUIRE(k.splitk > 1 && a->slab && a->bias == nullptr && a->act == PFA_ACT_NONE && a->psum == nullptr && a->n % 4 == 0,
                    "pfa_gather_gemm: PFA_FUSE_DEFER leaves K-slice partial planes: it needs splitk > 1, a slab, n %% 4 == 0 and no bias / activation / statistics");
    } else if (a->fuse_out) {
        PFA_REQUIRE(a->fuse_out == 1 || (a->fuse_out == 2 && a->ld2 > 0), "pfa_gather_gemm: fuse_out must name an existing output");
        PFA_REQUIRE(a->fuse_raw && (a->fuse_scale == nullptr) == (a->fuse_shift == nullptr) &&
                    (a->fuse_mean == nullptr) == (a->fuse_invstd == nullptr), "pfa_gather_gemm: incomplete fused-backward arguments");
        PFA_REQUIRE(a->bias == nullptr && a->act == PFA_ACT_NONE, "pfa_gather_gemm: fused backward excludes bias/activation");
        k.stat_col0 = a->fuse_out == 1 ? 0 : a->n1;
        k.stat_n = a->fuse_out == 1 ? a->ld1 : a->ld2;
    }
    k.slab_ld = (a->n + 3) & ~3;
    if (k.splitk > 1) PFA_REQUIRE(a->slab, "pfa_gather_gemm: split-K needs a slab of pfa_gather_slab_floats() floats");
    PFA_REQUIRE(a->act != PFA_ACT_TANH || a->n <= 16, "pfa_gather_gemm: tanh epilogue is only built for n <= 16");
    PFA_REQUIRE(!(a->fuse_out && a->fuse_out != PFA_FUSE_DEFER && k.splitk > 1) || (finish_vec_ok(a->n, a->n1, a->ld1, a->act) && k.slab_ld == a->n),
                "pfa_gather_gemm: a fused epilogue of a K-sliced launch needs n %% 4 == 0, n/4 dividing 256 (and ld1 == n for one output)");
    // geometry sanity: the source grid must be what the mode implies
    if (a->mode == PFA_CONV_S2) PFA_REQUIRE(a->hs == 2 * a->hq && a->ws == 2 * a->wq, "pfa_gather_gemm: CONV_S2 needs hs=2*hq");
    if (a->mode == PFA_CONV_S1) PFA_REQUIRE(a->hs == a->hq + 1 && a->ws == a->wq + 1, "pfa_gather_gemm: CONV_S1 needs hs=hq+1");
    if (a->mode == PFA_TCONV_S2) PFA_REQUIRE(a->hs == a->hq && a->ws == a->wq, "pfa_gather_gemm: TCONV_S2 needs hs=hq");
    if (a->mode == PFA_CONV_S1_DGRAD) PFA_REQUIRE(a->hs == a->hq - 1 && a->ws == a->wq - 1, "pfa_gather_gemm: CONV_S1_DGRAD needs hs=hq-1");
    return 0;
}

// ---- tile configuration
// shape 0: 128x128, 4 waves   shape 1: 128x64, 4 waves   shape 2: 256x16, 4 waves   shape 3: 256x128, 8 waves
// shape 4: 256x64, 8 waves, 3-stage ring where it applies (16 < n <= 64; off by default, knob 7 = 2 turns it on)
// (two LDS buffers each).  pfa_set_knob(0, v): v = 4 forces shape 3 on launches with n >= 128, anything else is the
// automatic choice.  pfa_set_knob(2, v): K-loop variant, 0 = default, v = 1..3 selects EARLY = v - 1.
struct TileCfg { int id, bm, bn; };
constexpr int RING_MIN_TILES = 64;     // tuned in situ: 1 and 128 are equal, 256 loses a third of the gain (pfa_set_knob(6, n) overrides)

inline TileCfg choose_cfg(const GatherK& k) {
    static const TileCfg cfgs[] = {{0, 128, 128}, {1, 128, 64}, {2, 256, 16}, {3, 256, 128}, {4, 256, 64}};
    if (k.n <= 16) return cfgs[2];
    if (k.n <= 64) {
        // 256x64 with the 3-stage ring (8 waves, 120 KiB) for the wide launches with enough tiles; else 128x64
        const long long tiles = (long long)((k.M + 255) / 256) * (k.mode == PFA_TCONV_S2 ? 4 : 1) * k.splitk;
        const bool ring_ok = k.wide && k.kper >= 3 && tiles >= (pfa_knob(6) > 0 ? pfa_knob(6) : RING_MIN_TILES);
        // (measured slower than 128x64 reads-first: d2 forward 75 vs 61 us, -1 % in the step: knob 7 = 2 enables it)
        if (pfa_knob(0) == 5 || (pfa_knob(0) == 0 && ring_ok && pfa_knob(7) == 2)) return cfgs[4];
        return cfgs[1];
    }
    // Default: 128x128 at two workgroups per CU.  The 256x128 tile (85 FLOP per staged byte) is 3-5 % faster per layer in
    // isolation (tools/layer_bench.py) but not inside the three-stream step, where its 96 KiB of LDS per workgroup leaves
    // less room for the co-running weight-gradient kernels (same-box A/B: 3168 vs 3187 pairs/s); it stays selectable.
    if (pfa_knob(0) == 4 || pfa_knob(0) == 8) return cfgs[3];     // 8: with the 3-stage ring where it applies
    if (pfa_knob(0) == 0) {
        // Default for n >= 128: the 256x128 tile with the 3-stage ring (one 8-wave workgroup owns the CU, the DMA queue
        // is never drained) wherever the launch still has enough 256-row tiles; +4.6 % pairs/s over 128x128 everywhere.
        const long long tiles256 = (long long)((k.M + 255) / 256) * ((k.n + 127) / 128) * (k.mode == PFA_TCONV_S2 ? 4 : 1) *
                                   k.splitk;
        const int thr = pfa_knob(6) > 0 ? pfa_knob(6) : RING_MIN_TILES;
        if (k.wide && k.kper >= 3 && tiles256 >= thr) return cfgs[3];
    }
    if (pfa_knob(0) == 6) return cfgs[1];              // experiment: 128x64 everywhere
    // (128x64 for unsplit launches with one 128x128 workgroup per CU: -14 % on e3/c3 forward in isolation, nothing in
    //  the step - not enabled.)
    return cfgs[0];
}

// geometry of a call whose workspaces are not allocated yet (the sizing queries below)
inline int fill_params_geometry(const pfa_gather_args* a, GatherK& k, int& classes) {
    if (!a || a->struct_size != (int)sizeof(pfa_gather_args)) return fill_params(a, k, classes);
    pfa_gather_args b = *a;
    if (!b.slab) b.slab = (float*)(uintptr_t)16;
    return fill_params(&b, k, classes);
}

// Launches that take the patch kernel of conv_patch.hip instead of the implicit GEMM: stride-1 convolutions (forward, and
// input gradient = the same walk with the taps reversed) in bf16 with one source and one unpadded output, no bias /
// activation, at least 128 workgroups (smaller launches rely on more K slices than the channel chunks give).  A fused
// epilogue rides on the finish pass, i.e. needs a K-sliced launch.  pfa_set_knob(3, 1) turns the route off, 2 takes it
// whenever the geometry fits (tests).
inline bool patch_route(const pfa_gather_args* a, const GatherK& k, int* tiles, int* bn_out = nullptr, int* splitk_out = nullptr) {
    if (pfa_knob(3) == 1) return false;
    if (!k.small32) return false;          // the patch kernel builds 32-bit byte offsets: tensors of 2 GiB and more stay on the implicit GEMM
    const bool fwd = a->mode == PFA_CONV_S1, grad = a->mode == PFA_CONV_S1_DGRAD;
    if (!((fwd || grad) && a->dtype == PFA_BF16 && a->c2 == 0 && a->bias == nullptr && a->act == PFA_ACT_NONE &&
          a->n1 >= a->n && a->ld1 == a->n && a->ld2 == 0 && a->fuse_out != 3 && a->fuse_out != PFA_FUSE_DEFER))
        return false;
    int t = 0, bn = 128, sk = k.splitk;
    if (pfa_knob(3) == 6) {                // the 16x16x32 form (A/B): 128-channel tiles, the caller's slices, fused epilogue only in the finish pass
        if (!(a->fuse_out == 0 || k.splitk > 1)) return false;
        if (!pfa_conv_s1_patch_ok(grad ? 1 : 0, a->nb, a->hs, a->ws, a->c1, a->n, k.splitk, &t)) return false;
    } else if (!pfa_conv_s1_plan(grad ? 1 : 0, a->nb, a->hs, a->ws, a->c1, a->n, k.splitk, &bn, &sk, &t)) {
        return false;
    }
    if (pfa_knob(3) != 2 && pfa_knob(3) != 7 && (long long)t * (a->n / bn) * sk < 128) return false;
    if (tiles) *tiles = t;
    if (bn_out) *bn_out = bn;
    if (splitk_out) *splitk_out = sk;
    return true;
}

}  // namespace

extern "C" size_t pfa_gather_slab_floats(const pfa_gather_args* a) {
    GatherK k; int classes;
    if (fill_params_geometry(a, k, classes)) return 0;
    if (k.splitk <= 1) return 0;
    return (size_t)k.splitk * k.npix * k.slab_ld;                       // [splitk][pixels][round4(n)]
}

extern "C" int pfa_gather_stat_tiles(const pfa_gather_args* a) {
    GatherK k; int classes;
    if (int rc = fill_params(a, k, classes)) return rc;
    int ptiles = 0, pbn = 128, psk = 1;
    if (patch_route(a, k, &ptiles, &pbn, &psk)) {
        if (psk == 1) return ptiles;
        k.splitk = psk;
    }
    int itiles = 0;
    if (k.splitk == 1 && pfa_conv_image_ok(k, a->dtype, &itiles)) return itiles;
    Patch2Plan plan2;
    if (pfa_conv_patch2_plan(k, a->dtype, &plan2)) {
        if (plan2.splitk == 1) return plan2.stat_tiles;
        k.splitk = plan2.splitk;
    }
    if (k.splitk > 1) {
        const int rpb = finish_rows_per_block(k.n, k.n1, k.ld1, k.act, k.npix);
        return (k.npix + rpb - 1) / rpb;
    }
    const int bm = choose_cfg(k).bm;
    return classes * ((k.M + bm - 1) / bm);
}

// Slices pfa_gather_gemm will run this call in (the planners may keep fewer than the caller asked for - down to one, where the
// tiles alone fill the chip): what a caller that wants to finish the slices itself (PFA_FUSE_DEFER) has to know beforehand.
extern "C" int pfa_gather_plan_splitk(const pfa_gather_args* a) {
    GatherK k; int classes;
    if (int rc = fill_params_geometry(a, k, classes)) return rc;
    const bool defer = a->fuse_out == PFA_FUSE_DEFER;
    if (defer) k.fuse_out = 0;
    int ptiles = 0, pbn = 128, psk = 1;
    if (!defer && patch_route(a, k, &ptiles, &pbn, &psk)) return psk;
    int itiles = 0;
    if (!defer && k.splitk == 1 && pfa_conv_image_ok(k, a->dtype, &itiles)) return 1;
    Patch2Plan plan2;
    if (pfa_conv_patch2_plan(k, a->dtype, &plan2)) return plan2.splitk;     // (one slice: the caller should not defer this launch)
    return k.splitk;
}

// Elements each order pulls through the 8 private L2s, to first order (an operand part an XCD needs is fetched by that XCD
// once; A = input elements, W = weight elements, C = parity classes, S = K slices, P = S*C "planes" of m x n tiles):
//   order 0 (plane-major): a plane reads the whole input and 1/P of the weights; with P < 8 a plane is cut over 8/P XCDs
//                          by pixel range          ->  A*min(P, 8) + W*max(1, 8/P)
//   order 1 (pixel-major): each XCD reads its pixels once and all the weights   ->  A + 8*W
//   order 2 (channel-major, needs >= 8 n-tiles): all pixels, its share of the weights   ->  8*A + W
// Measured with rocprofv3 FETCH_SIZE inside a step (tools/step_traffic.py): d2 forward 472 -> see DESIGN section 6.
static int choose_order(const GatherK& k) {
    if (pfa_knob(5) >= 1 && pfa_knob(5) <= 3) return pfa_knob(5) - 1;
    const double A = (double)k.nb * k.hs * k.ws * k.ctot;
    const double W = (double)k.n * 16 * k.ctot;
    const double P = (double)k.splitk * k.classes;
    const double cost0 = A * (P < 8.0 ? P : 8.0) + W * (P < 8.0 ? 8.0 / P : 1.0);
    const double cost1 = A + 8.0 * W;
    const double cost2 = k.ntiles >= 8 ? 8.0 * A + W : 1e300;
    if (cost2 < cost0 && cost2 < cost1) return 2;
    return cost1 < cost0 ? 1 : 0;
}

template <typename T, int BM, int BN, int WM, int WN, int EARLY>
static int launch_gather(GatherK k, int classes, hipStream_t st) {
    k.mtiles = (k.M + BM - 1) / BM;
    k.ntiles = (k.n + BN - 1) / BN;
    k.nwg = k.mtiles * k.ntiles * classes * k.splitk;
    k.classes = classes;
    k.order = choose_order(k);
    dim3 grid(k.nwg);
    const size_t lds = (EARLY >= 3 ? 3 : 2) * (BM * 128 + BN * 128);
    static std::atomic<bool> attr_done{false};   // (launches may come from several host threads)
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)k_gather_gemm<T, BM, BN, WM, WN, EARLY>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL((k_gather_gemm<T, BM, BN, WM, WN, EARLY>), grid, dim3(64 * WM * WN), lds, st, k);
    static char name[64];
    if (!name[0]) snprintf(name, sizeof(name), "k_gather_gemm<%s,%d,%d,%d,%d,%d>", sizeof(T) == 2 ? "bf16" : "f32", BM, BN, WM, WN, EARLY);
    g_last_gather_name = name;
    PFA_LAUNCHED("k_gather_gemm");
    return 0;
}

template <typename T>
static int dispatch_gather(const GatherK& k, int classes, int cfg, hipStream_t st) {
    // default K-loop per tile shape (tools/layer_bench.py): spreading the DMA pays on the 128x128 tile; the 128x64 tile
    // (16 MFMAs per wave and K-step, three to four workgroups per CU) is 30-40 % faster with the plain reads-first loop
    static const int default_early[] = {2, 0, 1, 1, 1};
    int early = pfa_knob(2) >= 1 && pfa_knob(2) <= 3 ? pfa_knob(2) - 1 : default_early[cfg];
    if ((pfa_knob(0) == 8 || pfa_knob(0) == 0) && cfg == 3 && pfa_knob(2) == 0) early = 2;    // 256x128: the 3-stage ring
    // round 4, opt-in (knob 1 bit 16384): the launches of the 256-row tile in bf16 on whole 64-channel chunks on the row kernel of
    // conv_rows.hip (the patch kernels' K loop over gathered rows; measured slower inside the three-stream step)
    if constexpr (std::is_same_v<T, bf16_t>) {
        if (cfg == 3 && pfa_knob(0) == 0 && pfa_knob(2) == 0 && pfa_knob(7) == 0 && pfa_conv_rows_ok(k, PFA_BF16)) {
            const char* nm = "";
            const int rc = pfa_conv_rows(k, classes, [](const GatherK& kk) { return choose_order(kk); }, st, &nm);
            g_last_gather_name = nm;
            return rc;
        }
    }
    // (narrow-staging launches on the 128x128 tile keep the spread loop: reads-first measured 40 % slower there)
    switch (cfg * 3 + early) {
        case 0: return launch_gather<T, 128, 128, 2, 2, 0>(k, classes, st);
        case 1: return launch_gather<T, 128, 128, 2, 2, 1>(k, classes, st);
        case 2: return launch_gather<T, 128, 128, 2, 2, 2>(k, classes, st);
        case 3:
            // knob 7 = 5: three-stage ring on the 128x64 tile (72 KiB: still two workgroups per CU)
            if (pfa_knob(7) == 5 && k.wide && k.kper >= 3) return launch_gather<T, 128, 64, 2, 2, 3>(k, classes, st);
            return launch_gather<T, 128, 64, 2, 2, 0>(k, classes, st);
        case 4: return launch_gather<T, 128, 64, 2, 2, 1>(k, classes, st);
        case 5: return launch_gather<T, 128, 64, 2, 2, 2>(k, classes, st);
        case 6: return launch_gather<T, 256, 16, 4, 1, 0>(k, classes, st);
        case 7: case 8: return launch_gather<T, 256, 16, 4, 1, 1>(k, classes, st);
        case 9: return launch_gather<T, 256, 128, 4, 2, 0>(k, classes, st);
        case 10: return launch_gather<T, 256, 128, 4, 2, 1>(k, classes, st);
        case 11:
            // knob 2 = 3 on the 256x128 tile: the 3-stage ring (wide launches with at least three K-steps per slice)
            if (k.wide && k.kper >= 3)
                return pfa_knob(7) == 4   ? launch_gather<T, 256, 128, 4, 2, 4>(k, classes, st)      // half-step pipelined ring
                       : pfa_knob(7) == 6 ? launch_gather<T, 256, 128, 4, 2, 5>(k, classes, st)      // staggered ring
                       : pfa_knob(7) == 7 ? launch_gather<T, 256, 128, 4, 2, 6>(k, classes, st)      // staggered, half-step phases
                                        : launch_gather<T, 256, 128, 4, 2, 3>(k, classes, st);
            return launch_gather<T, 256, 128, 4, 2, 1>(k, classes, st);
        default:
            if (k.wide && k.kper >= 3) return launch_gather<T, 256, 64, 4, 2, 3>(k, classes, st);
            return launch_gather<T, 256, 64, 4, 2, 1>(k, classes, st);
    }
}

extern "C" const char* pfa_last_gather_kernel(void) { return g_last_gather_name; }

extern "C" int pfa_gather_timing(int enable) {
    std::lock_guard<std::mutex> lk(g_timed_mu);
    if (enable) {
        for (auto& t : g_timed) { (void)hipEventDestroy(t.e0); (void)hipEventDestroy(t.e1); }
        g_timed.clear();
    }
    g_timing.store(enable != 0);
    return 0;
}

extern "C" int pfa_gather_timing_count(void) {
    std::lock_guard<std::mutex> lk(g_timed_mu);
    return (int)g_timed.size();
}

extern "C" int pfa_gather_timing_get(int i, float* ms, const char** name, double* flop, double* bytes, int* splitk) {
    std::lock_guard<std::mutex> lk(g_timed_mu);
    PFA_REQUIRE(i >= 0 && i < (int)g_timed.size() && ms && name && flop && bytes && splitk, "pfa_gather_timing_get: bad arguments");
    const TimedLaunch& t = g_timed[i];
    hipError_t e = hipEventElapsedTime(ms, t.e0, t.e1);     // hipErrorNotReady until the launch has completed
    if (e != hipSuccess) PFA_FAIL((int)e, "pfa_gather_timing_get: %s (synchronise the stream first)", hipGetErrorString(e));
    *name = t.name; *flop = t.flop; *bytes = t.bytes; *splitk = t.splitk;
    return 0;
}

extern "C" int pfa_gather_timing_grid(int i, int* nwg) {
    std::lock_guard<std::mutex> lk(g_timed_mu);
    PFA_REQUIRE(i >= 0 && i < (int)g_timed.size() && nwg, "pfa_gather_timing_grid: bad arguments");
    *nwg = g_timed[i].nwg;
    return 0;
}

extern "C" int pfa_gather_timing_start(int i, int ref, float* start_ms) {
    std::lock_guard<std::mutex> lk(g_timed_mu);
    PFA_REQUIRE(i >= 0 && i < (int)g_timed.size() && ref >= 0 && ref < (int)g_timed.size() && start_ms,
                "pfa_gather_timing_start: bad arguments");
    hipError_t e = hipEventElapsedTime(start_ms, g_timed[ref].e0, g_timed[i].e0);    // (events of different streams compare)
    if (e != hipSuccess) PFA_FAIL((int)e, "pfa_gather_timing_start: %s (synchronise first)", hipGetErrorString(e));
    return 0;
}

void* g_pfa_dbgbuf = nullptr;
size_t g_pfa_dbgbuf_bytes = 0;
extern "C" int pfa_set_debug_buffer(void* dev, size_t bytes) {
    g_pfa_dbgbuf = dev;
    g_pfa_dbgbuf_bytes = dev ? bytes : 0;
    return 0;
}

extern "C" int pfa_set_knob(int id, int value) {
    PFA_REQUIRE(id >= 0 && id < 16, "pfa_set_knob: id out of range");
    g_knob[id].store(value, std::memory_order_relaxed);
    return 0;
}

static int splitk_finish(const float* slab, int slab_ld, int splitk, int pixels, int n, int n1, const float* bias, int act,
                         float slope, int dtype, void* out1, int ld1, void* out2, int ld2, float* psum, float* psq,
                         int* ntiles_stat, pfa_stream s, const float* scale, const float* shift, int act2,
                         const FinishFuse& fz);

extern "C" int pfa_splitk_finish(const float* slab, int slab_ld, int splitk, int pixels, int n, int n1,
                                 const float* bias, int act, float slope, int dtype, void* out1, int ld1, void* out2,
                                 int ld2, float* psum, float* psq, int* ntiles_stat, pfa_stream s) {
    return splitk_finish(slab, slab_ld, splitk, pixels, n, n1, bias, act, slope, dtype, out1, ld1, out2, ld2, psum, psq,
                         ntiles_stat, s, nullptr, nullptr, PFA_ACT_NONE, FinishFuse{});
}

static int splitk_finish(const float* slab, int slab_ld, int splitk, int pixels, int n, int n1, const float* bias, int act,
                         float slope, int dtype, void* out1, int ld1, void* out2, int ld2, float* psum, float* psq,
                         int* ntiles_stat, pfa_stream s, const float* scale, const float* shift, int act2,
                         const FinishFuse& fz) {
    PFA_REQUIRE(slab && out1 && splitk >= 1 && pixels > 0 && n > 0 && slab_ld >= n, "pfa_splitk_finish: bad arguments");
    PFA_REQUIRE(n1 >= n || out2 || scale, "pfa_splitk_finish: out2 missing");
    PFA_REQUIRE(dtype == PFA_F32 || dtype == PFA_BF16, "pfa_splitk_finish: bad dtype");
    const bool vec = finish_vec_ok(n, n1, ld1, act) && n1 % 4 == 0 && slab_ld == n;
    PFA_REQUIRE(!(scale || fz.out) || vec,
                "pfa_gather_gemm: a fused epilogue of a K-sliced launch needs n %% 4 == 0, n/4 dividing 256 (and ld1 == n for one output)");
    const bool small = !vec && n1 >= n && ld1 <= 16 && !psum;
    const int passes = vec ? finish_passes(n, pixels) : 1;
    const int rpb = vec ? (256 / (n / 4)) * passes : 64;
    const int kpasses = (pfa_knob(1) & 1024) ? -passes : passes;
    const int tiles = (pixels + rpb - 1) / rpb;
    if (ntiles_stat) *ntiles_stat = tiles;
    hipStream_t st = (hipStream_t)s;
    if (vec) {
        if (dtype == PFA_BF16)
            hipLaunchKernelGGL(k_splitk_finish_vec<bf16_t>, dim3(tiles), dim3(256), 0, st, slab, splitk, pixels, n, n1,
                               bias, act, slope, (char*)out1, ld1, (char*)out2, ld2, psum, psq, n / 4, kpasses, scale, shift, act2, fz);
        else
            hipLaunchKernelGGL(k_splitk_finish_vec<float>, dim3(tiles), dim3(256), 0, st, slab, splitk, pixels, n, n1,
                               bias, act, slope, (char*)out1, ld1, (char*)out2, ld2, psum, psq, n / 4, kpasses, scale, shift, act2, fz);
    } else if (small) {
        const int grid = (int)(((size_t)pixels * ld1 + 255) / 256);
        if (dtype == PFA_BF16)
            hipLaunchKernelGGL(k_splitk_finish_small<bf16_t>, dim3(grid > 4096 ? 4096 : grid), dim3(256), 0, st, slab,
                               splitk, pixels, n, bias, act, slope, (bf16_t*)out1, ld1, slab_ld);
        else
            hipLaunchKernelGGL(k_splitk_finish_small<float>, dim3(grid > 4096 ? 4096 : grid), dim3(256), 0, st, slab,
                               splitk, pixels, n, bias, act, slope, (float*)out1, ld1, slab_ld);
    } else if (dtype == PFA_BF16) {
        hipLaunchKernelGGL(k_splitk_finish<bf16_t>, dim3(tiles), dim3(256), 0, st, slab, splitk, pixels, n, n1, bias,
                           act, slope, (char*)out1, ld1, (char*)out2, ld2, psum, psq, slab_ld);
    } else {
        hipLaunchKernelGGL(k_splitk_finish<float>, dim3(tiles), dim3(256), 0, st, slab, splitk, pixels, n, n1, bias,
                           act, slope, (char*)out1, ld1, (char*)out2, ld2, psum, psq, slab_ld);
    }
    PFA_LAUNCHED("k_splitk_finish");
    return 0;
}

extern "C" int pfa_gather_gemm(pfa_gather_args* a, pfa_stream s) {
    GatherK k; int classes;
    if (int rc = fill_params(a, k, classes)) return rc;
    PFA_REQUIRE(!(a->fuse_out == 1 || a->fuse_out == 2) || (a->fuse_mean != nullptr) == (a->psum != nullptr),
                "pfa_gather_gemm: a fused BatchNorm backward needs psum/psq, a fused activation backward must not pass them");
    hipStream_t st = (hipStream_t)s;
    // PFA_FUSE_DEFER: the K slices stay in the slab for a finish of the caller's (pfa_finish_bn_* / pfa_finish_coop_*): the
    // stride-2 / transposed patch kernel runs them where its plan keeps more than one slice, else the implicit GEMM (with the
    // caller's slices: pfa_gather_plan_splitk tells beforehand)
    const bool defer = a->fuse_out == PFA_FUSE_DEFER;
    if (defer) k.fuse_out = 0;
    int ptiles = 0, pbn = 128, psk = 1;
    const bool patch = !defer && patch_route(a, k, &ptiles, &pbn, &psk);
    if (patch) k.splitk = psk;                     // (un-sliced wherever the tiles alone fill the chip: never more than the caller's)
    int itiles = 0;
    const bool image = !defer && !patch && k.splitk == 1 && pfa_conv_image_ok(k, a->dtype, &itiles);
    Patch2Plan plan2;
    const bool patch2 = !patch && !image && pfa_conv_patch2_plan(k, a->dtype, &plan2) && (!defer || plan2.splitk > 1);
    if (patch2) k.splitk = plan2.splitk;          // never more slices than the caller sized the slab for
    const TileCfg cfg = choose_cfg(k);
    int rc;
    GatherK kk = k;
    if (k.splitk > 1) { kk.bias = nullptr; kk.psum = kk.psq = nullptr; }   // applied by the finish pass
    hipEvent_t te0, te1;
    const bool timing = pfa_timing_begin(st, &te0, &te1);
    if (patch) {
        if (pfa_knob(3) == 6)           // knob 3 = 6: the 16x16x32 form (A/B)
            rc = pfa_conv_s1_patch(a->mode == PFA_CONV_S1_DGRAD ? 1 : 0, a->src1, a->w, a->out1, k.slab, k.splitk, kk.psum, kk.psq,
                                   a->nb, a->hs, a->ws, a->c1, a->n, st);
        else
            rc = pfa_conv_s1_patch32(a->mode == PFA_CONV_S1_DGRAD ? 1 : 0, kk, pbn, a->hs, a->ws, a->c1, st);
        g_last_gather_name = pbn != 128 ? (pfa_knob(3) == 8 ? "k_conv_s1_patch<bf16,8x32,64>" : "k_conv_s1_inplace<bf16,8x32,64>")
                             : pfa_knob(3) == 6 || pfa_knob(3) == 8 ? "k_conv_s1_patch<bf16,8x32,128>" : "k_conv_s1_inplace<bf16,8x32,128>";
    } else if (image) {
        const char* nm = "";
        rc = pfa_conv_image(kk, st, &nm);
        g_last_gather_name = nm;
    } else if (patch2) {
        const char* nm = "";
        rc = pfa_conv_patch2(kk, plan2, st, &nm);
        g_last_gather_name = nm;
    } else {
        rc = a->dtype == PFA_BF16 ? dispatch_gather<bf16_t>(kk, classes, cfg.id, st)
                                  : dispatch_gather<float>(kk, classes, cfg.id, st);
    }
    if (timing) {
        const double es = a->dtype == PFA_BF16 ? 2.0 : 4.0, taps = a->mode == PFA_TCONV_S2 ? 4.0 : 16.0;
        // algorithmic FLOPs = 2 x the MACs of the convolution this launch is (or is the gradient of), on REAL channels
        // (a->c_real; SURVEY 8d counts the same): a stride-1 input gradient has the forward's MACs, i.e. its rows are
        // the source grid (hs x ws), not the padded position grid
        const double creal = a->c_real > 0 ? a->c_real : k.ctot;
        const double rows = a->mode == PFA_CONV_S1_DGRAD ? (double)a->nb * a->hs * a->ws : (double)k.M * classes;
        const double flop = 2.0 * rows * k.n * taps * creal;
        // unique operand + result bytes: both sources, the weight rows used, the output rows written
        const double bytes = es * ((double)a->nb * a->hs * a->ws * k.ctot + (double)k.n * 16.0 * k.ctot +
                                   (double)k.npix * (a->ld1 + a->ld2));
        pfa_timing_end(st, te0, te1, g_last_gather_name, flop, bytes, k.splitk, patch ? ptiles * (a->n / pbn) * k.splitk : 0);
    }
    if (rc) return rc;
    if (defer) {
        a->ntiles_stat = 0;
        a->splitk = k.splitk;                      // the planes actually written (no empty slices)
        return 0;
    }
    if (k.splitk > 1) {
        int tiles = 0;
        // split outputs: slab columns [0,n) -> out1 [0,n1) / out2; a single output passes n1 >= n
        const bool aff = a->fuse_out == 3;
        FinishFuse fz{};
        if (a->fuse_out == 1 || a->fuse_out == 2) {
            fz.out = a->fuse_out; fz.act1 = a->fuse_act1; fz.act2 = a->fuse_act2; fz.nreal = k.fuse_nreal;
            fz.stat_col0 = k.stat_col0; fz.stat_n = k.stat_n;
            fz.raw = (const char*)a->fuse_raw; fz.g2 = (const char*)a->fuse_g2;
            fz.scale = a->fuse_scale; fz.shift = a->fuse_shift; fz.mean = a->fuse_mean; fz.invstd = a->fuse_invstd;
            PFA_REQUIRE(!k.coef_unaligned, "pfa_gather_gemm: fused backward of a K-sliced launch needs 16-byte aligned coefficient arrays");
        }
        rc = splitk_finish(k.slab, k.slab_ld, k.splitk, k.npix, k.n, k.n1, k.bias, k.act, k.slope, a->dtype, k.out1, k.ld1,
                           k.out2, k.ld2, k.psum, k.psq, &tiles, s, aff ? a->fuse_scale : nullptr,
                           aff ? a->fuse_shift : nullptr, a->fuse_act2, fz);
        if (rc) return rc;
        a->ntiles_stat = tiles;
    } else {
        a->ntiles_stat = patch ? ptiles : image ? itiles : patch2 ? plan2.stat_tiles : classes * ((k.M + cfg.bm - 1) / cfg.bm);
    }
    return 0;
}
