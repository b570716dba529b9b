// Stride-2 4x4 convolution (PFA_CONV_S2) and transposed convolution (PFA_TCONV_S2) with the input PATCH kept in LDS.
//
// Replaces, for the MFMA-heavy layers: nn.Conv2d(k4,s2,p1) forward (models.py:177,230) and the input gradient of
// nn.ConvTranspose2d (CONV_S2 geometry over dy), nn.ConvTranspose2d(k4,s2,p1) forward (models.py:189,194) and the input
// gradient of the stride-2 Conv2d (TCONV_S2 geometry) - the launches pfa_gather_gemm used to run on its 256x128 ring tile.
//
// Both geometries are sums of FOUR 2x2-tap stride-1 convolutions:
//   TCONV_S2  output parity class (py, px) of a workgroup: out(2qy+py, 2qx+px) = sum_{a,b in {0,1}} in(qy+py-1+a, qx+px-1+b)
//             * w[(3-py-2a)*4 + (3-px-2b)]                                    - one patch per 64-channel chunk;
//   CONV_S2   out(qy, qx) = sum over the four input parity planes (ey, ex) of sum_{a,b} in(2(qy+a)-1+ey, 2(qx+b)-1+ex)
//             * w[(2a+ey)*4 + 2b+ex]                                          - one patch per chunk AND plane.
// A "super-step" is one such 2x2-tap convolution over one 64-channel chunk: the workgroup stages its (TY+1) x (TX+1)
// input pixels ONCE (LDS-DMA, 144 bytes per pixel: 16 consecutive pixels fall on 16 distinct bank groups, a tap shift is a
// wave-uniform address offset) and walks the four taps inside LDS; only the weight tile of a tap step (BN rows x 128 B)
// streams through a 3-slot ring.  DMA bytes per FLOP: 1/157 against 1/87 for the ring tile, and every input pixel is
// fetched once per workgroup instead of four times.
//
// K loop: ONE barrier per tap step, no phases.  The fragments of step i+1 are read from LDS into a second register set
// BETWEEN the MFMAs of step i (one ds_read_b128 per two MFMAs: the LDS read pipe and the MFMA pipe of a SIMD run side by
// side, nothing waits for a read issued right behind a barrier) and retired by one lgkmcnt(0) at the end of the step.
// Because a stage's fragments are in registers a full step before its MFMAs, its ring slot is refilled as soon as the
// step begins: two weight stages and up to two patch pieces are in flight across every barrier (counted vmcnt).
//   step i:  barrier | DMA W(i+3) -> slot i%3, patch pieces of a later super-step | 32 MFMAs(i) interleaved with the 16
//            reads of stage i+1 | lgkmcnt(0) | vmcnt(what this step issued)
//   RAW: data issued in step k is waited for at the end of step k+1 by every wave, i.e. visible after barrier k+2; stage
//        i+1 (issued in step i-2) is read during step i.  Patch S (steps 4S..4S+3, first read during step 4S-1) is
//        issued in steps 4S-5..4S-3.
//   WAR: slot i%3 was read during step i-1 (retired before barrier i); patch buffer S%2 was last read during step 4S-6
//        (the reads of step 4S-5), retired before barrier 4S-5.
#include "patch_epilogue.h"

#include <type_traits>

namespace {

constexpr int PSTRIDE = 144;                      // bytes per patch pixel: 128 + one unused 16-byte piece
constexpr int PPASS = 6;                          // 6 x 512 lanes x 16 B = 341 pixel slots >= 9x33 = 297, 17x17 = 289
constexpr int PATCH_BYTES = PPASS * 512 * 16;     // 49,152
constexpr int W_SLOTS = 3;

struct Patch2K {
    GatherK g;
    int tiles_y, tiles_x, ntiles, nwg;
    int nss, ssper;          // super-steps in all / per K slice
    int c1chunks;            // 64-channel chunks that come from src1
    unsigned long long* stamps;   // developer timing (pfa_set_debug_buffer): 4 clock stamps per workgroup, or NULL
};

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// MODE 0 = PFA_TCONV_S2, 1 = PFA_CONV_S2.  TILE16: 16 x 16 position tile instead of 8 x 32.  BN: 128 or 64 output channels.
// ABL: ablation bits of a diagnostic instantiation (1 no MFMA, 2 no fragment reads, 4 no DMA - results are wrong); 0 in every
// instantiation the library launches by default.
template <int MODE, int BN, int TILE16, int ABL = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void k_conv_patch2(const Patch2K q) {
    typedef bf16_t T;
    constexpr int TY = TILE16 ? 16 : 8, TX = TILE16 ? 16 : 32;
    constexpr int PH = TY + 1, PW = TX + 1, PSLOTS = PH * PW;
    constexpr int FN = BN / 32, FM = 4;                    // 4 x 2 waves: 64 positions x BN/2 channels per wave
    constexpr int W_BYTES = BN * 128, WP = BN / 64;        // weight tile of a step, DMA pieces per wave
    constexpr int NREAD = 2 * (FN + FM), NMFMA = 2 * FN * FM;
    static_assert(PSLOTS * 9 <= PPASS * 512, "patch does not fit its buffer");
    const GatherK& p = q.g;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int lrow = lane & 15, kk = lane >> 4;
    // work list: n-tile fastest (the workgroups that share a patch sit next to each other on one XCD), then K slice,
    // parity class, tile, image
    unsigned wid = xcd_remap(blockIdx.x, q.nwg);
    const int nt = wid % q.ntiles; wid /= q.ntiles;
    const int split = wid % p.splitk; wid /= p.splitk;
    int cls = 0;
    if (MODE == 0) { cls = wid & 3; wid >>= 2; }
    const int tx = wid % q.tiles_x; wid /= q.tiles_x;
    const int ty = wid % q.tiles_y, img = wid / q.tiles_y;
    const int py = cls >> 1, px = cls & 1;
    const int qy0 = ty * TY, qx0 = tx * TX, n0 = nt * BN;
    const char* zero = (const char*)g_pfa_zero_page;
    auto stamp = [&](int k) {          // shader clock at four points of the workgroup's life (a separate measurement, never a result)
        if (q.stamps && tid == 0) q.stamps[(size_t)blockIdx.x * 4 + k] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);

    // ---- patch staging: lane's piece j = (pixel slot, 16-byte piece g of its 128 bytes); the ninth piece is padding
    int ppix[PPASS];             // source pixel index (MODE 1: of plane (0,0); plane (ey,ex) adds ey*ws + ex)
    unsigned pok = 0;            // validity bits: MODE 0 bit j; MODE 1 bit plane*8 + j
#pragma unroll
    for (int j = 0; j < PPASS; ++j) {
        const int qq = j * 512 + tid, slot = qq / 9, g = qq - slot * 9;
        const int sy = slot / PW, sx = slot - sy * PW;
        const bool inb = g < 8 && slot < PSLOTS;
        if (MODE == 0) {
            const int iy = qy0 + py - 1 + sy, ix = qx0 + px - 1 + sx;
            const bool ok = inb && (unsigned)iy < (unsigned)p.hs && (unsigned)ix < (unsigned)p.ws;
            ppix[j] = ((img * p.hs + iy) * p.ws + ix) * 8 + (g & 7);      // pixel * 8 + piece: one multiply by C/8 later
            pok |= ok ? (1u << j) : 0u;
        } else {
            const int iy = 2 * (qy0 + sy) - 1, ix = 2 * (qx0 + sx) - 1;
            ppix[j] = ((img * p.hs + iy) * p.ws + ix) * 8 + (g & 7);
#pragma unroll
            for (int pl = 0; pl < 4; ++pl) {
                const int y = iy + (pl >> 1), x = ix + (pl & 1);
                const bool ok = inb && (unsigned)y < (unsigned)p.hs && (unsigned)x < (unsigned)p.ws;
                pok |= ok ? (1u << (pl * 8 + j)) : 0u;
            }
        }
    }
    const int ss0 = split * q.ssper;
    const int nss = min(q.nss, ss0 + q.ssper) - ss0;          // super-steps of this slice
    const int nsteps = nss * 4;
    // super-step S of the slice -> (source, channel chunk, plane)
    auto stage_patch = [&](int S, int j0, int j1) {
        const int gs = ss0 + S;
        const int chunk = MODE == 0 ? gs : gs >> 2, pl = MODE == 0 ? 0 : gs & 3;
        const bool first = chunk < q.c1chunks;
        const char* base = first ? p.src1 : p.src2;
        const int C = first ? p.c1 : p.c2, cc = first ? chunk : chunk - q.c1chunks;
        const int c8 = C >> 3;                               // 16-byte pieces per pixel
        const int pladd = MODE == 0 ? 0 : ((pl >> 1) * p.ws + (pl & 1)) * 8;
        char* dst = smem + (S & 1) * PATCH_BYTES;
#pragma unroll
        for (int j = j0; j < j1; ++j) {
            const bool ok = (pok >> ((MODE == 0 ? 0 : pl * 8) + j)) & 1u;
            // element offset = pixel * C + cc * 64 + g * 8  =  ((pixel*8 + g) + plane) ... with C = 8 * c8:
            const int pg = ppix[j] + pladd;                  // pixel * 8 + g
            const int pix = pg >> 3, g = pg & 7;
            const unsigned off = (unsigned)((pix * c8 * 8 + cc * 64 + g * 8) * 2);
            const char* gp = ok ? base + off : zero;
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gp), LDS_PTR(dst + (j * 512 + wave * 64) * 16), 16, 0, 0);
        }
    };
    // ---- weight staging: WP pieces per step (rows r0 + 64 i of the tile), chunk slot cs holds source chunk cs ^ swz(row)
    const int cs = tid & 7, r0 = tid >> 3;
    const int wgc = cs ^ ((r0 >> 1) & 7);
    unsigned woff[WP];
#pragma unroll
    for (int i = 0; i < WP; ++i) woff[i] = (unsigned)(((n0 + r0 + 64 * i) * 16 * p.ctot + wgc * 8) * 2);
    auto stage_w = [&](int step) {                            // step of this slice: super-step step >> 2, tap step & 3
        const int gs = ss0 + (step >> 2), t = step & 3, a = t >> 1, b = t & 1;
        int chunk, widx;
        if (MODE == 0) { chunk = gs; widx = (3 - py - 2 * a) * 4 + (3 - px - 2 * b); }
        else { chunk = gs >> 2; const int pl = gs & 3; widx = (2 * a + (pl >> 1)) * 4 + 2 * b + (pl & 1); }
        char* wb = smem + 2 * PATCH_BYTES + (step % W_SLOTS) * W_BYTES;
#pragma unroll
        for (int i = 0; i < WP; ++i) {
            const char* gp = p.w + woff[i] + (unsigned)((widx * p.ctot + chunk * 64) * 2);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gp), LDS_PTR(wb + (i * 512 + wave * 64) * 16), 16, 0, 0);
        }
    };

    f32x4 acc[FN][FM];
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // activation fragment b of this wave = 16 consecutive positions of one tile row
    const unsigned lds0 = (unsigned)(size_t)LDS_PTR(smem);
    int frow[FM], fcol[FM];
    unsigned abyte[FM];
#pragma unroll
    for (int b = 0; b < FM; ++b) {
        frow[b] = TILE16 ? wm * 4 + b : wm * 2 + (b >> 1);
        fcol[b] = TILE16 ? 0 : (b & 1) * 16;
        abyte[b] = lds0 + (frow[b] * PW + fcol[b] + lrow) * PSTRIDE + kk * 16;
    }
    unsigned wfb[2];
    {
        const unsigned sw = (lrow >> 1) & 7;
#pragma unroll
        for (int h = 0; h < 2; ++h) wfb[h] = lds0 + 2 * PATCH_BYTES + (wn * (BN / 2) + lrow) * 128 + (((kk + 4 * h) ^ sw) << 4);
    }
    struct Frags { bf16x8 w[2][FN], a[2][FM]; };
    // read r (0 .. NREAD-1) of a step: k-half h = r / (FN+FM), then the FN weight fragments, then the FM activation ones
    auto read_one = [&](Frags& f, int step, int r) {
        const int t = step & 3, S = step >> 2;
        const unsigned tb = (S & 1) * PATCH_BYTES + ((t >> 1) * PW + (t & 1)) * PSTRIDE;
        const unsigned rb = (step % W_SLOTS) * W_BYTES;
        const int h = r / (FN + FM), e = r - h * (FN + FM);
        if (e < FN) {
            const unsigned ad = wfb[h] + rb;
            if (e == 0) f.w[h][0] = lds_read16<bf16x8, 0>(ad);
            else if (e == 1) f.w[h][1] = lds_read16<bf16x8, 2048>(ad);
            else if (e == 2) f.w[h][e] = lds_read16<bf16x8, 4096>(ad);
            else f.w[h][e] = lds_read16<bf16x8, 6144>(ad);
        } else {
            const int b = e - FN;
            if (h == 0) f.a[0][b] = lds_read16<bf16x8, 0>(abyte[b] + tb);
            else f.a[1][b] = lds_read16<bf16x8, 64>(abyte[b] + tb);
        }
    };
    auto pin_all = [&](Frags& f) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int a = 0; a < FN; ++a) pin(f.w[h][a]);
#pragma unroll
            for (int b = 0; b < FM; ++b) pin(f.a[h][b]);
        }
    };

    // ---- prologue: patch 0, weight stages 0..2, the first two pieces of patch 1
    constexpr bool no_mfma = ABL & 1, no_read = ABL & 2, no_dma = ABL & 4;
    if (!no_dma) {
        stage_patch(0, 0, PPASS);
        stage_w(0);
        if (nsteps > 1) stage_w(1);
        if (nsteps > 2) stage_w(2);
        if (nss > 1) stage_patch(1, 0, 2);
    }
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    Frags fA, fB;
#pragma unroll
    for (int r = 0; r < NREAD; ++r) read_one(fA, 0, r);
    PFA_WAIT_LGKM(0);
    pin_all(fA);
    stamp(1);

    // one tap step; LAST: no step follows (nothing to read ahead)
    auto body = [&](int i, Frags& cur, Frags& nxt, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // DMA of this step
        bool w_issued = false, p_issued = false;
        if (!no_dma) {
            if (i + 3 < nsteps) { stage_w(i + 3); w_issued = true; }
            const int rel = i & 3, S = i >> 2;
            if (rel == 3) { if (S + 2 < nss) { stage_patch(S + 2, 0, 2); p_issued = true; } }
            else if (rel == 0) { if (S + 1 < nss) { stage_patch(S + 1, 2, 4); p_issued = true; } }
            else if (rel == 1) { if (S + 1 < nss) { stage_patch(S + 1, 4, 6); p_issued = true; } }
        }
        __builtin_amdgcn_sched_barrier(0);
        // MFMAs of step i with the reads of step i+1 between them
        int rnext = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int a = 0; a < FN; ++a)
#pragma unroll
                for (int b = 0; b < FM; ++b) {
                    if (!no_mfma) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur.w[h][a], cur.a[h][b], acc[a][b], 0, 0, 0);
                    const int m = (h * FN + a) * FM + b;                 // MFMA index 0 .. NMFMA-1
                    const int want = ((m + 1) * NREAD) / NMFMA;          // reads that should have been issued after it
                    if (want > rnext) {
                        if (!LAST && !no_read) {
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int r = rnext; r < want; ++r) read_one(nxt, i + 1, r);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        rnext = want;
                    }
                }
        __builtin_amdgcn_sched_barrier(0);
        if (!LAST && !no_read) {
            PFA_WAIT_LGKM(0);
            pin_all(nxt);
        }
        if (w_issued && p_issued) wait_vm<WP + 2>();
        else if (w_issued) wait_vm<WP>();
        else if (p_issued) wait_vm<2>();
        else wait_vm<0>();
    };
    std::false_type more;
    std::true_type last;
    int i = 0;                                   // nsteps is a multiple of 4
    for (; i + 2 < nsteps; i += 2) {
        body(i, fA, fB, more);
        body(i + 1, fB, fA, more);
    }
    body(i, fA, fB, more);
    body(i + 1, fB, fA, last);
    stamp(2);

    // ---- epilogue: through LDS, whole NHWC rows out (patch_epilogue.h)
    int prow[FM];
#pragma unroll
    for (int b = 0; b < FM; ++b) prow[b] = frow[b] * TX + fcol[b];
    auto pix_of = [&](int r, bool& valid, size_t& opix) {
        const int qy = qy0 + r / TX, qx = qx0 + r % TX;
        valid = qy < p.hq && qx < p.wq;
        if (MODE == 0) opix = valid ? ((size_t)img * p.ho + 2 * qy + py) * p.wo + 2 * qx + px : 0;
        else opix = valid ? ((size_t)img * p.hq + qy) * p.wq + qx : 0;
    };
    const size_t trow = (((size_t)img * q.tiles_y + ty) * q.tiles_x + tx) * (MODE == 0 ? 4 : 1) + cls;
    patch_epilogue<BN>(p, acc, prow, n0, wn, tid, split, trow, smem, pix_of);
    if (q.stamps) {
        __builtin_amdgcn_s_waitcnt(0);  // (stamps only: the stores have been issued AND accepted)
        stamp(3);
    }
}

template <int MODE, int BN, int TILE16>
int launch_patch2(const Patch2K& k, hipStream_t st) {
    constexpr int LDS = 2 * PATCH_BYTES + W_SLOTS * BN * 128;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)k_conv_patch2<MODE, BN, TILE16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_done = true;
    }
    hipLaunchKernelGGL((k_conv_patch2<MODE, BN, TILE16>), dim3(k.nwg), dim3(512), LDS, st, k);
    PFA_LAUNCHED("k_conv_patch2");
    return 0;
}

}  // namespace

// Launches the patch kernels take: bf16, stride-2 conv / transposed conv, every source a multiple of 64 channels, n a
// multiple of 64, maps at least 8 x 16 positions (smaller ones are weight-streaming launches: they stay K-sliced on the
// implicit GEMM), no tanh.  BN = 128 when that already gives the chip ~one workgroup per CU, else 64; K slices (over
// super-steps, at most the caller's count) only when even 64-channel tiles leave most CUs idle.
bool pfa_conv_patch2_plan(const GatherK& k, int dtype, Patch2Plan* plan) {
    if (pfa_knob(3) == 1 || pfa_knob(3) == 3) return false;             // knob 3: 1 = no patch kernels at all, 3 = stride-1 only
    if (dtype != PFA_BF16 || !(k.mode == PFA_TCONV_S2 || k.mode == PFA_CONV_S2) || !k.small32) return false;
    if (k.c1 % 64 || k.c2 % 64 || k.n % 64 || k.act == PFA_ACT_TANH) return false;
    if (k.hq < 8 || k.wq < 16) return false;
    Patch2Plan pl{};
    pl.tile16 = k.wq < 32 ? 1 : 0;
    const int tyy = pl.tile16 ? 16 : 8, txx = pl.tile16 ? 16 : 32;
    pl.tiles_y = (k.hq + tyy - 1) / tyy; pl.tiles_x = (k.wq + txx - 1) / txx;
    const int classes = k.mode == PFA_TCONV_S2 ? 4 : 1;
    const long long tiles = (long long)k.nb * pl.tiles_y * pl.tiles_x * classes;
    const int nss = (k.ctot / 64) * (k.mode == PFA_CONV_S2 ? 4 : 1);
    const int want = pfa_knob(6) > 0 ? pfa_knob(6) : 192;
    pl.splitk = 1;
    if (k.n % 128 == 0 && tiles * (k.n / 128) >= want) pl.bn = 128;
    else if (tiles * (k.n / 64) >= want || k.splitk <= 1) pl.bn = 64;
    else {
        pl.bn = 64;
        int sk = k.splitk < nss ? k.splitk : nss;
        while (sk > 2 && tiles * (k.n / 64) * (sk / 2) >= want) sk /= 2;     // no more slices than the chip needs
        pl.splitk = sk;
    }
    if (pfa_knob(3) != 2 && tiles * (k.n / pl.bn) * pl.splitk < 96) return false;      // too small to be worth a patch launch
    pl.stat_tiles = (int)tiles;
    if (plan) *plan = pl;
    return true;
}

int pfa_conv_patch2(const GatherK& k, const Patch2Plan& plan, hipStream_t st, const char** kernel_name) {
    Patch2K q;
    q.g = k;
    q.g.splitk = plan.splitk;
    q.tiles_y = plan.tiles_y; q.tiles_x = plan.tiles_x;
    q.ntiles = k.n / plan.bn;
    const int classes = k.mode == PFA_TCONV_S2 ? 4 : 1;
    q.nwg = k.nb * plan.tiles_y * plan.tiles_x * classes * q.ntiles * plan.splitk;
    q.nss = (k.ctot / 64) * (k.mode == PFA_CONV_S2 ? 4 : 1);
    q.ssper = (q.nss + plan.splitk - 1) / plan.splitk;
    PFA_REQUIRE((q.nss + q.ssper - 1) / q.ssper == plan.splitk, "pfa_conv_patch2: empty K slice");
    q.c1chunks = k.c1 / 64;
    q.stamps = nullptr;
    if (g_pfa_dbgbuf && g_pfa_dbgbuf_bytes >= (size_t)q.nwg * 32) q.stamps = (unsigned long long*)g_pfa_dbgbuf;
    const int sel = (k.mode == PFA_CONV_S2 ? 4 : 0) + (plan.bn == 64 ? 2 : 0) + (plan.tile16 ? 1 : 0);
    static const char* names[8] = {"k_conv_patch2<tconv,128,8x32>", "k_conv_patch2<tconv,128,16x16>", "k_conv_patch2<tconv,64,8x32>",
                                   "k_conv_patch2<tconv,64,16x16>", "k_conv_patch2<conv_s2,128,8x32>", "k_conv_patch2<conv_s2,128,16x16>",
                                   "k_conv_patch2<conv_s2,64,8x32>", "k_conv_patch2<conv_s2,64,16x16>"};
    if (kernel_name) *kernel_name = names[sel];
    switch (sel) {
        case 0: return launch_patch2<0, 128, 0>(q, st);
        case 1: return launch_patch2<0, 128, 1>(q, st);
        case 2: return launch_patch2<0, 64, 0>(q, st);
        case 3: return launch_patch2<0, 64, 1>(q, st);
        case 4: return launch_patch2<1, 128, 0>(q, st);
        case 5: return launch_patch2<1, 128, 1>(q, st);
        case 6: return launch_patch2<1, 64, 0>(q, st);
        default: return launch_patch2<1, 64, 1>(q, st);
    }
}
