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
// input pixels ONCE (LDS-DMA, 144 bytes per pixel: a tap shift is a constant address offset) and walks the four taps
// inside LDS; only the weight tile of a tap step (BN rows x 128 B) streams through a 4-slot ring.  DMA bytes per FLOP:
// 1/157 against 1/87 for the ring tile, and every input pixel is fetched once per workgroup instead of four times.
//
// MFMA 32x32x16 (a wave = 64 positions x BN/2 channels = 2 x {2|1} tiles of 32x32): per FLOP it holds the SIMD's issue
// port half as long as 16x16x32 - the K loop of the first version (16x16x32) was bound by instruction issue, not by the
// matrix pipe: 1750 cycles per step against 1024 of MFMA - and all 16 lanes of a ds_read_b128 lane group then read the
// SAME k-chunk, so 32 consecutive 144-byte pixels fall on distinct bank slots (with 16x16x32 a lane group mixes two
// k-chunks, which collide at that pitch: every activation read took two passes, measured as ~137 B/clk/CU).
//
// K loop: ONE barrier per tap step, and ONE fragment register set: a step is four k-steps (16 channels each); right after
// the MFMAs of k-step s have been issued, the fragments of k-step s of the NEXT tap step are read into the registers they
// just consumed (ds_read_b128, retired by counted lgkmcnt one full step later: LDS latency never shows, nothing waits for a
// read issued behind a barrier).  The loop is unrolled over the 8 (patch buffer, tap) combinations so that every LDS read
// address is a per-lane register plus an immediate: no address arithmetic in the loop.
// Because a stage's fragments are read a full step before its MFMAs, its ring slot is free early: two weight stages and
// up to two patch pieces are in flight across every barrier (counted vmcnt).
//   step i:  barrier | per k-step s: lgkmcnt(3 k-steps) - 4 MFMAs(i, s) - 4 reads(i+1, s) - part of the DMA (W(i+3) ->
//            slot (i+3)%4, patch pieces of a later super-step) | vmcnt(what this step issued)
//   RAW: data issued in step k is waited for at the end of step k+1 by every wave, i.e. visible after barrier k+2; the
//        fragments of stage i+1 (issued in step i-2) are read during step i.  Patch S (steps 4S..4S+3, first read during
//        step 4S-1) is issued in steps 4S-4 and 4S-3, three pieces each.
//   WAR: slot (i+3)%4 held stage i-1, whose reads (issued during step i-2) every wave retired during step i-1, i.e.
//        before barrier i; patch buffer S%2 was last read by the fragments of step 4S-5, issued during step 4S-6 and
//        retired during step 4S-5, i.e. before barrier 4S-4.
#include "patch_epilogue.h"

#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int PSTRIDE = 144;                      // bytes per patch pixel: 128 + one unused 16-byte piece
constexpr int PPASS = 6;                          // 6 x 512 lanes x 16 B = 341 pixel slots >= 9x33 = 297, 17x17 = 289
constexpr int PATCH_BYTES = PPASS * 512 * 16;     // 49,152
constexpr int W_SLOTS = 4;

struct Patch2K {
    GatherK g;
    int tiles_y, tiles_x, ntiles, nwg;
    int nss, ssper;          // super-steps in all / per K slice
    int c1chunks;            // 64-channel chunks that come from src1
    unsigned long long* stamps;   // developer timing (pfa_set_debug_buffer): 16 slots per workgroup (8 shader-clock stamps, 2 of the 100 MHz real-time counter), or NULL
};

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// Fragment read INTO the register that held the same fragment of the previous step ("+v": the allocator must reuse it -
// with a fresh output per read it spreads the fragments of the unrolled steps over different registers and pays ~70
// v_mov per step to line them up again at the loop's back edge).  The MFMA that consumed the old value was issued before
// this instruction and has read its A/B operands long before the LDS data (>= 64 cycles away) lands.
template <int OFF> __device__ __forceinline__ void lds_rd(unsigned addr, bf16x8& v) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(v) : "v"(addr), "n"(OFF));
}

// MODE 0 = PFA_TCONV_S2, 1 = PFA_CONV_S2.  TILE16: 0 = 8 x 32 position tile, 1 = 16 x 16, 2 = FOUR IMAGES x 8 x 8 (the 8 x 8 maps of the
// generator's fifth level: e5 / d5 and their input gradients - four 9 x 9 patches in the patch buffer, a wave per image).
// BN: 128 or 64 output channels.
// ABL: ablation bits of a diagnostic instantiation (1 no MFMA, 2 no fragment reads, 4 no DMA - results are wrong); 0 in every
// instantiation the library launches.
template <int MODE, int BN, int TILE16, int ABL = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void k_conv_patch2(const Patch2K q) {
    typedef bf16_t T;
    constexpr int NIMG = TILE16 == 2 ? 4 : 1;              // images of a tile
    constexpr int TY = TILE16 == 1 ? 16 : 8, TX = TILE16 == 1 ? 16 : TILE16 == 2 ? 8 : 32;
    constexpr int PH = TY + 1, PW = TX + 1, PSLOTS = NIMG * PH * PW;
    constexpr int FN = BN / 64, FM = 2;                    // 4 x 2 waves: 64 positions x BN/2 channels per wave, 32x32 tiles
    constexpr int KS = 4;                                  // MFMA k-steps (16 channels) per 64-channel chunk
    constexpr int W_BYTES = BN * 128, WP = BN / 64;        // weight tile of a step, DMA pieces per wave
    constexpr int NREAD = KS * (FN + FM), NMFMA = KS * FN * FM;
    constexpr int WBASE = 2 * PATCH_BYTES;
    static_assert(PSLOTS * 9 <= PPASS * 512, "patch does not fit its buffer");
    static_assert(3 * W_BYTES < 65536 && PATCH_BYTES + (PW + 1) * PSTRIDE + 128 < 65536, "ds offset field");
    const GatherK& p = q.g;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int lr = lane & 31, lh = lane >> 5;
    // work list: n-tile fastest (the workgroups that share a patch sit next to each other on one XCD), then K slice,
    // parity class, tile, image
    unsigned wid = xcd_remap(blockIdx.x, q.nwg);
    const int nt = wid % q.ntiles; wid /= q.ntiles;
    const int split = wid % p.splitk; wid /= p.splitk;
    int cls = 0;
    if (MODE == 0) { cls = wid & 3; wid >>= 2; }
    const int tx = wid % q.tiles_x; wid /= q.tiles_x;
    const int ty = wid % q.tiles_y, img = (int)(wid / q.tiles_y) * NIMG;      // (the tile's first image)
    const int py = cls >> 1, px = cls & 1;
    const int qy0 = ty * TY, qx0 = tx * TX, n0 = nt * BN;
    auto stamp = [&](int k) {          // shader clock at four points of the workgroup's life (a separate measurement, never a result)
        if (q.stamps && tid == 0) q.stamps[(size_t)blockIdx.x * 16 + k] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);
    if (q.stamps && tid == 0) q.stamps[(size_t)blockIdx.x * 16 + 8] = __builtin_amdgcn_s_memrealtime();     // 100 MHz, chip-wide

    constexpr bool no_mfma = ABL & 1, no_read = ABL & 2, no_dma = ABL & 4, no_vmwait = ABL & 8, no_barrier = ABL & 16;
    const int ss0 = split * q.ssper;
    const int nss = min(q.nss, ss0 + q.ssper) - ss0;          // super-steps of this slice
    const int nsteps = nss * 4;
    const unsigned C2 = (unsigned)p.c1 * 2, C2b = (unsigned)p.c2 * 2;        // bytes per pixel of the two sources
    const unsigned ctot2 = (unsigned)p.ctot * 2;
    // All DMA is the buffer form of pfa_common.h (buf_dma16): a per-lane offset that is fixed for the whole K loop plus ONE
    // scalar per piece for what changes from step to step; a lane that is to read zeros carries PFA_DMA_OOB.
    // A super-step (S) behind the slice's last one gets out-of-range scalars: its pieces are still issued (zeros into ring
    // slots and a patch buffer nobody reads, no memory traffic), so every step issues the same pieces and waits with the same
    // counted vmcnt - no conditions in the loop.
    // ---- weight staging: WP pieces per step (rows r0 + 64 i of the tile), chunk slot cs holds source chunk cs ^ swz(row).
    // Scalar: super-step's share (channel chunk; MODE 1: the plane's part of the tap index) + the tap's share, one of four
    // per-workgroup constants.  Issued FIRST: the first stages' latency runs under the patch's address arithmetic below.
    const int cs = tid & 7, r0 = tid >> 3;
    const int wgc = cs ^ ((r0 >> 1) & 7);
    unsigned woff[WP];
#pragma unroll
    for (int i = 0; i < WP; ++i) woff[i] = (unsigned)(((n0 + r0 + 64 * i) * 16 * p.ctot + wgc * 8) * 2);
    unsigned wtap[4];            // tap t = (a, b): MODE 0 tap index (3 - py - 2a) * 4 + (3 - px - 2b), MODE 1 (2a) * 4 + 2b (+ the plane's part)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int a = t >> 1, b = t & 1;
        wtap[t] = (unsigned)(MODE == 0 ? (3 - py - 2 * a) * 4 + (3 - px - 2 * b) : 8 * a + 2 * b) * ctot2;
    }
    auto wsoff_of = [&](int S) -> unsigned {
        const int gs = ss0 + S;
        const int chunk = MODE == 0 ? gs : gs >> 2, pl = MODE == 0 ? 0 : gs & 3;
        return S < nss ? (unsigned)chunk * 128 + (MODE == 0 ? 0u : (unsigned)((pl >> 1) * 4 + (pl & 1)) * ctot2) : PFA_DMA_OOB;
    };
    auto stage_w_piece = [&](unsigned wsoff, auto t_, int i) {  // tap t (= weight slot) of the super-step with scalar wsoff, piece i
        constexpr int TP = decltype(t_)::value;
        char* wb = smem + WBASE + TP * W_BYTES;
        buf_dma16(p.w, wb + (i * 512 + wave * 64) * 16, woff[i], wsoff + wtap[TP]);
    };
    if (!no_dma) {               // steps 0 .. 2 (a slice has at least one super-step = 4 steps)
        const unsigned w0 = wsoff_of(0);
#pragma unroll
        for (int i = 0; i < WP; ++i) {
            stage_w_piece(w0, std::integral_constant<int, 0>{}, i);
            stage_w_piece(w0, std::integral_constant<int, 1>{}, i);
            stage_w_piece(w0, std::integral_constant<int, 2>{}, i);
        }
    }
    // ---- patch staging: lane's piece j = (pixel slot, 16-byte piece g of its 128 bytes); the ninth piece is padding.
    // pvoff[j]: the lane's byte offset inside the source tensor - pixel * pixel pitch + 16 g - fixed while the K loop walks
    // one source (pvoff2: the same for the second source's pitch, copied over pvoff when the walk gets there).  Scalar: the
    // channel chunk and, MODE 1, the parity plane.  MODE 1: pvoff is the plane-(0,0) pixel moved down-right by one row and one
    // pixel, and the source base is moved back by as much: the offsets of pixels that exist in SOME plane are then never
    // negative.  pok: validity bit plane * 8 + j.
    unsigned pvoff[PPASS], pvoff2[PPASS];
    unsigned pok = 0;
#pragma unroll
    for (int j = 0; j < PPASS; ++j) {
        const int qq = j * 512 + tid, slot = qq / 9, g = qq - slot * 9;
        const int il = NIMG == 1 ? 0 : slot / (PH * PW), rs = slot - il * (PH * PW);      // image of the tile, slot inside its patch
        const int sy = rs / PW, sx = rs - sy * PW;
        const bool inb = g < 8 && slot < PSLOTS && img + il < p.nb;
        unsigned pix, any;               // (validity as 0 / 1 integers and bitwise operators: `&&` / `||` / `?:` on per-lane
                                         //  conditions compile to exec-masked branches, two dozen of them in this prologue)
        if (MODE == 0) {
            const int iy = qy0 + py - 1 + sy, ix = qx0 + px - 1 + sx;
            any = (unsigned)inb & (unsigned)((unsigned)iy < (unsigned)p.hs) & (unsigned)((unsigned)ix < (unsigned)p.ws);
            pix = (unsigned)(((img + il) * p.hs + iy) * p.ws + ix);
        } else {
            const int iy = 2 * (qy0 + sy) - 1, ix = 2 * (qx0 + sx) - 1;
            const unsigned y0 = (unsigned)((unsigned)iy < (unsigned)p.hs), y1 = (unsigned)((unsigned)(iy + 1) < (unsigned)p.hs);
            const unsigned x0 = (unsigned)((unsigned)ix < (unsigned)p.ws), x1 = (unsigned)((unsigned)(ix + 1) < (unsigned)p.ws);
            const unsigned in1 = (unsigned)inb;
            const unsigned b4 = (0u - in1) & ((y0 & x0) | ((y0 & x1) << 8) | ((y1 & x0) << 16) | ((y1 & x1) << 24));     // plane (ey, ex) -> bit 8 (2 ey + ex)
            pok |= b4 << j;
            any = b4 != 0u;
            pix = (unsigned)(((img + il) * p.hs + iy + 1) * p.ws + ix + 1);
        }
        const unsigned keep = 0u - any;          // all ones where the lane reads pixels in some step
        pvoff[j] = ((pix * C2 + (unsigned)g * 16) & keep) | (PFA_DMA_OOB & ~keep);
        pvoff2[j] = ((pix * C2b + (unsigned)g * 16) & keep) | (PFA_DMA_OOB & ~keep);
    }
    const char* const src1s = MODE == 0 ? p.src1 : p.src1 - (size_t)(p.ws + 1) * C2;
    const char* const src2s = MODE == 0 ? p.src2 : p.src2 - (size_t)(p.ws + 1) * C2b;
    // What changes from super-step to super-step is formed ONCE per super-step (SS).  Inside a step a DMA piece then costs
    // its M0 move, MODE 1 one select on the per-lane offset, and the instruction itself.
    struct SS { unsigned wsoff; const char* pbase; unsigned psoff; unsigned bits; };
    auto make_ss = [&](int S) -> SS {                        // super-step S of this slice
        const int gs = ss0 + S;
        const int chunk = MODE == 0 ? gs : gs >> 2, pl = MODE == 0 ? 0 : gs & 3;
        const bool first = chunk < q.c1chunks;
        const int cc = first ? chunk : chunk - q.c1chunks;
        SS r;
        r.wsoff = wsoff_of(S);
        r.pbase = first ? src1s : src2s;
        r.psoff = S < nss ? (unsigned)cc * 128 + (MODE == 0 ? 0u : (unsigned)((pl >> 1) * p.ws + (pl & 1)) * (first ? C2 : C2b)) : PFA_DMA_OOB;
        if (!first) {            // the walk is in the second source (every piece of the first one has been issued)
#pragma unroll
            for (int j = 0; j < PPASS; ++j) pvoff[j] = pvoff2[j];
        }
        r.bits = MODE == 0 ? 0u : pok >> (pl * 8);
        return r;
    };
    auto stage_patch = [&](const SS& ss, int buf, int j0, int j1) {      // pieces j0 .. j1-1 into patch buffer buf
        char* dst = smem + buf * PATCH_BYTES;
#pragma unroll
        for (int j = j0; j < j1; ++j) {
            const unsigned v = MODE == 0 ? pvoff[j] : (((ss.bits >> j) & 1u) ? pvoff[j] : PFA_DMA_OOB);
            buf_dma16(ss.pbase, dst + (j * 512 + wave * 64) * 16, v, ss.psoff);
        }
    };
    SS cur = make_ss(0), nxt = cur;              // the super-step of the running step and the one behind it
    if (!no_dma) stage_patch(cur, 0, 0, PPASS);

    f32x16 acc[FN][FM];
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    // ---- fragment addresses.  32x32x16 operands: lane (lr = lane & 31, lh = lane >> 5) supplies row lr, k = 16 s + 8 lh .. +7
    // of k-step s, i.e. the 16-byte chunk 2 s + lh of the row's 128 bytes.
    // Activation tile b of this wave = 32 positions: 8x32 tile: tile row wm*2 + b, columns lr; 16x16 tile: tile rows
    // wm*4 + 2b + (lr >> 4), column ((lr & 15) - (lr >> 4) * (PW & 15)) & 15 - the second row rotated by the row pitch, so that
    // the 32 pixel slots are distinct mod 16 (conflict-free ds_read_b128: 16 lanes of a group, one k-chunk, pitch 9 * 16 B).
    const unsigned lds0 = (unsigned)(size_t)LDS_PTR(smem);
    int tpos[FM];                // tile-local position (row-major TY x TX) of this lane's row of tile b
    unsigned abyte[FM];          // its patch byte address at tap (0, 0), patch buffer 0, k-step 0
#pragma unroll
    for (int b = 0; b < FM; ++b) {
        int trow_, tcol_;
        if (TILE16 == 2) {       // wave wm = image wm of the tile; tile b = its rows 4 b .. 4 b + 3 (4 x 8 positions)
            trow_ = 4 * b + (lr >> 3); tcol_ = lr & 7;
            tpos[b] = wm * 64 + trow_ * 8 + tcol_;
            abyte[b] = lds0 + ((wm * PH + trow_) * PW + tcol_) * PSTRIDE + lh * 16;
            continue;
        }
        if (TILE16 == 1) { trow_ = wm * 4 + 2 * b + (lr >> 4); tcol_ = ((lr & 15) - (lr >> 4) * (PW & 15)) & 15; }
        else { trow_ = wm * 2 + b; tcol_ = lr; }
        tpos[b] = trow_ * TX + tcol_;
        abyte[b] = lds0 + (trow_ * PW + tcol_) * PSTRIDE + lh * 16;
    }
    unsigned wadr[FN][KS];       // weight fragment (channel tile a, k-step s): row wn*BN/2 + a*32 + lr, swizzled chunk
#pragma unroll
    for (int a = 0; a < FN; ++a) {
        const int row = wn * (BN / 2) + a * 32 + lr;
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) wadr[a][s2] = lds0 + WBASE + row * 128 + (((2 * s2 + lh) ^ ((row >> 1) & 7)) << 4);
    }
    struct Frags { bf16x8 w[FN][KS], a[FM][KS]; };
    // read r (0 .. NREAD-1) of the step with compile-time (patch buffer PB, tap TP, weight slot WS): k-step major, the FN
    // weight fragments then the FM activation ones
    auto read_one = [&](Frags& f, auto pb_, auto tp_, auto ws_, auto r_) {
        constexpr int PB = decltype(pb_)::value, TP = decltype(tp_)::value, WS = decltype(ws_)::value, R = decltype(r_)::value;
        constexpr int S2 = R / (FN + FM), E = R % (FN + FM);
        if constexpr (E < FN) lds_rd<WS * W_BYTES>(wadr[E][S2], f.w[E][S2]);
        else lds_rd<PB * PATCH_BYTES + ((TP >> 1) * PW + (TP & 1)) * PSTRIDE + S2 * 32>(abyte[E - FN], f.a[E - FN][S2]);
    };
    auto pin_all = [&](Frags& f) {
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) {
#pragma unroll
            for (int a = 0; a < FN; ++a) pin(f.w[a][s2]);
#pragma unroll
            for (int b = 0; b < FM; ++b) pin(f.a[b][s2]);
        }
    };
    auto read_all = [&](Frags& f, auto pb_, auto tp_, auto ws_) {
        [&]<int... R>(std::integer_sequence<int, R...>) {
            (read_one(f, pb_, tp_, ws_, std::integral_constant<int, R>{}), ...);
        }(std::make_integer_sequence<int, NREAD>{});
    };

    // ---- prologue: patch 0 and the weight stages 0..2 are on their way (issued above)
    constexpr int NR = FN + FM;                  // fragment reads of one k-step
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    Frags fr;                                    // ONE fragment set: k-step s of step i+1 is read into the registers of
                                                 // k-step s of step i right after that k-step's MFMAs have been issued
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) {
#pragma unroll
        for (int a = 0; a < FN; ++a) fr.w[a][s2] = bf16x8{};
#pragma unroll
        for (int b = 0; b < FM; ++b) fr.a[b][s2] = bf16x8{};
    }
    read_all(fr, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    stamp(1);

    // one tap step, V = step index mod 8 (compile time): tap V & 3, patch buffer (V >> 2) & 1, weight slot V & 3;
    // more: a step follows (its fragments are read ahead here).  LDS reads return in order: when k-step s starts, the reads
    // issued after its own fragments are the 3 * NR of the following three k-steps (counted lgkmcnt; fewer in the last step).
    auto body = [&](int i, auto v_, const bool more) {
        constexpr int V = decltype(v_)::value;
        constexpr int NV = (V + 1) & 7;                       // the next step's combination
        constexpr int rel = V & 3;
        if constexpr (rel == 0) { cur = nxt; nxt = make_ss((i >> 2) + 1); }     // (before the barrier: scalar work, nothing waits for it)
        if (!no_barrier) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // DMA of this step: the WP weight pieces of stage i+3 and, in the first two steps of a super-step, three pieces each of
        // the next super-step's patch (behind the slice's end: out-of-range offsets, see make_ss)
        constexpr int pj = rel == 0 ? 0 : 3;
        constexpr bool p_issued = !no_dma && rel < 2;
        auto dma_op = [&](int k) {               // op k of this step's list: W pieces first, then the three patch pieces
            if (k < WP) { if (!no_dma) stage_w_piece(rel == 0 ? cur.wsoff : nxt.wsoff, std::integral_constant<int, (V + 3) & 3>{}, k); }
            else if (p_issued) stage_patch(nxt, ((V >> 2) + 1) & 1, pj + (k - WP), pj + (k - WP) + 1);
        };
        constexpr int NOPS = WP + 3;
        auto kstep = [&](auto s2_) {
            constexpr int S2 = decltype(s2_)::value;
            // fragments of k-step S2 have landed
            if (!no_read) {
                if (more) PFA_WAIT_LGKM((KS - 1) * NR); else PFA_WAIT_LGKM((KS - 1 - S2) * NR);
#pragma unroll
                for (int a = 0; a < FN; ++a) pin(fr.w[a][S2]);
#pragma unroll
                for (int b = 0; b < FM; ++b) pin(fr.a[b][S2]);
                __builtin_amdgcn_sched_barrier(0);
            }
            // The k-step's other instructions sit in the shadows of its own MFMAs (a wave issues in order: behind four MFMAs
            // in a row they would all wait for the matrix pipe to drain first).  Order: M(w0,a0) [DMA] M(w1,a0) R(a0) M(w0,a1)
            // R(w0) M(w1,a1) R(w1) R(a1) - a fragment of the NEXT step is read into its registers as soon as the last MFMA
            // that takes the old value has been issued.  knob 14 = 1 (diagnostic instantiation INTER = 0): the old order.
            auto mfma = [&](auto a_, auto b_) {
                constexpr int A = decltype(a_)::value, B = decltype(b_)::value;
                if (!no_mfma) acc[A][B] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.w[A][S2], fr.a[B][S2], acc[A][B], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            auto rd = [&](auto e_) {             // fragment e (0 .. FN-1: weights, FN ..: activations) of this k-step, next tap step
                constexpr int E = decltype(e_)::value;
                if (!no_read && more)
                    read_one(fr, std::integral_constant<int, (NV >> 2) & 1>{}, std::integral_constant<int, NV & 3>{},
                             std::integral_constant<int, NV & 3>{}, std::integral_constant<int, S2 * NR + E>{});
                __builtin_amdgcn_sched_barrier(0);
            };
            auto dma = [&]() {
#pragma unroll
                for (int k = 0; k < NOPS; ++k)
                    if (S2 == (k * KS) / NOPS) dma_op(k);
                __builtin_amdgcn_sched_barrier(0);
            };
            using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
            using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
            if constexpr (FN == 2) {
                mfma(I0{}, I0{}); dma(); mfma(I1{}, I0{}); rd(I2{}); mfma(I0{}, I1{}); rd(I0{}); mfma(I1{}, I1{}); rd(I1{}); rd(I3{});
            } else {
                mfma(I0{}, I0{}); dma(); rd(I1{}); mfma(I0{}, I1{}); rd(I0{}); rd(I2{});
            }
        };
        kstep(std::integral_constant<int, 0>{});
        kstep(std::integral_constant<int, 1>{});
        kstep(std::integral_constant<int, 2>{});
        kstep(std::integral_constant<int, 3>{});
        static_assert(KS == 4, "k-steps of a tap step");
        if (no_dma) wait_vm<0>();
        else if (!no_vmwait) wait_vm<p_issued ? WP + 3 : WP>();
    };
    // Each step body exists ONCE (duplicated tail copies cost registers, and a spilled fragment register would be saved
    // before its in-flight LDS read has landed): `more` is a run-time flag only in the last step of a super-step.
#define PFA_STEP(V, MORE) body(i + V, std::integral_constant<int, V>{}, MORE)
    for (int i = 0; i < nsteps; i += 8) {        // nsteps is a multiple of 4
        const bool second = i + 4 < nsteps;      // this group has its second super-step
        PFA_STEP(0, true); PFA_STEP(1, true); PFA_STEP(2, true); PFA_STEP(3, second);
        if (!second) break;
        PFA_STEP(4, true); PFA_STEP(5, true); PFA_STEP(6, true); PFA_STEP(7, i + 8 < nsteps);
    }
#undef PFA_STEP
    wait_vm<0>();                                // (the pieces issued behind the slice's end land before the buffers are reused)
    stamp(2);

    // ---- epilogue: accumulators -> fp32 [position][channel] image in LDS -> whole NHWC rows out (patch_epilogue.h).
    // 32x32 accumulator: lane holds, for register group g = 0..3, channels 8 g + 4 lh .. +3 of position lr of the tile.
    constexpr int RS = patch_image_stride<BN>();
    __syncthreads();                             // every wave is done with the staging buffers
#pragma unroll
    for (int b = 0; b < FM; ++b) {
        char* row = smem + tpos[b] * RS;
#pragma unroll
        for (int a = 0; a < FN; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
                *(f32x4*)(row + (wn * (BN / 2) + a * 32 + 8 * g + 4 * lh) * 4) = v;
            }
    }
    __syncthreads();
    stamp(3);
    auto pix_of = [&](int r, bool& valid, size_t& opix) {
        const int il = NIMG == 1 ? 0 : r / (TY * TX), rr = r - il * (TY * TX);
        const int qy = qy0 + rr / TX, qx = qx0 + rr % TX, im = img + il;
        valid = qy < p.hq && qx < p.wq && im < p.nb;
        if (MODE == 0) opix = valid ? ((size_t)im * p.ho + 2 * qy + py) * p.wo + 2 * qx + px : 0;
        else opix = valid ? ((size_t)im * p.hq + qy) * p.wq + qx : 0;
    };
    const size_t trow = (((size_t)(img / NIMG) * q.tiles_y + ty) * q.tiles_x + tx) * (MODE == 0 ? 4 : 1) + cls;
    patch_epilogue_rows<BN>(p, n0, tid, split, trow, smem, pix_of, q.stamps ? q.stamps + (size_t)blockIdx.x * 16 : nullptr);
    if (q.stamps) {
        stamp(6);
        __builtin_amdgcn_s_waitcnt(0);  // (stamps only: the stores have been issued AND accepted)
        stamp(7);
        if (tid == 0) q.stamps[(size_t)blockIdx.x * 16 + 9] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int MODE, int BN, int TILE16, int ABL = 0>
int launch_patch2(const Patch2K& k, hipStream_t st) {
    constexpr int LDS = 2 * PATCH_BYTES + W_SLOTS * BN * 128;
    static std::atomic<bool> attr_done{false};   // (launches may come from several host threads)
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)k_conv_patch2<MODE, BN, TILE16, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_done = true;
    }
    hipLaunchKernelGGL((k_conv_patch2<MODE, BN, TILE16, ABL>), dim3(k.nwg), dim3(512), LDS, st, k);
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
    const bool tile_img = k.hq == 8 && k.wq == 8 && !(pfa_knob(1) & 4096);   // the 8 x 8 maps: four images per tile (knob 1 bit 4096: off, A/B)
    if (!tile_img && (k.hq < 8 || k.wq < 16)) return false;
    // buffer-form LDS-DMA: every in-range offset stays below 2 GiB
    if (((long long)k.nb * k.hs * k.ws + k.ws + 2) * (k.c1 > k.c2 ? k.c1 : k.c2) * 2 >= (1LL << 31) || (long long)k.n * 16 * k.ctot * 2 >= (1LL << 31))
        return false;
    Patch2Plan pl{};
    pl.tile16 = tile_img ? 2 : k.wq < 32 ? 1 : 0;
    const int tyy = pl.tile16 == 1 ? 16 : 8, txx = pl.tile16 == 1 ? 16 : pl.tile16 == 2 ? 8 : 32;
    pl.tiles_y = (k.hq + tyy - 1) / tyy; pl.tiles_x = (k.wq + txx - 1) / txx;
    const int classes = k.mode == PFA_TCONV_S2 ? 4 : 1;
    const int groups = pl.tile16 == 2 ? (k.nb + 3) / 4 : k.nb;           // image groups of the work list
    const long long tiles = (long long)groups * pl.tiles_y * pl.tiles_x * classes;
    const int nss = (k.ctot / 64) * (k.mode == PFA_CONV_S2 ? 4 : 1);
    // (in-situ sweep of the threshold: 192 -> 3.41 ms per step, 128 -> 3.37, 256 -> 3.47, 320 -> 3.65: a launch that leaves CUs
    // idle is not lost inside the three-stream step, narrower tiles are)
    // (knob 15, set by the engines around passes that run alone on the GPU - the generator phase's discriminator pass: no other
    // stream fills the CUs a 128-workgroup launch leaves idle there, so such a launch narrows its channel tile)
    const int want = pfa_knob(6) > 0 ? pfa_knob(6) : (pfa_knob(15) ? 192 : 128);
    pl.splitk = 1;
    if (k.n % 128 == 0 && tiles * (k.n / 128) >= want) pl.bn = 128;
    else if (tile_img && k.n % 128 == 0 && k.splitk > 1) {
        // the 8 x 8 maps have 4 .. 16 position tiles at batch 16: 128-channel tiles (weight bytes per FLOP) and K slices
        pl.bn = 128;
        int sk = k.splitk < nss ? k.splitk : nss;
        while (sk > 2 && tiles * (k.n / 128) * (sk / 2) >= want) sk /= 2;
        while (nss % sk) --sk;                                             // (whole super-steps per slice, none empty)
        pl.splitk = sk;
    }
    else if (tiles * (k.n / 64) >= want || k.splitk <= 1) pl.bn = 64;
    else {
        pl.bn = 64;
        int sk = k.splitk < nss ? k.splitk : nss;
        while (sk > 2 && tiles * (k.n / 64) * (sk / 2) >= want) sk /= 2;     // no more slices than the chip needs
        pl.splitk = sk;
    }
    if (pfa_knob(3) != 2 && pfa_knob(3) != 5 && tiles * (k.n / pl.bn) * pl.splitk < 96) return false;      // too small to be worth a patch launch
    pl.stat_tiles = (int)tiles;
    // a transposed conv on 64-channel tiles is DMA-bound with one class per workgroup: all four classes over one patch
    pl.tconv4 = (k.mode == PFA_TCONV_S2 && pl.bn == 64 && !pl.tile16 && pl.splitk == 1 && pfa_tconv4_ok(k, dtype, pl.tiles_y, pl.tiles_x)) ? 1 : 0;
    if (plan) *plan = pl;
    return true;
}

int pfa_conv_patch2(const GatherK& k, const Patch2Plan& plan, hipStream_t st, const char** kernel_name) {
    if (plan.tconv4) return pfa_tconv4(k, plan.tiles_y, plan.tiles_x, st, kernel_name);
    Patch2K q;
    q.g = k;
    q.g.splitk = plan.splitk;
    q.tiles_y = plan.tiles_y; q.tiles_x = plan.tiles_x;
    q.ntiles = k.n / plan.bn;
    const int classes = k.mode == PFA_TCONV_S2 ? 4 : 1;
    q.nwg = (plan.tile16 == 2 ? (k.nb + 3) / 4 : k.nb) * plan.tiles_y * plan.tiles_x * classes * q.ntiles * plan.splitk;
    q.nss = (k.ctot / 64) * (k.mode == PFA_CONV_S2 ? 4 : 1);
    q.ssper = (q.nss + plan.splitk - 1) / plan.splitk;
    PFA_REQUIRE((q.nss + q.ssper - 1) / q.ssper == plan.splitk, "pfa_conv_patch2: empty K slice");
    q.c1chunks = k.c1 / 64;
    q.stamps = nullptr;
    if (g_pfa_dbgbuf && g_pfa_dbgbuf_bytes >= (size_t)q.nwg * 128) q.stamps = (unsigned long long*)g_pfa_dbgbuf;
    if (plan.tile16 == 2) {
        static const char* names2[4] = {"k_conv_patch2<tconv,128,4x8x8>", "k_conv_patch2<tconv,64,4x8x8>", "k_conv_patch2<conv_s2,128,4x8x8>",
                                        "k_conv_patch2<conv_s2,64,4x8x8>"};
        const int sel2 = (k.mode == PFA_CONV_S2 ? 2 : 0) + (plan.bn == 64 ? 1 : 0);
        if (kernel_name) *kernel_name = names2[sel2];
        switch (sel2) {
            case 0: return launch_patch2<0, 128, 2>(q, st);
            case 1: return launch_patch2<0, 64, 2>(q, st);
            case 2: return launch_patch2<1, 128, 2>(q, st);
            default: return launch_patch2<1, 64, 2>(q, st);
        }
    }
    const int sel = (k.mode == PFA_CONV_S2 ? 4 : 0) + (plan.bn == 64 ? 2 : 0) + (plan.tile16 ? 1 : 0);
    static const char* names[8] = {"k_conv_patch2<tconv,128,8x32>", "k_conv_patch2<tconv,128,16x16>", "k_conv_patch2<tconv,64,8x32>",
                                   "k_conv_patch2<tconv,64,16x16>", "k_conv_patch2<conv_s2,128,8x32>", "k_conv_patch2<conv_s2,128,16x16>",
                                   "k_conv_patch2<conv_s2,64,8x32>", "k_conv_patch2<conv_s2,64,16x16>"};
    if (kernel_name) *kernel_name = names[sel];
    if (sel == 4 && pfa_knob(7) == 18) return launch_patch2<1, 128, 0, 8>(q, st);       // no counted vmcnt wait in the loop (races: timing only)
    if (sel == 4 && pfa_knob(7) == 19) return launch_patch2<1, 128, 0, 24>(q, st);      // nor the step's barrier
    if (sel == 4 && pfa_knob(7) >= 11 && pfa_knob(7) <= 17) {      // developer ablations of the stride-2 128-channel kernel (results are wrong)
        switch (pfa_knob(7) - 10) {
            case 1: return launch_patch2<1, 128, 0, 1>(q, st);     // no MFMA
            case 2: return launch_patch2<1, 128, 0, 2>(q, st);     // no fragment reads
            case 3: return launch_patch2<1, 128, 0, 3>(q, st);
            case 4: return launch_patch2<1, 128, 0, 4>(q, st);     // no DMA
            case 5: return launch_patch2<1, 128, 0, 5>(q, st);
            case 6: return launch_patch2<1, 128, 0, 6>(q, st);
            default: return launch_patch2<1, 128, 0, 7>(q, st);    // barriers, waits and address arithmetic only
        }
    }
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
