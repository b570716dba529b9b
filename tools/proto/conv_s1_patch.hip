// PROTOTYPE (not part of libpfa_hip.so, not on any product path): Conv2d(k4, s1, p1) forward with the input PATCH kept in LDS.
//
// Round-2 groundwork for DESIGN.md section 6 "Next" item 1.  The product kernel (csrc/gather_gemm.hip) stages, for every
// (tap, channel chunk) K-step, one 128-byte piece per output pixel: 16 taps x the same input pixels.  Here a workgroup owns
// an 8 x 32 patch of output pixels of one image, stages the (8+3) x (32+3) input pixels of a 64-channel chunk ONCE
// (double-buffered, its 7 DMA pieces spread over the first 7 tap steps of the previous chunk) and walks the 16 taps
// inside LDS; only the weight tile (128 rows x 128 B per step) still streams through a 3-stage ring with a counted
// vmcnt.  DMA instructions per K-step and wave: 2 (+7/16) instead of 6.
//
// Built by tools/proto/Makefile into tools/proto/libpfa_proto.so; tools/proto/bench_conv_s1_patch.py checks it against
// pfa_gather_gemm on the discriminator's c4 shape and times both.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define WAIT_LGKM(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory")

static __device__ __attribute__((aligned(256))) uint32_t g_zero_page[64];

__device__ __forceinline__ bf16x8 lds_read16(unsigned addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
template <int OFF> __device__ __forceinline__ bf16x8 lds_read16o(unsigned addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <typename F> __device__ __forceinline__ void pin(F& v) { asm volatile("" : "+v"(v)); }

struct ProtoK {
    const char* src;     // [nb][hin][win][cin] bf16
    const char* w;       // [n][16][cin] bf16 (the product's "same" weight layout)
    char* out;           // [nb][hout][wout][n] bf16
    int nb, hin, win, cin, n, hout, wout, tiles_y, ntiles;
    int dbg;             // ablation bits (results are wrong when set): 1 no MFMA, 2 no LDS fragment reads after step 0, 4 no DMA after the prologue
};

constexpr int TY = 8, TX = 32;                    // output patch of a workgroup: 256 pixels
constexpr int PH = TY + 3, PW = TX + 3;           // input patch with halo
constexpr int PSLOTS = PH * PW;                   // 385 pixels x 128 B
constexpr int PSTRIDE = 144;                      // 8-wave kernel: 128 B + one unused 16-byte piece per input pixel - 16
                                                  // consecutive pixels then fall on 16 distinct bank groups for every chunk
                                                  // WITHOUT an XOR swizzle, so a tap shift is a plain address offset
constexpr int PPASS = 7;                          // 7 x 512 lanes x 16 B >= 385 x 144 B
constexpr int PATCH_BYTES = PPASS * 512 * 16;     // 57,344 (padded: every lane of every pass writes)
constexpr int W_BYTES = 128 * 128;                // 128 weight rows x 128 B
constexpr int LDS_BYTES = 2 * PATCH_BYTES + 3 * W_BYTES;      // 163,840: the whole LDS of a CU

// STAG: the two waves of every SIMD (waves w and w + 4) run in opposite read / MFMA phases (the staggered ring of
// csrc/gather_gemm.hip, knob 7 = 6): with 2-3 DMA pieces per step instead of 6 the read phase now fits under the other
// group's 32 MFMAs.
template <bool STAG>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void k_conv_s1_patch(const ProtoK p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;                 // 4 x 2 waves, 64 x 64 per wave
    const int lrow = lane & 15, kk = lane >> 4;
    unsigned wid = blockIdx.x;
    const int nt = wid % p.ntiles; wid /= p.ntiles;
    const int ty = wid % p.tiles_y, img = wid / p.tiles_y;
    const int oy0 = ty * TY, n0 = nt * 128;
    const char* zero = (const char*)g_zero_page;

    // ---- patch staging: lane's 7 pieces (slot = input pixel of the patch, chunk slot cp holds source chunk cp ^ swz)
    unsigned poff[PPASS];
    unsigned pok = 0;
#pragma unroll
    for (int j = 0; j < PPASS; ++j) {
        const int q = j * 512 + tid, slot = q / 9, g = q - slot * 9;          // 9 pieces per pixel, the ninth unused
        const int py = slot / PW, px = slot - py * PW;
        const int iy = oy0 - 1 + py, ix = px - 1;
        const bool ok = g < 8 && slot < PSLOTS && (unsigned)iy < (unsigned)p.hin && (unsigned)ix < (unsigned)p.win;
        poff[j] = ok ? (unsigned)((((img * p.hin + iy) * p.win + ix) * p.cin + g * 8) * 2) : 0u;
        pok |= ok ? (1u << j) : 0u;
    }
    auto stage_patch_piece = [&](int chunk, int j) {
        const char* gp = ((pok >> j) & 1u) ? p.src + poff[j] + chunk * 128 : zero;
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gp), LDS_PTR(smem + (chunk & 1) * PATCH_BYTES + (j * 512 + wave * 64) * 16), 16, 0, 0);
    };
    // ---- weight staging: 2 pieces per step (rows r0 + 64 i), chunk slot cs holds source chunk cs ^ swz(row)
    const int cs = tid & 7, r0 = tid >> 3;
    const int wg = cs ^ ((r0 >> 1) & 7);
    unsigned woff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) woff[i] = (unsigned)(((n0 + r0 + 64 * i) * 16 * p.cin + wg * 8) * 2);
    auto stage_w = [&](int step, int buf) {                 // step = chunk * 16 + tap
        const int chunk = step >> 4, tap = step & 15;
        char* wb = smem + 2 * PATCH_BYTES + buf * W_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const char* gp = p.w + woff[i] + (tap * p.cin + chunk * 64) * 2;
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gp), LDS_PTR(wb + (i * 512 + wave * 64) * 16), 16, 0, 0);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addressing: activation fragment b of this wave = 16 consecutive output pixels of patch row wm*2 + b/2
    const unsigned lds0 = (unsigned)(size_t)LDS_PTR(smem);
    unsigned abyte[4];                   // byte offset of this lane's piece of fragment b inside a patch buffer, tap (0,0), half 0
#pragma unroll
    for (int b = 0; b < 4; ++b) abyte[b] = ((wm * 2 + (b >> 1)) * PW + (b & 1) * 16 + lrow) * PSTRIDE + kk * 16;
    unsigned wfb[2];
    {
        const unsigned sw = (lrow >> 1) & 7;
#pragma unroll
        for (int h = 0; h < 2; ++h) wfb[h] = lds0 + 2 * PATCH_BYTES + (wn * 64 + lrow) * 128 + (((kk + 4 * h) ^ sw) << 4);
    }

    const int nchunks = p.cin / 64, nsteps = nchunks * 16;
    // prologue: patch 0, weight stages 0 and 1
#pragma unroll
    for (int j = 0; j < PPASS; ++j) stage_patch_piece(0, j);
    stage_w(0, 0);
    if (nsteps > 1) stage_w(1, 1);
    auto read_frags = [&](int i, bf16x8 (*wf)[4], bf16x8 (*af)[4]) {
        // one VALU add per address register: the tap shift and the ring stage are wave-uniform offsets, the 16-row
        // fragment pitch and the K half are immediates
        const int chunk = i >> 4, tap = i & 15;
        const unsigned tb = lds0 + (chunk & 1) * PATCH_BYTES + ((tap >> 2) * PW + (tap & 3)) * PSTRIDE;
        const unsigned rb = (i % 3) * W_BYTES;
        const unsigned w0 = wfb[0] + rb, w1 = wfb[1] + rb;
        wf[0][0] = lds_read16o<0>(w0); wf[0][1] = lds_read16o<2048>(w0); wf[0][2] = lds_read16o<4096>(w0); wf[0][3] = lds_read16o<6144>(w0);
#pragma unroll
        for (int b = 0; b < 4; ++b) af[0][b] = lds_read16o<0>(abyte[b] + tb);
        wf[1][0] = lds_read16o<0>(w1); wf[1][1] = lds_read16o<2048>(w1); wf[1][2] = lds_read16o<4096>(w1); wf[1][3] = lds_read16o<6144>(w1);
#pragma unroll
        for (int b = 0; b < 4; ++b) af[1][b] = lds_read16o<64>(abyte[b] + tb);
    };
    auto mfma_half = [&](bf16x8 (*wf)[4], bf16x8 (*af)[4], int h) {
#pragma unroll
        for (int a = 0; a < 4; ++a) pin(wf[h][a]);
#pragma unroll
        for (int b = 0; b < 4; ++b) pin(af[h][b]);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[h][a], af[h][b], acc[a][b], 0, 0, 0);
    };
    bf16x8 wf[2][4], af[2][4];
    if constexpr (STAG) {
        // read phase of step i: fragments of stage i, DMA of [patch piece] + weight stage i+2, counted wait for stage
        // i+1, every read retired -> barrier -> MFMA phase -> barrier; the late group runs one barrier behind
        const bool late = wave >= 4;
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");          // patch 0 and weight stage 0 have landed
        __builtin_amdgcn_s_barrier();
        if (late) __builtin_amdgcn_s_barrier();
        for (int i = 0; i < nsteps; ++i) {
            const int chunk = i >> 4, tap = i & 15;
            __builtin_amdgcn_sched_barrier(0);
            read_frags(i, wf, af);
            const bool pp = tap < PPASS && chunk + 1 < nchunks;
            if (pp) stage_patch_piece(chunk + 1, tap);
            if (i + 2 < nsteps) {
                stage_w(i + 2, (i + 2) % 3);
                if (pp) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            WAIT_LGKM(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
            mfma_half(wf, af, 0);
            mfma_half(wf, af, 1);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
        if (!late) __builtin_amdgcn_s_barrier();
    } else {
    bool prev_patch_piece = false;
    for (int i = 0; i < nsteps; ++i) {
        const int chunk = i >> 4, tap = i & 15;
        // everything older than the weight pieces of step i+1 (and the patch piece issued in step i-1, if any) has landed
        if (i + 1 >= nsteps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (prev_patch_piece) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        prev_patch_piece = tap < PPASS && chunk + 1 < nchunks && !(p.dbg & 4);
        if (prev_patch_piece) stage_patch_piece(chunk + 1, tap);
        if (i + 2 < nsteps && !(p.dbg & 4)) stage_w(i + 2, (i + 2) % 3);
        __builtin_amdgcn_sched_barrier(0);
        if (!(p.dbg & 2) || i == 0) read_frags(i, wf, af);
        WAIT_LGKM(8);
        if (!(p.dbg & 1)) mfma_half(wf, af, 0);
        __builtin_amdgcn_sched_barrier(0);
        WAIT_LGKM(0);
        if (!(p.dbg & 1)) mfma_half(wf, af, 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    }

    // ---- epilogue: lane holds D[n = .. + 4 kk + r][pixel = .. + lrow]
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int oy = oy0 + wm * 2 + (b >> 1), ox = (b & 1) * 16 + lrow;
        if (oy >= p.hout || ox >= p.wout) continue;
        char* orow = p.out + ((size_t)((img * p.hout + oy) * p.wout + ox) * p.n) * 2;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int col = n0 + wn * 64 + a * 16 + 4 * kk;
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16_t)acc[a][b][r];
            *(bf16x4*)(orow + col * 2) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Variant "big wave tile": the same 256 x 128 workgroup tile on FOUR waves (2 x 2), each 128 pixels x 64 channels: 128
// accumulator registers, 12 fragment reads per 32 MFMAs instead of 8 per 16 (-25 % LDS read traffic).  One wave per SIMD
// (up to 512 registers): both K halves of a step are read up front, so the second half's reads return under the first
// half's 32 MFMAs.
constexpr int PPASS4 = 13;                        // 13 x 256 lanes x 16 B >= 385 x 128 B
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1))) void k_conv_s1_patch_w4(const ProtoK p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;                 // 2 x 2 waves, 128 x 64 per wave
    const int lrow = lane & 15, kk = lane >> 4;
    unsigned wid = blockIdx.x;
    const int nt = wid % p.ntiles; wid /= p.ntiles;
    const int ty = wid % p.tiles_y, img = wid / p.tiles_y;
    const int oy0 = ty * TY, n0 = nt * 128;
    const char* zero = (const char*)g_zero_page;

    unsigned poff[PPASS4];
    unsigned pok = 0;
#pragma unroll
    for (int j = 0; j < PPASS4; ++j) {
        const int q = j * 256 + tid, slot = q >> 3, cp = q & 7;
        const int g = cp ^ ((slot >> 1) & 7);
        const int py = slot / PW, px = slot - py * PW;
        const int iy = oy0 - 1 + py, ix = px - 1;
        const bool ok = slot < PSLOTS && (unsigned)iy < (unsigned)p.hin && (unsigned)ix < (unsigned)p.win;
        poff[j] = ok ? (unsigned)((((img * p.hin + iy) * p.win + ix) * p.cin + g * 8) * 2) : 0u;
        pok |= ok ? (1u << j) : 0u;
    }
    auto stage_patch_piece = [&](int chunk, int j) {
        const char* gp = ((pok >> j) & 1u) ? p.src + poff[j] + chunk * 128 : zero;
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gp), LDS_PTR(smem + (chunk & 1) * PATCH_BYTES + (j * 256 + wave * 64) * 16), 16, 0, 0);
    };
    const int cs = tid & 7, r0 = tid >> 3;                   // 32 weight rows per pass, 4 passes
    const int wg = cs ^ ((r0 >> 1) & 7);
    unsigned woff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) woff[i] = (unsigned)(((n0 + r0 + 32 * i) * 16 * p.cin + wg * 8) * 2);
    auto stage_w = [&](int step, int buf) {
        const int chunk = step >> 4, tap = step & 15;
        char* wb = smem + 2 * PATCH_BYTES + buf * W_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const char* gp = p.w + woff[i] + (tap * p.cin + chunk * 64) * 2;
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gp), LDS_PTR(wb + (i * 256 + wave * 64) * 16), 16, 0, 0);
        }
    };

    f32x4 acc[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned lds0 = (unsigned)(size_t)LDS_PTR(smem);
    int aslot[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) aslot[b] = (wm * 4 + (b >> 1)) * PW + (b & 1) * 16 + lrow;
    unsigned wfb[2];
    {
        const unsigned sw = (lrow >> 1) & 7;
#pragma unroll
        for (int h = 0; h < 2; ++h) wfb[h] = lds0 + 2 * PATCH_BYTES + (wn * 64 + lrow) * 128 + (((kk + 4 * h) ^ sw) << 4);
    }
    const int nchunks = p.cin / 64, nsteps = nchunks * 16;
#pragma unroll
    for (int j = 0; j < PPASS4; ++j) stage_patch_piece(0, j);
    stage_w(0, 0);
    if (nsteps > 1) stage_w(1, 1);
    bf16x8 wf[2][4], af[2][8];
    bool prev_patch_piece = false;
    for (int i = 0; i < nsteps; ++i) {
        const int chunk = i >> 4, tap = i & 15;
        if (i + 1 >= nsteps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (prev_patch_piece) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        prev_patch_piece = tap < PPASS4 && chunk + 1 < nchunks && !(p.dbg & 4);
        if (prev_patch_piece) stage_patch_piece(chunk + 1, tap);
        if (i + 2 < nsteps && !(p.dbg & 4)) stage_w(i + 2, (i + 2) % 3);
        __builtin_amdgcn_sched_barrier(0);
        if (!(p.dbg & 2) || i == 0) {
            const unsigned pbase = lds0 + (chunk & 1) * PATCH_BYTES;
            const int tapslot = (tap >> 2) * PW + (tap & 3);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int a = 0; a < 4; ++a) wf[h][a] = lds_read16(wfb[h] + (i % 3) * W_BYTES + a * 2048);
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const int slot = aslot[b] + tapslot;
                    af[h][b] = lds_read16(pbase + slot * 128 + (((kk + 4 * h) ^ ((slot >> 1) & 7)) << 4));
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 0) WAIT_LGKM(12); else WAIT_LGKM(0);
#pragma unroll
            for (int a = 0; a < 4; ++a) pin(wf[h][a]);
#pragma unroll
            for (int b = 0; b < 8; ++b) pin(af[h][b]);
            if (!(p.dbg & 1)) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 8; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[h][a], af[h][b], acc[a][b], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const int oy = oy0 + wm * 4 + (b >> 1), ox = (b & 1) * 16 + lrow;
        if (oy >= p.hout || ox >= p.wout) continue;
        char* orow = p.out + ((size_t)((img * p.hout + oy) * p.wout + ox) * p.n) * 2;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int col = n0 + wn * 64 + a * 16 + 4 * kk;
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16_t)acc[a][b][r];
            *(bf16x4*)(orow + col * 2) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Transposed conv (k4, s2, p1) = the product's PFA_TCONV_S2 (also the input gradient of a stride-2 conv): one output parity
// class (py, px) per workgroup, 8 x 32 source-grid positions, (8+1) x (32+1) input patch per 64-channel chunk, FOUR tap
// steps per chunk.  The 6 patch pieces of the next chunk go out in the first two steps of the current one (3 + 3), so that
// the wait of the third step has retired them before the chunk changes.  Staggered phases, 144-byte patch pixels.
struct ProtoT {
    const char* src;     // [nb][h][w][cin] bf16
    const char* w;       // [n][16][cin] bf16 (the product's "tr" layout: row = output channel)
    char* out;           // [nb][2h][2w][n] bf16
    int nb, h, w_, cin, n, tiles_y, tiles_x, ntiles;
};
constexpr int TPH = TY + 1, TPW = TX + 1, TPSLOTS = TPH * TPW;        // 9 x 33 = 297 pixels
constexpr int TPPASS = 6;                                               // 6 x 512 x 16 B >= 297 x 144 B
constexpr int TPATCH_BYTES = TPPASS * 512 * 16;                         // 49,152
constexpr int TLDS_BYTES = 2 * TPATCH_BYTES + 3 * W_BYTES;              // 147,456

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void k_tconv_patch(const ProtoT p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int lrow = lane & 15, kk = lane >> 4;
    unsigned wid = blockIdx.x;
    const int nt = wid % p.ntiles; wid /= p.ntiles;
    const int cls = wid & 3; wid >>= 2;
    const int tx = wid % p.tiles_x; wid /= p.tiles_x;
    const int ty = wid % p.tiles_y, img = wid / p.tiles_y;
    const int py = cls >> 1, px = cls & 1;
    const int qy0 = ty * TY, qx0 = tx * TX, n0 = nt * 128;
    const char* zero = (const char*)g_zero_page;

    unsigned poff[TPPASS];
    unsigned pok = 0;
#pragma unroll
    for (int j = 0; j < TPPASS; ++j) {
        const int q = j * 512 + tid, slot = q / 9, g = q - slot * 9;
        const int sy = slot / TPW, sx = slot - sy * TPW;
        const int iy = qy0 + py - 1 + sy, ix = qx0 + px - 1 + sx;
        const bool ok = g < 8 && slot < TPSLOTS && (unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.w_;
        poff[j] = ok ? (unsigned)((((img * p.h + iy) * p.w_ + ix) * p.cin + g * 8) * 2) : 0u;
        pok |= ok ? (1u << j) : 0u;
    }
    auto stage_patch_piece = [&](int chunk, int j) {
        const char* gp = ((pok >> j) & 1u) ? p.src + poff[j] + chunk * 128 : zero;
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gp), LDS_PTR(smem + (chunk & 1) * TPATCH_BYTES + (j * 512 + wave * 64) * 16), 16, 0, 0);
    };
    const int cs = tid & 7, r0 = tid >> 3;
    const int wg = cs ^ ((r0 >> 1) & 7);
    unsigned woff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) woff[i] = (unsigned)(((n0 + r0 + 64 * i) * 16 * p.cin + wg * 8) * 2);
    auto stage_w = [&](int step, int buf) {                 // step = chunk * 4 + tap
        const int chunk = step >> 2, t = step & 3;
        const int widx = (1 - py + 2 * (t >> 1)) * 4 + (1 - px + 2 * (t & 1));
        char* wb = smem + 2 * TPATCH_BYTES + buf * W_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const char* gp = p.w + woff[i] + (widx * p.cin + chunk * 64) * 2;
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gp), LDS_PTR(wb + (i * 512 + wave * 64) * 16), 16, 0, 0);
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned lds0 = (unsigned)(size_t)LDS_PTR(smem);
    unsigned abyte[4];       // position q = (row wm*2 + b/2, col (b&1)*16 + lrow); tap (ty,tx) reads patch (row + 1 - ty, col + 1 - tx)
#pragma unroll
    for (int b = 0; b < 4; ++b) abyte[b] = ((wm * 2 + (b >> 1)) * TPW + (b & 1) * 16 + lrow) * PSTRIDE + kk * 16;
    unsigned wfb[2];
    {
        const unsigned sw = (lrow >> 1) & 7;
#pragma unroll
        for (int h = 0; h < 2; ++h) wfb[h] = lds0 + 2 * TPATCH_BYTES + (wn * 64 + lrow) * 128 + (((kk + 4 * h) ^ sw) << 4);
    }
    const int nchunks = p.cin / 64, nsteps = nchunks * 4;
#pragma unroll
    for (int j = 0; j < TPPASS; ++j) stage_patch_piece(0, j);
    stage_w(0, 0);
    stage_w(1, 1);
    bf16x8 wf[2][4], af[2][4];
    const bool late = wave >= 4;
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (late) __builtin_amdgcn_s_barrier();
    for (int i = 0; i < nsteps; ++i) {
        const int chunk = i >> 2, t = i & 3;
        __builtin_amdgcn_sched_barrier(0);
        {
            const unsigned tb = lds0 + (chunk & 1) * TPATCH_BYTES + ((1 - (t >> 1)) * TPW + (1 - (t & 1))) * PSTRIDE;
            const unsigned rb = (i % 3) * W_BYTES;
            const unsigned w0 = wfb[0] + rb, w1 = wfb[1] + rb;
            wf[0][0] = lds_read16o<0>(w0); wf[0][1] = lds_read16o<2048>(w0); wf[0][2] = lds_read16o<4096>(w0); wf[0][3] = lds_read16o<6144>(w0);
#pragma unroll
            for (int b = 0; b < 4; ++b) af[0][b] = lds_read16o<0>(abyte[b] + tb);
            wf[1][0] = lds_read16o<0>(w1); wf[1][1] = lds_read16o<2048>(w1); wf[1][2] = lds_read16o<4096>(w1); wf[1][3] = lds_read16o<6144>(w1);
#pragma unroll
            for (int b = 0; b < 4; ++b) af[1][b] = lds_read16o<64>(abyte[b] + tb);
        }
        const bool pp = t < 2 && chunk + 1 < nchunks;        // 3 patch pieces of the next chunk in each of the first two steps
        if (pp) {
#pragma unroll
            for (int j = 0; j < 3; ++j) stage_patch_piece(chunk + 1, t * 3 + j);
        }
        if (i + 2 < nsteps) {
            stage_w(i + 2, (i + 2) % 3);
            if (pp) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        WAIT_LGKM(0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int a = 0; a < 4; ++a) pin(wf[h][a]);
#pragma unroll
            for (int b = 0; b < 4; ++b) pin(af[h][b]);
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[h][a], af[h][b], acc[a][b], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    }
    if (!late) __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int qy = qy0 + wm * 2 + (b >> 1), qx = qx0 + (b & 1) * 16 + lrow;
        if (qy >= p.h || qx >= p.w_) continue;
        char* orow = p.out + ((size_t)((img * 2 * p.h + 2 * qy + py) * 2 * p.w_ + 2 * qx + px) * p.n) * 2;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int col = n0 + wn * 64 + a * 16 + 4 * kk;
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16_t)acc[a][b][r];
            *(bf16x4*)(orow + col * 2) = o;
        }
    }
}

extern "C" int pfa_proto_tconv_patch(const void* src, const void* w, void* out, int nb, int h, int wd, int cin, int n, void* stream) {
    if (!src || !w || !out || cin % 64 || n % 128 || cin < 128) return -1;
    ProtoT k;
    k.src = (const char*)src; k.w = (const char*)w; k.out = (char*)out;
    k.nb = nb; k.h = h; k.w_ = wd; k.cin = cin; k.n = n;
    k.tiles_y = (h + TY - 1) / TY; k.tiles_x = (wd + TX - 1) / TX; k.ntiles = n / 128;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)k_tconv_patch, hipFuncAttributeMaxDynamicSharedMemorySize, TLDS_BYTES);
        attr_done = true;
    }
    hipLaunchKernelGGL(k_tconv_patch, dim3(nb * k.tiles_y * k.tiles_x * 4 * k.ntiles), dim3(512), TLDS_BYTES, (hipStream_t)stream, k);
    return (int)hipGetLastError();
}

// src [nb][hin][win][cin] bf16, w [n][16][cin] bf16, out [nb][hin-1][win-1][n] bf16; cin % 64 == 0, n % 128 == 0, win - 1 <= 32
extern "C" int pfa_proto_conv_s1_patch(const void* src, const void* w, void* out, int nb, int hin, int win, int cin, int n,
                                       int staggered, int dbg, void* stream) {
    if (!src || !w || !out || cin % 64 || n % 128 || win - 1 > TX || hin < 2 || win < 2) return -1;
    ProtoK k;
    k.src = (const char*)src; k.w = (const char*)w; k.out = (char*)out;
    k.nb = nb; k.hin = hin; k.win = win; k.cin = cin; k.n = n; k.hout = hin - 1; k.wout = win - 1;
    k.tiles_y = (k.hout + TY - 1) / TY; k.ntiles = n / 128; k.dbg = dbg;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)k_conv_s1_patch<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)k_conv_s1_patch<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_done = true;
    }
    const dim3 grid(nb * k.tiles_y * k.ntiles);
    if (staggered == 2) {            // the 4-wave variant with 128 x 64 per-wave tiles
        static bool attr4 = false;
        if (!attr4) {
            (void)hipFuncSetAttribute((const void*)k_conv_s1_patch_w4, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
            attr4 = true;
        }
        hipLaunchKernelGGL(k_conv_s1_patch_w4, grid, dim3(256), LDS_BYTES, (hipStream_t)stream, k);
        return (int)hipGetLastError();
    }
    if (staggered) hipLaunchKernelGGL(k_conv_s1_patch<true>, grid, dim3(512), LDS_BYTES, (hipStream_t)stream, k);
    else hipLaunchKernelGGL(k_conv_s1_patch<false>, grid, dim3(512), LDS_BYTES, (hipStream_t)stream, k);
    return (int)hipGetLastError();
}
