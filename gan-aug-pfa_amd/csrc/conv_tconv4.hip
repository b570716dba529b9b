// Transposed convolution (PFA_TCONV_S2) onto 64 output channels with ALL FOUR output parity classes in one workgroup.
//
// conv_patch2.hip gives every parity class of a position tile its own workgroup: each stages its own (shifted) input patch,
// and with a 64-channel tile (FN = 1) a tap step is 512 cycles of MFMA against 20 KiB of LDS-DMA - at the ~30 B/clk a CU takes
// in from L2 the loop is DMA-bound (d2 forward, the input gradients of e2 / c2: 5 launches, 240 us per step at 17 % of the
// MFMA peak).  The 16 (class, tap) pairs of a transposed convolution read only the 3 x 3 neighbourhood of an input position,
// and each of the 16 kernel taps belongs to exactly one class:
//     out(2qy+py, 2qx+px) = sum_{a,b in {0,1}} in(qy+py-1+a, qx+px-1+b) * w[(3-py-2a)*4 + (3-px-2b)]
// so ONE (8+2) x (32+2) patch per 64-channel chunk serves all four classes: a chunk is 16 tap steps (class-major) over the
// same patch - patch DMA / 4, 11 KiB per tap step - with four accumulator sets (128 registers) in the wave.  A workgroup
// writes a 16 x 64 block of output pixels (four epilogue passes through the whole-row epilogue of patch_epilogue.h).
//
// K loop as in conv_patch2.hip: one barrier per tap step, four k-steps of MFMA 32x32x16, ONE fragment register set re-read in
// place a step ahead (counted lgkmcnt), weight tiles through a 4-slot ring three steps ahead, the next chunk's patch in the
// first six steps of a chunk, counted vmcnt.  The 16 (class, tap) bodies are compiled once each: the tap offset, the weight
// slot and the accumulator set are compile-time; the patch buffer of a chunk is a register pair selected per chunk.
#include "patch_epilogue.h"

#include <type_traits>
#include <utility>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int T4_TY = 8, T4_TX = 32, T4_PH = T4_TY + 2, T4_PW = T4_TX + 2, T4_SLOTS = T4_PH * T4_PW;     // 340 pixels
constexpr int T4_PSTRIDE = 144, T4_PPASS = 6, T4_PATCH = T4_PPASS * 512 * 16;                           // 49,152 bytes
constexpr int T4_BN = 64, T4_KS = 4, T4_WBYTES = T4_BN * 128, T4_WSLOTS = 4;
constexpr int T4_WBASE = 2 * T4_PATCH;
constexpr int T4_LDS_LOOP = T4_WBASE + T4_WSLOTS * T4_WBYTES;
static_assert(T4_SLOTS * 9 <= T4_PPASS * 512, "patch does not fit its buffer");

struct Tconv4K {
    GatherK g;
    int tiles_y, tiles_x, ntiles, nwg;
    int nchunks, c1chunks;
};

template <int N> __device__ __forceinline__ void t4_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int OFF> __device__ __forceinline__ void t4_rd(unsigned addr, bf16x8& v) {      // in place: see conv_patch2.hip lds_rd
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(v) : "v"(addr), "n"(OFF));
}

// patch byte offset of (class, tap) = V: class (py, px) = (V >> 3, (V >> 2) & 1), tap (a, b) = ((V >> 1) & 1, V & 1)
constexpr int t4_tap_off(int v) {
    return (((v >> 3) + ((v >> 1) & 1)) * T4_PW + (((v >> 2) & 1) + (v & 1))) * T4_PSTRIDE;
}
// kernel tap of (class, tap)
constexpr int t4_widx(int v) {
    const int py = v >> 3, px = (v >> 2) & 1, a = (v >> 1) & 1, b = v & 1;
    return (3 - py - 2 * a) * 4 + (3 - px - 2 * b);
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void k_tconv4(const Tconv4K q) {
    constexpr int FM = 2, KS = T4_KS, NR = 1 + FM;           // fragment reads of a k-step: one weight, two activation
    const GatherK& p = q.g;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int lr = lane & 31, lh = lane >> 5;
    unsigned wid = xcd_remap(blockIdx.x, q.nwg);
    const int nt = wid % q.ntiles; wid /= q.ntiles;
    const int tx = wid % q.tiles_x; wid /= q.tiles_x;
    const int ty = wid % q.tiles_y, img = wid / q.tiles_y;
    const int qy0 = ty * T4_TY, qx0 = tx * T4_TX, n0 = nt * T4_BN;
    const int nsteps = q.nchunks * 16;
    // All DMA in the buffer form of pfa_common.h (buf_dma16): per-lane offsets fixed for the whole loop, the channel chunk (and
    // the tap, for the weights) in ONE scalar per piece, PFA_DMA_OOB where the hardware is to write zeros - out-of-image
    // pixels, the padding piece, every piece of a chunk behind the last one (still issued: all steps issue and wait alike).
    // ---- weight staging: one piece per lane and step: row r0 of the tile, chunk slot cs holds source chunk cs ^ swz(row).
    // Issued first: the latency of stages 0..2 runs under the patch's address arithmetic.
    const int cs = tid & 7, r0 = tid >> 3;
    const unsigned woff = (unsigned)(((n0 + r0) * 16 * p.ctot + (cs ^ ((r0 >> 1) & 7)) * 8) * 2);
    const unsigned ctot2 = (unsigned)p.ctot * 2;
    auto stage_w = [&](unsigned chunk_off, auto v_) {         // (class, tap) = V of the chunk with scalar chunk_off (chunk * 128, or out of range)
        constexpr int V = decltype(v_)::value;
        buf_dma16(p.w, smem + T4_WBASE + (V & (T4_WSLOTS - 1)) * T4_WBYTES + wave * 1024, woff, chunk_off + (unsigned)t4_widx(V) * ctot2);
    };
    stage_w(0u, std::integral_constant<int, 0>{}); stage_w(0u, std::integral_constant<int, 1>{}); stage_w(0u, std::integral_constant<int, 2>{});
    // ---- patch staging: lane's piece j = (pixel slot, 16-byte piece g of its 128 bytes); the ninth piece is padding.  pvoff:
    // pixel * pixel pitch + 16 g for the source the loop is walking (pvoff2: for the second source's pitch, copied over when the
    // walk gets there)
    const unsigned C2 = (unsigned)p.c1 * 2, C2b = (unsigned)p.c2 * 2;
    unsigned pvoff[T4_PPASS], pvoff2[T4_PPASS];
#pragma unroll
    for (int j = 0; j < T4_PPASS; ++j) {
        const int qq = j * 512 + tid, slot = qq / 9, g = qq - slot * 9;
        const int sy = slot / T4_PW, sx = slot - sy * T4_PW;
        const int iy = qy0 - 1 + sy, ix = qx0 - 1 + sx;
        const unsigned ok = (unsigned)(g < 8) & (unsigned)(slot < T4_SLOTS) & (unsigned)((unsigned)iy < (unsigned)p.hs) & (unsigned)((unsigned)ix < (unsigned)p.ws);
        const unsigned keep = 0u - ok, pix = (unsigned)((img * p.hs + iy) * p.ws + ix);
        pvoff[j] = ((pix * C2 + (unsigned)g * 16) & keep) | (PFA_DMA_OOB & ~keep);
        pvoff2[j] = ((pix * C2b + (unsigned)g * 16) & keep) | (PFA_DMA_OOB & ~keep);
    }
    // scalars of a chunk: weights chunk * 128; patch: source base and chunk-in-source * 128
    struct CK { unsigned woff; const char* pbase; unsigned poff; };
    auto make_ck = [&](int chunk) -> CK {
        const bool live = chunk < q.nchunks, first = chunk < q.c1chunks;
        CK r;
        r.woff = live ? (unsigned)chunk * 128 : PFA_DMA_OOB;
        r.pbase = first ? p.src1 : p.src2;
        r.poff = live ? (unsigned)(first ? chunk : chunk - q.c1chunks) * 128 : PFA_DMA_OOB;
        if (!first) {            // the walk is in the second source (every piece of the first one has been issued)
#pragma unroll
            for (int j = 0; j < T4_PPASS; ++j) pvoff[j] = pvoff2[j];
        }
        return r;
    };
    auto stage_patch_piece = [&](const CK& ck, int buf, int j) {
        buf_dma16(ck.pbase, smem + buf * T4_PATCH + (j * 512 + wave * 64) * 16, pvoff[j], ck.poff);
    };
    CK ck_cur = make_ck(0), ck_nxt = ck_cur;
#pragma unroll
    for (int j = 0; j < T4_PPASS; ++j) stage_patch_piece(ck_cur, 0, j);

    f32x16 acc[4][FM];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int b = 0; b < FM; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[c][b][e] = 0.f;

    // ---- fragment addresses (32x32x16: lane (lr, lh) supplies row lr, the 16-byte chunk 2 s + lh of k-step s)
    const unsigned lds0 = (unsigned)(size_t)LDS_PTR(smem);
    int tpos[FM];
    unsigned abuf[2][FM];        // activation tile b (tile row wm*2 + b, column lr) at patch origin, in patch buffer 0 / 1
#pragma unroll
    for (int b = 0; b < FM; ++b) {
        tpos[b] = (wm * 2 + b) * T4_TX + lr;
        abuf[0][b] = lds0 + ((wm * 2 + b) * T4_PW + lr) * T4_PSTRIDE + lh * 16;
        abuf[1][b] = abuf[0][b] + T4_PATCH;
    }
    unsigned wadr[KS];
    {
        const int row = wn * 32 + lr;
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) wadr[s2] = lds0 + T4_WBASE + row * 128 + (((2 * s2 + lh) ^ ((row >> 1) & 7)) << 4);
    }
    bf16x8 fw[KS], fa[FM][KS];   // the ONE fragment set
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) {
        fw[s2] = bf16x8{};
#pragma unroll
        for (int b = 0; b < FM; ++b) fa[b][s2] = bf16x8{};
    }
    // fragment E (0: weights, 1 + b: activation tile b) of k-step S2 of the step with (class, tap) = V, weight slot V & 3,
    // activation base registers ab[]
    auto read_frag = [&](auto v_, auto s2_, auto e_, const unsigned (&ab)[FM]) {
        constexpr int V = decltype(v_)::value, S2 = decltype(s2_)::value, E = decltype(e_)::value;
        if constexpr (E == 0) t4_rd<(V & 3) * T4_WBYTES>(wadr[S2], fw[S2]);
        else t4_rd<t4_tap_off(V) + S2 * 32>(ab[E - 1], fa[E - 1][S2]);
    };
    auto read_kstep = [&](auto v_, auto s2_, const unsigned (&ab)[FM]) {
        read_frag(v_, s2_, std::integral_constant<int, 0>{}, ab);
        read_frag(v_, s2_, std::integral_constant<int, 1>{}, ab);
        read_frag(v_, s2_, std::integral_constant<int, 2>{}, ab);
    };

    // ---- prologue: patch 0 and the weight stages 0..2 are on their way
    t4_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    unsigned acur[FM] = {abuf[0][0], abuf[0][1]}, aoth[FM] = {abuf[1][0], abuf[1][1]};
    read_kstep(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, acur);
    read_kstep(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, acur);
    read_kstep(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}, acur);
    read_kstep(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{}, acur);

    // one tap step of chunk `chunk`: (class, tap) = V; `more`: a step follows (its fragments are read ahead here)
    auto body = [&](int chunk, auto v_, const bool more) {
        constexpr int V = decltype(v_)::value, NV = (V + 1) & 15, CLS = V >> 2;
        if constexpr (V == 0) { ck_cur = ck_nxt; ck_nxt = make_ck(chunk + 1); }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        constexpr bool p_issued = V < T4_PPASS;
        auto kstep = [&](auto s2_) {
            constexpr int S2 = decltype(s2_)::value;
            if (more) PFA_WAIT_LGKM((KS - 1) * NR); else PFA_WAIT_LGKM((KS - 1 - S2) * NR);
            pin(fw[S2]);
#pragma unroll
            for (int b = 0; b < FM; ++b) pin(fa[b][S2]);
            __builtin_amdgcn_sched_barrier(0);
            // the k-step's reads and DMA in the shadows of its two MFMAs (see conv_patch2.hip): M(w,a0) [DMA] R(a0) M(w,a1) R(w) R(a1)
            auto rd = [&](auto e_) {                         // the same k-step of the next tap step, into the registers just consumed
                if (more) {
                    if (V == 15) read_frag(std::integral_constant<int, NV>{}, s2_, e_, aoth);     // the next chunk's patch buffer
                    else read_frag(std::integral_constant<int, NV>{}, s2_, e_, acur);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            acc[CLS][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[S2], fa[0][S2], acc[CLS][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (S2 == 0) stage_w(V + 3 < 16 ? ck_cur.woff : ck_nxt.woff, std::integral_constant<int, (V + 3) & 15>{});      // stage i+3
            if (S2 == 2 && p_issued) stage_patch_piece(ck_nxt, (chunk + 1) & 1, V < T4_PPASS ? V : 0);
            __builtin_amdgcn_sched_barrier(0);
            rd(std::integral_constant<int, 1>{});
            acc[CLS][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[S2], fa[1][S2], acc[CLS][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            rd(std::integral_constant<int, 0>{});
            rd(std::integral_constant<int, 2>{});
        };
        kstep(std::integral_constant<int, 0>{});
        kstep(std::integral_constant<int, 1>{});
        kstep(std::integral_constant<int, 2>{});
        kstep(std::integral_constant<int, 3>{});
        t4_wait_vm<p_issued ? 2 : 1>();
    };
    for (int chunk = 0; chunk < q.nchunks; ++chunk) {         // one 64-channel chunk per trip; every body exists once
        const bool next = chunk + 1 < q.nchunks;
        [&]<int... V>(std::integer_sequence<int, V...>) {
            (body(chunk, std::integral_constant<int, V>{}, V < 15 || next), ...);
        }(std::make_integer_sequence<int, 16>{});
#pragma unroll
        for (int b = 0; b < FM; ++b) { const unsigned t = acur[b]; acur[b] = aoth[b]; aoth[b] = t; }
    }
    t4_wait_vm<0>();                             // (the pieces issued behind the last chunk)

    // ---- epilogue: four passes (one per parity class) through the fp32 [position][channel] image in LDS
    constexpr int RS = patch_image_stride<T4_BN>();
    // (one pass per class through a fold expression, not a `#pragma unroll` loop: were the compiler to leave that loop rolled -
    // its body is the whole row epilogue - acc[cls] would be indexed at run time and the accumulators would live in scratch)
    auto class_pass = [&](auto cls_) {
        constexpr int cls = decltype(cls_)::value;
        __syncthreads();                         // staging buffers / the previous class's image are free
#pragma unroll
        for (int b = 0; b < FM; ++b) {
            char* row = smem + tpos[b] * RS;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {acc[cls][b][4 * g], acc[cls][b][4 * g + 1], acc[cls][b][4 * g + 2], acc[cls][b][4 * g + 3]};
                *(f32x4*)(row + (wn * 32 + 8 * g + 4 * lh) * 4) = v;
            }
        }
        __syncthreads();
        constexpr int py = cls >> 1, px = cls & 1;
        auto pix_of = [&](int r, bool& valid, size_t& opix) {
            const int qy = qy0 + r / T4_TX, qx = qx0 + r % T4_TX;
            valid = qy < p.hq && qx < p.wq;
            opix = valid ? ((size_t)img * p.ho + 2 * qy + py) * p.wo + 2 * qx + px : 0;
        };
        const size_t trow = (((size_t)img * q.tiles_y + ty) * q.tiles_x + tx) * 4 + cls;
        patch_epilogue_rows<T4_BN>(p, n0, tid, 0, trow, smem, pix_of);
    };
    class_pass(std::integral_constant<int, 0>{});
    class_pass(std::integral_constant<int, 1>{});
    class_pass(std::integral_constant<int, 2>{});
    class_pass(std::integral_constant<int, 3>{});
}

}  // namespace

// PFA_TCONV_S2 launches with n % 64 == 0 that the per-class kernel would run on 64-channel tiles without K slices, when the
// four-class workgroups alone still fill the chip.
bool pfa_tconv4_ok(const GatherK& k, int dtype, int tiles_y, int tiles_x) {
    if (pfa_knob(3) == 4) return false;                      // knob 3 = 4: per-class workgroups only (A/B)
    if (dtype != PFA_BF16 || k.mode != PFA_TCONV_S2 || k.wq < 32 || k.ctot < 64) return false;
    if (pfa_knob(3) == 5) return true;                       // knob 3 = 5: whenever the geometry allows (tests)
    return (long long)k.nb * tiles_y * tiles_x * (k.n / T4_BN) >= (pfa_knob(9) > 0 ? pfa_knob(9) : 192);      // knob 9: the threshold
}

int pfa_tconv4(const GatherK& k, int tiles_y, int tiles_x, hipStream_t st, const char** kernel_name) {
    Tconv4K q;
    q.g = k;
    q.g.splitk = 1;
    q.tiles_y = tiles_y; q.tiles_x = tiles_x; q.ntiles = k.n / T4_BN;
    q.nwg = k.nb * tiles_y * tiles_x * q.ntiles;
    q.nchunks = k.ctot / 64; q.c1chunks = k.c1 / 64;
    constexpr int LDS = T4_LDS_LOOP > 256 * patch_image_stride<T4_BN>() ? T4_LDS_LOOP : 256 * patch_image_stride<T4_BN>();
    static std::atomic<bool> attr_done{false};   // (launches may come from several host threads)
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)k_tconv4, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_done = true;
    }
    if (kernel_name) *kernel_name = "k_tconv4<64,8x32>";
    hipLaunchKernelGGL(k_tconv4, dim3(q.nwg), dim3(512), LDS, st, q);
    PFA_LAUNCHED("k_tconv4");
    return 0;
}
