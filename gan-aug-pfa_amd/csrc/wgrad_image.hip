// Weight gradient of the IMAGE-SIDE layers: 8-channel gathered operand (one or two sources of 3 real channels padded to 8),
// 64 or 128 direct channels, stride-2 gather - the generator's e1 = Conv2d(3, 64) and d1 = ConvTranspose2d(128, 3)
// (models.py:183,184) and the discriminator's c1 = Conv2d(6, 64) (models.py:223), as train_gan.py:62,70 differentiate them.
//
//   dw[i][ky*4+kx][c] += sum_positions direct[p][i] * gathered[2 p - 1 + (ky, kx)][c]
//
// On wgrad.hip these launches are one 128 x 128 output tile with ~256 reduction slices: every K step stages 16 pieces of 16
// bytes per position for the gathered operand (one per tap - the same pixel up to four times per step, 134 MB of L2 -> LDS
// traffic for an 8 MB image) next to a half-empty 128-channel direct tile: 19-36 us per launch on the tail of every backward
// pass.  Here a workgroup walks a slice of 4 x 32 position tiles: per tile it stages the direct rows [128 positions][CD] and
// the (10 x 66)-pixel input patch ONCE (odd / even column planes of 16-byte pixels, as conv_image.hip: consecutive positions
// of a tap are consecutive pixels) and takes both MFMA operands through the transpose read ds_read_b64_tr_b16 (the contraction
// runs over positions, both operands are position-major).  An output tile column block of 32 = one tap row ky x four kx x 8
// channels; the 2-8 (row tile, column tile) pairs of the [CD][128 | 256] output are dealt to the 8 waves.  Partial tiles of the
// slices go to a slab, a second kernel adds them in slice order and stores the real channels (bitwise reproducible).
#include "pfa_common.h"
#include <atomic>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

struct WgiK {
    const char* dir1;
    const char* dir2;            // second direct source (d1: the skip half), or NULL
    const char* gat1;
    const char* gat2;            // second gathered source (c1: the B image), or NULL
    float* dw;
    float* slab;                 // [slices][CD][16 taps][8 * NG]
    int d1p, cd, ng;             // channels of direct source 1, direct channels in all (64 | 128), gathered sources (1 | 2)
    int d1r, d2r, g1r, g2r;      // real channels
    int nb, hq, wq, hg, wg;
    int tiles_y, tiles_x, ntiles, slices, sper;
    int fresh;
};

constexpr int GTY = 4, GTX = 32, GPOS = GTY * GTX;           // positions per tile
constexpr int GPH = 2 * GTY + 2, GPL = GTX + 1;               // patch rows; pixels per plane row
constexpr int GSLOTS = GPH * 2 * GPL;                         // 660 pixels of 16 bytes
constexpr int GPATCH = 16 * 1024;                             // per source: 16 wave pieces (2 per wave: a uniform DMA count), 10.3 KiB used

template <int N> __device__ __forceinline__ void wgi_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// CD: direct channels (64 | 128); NG: gathered sources (1 | 2)
template <int CD, int NG>
__global__ __launch_bounds__(512) void k_wgrad_image(const WgiK p) {
    constexpr int DROW = CD * 2;                              // bytes of a direct row (one position)
    constexpr int DBYTES = GPOS * DROW;                       // 16 | 32 KiB
    constexpr int STAGE = DBYTES + NG * GPATCH;
    constexpr int DPIECES = DBYTES / 8192;                    // DMA pieces per lane for the direct tile (2 | 4)
    constexpr int PPIECES = GPATCH / 1024;                    // wave pieces per gathered source (16: two per wave)
    constexpr int NSTG = 3 * STAGE <= 163840 ? 3 : 2;         // ring of stages: NSTG - 1 tiles in flight behind the one computed
    constexpr int OPS = DPIECES + 2 * NG;                     // DMA instructions per wave and tile
    constexpr int MT = CD / 32, NT = 4 * NG;                  // output tiles of 32 x 32
    constexpr int PAIRS = MT * NT / 8;                        // (row tile, column tile) pairs per wave: 1 | 2 | 4
    static_assert(MT * NT % 8 == 0 && PAIRS >= 1, "pairs per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* zero = (const char*)g_pfa_zero_page;
    const int slice = blockIdx.x;
    const int t0 = slice * p.sper, t1 = min(p.ntiles, t0 + p.sper);

    // ---- DMA of tile t into stage st
    auto stage = [&](int t, int st) {
        const int per = p.tiles_y * p.tiles_x;
        const int img = t / per, rem = t - img * per;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const int qy0 = ty * GTY, qx0 = tx * GTX;
        char* sb = smem + st * STAGE;
        // direct rows: piece q = (position, 16-byte chunk); chunk c of a row lives at c ^ swz(position) (bank spread of the
        // transpose reads: 4 consecutive positions x 32 bytes per 16-lane group)
#pragma unroll
        for (int j = 0; j < DPIECES; ++j) {
            const int q = j * 512 + tid;
            const int pos = q / (CD / 8), cpos = q - pos * (CD / 8);
            const int c = cpos ^ ((pos & 3) << 1);                                  // (chunk pairs stay together: a read spans 32 B)
            const int py = pos / GTX, px = pos - py * GTX;
            const int ch = c * 8;
            const bool first = ch < p.d1p;
            const char* base = first ? p.dir1 : p.dir2;
            const int cw = first ? p.d1p : CD - p.d1p, cc = first ? ch : ch - p.d1p;
            const char* gp = base + ((size_t)((img * p.hq + qy0 + py) * p.wq + qx0 + px) * cw + cc) * 2;
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gp), LDS_PTR(sb + (j * 512 + wave * 64) * 16), 16, 0, 0);
        }
        // gathered patches: slot s = (row r, plane pl, index i): input pixel (2 qy0 - 1 + r, 2 (qx0 + i) - 1 + pl)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const char* base = g == 0 ? p.gat1 : p.gat2;
            for (int pc = wave; pc < PPIECES; pc += 8) {
                const int s = pc * 64 + lane;
                const int r = s / (2 * GPL), jj = s - r * (2 * GPL), pl = jj / GPL, i = jj - pl * GPL;
                const int y = 2 * qy0 - 1 + r, x = 2 * (qx0 + i) - 1 + pl;
                const bool ok = s < GSLOTS && (unsigned)y < (unsigned)p.hg && (unsigned)x < (unsigned)p.wg;
                const char* gp = ok ? base + ((size_t)(img * p.hg + y) * p.wg + x) * 16 : zero;
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gp), LDS_PTR(sb + DBYTES + g * GPATCH + pc * 1024), 16, 0, 0);
            }
        }
    };

    // ---- this wave's output tile pairs and fragment addresses (ds_read_b64_tr_b16: lane 4 q + pp of a 16-lane group supplies
    // row q = one of 4 consecutive positions, 4 consecutive columns 4 pp .. 4 pp + 3 of the group's 16; 32x32x16 operand: lanes
    // 0-15 / 16-31 = columns 0-15 / 16-31 of the tile, lanes 32-63 the second 8 positions of the k-step)
    const int grp = lane >> 4, cb = 16 * (grp & 1), lh = grp >> 1, rq = (lane & 15) >> 2, pp = lane & 3;
    int mt[PAIRS], nt[PAIRS];
    unsigned aoff[PAIRS], boff[PAIRS];
#pragma unroll
    for (int k = 0; k < PAIRS; ++k) {
        const int pair = wave * PAIRS + k;
        mt[k] = pair / NT; nt[k] = pair - mt[k] * NT;
        // direct: channel ch of position pos at pos * DROW + ((ch / 8) ^ swz(pos)) * 16 + (ch & 7) * 2, pos = 8 lh + rq (+ 16 s, + 4)
        const int ch = mt[k] * 32 + cb + 4 * pp;
        aoff[k] = (unsigned)((8 * lh + rq) * DROW + ((((ch >> 3) ^ ((rq & 3) << 1))) << 4) + (ch & 7) * 2);
        // gathered: column n = cb + 4 pp of column tile nt: source g = nt / 4, tap row ky = nt & 3, kx = n / 8, channel n & 7
        const int g = nt[k] >> 2, ky = nt[k] & 3, n = cb + 4 * pp, kx = n >> 3, c = n & 7;
        // position (row py, column px) -> patch slot (2 py + ky) * 2 GPL + (kx & 1 ? GPL : 0) + px + (kx >> 1)
        boff[k] = (unsigned)(DBYTES + g * GPATCH + ((ky * 2 * GPL + ((kx & 1) ? GPL : 0) + (kx >> 1) + 8 * lh + rq) * 16) + c * 2);
    }
    f32x16 acc[PAIRS];
#pragma unroll
    for (int k = 0; k < PAIRS; ++k)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;

    auto tr = [&](unsigned addr) -> bf16x4 {
        return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(smem + addr));
    };

    if (t0 < t1) stage(t0, 0);
    if (NSTG == 3 && t0 + 1 < t1) stage(t0 + 1, 1);
    int st = 0;
    for (int t = t0; t < t1; ++t) {
        // tile t has landed (with three stages the younger OPS belong to tile t + 1, if any); after the barrier every wave is
        // also done with tile t - 1, whose stage the next tile to fetch takes
        if (NSTG == 3 && t + 1 < t1) wgi_wait_vm<OPS>(); else wgi_wait_vm<0>();
        __builtin_amdgcn_s_barrier();                        // (raw: __syncthreads() would drain the DMA queue)
        if (t + NSTG - 1 < t1) stage(t + NSTG - 1, st == 0 ? NSTG - 1 : st - 1);
        const unsigned sb = st * STAGE;
        // 8 k-steps of 16 positions: k-step s = tile row s / 2, columns 16 (s & 1) ..; the transpose reads of four k-steps
        // are issued together (their latency overlaps), then their MFMAs
#pragma unroll
        for (int s0 = 0; s0 < GPOS / 16; s0 += 4) {
            bf16x4 fa[4][PAIRS][2], fb[4][PAIRS][2];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = s0 + u, py = s >> 1, px0 = (s & 1) * 16;
                const unsigned dpos = (unsigned)((py * GTX + px0) * DROW);
                const unsigned gpos = (unsigned)((2 * py * 2 * GPL + px0) * 16);
#pragma unroll
                for (int k = 0; k < PAIRS; ++k) {
                    // (the swizzle term uses position & 3 = rq: the k-step and half offsets are multiples of 4 positions)
                    fa[u][k][0] = tr(sb + aoff[k] + dpos); fa[u][k][1] = tr(sb + aoff[k] + dpos + 4 * DROW);
                    fb[u][k][0] = tr(sb + boff[k] + gpos); fb[u][k][1] = tr(sb + boff[k] + gpos + 4 * 16);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < PAIRS; ++k) {
                    const bf16x8 av = __builtin_shufflevector(fa[u][k][0], fa[u][k][1], 0, 1, 2, 3, 4, 5, 6, 7);
                    const bf16x8 bv = __builtin_shufflevector(fb[u][k][0], fb[u][k][1], 0, 1, 2, 3, 4, 5, 6, 7);
                    acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[k], 0, 0, 0);
                }
        }
        st = st == NSTG - 1 ? 0 : st + 1;
    }

    // ---- partial tiles -> slab [slice][CD][NT * 32]: lane holds D[i = 8 g + 4 lh' + r][n = lane & 31] of its tile
    const int col = lane & 31, rh = lane >> 5;
    float* dst = p.slab + (size_t)slice * CD * (NT * 32);
#pragma unroll
    for (int k = 0; k < PAIRS; ++k)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = mt[k] * 32 + (e & 3) + 8 * (e >> 2) + 4 * rh;
            dst[(size_t)i * (NT * 32) + nt[k] * 32 + col] = acc[k][e];
        }
}

// dw[i][tap][c] (+)= sum over the slices.  A block = 64 elements (direct channel, tap, padded gathered channel) x 4 quarters of
// the slices; the quarters meet in LDS in a fixed order (bitwise reproducible).
__global__ __launch_bounds__(256) void k_wgrad_image_reduce(const WgiK p) {
    __shared__ float part[4][64];
    const int ncol = p.ng * 128;
    const int el = threadIdx.x & 63, quarter = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + el;                       // (cd * ncol is a multiple of 64)
    const size_t stride = (size_t)p.cd * ncol;
    const int per = (p.slices + 3) / 4, k0 = quarter * per, k1 = min(p.slices, k0 + per);
    const float* src = p.slab + e + (size_t)k0 * stride;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    int k = k0;
    for (; k + 4 <= k1; k += 4, src += 4 * stride) {
#pragma unroll
        for (int u = 0; u < 4; ++u) s[u] += src[(size_t)u * stride];
    }
    for (int u = 0; k < k1; ++k, ++u, src += stride) s[u] += *src;
    part[quarter][el] = (s[0] + s[1]) + (s[2] + s[3]);
    __syncthreads();
    if (quarter != 0) return;
    const float tot = (part[0][el] + part[1][el]) + (part[2][el] + part[3][el]);
    const int i = e / ncol, n = e - i * ncol;
    const int g = n >> 7, ky = (n >> 5) & 3, kx = (n >> 3) & 3, c = n & 7;
    const int creal = g == 0 ? (c < p.g1r ? c : -1) : (c < p.g2r ? p.g1r + c : -1);
    const int ireal = i < p.d1p ? (i < p.d1r ? i : -1) : ((i - p.d1p) < p.d2r ? p.d1r + i - p.d1p : -1);
    if (creal < 0 || ireal < 0) return;
    float* dst = p.dw + ((size_t)ireal * 16 + ky * 4 + kx) * (p.g1r + p.g2r) + creal;
    *dst = p.fresh ? tot : *dst + tot;
}

template <int CD, int NG>
int launch_wgi(const WgiK& k, hipStream_t st) {
    constexpr int STAGE = GPOS * CD * 2 + NG * GPATCH;
    constexpr int LDS = (3 * STAGE <= 163840 ? 3 : 2) * STAGE;
    static_assert(LDS <= 163840, "LDS");
    static std::atomic<bool> attr_done{false};   // (launches may come from several host threads)
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)k_wgrad_image<CD, NG>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_done = true;
    }
    hipLaunchKernelGGL((k_wgrad_image<CD, NG>), dim3(k.slices), dim3(512), LDS, st, k);
    PFA_LAUNCHED("k_wgrad_image");
    return 0;
}

}  // namespace

// bf16, stride 2, gathered sources of 8 padded channels, 64 or 128 direct channels in 64-channel sources, position grid a
// multiple of the 4 x 32 tile; needs the caller's slab (pfa_wgrad_slab_floats).
bool pfa_wgrad_image_plan(const pfa_wgrad_args* a, int* slices) {
    if (pfa_knob(4) == 7 || pfa_knob(4) == 9) return false;                // knob 4 = 9: never this kernel (A/B)
    if (a->dtype != PFA_BF16 || a->stride != 2) return false;
    if (a->g1p != 8 || !(a->g2p == 0 || a->g2p == 8)) return false;
    const int cd = a->d1p + a->d2p;
    if (!(cd == 64 || cd == 128) || a->d1p % 64 || a->d2p % 64) return false;
    if (a->hq % GTY || a->wq % GTX || a->hg != 2 * a->hq || a->wg != 2 * a->wq) return false;
    if ((long long)a->nb * a->hg * a->wg * 16 >= (1ll << 31) || (long long)a->nb * a->hq * a->wq * cd * 2 >= (1ll << 31)) return false;
    const int ntiles = a->nb * (a->hq / GTY) * (a->wq / GTX);
    if (ntiles < 32) return false;
    const int want = pfa_knob(12) > 0 ? pfa_knob(12) : 256;               // workgroups a launch aims for (knob 12)
    int s = ntiles < want ? ntiles : want;
    const int sper = (ntiles + s - 1) / s;
    s = (ntiles + sper - 1) / sper;
    if (slices) *slices = s;
    return true;
}

size_t pfa_wgrad_image_slab_floats(const pfa_wgrad_args* a, int slices) {
    return (size_t)slices * (a->d1p + a->d2p) * 128 * (a->g2p ? 2 : 1);
}

int pfa_wgrad_image(const pfa_wgrad_args* a, hipStream_t st, const char** name) {
    int slices;
    PFA_REQUIRE(pfa_wgrad_image_plan(a, &slices) && a->slab, "pfa_wgrad_image: geometry not supported or no slab");
    WgiK k;
    k.dir1 = (const char*)a->dir1; k.dir2 = (const char*)a->dir2; k.gat1 = (const char*)a->gat1; k.gat2 = (const char*)a->gat2;
    k.dw = a->dw; k.slab = a->slab; k.fresh = a->fresh;
    k.d1p = a->d1p; k.cd = a->d1p + a->d2p; k.ng = a->g2p ? 2 : 1;
    k.d1r = a->d1r; k.d2r = a->d2r; k.g1r = a->g1r; k.g2r = a->g2r;
    k.nb = a->nb; k.hq = a->hq; k.wq = a->wq; k.hg = a->hg; k.wg = a->wg;
    k.tiles_y = a->hq / GTY; k.tiles_x = a->wq / GTX; k.ntiles = a->nb * k.tiles_y * k.tiles_x;
    k.slices = slices; k.sper = (k.ntiles + slices - 1) / slices;
    int rc;
    if (k.cd == 64) rc = k.ng == 1 ? launch_wgi<64, 1>(k, st) : launch_wgi<64, 2>(k, st);
    else rc = k.ng == 1 ? launch_wgi<128, 1>(k, st) : launch_wgi<128, 2>(k, st);
    if (name) *name = k.cd == 64 ? (k.ng == 1 ? "k_wgrad_image<64,8ch>" : "k_wgrad_image<64,2x8ch>") : (k.ng == 1 ? "k_wgrad_image<128,8ch>" : "k_wgrad_image<128,2x8ch>");
    if (rc) return rc;
    const int total = k.cd * k.ng * 128;
    hipLaunchKernelGGL(k_wgrad_image_reduce, dim3(total / 64), dim3(256), 0, st, k);
    PFA_LAUNCHED("k_wgrad_image_reduce");
    return 0;
}
