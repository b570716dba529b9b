// Stride-2 4x4 convolution of an IMAGE-side layer: few input channels (one or two sources of 8 padded bf16 channels),
// 64 output channels - the generator's outermost Conv2d(3, 64) (models.py:183) and the discriminator's first
// Conv2d(6, 64) + bias + LeakyReLU (models.py:223), forward.
//
// These launches are HBM-shaped, not MFMA-shaped: K = 16 taps x 8|16 channels = 128|256, 0.8-1.6 GMAC against 50-80 MB of
// activations.  On the implicit GEMM (128 x 64 tile, one K-step, per-lane tap decoding) they took 37-50 us, a third of
// the HBM rate.  Here a workgroup owns 8 x 32 output positions x 64 channels: it loads its whole (18 x 66)-pixel input
// patch - as odd / even column planes, so that the stride-2 taps read consecutive 16-byte pixels - and the whole weight
// matrix into LDS in ONE round of LDS-DMA, runs its 8|16 MFMA k-steps out of LDS, and leaves through the whole-row
// epilogue of the patch kernels (patch_epilogue.h: bias, activation, the identity / eval-mode affine map with two
// activation outputs, BatchNorm partial sums).  Two workgroups fit a CU: one's epilogue overlaps the other's loads.
#include "patch_epilogue.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

struct ImageK {
    GatherK g;
    int tiles_y, tiles_x, ntiles, nwg;      // ntiles: 64-channel tiles of the output (n = 64 or 128)
    unsigned long long* stamps;             // developer timing (pfa_set_debug_buffer): 16 slots per workgroup, or NULL
};

constexpr int ITY = 8, ITX = 32;
constexpr int IPH = 2 * ITY + 2, IPL = ITX + 1;           // patch rows; pixels per plane row (odd plane / even plane)
constexpr int IPSLOTS = IPH * 2 * IPL;                    // 18 x 66 = 1188 pixel slots of 16 bytes per source

// NSRC: 1 (K = 128) or 2 (K = 256: k-step = one tap, the two 8-channel halves from the two sources)
template <int NSRC>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4))) void k_conv_image(const ImageK q) {
    constexpr int K = 128 * NSRC, KS = K / 16;            // contraction length, MFMA k-steps
    constexpr int PATCH_BYTES = ((IPSLOTS * 16 + 1023) / 1024) * 1024;      // per source, whole wave pieces
    constexpr int W_BYTES = 64 * K * 2;
    constexpr int PPIECES = PATCH_BYTES / 1024;           // 1-KiB wave pieces per source (19)
    constexpr int WPIECES = W_BYTES / 1024;               // 16 | 32
    const GatherK& p = q.g;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    unsigned wid = blockIdx.x;
    const int nt = wid % q.ntiles; wid /= q.ntiles;          // (the n-tiles of a patch are neighbours: the second finds it in L2)
    const int tx = wid % q.tiles_x; wid /= q.tiles_x;
    const int ty = wid % q.tiles_y, img = wid / q.tiles_y;
    const int qy0 = ty * ITY, qx0 = tx * ITX, n0 = nt * 64;
    const char* zero = (const char*)g_pfa_zero_page;
    unsigned long long* stamps = q.stamps ? q.stamps + (size_t)blockIdx.x * 16 : nullptr;
    auto stamp = [&](int k) { if (stamps && tid == 0) stamps[k] = __builtin_amdgcn_s_memtime(); };
    stamp(0);
    if (stamps && tid == 0) stamps[8] = __builtin_amdgcn_s_memrealtime();

    // ---- one round of DMA: the patch of every source (pieces dealt round-robin over the waves), then the weights.
    // Patch slot s = (row r, plane pl, index i): input pixel (2 qy0 - 1 + r, 2 (qx0 + i) - 1 + pl).
#pragma unroll
    for (int src = 0; src < NSRC; ++src) {
        const char* base = src == 0 ? p.src1 : p.src2;
        for (int pc = wave; pc < PPIECES; pc += 8) {
            const int s = pc * 64 + lane;
            const int r = s / (2 * IPL), j = s - r * (2 * IPL), pl = j / IPL, i = j - pl * IPL;
            const int y = 2 * qy0 - 1 + r, x = 2 * (qx0 + i) - 1 + pl;
            const bool ok = s < IPSLOTS && (unsigned)y < (unsigned)p.hs && (unsigned)x < (unsigned)p.ws;
            const char* gp = ok ? base + ((size_t)(img * p.hs + y) * p.ws + x) * 16 : zero;
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gp), LDS_PTR(smem + src * PATCH_BYTES + pc * 1024), 16, 0, 0);
        }
    }
    // weights [64 n][K] (the layer's "same" shadow: [n][16 taps][c1 + c2]), 16-byte chunk c of row n at c ^ (n & 15)
    {
        char* wl = smem + NSRC * PATCH_BYTES;
        constexpr int CPR = K / 8;                        // chunks per row
        for (int pc = wave; pc < WPIECES; pc += 8) {
            const int s = pc * 64 + lane, n = s / CPR, cpos = s - n * CPR;
            const int c = (cpos & ~15) | ((cpos ^ n) & 15);
            const char* gp = n0 + n < p.n ? p.w + ((size_t)(n0 + n) * K + c * 8) * 2 : zero;
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gp), LDS_PTR(wl + pc * 1024), 16, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    stamp(1);

    // ---- MFMA 32x32x16: wave = 32 positions (tile row wave, all 32 columns) x 64 channels (two 32-row weight tiles).
    // k-step s, lane half lh: NSRC 1: tap 2 s + lh, the pixel's 8 channels; NSRC 2: tap s, source lh.
    f32x16 acc[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    const char* wl = smem + NSRC * PATCH_BYTES;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int tap = NSRC == 1 ? 2 * s + lh : s;
        const int tyy = tap >> 2, txx = tap & 3;
        // output (row wave, column lr), tap (tyy, txx): input row 2 wave + tyy of the patch, column 2 lr - 1 + txx of the
        // tile = plane (txx & 1 ? even : odd) index lr + (txx >> 1)
        const int slot = (2 * wave + tyy) * (2 * IPL) + ((txx & 1) ? IPL : 0) + lr + (txx >> 1);
        const bf16x8 av = *(const bf16x8*)(smem + (NSRC == 2 ? lh * PATCH_BYTES : 0) + slot * 16);
        const int chunk = 2 * s + lh;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int n = a * 32 + lr;
            const int cpos = (chunk & ~15) | ((chunk ^ n) & 15);
            const bf16x8 wv = *(const bf16x8*)(wl + (n * (K / 8) + cpos) * 16);
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, av, acc[a], 0, 0, 0);
        }
    }

    // ---- epilogue: accumulators -> fp32 [position][channel] image -> whole rows out
    constexpr int RS = patch_image_stride<64>();
    __syncthreads();                             // every wave is done with the patch and the weights
    stamp(2);
    {
        char* row = smem + (wave * ITX + lr) * RS;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {acc[a][4 * g], acc[a][4 * g + 1], acc[a][4 * g + 2], acc[a][4 * g + 3]};
                *(f32x4*)(row + (a * 32 + 8 * g + 4 * lh) * 4) = v;
            }
    }
    __syncthreads();
    stamp(3);
    auto pix_of = [&](int r, bool& valid, size_t& opix) {
        const int qy = qy0 + r / ITX, qx = qx0 + r % ITX;
        valid = qy < p.hq && qx < p.wq;
        opix = valid ? ((size_t)img * p.hq + qy) * p.wq + qx : 0;
    };
    const size_t trow = ((size_t)img * q.tiles_y + ty) * q.tiles_x + tx;
    patch_epilogue_rows<64>(p, n0, tid, 0, trow, smem, pix_of, stamps);
    if (stamps) {
        stamp(6);
        __builtin_amdgcn_s_waitcnt(0);
        stamp(7);
        if (tid == 0) stamps[9] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int NSRC>
int launch_image(const ImageK& k, hipStream_t st) {
    constexpr int K = 128 * NSRC;
    constexpr int PATCH_BYTES = ((IPSLOTS * 16 + 1023) / 1024) * 1024;
    constexpr int IN_BYTES = NSRC * PATCH_BYTES + 64 * K * 2;
    constexpr int OUT_BYTES = 256 * patch_image_stride<64>();
    constexpr int LDS = IN_BYTES > OUT_BYTES ? IN_BYTES : OUT_BYTES;
    static std::atomic<bool> attr_done{false};   // (launches may come from several host threads)
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)k_conv_image<NSRC>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_done = true;
    }
    hipLaunchKernelGGL((k_conv_image<NSRC>), dim3(k.nwg), dim3(512), LDS, st, k);
    PFA_LAUNCHED("k_conv_image");
    return 0;
}

}  // namespace

// bf16 PFA_CONV_S2 launches with 8-channel sources (one or two), n = 64 or 128 (two 64-channel tiles per patch: the input
// gradient of the generator's output layer, whose two halves are the skip gradient and the fused BatchNorm backward of d2),
// no K slices, no tanh.
bool pfa_conv_image_ok(const GatherK& k, int dtype, int* stat_tiles) {
    if (pfa_knob(3) == 1 || pfa_knob(3) == 3) return false;
    if (dtype != PFA_BF16 || k.mode != PFA_CONV_S2 || !k.small32) return false;
    if (k.c1 != 8 || !(k.c2 == 0 || k.c2 == 8) || !(k.n == 64 || k.n == 128) || k.act == PFA_ACT_TANH) return false;
    if (k.fuse_out == PFA_FUSE_DEFER) return false;
    if (k.hq < 8 || k.wq < 16) return false;
    if (stat_tiles) *stat_tiles = k.nb * ((k.hq + ITY - 1) / ITY) * ((k.wq + ITX - 1) / ITX);
    return true;
}

int pfa_conv_image(const GatherK& k, hipStream_t st, const char** kernel_name) {
    ImageK q;
    q.g = k;
    q.g.splitk = 1;
    q.tiles_y = (k.hq + ITY - 1) / ITY; q.tiles_x = (k.wq + ITX - 1) / ITX;
    q.ntiles = k.n / 64;
    q.nwg = k.nb * q.tiles_y * q.tiles_x * q.ntiles;
    q.stamps = (g_pfa_dbgbuf && g_pfa_dbgbuf_bytes >= (size_t)q.nwg * 128) ? (unsigned long long*)g_pfa_dbgbuf : nullptr;
    if (kernel_name) *kernel_name = k.c2 ? "k_conv_image<2x8ch,64>" : "k_conv_image<8ch,64>";
    return k.c2 ? launch_image<2>(q, st) : launch_image<1>(q, st);
}
