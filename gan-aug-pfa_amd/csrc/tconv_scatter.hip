// Transposed 4x4 / stride-2 / pad-1 convolution onto a FEW output channels (n <= 8), gfx950.
//
// Replaces, for the image-side layers only: the outermost ConvTranspose2d(2*ngf -> 3) + bias + Tanh of the generator
// (reference models.py:184-186) and the input gradient of the discriminator's first Conv2d(6 -> ndf, s2)
// (autograd of models.py:223; the G phase needs dL/dfake, train_gan.py:66-70).  Through the implicit-GEMM kernel
// these launches are bound by L2->LDS gather traffic: with only 3..8 output columns every input pixel is staged
// 16 times (4 taps x 4 output-parity classes) for a sliver of MFMA work.
//
// Here the scatter form of the same sum is used (oracle/ops.py conv_transpose2d):
//     out[b, 2*iy-1+ky, 2*ix-1+kx, co] += sum_c x[b, iy, ix, c] * w[co][ky*4+kx][c]
// 1. A workgroup takes an 8 x 32 patch of INPUT pixels (a 6 x 30 interior plus a one-pixel halo), stages each
//    128-byte channel chunk of it once (LDS-DMA, same swizzled row image as gather_gemm.hip) and computes the
//    pointwise GEMM  Y[pixel][tap*NCO + co] = X[pixel][:] . W[tap,co][:]  on MFMA: N = 16 taps x NCO columns.
// 2. col2im inside LDS: every output pixel of the 12 x 60 tile owned by the interior sums its four (tap, input
//    pixel) contributions from Y (fp32), adds the bias, applies the activation and is stored once.
// The input is read ~1.4x (halo) instead of 16x; results do not depend on the tiling (each output's four terms are
// summed in a fixed order: ky ascending, kx ascending).
#include "pfa_common.h"
#include <atomic>

namespace {

struct ScatterK {
    const char* src1;
    const char* src2;
    const char* w;
    const float* bias;
    char* out;
    int nb, hs, ws;          // input grid
    int c1, c2, ctot;        // padded channels of the two sources (virtual concat), their sum
    int n, ld;               // real output channels, padded output row length (elements)
    int act;
    float slope;
    int relu1;               // source 1's fragments pass through ReLU (pfa_scatter_args.relu_src1)
    int ty_tiles, tx_tiles;
    int kchunks, k1chunks;   // 128-byte K chunks in total / coming from src1
    const char* l1_fake;     // fused L1 + Tanh' epilogue (pfa_hip.h), or NULL
    const char* l1_real;
    float l1_lambda;
    int l1_creal;
};

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ f32x4 run(const frag& a, const frag& b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    // max(x, 0) on eight bf16: as 16-bit integers a negative value (sign bit) is a negative integer - one v_pk_max_i16 per pair
    static __device__ __forceinline__ frag relu(const frag& a) {
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        return __builtin_bit_cast(frag, __builtin_elementwise_max(__builtin_bit_cast(s16x8, a), z));
    }
};
template <> struct Mma<float> {
    typedef f32x4 frag;
    static __device__ __forceinline__ f32x4 run(const frag& a, const frag& b, f32x4 c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
        return c;
    }
    static __device__ __forceinline__ frag relu(const frag& a) {
        frag r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = a[j] <= 0.f ? 0.f : a[j];
        return r;
    }
};

constexpr int PH = 8, PW = 32;            // input patch (with halo); interior (PH-2) x (PW-2)
constexpr int NPIX = PH * PW;             // 256 rows of the pointwise GEMM = 4 waves x 64

template <int NCO> constexpr int scatter_lds_bytes() {
    constexpr int N = 16 * NCO;
    constexpr int stage = NPIX * 128 + N * 128;
    constexpr int ybytes = NPIX * ((N / 2) * 4 + 16);
    return stage > ybytes ? stage : ybytes;
}

template <typename T, int NCO>
__global__ __launch_bounds__(256) void k_tconv_scatter(const ScatterK p) {
    constexpr int CH = ElemTraits<T>::CH;
    constexpr int N = 16 * NCO, FN = N / 16, FM = 4;
    constexpr int X_BYTES = NPIX * 128;
    constexpr int WPASS = N / 32;                 // weight rows staged per thread
    constexpr int NP = N / 2;                     // Y columns of one output-row parity
    constexpr int YSTRIDE = NP * 4 + 16;          // bytes; +16 spreads the 16-byte column slots of 8 rows over all banks
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 15, kk = lane >> 4;
    unsigned wid = blockIdx.x;
    const int tx = wid % p.tx_tiles; wid /= p.tx_tiles;
    const int ty = wid % p.ty_tiles;
    const int img = wid / p.ty_tiles;
    const int Y0 = ty * (PH - 2), X0 = tx * (PW - 2);      // first interior input pixel; the patch starts one before
    const char* zero = (const char*)g_pfa_zero_page;

    // ---- staging roles: chunk slot cs of patch rows r0 + 32*i, i.e. patch pixel (py = i, px = r0)
    const int cs = tid & 7, r0 = tid >> 3;
    const int g = cs ^ ((r0 >> 1) & 7);                     // swizzle on the source side ((row >> 1) & 7 is i-independent)
    int pixoff[PH];
    unsigned okx = 0;
#pragma unroll
    for (int i = 0; i < PH; ++i) {
        const int iy = Y0 - 1 + i, ix = X0 - 1 + r0;
        const bool ok = (unsigned)iy < (unsigned)p.hs && (unsigned)ix < (unsigned)p.ws;
        pixoff[i] = ok ? (img * p.hs + iy) * p.ws + ix : 0;
        okx |= ok ? (1u << i) : 0u;
    }
    int woff[WPASS];
    unsigned okw = 0;
#pragma unroll
    for (int j = 0; j < WPASS; ++j) {
        const int wr = r0 + 32 * j;                         // GEMM column = tap * NCO + co
        const int tap = wr / NCO, co = wr % NCO;
        woff[j] = (co * 16 + tap) * p.ctot;
        okw |= (co < p.n) ? (1u << j) : 0u;
    }

    f32x4 acc[FN][FM];
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kc = 0; kc < p.kchunks; ++kc) {
        const bool second = kc >= p.k1chunks;
        const char* base = second ? p.src2 : p.src1;
        const int csrc = second ? p.c2 : p.c1;
        const int cc = ((second ? kc - p.k1chunks : kc) * 8 + g) * CH;      // channel inside its source
        const int cw = (kc * 8 + g) * CH;                                   // channel inside the weight row
#pragma unroll
        for (int i = 0; i < PH; ++i) {
            const bool ok = (okx >> i) & 1u;
            const char* gp = ok ? base + ((size_t)pixoff[i] * csrc + cc) * sizeof(T) : zero;
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gp), LDS_PTR(smem + (i * 256 + wave * 64) * 16), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < WPASS; ++j) {
            const bool ok = (okw >> j) & 1u;
            const char* gp = ok ? p.w + ((size_t)woff[j] + cw) * sizeof(T) : zero;
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gp), LDS_PTR(smem + X_BYTES + (j * 256 + wave * 64) * 16), 16, 0, 0);
        }
        __syncthreads();                                    // the chunk has landed (vmcnt(0))
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ch = kk + 4 * h;
            typename Mma<T>::frag wf[FN], af[FM];
#pragma unroll
            for (int a = 0; a < FN; ++a) {
                const int row = a * 16 + lrow;
                wf[a] = *(const typename Mma<T>::frag*)(smem + X_BYTES + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int b = 0; b < FM; ++b) {
                const int row = wave * 64 + b * 16 + lrow;
                af[b] = *(const typename Mma<T>::frag*)(smem + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4));
                if (p.relu1 && !second) af[b] = Mma<T>::relu(af[b]);
            }
#pragma unroll
            for (int a = 0; a < FN; ++a)
#pragma unroll
                for (int b = 0; b < FM; ++b) acc[a][b] = Mma<T>::run(wf[a], af[b], acc[a][b]);
        }
        __syncthreads();                                    // everyone is done with this chunk's image
    }

    // ---- col2im.  Lane holds Y[pixel = wave*64 + b*16 + lrow][column = a*16 + 4*kk + r].
    //      NCO == 4: column = tap*4 + co      -> tap = 4a + kk,          co = r      (ky = a,      kx = kk)
    //      NCO == 8: column = tap*8 + co      -> tap = 2a + (kk >> 1),   co = 4*(kk&1) + r   (ky = a >> 1)
    // Output rows of one parity use two ky values; each pass writes that half of Y (fp32) and gathers from it.
    const int HO = 2 * p.hs, WO = 2 * p.ws;
    const float l1_inv = 1.f / ((float)((int64_t)p.nb * HO * WO) * (float)p.l1_creal);      // (pfa_l1_tanh_bwd's 1 / (count * c_real))
    float bv[NCO];                       // bias of the real channels, fetched once (uniform addresses: scalar loads)
#pragma unroll
    for (int e = 0; e < NCO; ++e) bv[e] = 0.f;
    if (p.bias) {
#pragma unroll
        for (int e = 0; e < NCO; ++e)
            if (e < p.n) bv[e] = p.bias[e];
    }
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        // pass 0: ky in {1, 3} -> even output rows;  pass 1: ky in {0, 2} -> odd output rows
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            const int ky = NCO == 4 ? a : (a >> 1);
            if ((ky & 1) != (pass == 0 ? 1 : 0)) continue;
            const int kx = NCO == 4 ? kk : (2 * (a & 1) + (kk >> 1));
            const int col = ((ky >> 1) * 4 + kx) * NCO + (NCO == 4 ? 0 : 4 * (kk & 1));
#pragma unroll
            for (int b = 0; b < FM; ++b) {
                const int m = wave * 64 + b * 16 + lrow;
                *(f32x4*)(smem + m * YSTRIDE + col * 4) = acc[a][b];
            }
        }
        __syncthreads();
        constexpr int TW = 2 * (PW - 2), TH = PH - 2;       // output columns of the tile, output rows of this parity
        for (int idx = tid; idx < TH * TW; idx += 256) {
            const int j = idx / TW, x = idx - j * TW;
            const int oy = 2 * (Y0 + j) + pass, ox = 2 * X0 + x;
            if (oy >= HO || ox >= WO) continue;
            // patch row of the two ky terms (ascending ky): pass 0: ky=1 -> iy=Y0+j, ky=3 -> iy=Y0+j-1
            //                                               pass 1: ky=0 -> iy=Y0+j+1, ky=2 -> iy=Y0+j
            const int py0 = pass == 0 ? j + 1 : j + 2;      // patch row of the first (smaller) ky; the second is py0 - 1
            const int xi = x >> 1, xodd = x & 1;
            // x even: kx=1 -> ix=X0+xi, kx=3 -> ix=X0+xi-1;   x odd: kx=0 -> ix=X0+xi+1, kx=2 -> ix=X0+xi
            const int px0 = xodd ? xi + 2 : xi + 1;         // patch column of the first (smaller) kx; the second is px0 - 1
            const int kx0 = xodd ? 0 : 1;
            float v[NCO];
#pragma unroll
            for (int e = 0; e < NCO; ++e) v[e] = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)                      // ky ascending
#pragma unroll
                for (int u = 0; u < 2; ++u) {                // kx ascending
                    const int pix = (py0 - t) * PW + (px0 - u);
                    const int col = (t * 4 + kx0 + 2 * u) * NCO;
                    const float* yp = (const float*)(smem + pix * YSTRIDE + col * 4);
#pragma unroll
                    for (int e = 0; e < NCO; e += 4) {
                        const f32x4 q = *(const f32x4*)(yp + e);
                        v[e] += q[0]; v[e + 1] += q[1]; v[e + 2] += q[2]; v[e + 3] += q[3];
                    }
                }
            const size_t orow = ((size_t)(img * HO + oy) * WO + ox) * p.ld;
            T* op = (T*)p.out + orow;
#pragma unroll
            for (int e0 = 0; e0 < 8; e0 += CH) {            // the padded row is 4 (f32) or 8 elements wide at most
                if (e0 >= p.ld) break;
                Chunk<T> o, f, rl;
                if (p.l1_fake) {
                    f.load((const T*)p.l1_fake + orow + e0);
                    rl.load((const T*)p.l1_real + orow + e0);
                }
#pragma unroll
                for (int e = 0; e < CH; ++e) {
                    const int c = e0 + e;
                    float t = 0.f;                           // pad lanes are exact zeros
                    if (c < NCO && c < p.n) t = apply_act(v[c < NCO ? c : 0] + bv[c < NCO ? c : 0], p.act, p.slope);
                    if (p.l1_fake) {                         // dpre from dL/dfake as it would have been stored: the
                        const float dgan = to_f32(from_f32<T>(t));     // (pfa_l1_tanh_bwd reads the stored value)
                        t = c < p.l1_creal ? l1_tanh_grad(f.v[e], rl.v[e], dgan, p.l1_lambda, l1_inv) : 0.f;
                    }
                    o.v[e] = t;
                }
                o.store(op + e0);
            }
        }
        __syncthreads();                                    // Y is rewritten by the next pass
    }
}

template <typename T, int NCO>
int launch_scatter(const ScatterK& k, hipStream_t st) {
    constexpr int lds = scatter_lds_bytes<NCO>();
    static std::atomic<bool> attr_done{false};   // (launches may come from several host threads)
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)k_tconv_scatter<T, NCO>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    const unsigned grid = (unsigned)(k.nb * k.ty_tiles * k.tx_tiles);
    hipLaunchKernelGGL((k_tconv_scatter<T, NCO>), dim3(grid), dim3(256), lds, st, k);
    PFA_LAUNCHED("k_tconv_scatter");
    return 0;
}

}  // namespace

extern "C" int pfa_tconv_scatter_supported(int dtype, int c1, int c2, int n, int ld) {
    if (dtype != PFA_BF16 && dtype != PFA_F32) return 0;
    const int kch = 8 * pfa_ch(dtype);                      // channels per 128-byte chunk
    if (c1 <= 0 || c1 % kch || c2 < 0 || c2 % kch) return 0;
    if (n < 1 || n > 8 || ld < n || ld > 8 || ld % pfa_ch(dtype)) return 0;
    return 1;
}

extern "C" int pfa_tconv_scatter(const pfa_scatter_args* a, pfa_stream s) {
    hipStream_t st = (hipStream_t)s;
    PFA_REQUIRE(a && a->struct_size == sizeof(pfa_scatter_args), "pfa_tconv_scatter: bad struct_size");
    PFA_REQUIRE(pfa_tconv_scatter_supported(a->dtype, a->c1, a->c2, a->n, a->ld),
                "pfa_tconv_scatter: needs c1, c2 multiples of one 128-byte chunk, 1 <= n <= 8, n <= ld <= 8 (got c1=%d c2=%d n=%d ld=%d)",
                a->c1, a->c2, a->n, a->ld);
    PFA_REQUIRE(a->nb > 0 && a->hs > 0 && a->ws > 0, "pfa_tconv_scatter: empty grid");
    PFA_REQUIRE(a->src1 && a->w && a->out && (a->c2 == 0 || a->src2), "pfa_tconv_scatter: null operand");
    PFA_REQUIRE((long long)a->nb * a->hs * a->ws * (a->c1 > a->c2 ? a->c1 : a->c2) < (1ll << 31),
                "pfa_tconv_scatter: source larger than 2^31 elements");
    PFA_REQUIRE(a->act == PFA_ACT_NONE || a->act == PFA_ACT_TANH || a->act == PFA_ACT_RELU || a->act == PFA_ACT_LRELU,
                "pfa_tconv_scatter: unknown activation");
    ScatterK k;
    k.src1 = (const char*)a->src1; k.src2 = (const char*)a->src2; k.w = (const char*)a->w;
    k.bias = a->bias; k.out = (char*)a->out;
    k.nb = a->nb; k.hs = a->hs; k.ws = a->ws;
    k.c1 = a->c1; k.c2 = a->c2; k.ctot = a->c1 + a->c2;
    k.n = a->n; k.ld = a->ld; k.act = a->act; k.slope = a->slope; k.relu1 = a->relu_src1 != 0;
    PFA_REQUIRE((a->l1_fake == nullptr) == (a->l1_real == nullptr), "pfa_tconv_scatter: l1_fake and l1_real come together");
    PFA_REQUIRE(!a->l1_fake || (a->act == PFA_ACT_NONE && !a->bias && a->l1_creal > 0 && a->l1_creal <= a->n),
                "pfa_tconv_scatter: the fused L1 epilogue takes a plain input-gradient launch (no bias / activation), 0 < l1_creal <= n");
    k.l1_fake = (const char*)a->l1_fake; k.l1_real = (const char*)a->l1_real; k.l1_lambda = a->l1_lambda; k.l1_creal = a->l1_creal;
    k.ty_tiles = (a->hs + PH - 3) / (PH - 2);
    k.tx_tiles = (a->ws + PW - 3) / (PW - 2);
    const int kch = 8 * pfa_ch(a->dtype);
    k.kchunks = k.ctot / kch; k.k1chunks = a->c1 / kch;
    const bool wide = a->n > 4;
    if (a->dtype == PFA_BF16) return wide ? launch_scatter<bf16_t, 8>(k, st) : launch_scatter<bf16_t, 4>(k, st);
    return wide ? launch_scatter<float, 8>(k, st) : launch_scatter<float, 4>(k, st);
}
