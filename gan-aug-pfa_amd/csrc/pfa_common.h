// Shared host/device helpers for libpfa_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>

#include "../../include/pfa_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// ---------------------------------------------------------------- host side plumbing
extern thread_local char g_pfa_err[512];
extern std::atomic<uint64_t> g_pfa_launches;

#define PFA_FAIL(code, ...)                                   \
    do {                                                      \
        snprintf(g_pfa_err, sizeof(g_pfa_err), __VA_ARGS__);  \
        return (code);                                        \
    } while (0)

#define PFA_REQUIRE(cond, ...) \
    do {                       \
        if (!(cond)) PFA_FAIL(-1, __VA_ARGS__); \
    } while (0)

// Call after every kernel launch: counts it and turns a launch error into a return code.
#define PFA_LAUNCHED(name)                                                              \
    do {                                                                                \
        g_pfa_launches.fetch_add(1, std::memory_order_relaxed);                         \
        hipError_t e__ = hipGetLastError();                                             \
        if (e__ != hipSuccess) PFA_FAIL((int)e__, "%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

// developer knobs and the per-launch timing store (csrc/gather_gemm.hip)
extern std::atomic<int> g_knob[16];
static inline int pfa_knob(int i) { return g_knob[i].load(std::memory_order_relaxed); }
bool pfa_timing_begin(hipStream_t st, hipEvent_t* e0, hipEvent_t* e1);      // false: timing is off (or the store is full)
void pfa_timing_end(hipStream_t st, hipEvent_t e0, hipEvent_t e1, const char* name, double flop, double bytes, int splitk, int nwg = 0);   // nwg: the launch's workgroups where the caller knows them (else 0)

extern void* g_pfa_dbgbuf;              // pfa_set_debug_buffer: device memory a kernel may write developer stamps to
extern size_t g_pfa_dbgbuf_bytes;

// csrc/conv_patch.hip: stride-1 convolution forward with the input patch kept in LDS (dispatched by pfa_gather_gemm)
bool pfa_conv_s1_patch_ok(int grad, int nb, int hin, int win, int cin, int n, int splitk, int* stat_tiles);
int pfa_conv_s1_patch(int grad, const void* src, const void* w, void* out, float* slab, int splitk, float* psum, float* psq,
                      int nb, int hin, int win, int cin, int n, hipStream_t st);

// csrc/wgrad_patch.hip: weight gradient with the gathered patch kept in LDS (dispatched by pfa_wgrad)
bool pfa_wgrad_patch_plan(const pfa_wgrad_args* a, int* slices, int* tx_out, int* steps_out);
size_t pfa_wgrad_patch_slab_floats(const pfa_wgrad_args* a, int slices);
int pfa_wgrad_patch(const pfa_wgrad_args* a, hipStream_t st, const char** name);

// csrc/wgrad_image.hip: weight gradient of the image-side layers (8-channel gathered operand)
bool pfa_wgrad_image_plan(const pfa_wgrad_args* a, int* slices);
size_t pfa_wgrad_image_slab_floats(const pfa_wgrad_args* a, int slices);
int pfa_wgrad_image(const pfa_wgrad_args* a, hipStream_t st, const char** name);

static inline int pfa_ch(int dtype) { return dtype == PFA_BF16 ? 8 : 4; }
static inline int pfa_esize(int dtype) { return dtype == PFA_BF16 ? 2 : 4; }
static inline int ilog2_exact(int v) {  // log2 if power of two else -1
    if (v <= 0 || (v & (v - 1))) return -1;
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

// ---------------------------------------------------------------- device helpers
template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
    static constexpr int CH = 4;   // elements per 16-byte chunk
    static constexpr int DT = PFA_F32;
};
template <> struct ElemTraits<bf16_t> {
    static constexpr int CH = 8;
    static constexpr int DT = PFA_BF16;
};

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// One 16-byte chunk of T as floats.
template <typename T> struct Chunk;
template <> struct Chunk<float> {
    float v[4];
    __device__ __forceinline__ void load(const float* p) {
        f32x4 t = *(const f32x4*)p;
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    }
    __device__ __forceinline__ void store(float* p) const {
        f32x4 t = {v[0], v[1], v[2], v[3]};
        *(f32x4*)p = t;
    }
};
template <> struct Chunk<bf16_t> {
    float v[8];
    __device__ __forceinline__ void load(const bf16_t* p) {
        bf16x8 t = *(const bf16x8*)p;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
    }
    __device__ __forceinline__ void store(bf16_t* p) const {
        bf16x8 t;
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = (bf16_t)v[i];
        *(bf16x8*)p = t;
    }
};

// The piecewise-linear activations are ONE select with a uniform coefficient: act(v) = v > 0 ? v : neg * v, act'(y) = y > 0 ? 1 : neg
// (LeakyReLU: neg = slope, ReLU: 0, none: 1).  A `switch (act)` per element compiles to scalar branches per element - the
// whole-row epilogues spent most of their time in them; kernels hoist act_neg() out of their element loops.
__device__ __forceinline__ float act_neg(int act, float slope) { return act == PFA_ACT_LRELU ? slope : act == PFA_ACT_RELU ? 0.f : 1.f; }
// Edge cases as torch: a NaN stays a NaN (the compare is false for it: a diverged run must show, not train on zeros);
// -inf (a bf16 conv output that overflowed) gives -inf under LeakyReLU and 0 under ReLU, where -inf * 0 would be NaN and
// poison the BatchNorm statistics behind it: the non-positive branch clamps at -FLT_MAX when neg == 0 (the bound is uniform
// and hoists with neg).  Four VALU operations per element - the whole-row epilogues are VALU-bound: a second compare and
// select for the NaN case cost the inference forward 8 %.
__device__ __forceinline__ float act_lin(float v, float neg) {
    const float lo = neg != 0.f ? -__builtin_inff() : -3.402823466e38f;
    return v <= 0.f ? fmaxf(v, lo) * neg : v;
}
__device__ __forceinline__ float act_grad_lin(float y, float neg) { return y > 0.f ? 1.f : neg; }
__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    if (act == PFA_ACT_TANH) return tanhf(v);
    return act_lin(v, act_neg(act, slope));
}
// derivative of the activation evaluated at its INPUT y (torch: x > 0 ? g : g*slope)
__device__ __forceinline__ float act_grad(float y, int act, float slope) { return act_grad_lin(y, act_neg(act, slope)); }

// d loss_g / d (pre-Tanh output) of one element (train_gan.py:67-69 through models.py:186): (lambda sign(fake - real) / N + dgan)
// (1 - fake^2), every product and sum rounded on its own (no contraction) - two kernels form it (pfa_l1_tanh_bwd and the
// fused epilogue of pfa_tconv_scatter) and must give the same bits.
__device__ __forceinline__ float l1_tanh_grad(float f, float r, float dgan, float lambda, float inv) {
#pragma clang fp contract(off)          // (HIP's __fmul_rn / __fadd_rn are plain operators: the compiler would still fuse them)
    const float df = f - r;
    const float sgn = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
    const float a = lambda * sgn;
    const float b = a * inv;
    const float d = b + dgan;
    const float ff = f * f;
    const float t = 1.f - ff;
    return d * t;
}

// LDS-DMA in its BUFFER form (buffer_load_dwordx4 v_off, s[rsrc], s_off offen lds): lane's 16 bytes from
// base + voff (per-lane VGPR) + soff (wave-uniform SGPR) land at lds + lane * 16 (lds wave-uniform, through M0).  What it buys
// over global_load_lds with a per-lane 64-bit pointer: the part of a piece's address that changes from step to step (channel
// chunk, tap, parity plane) is ONE scalar, the per-lane part is computed once per workgroup, and a lane that must read zeros
// (out-of-image pixel, padding piece) carries the offset PFA_DMA_OOB - the range check against num_records = 2 GiB (VGPR +
// SGPR offset, measured on gfx950: tools/micro/bufdma_check.hip) makes the hardware write zeros to its LDS bytes: no zero
// page, no select, no 64-bit address arithmetic.  In the patch kernels' K loops that arithmetic was ~2/3 of the instruction
// stream (profiles/r04_issue_mix.txt).  Every in-range voff + soff must stay below 2 GiB.
// (A plain function on purpose: written inside a kernel template's lambda the builtin made hipcc drop the kernel's host stub
// without a diagnostic.)
constexpr unsigned PFA_DMA_OOB = 0x80000000u;
__device__ __forceinline__ void buf_dma16(const void* base, void* lds, unsigned voff, unsigned soff) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)0x80000000u, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(lds), 16, (int)voff, (int)soff, 0, 0);
}

// XCD-aware work-item id (cdna guide T1, bijective form).  Workgroups are dealt round-robin over the 8 XCDs,
// each with a private L2: give every XCD one CONTIGUOUS chunk of the work list so that workgroups which share an
// operand tile hit the same L2.  Placement only changes speed, never results.
__device__ __forceinline__ unsigned xcd_remap(unsigned orig, unsigned nwg) {
    const unsigned xcd = orig & 7u, q = nwg >> 3, r = nwg & 7u;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

// 256 bytes of zeros in the code object: out-of-image taps, K tails and missing weight rows
// point their LDS-DMA source here instead of branching.
// (static: each translation unit carries its own copy, so no relocatable device code is needed.)
static __device__ __attribute__((aligned(256))) uint32_t g_pfa_zero_page[64];
