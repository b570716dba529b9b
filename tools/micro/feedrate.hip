// How fast can one CU take operand bytes in from L2 while its matrix pipes run?  A K loop with the shape of the patch
// kernels (512 threads, one workgroup per CU, 16 MFMA 32x32x16 + 16 ds_read_b128 per wave and step, one barrier per
// step) is fed three ways: by LDS-DMA pieces (global_load_lds_dwordx4, 1 KiB per wave instruction), by register staging
// (global_load_dwordx4 -> ds_write_b128 one step later) and by both at once.  Cycles per step from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/feedrate.hip -o gan-aug-pfa_amd/build/feedrate && gan-aug-pfa_amd/build/feedrate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int SLOT = 48 * 1024;      // one ring slot: 16 KiB "weights" (register path or DMA) + up to 32 KiB of DMA pieces
constexpr int NSLOT = 3;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }

// LDS-DMA, buffer form: lane's 16 bytes from base + voff (VGPR) + soff (SGPR) to lds + lane * 16.  (A plain function: inside the
// kernel template's lambdas the builtin makes hipcc drop the kernel's host stub without a diagnostic.)
__device__ __forceinline__ void buf_dma16(const void* base, void* lds, unsigned voff, unsigned soff) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(lds), 16, voff, soff, 0, 0);
}

// NDMA: DMA pieces per step and workgroup (0..32; waves take them round robin).  NREG: global_load_dwordx4 per thread and
// step on the register path (0, 2, 4: 512 threads x 16 B = 8 KiB each).  MFMA: run the 16 reads + 16 MFMAs per wave.
template <int NDMA, int NREG, int MFMA, int BUF = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void k_feed(const char* tab, int tab_rows, int pitch, int steps,
                                                                                     unsigned long long* out, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)LDS_PTR(smem);
    static_assert(NDMA % 8 == 0 && NDMA * 1024 + NREG * 8192 <= SLOT, "every wave issues the same number of pieces; the slot holds both regions");
    constexpr int PW = NDMA / 8;                           // DMA pieces of each wave per step
    // per-lane source: piece = 8 rows x 128 B
    const int prow = lane >> 3, pchunk = lane & 7;
    const int rrow = tid >> 3, rchunk = tid & 7;           // register path: 64 rows x 128 B per load
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    bf16x8 fa[8], fb[8];
    for (int i = 0; i < 8; ++i) { fa[i] = bf16x8{}; fb[i] = bf16x8{}; }
    u32x4 rg[NREG > 0 ? NREG : 1];
    for (int i = 0; i < (NREG > 0 ? NREG : 1); ++i) rg[i] = u32x4{0, 0, 0, 0};
    const int row0 = (blockIdx.x * 8) % tab_rows;
    unsigned bvoff[PW > 0 ? PW : 1];
    for (int j = 0; j < PW; ++j) bvoff[j] = (unsigned)(((row0 + (j * 8 + wave) * 8 + prow) % tab_rows) * pitch + pchunk * 16);
    auto issue_dma = [&](int s) {
        char* dst = smem + (s % NSLOT) * SLOT + NREG * 8192;
        const unsigned kofs = (unsigned)((s * 128) % pitch);
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const int piece = j * 8 + wave;
            if constexpr (BUF != 0) {        // buffer form: per-lane offset fixed for the whole loop, the step's offset in an SGPR
                buf_dma16(tab, dst + (piece * 64) * 16, bvoff[j], kofs);
            } else {
                const int r = (row0 + piece * 8 + prow) % tab_rows;
                const char* gp = tab + (size_t)r * pitch + kofs + pchunk * 16;
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gp), LDS_PTR(dst + (piece * 64) * 16), 16, 0, 0);
            }
        }
    };
    auto issue_reg = [&](int s) {
        const unsigned kofs = (unsigned)((s * 128 + 64 * 128) % pitch);
#pragma unroll
        for (int i = 0; i < NREG; ++i) {
            const int r = (row0 + 37 + i * 64 + rrow) % tab_rows;
            const char* gp = tab + (size_t)r * pitch + kofs + rchunk * 16;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rg[i]) : "v"(gp) : "memory");
        }
    };
    auto write_reg = [&](int s) {          // data of step s (loaded a step earlier) -> slot s % NSLOT
        const unsigned dst = lds0 + (s % NSLOT) * SLOT;
#pragma unroll
        for (int i = 0; i < NREG; ++i)
            asm volatile("ds_write_b128 %0, %1" ::"v"(dst + (i * 512 + tid) * 16), "v"(rg[i]) : "memory");
    };
    // prologue
    issue_dma(0); issue_dma(1);          // (no register load here: an asm load whose result is never read lands in registers the compiler has reused)
    wait_vm<0>();
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; ++s) {
        // register loads issued in step s-1 (and every DMA older than them) have landed
        if (NREG > 0) {
            if (s > 0) {
#pragma unroll
                for (int i = 0; i < NREG; ++i) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(rg[i]) : "n"(PW));
                write_reg(s + 1);
            }
        } else wait_vm<PW>();
        wait_lgkm<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (NREG > 0) issue_reg(s + 2);
        if (NDMA > 0) issue_dma(s + 2);
        if (MFMA) {
            const unsigned base = lds0 + (s % NSLOT) * SLOT + lane * 16;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(fa[i]) : "v"(base), "n"(i * 1024));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(fb[i]) : "v"(base), "n"(8192 + i * 1024));
            }
            wait_lgkm<0>();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[k * 2 + a], fb[k * 2 + b], acc[a][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    wait_vm<0>();
#pragma unroll
    for (int i = 0; i < (NREG > 0 ? NREG : 1); ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rg[i]));      // the last loads' results are read below
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[blockIdx.x] = t1 - t0;
    float x = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) x += acc[a][b][0] + acc[a][b][7];
    for (int i = 0; i < (NREG > 0 ? NREG : 1); ++i) x += (float)rg[i][0];
    sink[blockIdx.x * 512 + tid] = x + smem[tid * 4];
}

template <int NDMA, int NREG, int MFMA, int BUF = 0>
static void run(const char* tab, int tab_rows, int pitch, unsigned long long* out, float* sink, int nwg) {
    const int steps = 2048, lds = NSLOT * SLOT;
    hipFuncSetAttribute((const void*)k_feed<NDMA, NREG, MFMA, BUF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    std::vector<unsigned long long> h(nwg);
    double best = 1e30, best_ms = 0;
    for (int rep = 0; rep < 4; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_feed<NDMA, NREG, MFMA, BUF>), dim3(nwg), dim3(512), lds, 0, tab, tab_rows, pitch, steps, out, sink);
        hipEventRecord(e1);
        hipError_t err = hipEventSynchronize(e1);
        if (err != hipSuccess || (err = hipGetLastError()) != hipSuccess) { printf("dma %d reg %d mfma %d: %s\n", NDMA, NREG, MFMA, hipGetErrorString(err)); fflush(stdout); return; }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), out, nwg * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double med = (double)h[nwg / 2] / steps;
        if (med < best) { best = med; best_ms = ms; }
    }
    const double bytes = NDMA * 1024.0 + NREG * 8192.0;
    printf("dma %2d KiB%s reg %2d KiB  mfma %d : %7.1f cycles/step  %5.1f B/clk/CU  (launch %.3f ms -> %.2f GHz implied, %.1f GB/s/CU)\n", NDMA, BUF ? " buf" : "    ", NREG * 8, MFMA, best,
           bytes / best, best_ms, best * steps / (best_ms * 1e6), bytes * steps / (best_ms * 1e-3) / 1e9);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int nwg = argc > 1 ? atoi(argv[1]) : 256;
    const int pitch = 8192, tab_rows = argc > 2 ? atoi(argv[2]) : 256;      // 2 MiB table: resident in every XCD's L2
    char* tab; unsigned long long* out; float* sink;
    hipMalloc(&tab, (size_t)tab_rows * pitch); hipMalloc(&out, nwg * 8); hipMalloc(&sink, (size_t)nwg * 512 * 4);
    std::vector<unsigned short> h((size_t)tab_rows * pitch / 2);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3c00 + ((x >> 9) & 0x3ff) + ((x >> 3) & 0x8000)); }   // bf16 in +-[0.0078, 0.0156): random mantissas
    hipMemcpy(tab, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    printf("%d workgroups, table %d rows x %d B\n", nwg, tab_rows, pitch);
    run<0, 0, 1>(tab, tab_rows, pitch, out, sink, nwg);
    run<8, 0, 0>(tab, tab_rows, pitch, out, sink, nwg);
    run<16, 0, 0>(tab, tab_rows, pitch, out, sink, nwg);
    run<24, 0, 0>(tab, tab_rows, pitch, out, sink, nwg);
    run<32, 0, 0>(tab, tab_rows, pitch, out, sink, nwg);
    run<0, 2, 0>(tab, tab_rows, pitch, out, sink, nwg);
    run<0, 4, 0>(tab, tab_rows, pitch, out, sink, nwg);
    run<8, 2, 0>(tab, tab_rows, pitch, out, sink, nwg);
    run<16, 2, 0>(tab, tab_rows, pitch, out, sink, nwg);
    run<16, 4, 0>(tab, tab_rows, pitch, out, sink, nwg);
    run<32, 2, 0>(tab, tab_rows, pitch, out, sink, nwg);
    run<16, 0, 1>(tab, tab_rows, pitch, out, sink, nwg);
    run<24, 0, 1>(tab, tab_rows, pitch, out, sink, nwg);
    run<32, 0, 1>(tab, tab_rows, pitch, out, sink, nwg);
    run<0, 2, 1>(tab, tab_rows, pitch, out, sink, nwg);
    run<0, 4, 1>(tab, tab_rows, pitch, out, sink, nwg);
    run<8, 2, 1>(tab, tab_rows, pitch, out, sink, nwg);
    run<16, 2, 1>(tab, tab_rows, pitch, out, sink, nwg);
    run<16, 4, 1>(tab, tab_rows, pitch, out, sink, nwg);
    run<16, 0, 0, 1>(tab, tab_rows, pitch, out, sink, nwg);
    run<32, 0, 0, 1>(tab, tab_rows, pitch, out, sink, nwg);
    run<16, 0, 1, 1>(tab, tab_rows, pitch, out, sink, nwg);
    run<32, 0, 1, 1>(tab, tab_rows, pitch, out, sink, nwg);
    return 0;
}
