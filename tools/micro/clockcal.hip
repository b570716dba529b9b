// What does s_memtime count?  Per workgroup: ticks of s_memtime and of s_memrealtime (100 MHz) across (a) a chain of
// dependent v_fma (VALU only) and (b) a chain of MFMAs, plus the HIP-event time of each launch.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/clockcal.hip -o /tmp/clockcal && /tmp/clockcal
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void k(int mode, int iters, unsigned long long* out, float* sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float x = threadIdx.x * 1e-3f;
    f32x16 acc = {};
    bf16x8 a = {}, b = {};
    if (mode == 0) {
        for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f;
    } else {
        for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = r1 - r0; }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = x + acc[0];
}
int main() {
    const int blocks = 2048, threads = 512;
    unsigned long long* out; float* sink;
    hipMalloc(&out, blocks * 16); hipMalloc(&sink, blocks * threads * 4);
    unsigned long long* h = (unsigned long long*)malloc(blocks * 16);
    for (int mode = 0; mode < 2; ++mode) {
        const int iters = mode == 0 ? 200000 : 20000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, mode, iters, out, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, out, blocks * 16, hipMemcpyDeviceToHost);
            double t = 0, r = 0;
            for (int i = 0; i < blocks; ++i) { t += h[2 * i]; r += h[2 * i + 1]; }
            t /= blocks; r /= blocks;
            printf("mode %s: %d iterations per wave: s_memtime %.0f ticks, s_memrealtime %.0f ticks (= %.1f us at 100 MHz) -> s_memtime runs at %.1f MHz; "
                   "%.3f ticks per iteration; launch %.1f us\n", mode ? "mfma32x32x16" : "v_fma chain", iters, t, r, r / 100.0, t / (r / 100.0), t / iters, ms * 1e3);
        }
    }
    return 0;
}
