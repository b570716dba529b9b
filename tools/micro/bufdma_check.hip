// Semantics of the LDS-DMA buffer form on gfx950 (buffer_load_dwordx4 v_off, s[rsrc], s_off offen lds), which the patch
// kernels use so that a DMA piece costs an SGPR offset instead of per-lane 64-bit address arithmetic:
//  (1) a lane whose VGPR offset is >= num_records WRITES ZEROS to its 16 LDS bytes (it does not leave them untouched);
//  (2) the SGPR offset takes no part in the range check (offset = VGPR offset + instruction offset only);
//  (3) the LDS destination is M0 base + lane * 16 as with global_load_lds.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/bufdma_check.hip -o gan-aug-pfa_amd/build/bufdma_check && gan-aug-pfa_amd/build/bufdma_check
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
__global__ void k(const char* src, unsigned num_records, unsigned soff, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* w = (unsigned*)smem;
    for (int i = threadIdx.x; i < 64 * 4 * 2; i += 64) w[i] = 0xdeadbeefu;          // stale pattern in both slots
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, num_records, 0x00020000);
    unsigned voff = threadIdx.x * 16;
    if ((threadIdx.x & 3) == 1) voff = 0x80000000u;            // far out of range
    if ((threadIdx.x & 3) == 2) voff = num_records;            // first byte out of range
    if ((threadIdx.x & 3) == 3) voff = num_records - 8;        // straddles the end
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(smem + 1024), 16, voff, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 4 * 2; i += 64) out[i] = w[i];
}
int main() {
    const unsigned bytes = 1 << 20;
    char* src; unsigned* out;
    hipMalloc(&src, 2 * bytes); hipMalloc(&out, 64 * 4 * 2 * 4);
    std::vector<unsigned> h(2 * bytes / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x10000000u + (unsigned)i;       // word i holds its own index
    hipMemcpy(src, h.data(), 2 * bytes, hipMemcpyHostToDevice);
    std::vector<unsigned> o(64 * 4 * 2);
    int bad = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const unsigned num_records = 4096, soff = pass ? bytes : 0;        // pass 1: SGPR offset far beyond num_records
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, src, num_records, soff, out);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost);
        for (int i = 0; i < 256; ++i) if (o[i] != 0xdeadbeefu) { ++bad; if (bad < 5) printf("pass %d: slot 0 word %d touched: %08x\n", pass, i, o[i]); }
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 4; ++e) {
                const unsigned got = o[256 + l * 4 + e];
                unsigned want;
                if ((l & 3) == 0) want = 0x10000000u + (soff + l * 16) / 4 + e;
                else if ((l & 3) == 3) want = 0xffffffffu;      // straddling: reported, not judged
                else want = 0;
                if (want == 0xffffffffu) { if (l == 3) printf("pass %d: straddling lane word %d = %08x\n", pass, e, got); continue; }
                if (got != want) { ++bad; if (bad < 12) printf("pass %d lane %d word %d: got %08x want %08x\n", pass, l, e, got, want); }
            }
        printf("pass %d (soffset %u, num_records %u): %s\n", pass, soff, num_records, bad ? "MISMATCH" : "in-range lanes loaded, out-of-range lanes wrote zeros");
    }
    return bad ? 1 : 0;
}
