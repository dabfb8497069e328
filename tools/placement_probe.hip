// Diagnostic only (not part of libwfhip.so): WHERE the workgroups of a small grid land.  One-wave workgroups with the lane
// detectors' footprint (64 threads, 24.6 KB of LDS) spin for ~100 us and record HW_ID / XCC_ID; the host prints how many
// distinct CUs and SIMDs a grid of G workgroups occupied and the largest number of waves that shared one SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/placement_probe.hip -o /tmp/placement_probe && /tmp/placement_probe 610 814 977 1221
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <map>
#include <vector>

__global__ __launch_bounds__(64) void probe(double *sink, uint32_t *where, int iters)
{
    extern __shared__ double lds[];
    double a = threadIdx.x * 1e-3 + 1.0, b = 1.0000001, c = 0.9999999, d = a + 0.5;
    lds[threadIdx.x] = a;
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    for (int i = 0; i < iters; ++i) {
        a = fma(a, b, c); d = fma(d, c, b); b = fma(b, c, 1e-9); c = fma(c, 0.9999999, 1e-7);
    }
    if (threadIdx.x == 0) {
        where[2 * blockIdx.x] = hw;
        where[2 * blockIdx.x + 1] = xcc;
    }
    if (a + d + b + c + lds[63 - threadIdx.x] == 12345.678) sink[0] = a;
}

int main(int argc, char **argv)
{
    double *sink;
    uint32_t *where;
    hipMalloc(&sink, 8);
    hipMalloc(&where, 8 * 65536);
    for (int k = 1; k < argc; ++k) {
        const int g = atoi(argv[k]);
        for (int lds_kb : {24, 1}) {
            hipMemset(where, 0xff, 8 * 65536);
            hipLaunchKernelGGL(probe, dim3(g), dim3(64), lds_kb * 1024 + 640, 0, sink, where, 15000);
            hipDeviceSynchronize();
            std::vector<uint32_t> h(2 * g);
            hipMemcpy(h.data(), where, 8 * g, hipMemcpyDeviceToHost);
            std::map<uint32_t, int> per_simd, per_cu;
            for (int i = 0; i < g; ++i) {
                const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
                const uint32_t simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
                const uint32_t cu_key = (xcc << 16) | (se << 8) | (sh << 4) | cu;
                per_cu[cu_key]++;
                per_simd[(cu_key << 2) | simd]++;
            }
            int max_simd = 0, max_cu = 0;
            std::map<int, int> hist;
            for (auto &kv : per_simd) { if (kv.second > max_simd) max_simd = kv.second; hist[kv.second]++; }
            for (auto &kv : per_cu) if (kv.second > max_cu) max_cu = kv.second;
            printf("grid %5d, %2d KB LDS: %3zu CUs, %4zu SIMDs in use; most waves on one CU %d, on one SIMD %d; SIMDs by wave count:", g, lds_kb, per_cu.size(),
                   per_simd.size(), max_cu, max_simd);
            for (auto &kv : hist) printf(" %dx%d", kv.first, kv.second);
            printf("\n");
        }
    }
    return 0;
}
