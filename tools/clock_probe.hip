// SUPERSEDED for instruction costs by tools/valu_probe.hip: the compiler adds three v_mov_b64 per iteration to
// the fp64 loop below (7 instructions, not 4), so the 7.7 "cycles per fp64 FMA" this printed were wrong — the
// real figure is 4.2.  Kept for the clock reading only.
// Diagnostic only (not part of libwfhip.so): the shader clock the chip sustains under a dense fp64
// VALU load, measured in-kernel as delta s_memtime / delta s_memrealtime x 100 MHz
// (MI355X_MICROARCH.md, "DVFS give-back" item 6).  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <algorithm>
#include <vector>

__global__ void probe(double *sink, uint64_t *stamps, int iters, int mode)
{
    double a = threadIdx.x * 1e-3 + 1.0, b = 1.0000001, c = 0.9999999, d = a + 0.5;
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (mode == 0) {
        for (int i = 0; i < iters; ++i) {       // 4 independent fp64 FMA chains
            a = fma(a, b, c); d = fma(d, c, b); b = fma(b, c, 1e-9); c = fma(c, 0.9999999, 1e-7);
        }
    } else if (mode == 2) {
        // 32 x 32 -> 64-bit multiply-add (v_mad_u64_u32: the Philox round's instruction)
        uint64_t x0 = threadIdx.x + 1, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7;
        for (int i = 0; i < iters; ++i) {
            x0 = (uint64_t)(uint32_t)x0 * 0xD2511F53u + (x1 >> 32); x1 = (uint64_t)(uint32_t)x1 * 0xCD9E8D57u + (x2 >> 32);
            x2 = (uint64_t)(uint32_t)x2 * 0xD2511F53u + (x3 >> 32); x3 = (uint64_t)(uint32_t)x3 * 0xCD9E8D57u + (x0 >> 32);
        }
        a = (double)(x0 ^ x1 ^ x2 ^ x3);
    } else if (mode == 3) {
        // 32-bit integer / logic ops (v_xor / v_add_u32)
        uint32_t y0 = threadIdx.x + 1, y1 = y0 * 3, y2 = y0 * 5, y3 = y0 * 7;
        for (int i = 0; i < iters; ++i) {
            y0 = (y0 ^ y1) + 0x9E3779B9u; y1 = (y1 ^ y2) + 0xBB67AE85u; y2 = (y2 ^ y3) + 0x9E3779B9u; y3 = (y3 ^ y0) + 0xBB67AE85u;
        }
        a = (double)(y0 ^ y1 ^ y2 ^ y3);
    } else {
        float fa = (float)a, fb = 1.0000001f, fc = 0.9999999f, fd = fa + 0.5f;
        for (int i = 0; i < iters; ++i) {
            fa = fmaf(fa, fb, fc); fd = fmaf(fd, fc, fb); fb = fmaf(fb, fc, 1e-9f); fc = fmaf(fc, 0.9999999f, 1e-7f);
        }
        a = fa + fd + fb + fc;
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
    if (a + b + c + d == 1.2345) sink[0] = a;
}

int main()
{
    const int blocks = 256 * 8, iters = 1 << 15;
    double *sink; uint64_t *stamps;
    hipMalloc(&sink, 8); hipMalloc(&stamps, blocks * 16);
    std::vector<uint64_t> h(2 * blocks);
    for (int mode = 0; mode < 4; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        double last = 0;
        for (int rep = 0; rep < 400; ++rep) {   // ~2 s of back-to-back launches, stamp the last
            hipEventRecord(e0);
            hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, sink, stamps, iters, mode);
            hipEventRecord(e1);
        }
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1); last = ms;
        hipMemcpy(h.data(), stamps, blocks * 16, hipMemcpyDeviceToHost);
        std::vector<double> mhz;
        for (int b = 0; b < blocks; ++b) if (h[2 * b + 1]) mhz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 100.0);
        std::sort(mhz.begin(), mhz.end());
        // 4 FMA per iteration per lane: wave-instructions = blocks*4 waves*iters*4
        const double winst = (double)blocks * 4 * iters * (mode == 3 ? 8 : 4);   // mode 3: xor + add per statement
        const double clk = mhz[mhz.size() / 2] * 1e6;
        printf("{\"load\": \"%s\", \"shader_clock_mhz_median\": %.0f, \"min\": %.0f, \"max\": %.0f, \"kernel_ms\": %.3f, "
               "\"cycles_per_wave_instruction_per_simd\": %.2f}\n", mode == 0 ? "fp64 fma, 8 waves per SIMD" : mode == 1 ? "fp32 fma, 8 waves per SIMD" : mode == 2 ? "v_mad_u64_u32, 8 waves per SIMD" : "32-bit xor/add, 8 waves per SIMD",
               mhz[mhz.size() / 2], mhz.front(), mhz.back(), last, (last * 1e-3 * clk) / (winst / 1024.0));
    }
    return 0;
}
