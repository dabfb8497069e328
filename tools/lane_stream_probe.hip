// Diagnostic only (not part of libwfhip.so): what the LANE detectors' row-fetch pattern can draw from HBM, without the detector.
// One-wave workgroups, a ring of 3 slots of 8 KB filled by LDS-DMA (global_load_lds_dwordx4, as wf_cpm_lanes.hip), a counted
// s_waitcnt per slot, `pace` dependent FMAs per slot in place of the detector's arithmetic.  Two layouts of the same rows:
//   A  chunk-major (what the front ends write today): chunk c's rows are one contiguous stream; a wave's slot = 64 x 128 B,
//      CH x rb bytes apart (rb = bytes per call: 64 PCM/FM, 256 ARTM)
//   B  wave-interleaved: the 64 chunks of a wave interleaved in units of 256 B; a wave's slot pair = 16 KB contiguous
// Prints GB/s for grids of 489 .. 1221 waves.
//   hipcc --offload-arch=gfx950 -O3 tools/lane_stream_probe.hip -o /tmp/lane_stream_probe && /tmp/lane_stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

#define SLOT 8192
#define BIAS 4096

__device__ __forceinline__ void dma4(unsigned v0, unsigned v1, unsigned v2, unsigned v3, const void *sb, unsigned lds)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %6\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %5 nt\n\t"
                 "global_load_lds_dwordx4 %2, %5 offset:1024 nt\n\t"
                 "global_load_lds_dwordx4 %3, %5 offset:2048 nt\n\t"
                 "global_load_lds_dwordx4 %4, %5 offset:3072 nt\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(sb), "s"(lds)
                 : "memory");
}

// nslots slots per wave; layout A: slot s of chunk c at c * stream_bytes + 128 s; layout B: slot s of the wave at 8192 s (+ the
// 128-B lines of a chunk 256 B apart inside a 16 KB pair: the wave reads lines 2 c + (s & 1) of pair s >> 1)
template <int LAYOUT>
__global__ __launch_bounds__(64) void probe(const char *rows, double *sink, long stream_bytes, int nslots, int pace)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x, c8 = lane >> 3, j8 = lane & 7;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const char *wave_base = rows + (long)blockIdx.x * 64 * stream_bytes;
    unsigned voff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned c = 8 * i + c8;
        voff[i] = (LAYOUT == 0 ? c * (unsigned)stream_bytes : c * 256u) + j8 * 16u + BIAS - 1024u * (i & 3);
    }
    auto fetch = [&](int s, unsigned lds) {
        const int ss = s < nslots ? s : nslots - 1;
        const long off = LAYOUT == 0 ? (long)ss * 128 : (long)(ss >> 1) * 16384 + (ss & 1) * 128;
        const char *sb = wave_base + off - BIAS;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)sb), hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)sb >> 32));
        const char *sbu = (const char *)(((uintptr_t)hi << 32) | lo);
        dma4(voff[0], voff[1], voff[2], voff[3], sbu, lds);
        dma4(voff[4], voff[5], voff[6], voff[7], sbu, lds + 4096);
    };
    fetch(0, lds0);
    fetch(1, lds0 + SLOT);
    double a = lane * 1e-3, b = 1.0000001;
    unsigned p_fetch = 2, p_read = 0;
    for (int s = 0; s < nslots; ++s) {
        fetch(s + 2, lds0 + p_fetch * SLOT);
        p_fetch = p_fetch == 2 ? 0 : p_fetch + 1;
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        a += *reinterpret_cast<const double *>(smem + p_read * SLOT + lane * 128);
        p_read = p_read == 2 ? 0 : p_read + 1;
        for (int k = 0; k < pace; ++k) a = fma(a, b, 1e-9);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a == 12345.678) sink[0] = a;
}

int main()
{
    const size_t bytes = 4ull << 30;
    char *rows;
    double *sink;
    hipMalloc(&rows, bytes);
    hipMalloc(&sink, 8);
    hipMemset(rows, 0, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int rbs[2] = {64, 256}, calls[2] = {320, 240};
    for (int w = 0; w < 2; ++w) {
        const int rb = rbs[w];
        for (int waves : {489, 610, 814, 1024, 1221, 2048}) {
            for (int pace : {0, rb == 64 ? 150 : 300}) {
                for (int layout = 0; layout < 2; ++layout) {
                    const long stream = (long)calls[w] * rb;              // bytes a chunk's lane reads (own + warm-up)
                    const int nslots = (int)(stream / 128);
                    if ((size_t)waves * 64 * stream > bytes) continue;
                    float best = 1e9f;
                    for (int rep = 0; rep < 5; ++rep) {
                        hipEventRecord(e0);
                        if (layout == 0) hipLaunchKernelGGL(probe<0>, dim3(waves), dim3(64), 3 * SLOT, 0, rows, sink, stream, nslots, pace);
                        else hipLaunchKernelGGL(probe<1>, dim3(waves), dim3(64), 3 * SLOT, 0, rows, sink, stream, nslots, pace);
                        hipEventRecord(e1);
                        hipEventSynchronize(e1);
                        float ms;
                        hipEventElapsedTime(&ms, e0, e1);
                        if (ms < best) best = ms;
                    }
                    const double gb = (double)waves * 64 * stream / 1e9;
                    printf("rb %3d  waves %4d  pace %3d  layout %c: %.3f ms  %.2f GB  %.2f TB/s\n", rb, waves, pace, layout ? 'B' : 'A', best, gb, gb / best);
                }
            }
        }
    }
    return 0;
}
