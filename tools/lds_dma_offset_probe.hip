// Does the immediate offset of global_load_lds_dwordx4 move the LDS destination as well as the global source?
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_dma_offset_probe tools/lds_dma_offset_probe.hip && /tmp/lds_dma_offset_probe
// Prints, for offset:1024 with M0 = 0: which global bytes arrived, and at which LDS address.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void probe(const uint32_t *src, uint32_t *out)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int lane = threadIdx.x;
    for (int k = lane; k < 2048; k += 64) lds[k] = 0xDEAD0000u + k;
    __builtin_amdgcn_s_waitcnt(0);
    unsigned voff = lane * 16;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, 0\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "s_waitcnt vmcnt(0)\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(src) : "memory");
    __builtin_amdgcn_s_waitcnt(0);
    for (int k = lane; k < 2048; k += 64) out[k] = lds[k];
}

int main()
{
    uint32_t *d_src, *d_out, h_src[4096], h_out[2048];
    for (int i = 0; i < 4096; ++i) h_src[i] = i;      // word i at byte 4 i
    hipMalloc(&d_src, sizeof h_src); hipMalloc(&d_out, sizeof h_out);
    hipMemcpy(d_src, h_src, sizeof h_src, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 8192, 0, d_src, d_out);
    hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost);
    int first = -1;
    for (int k = 0; k < 2048; ++k) if (h_out[k] != 0xDEAD0000u + k) { first = k; break; }
    if (first < 0) { printf("{\"landed\": false}\n"); return 1; }
    printf("{\"lds_byte_of_first_word\": %d, \"global_word_there\": %u, \"global_byte_there\": %u}\n", 4 * first, h_out[first], 4 * h_out[first]);
    return 0;
}
