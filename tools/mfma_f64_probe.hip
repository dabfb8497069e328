// v_mfma_f64_16x16x4_f64 on gfx950: (a) operand / result layout, (b) HOW it accumulates — is
// D[i][j] = C[i][j] + sum_k A[i][k] B[k][j] the chain fma(A[i][3],B[3][j], fma(A[i][2],B[2][j], fma(.., fma(A[i][0],B[0][j], C))))
// (k ascending, one rounding per product), the reverse, or something wider? — and (c) its issue rate beside VALU work.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_f64_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void mfma_once(const double *A, const double *B, const double *C, double *D)
{
    // A: [16][4] row-major, B: [4][16], C/D: [16][16]
    const int l = threadIdx.x;
    const double a = A[(l & 15) * 4 + (l >> 4)];        // A[i = l & 15][k = l >> 4]
    const double b = B[(l >> 4) * 16 + (l & 15)];       // B[k = l >> 4][j = l & 15]
    d4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[((l >> 4) + 4 * r) * 16 + (l & 15)];   // row = (l >> 4) + 4 r, col = l & 15
    const d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = d[r];
}

__global__ void mfma_rate(double *out, int iters, int valu_per_mfma)
{
    const int l = threadIdx.x & 63;
    double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    d4 c0 = {0, 0, 0, 0}, c1 = c0;
    double v0 = a, v1 = b, v2 = a + b, v3 = a - b;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        for (int q = 0; q < valu_per_mfma; q += 4) {
            v0 = fma(v0, 0.999999, 1e-9); v1 = fma(v1, 0.999998, 1e-9); v2 = fma(v2, 0.999997, 1e-9); v3 = fma(v3, 0.999996, 1e-9);
        }
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
        for (int q = 0; q < valu_per_mfma; q += 4) {
            v0 = fma(v0, 0.999999, 1e-9); v1 = fma(v1, 0.999998, 1e-9); v2 = fma(v2, 0.999997, 1e-9); v3 = fma(v3, 0.999996, 1e-9);
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + v0 + v1 + v2 + v3;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0) / (2.0 * iters);
}

int main()
{
    double hA[64], hB[64], hC[256], hD[256];
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC); hipMalloc(&dD, sizeof hD);
    srand48(7);
    long long n_fwd = 0, n_rev = 0, n_tot = 0, n_layout_bad = 0;
    for (int trial = 0; trial < 4000; ++trial) {
        for (int q = 0; q < 64; ++q) { hA[q] = (drand48() - 0.5) * exp2((double)(lrand48() % 40 - 20)); hB[q] = (drand48() - 0.5) * exp2((double)(lrand48() % 40 - 20)); }
        for (int q = 0; q < 256; ++q) hC[q] = (drand48() - 0.5) * exp2((double)(lrand48() % 40 - 20));
        hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice); hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mfma_once, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double f = hC[i * 16 + j], r = hC[i * 16 + j], ex = hC[i * 16 + j];
                for (int k = 0; k < 4; ++k) f = fma(hA[i * 4 + k], hB[k * 16 + j], f);
                for (int k = 3; k >= 0; --k) r = fma(hA[i * 4 + k], hB[k * 16 + j], r);
                for (int k = 0; k < 4; ++k) ex += hA[i * 4 + k] * hB[k * 16 + j];
                const double d = hD[i * 16 + j];
                ++n_tot;
                n_fwd += memcmp(&d, &f, 8) == 0;
                n_rev += memcmp(&d, &r, 8) == 0;
                if (fabs(d - ex) > 1e-9 * (fabs(ex) + 1e-300) + 1e-280) ++n_layout_bad;
            }
    }
    printf("{\"elements\": %lld, \"bitwise_equal_fma_chain_k_ascending\": %lld, \"bitwise_equal_fma_chain_k_descending\": %lld, \"layout_mismatch\": %lld",
           n_tot, n_fwd, n_rev, n_layout_bad);
    double *dout; hipMalloc(&dout, (size_t)1024 * 1024 * 8);      // 1024 blocks x up to 1024 threads
    for (int waves = 1; waves <= 4; waves *= 2)
        for (int v = 0; v <= 32; v += 8) {
            hipLaunchKernelGGL(mfma_rate, dim3(1024), dim3(64 * waves * 4), 0, 0, dout, 2000, v);
            hipDeviceSynchronize();
            hipLaunchKernelGGL(mfma_rate, dim3(1024), dim3(64 * waves * 4), 0, 0, dout, 2000, v);
            double cyc; hipMemcpy(&cyc, dout, 8, hipMemcpyDeviceToHost);
            printf(", \"cycles_per_mfma_w%d_valu%d\": %.1f", waves, v, cyc);
        }
    printf("}\n");
    return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
