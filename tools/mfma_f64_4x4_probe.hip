// v_mfma_f64_4x4x4_4b_f64 on gfx950 (four independent 4 x 4 x 4 products per instruction, one f64 of A, B, C / D per lane):
// (a) which lanes hold which element — found by one-hot launches, no assumption —, (b) how it accumulates (k ascending
// fma chain?) and (c) its issue rate alone and beside fp64 vector work (tools/mfma_f64_probe.hip measured 64.2 shader
// cycles for the 16 x 16 x 4 form: the same multiply-adds per cycle as the vector pipe).
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_f64_4x4_probe.hip -o /tmp/mfma4_probe && /tmp/mfma4_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

__global__ void mfma_once(const double *A, const double *B, const double *C, double *D)
{
    const int l = threadIdx.x;
    D[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], C[l], 0, 0, 0);
}

__global__ void mfma_rate(double *out, int iters, int valu_per_mfma)
{
    const int l = threadIdx.x & 63;
    double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    double v0 = a, v1 = b, v2 = a + b, v3 = a - b;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, a, c1, 0, 0, 0);
        for (int q = 0; q < valu_per_mfma; q += 4) {
            v0 = fma(v0, 0.999999, 1e-9); v1 = fma(v1, 0.999998, 1e-9); v2 = fma(v2, 0.999997, 1e-9); v3 = fma(v3, 0.999996, 1e-9);
        }
        c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, b, c3, 0, 0, 0);
        for (int q = 0; q < valu_per_mfma; q += 4) {
            v0 = fma(v0, 0.999999, 1e-9); v1 = fma(v1, 0.999998, 1e-9); v2 = fma(v2, 0.999997, 1e-9); v3 = fma(v3, 0.999996, 1e-9);
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + v0 + v1 + v2 + v3;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0) / (4.0 * iters);
}

// the same with ONE dependent chain (what a bank's accumulation is)
__global__ void mfma_rate_dep(double *out, int iters)
{
    const int l = threadIdx.x & 63;
    double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3, c0 = 0;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, a, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, a, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, b, c0, 0, 0, 0);
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0) / (4.0 * iters);
}

int main()
{
    double hA[64], hB[64], hC[64], hD[64];
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC); hipMalloc(&dD, sizeof hD);
    // (a) layout: A one-hot at lane la, B[l] = 1 + l, C = 0  =>  D[l] = 1 + (the B lane it met), in the lanes of A's row
    int a_row[64], a_k[64], a_blk[64], b_col[64], b_k[64], b_blk[64];
    memset(a_row, -1, sizeof a_row); memset(b_col, -1, sizeof b_col);
    int pair_out[64][4], pair_b[64][4], npair[64];
    for (int la = 0; la < 64; ++la) {
        for (int q = 0; q < 64; ++q) { hA[q] = q == la ? 1.0 : 0.0; hB[q] = 1.0 + q; hC[q] = 0.0; }
        hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice); hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mfma_once, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        npair[la] = 0;
        for (int l = 0; l < 64; ++l)
            if (hD[l] != 0.0 && npair[la] < 4) { pair_out[la][npair[la]] = l; pair_b[la][npair[la]] = (int)hD[l] - 1; ++npair[la]; }
    }
    printf("{\"layout_one_hot\": [");
    for (int la = 0; la < 64; ++la) {
        printf("%s[%d", la ? ", " : "", la);
        for (int q = 0; q < npair[la]; ++q) printf(", [%d, %d]", pair_out[la][q], pair_b[la][q]);
        printf("]");
    }
    printf("]");
    // hypothesis to check on random data: block = l >> 4; A[i = l & 3][k = (l >> 2) & 3]; B[k = (l >> 2) & 3][j = l & 3]; D[i = (l >> 2) & 3][j = l & 3]
    srand48(7);
    long long n_fwd = 0, n_rev = 0, n_tot = 0, n_bad = 0;
    for (int trial = 0; trial < 4000; ++trial) {
        for (int q = 0; q < 64; ++q) {
            hA[q] = (drand48() - 0.5) * exp2((double)(lrand48() % 40 - 20)); hB[q] = (drand48() - 0.5) * exp2((double)(lrand48() % 40 - 20));
            hC[q] = (drand48() - 0.5) * exp2((double)(lrand48() % 40 - 20));
        }
        hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice); hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mfma_once, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l) {
            const int blk = l >> 4, i = (l >> 2) & 3, j = l & 3;
            double f = hC[l], r = hC[l], ex = hC[l];
            for (int k = 0; k < 4; ++k) f = fma(hA[16 * blk + 4 * k + i], hB[16 * blk + 4 * k + j], f);
            for (int k = 3; k >= 0; --k) r = fma(hA[16 * blk + 4 * k + i], hB[16 * blk + 4 * k + j], r);
            for (int k = 0; k < 4; ++k) ex += hA[16 * blk + 4 * k + i] * hB[16 * blk + 4 * k + j];
            ++n_tot;
            n_fwd += memcmp(&hD[l], &f, 8) == 0;
            n_rev += memcmp(&hD[l], &r, 8) == 0;
            if (fabs(hD[l] - ex) > 1e-9 * (fabs(ex) + 1e-300) + 1e-280) ++n_bad;
        }
    }
    printf(", \"hypothesis\": \"blk = l >> 4, A[i = l & 3][k = (l >> 2) & 3], B[k = (l >> 2) & 3][j = l & 3], D[i = (l >> 2) & 3][j = l & 3]\", \"elements\": %lld, "
           "\"bitwise_equal_fma_chain_k_ascending\": %lld, \"bitwise_equal_fma_chain_k_descending\": %lld, \"layout_mismatch\": %lld",
           n_tot, n_fwd, n_rev, n_bad);
    double *dout; hipMalloc(&dout, (size_t)1024 * 1024 * 8);
    for (int waves = 1; waves <= 4; waves *= 2)
        for (int v = 0; v <= 32; v += 8) {
            hipLaunchKernelGGL(mfma_rate, dim3(1024), dim3(64 * waves * 4), 0, 0, dout, 2000, v);
            hipDeviceSynchronize();
            hipLaunchKernelGGL(mfma_rate, dim3(1024), dim3(64 * waves * 4), 0, 0, dout, 2000, v);
            double cyc; hipMemcpy(&cyc, dout, 8, hipMemcpyDeviceToHost);
            printf(", \"cycles_per_mfma_w%d_valu%d\": %.1f", waves, v, cyc);
        }
    for (int waves = 1; waves <= 4; waves *= 2) {
        hipLaunchKernelGGL(mfma_rate_dep, dim3(1024), dim3(64 * waves * 4), 0, 0, dout, 2000);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(mfma_rate_dep, dim3(1024), dim3(64 * waves * 4), 0, 0, dout, 2000);
        double cyc; hipMemcpy(&cyc, dout, 8, hipMemcpyDeviceToHost);
        printf(", \"cycles_per_dependent_mfma_w%d\": %.1f", waves, cyc);
    }
    printf("}\n");
    return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
