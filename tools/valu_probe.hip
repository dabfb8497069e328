// Diagnostic only (not part of libwfhip.so): issue cost of single VALU instructions on gfx950, in
// shader cycles per wave64 instruction per SIMD.  Every loop body is inline asm, so the instruction
// count is exact (tools/clock_probe.hip let the compiler add v_mov_b64 copies to its fp64 loop, which
// made v_fma_f64 look like 7.7 cycles).  CHAINS independent dependency chains per wave, WAVES waves per
// SIMD; the shader clock is read in-kernel (s_memtime / s_memrealtime x 100 MHz).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <algorithm>
#include <vector>

#define REP8(X) X X X X X X X X

// Each BODY issues 8 instructions on 8 different destination registers (8 chains; CHAINS4 variants
// use 4 registers twice = 4 chains of dependent pairs).
#define DEF_KERNEL(NAME, DECL, BODY, SINK)                                                     \
    __global__ void NAME(double *sink, uint64_t *stamps, int iters)                            \
    {                                                                                          \
        DECL                                                                                   \
        const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
        for (int i = 0; i < iters; ++i) { REP8(BODY) }                                         \
        const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
        if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; } \
        if ((SINK) == 1.2345) sink[0] = 1.0;                                                   \
    }

#define D8 double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
           double b = 1.0000001, c = 1e-9;
#define U8 uint32_t a0 = threadIdx.x + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
           uint32_t b = 0xD2511F53u, c = 0x9E3779B9u;
#define Q8 uint64_t a0 = threadIdx.x + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
           uint32_t b = 0xD2511F53u; uint64_t c = 0x9E3779B97F4A7C15ull;
#define F8 float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
           float b = 1.0000001f, c = 1e-9f;

#define OP3(INS, R) asm volatile(INS " %0, %0, %1, %2" : "+v"(R) : "v"(b), "v"(c));
#define OP2(INS, R) asm volatile(INS " %0, %0, %1" : "+v"(R) : "v"(b));
#define ALL8(M, INS) M(INS, a0) M(INS, a1) M(INS, a2) M(INS, a3) M(INS, a4) M(INS, a5) M(INS, a6) M(INS, a7)
#define ALL4x2(M, INS) M(INS, a0) M(INS, a1) M(INS, a2) M(INS, a3) M(INS, a0) M(INS, a1) M(INS, a2) M(INS, a3)
#define ALL2x4(M, INS) M(INS, a0) M(INS, a1) M(INS, a0) M(INS, a1) M(INS, a0) M(INS, a1) M(INS, a0) M(INS, a1)
#define ALL1x8(M, INS) M(INS, a0) M(INS, a0) M(INS, a0) M(INS, a0) M(INS, a0) M(INS, a0) M(INS, a0) M(INS, a0)
#define SUMD (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
#define SUMU ((double)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7))

DEF_KERNEL(k_fma_f64, D8, ALL8(OP3, "v_fma_f64"), SUMD)
DEF_KERNEL(k_fma_f64_c4, D8, ALL4x2(OP3, "v_fma_f64"), SUMD)
DEF_KERNEL(k_fma_f64_c2, D8, ALL2x4(OP3, "v_fma_f64"), SUMD)
DEF_KERNEL(k_fma_f64_c1, D8, ALL1x8(OP3, "v_fma_f64"), SUMD)
DEF_KERNEL(k_mul_f64, D8, ALL8(OP2, "v_mul_f64"), SUMD)
DEF_KERNEL(k_add_f64, D8, ALL8(OP2, "v_add_f64"), SUMD)
DEF_KERNEL(k_min_f64, D8, ALL8(OP2, "v_min_f64"), SUMD)
DEF_KERNEL(k_fma_f32, F8, ALL8(OP3, "v_fma_f32"), SUMD)
DEF_KERNEL(k_fma_f32_c1, F8, ALL1x8(OP3, "v_fma_f32"), SUMD)
DEF_KERNEL(k_xor_b32, U8, ALL8(OP2, "v_xor_b32"), SUMU)
DEF_KERNEL(k_add_u32, U8, ALL8(OP2, "v_add_u32"), SUMU)
DEF_KERNEL(k_mul_lo_u32, U8, ALL8(OP2, "v_mul_lo_u32"), SUMU)
DEF_KERNEL(k_mul_hi_u32, U8, ALL8(OP2, "v_mul_hi_u32"), SUMU)
#define MAD64(INS, R) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(R) : "v"(b), "v"((uint32_t)c) : "vcc");
DEF_KERNEL(k_mad_u64_u32, Q8, ALL8(MAD64, ""), SUMU)
#define SHL64(INS, R) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(R));
DEF_KERNEL(k_lshl_b64, Q8, ALL8(SHL64, ""), SUMU)
#define MOV64(INS, R) asm volatile("v_mov_b64 %0, %1" : "+v"(R) : "v"(c));
DEF_KERNEL(k_mov_b64, Q8, ALL8(MOV64, ""), SUMU)
#define CND(INS, R) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(R) : "v"(b) : "vcc");
DEF_KERNEL(k_cndmask, U8, ALL8(CND, ""), SUMU)
#define CMPF64(INS, R) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(R), "v"(b) : "vcc");
DEF_KERNEL(k_cmp_f64, D8, ALL8(CMPF64, ""), SUMD)
#define RSQ(INS, R) asm volatile("v_rsq_f64 %0, %0" : "+v"(R));
DEF_KERNEL(k_rsq_f64, D8, ALL8(RSQ, ""), SUMD)
#define RCP(INS, R) asm volatile("v_rcp_f64 %0, %0" : "+v"(R));
DEF_KERNEL(k_rcp_f64, D8, ALL8(RCP, ""), SUMD)
#define CVT(INS, R) asm volatile("v_cvt_f64_u32 %0, %1" : "+v"(R) : "v"((uint32_t)threadIdx.x));
DEF_KERNEL(k_cvt_f64_u32, D8, ALL8(CVT, ""), SUMD)
#define PKF32(INS, R) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(R) : "v"(b));
DEF_KERNEL(k_pk_fma_f32, D8, ALL8(PKF32, ""), SUMD)
#define DPPMOV(INS, R) asm volatile("v_mov_b32_dpp %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(R));
DEF_KERNEL(k_mov_dpp, U8, ALL8(DPPMOV, ""), SUMU)

#define CNDS(INS, R) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(R) : "v"(b), "s"(m64));
#define UM8 U8 const uint64_t m64 = __builtin_amdgcn_ballot_w64((threadIdx.x & 3) == 1);
DEF_KERNEL(k_cndmask_sgpr, UM8, ALL8(CNDS, ""), SUMU)
#define CMPCND(INS, R) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(R) : "v"(b) : "vcc");
DEF_KERNEL(k_cmp_cndmask_pair, U8, ALL8(CMPCND, ""), SUMU)
#define CMPU(INS, R) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(R), "v"(b) : "vcc");
DEF_KERNEL(k_cmp_u32, U8, ALL8(CMPU, ""), SUMU)
#define OP3U(INS, R) asm volatile(INS " %0, %0, %1, %2" : "+v"(R) : "v"(b), "v"(c));
DEF_KERNEL(k_add3_u32, U8, ALL8(OP3U, "v_add3_u32"), SUMU)
DEF_KERNEL(k_xad_u32, U8, ALL8(OP3U, "v_xad_u32"), SUMU)
DEF_KERNEL(k_lshl_add_u32, U8, ALL8(OP3U, "v_lshl_add_u32"), SUMU)
DEF_KERNEL(k_alignbit, U8, ALL8(OP3U, "v_alignbit_b32"), SUMU)
DEF_KERNEL(k_bfe_u32, U8, ALL8(OP3U, "v_bfe_u32"), SUMU)
DEF_KERNEL(k_perm_b32, U8, ALL8(OP3U, "v_perm_b32"), SUMU)
DEF_KERNEL(k_mad_u32_u24, U8, ALL8(OP3U, "v_mad_u32_u24"), SUMU)
DEF_KERNEL(k_and_or_b32, U8, ALL8(OP3U, "v_and_or_b32"), SUMU)
#define BITOP3(INS, R) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(R) : "v"(b), "v"(c));
DEF_KERNEL(k_bitop3_xor3, U8, ALL8(BITOP3, ""), SUMU)
#define BITOP3S(INS, R) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(R) : "v"(b), "s"(c));
DEF_KERNEL(k_bitop3_xor3_sgpr, U8, ALL8(BITOP3S, ""), SUMU)
#define XOR2(INS, R) asm volatile("v_xor_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %2" : "+v"(R) : "v"(b), "v"(c));
DEF_KERNEL(k_xor_pair, U8, ALL8(XOR2, ""), SUMU)
#define ADD64(INS, R) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(R) : "v"(c));
DEF_KERNEL(k_lshl_add_u64, Q8, ALL8(ADD64, ""), SUMU)
#define ADDCO(INS, R) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(R) : "v"(b) : "vcc");
DEF_KERNEL(k_add_co_u32, U8, ALL8(ADDCO, ""), SUMU)
#define RDLANE(INS, R) { uint32_t t_; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(t_) : "v"(R)); asm volatile("" :: "s"(t_)); }
DEF_KERNEL(k_readlane, U8, ALL8(RDLANE, ""), SUMU)
#define LDEXP(INS, R) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(R) : "v"((int)threadIdx.x & 1));
DEF_KERNEL(k_ldexp_f64, D8, ALL8(LDEXP, ""), SUMD)
#define FREXPM(INS, R) asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(R));
DEF_KERNEL(k_frexp_mant_f64, D8, ALL8(FREXPM, ""), SUMD)
#define FRACT(INS, R) asm volatile("v_fract_f64 %0, %0" : "+v"(R));
DEF_KERNEL(k_fract_f64, D8, ALL8(FRACT, ""), SUMD)
#define CVTI(INS, R) { int t_; asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(t_) : "v"(R)); asm volatile("" :: "v"(t_)); }
DEF_KERNEL(k_cvt_i32_f64, D8, ALL8(CVTI, ""), SUMD)
#define SQRT(INS, R) asm volatile("v_sqrt_f64 %0, %0" : "+v"(R));
DEF_KERNEL(k_sqrt_f64, D8, ALL8(SQRT, ""), SUMD)
#define LOGF(INS, R) asm volatile("v_log_f32 %0, %0" : "+v"(R));
DEF_KERNEL(k_log_f32, F8, ALL8(LOGF, ""), SUMD)
#define FMADPP(INS, R) asm volatile("v_add_f64_dpp %0, %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(R) : "v"(b));
// LDS: one wave-private 16 B slot per lane (conflict-free), b128 / b64 reads and b64 writes; bpermute
#define LDSDECL U8 __shared__ __attribute__((aligned(16))) uint32_t lds[256 * 4 + 64]; const uint32_t la = threadIdx.x * 16; \
                lds[threadIdx.x * 4] = a0; lds[threadIdx.x * 4 + 1] = a1; lds[threadIdx.x * 4 + 2] = a2; lds[threadIdx.x * 4 + 3] = a3; __syncthreads();
#define LDR128(INS, R) { typedef uint32_t u4 __attribute__((ext_vector_type(4))); u4 t_; asm volatile("ds_read_b128 %0, %1" : "=v"(t_) : "v"(la)); asm volatile("" :: "v"(t_)); }
#define LDR64(INS, R) { uint64_t t_; asm volatile("ds_read_b64 %0, %1" : "=v"(t_) : "v"(la)); asm volatile("" :: "v"(t_)); }
#define LDR32(INS, R) { uint32_t t_; asm volatile("ds_read_b32 %0, %1" : "=v"(t_) : "v"(la)); asm volatile("" :: "v"(t_)); }
#define LDW64(INS, R) { asm volatile("ds_write_b64 %0, %1" : : "v"(la), "v"((uint64_t)R) : "memory"); }
#define BPERM(INS, R) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(R) : "v"(la & 252));
#define WAITL asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
DEF_KERNEL(k_ds_read_b128, LDSDECL, ALL8(LDR128, "") WAITL, SUMU)
DEF_KERNEL(k_ds_read_b64, LDSDECL, ALL8(LDR64, "") WAITL, SUMU)
DEF_KERNEL(k_ds_read_b32, LDSDECL, ALL8(LDR32, "") WAITL, SUMU)
DEF_KERNEL(k_ds_write_b64, LDSDECL, ALL8(LDW64, "") WAITL, SUMU)
DEF_KERNEL(k_ds_bpermute, LDSDECL, ALL8(BPERM, ""), SUMU)
#define BARR(INS, R) asm volatile("s_barrier" ::: "memory");
DEF_KERNEL(k_s_barrier, U8, ALL8(BARR, ""), SUMU)

typedef void (*kern_t)(double *, uint64_t *, int);
struct entry { const char *name; kern_t k; };

int main(int argc, char **argv)
{
    const bool quick = argc > 1;      // `valu_probe quick`: 3 launches per kernel, 8 waves per SIMD only (for runs under rocprofv3 --pmc)
    const entry tab[] = {
        {"v_fma_f64 (8 chains)", k_fma_f64}, {"v_fma_f64 (4 chains)", k_fma_f64_c4}, {"v_fma_f64 (2 chains)", k_fma_f64_c2},
        {"v_fma_f64 (1 chain)", k_fma_f64_c1}, {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64}, {"v_min_f64", k_min_f64},
        {"v_fma_f32 (8 chains)", k_fma_f32}, {"v_fma_f32 (1 chain)", k_fma_f32_c1}, {"v_xor_b32", k_xor_b32}, {"v_bitop3_b32 (xor3)", k_bitop3_xor3}, {"v_bitop3_b32 (xor3, one SGPR)", k_bitop3_xor3_sgpr}, {"2 x v_xor_b32 (= xor3; counted as ONE)", k_xor_pair}, {"v_add_u32", k_add_u32},
        {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_hi_u32", k_mul_hi_u32}, {"v_mad_u64_u32", k_mad_u64_u32}, {"v_lshlrev_b64", k_lshl_b64},
        {"v_mov_b64", k_mov_b64}, {"v_cndmask_b32", k_cndmask}, {"v_cmp_lt_f64", k_cmp_f64}, {"v_rsq_f64", k_rsq_f64},
        {"v_rcp_f64", k_rcp_f64}, {"v_cvt_f64_u32", k_cvt_f64_u32}, {"v_pk_fma_f32", k_pk_fma_f32}, {"v_mov_b32 dpp row_ror", k_mov_dpp},
        {"v_cndmask_b32 (sgpr mask)", k_cndmask_sgpr}, {"v_cmp_lt_u32 + v_cndmask_b32 (pair = 1)", k_cmp_cndmask_pair}, {"v_cmp_lt_u32", k_cmp_u32},
        {"v_add3_u32", k_add3_u32}, {"v_xad_u32", k_xad_u32}, {"v_lshl_add_u32", k_lshl_add_u32}, {"v_alignbit_b32", k_alignbit},
        {"v_bfe_u32", k_bfe_u32}, {"v_perm_b32", k_perm_b32}, {"v_mad_u32_u24", k_mad_u32_u24}, {"v_and_or_b32", k_and_or_b32},
        {"v_lshl_add_u64", k_lshl_add_u64}, {"v_add_co_u32", k_add_co_u32}, {"v_readlane_b32", k_readlane}, {"v_ldexp_f64", k_ldexp_f64},
        {"v_frexp_mant_f64", k_frexp_mant_f64}, {"v_fract_f64", k_fract_f64}, {"v_cvt_i32_f64", k_cvt_i32_f64}, {"v_sqrt_f64", k_sqrt_f64},
        {"v_log_f32", k_log_f32}, {"ds_read_b128", k_ds_read_b128}, {"ds_read_b64", k_ds_read_b64}, {"ds_read_b32", k_ds_read_b32},
        {"ds_write_b64", k_ds_write_b64}, {"ds_bpermute_b32 (+wait)", k_ds_bpermute}, {"s_barrier (4 waves)", k_s_barrier},
    };
    const int iters = 1 << 11;
    double *sink; uint64_t *stamps;
    hipMalloc(&sink, 8);
    hipMalloc(&stamps, 256 * 8 * 16);   // return codes unchecked: diagnostic
    std::vector<uint64_t> h(2 * 256 * 8);
    printf("[\n");
    bool first = true;
    for (int waves : {8, 4, 1}) {
        if (quick && waves != 8) continue;                    // waves per SIMD: 256 CUs x (waves) blocks of 256 threads
        const int blocks = 256 * waves;
        for (const entry &e : tab) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < (quick ? 3 : 60); ++rep) {        // back-to-back launches, time the last one
                hipEventRecord(e0);
                hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, sink, stamps, iters);
                hipEventRecord(e1);
            }
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), stamps, blocks * 16, hipMemcpyDeviceToHost);
            std::vector<double> mhz, cyc;
            for (int b = 0; b < blocks; ++b) if (h[2 * b + 1]) { mhz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 100.0); cyc.push_back((double)h[2 * b]); }
            std::sort(mhz.begin(), mhz.end()); std::sort(cyc.begin(), cyc.end());
            const double per_wave = 64.0 * iters;       // instructions issued by one wave
            // in-kernel: cycles of one wave's loop / its instructions, times the waves sharing the SIMD
            const double cpi_inkernel = cyc[cyc.size() / 2] / per_wave / waves;
            const double clk = mhz[mhz.size() / 2] * 1e6;
            const double cpi_event = (ms * 1e-3 * clk) / (per_wave * blocks * 4 / 1024.0);
            printf("%s{\"instruction\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_wave_instruction_per_simd\": %.2f, "
                   "\"from_event_time\": %.2f, \"shader_clock_mhz\": %.0f}", first ? "" : ",\n", e.name, waves, cpi_inkernel, cpi_event, mhz[mhz.size() / 2]);
            first = false;
        }
    }
    printf("\n]\n");
    return 0;
}
