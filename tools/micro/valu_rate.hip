// Micro-benchmark (analysis only): issue rate of the VALU instructions the anchor kernel leans on.
// One workgroup of 1024 threads on one CU (4 waves per SIMD), each wave runs a long unrolled chain of
// 8 independent accumulators through ONE opcode; reports cycles per wave-instruction per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define OP_LIST(X) \
    X(0, "v_xor_b32",        "v_xor_b32 %0, %0, %1") \
    X(1, "v_pk_min_u16",     "v_pk_min_u16 %0, %0, %1") \
    X(2, "v_pk_sub_u16 opsel", "v_pk_sub_u16 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,1]") \
    X(3, "v_mad_u32_u24",    "v_mad_u32_u24 %0, %0, %1, %1") \
    X(4, "v_min_u32",        "v_min_u32 %0, %0, %1") \
    X(5, "v_mov_dpp shr",    "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf") \
    X(6, "v_alignbit_b32",   "v_alignbit_b32 %0, %0, %1, 11") \
    X(7, "v_perm_b32",       "v_perm_b32 %0, %0, %1, %1") \
    X(8, "v_bitop3_b32",     "v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96") \
    X(9, "v_pk_add_u16",     "v_pk_add_u16 %0, %0, %1") \
    X(10, "v_sub_u16",       "v_sub_u16 %0, %0, %1") \
    X(11, "v_mul_lo_u32",    "v_mul_lo_u32 %0, %0, %1") \
    X(12, "v_mul_u32_u24",   "v_mul_u32_u24 %0, %0, %1") \
    X(13, "v_lshl_add_u32",  "v_lshl_add_u32 %0, %0, 3, %1") \
    X(14, "v_and_or_b32",    "v_and_or_b32 %0, %0, %1, %1") \
    X(15, "v_min3_u32",      "v_min3_u32 %0, %0, %1, %1") \
    X(16, "v_cmp_eq_u16 sdwa", "v_cmp_eq_u16_sdwa vcc, %0, %1 src0_sel:WORD_1 src1_sel:WORD_0") \
    X(17, "v_cmp_eq_u32",    "v_cmp_eq_u32 vcc, %0, %1") \
    X(18, "v_sad_u16",       "v_sad_u16 %0, %0, %1, %1") \
    X(19, "v_pk_max_u16",    "v_pk_max_u16 %0, %0, %1")

template <int OP>
__global__ __launch_bounds__(1024) void k(uint32_t *out, long long *cycles, int iters) {
    uint32_t a[8], b = threadIdx.x * 2654435761u | 1u;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i * 977u;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#define X(N, NAME, ASM) if (OP == N) asm volatile(ASM : "+v"(a[i]) : "v"(b));
                OP_LIST(X)
#undef X
            }
        }
    }
    const long long t1 = clock64();
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r ^= a[i];
    out[threadIdx.x] = r;
    if (threadIdx.x == 0) cycles[0] = t1 - t0;
}

template <int OP>
void run(const char *name, uint32_t *out, long long *cyc) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(1024), 0, 0, out, cyc, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(1024), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    // every SIMD of every CU: 4 waves x iters x 64 wave-instructions
    const double n = 4.0 * iters * 64;
    printf("%-22s %8.3f ms  %.3f ns per wave-instruction per SIMD  (clock64 ticks/instr %.3f)\n", name, ms, ms * 1e6 / n, double(c) / n);
}

int main() {
    uint32_t *out; long long *cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
#define X(N, NAME, ASM) run<N>(NAME, out, cyc);
    OP_LIST(X)
#undef X
    return 0;
}
