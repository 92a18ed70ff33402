// valu_rate.hip — micro-benchmark: sustained VALU issue rate per SIMD on gfx950 for the instruction
// kinds the render kernels use (plain fp32 FMA, packed FMA, DPP add, v_exp, v_cndmask, readlane).
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o /tmp/valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* cyc, int active) {
    if ((int)(threadIdx.x & 63) >= active) return;  // EXEC = the low `active` lanes for the whole kernel (partial-EXEC issue cost)
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f, c = 0.5f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {  // 64 independent-ish v_fma_f32 (8 chains)
            REP16(asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(a0), "+v"(a1) : "v"(b), "v"(c));
                  asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
        } else if (KIND == 1) {  // v_pk_fma_f32 on register pairs
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = { a0, a1 }, p1 = { a2, a3 }, p2 = { a4, a5 }, p3 = { a6, a7 }, bb = { b, b }, cc = { c, c };
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3" : "+v"(p0), "+v"(p1) : "v"(bb), "v"(cc));
                  asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3" : "+v"(p2), "+v"(p3) : "v"(bb), "v"(cc));)
            a0 = p0.x + p1.x + p2.x + p3.x; a1 = p0.y + p1.y + p2.y + p3.y;
        } else if (KIND == 2) {  // DPP adds (row_ror:8), 4 chains
            REP16(asm volatile("v_add_f32_dpp %0, %1, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %3, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                               "v_add_f32_dpp %1, %0, %1 row_ror:4 row_mask:0xf bank_mask:0x5\n v_add_f32_dpp %3, %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (KIND == 3) {  // v_exp_f32
            REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (KIND == 4) {  // v_mul_f32 + v_add_f32 mix
            REP16(asm volatile("v_mul_f32 %0, %0, %2\n v_add_f32 %1, %1, %3" : "+v"(a0), "+v"(a1) : "v"(b), "v"(c));
                  asm volatile("v_mul_f32 %0, %0, %2\n v_add_f32 %1, %1, %3" : "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
        } else if (KIND == 5) {  // v_cndmask_b32 with SGPR mask
            unsigned long long m = 0xCCCCCCCCCCCCCCCCull;
            REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %1, %4\n v_cndmask_b32_e64 %1, %1, %2, %4\n v_cndmask_b32_e64 %2, %2, %3, %4\n v_cndmask_b32_e64 %3, %3, %0, %4"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(m));)
        } else if (KIND == 6) {  // pure SALU
            unsigned s0 = i, s1 = i + 1, s2 = i + 2, s3 = i + 3;
            REP16(asm volatile("s_add_u32 %0, %0, 3\n s_and_b32 %1, %1, 0xffff\n s_lshl_b32 %2, %2, 1\n s_or_b32 %3, %3, 5" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3));)
            a0 += (float)(s0 + s1 + s2 + s3);
        } else if (KIND == 7) {  // 1 VALU : 1 SALU interleaved
            unsigned s0 = i, s1 = i + 1;
            REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n s_add_u32 %2, %2, 3\n v_fma_f32 %1, %1, %4, %5\n s_and_b32 %3, %3, 0xffff" : "+v"(a0), "+v"(a1), "+s"(s0), "+s"(s1) : "v"(b), "v"(c));
                  asm volatile("v_fma_f32 %0, %0, %4, %5\n s_add_u32 %2, %2, 3\n v_fma_f32 %1, %1, %4, %5\n s_and_b32 %3, %3, 0xffff" : "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1) : "v"(b), "v"(c));)
            a0 += (float)(s0 + s1);
        } else if (KIND == 8) {  // 1 VALU : 3 SALU
            unsigned s0 = i, s1 = i + 1, s2 = i + 2;
            REP16(asm volatile("v_fma_f32 %0, %0, %5, %6\n s_add_u32 %2, %2, 3\n s_and_b32 %3, %3, 0xffff\n s_lshl_b32 %4, %4, 1" : "+v"(a0), "+v"(a1), "+s"(s0), "+s"(s1), "+s"(s2) : "v"(b), "v"(c));)
            a0 += (float)(s0 + s1 + s2);
        } else if (KIND == 9) {  // v_cmp (VOPC -> vcc) + v_cndmask vcc (VOP2)
            REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");)
        } else if (KIND == 11) {  // v_permlane32_swap (both operands read-write)
            REP16(asm volatile("v_permlane32_swap_b32_e32 %0, %1\n v_permlane32_swap_b32_e32 %2, %3\n v_permlane32_swap_b32_e32 %0, %2\n v_permlane32_swap_b32_e32 %1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (KIND == 12) {  // v_permlane16_swap
            REP16(asm volatile("v_permlane16_swap_b32_e32 %0, %1\n v_permlane16_swap_b32_e32 %2, %3\n v_permlane16_swap_b32_e32 %0, %2\n v_permlane16_swap_b32_e32 %1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (KIND == 13) {  // swap + dependent add (the reduce-scatter step)
            REP16(asm volatile("v_permlane32_swap_b32_e32 %0, %1\n v_add_f32 %0, %0, %1\n v_permlane32_swap_b32_e32 %2, %3\n v_add_f32 %2, %2, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (KIND == 14) {  // 8 INDEPENDENT full-mask DPP adds (a dependent one only every 8 instructions)
            REP4(asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_ror:12 row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_ror:4 row_mask:0xf bank_mask:0xf"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
                 asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_ror:12 row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_ror:4 row_mask:0xf bank_mask:0xf"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 15) {  // the same with bank-masked writes (the reduce-scatter's stage A / B form)
            REP4(asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xa\n"
                              "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc\n v_add_f32_dpp %3, %3, %3 row_ror:12 row_mask:0xf bank_mask:0x5\n"
                              "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0x3\n v_add_f32_dpp %5, %5, %5 row_ror:12 row_mask:0xf bank_mask:0x5\n"
                              "v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xc\n v_add_f32_dpp %7, %7, %7 row_ror:4 row_mask:0xf bank_mask:0xa"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
                 asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xa\n"
                              "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc\n v_add_f32_dpp %3, %3, %3 row_ror:12 row_mask:0xf bank_mask:0x5\n"
                              "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0x3\n v_add_f32_dpp %5, %5, %5 row_ror:12 row_mask:0xf bank_mask:0x5\n"
                              "v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xc\n v_add_f32_dpp %7, %7, %7 row_ror:4 row_mask:0xf bank_mask:0xa"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 16) {  // 8 independent plain adds, same structure
            REP4(asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3\n v_add_f32 %4, %4, %4\n v_add_f32 %5, %5, %5\n v_add_f32 %6, %6, %6\n v_add_f32 %7, %7, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
                 asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3\n v_add_f32 %4, %4, %4\n v_add_f32 %5, %5, %5\n v_add_f32 %6, %6, %6\n v_add_f32 %7, %7, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 17) {  // v_mov_b32_dpp (move only) + separate plain add: the two-instruction alternative
            REP4(asm volatile("v_mov_b32_dpp %4, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %6, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %3 row_ror:12 row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %6\n v_add_f32 %3, %3, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
                 asm volatile("v_mov_b32_dpp %4, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %6, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %3 row_ror:12 row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %6\n v_add_f32 %3, %3, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 10) {  // ds_read_b128 broadcast (uniform address)
            __shared__ float4 sm[256];
            if (i == 0) sm[threadIdx.x] = make_float4(a0, a1, a2, a3);
            float4 q0, q1;
            int addr = (i & 63) * 16;
            REP16(asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:1024\n s_waitcnt lgkmcnt(0)" : "=v"(q0), "=v"(q1) : "v"(addr) : "memory"); a0 += q0.x + q1.y;)
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND> void run(const char* name, int insts_per_iter, int waves_per_simd, int active = 64) {
    const int iters = 2000;
    const int blocks = 256 * waves_per_simd;  // 256 CUs x (waves_per_simd blocks of 4 waves)
    float* out; unsigned long long* cyc;
    hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<blocks, 256>>>(out, 10, cyc, active);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<blocks, 256>>>(out, iters, cyc, active);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double insts_per_simd = (double)insts_per_iter * iters * waves_per_simd;
    // s_memtime ticks at 100 MHz on this part: derive cycles from wall time at a nominal 2.4 GHz as well
    printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f cycles@2.4GHz per wave-instruction per SIMD (memtime ticks %llu)\n", name, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / insts_per_simd, c);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : { 2, 8 }) {
        run<0>("v_fma_f32", 64, w);
        run<1>("v_pk_fma_f32", 64, w);
        run<2>("v_add_f32_dpp", 64, w);
        run<3>("v_exp_f32", 64, w);
        run<4>("v_mul/v_add", 64, w);
        run<5>("v_cndmask(sgpr mask)", 64, w);
        run<6>("SALU only", 64, w);
        run<7>("1 VALU : 1 SALU (counts both)", 128, w);
        run<8>("1 VALU : 3 SALU (counts all)", 64, w);
        run<9>("v_cmp+v_cndmask vcc", 64, w);
        run<11>("v_permlane32_swap", 64, w);
        run<12>("v_permlane16_swap", 64, w);
        run<13>("swap32 + dependent add", 64, w);
        run<14>("dpp add, 8 independent", 64, w);
        run<15>("dpp add bank-masked, 8 indep", 64, w);
        run<16>("plain add, 8 independent", 64, w);
        run<17>("mov_dpp + plain add (counts both)", 64, w);
        run<0>("v_fma_f32, EXEC = lanes 0-31", 64, w, 32);
        run<0>("v_fma_f32, EXEC = lanes 0-15", 64, w, 16);
        run<3>("v_exp_f32, EXEC = lanes 0-31", 64, w, 32);
        run<5>("v_cndmask, EXEC = lanes 0-31", 64, w, 32);
        run<14>("dpp add,   EXEC = lanes 0-31", 64, w, 32);
    }
    return 0;
}
