// Microbenchmark: issue cost (clocks per wave64 instruction) of the instruction forms the solver is built from, with 1 and 2 waves
// per SIMD.  hipcc --offload-arch=gfx950 -O3 -o issue_rate issue_rate.hip && ./issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
template <int MODE> __global__ void __launch_bounds__(64) k(float *out, unsigned long long *cyc, int iters) {
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, s = 0.5f + threadIdx.x, t = 1e-3f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if constexpr (MODE == 0) {        // plain v_fmac_f32, 8 independent accumulators
            REP16(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(t));)
        } else if constexpr (MODE == 1) { // v_fmac_f32_dpp row_newbcast
            REP16(asm volatile("v_fmac_f32_dpp %0, %8, %9 row_newbcast:0 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %2, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %4, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %5, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %6, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %7, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(t));)
        } else if constexpr (MODE == 2) { // v_fmac_f32_dpp row_shr:1
            REP16(asm volatile("v_fmac_f32_dpp %0, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %2, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %4, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %5, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %6, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %7, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(t));)
        } else if constexpr (MODE == 3) { // dependent chain of plain fmac (latency)
            REP64(asm volatile("v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %1, %2" : "+v"(a0) : "v"(s), "v"(t));)
        } else if constexpr (MODE == 4) { // dependent chain of dpp fmac: acc feeds src (the 2-wait-state hazard: s_nop 1)
            REP64(asm volatile("s_nop 1\n v_fmac_f32_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_fmac_f32_dpp %0, %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(a0) : "v"(t));)
        } else if constexpr (MODE == 5) { // v_pk_fma_f32, 4 independent pairs
            REP16(asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3" : "+v"(*(double *)&a0), "+v"(*(double *)&a2), "+v"(*(double *)&a4), "+v"(*(double *)&a6) : "v"(*(double *)&s), "v"(*(double *)&s));)
        } else if constexpr (MODE == 6) { // v_mov_b32_dpp row_newbcast (8 independent)
            REP16(asm volatile("v_mov_b32_dpp %0, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 row_newbcast:7 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));)
        } else if constexpr (MODE == 7) { // v_permlane16_swap (8 per statement)
            REP16(asm volatile("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if constexpr (MODE == 8) { // ds_bpermute round trip, dependent
            int idx = ((threadIdx.x + 5) & 63) * 4;
            REP16(asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a0) : "v"(idx));)
        } else if constexpr (MODE == 9) { // v_readlane + s use (8)
            int r;
            REP16(asm volatile("v_readlane_b32 %0, %1, 3\n v_readlane_b32 %0, %1, 5\n v_readlane_b32 %0, %1, 7\n v_readlane_b32 %0, %1, 9\n v_readlane_b32 %0, %1, 11\n v_readlane_b32 %0, %1, 13\n v_readlane_b32 %0, %1, 15\n v_readlane_b32 %0, %1, 17" : "=s"(r) : "v"(a0));)
            a1 += r;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char *name, int per_iter, float *out, unsigned long long *cyc) {
    const int iters = 200;
    for (int wps : {1, 2, 4}) {                 // waves per SIMD: 256 CUs x 4 SIMDs x wps single-wave workgroups
        const int nb = 256 * 4 * wps;
        hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(64), 0, 0, out, cyc, iters);
        hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(64), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        unsigned long long h[8192]; hipMemcpy(h, cyc, nb * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < nb; i++) s += h[i];
        // s_memtime counts at 100 MHz on gfx9; convert with the shader clock measured by mode 0
        printf("%-28s waves/SIMD %d: %.3f memtime ticks per instruction\n", name, wps, s / nb / iters / per_iter);
    }
}
int main() {
    float *out; unsigned long long *cyc; hipMalloc(&out, 8192 * 64 * 4); hipMalloc(&cyc, 8192 * 8);
    run<0>("v_fmac_f32 x8 indep", 128, out, cyc);
    run<1>("v_fmac_f32_dpp newbcast", 128, out, cyc);
    run<2>("v_fmac_f32_dpp row_shr", 128, out, cyc);
    run<3>("v_fmac_f32 dependent", 128, out, cyc);
    run<4>("nop1+fmac_dpp dependent", 128, out, cyc);
    run<5>("v_pk_fma_f32", 128, out, cyc);
    run<6>("v_mov_b32_dpp newbcast", 128, out, cyc);
    run<7>("v_permlane16_swap", 128, out, cyc);
    run<8>("ds_bpermute dependent", 128, out, cyc);
    run<9>("v_readlane", 128, out, cyc);
    return 0;
}
