// v_mfma_f32_32x32x1f32 (2 blocks of 32x32, K = 1): operand / result layout, and the half-wave exchange that brings block b into the 32 lanes of
// half b with every matrix row in a register of its own (HessAcc32 in solve_g.h).
// hipcc --offload-arch=gfx950 -O2 -o mfma_layout32 mfma_layout32.hip && ./mfma_layout32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float v32f __attribute__((ext_vector_type(32)));
__global__ void k(float *out) {
    const int l = threadIdx.x;
    v32f z;
    for (int v = 0; v < 32; v++) z[v] = 0.f;
    v32f a = __builtin_amdgcn_mfma_f32_32x32x1f32(1000.f * (l / 32) + 10.f * (l % 32), 1.f, z, 0, 0, 0);       // 1000 b + 10 i
    v32f d = __builtin_amdgcn_mfma_f32_32x32x1f32(1.f, 0.01f * (l % 32), a, 0, 0, 0);                           // + 0.01 j
    for (int v = 0; v < 32; v++) { const float f = d[v]; out[v * 64 + l] = f; }
    unsigned r[32];
    for (int v = 0; v < 32; v++) { const float f = d[v]; r[v] = __float_as_uint(f); }
    for (int t = 0; t < 16; t++) { auto sw = __builtin_amdgcn_permlane32_swap(r[t], r[16 + t], false, false); r[t] = sw[0]; r[16 + t] = sw[1]; }
    for (int v = 0; v < 32; v++) out[(32 + v) * 64 + l] = __uint_as_float(r[v]);
}
int main() {
    float *d; (void)hipMalloc(&d, 64 * 64 * sizeof(float));
    k<<<1, 64>>>(d);
    static float h[64 * 64];
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("D layout (lane 0 / 32 / 5 / 37): register -> block, row i, column j\n");
    for (int v = 0; v < 32; v++) {
        printf("v%2d:", v);
        for (int l : {0, 32, 5, 37}) { const float x = h[v * 64 + l]; const int b = (int)(x / 1000), i = (int)((x - 1000 * b) / 10 + 1e-3f); printf("  l%02d=(b%d i%2d j%4.1f)", l, b, i, (x - 1000 * b - 10 * i) * 100); }
        printf("\n");
    }
    int bad = 0;
    for (int i = 0; i < 32; i++) for (int l = 0; l < 64; l++) {
        const int reg = ((i / 4) % 2 ? 16 : 0) + 4 * (i / 8) + i % 4;
        const float want = 1000.f * (l / 32) + 10.f * i + 0.01f * (l % 32);
        if (fabsf(h[(32 + reg) * 64 + l] - want) > 2e-3f) { if (bad < 8) printf("row %d lane %d (reg %d): %.2f want %.2f\n", i, l, reg, h[(32 + reg) * 64 + l], want); bad++; }
    }
    printf("exchange mismatches: %d\n", bad);
    return bad != 0;
}
