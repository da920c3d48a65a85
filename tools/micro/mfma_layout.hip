// Where does v_mfma_f32_16x16x1f32 (4 blocks of 16x16, K = 1) keep its operands and results?  One wave; A = 1000 block + i + 1 against
// B = 1 gives D[block][i][j] = the row index, the other way round the column index.  Then the 4x4 transposition of (lane row, register
// group) with v_permlane32_swap / v_permlane16_swap that brings block b into the 16 lanes of row b, register k = matrix row k.
// hipcc --offload-arch=gfx950 -O2 -o mfma_layout mfma_layout.hip && ./mfma_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void k(float *out) {
    const int l = threadIdx.x;
    const float code = 1000.f * (l / 16) + (l % 16) + 1.f;
    v16f z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    v16f d1 = __builtin_amdgcn_mfma_f32_16x16x1f32(code, 1.f, z, 0, 0, 0);
    v16f d2 = __builtin_amdgcn_mfma_f32_16x16x1f32(1.f, code, z, 0, 0, 0);
    for (int v = 0; v < 16; v++) { out[(0 * 16 + v) * 64 + l] = d1[v]; out[(1 * 16 + v) * 64 + l] = d2[v]; }
    // transposition: element = 10000 block + 100 i + j
    v16f d3 = z;
    {   // build it from the decoded layout instead of assuming one: a second pair of products
        v16f a = __builtin_amdgcn_mfma_f32_16x16x1f32(1000.f * (l / 16) + 10.f * (l % 16), 1.f, z, 0, 0, 0);      // 1000 b + 10 i
        d3 = __builtin_amdgcn_mfma_f32_16x16x1f32(1.f, 0.01f * (l % 16), a, 0, 0, 0);                             // + 0.01 j (chained accumulation)
    }
    unsigned r[16];
    for (int v = 0; v < 16; v++) { const float f = d3[v]; r[v] = __float_as_uint(f); }
    for (int g = 0; g < 2; g++) for (int v = 0; v < 4; v++) {       // register groups g and g + 2: upper half of the wave <-> lower half
        auto sw = __builtin_amdgcn_permlane32_swap(r[4 * g + v], r[4 * (g + 2) + v], false, false);
        r[4 * g + v] = sw[0]; r[4 * (g + 2) + v] = sw[1];
    }
    for (int v = 0; v < 16; v++) out[(3 * 16 + v) * 64 + l] = __uint_as_float(r[v]);
    for (int g = 0; g < 4; g += 2) for (int v = 0; v < 4; v++) {    // register groups g and g + 1: odd rows <-> even rows
        auto sw = __builtin_amdgcn_permlane16_swap(r[4 * g + v], r[4 * (g + 1) + v], false, false);
        r[4 * g + v] = sw[0]; r[4 * (g + 1) + v] = sw[1];
    }
    for (int v = 0; v < 16; v++) out[(2 * 16 + v) * 64 + l] = __uint_as_float(r[v]);
}
int main() {
    float *d; hipMalloc(&d, 4 * 16 * 64 * sizeof(float));
    k<<<1, 64>>>(d);
    static float h[4 * 16 * 64];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("D layout: register v, lane l -> (block, i) from A, (block, j) from B\n");
    for (int v = 0; v < 16; v++) {
        printf("v%2d:", v);
        for (int l = 0; l < 64; l += 5) { const int a = (int)h[(0 * 16 + v) * 64 + l] - 1, b = (int)h[(1 * 16 + v) * 64 + l] - 1; printf("  l%02d=(b%d i%2d | b%d j%2d)", l, a / 1000, a % 1000, b / 1000, b % 1000); }
        printf("\n");
    }
    for (int st = 3; st >= 2; st--) {
        printf("after stage %d: register group (v = 0 of it) x lane row -> block, quarter\n", st == 3 ? 1 : 2);
        for (int g = 0; g < 4; g++) { printf(" group %d:", g); for (int q = 0; q < 4; q++) { const float x = h[(st * 16 + 4 * g) * 64 + 16 * q]; printf("  row%d=(b%d q%d)", q, (int)(x / 1000), ((int)x % 1000) / 40); } printf("\n"); }
    }
    int bad = 0;
    for (int v = 0; v < 16; v++) for (int l = 0; l < 64; l++) {
        const float want = 1000.f * (l / 16) + 10.f * v + 0.01f * (l % 16);       // lane row = block, register = matrix row, lane in row = column
        if (fabsf(h[(2 * 16 + v) * 64 + l] - want) > 1e-3f) { if (bad < 8) printf("transposed v%d l%d: %.2f want %.2f\n", v, l, h[(2 * 16 + v) * 64 + l], want); bad++; }
    }
    printf("transposition mismatches: %d\n", bad);
    return bad != 0;
}
