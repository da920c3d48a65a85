// chol_mfma32_fwd (below: the dense 25 x 25 factorisation of the two 32-lane envs of a wave on the matrix core) against chol_g_fwd (DPP)
// on random SPD matrices: factor rows, reciprocal pivots and the forward-substituted right-hand side must agree; cycles of both.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../hsr_env_amd/csrc -o chol_mfma32 chol_mfma32.hip && ./chol_mfma32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "solve_g.h"
// ---- the experiment (round 4; NOT in the product: measured 5.9 k cycles against 5.2 k of chol_g_fwd on an idle CU - with 32 lanes per env every
// broadcast is two v_readlane + a select or a v_permlane16_swap + DPP, and the variant needs three per pivot step where the DPP one needs one;
// the first version, without the look-ahead, waited 16 passes for the matrix core at every step: 6.2 k) ----
// Dense factorisation + forward substitution of the two 32-lane envs of a wave ON THE MATRIX CORE (round 4; the 25 x 25 Newton Hessian of three
// blocks + arm): the matrix sits in the accumulator of v_mfma_f32_32x32x1 (2 blocks = the two envs, layout as HessAcc32), and the trailing
// update of pivot step j, A -= l l^T, is ONE instruction (A operand l, B operand -l, K = 1) instead of NK - j - 1 DPP-FMAs.  Column j of the
// current Schur complement comes back into the row layout (lane c of each env: A[c][j]) as row j of the accumulator - register
// 4 (j / 8) + j % 4 of the lane half (j / 4) % 2, both envs' registers exchanged by one v_permlane32_swap.  Same interface as chol_g_fwd:
// row[] = rows of L, invd, y; the operands are the lower triangle as the row lanes hold it (row c's entries k < c), like there.
// The matrix core's result is 16 passes away, so the pivot chain never waits for it: column j is taken out of the accumulator two steps ahead
// (at step j - 2, before that step's update is issued: it then holds the updates 0 .. j - 3, long finished) and the two updates it misses
// are applied to it by hand (a broadcast and an FMA each - the same FMAs the matrix core does: results bit-identical with chol_g_fwd).
template <int NK> __device__ __forceinline__ bool chol_mfma32_fwd(float (&row)[32], float &invd, int c, float b, float &y) {
    hess_v32f acc;
    static_for<0, 4>([&](auto ac) {
        static_for<0, 4>([&](auto ic) {
            constexpr int a = decltype(ac)::value, i = decltype(ic)::value, k0 = 8 * a + i, k1 = 8 * a + 4 + i;
            const unsigned x = k0 < NK ? __float_as_uint(row[k0 < NK ? k0 : 0]) : 0u, yv = k1 < NK ? __float_as_uint(row[k1 < NK ? k1 : 0]) : 0u;
            const auto sw = __builtin_amdgcn_permlane32_swap(x, yv, false, false);
            acc[4 * a + i] = __uint_as_float((unsigned)sw[0]); acc[16 + 4 * a + i] = __uint_as_float((unsigned)sw[1]);
        });
    });
    invd = 1.f;
    float sacc = b;
    y = 0.f;
    // col: column j (updates 0 .. j - 2 applied), nxt: column j + 1 (updates 0 .. j - 2 applied); by symmetry, as the accumulator holds the
    // matrix, the untouched columns 0 and 1 are every lane's own entries 0 and 1
    float col = row[0], nxt = NK > 1 ? row[1] : 0.f;
    float lprev = 0.f;           // L[c][j - 1]
    static_for<0, NK>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j > 0) col = fmaf(-lprev, gbcast<32, j>(lprev), col);          // the update of step j - 1
        const float ajj = gbcast<32, j>(col);
        const float inv = __builtin_amdgcn_rsqf(ajj);
        const float l = c >= j ? col * inv : 0.f;
        const float t = sacc * inv;
        if (c == j) { invd = inv; y = t; }
        row[j] = l;
        // column j + 1 gets this step's update; column j + 2 comes out of the accumulator as it stands (updates 0 .. j - 1), THEN this step's
        // update is issued
        if constexpr (j + 1 < NK) col = nxt;          // (its update of step j is applied at the top of the next step)
        if constexpr (j + 2 < NK) {
            constexpr int reg = 4 * ((j + 2) / 8) + (j + 2) % 4, half = ((j + 2) / 4) % 2;
            const float r0 = acc[reg], r1 = acc[16 + reg];
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(r0), __float_as_uint(r1), false, false);
            nxt = fmaf(-l, gbcast<32, j + 2>(l), __uint_as_float((unsigned)sw[half]));          // this step's update, by hand
            if constexpr (j + 3 < NK) acc = __builtin_amdgcn_mfma_f32_32x32x1f32(l, -l, acc, 0, 0, 0);
        }
        fmac_bcast<32, j, true>(sacc, c > j ? -l : 0.f, bc_prepare<32>(t));      // sacc -= L[c][j] y_j
        lprev = l;
    });
    return chol_pivots_ok<32>(invd);
}
constexpr int NK = 25;
__global__ void k(const float *A, const float *rhs, float *out, unsigned long long *cyc) {
    const int l = threadIdx.x, c = l % 32, e = l / 32;
    float r1[32], r2[32];
    for (int kk = 0; kk < 32; kk++) { r1[kk] = (kk < NK && c < NK) ? A[(e * 32 + c) * 32 + kk] : (kk == c ? 1.f : 0.f); r2[kk] = r1[kk]; }
    if (c >= NK) for (int kk = 0; kk < 32; kk++) { r1[kk] = r2[kk] = (kk == c && kk < NK) ? 1.f : 0.f; }
    const float b = c < NK ? rhs[e * 32 + c] : 0.f;
    float i1, y1, i2, y2;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    bool ok1 = chol_g_fwd<32, NK>(r1, i1, NK, c, b, y1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    bool ok2 = chol_mfma32_fwd<NK>(r2, i2, c, b, y2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long t2 = __builtin_amdgcn_s_memtime();
    for (int kk = 0; kk < NK; kk++) { out[(0 * 64 + l) * 32 + kk] = kk <= c ? r1[kk] : 0.f; out[(1 * 64 + l) * 32 + kk] = kk <= c ? r2[kk] : 0.f; }
    out[2 * 64 * 32 + l] = i1; out[2 * 64 * 32 + 64 + l] = i2; out[2 * 64 * 32 + 128 + l] = y1; out[2 * 64 * 32 + 192 + l] = y2;
    if (l == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = ok1; cyc[3] = ok2; }
}
int main() {
    std::vector<float> A(2 * 32 * 32, 0.f), rhs(64);
    srand(1);
    for (int e = 0; e < 2; e++) {
        float B[NK][NK];
        for (int i = 0; i < NK; i++) for (int j = 0; j < NK; j++) B[i][j] = (rand() / (float)RAND_MAX - 0.5f);
        for (int i = 0; i < NK; i++) for (int j = 0; j < NK; j++) { float s = i == j ? 1.0f + e : 0.f; for (int k2 = 0; k2 < NK; k2++) s += B[i][k2] * B[j][k2]; A[(e * 32 + i) * 32 + j] = s; }
        for (int i = 0; i < 32; i++) rhs[e * 32 + i] = rand() / (float)RAND_MAX - 0.5f;
    }
    float *dA, *dr, *dout; unsigned long long *dc;
    (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dr, 64 * 4); (void)hipMalloc(&dout, (2 * 64 * 32 + 256) * 4); (void)hipMalloc(&dc, 32);
    (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dr, rhs.data(), 64 * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; rep++) k<<<1, 64>>>(dA, dr, dout, dc);
    std::vector<float> out(2 * 64 * 32 + 256); unsigned long long cyc[4];
    (void)hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(cyc, dc, 32, hipMemcpyDeviceToHost);
    double dl = 0, di = 0, dy = 0;
    for (int l = 0; l < 64; l++) if (l % 32 < NK) {
        for (int kk = 0; kk < NK; kk++) dl = fmax(dl, fabs(out[(0 * 64 + l) * 32 + kk] - out[(1 * 64 + l) * 32 + kk]));
        di = fmax(di, fabs(out[2 * 64 * 32 + l] - out[2 * 64 * 32 + 64 + l]) / fabs(out[2 * 64 * 32 + l]));
        dy = fmax(dy, fabs(out[2 * 64 * 32 + 128 + l] - out[2 * 64 * 32 + 192 + l]));
    }
    printf("max |L_dpp - L_mfma| %.3e, rel |invd diff| %.3e, |y diff| %.3e; ok %llu %llu; cycles dpp %llu mfma %llu\n", dl, di, dy, cyc[2], cyc[3], cyc[0], cyc[1]);
    return !(dl < 1e-4 && di < 1e-4 && dy < 1e-4 && cyc[2] && cyc[3]);
}
