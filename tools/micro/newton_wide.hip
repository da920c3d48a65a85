// What would the pieces of a Newton iteration cost that a 64-lane layout for ONE env changes?  (round-5 verdict, item 1, step 1; DESIGN.md "what a lone wave could gain")
// One wave on an idle CU, a cfg3-sized problem: nv = 13 (16 dof lanes), 7 contacts x (6 rows + the 2 cone rows) = 14 row quads.  Layout: lane = (dof i = l % 16,
// slot k = l / 16), register t = row quad: Jt[t] holds J[4 t + k][i]; a per-row scalar x_r lives replicated in the 16 lanes of DPP row k of register t.
//   cone   the two cone rows of every contact, z6 = sum_j gn_j J_j and z7 = sum_j u_j J_j: multiply, add across the four DPP rows (permlane16/32 swap-adds), select
//          into slots 2 and 3 of the contact's second quad
//   hess   J^T W J as 14 x v_mfma_f32_16x16x4_f32 (A = Jt[t], B = w[t] Jt[t]), then the result (lane (j, q), register v: H[4 q + v][j]) into the row layout the
//          16-lane factorisation takes (lane c: row c in 13 registers): 12 permlane swaps
//   grad   J^T g: 14 FMAs + two swap-adds
//   jmul   J s: per quad one multiply + a four-step DPP row reduction
//   xlds   the layout traffic the cone arithmetic needs (it stays lane = contact): J s out of the replicated layout to the contact lanes, the row weights / cone
//          vectors / gradient back, through LDS
// Each piece runs REPS times in a dependent chain between two s_memtime stamps; the Hessian is checked against the host.  The pieces a wide layout does NOT change
// (factorisation, line search, evaluation: DESIGN.md) are not here.   hipcc --offload-arch=gfx950 -O3 -o newton_wide newton_wide.hip && ./newton_wide
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int NQ = 14, NC = 7, REPS = 200;

__device__ __forceinline__ float xrow_sum(float v) {      // sum over the four DPP rows (lanes i, i + 16, i + 32, i + 48), in every row
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float s1 = __builtin_bit_cast(float, (unsigned)a[0]) + __builtin_bit_cast(float, (unsigned)a[1]);
    const unsigned u2 = __builtin_bit_cast(unsigned, s1);
    const auto b = __builtin_amdgcn_permlane32_swap(u2, u2, false, false);
    return __builtin_bit_cast(float, (unsigned)b[0]) + __builtin_bit_cast(float, (unsigned)b[1]);
}
template <int CTRL> __device__ __forceinline__ float dpp(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true)); }
__device__ __forceinline__ float row_sum(float v) { v += dpp<0x128>(v); v += dpp<0x124>(v); v += dpp<0x122>(v); v += dpp<0x121>(v); return v; }      // over the 16 lanes of the DPP row
__device__ __forceinline__ unsigned long long now() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }

__global__ void __launch_bounds__(64) k(const float *J, const float *wrow, const float *gnrow, const float *urow, const float *grow, const float *svec, float *Hout, float *vout, unsigned long long *cyc) {
    __shared__ __align__(16) float sh[256];
    const int l = threadIdx.x, i = l % 16, kq = l / 16;
    float Jt[NQ], w[NQ], gn[NQ], uu[NQ], g[NQ];
    for (int t = 0; t < NQ; t++) { const int r = 4 * t + kq; Jt[t] = J[r * 16 + i]; w[t] = wrow[r]; gn[t] = gnrow[r]; uu[t] = urow[r]; g[t] = grow[r]; }
    float s = svec[i];
    float chain = 0.f;      // threads the repetitions together so that none can be hoisted or dropped
    // ---- cone rows
    unsigned long long t0 = now();
    for (int rep = 0; rep < REPS; rep++) {
#pragma unroll
        for (int ci = 0; ci < NC; ci++) {
            const int a = 2 * ci, b = 2 * ci + 1;
            const float j1 = kq < 2 ? Jt[b] : 0.f;
            const float z6 = xrow_sum(Jt[a] * gn[a] + j1 * gn[b] + chain), z7 = xrow_sum(Jt[a] * uu[a] + j1 * uu[b]);
            Jt[b] = kq == 2 ? z6 : (kq == 3 ? z7 : Jt[b]);
        }
        chain = Jt[1] * 1e-30f;
    }
    unsigned long long t1 = now();
    cyc[0] = t1 - t0;
    // ---- Hessian on the matrix core + the move into the row layout
    float Hrow[16];
    t0 = now();
    for (int rep = 0; rep < REPS; rep++) {
        v4f acc = {chain, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NQ; t++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Jt[t], w[t] * Jt[t], acc, 0, 0, 0);
        unsigned r[16];
#pragma unroll
        for (int v = 0; v < 4; v++) { const float f = acc[v]; r[v] = __float_as_uint(f); r[4 + v] = 0; r[8 + v] = 0; r[12 + v] = 0; }
#pragma unroll
        for (int v = 0; v < 4; v++) { const auto sw = __builtin_amdgcn_permlane32_swap(r[v], r[8 + v], false, false); r[v] = sw[0]; r[8 + v] = sw[1]; }
#pragma unroll
        for (int gq = 0; gq < 4; gq += 2)
#pragma unroll
            for (int v = 0; v < 4; v++) { const auto sw = __builtin_amdgcn_permlane16_swap(r[4 * gq + v], r[4 * (gq + 1) + v], false, false); r[4 * gq + v] = sw[0]; r[4 * (gq + 1) + v] = sw[1]; }
#pragma unroll
        for (int q = 0; q < 16; q++) Hrow[q] = __uint_as_float(r[q]);
        chain = Hrow[3] * 1e-30f;
    }
    t1 = now();
    cyc[1] = t1 - t0;
    if (l < 16) for (int q = 0; q < 16; q++) Hout[l * 16 + q] = Hrow[q];
    // ---- gradient J^T g
    float grad = 0.f;
    t0 = now();
    for (int rep = 0; rep < REPS; rep++) {
        float p = chain;
#pragma unroll
        for (int t = 0; t < NQ; t++) p += Jt[t] * g[t];
        grad = xrow_sum(p);
        chain = grad * 1e-30f;
    }
    t1 = now();
    cyc[2] = t1 - t0;
    // ---- J search
    float jv[NQ];
    t0 = now();
    for (int rep = 0; rep < REPS; rep++) {
#pragma unroll
        for (int t = 0; t < NQ; t++) jv[t] = row_sum(Jt[t] * s + chain);
        chain = jv[5] * 1e-30f;
    }
    t1 = now();
    cyc[3] = t1 - t0;
    // ---- layout traffic through LDS: J s to the contact lanes (lane = contact reads its 8 slots), weights / cone vectors / gradient rows back (4 x 14 replicated reads)
    t0 = now();
    for (int rep = 0; rep < REPS; rep++) {
        if (i == 0) {
#pragma unroll
            for (int t = 0; t < NQ; t++) sh[4 * t + kq] = jv[t] + chain;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        float4 a = make_float4(0, 0, 0, 0), b = a;
        if (l < NC) { a = *reinterpret_cast<const float4 *>(sh + 8 * l); b = *reinterpret_cast<const float4 *>(sh + 8 * l + 4); }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        if (l < NC) {      // (the cone arithmetic of the contact lane would sit here)
#pragma unroll
            for (int f = 0; f < 4; f++) { *reinterpret_cast<float4 *>(sh + 64 * f + 8 * l) = make_float4(a.x + f, a.y, a.z, a.w); *reinterpret_cast<float4 *>(sh + 64 * f + 8 * l + 4) = b; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < NQ; t++) { w[t] = sh[4 * t + kq]; gn[t] = sh[64 + 4 * t + kq]; uu[t] = sh[128 + 4 * t + kq]; g[t] = sh[192 + 4 * t + kq]; }
        chain = (w[2] + gn[3] + uu[4] + g[5]) * 1e-30f;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
    t1 = now();
    cyc[4] = t1 - t0;
    vout[l] = grad + jv[0] + chain + w[1];
}

int main() {
    std::vector<float> J(64 * 16, 0.f), w(64, 0.f), gn(64, 0.f), u(64, 0.f), g(64, 0.f), s(16, 0.f);
    unsigned x = 12345;
    auto rnd = [&]() { x = x * 1664525u + 1013904223u; return ((x >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (int r = 0; r < 56; r++) {
        const int j = r % 8;
        if (j < 6) { for (int i = 0; i < 13; i++) J[r * 16 + i] = rnd(); gn[r] = rnd(); u[r] = rnd(); g[r] = rnd(); }
        w[r] = j < 6 ? 10.f + 20.f * (rnd() + 0.5f) : (j == 6 ? 3.f : -0.5f);      // row weights, then Dm and -k3 of the cone rows
    }
    for (int i = 0; i < 13; i++) s[i] = rnd();
    // host Hessian: H = sum_rows w_r J_r J_r^T with the cone rows z6 = sum_j gn_j J_j, z7 = sum_j u_j J_j of every contact
    std::vector<double> H(16 * 16, 0.0);
    for (int ci = 0; ci < NC; ci++) {
        double z6[16] = {0}, z7[16] = {0};
        for (int j = 0; j < 6; j++) for (int i = 0; i < 16; i++) { z6[i] += (double)gn[8 * ci + j] * J[(8 * ci + j) * 16 + i]; z7[i] += (double)u[8 * ci + j] * J[(8 * ci + j) * 16 + i]; }
        for (int a = 0; a < 16; a++) for (int b = 0; b < 16; b++) {
            for (int j = 0; j < 6; j++) H[a * 16 + b] += (double)w[8 * ci + j] * J[(8 * ci + j) * 16 + a] * J[(8 * ci + j) * 16 + b];
            H[a * 16 + b] += (double)w[8 * ci + 6] * z6[a] * z6[b] + (double)w[8 * ci + 7] * z7[a] * z7[b];
        }
    }
    float *dJ, *dw, *dgn, *du, *dg, *ds, *dH, *dv; unsigned long long *dc;
    hipMalloc(&dJ, J.size() * 4); hipMalloc(&dw, 256); hipMalloc(&dgn, 256); hipMalloc(&du, 256); hipMalloc(&dg, 256); hipMalloc(&ds, 64); hipMalloc(&dH, 1024); hipMalloc(&dv, 256); hipMalloc(&dc, 64);
    hipMemcpy(dJ, J.data(), J.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dw, w.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dgn, gn.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(du, u.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dg, g.data(), 256, hipMemcpyHostToDevice); hipMemcpy(ds, s.data(), 64, hipMemcpyHostToDevice);
    for (int run = 0; run < 2; run++) k<<<1, 64>>>(dJ, dw, dgn, du, dg, ds, dH, dv, dc);
    float h[256]; unsigned long long c[8];
    hipMemcpy(h, dH, 1024, hipMemcpyDeviceToHost); hipMemcpy(c, dc, 64, hipMemcpyDeviceToHost);
    double worst = 0, scale = 0;
    for (int a = 0; a < 13; a++) for (int b = 0; b < 13; b++) { worst = fmax(worst, fabs(h[a * 16 + b] - H[a * 16 + b])); scale = fmax(scale, fabs(H[a * 16 + b])); }
    printf("Hessian on 16x16x4 MFMAs against the host: max |dH| = %.3g (largest entry %.3g)\n", worst, scale);
    const char *nm[5] = {"cone rows (7 contacts)", "Hessian: 14 MFMAs + move to the row layout", "J^T g", "J search (14 row quads)", "layout traffic through LDS"};
    double tot = 0;
    for (int p = 0; p < 5; p++) { printf("  %-44s %7.0f cycles per iteration\n", nm[p], (double)c[p] / REPS); tot += (double)c[p] / REPS; }
    printf("  %-44s %7.0f cycles   (16-lane layout in the kernel, slowest workgroups, per iteration: Hessian 2.5 k + gradient 0.7 k + J search 1.2 k = 4.4 k)\n", "sum of the pieces a wide layout changes", tot);
    return worst > 1e-3 * scale;
}
