// Microbenchmark (round 5): the convex-pair narrowphase on a BIG hull (246 vertices, the HSR base) against a box that stays a few millimetres away - the situation of the
// slowest workgroups of cfg1 / cfg2 (the base next to the table): cycles of the two hull rescans along a cached direction and of a cold mpr_penetration run that ends with a
// separating direction, for 8- and 16-lane sub-groups, one sub-group busy.  s_memtime against s_memrealtime (100 MHz) gives the clock of the tick.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I hsr_env_amd/csrc tools/micro/mpr_bighull.hip -o /tmp/mpr_bighull && /tmp/mpr_bighull
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <vector>
#include "collide.h"

template <int W> __global__ void k_bench(const float4 *verts, int nvert, int reps, float gap, unsigned long long *out, float *res) {
    const int tid = threadIdx.x, sg = tid / W;
    Geom A, B;
    A.type = GEOM_MESH; A.nvert = nvert; A.verts = verts; A.size = mk3(0, 0, 0);
    B.type = GEOM_BOX; B.nvert = 0; B.verts = nullptr; B.size = mk3(0.05f, 0.025f, 0.017f);
#pragma unroll
    for (int k = 0; k < 9; k++) { A.mat.a[k] = (k % 4 == 0) ? 1.f : 0.f; B.mat.a[k] = (k % 4 == 0) ? 1.f : 0.f; }
    A.pos = mk3(0, 0, 0); A.bc = A.pos; A.bh = mk3(0.22f, 0.22f, 0.1f); B.bc = mk3(0, 0, 0); B.bh = B.size;
    float acc = 0;
    unsigned long long t0 = 0, t1 = 0, t2 = 0, r0 = 0, r1 = 0;
    int nsup_tot = 0;
    if (sg < 1) {
        r0 = __builtin_amdgcn_s_memrealtime();
        t0 = __builtin_amdgcn_s_memtime();
        for (int r = 0; r < reps; r++) {          // rescans along a cached direction
            B.pos = mk3(0.22f + 0.05f + gap + 1e-6f * r, 0.003f, 0.0f);
            const v3 d = normalized(mk3(-1.f, 1e-4f * r, 0.f));
            acc += -dot(support<W>(A, d) - support<W>(B, -d), d);
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int r = 0; r < reps; r++) {          // cold MPR runs
            B.pos = mk3(0.22f + 0.05f + gap + 1e-6f * r, 0.003f, 0.0f);
            const float a = 0.3f + 1e-4f * r;
            B.mat.a[0] = cosf(a); B.mat.a[1] = -sinf(a); B.mat.a[3] = sinf(a); B.mat.a[4] = cosf(a);
            float depth; v3 dir, pos, sep; int nsup = 0;
            const bool hit = mpr_penetration<W>(A, B, 1e-6f, 50, depth, dir, pos, sep, nsup, nullptr);
            acc += hit ? depth + pos.x + dir.z : sep.x;
            nsup_tot += nsup;
        }
        t2 = __builtin_amdgcn_s_memtime();
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    if (tid == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = r1 - r0; out[3] = nsup_tot; res[0] = acc; }
}

int main() {
    std::vector<float4> v;
    for (int i = 0; i < 246; i++) {
        const float z = 1.f - 2.f * (i + 0.5f) / 246.f, r = sqrtf(1 - z * z), ph = 2.39996323f * i;
        v.push_back(make_float4(0.22f * r * cosf(ph), 0.22f * r * sinf(ph), 0.1f * z, 0));
    }
    float4 *dv; unsigned long long *dt; float *dr;
    hipMalloc(&dv, v.size() * sizeof(float4)); hipMemcpy(dv, v.data(), v.size() * sizeof(float4), hipMemcpyHostToDevice);
    hipMalloc(&dt, 8 * sizeof(unsigned long long)); hipMalloc(&dr, 8 * sizeof(float));
    const int reps = 2000;
    for (float gap : {0.004f, -0.003f})
    for (int w : {8, 16}) {
        for (int rep = 0; rep < 2; rep++) {
            if (w == 8) hipLaunchKernelGGL(k_bench<8>, dim3(1), dim3(64), 0, 0, dv, 246, reps, gap, dt, dr);
            else hipLaunchKernelGGL(k_bench<16>, dim3(1), dim3(64), 0, 0, dv, 246, reps, gap, dt, dr);
        }
        hipDeviceSynchronize();
        unsigned long long t[4]; float r[1];
        hipMemcpy(t, dt, sizeof t, hipMemcpyDeviceToHost); hipMemcpy(r, dr, sizeof r, hipMemcpyDeviceToHost);
        const double tick_mhz = 100.0 * (double)(t[0] + t[1]) / (double)t[2];
        printf("gap %+.0f mm, %2d-lane sub-group: rescans %.0f ticks per pair of supports, MPR run %.0f ticks (%.2f supports per run); s_memtime tick = %.0f MHz; checksum %g\n",
               1e3 * gap, w, (double)t[0] / reps, (double)t[1] / reps, (double)t[3] / reps, tick_mhz, r[0]);
    }
    return 0;
}
