// Microbenchmark of the convex-pair narrowphase of the persistent kernel (collide.h: mpr_penetration<8>): cycles per run for a box
// against a 32-vertex hull, cold and warm-started, with 1 or 8 sub-groups of the wave busy.  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I hsr_env_amd/csrc tools/micro/mpr_bench.hip -o /tmp/mpr_bench && /tmp/mpr_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <vector>
#include "collide.h"

__global__ void k_bench(const float4 *verts, int nvert, int nsub, int reps, int warm, unsigned long long *out, float *res) {
    const int tid = threadIdx.x, sg = tid / 8;
    Geom A, B;
    A.type = GEOM_MESH; A.nvert = nvert; A.verts = verts; A.size = mk3(0, 0, 0);
    B.type = GEOM_BOX; B.nvert = 0; B.verts = nullptr; B.size = mk3(0.05f, 0.025f, 0.017f);
#pragma unroll
    for (int k = 0; k < 9; k++) { A.mat.a[k] = (k % 4 == 0) ? 1.f : 0.f; B.mat.a[k] = (k % 4 == 0) ? 1.f : 0.f; }
    A.pos = mk3(0, 0, 0); A.bc = A.pos; A.bh = mk3(0.03f, 0.03f, 0.03f); B.bc = mk3(0, 0, 0); B.bh = B.size;
    int wid[3] = {0, 0, 0};
    float acc = 0;
    unsigned long long t0 = 0, t1 = 0;
    if (sg < nsub) {
        t0 = __builtin_amdgcn_s_memtime();
        for (int r = 0; r < reps; r++) {
            // the box dips 3 mm into the top of the hull and drifts sideways a little every run
            B.pos = mk3(0.004f + 1e-5f * r + 0.001f * sg, 0.003f, 0.02f + 0.017f - 0.003f);
            const float a = 0.3f + 1e-4f * r;
            B.mat.a[0] = cosf(a); B.mat.a[1] = -sinf(a); B.mat.a[3] = sinf(a); B.mat.a[4] = cosf(a);
            float depth; v3 dir, pos, sep; int nsup = 0;
            if (!warm) { wid[0] = wid[1] = wid[2] = 0; }
            const bool hit = mpr_penetration<8>(A, B, 1e-6f, 50, depth, dir, pos, sep, nsup, wid);
            acc += hit ? depth + pos.x + dir.z + 1e-3f * nsup : sep.x;
        }
        t1 = __builtin_amdgcn_s_memtime();
    }
    if (tid % 8 == 0 && sg < nsub) { out[sg] = t1 - t0; res[sg] = acc; }
}

int main() {
    // 32 vertices of a squashed sphere (a convex polytope after hulling; MPR only ever asks for support points)
    std::vector<float4> v;
    for (int i = 0; i < 32; i++) {
        const float z = 1.f - 2.f * (i + 0.5f) / 32.f, r = sqrtf(1 - z * z), ph = 2.39996323f * i;
        v.push_back(make_float4(0.03f * r * cosf(ph), 0.025f * r * sinf(ph), 0.02f * z, 0));
    }
    float4 *dv; unsigned long long *dt; float *dr;
    hipMalloc(&dv, v.size() * sizeof(float4)); hipMemcpy(dv, v.data(), v.size() * sizeof(float4), hipMemcpyHostToDevice);
    hipMalloc(&dt, 8 * sizeof(unsigned long long)); hipMalloc(&dr, 8 * sizeof(float));
    const int reps = 2000;
    for (int warm = 0; warm < 2; warm++)
        for (int nsub : {1, 8}) {
            hipLaunchKernelGGL(k_bench, dim3(1), dim3(64), 0, 0, dv, 32, nsub, reps, warm, dt, dr);
            hipLaunchKernelGGL(k_bench, dim3(1), dim3(64), 0, 0, dv, 32, nsub, reps, warm, dt, dr);
            hipDeviceSynchronize();
            unsigned long long t[8]; float r[8];
            hipMemcpy(t, dt, sizeof t, hipMemcpyDeviceToHost); hipMemcpy(r, dr, sizeof r, hipMemcpyDeviceToHost);
            printf("%s start, %d sub-group(s) busy: %.0f s_memtime ticks per run (x 24 = %.0f shader clocks at 2.4 GHz / 100 MHz), checksum %g\n", warm ? "warm" : "cold", nsub,
                   (double)t[0] / reps, 24.0 * t[0] / reps, r[0]);
        }
    return 0;
}
