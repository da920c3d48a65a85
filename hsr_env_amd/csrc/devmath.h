// Small fp32 vector helpers shared by the HIP kernels (device only).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

// Single-wave workgroups (64 threads): LDS operations of one wave are performed in issue order, so a "barrier" only has to
// keep the compiler from moving memory operations across it - no s_barrier and, above all, no s_waitcnt vmcnt(0) on whatever
// global stores / prefetches are in flight, which __syncthreads() implies.  Not for exchanging data through GLOBAL memory.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ bool wave_any(bool x) { return __ballot(x) != 0ull; }

// compile-time unrolled loop: f(std::integral_constant<int, I>) for I in [I0, N)
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

#define HSR_MINVAL 1e-15f
#define HSR_MINIMP 0.0001f
#define HSR_MAXIMP 0.9999f
#define HSR_EPS 1.1920929e-7f

// 1-ulp hardware reciprocal / square root (v_rcp_f32, v_sqrt_f32, v_rsq_f32) instead of the ~10-instruction IEEE
// sequences: the fp32 path is checked against the fp64 oracle at tolerances far above an ulp
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float frsq(float x) { return __builtin_amdgcn_rsqf(x); }

// sin and cos for joint half-angles and integration steps (|x| of a few pi at most): one three-term Cody-Waite reduction to
// [-pi/4, pi/4] and the Cephes single-precision minimax polynomials (1 ulp on the reduced range) - the library sincosf
// carries a Payne-Hanek path for huge arguments that costs ~400 instructions of code at every call site
__device__ __forceinline__ void fast_sincos(float x, float *sn, float *cs) {
    const float k = rintf(x * 0.63661977236758134f);                       // x / (pi/2)
    float r = fmaf(k, -1.5703125f, x);                                     // pi/2 in three pieces: 1e-7 absolute up to |x| ~ 20
    r = fmaf(k, -4.837512969970703125e-4f, r);
    r = fmaf(k, -7.54978995489188e-8f, r);
    const float z = r * r;
    const float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
    const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z, fmaf(-0.5f, z, 1.0f));
    const int q = (int)k & 3;
    const float s0 = (q & 1) ? pc : ps, c0 = (q & 1) ? ps : pc;
    *sn = (q & 2) ? -s0 : s0;
    *cs = ((q + 1) & 2) ? -c0 : c0;
}

struct v3 { float x, y, z; };
struct m3 { float a[9]; };   // row-major

__device__ __forceinline__ v3 mk3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ v3 operator+(v3 a, v3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 operator-(v3 a, v3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 operator-(v3 a) { return mk3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ v3 operator*(v3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ v3 operator*(float s, v3 a) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ v3 cross(v3 a, v3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float norm(v3 a) { return fsqrt(dot(a, a)); }
__device__ __forceinline__ v3 normalized(v3 a) {
    const float d = dot(a, a);
    if (d < HSR_MINVAL * HSR_MINVAL) return mk3(1.f, 0.f, 0.f);
    return a * frsq(d);
}
__device__ __forceinline__ float comp(v3 a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }
__device__ __forceinline__ v3 mulmv(const m3 &m, v3 v) {
    return mk3(m.a[0] * v.x + m.a[1] * v.y + m.a[2] * v.z, m.a[3] * v.x + m.a[4] * v.y + m.a[5] * v.z,
               m.a[6] * v.x + m.a[7] * v.y + m.a[8] * v.z);
}
__device__ __forceinline__ v3 mulmtv(const m3 &m, v3 v) {
    return mk3(m.a[0] * v.x + m.a[3] * v.y + m.a[6] * v.z, m.a[1] * v.x + m.a[4] * v.y + m.a[7] * v.z,
               m.a[2] * v.x + m.a[5] * v.y + m.a[8] * v.z);
}
__device__ __forceinline__ m3 mulmm(const m3 &a, const m3 &b) {
    m3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.a[3 * i + j] = a.a[3 * i] * b.a[j] + a.a[3 * i + 1] * b.a[3 + j] + a.a[3 * i + 2] * b.a[6 + j];
    return r;
}
__device__ __forceinline__ v3 col(const m3 &m, int k) { return mk3(m.a[k], m.a[3 + k], m.a[6 + k]); }

struct q4 { float w, x, y, z; };
__device__ __forceinline__ q4 qmul(q4 a, q4 b) {
    q4 r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
    r.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
    return r;
}
__device__ __forceinline__ q4 qnormalized(q4 q) {
    const float d = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
    if (d < HSR_MINVAL * HSR_MINVAL) { q.w = 1.f; q.x = q.y = q.z = 0.f; return q; }
    const float inv = frsq(d);
    q.w *= inv; q.x *= inv; q.y *= inv; q.z *= inv;
    return q;
}
__device__ __forceinline__ m3 q2m(q4 q) {
    m3 m;
    float w = q.w, x = q.x, y = q.y, z = q.z;
    m.a[0] = 1 - 2 * (y * y + z * z); m.a[1] = 2 * (x * y - w * z); m.a[2] = 2 * (x * z + w * y);
    m.a[3] = 2 * (x * y + w * z); m.a[4] = 1 - 2 * (x * x + z * z); m.a[5] = 2 * (y * z - w * x);
    m.a[6] = 2 * (x * z - w * y); m.a[7] = 2 * (y * z + w * x); m.a[8] = 1 - 2 * (x * x + y * y);
    return m;
}

// strided per-thread array view: element i lives at p[i * s]  (global SoA: s = N; LDS: s = blockDim)
struct View {
    float *p;
    int s;
    __device__ __forceinline__ float &operator[](int i) const { return p[(size_t)i * (size_t)s]; }
    __device__ __forceinline__ View sub(int off) const { View v; v.p = p + (size_t)off * (size_t)s; v.s = s; return v; }
    __device__ __forceinline__ v3 get3(int i) const { return mk3((*this)[3 * i], (*this)[3 * i + 1], (*this)[3 * i + 2]); }
    __device__ __forceinline__ void set3(int i, v3 v) const { (*this)[3 * i] = v.x; (*this)[3 * i + 1] = v.y; (*this)[3 * i + 2] = v.z; }
    __device__ __forceinline__ m3 getm(int i) const { m3 m;
#pragma unroll
        for (int k = 0; k < 9; k++) m.a[k] = (*this)[9 * i + k];
        return m; }
    __device__ __forceinline__ void setm(int i, const m3 &m) const {
#pragma unroll
        for (int k = 0; k < 9; k++) (*this)[9 * i + k] = m.a[k]; }
};
struct IView {
    int *p;
    int s;
    __device__ __forceinline__ int &operator[](int i) const { return p[(size_t)i * (size_t)s]; }
};

__device__ __forceinline__ const float *cptr(const float *p) { return p; }
__device__ __forceinline__ v3 ld3(const float *p, int i) { return mk3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }
__device__ __forceinline__ m3 ldm(const float *p, int i) { m3 m;
#pragma unroll
    for (int k = 0; k < 9; k++) m.a[k] = p[9 * i + k];
    return m; }
