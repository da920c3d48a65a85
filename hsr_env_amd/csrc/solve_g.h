// Cooperative dynamics + constraint + Newton + Euler kernel: one group of G lanes (16 or 32) per env,
// 64/G envs per single-wave workgroup, all per-env working data in LDS / registers.
//
// Why this shape on MI355X: at the benchmark size (8192 envs per GPU) one lane per env gives 128 waves
// for 1024 SIMDs; 16 lanes per env give 2048 waves, the Jacobian / Hessian blocks (<= 48 x 13) fit the
// 160 KB LDS of a CU several times over, and every cross-lane step is a width-16 shuffle or an LDS
// broadcast read.  Lane roles change per phase: lane = dof column (inertia, gradient, Hessian rows,
// Cholesky), lane = constraint row (J a, J search), lane = contact (cone evaluation).
//
// Same reference path as solve.h (mj_crb, mj_rne, actuation, mj_makeConstraint, mj_fwdConstraint Newton
// with elliptic cones, mj_Euler, HSREnv.step goal test: hsr/env.py:115-135; SURVEY.md 8 a-2.2 .. a-4).
#pragma once
#include "devmath.h"
#include "model.h"
#include "solve.h"
#include <type_traits>

// relative stop of the exact line search: |phi'(alpha)| < HSR_LS_REL |phi'(0)|
#ifndef HSR_LAST_DEC
#define HSR_LAST_DEC 1e-6f      // 100 x the solver tolerance of 1e-8 (solve_body.inc, Newton loop)
#endif
#ifndef HSR_LS_REL
#define HSR_LS_REL 1e-3f
#endif
#ifdef HSR_PHASE_TIMING
// diagnostic build only: stamp = one asm statement (s_memtime + its wait) fenced by sched_barriers, sums kept in
// registers and flushed once at the end (cdna_hip_programming.md section 7, in-kernel stamps)
__device__ __forceinline__ unsigned long long stamp_() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define DBGCNT(i, v) do { dc_[i] += (v); } while (0)
#define PHASE_T0() unsigned long long dc_[4] = {0, 0, 0, 0}; unsigned long long pt_[32] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long t_prev_ = stamp_(); const unsigned long long t_first_ = t_prev_, r_first_ = __builtin_amdgcn_s_memrealtime()
#define PHASE(idx) do { const unsigned long long t_now_ = stamp_(); pt_[idx] += t_now_ - t_prev_; t_prev_ = t_now_; } while (0)
#define PHASE_FLUSH() do { if (tid == 0) { if (blockIdx.x < 8192) { unsigned long long *bt_ = s.phase_cyc + 32 + 40 * blockIdx.x; _Pragma("unroll") for (int i_ = 0; i_ < 32; i_++) bt_[8 + i_] = pt_[i_]; bt_[4] = dc_[0]; bt_[5] = dc_[1]; bt_[6] = dc_[2]; bt_[7] = dc_[3]; bt_[0] = r_first_; bt_[1] = __builtin_amdgcn_s_memrealtime(); bt_[2] = __builtin_amdgcn_s_getreg(63492); bt_[3] = __builtin_amdgcn_s_getreg(63508); } atomicAdd(&s.phase_cyc[26], stamp_() - t_first_); atomicAdd(&s.phase_cyc[27], __builtin_amdgcn_s_memrealtime() - r_first_); _Pragma("unroll") for (int i_ = 0; i_ < 26; i_++) atomicAdd(&s.phase_cyc[i_], pt_[i_]); } } while (0)
#else
#define PHASE_T0() do {} while (0)
#define DBGCNT(i, v) do {} while (0)
#define PHASE(idx) do {} while (0)
#define PHASE_FLUSH() do {} while (0)
#endif
// DPP controls (gfx90a+): row_shr:n = 0x110+n, row_ror:n = 0x120+n, row_newbcast:n = 0x150+n; a "row" is 16 lanes,
// exactly one 16-lane env group, so these are single full-rate VALU modifiers instead of ds_bpermute round trips
template <int CTRL, bool ZERO_OOB> __device__ __forceinline__ float dpp_f(float v) {
    // mov_dpp leaves "old" undefined, so no init move is emitted and the DPP combine pass can fold the move into
    // its consumer (v_fmac_f32_dpp); every control used here reads a valid lane or (ZERO_OOB) wants 0
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, ZERO_OOB));
}
template <int CTRL, bool ZERO_OOB> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, ZERO_OOB); }
// G = 32 (two envs per wave): a DPP row is 16 lanes, so the cross-row half of every group operation goes through
// v_readlane (an SGPR per env, selected by the lane's env) instead of a ds_bpermute round trip through the LDS crossbar.
// v_readlane reads its lane regardless of EXEC, which is what a group-uniform branch of one env needs.
__device__ __forceinline__ int rl_(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
template <int L0, int L1> __device__ __forceinline__ int pick32_(int v) { const int a = rl_(v, L0), b = rl_(v, L1); return (threadIdx.x & 32) ? b : a; }
// value of lane LANE of the group, in every lane of the group
template <int G, int LANE> __device__ __forceinline__ float gbcast(float v) {
    if constexpr (G == 16) return dpp_f<0x150 + LANE, true>(v);
    else return __builtin_bit_cast(float, pick32_<LANE, 32 + LANE>(__builtin_bit_cast(int, v)));
}
// acc += t * (value of lane LANE of the group of src): one v_fmac_f32_dpp.  LLVM forms v_fmac only after its DPP combine has
// run on the VOP3 v_fma (no DPP encoding on gfx9), so the builtin form costs a v_mov_b32_dpp plus the fma.  The hand-written
// form is invisible to the hazard recogniser and the scheduler may place the VALU instruction that produces src right in front
// of it, so FIRST = true (the first use of a freshly computed src) carries the two wait states of the VALU-write -> DPP-read
// hazard inside the same asm statement; later uses of the same src need none.
// The broadcast source of a run of fmac_bcast.  G = 16: the value itself.  G = 32 (an env spans two DPP rows): one gfx950
// v_permlane16_swap of the value with itself yields the two registers "first row of my group in both rows" and "second row
// in both rows"; a row_newbcast on the right one then reaches any of the 32 lanes, so the whole run stays v_fmac_f32_dpp.
template <int G> struct BcSrc { float lo, hi; };
template <int G> __device__ __forceinline__ BcSrc<G> bc_prepare(float v) {
    BcSrc<G> r;
    if constexpr (G == 32) {
        const unsigned u = __builtin_bit_cast(unsigned, v);
        const auto sw = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        r.lo = __builtin_bit_cast(float, (unsigned)sw[0]); r.hi = __builtin_bit_cast(float, (unsigned)sw[1]);
    } else { r.lo = v; r.hi = v; }
    return r;
}
template <int G, int LANE, bool FIRST = false> __device__ __forceinline__ void fmac_bcast(float &acc, float t, const BcSrc<G> &src) {
    const float sv = (G == 32 && LANE >= 16) ? src.hi : src.lo;
    if constexpr (FIRST) asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(sv), "v"(t), "n"(LANE % 16));
    else asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(sv), "v"(t), "n"(LANE % 16));
}
// runs that cross lane 16 touch src.hi for the first time there: that use needs the wait states again (G = 32)
template <int G, int LANE, int FIRST_LANE> constexpr bool bc_first() { return LANE == FIRST_LANE || (G == 32 && LANE == 16 && FIRST_LANE < 16); }
// group broadcast of a value that the preceding hand-written v_fmac_f32_dpp may have produced: the hazard recogniser cannot
// see through inline asm, so this one carries its own wait states
template <int G, int LANE> __device__ __forceinline__ float gbcast_after_asm(float v) {
    if constexpr (G == 16) {
        float r;
        asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(LANE));
        return r;
    } else return gbcast<G, LANE>(v);
}
template <int G> __device__ __forceinline__ float gsum(float v) {
    v += dpp_f<0x128, true>(v); v += dpp_f<0x124, true>(v); v += dpp_f<0x122, true>(v); v += dpp_f<0x121, true>(v);      // every lane: sum of its row
    if constexpr (G == 16) return v;
    else {
        const unsigned u = __builtin_bit_cast(unsigned, v);
        const auto sw = __builtin_amdgcn_permlane16_swap(u, u, false, false);      // [first row sum, second row sum] in every row of the group
        return __builtin_bit_cast(float, (unsigned)sw[0]) + __builtin_bit_cast(float, (unsigned)sw[1]);
    }
}
template <int G> __device__ __forceinline__ int gscan_incl(int v, int c) {
    v += dpp_i<0x111, true>(v); v += dpp_i<0x112, true>(v); v += dpp_i<0x114, true>(v); v += dpp_i<0x118, true>(v);      // inclusive scan inside the row
    if constexpr (G == 16) return v;
    else { const int first = pick32_<15, 47>(v); return (threadIdx.x & 16) ? v + first : v; }                                // second row: plus the total of the first
}
template <int G> __device__ __forceinline__ int gor(int v) {     // bitwise OR over the group, in every lane
    v |= dpp_i<0x128, true>(v); v |= dpp_i<0x124, true>(v); v |= dpp_i<0x122, true>(v); v |= dpp_i<0x121, true>(v);
    if constexpr (G == 16) return v;
    else { const auto sw = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false); return (int)(sw[0] | sw[1]); }
}
template <int G> __device__ __forceinline__ int gmax(int v) {    // maximum over the group, in every lane
    { const int t = dpp_i<0x128, true>(v); v = t > v ? t : v; } { const int t = dpp_i<0x124, true>(v); v = t > v ? t : v; }
    { const int t = dpp_i<0x122, true>(v); v = t > v ? t : v; } { const int t = dpp_i<0x121, true>(v); v = t > v ? t : v; }
    if constexpr (G == 16) return v;
    else { const auto sw = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false); const int a = (int)sw[0], b = (int)sw[1]; return a > b ? a : b; }
}
// OR / maximum of a group-uniform value over the env groups of the wave, as a scalar (v_readlane ignores EXEC: call it from
// wave-uniform control flow only, with the value defined in every lane)
template <int G> __device__ __forceinline__ int wave_or_groups(int v) {
    if constexpr (G == 16) return rl_(v, 0) | rl_(v, 16) | rl_(v, 32) | rl_(v, 48); else return rl_(v, 0) | rl_(v, 32);
}
template <int G> __device__ __forceinline__ int wave_max_groups(int v) {
    if constexpr (G == 16) { const int a = rl_(v, 0), b = rl_(v, 16), c = rl_(v, 32), d = rl_(v, 48); const int x = a > b ? a : b, y = c > d ? c : d; return x > y ? x : y; }
    else { const int a = rl_(v, 0), b = rl_(v, 32); return a > b ? a : b; }
}
template <int G> __device__ __forceinline__ int glast(int v) {   // value of the last lane of the group
    if constexpr (G == 16) return dpp_i<0x15F, false>(v); else return pick32_<31, 63>(v);
}

// in-register cooperative Cholesky: lane c holds row c (entries k <= c) of an SPD matrix; on return row c of L in
// row[0..c] (entries k > c are scratch) and invd = 1 / L[c][c].  Columns j >= ndense are known to have no
// off-diagonal entries (block-diagonal tail of M): only their pivots are taken.
template <int G, int NK = G> __device__ __forceinline__ bool chol_g(float (&row)[G], float &invd, int nv, int ndense, int c) {
    bool ok = true;
    invd = 1.f;
    static_for<0, NK>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if (j < nv) {
            float ajj = gbcast_after_asm<G, j>(row[j]);
            if (!(ajj >= HSR_MINVAL)) { ok = false; ajj = 1.f; }
            const float inv = __builtin_amdgcn_rsqf(ajj);            // 1 ulp; the factor only shapes a Newton / Euler solve
            const float lcj = row[j] * inv;                          // lane j: ajj * rsq(ajj) = sqrt(ajj)
            if (c == j) invd = inv;
            row[j] = lcj;
            if (j < ndense) {
                const float nl = -lcj;
                const BcSrc<G> bl = bc_prepare<G>(lcj);
                static_for<j + 1, NK>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    fmac_bcast<G, i, bc_first<G, i, j + 1>()>(row[i], nl, bl);   // row[i] -= lcj * L[i][j]; unconditional: entries i > c are never read
                });
            }
        }
    });
    return ok;
}
// solve L L^T x = b with lane c holding row c of L, invd and b_c; tile = G*(G+1) floats of LDS scratch.
// Contains two workgroup barriers: must be called by every thread of the block.
template <int G> __device__ __forceinline__ float chol_solve_g(const float (&row)[G], float invd, float b, int nv, int c, float *tile) {
    float lo[G];
#pragma unroll
    for (int k = 0; k < G; k++) lo[k] = (k < c) ? row[k] : 0.f;     // strictly lower part, zero elsewhere
    float sacc = b, y = 0.f;
    static_for<0, G>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if (j < nv) {
            const float yj = gbcast<G, j>(sacc * invd);
            if (c == j) y = yj;
            sacc -= lo[j] * yj;
        }
    });
    __syncthreads();
#pragma unroll
    for (int k = 0; k < G; k++) tile[c * (G + 1) + k] = lo[k];
    __syncthreads();
    float lt[G];
#pragma unroll
    for (int k = 0; k < G; k++) lt[k] = tile[k * (G + 1) + c];      // L[k][c] for k > c, 0 otherwise
    float s2 = y, x = 0.f;
    static_for<0, G>([&](auto jc) {
        constexpr int j = G - 1 - decltype(jc)::value;
        if (j < nv) {
            const float xj = gbcast<G, j>(s2 * invd);
            if (c == j) x = xj;
            s2 -= lt[j] * xj;
        }
    });
    return x;
}

// elliptic cone at residual x: cost, gradient g, and the Hessian in the form
//   H = diag(dw) + Dm gn gn^T - k3 u u^T      (zone 0 top: all zero; 1 bottom: dw = D; 2 middle)
struct ConeOut { float cost, Dm, k3; int zone; float g[6], dw[6], gn[6], u[6]; };
__device__ __forceinline__ void cone_eval2(int dim, float mu, const float *fri, const float *D, const float *x, ConeOut &o) {
    float U[6], T2 = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) { o.g[j] = 0; o.dw[j] = 0; o.gn[j] = 0; o.u[j] = 0; }
    o.cost = 0; o.Dm = 0; o.k3 = 0; o.zone = 0;
    U[0] = x[0] * mu;
    const float Nn = U[0];
#pragma unroll
    for (int j = 1; j < 6; j++) { U[j] = (j < dim) ? x[j] * fri[j - 1] : 0.f; T2 += U[j] * U[j]; }
    const float T = fsqrt(T2);
    if (Nn >= mu * T || (T <= 0 && Nn >= 0)) return;
    if (mu * Nn + T <= 0 || (T <= 0 && Nn < 0)) {
        o.zone = 1;
#pragma unroll
        for (int j = 0; j < 6; j++) if (j < dim) { o.cost += 0.5f * D[j] * x[j] * x[j]; o.g[j] = D[j] * x[j]; o.dw[j] = D[j]; }
        return;
    }
    o.zone = 2;
    const float Dm = D[0] * frcp(mu * mu * (1 + mu * mu)), NT = Nn - mu * T, invT = frcp(T);
    const float kappa = -Dm * NT * mu;
    o.Dm = Dm; o.k3 = kappa * invT * frcp(T2);
    o.gn[0] = mu;
#pragma unroll
    for (int j = 1; j < 6; j++) if (j < dim) {
        o.gn[j] = -mu * U[j] * fri[j - 1] * invT;
        o.u[j] = fri[j - 1] * U[j];
        o.dw[j] = kappa * fri[j - 1] * fri[j - 1] * invT;
    }
#pragma unroll
    for (int j = 0; j < 6; j++) o.g[j] = Dm * NT * o.gn[j];
    o.cost = 0.5f * Dm * NT * NT;
}

// cost only (the warm-start comparison needs nothing else at the unconstrained point)
__device__ __forceinline__ float cone_cost(int dim, float mu, const float *fri, const float *D, const float *x) {
    const float Nn = x[0] * mu;
    float T2 = 0;
#pragma unroll
    for (int j = 1; j < 6; j++) { const float Uj = (j < dim) ? x[j] * fri[j - 1] : 0.f; T2 += Uj * Uj; }
    const float T = fsqrt(T2);
    if (Nn >= mu * T || (T <= 0 && Nn >= 0)) return 0.f;
    if (mu * Nn + T <= 0 || (T <= 0 && Nn < 0)) {
        float cst = 0;
#pragma unroll
        for (int j = 0; j < 6; j++) if (j < dim) cst += 0.5f * D[j] * x[j] * x[j];
        return cst;
    }
    const float Dm = D[0] * frcp(mu * mu * (1 + mu * mu)), NT = Nn - mu * T;
    return 0.5f * Dm * NT * NT;
}
// first and second derivative along v of the elliptic-cone cost at residual x (what the line search needs): the same
// zones and formulas as cone_eval2 contracted with v analytically, d1 = g . v, d2 = v^T (diag(dw) + Dm gn gn^T - k3 u u^T) v
__device__ __forceinline__ void cone_dd(int dim, float mu, const float *fri, const float *D, const float *x, const float *v, float &d1, float &d2) {
    d1 = 0; d2 = 0;
    const float Nn = x[0] * mu;
    float T2 = 0, S1 = 0, S2 = 0;
#pragma unroll
    for (int j = 1; j < 6; j++) {
        const float f = (j < dim) ? fri[j - 1] : 0.f, Uj = x[j] * f, fv = f * v[j];
        T2 += Uj * Uj; S1 += Uj * fv; S2 += fv * fv;
    }
    const float T = fsqrt(T2);
    if (Nn >= mu * T || (T <= 0 && Nn >= 0)) return;
    if (mu * Nn + T <= 0 || (T <= 0 && Nn < 0)) {
#pragma unroll
        for (int j = 0; j < 6; j++) if (j < dim) { d1 += D[j] * x[j] * v[j]; d2 += D[j] * v[j] * v[j]; }
        return;
    }
    const float Dm = D[0] * frcp(mu * mu * (1 + mu * mu)), NT = Nn - mu * T, invT = frcp(T);
    const float kappa = -Dm * NT * mu, gnv = mu * (v[0] - invT * S1);
    d1 = Dm * NT * gnv;
    d2 = kappa * invT * S2 + Dm * gnv * gnv - (kappa * invT * frcp(T2)) * S1 * S1;
}

// per-contact record in LDS (floats)
enum { CR_POS = 0, CR_N = 3, CR_T1 = 6, CR_T2 = 9, CR_DIST = 12, CR_MU = 13, CR_PAIR = 14, CR_ADR = 15, CR_DIM = 16,
       CR_ZONE = 17, CR_DM = 18, CR_K3 = 19, CR_GN = 20, CR_U = 26, CR_L1 = 32, CR_L2 = 33, CR_B = 34, CR_KD = 35, CR_FRI = 36, CR_SIZE = 41 };
enum { NLMAX = 16, NVEC = 8 };

// LDS floats of one env.  Transient tables alias longer-lived regions:
//   kin axes + link table live inside the (not yet written) Jacobian region during phases A-C,
//   the inertia matrix M shares the Cholesky-solve tile (M is copied to registers before the first solve).
template <int G> struct SolveLayout {
    int R, RS, MS, oJ, oD, oAref, oJar, oJv, oGr, oDw, oVec, oM, oTile, oAng, oLin, oAnc, oLk, oCon, total;
    __host__ __device__ SolveLayout(int rows) {
        R = rows; RS = G + 4; MS = G + 1;
        int o = 0;
        oJ = o;
        oAng = o; oLin = 0; oAnc = 0; oLk = 0;       // flat copy of the env's kin_aos record (kstride floats <= R*RS, checked on host)
        o += R * RS;
        oD = o; o += R; oAref = o; o += R; oJar = o; o += R; oJv = o; o += R; oGr = o; o += R; oDw = o; o += R;
        oVec = o; o += NVEC * G;
        oM = o; oTile = o; o += G * MS;
        oCon = o; o += CR_SIZE * G;
        total = (o + 3) & ~3;
    }
    __host__ __device__ bool fits(int kstride) const { return kstride <= R * RS; }
};

template <int G>
__global__ void __launch_bounds__(64) k_solve_g(DevModel m, DevState s, int mode, int goal_body, float geofence, int debug) {
    extern __shared__ __align__(16) float lds[];
    constexpr int EPB = 64 / G;
    const SolveLayout<G> L(m.njmax);
    const int tid = threadIdx.x, g = tid / G, c = tid % G;
    const int e_raw = blockIdx.x * EPB + g;
    const bool valid = e_raw < s.N && !s.done[e_raw < s.N ? e_raw : 0];
    if (!__syncthreads_or(valid)) return;
    const int e = valid ? e_raw : 0;
    const int N = s.N, nv = m.nv, R = L.R, RS = L.RS, MS = L.MS;
    float *E = lds + (size_t)g * L.total;
    float *J = E + L.oJ, *rD = E + L.oD, *rAref = E + L.oAref, *rJar = E + L.oJar, *rJv = E + L.oJv, *rGr = E + L.oGr, *rDw = E + L.oDw;
    float *vQvel = E + L.oVec, *vQfs = vQvel + G, *vQas = vQfs + G, *vQacc = vQas + G, *vMa = vQacc + G, *vSearch = vMa + G,
          *vWarm = vSearch + G, *vQfc = vWarm + G;
    float *M = E + L.oM, *tile = E + L.oTile, *con = E + L.oCon;
    float *kAng = E + L.oAng, *kLin = kAng + 3 * nv, *kAnc = kAng + 6 * nv, *lk = kAng + 9 * nv;     // layout of kin_aos
    const bool isdof = c < nv;
    int bad = 0;
    // model constants used inside loops: one copy per workgroup in LDS (no dependent global loads later)
    __shared__ int sParent[32], sMask[NLMAX];
    __shared__ float sMass[NLMAX];
    if (tid < nv) sParent[tid] = m.dof_parent[tid];
    if (tid < m.nlink && tid < NLMAX) { sMask[tid] = m.link_dofmask[tid]; sMass[tid] = m.link_mass[tid]; }

    PHASE_T0();
    // ---------------- phase A: every first-level global load of the kernel is issued here, back to back, so that
    // their (fabric / Infinity-Cache) latencies overlap; the data was written by the previous kernels.
    constexpr int MAXCH = 384 / G;                         // pair-count chunks of G pairs (npair <= 384)
    int cnt_ch[MAXCH];
    {
        const int *cp = s.ncon_pair + (size_t)e * m.npair_pad;
#pragma unroll
        for (int ch = 0; ch < MAXCH; ch++) { const int p = ch * G + c; cnt_ch[ch] = (valid && p < m.npair) ? cp[p] : 0; }
    }
    v3 a_c = mk3(0, 0, 0), l_c = mk3(0, 0, 0), n_c = mk3(0, 0, 0);
    float qvel_c = 0, warm_c = 0, my_q = 0, my_ctrl = 0, damp_c = 0;
    int my_type = -1, my_qadr = 0, my_quat_lane = -1, my_limited = 0, my_act = -1;
    float lim_lo = 0, lim_hi = 0, lim_sr0 = 1, lim_sr1 = 1, lim_iw = 0, lim_si[5] = {0, 0, 0, 0, 0};
    float act_p[7] = {0, 0, 0, 0, 0, 0, 0};
    if (isdof) {
        my_type = m.dof_type[c]; my_qadr = m.dof_qposadr[c]; my_limited = m.dof_limited[c]; my_act = m.dof_act[c];
        my_quat_lane = m.link_dofadr[m.dof_link[c]] + 3;
        damp_c = m.dof_damping[c];
        qvel_c = s.qvel[(size_t)c * N + e]; warm_c = s.warm[(size_t)c * N + e];
        lim_lo = m.dof_range[2 * c]; lim_hi = m.dof_range[2 * c + 1]; lim_sr0 = m.dof_solref[2 * c]; lim_sr1 = m.dof_solref[2 * c + 1];
        lim_iw = m.dof_invweight0[c];
#pragma unroll
        for (int j = 0; j < 5; j++) lim_si[j] = m.dof_solimp[5 * c + j];
        my_q = s.qpos[(size_t)my_qadr * N + e];
        if (my_act >= 0) {
            my_ctrl = s.ctrl[(size_t)my_act * N + e];
            act_p[0] = m.act_kp[my_act]; act_p[1] = m.act_gear[my_act];
            act_p[2] = m.act_ctrlrange[2 * my_act]; act_p[3] = m.act_ctrlrange[2 * my_act + 1];
            act_p[4] = m.act_forcerange[2 * my_act]; act_p[5] = m.act_forcerange[2 * my_act + 1];
        }
        if (!(fabsf(qvel_c) <= 1e10f) || !(fabsf(my_q) <= 1e10f)) bad = 1;
    }
    {   // the env's kinematic record: kstride floats, env-major, copied with 16-byte loads (256 B per group instruction)
        const float4 *src = reinterpret_cast<const float4 *>(s.kin_aos + (size_t)e * s.kstride);
        float4 *dst = reinterpret_cast<float4 *>(kAng);
        const int n4 = s.kstride / 4;
        float4 tmp[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { const int idx = c + i * G; tmp[i] = idx < n4 ? src[idx] : make_float4(0, 0, 0, 0); }
#pragma unroll
        for (int i = 0; i < 8; i++) { const int idx = c + i * G; if (idx < n4) dst[idx] = tmp[i]; }
    }
    // values only needed at the very end (quaternion of a free joint, goal test, counters) are fetched now as well
    q4 quat0; quat0.w = 1; quat0.x = quat0.y = quat0.z = 0;
    if (valid && isdof && c == my_quat_lane && my_type == DOF_FREE_ANG) {
        quat0.w = s.qpos[(size_t)my_qadr * N + e]; quat0.x = s.qpos[(size_t)(my_qadr + 1) * N + e];
        quat0.y = s.qpos[(size_t)(my_qadr + 2) * N + e]; quat0.z = s.qpos[(size_t)(my_qadr + 3) * N + e];
    }
    bool reach = false;
    float time0 = 0;
    int nsteps0 = 0;
    if (valid && c == 0) {
        time0 = s.time[e]; nsteps0 = s.nsteps[e];
        if (goal_body >= 0 && mode != 0) {
            // a-3 goal test on the xpos of this substep's forward pass (computed by k_kinematics)
            const v3 goal = mk3(s.mocap[e], s.mocap[N + e], s.mocap[2 * N + e]);
            v3 bp = goal;
            if (!m.body_mocap[goal_body]) {
                const int l = m.body_link[goal_body];
                View xpos{s.xpos + e, N}, xmat{s.xmat + e, N};
                bp = xpos.get3(l) + mulmv(xmat.getm(l), ld3(m.body_pos, goal_body));
            }
            reach = norm(bp - goal) < geofence;
        }
    }
    vQvel[c] = qvel_c; vWarm[c] = warm_c;
#pragma unroll
    for (int k = 0; k < G + 1; k++) M[c * MS + k] = 0.f;
    __syncthreads();
    if (isdof) {
        a_c = mk3(kAng[3 * c], kAng[3 * c + 1], kAng[3 * c + 2]);
        l_c = mk3(kLin[3 * c], kLin[3 * c + 1], kLin[3 * c + 2]);
        n_c = mk3(kAnc[3 * c], kAnc[3 * c + 1], kAnc[3 * c + 2]);
    }

    PHASE(0);
    // ---------------- phase B/C: inertia rows (a-2.2) and bias force (a-2.5), lane = dof
    float bias_c = 0;
    if (isdof) {
        // ancestors of dof c (c itself first), gathered once; per ancestor an accumulator in registers, so the
        // link loop has no LDS read-modify-write chain
        int anc[8];
        int nd = 0;
        {
            int k = c;
#pragma unroll
            for (int d = 0; d < 8; d++) { anc[d] = k >= 0 ? k : 0; if (k >= 0) { nd = d + 1; k = sParent[k]; } }
        }
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int l = 1; l < m.nlink; l++) {
            if (!((sMask[l] >> c) & 1)) continue;
            const float *q = lk + 15 * l;
            const v3 com = mk3(q[0], q[1], q[2]), F = mk3(q[9], q[10], q[11]), Nt = mk3(q[12], q[13], q[14]);
            const v3 jpc = l_c + cross(a_c, com - n_c);
            const v3 v = jpc * sMass[l];
            const v3 u = mk3(q[3] * a_c.x + q[6] * a_c.y + q[7] * a_c.z, q[6] * a_c.x + q[4] * a_c.y + q[8] * a_c.z, q[7] * a_c.x + q[8] * a_c.y + q[5] * a_c.z);
            bias_c += dot(jpc, F) + dot(a_c, Nt);
#pragma unroll
            for (int d = 0; d < 8; d++) {
                if (d < nd) {
                    const int k = anc[d];
                    const v3 ak = mk3(kAng[3 * k], kAng[3 * k + 1], kAng[3 * k + 2]);
                    const v3 jpk = mk3(kLin[3 * k], kLin[3 * k + 1], kLin[3 * k + 2]) + cross(ak, com - mk3(kAnc[3 * k], kAnc[3 * k + 1], kAnc[3 * k + 2]));
                    acc[d] += dot(jpk, v) + dot(ak, u);
                }
            }
        }
#pragma unroll
        for (int d = 0; d < 8; d++) if (d < nd) M[c * MS + anc[d]] = acc[d];
    }
    float qfs_c = 0;
    if (isdof) {
        qfs_c = -damp_c * qvel_c - bias_c;
        if (my_act >= 0) {
            // position actuator: force = kp*clamp(ctrl) - kp*gear*q, clamped to forcerange; qfrc = gear*force
            const float ct = fminf(fmaxf(my_ctrl, act_p[2]), act_p[3]);
            float f = act_p[0] * ct - act_p[0] * act_p[1] * my_q;
            f = fminf(fmaxf(f, act_p[4]), act_p[5]);
            qfs_c += act_p[1] * f;
        }
    }
    vQfs[c] = qfs_c;
    __syncthreads();
    if (isdof) for (int k = sParent[c]; k >= 0; k = sParent[k]) M[k * MS + c] = M[c * MS + k];   // mirror
    if (!isdof) M[c * MS + c] = 1.f;
    __syncthreads();
    if (debug && valid && isdof) for (int k = 0; k <= c; k++) s.M[(size_t)(c * (c + 1) / 2 + k) * N + e] = M[c * MS + k];

    PHASE(1);
    // ---------------- qacc_smooth = M^-1 qfrc_smooth
    float Mrow[G];
#pragma unroll
    for (int k = 0; k < G; k++) Mrow[k] = M[c * MS + k];
    float qas_c;
    {
        float Lr[G];
#pragma unroll
        for (int k = 0; k < G; k++) Lr[k] = Mrow[k];
        float invd;
        if (!chol_g<G>(Lr, invd, nv, m.ndense, c)) bad = 1;
        qas_c = chol_solve_g<G>(Lr, invd, qfs_c, nv, c, tile);
        if (!isdof) qas_c = 0;
    }
    vQas[c] = qas_c;

    PHASE(2);
    // ---------------- phase E: constraint assembly (a-2.4)
    // E1 joint limits, lane = dof, rows in dof order (lower side then upper side)
    int nlim;
    {
        int a0 = 0, a1 = 0;
        float dlo = 0, dhi = 0;
        if (valid && my_limited) {
            dlo = my_q - lim_lo; dhi = lim_hi - my_q;
            a0 = dlo < 0; a1 = dhi < 0;
        }
        const int incl = gscan_incl<G>(a0 + a1, c);
        nlim = glast<G>(incl);
        int r = incl - (a0 + a1);
#pragma unroll
        for (int side = 0; side < 2; side++) {
            if ((side == 0 ? a0 : a1) && r < R) {
                const float dist = side == 0 ? dlo : dhi, sg = side == 0 ? 1.f : -1.f;
                const float imp = impedance(lim_si, dist);
                const float dmax = fminf(fmaxf(lim_si[1], HSR_MINIMP), HSR_MAXIMP);
                const float Kimp = imp / (dmax * dmax * lim_sr0 * lim_sr0 * lim_sr1 * lim_sr1), B = 2.0f / (dmax * lim_sr0);
                const float Rr = fmaxf((1 - imp) / imp * lim_iw, HSR_MINVAL);
                for (int k = 0; k < G; k++) J[r * RS + k] = (k == c) ? sg : 0.f;
                rAref[r] = -B * sg * qvel_c - Kimp * dist;
                rD[r] = 1.0f / Rr;
                r++;
            }
        }
        if (nlim > R) nlim = R;
    }
    // E2 contact compaction in (pair, index) order: width-G scans over the per-pair counts (ballot-style)
    int ncon = 0;
    {
        int base = 0;
#pragma unroll
        for (int ch = 0; ch < MAXCH; ch++) {
            if (ch * G < m.npair) {
                const int p = ch * G + c, cnt = cnt_ch[ch];
                const int incl = gscan_incl<G>(cnt, c);
                const int tot = glast<G>(incl);
                if (cnt > 0) {
                    const int slot0 = m.pair_slot[p];
                    for (int i = 0; i < cnt; i++) {
                        const int ci = base + incl - cnt + i;
                        if (ci < G) { float *cr = con + CR_SIZE * ci; cr[CR_PAIR] = (float)p; cr[CR_ZONE] = (float)(slot0 + i); }   // ZONE reused for the source slot
                    }
                }
                base += tot;
            }
        }
        ncon = base < G ? base : G;
        if (ncon > m.nconmax) ncon = m.nconmax;
    }
    __syncthreads();
    PHASE(3);
    // E3 lane = contact: frame, impedance, regulariser, row addresses
    {
        int dim = 0;
        float *cr = con + CR_SIZE * c;
        float4 pr[4] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)}, cd0 = make_float4(0, 0, 0, 0), cd1 = cd0;
        if (c < ncon) {
            const int p = (int)cr[CR_PAIR], slot = (int)cr[CR_ZONE];
            const float4 *prp = reinterpret_cast<const float4 *>(m.pair_rec + 16 * p);
            const float4 *cdp = reinterpret_cast<const float4 *>(s.con + ((size_t)e * m.nslot + slot) * 8);
            pr[0] = prp[0]; pr[1] = prp[1]; pr[2] = prp[2]; pr[3] = prp[3]; cd0 = cdp[0]; cd1 = cdp[1];
            dim = (int)pr[0].x;
        }
        const int incl = gscan_incl<G>(dim, c);
        const int adr = nlim + incl - dim;
        const bool ovf = c < ncon && adr + dim > R;
        // first overflowing contact truncates the list (njmax semantics)
        unsigned long long bal = __ballot(ovf);
        const unsigned int gm = (unsigned int)((bal >> (g * G)) & ((G == 32) ? 0xffffffffull : 0xffffull));
        if (gm) ncon = __ffs(gm) - 1;
        if (c < ncon) {
            const v3 pos = mk3(cd0.x, cd0.y, cd0.z), nrm = mk3(cd0.w, cd1.x, cd1.y);
            const float dist = cd1.z;
            v3 t1 = (nrm.y > -0.5f && nrm.y < 0.5f) ? mk3(0, 1, 0) : mk3(0, 0, 1);
            t1 = normalized(t1 - nrm * dot(nrm, t1));
            const v3 t2 = cross(nrm, t1);
            const float fri[5] = {pr[1].x, pr[1].y, pr[1].z, pr[1].w, pr[2].x};
            const float solimp[5] = {pr[2].w, pr[3].x, pr[3].y, pr[3].z, pr[3].w};
            const float tc = pr[2].y, dr = pr[2].z, tran = pr[0].w;
            const float imp = impedance(solimp, dist), dmax = fminf(fmaxf(solimp[1], HSR_MINIMP), HSR_MAXIMP);
            const float B = 2.0f / (dmax * tc), Kimp = imp / (dmax * dmax * tc * tc * dr * dr);
            const float R0 = fmaxf((1 - imp) / imp * tran, HSR_MINVAL), R1 = R0 / fmaxf(m.impratio, HSR_MINVAL);
            cr[CR_POS] = pos.x; cr[CR_POS + 1] = pos.y; cr[CR_POS + 2] = pos.z;
            cr[CR_N] = nrm.x; cr[CR_N + 1] = nrm.y; cr[CR_N + 2] = nrm.z;
            cr[CR_T1] = t1.x; cr[CR_T1 + 1] = t1.y; cr[CR_T1 + 2] = t1.z;
            cr[CR_T2] = t2.x; cr[CR_T2 + 1] = t2.y; cr[CR_T2 + 2] = t2.z;
            cr[CR_DIST] = dist; cr[CR_MU] = dim > 1 ? fri[0] * sqrtf(R1 / R0) : fri[0];
            cr[CR_ADR] = (float)adr; cr[CR_DIM] = (float)dim;
            cr[CR_L1] = pr[0].y; cr[CR_L2] = pr[0].z;
            cr[CR_B] = B; cr[CR_KD] = Kimp * dist;
#pragma unroll
            for (int j = 0; j < 5; j++) cr[CR_FRI + j] = fri[j];
            for (int j = 0; j < dim; j++) {
                const float Rj = j == 0 ? R0 : (j == 1 ? R1 : R1 * fri[0] * fri[0] / (fri[j - 1] * fri[j - 1]));
                rD[adr + j] = 1.0f / Rj;
            }
        }
    }
    __syncthreads();
    PHASE(4);
    // E4 lane = dof column: Jacobian entries of every contact row
    int nefc = nlim;
    for (int ci = 0; ci < ncon; ci++) {
        const float *cr = con + CR_SIZE * ci;
        const int adr = (int)cr[CR_ADR], dim = (int)cr[CR_DIM], l1 = (int)cr[CR_L1], l2 = (int)cr[CR_L2];
        const int in1 = (sMask[l1] >> c) & 1, in2 = (sMask[l2] >> c) & 1;
        const float sg = (float)(in2 - in1);
        const v3 pos = mk3(cr[CR_POS], cr[CR_POS + 1], cr[CR_POS + 2]);
        const v3 vp = (l_c + cross(a_c, pos - n_c)) * sg, wr = a_c * sg;
        for (int j = 0; j < dim; j++) {
            const int jj = j % 3;
            const v3 ax = mk3(cr[3 + 3 * jj], cr[4 + 3 * jj], cr[5 + 3 * jj]);
            J[(adr + j) * RS + c] = isdof ? dot(j < 3 ? vp : wr, ax) : 0.f;
        }
        nefc = adr + dim;
    }
    __syncthreads();
    // E5 lane = row: reference acceleration of contact rows
    for (int ci = c; ci < ncon; ci += G) {
        const float *cr = con + CR_SIZE * ci;
        const int adr = (int)cr[CR_ADR], dim = (int)cr[CR_DIM];
        for (int j = 0; j < dim; j++) {
            float vel = 0;
            for (int k = 0; k < nv; k++) vel += J[(adr + j) * RS + k] * vQvel[k];
            rAref[adr + j] = -cr[CR_B] * vel - (j == 0 ? cr[CR_KD] : 0.f);
        }
    }
    __syncthreads();

    PHASE(5);
    // ---------------- phase F: Newton solver (a-2.6)
    const float tol = m.tolerance, scale = 1.0f / (m.meaninertia * (nv > 1 ? nv : 1));
    float cost = 0, Ma_c = 0;
    // total cost at the acceleration stored in `va`; leaves Ma, jar, gr, dw and the cone records in LDS
    // zones_changed: some constraint switched between its quadratic pieces (limit on/off, cone top/bottom) or is in
    // the curved middle zone, relative to the previous evaluation.  If an exact Newton step + exact line search
    // lands on a point with unchanged pieces, that point minimises the current quadratic piece: converged.
    bool zones_changed = true;
    auto eval_at = [&](const float *va) -> float {
        float ma = 0, unstable = 0.f;
#pragma unroll
        for (int k = 0; k < G; k++) ma += Mrow[k] * va[k];
        Ma_c = isdof ? ma : 0.f;
        float part = isdof ? 0.5f * (va[c] - qas_c) * (Ma_c - qfs_c) : 0.f;
        for (int r = c; r < nefc; r += G) {
            float sacc = -rAref[r];
            for (int k = 0; k < nv; k++) sacc += J[r * RS + k] * va[k];
            rJar[r] = sacc;
            if (r < nlim) {
                const bool was = rDw[r] != 0.f;
                if (sacc < 0) { part += 0.5f * rD[r] * sacc * sacc; rGr[r] = rD[r] * sacc; rDw[r] = rD[r]; unstable += was ? 0.f : 1.f; }
                else { rGr[r] = 0; rDw[r] = 0; unstable += was ? 1.f : 0.f; }
            }
        }
        __syncthreads();
        if (c < ncon) {
            float *cr = con + CR_SIZE * c;
            const int adr = (int)cr[CR_ADR], dim = (int)cr[CR_DIM];
            float D[6], x[6], fri[5];
#pragma unroll
            for (int j = 0; j < 5; j++) fri[j] = cr[CR_FRI + j];
#pragma unroll
            for (int j = 0; j < 6; j++) if (j < dim) { D[j] = rD[adr + j]; x[j] = rJar[adr + j]; } else { D[j] = 0; x[j] = 0; }
            ConeOut o;
            cone_eval2(dim, cr[CR_MU], fri, D, x, o);
            part += o.cost;
            // a cone that changed zone, or sits in the (non-quadratic) middle zone, keeps the Newton loop going
            if ((int)cr[CR_ZONE] != o.zone || o.zone == 2) unstable += 1.f;
            cr[CR_ZONE] = (float)o.zone; cr[CR_DM] = o.Dm; cr[CR_K3] = o.k3;
#pragma unroll
            for (int j = 0; j < 6; j++) { cr[CR_GN + j] = o.gn[j]; cr[CR_U + j] = o.u[j]; if (j < dim) { rGr[adr + j] = o.g[j]; rDw[adr + j] = o.dw[j]; } }
        }
        const float tot = gsum<G>(part);
        zones_changed = gsum<G>(unstable) > 0.f;
        __syncthreads();
        return tot;
    };

    bool active = valid && nefc > 0;
    int iter = 0;
    {
        const float cost_s = eval_at(vQas);
        const float cost_w = eval_at(vWarm);
        // block-uniform control flow: if any group prefers qacc_smooth, everyone re-evaluates at its own start
        const bool use_warm = cost_w < cost_s;
        vQacc[c] = use_warm ? vWarm[c] : vQas[c];
        cost = use_warm ? cost_w : cost_s;
        if (__syncthreads_or(!use_warm)) cost = eval_at(vQacc);
        __syncthreads();
    }
    PHASE(6);
    for (int it = 0; it < m.iterations; it++) {
        if (!__syncthreads_or(active)) break;
        // gradient, lane = dof
        float grad_c = 0;
        if (isdof) {
            grad_c = Ma_c - qfs_c;
            for (int r = 0; r < nefc; r++) grad_c += J[r * RS + c] * rGr[r];
        }
        const float gnorm = sqrtf(gsum<G>(grad_c * grad_c));
        if (scale * gnorm < tol) active = false;
        // Hessian rows H = M + J^T (d2s) J, lane = row c of H.  with_neg = false drops the negative rank-1 part
        // of the middle-zone cone Hessians (a PSD majorant), used only if the fp32 factorisation fails.
        float Hrow[G];
        auto build_H = [&](bool with_neg) {
#pragma unroll
            for (int k = 0; k < G; k++) Hrow[k] = Mrow[k];
            if (active) {
                for (int r = 0; r < nefc; r++) {
                    const float w = rDw[r];
                    if (w != 0.f) {
                        const float t = w * J[r * RS + c];
                        const float4 *jr = reinterpret_cast<const float4 *>(J + r * RS);
#pragma unroll
                        for (int k4 = 0; k4 < G / 4; k4++) {
                            const float4 q = jr[k4];
                            Hrow[4 * k4] += t * q.x; Hrow[4 * k4 + 1] += t * q.y; Hrow[4 * k4 + 2] += t * q.z; Hrow[4 * k4 + 3] += t * q.w;
                        }
                    }
                }
                for (int ci = 0; ci < ncon; ci++) {
                    const float *cr = con + CR_SIZE * ci;
                    if ((int)cr[CR_ZONE] != 2) continue;
                    const int adr = (int)cr[CR_ADR], dim = (int)cr[CR_DIM];
                    float pc = 0, wc = 0;
                    for (int j = 0; j < dim; j++) { const float jc = J[(adr + j) * RS + c]; pc += jc * cr[CR_GN + j]; wc += jc * cr[CR_U + j]; }
                    const float Dm = cr[CR_DM], k3 = with_neg ? cr[CR_K3] : 0.f;
                    static_for<0, G>([&](auto kc) {
                        constexpr int k = decltype(kc)::value;
                        const float pk = gbcast<G, k>(pc), wk = gbcast<G, k>(wc);
                        Hrow[k] += Dm * pc * pk - k3 * wc * wk;
                    });
                }
            }
            if (!isdof) {
#pragma unroll
                for (int k = 0; k < G; k++) Hrow[k] = (k == c) ? 1.f : 0.f;
            }
        };
        PHASE(7);
        build_H(true);
        PHASE(8);
        float hinvd;
        bool hfail = !chol_g<G>(Hrow, hinvd, nv, nv, c) && active;
        if (__syncthreads_or(hfail)) {
            // rare: rebuild with the PSD majorant for every group of the block (cheap, keeps barriers uniform)
            const bool use_neg = !hfail;
            build_H(use_neg);
            hfail = !chol_g<G>(Hrow, hinvd, nv, nv, c) && active;
            if (hfail) active = false;
        }
        float search_c = chol_solve_g<G>(Hrow, hinvd, -grad_c, nv, c, tile);
        if (!isdof) search_c = 0;
        __syncthreads();
        vSearch[c] = search_c;
        __syncthreads();
        PHASE(9);
        // exact line search (safeguarded 1-D Newton on phi'), all reductions by shuffles
        float mv = 0;
#pragma unroll
        for (int k = 0; k < G; k++) mv += Mrow[k] * vSearch[k];
        const float g1 = gsum<G>(search_c * (Ma_c - qfs_c)), g2 = gsum<G>(isdof ? search_c * mv : 0.f), snorm = sqrtf(gsum<G>(search_c * search_c));
        for (int r = c; r < nefc; r += G) {
            float sacc = 0;
            for (int k = 0; k < nv; k++) sacc += J[r * RS + k] * vSearch[k];
            rJv[r] = sacc;
        }
        __syncthreads();
        float alpha = 0;
        if (active) {
            // per-lane constants of the 1-D function
            float cD[6], cx[6], cv[6], cfri[5], cmu = 0;
            int cdim = 0;
            if (c < ncon) {
                const float *cr = con + CR_SIZE * c;
                const int adr = (int)cr[CR_ADR];
                cdim = (int)cr[CR_DIM]; cmu = cr[CR_MU];
#pragma unroll
                for (int j = 0; j < 5; j++) cfri[j] = cr[CR_FRI + j];
#pragma unroll
                for (int j = 0; j < 6; j++) if (j < cdim) { cD[j] = rD[adr + j]; cx[j] = rJar[adr + j]; cv[j] = rJv[adr + j]; } else { cD[j] = 0; cx[j] = 0; cv[j] = 0; }
            }
            auto ls_eval = [&](float al, float &dphi, float &ddphi) {
                float dp = 0, hp = 0;
                for (int r = c; r < nlim; r += G) {
                    const float jvi = rJv[r], x = rJar[r] + al * jvi;
                    if (x < 0) { dp += rD[r] * x * jvi; hp += rD[r] * jvi * jvi; }
                }
                if (c < ncon) {
                    float x[6];
#pragma unroll
                    for (int j = 0; j < 6; j++) x[j] = cx[j] + al * cv[j];
                    ConeOut o;
                    cone_eval2(cdim, cmu, cfri, cD, x, o);
                    float gv = 0, gnv = 0, uv = 0, dwv = 0;
#pragma unroll
                    for (int j = 0; j < 6; j++) { gv += o.g[j] * cv[j]; gnv += o.gn[j] * cv[j]; uv += o.u[j] * cv[j]; dwv += o.dw[j] * cv[j] * cv[j]; }
                    dp += gv; hp += dwv + o.Dm * gnv * gnv - o.k3 * uv * uv;
                }
                dphi = g1 + al * g2 + gsum<G>(dp);
                ddphi = g2 + gsum<G>(hp);
            };
            const float gtol = tol * m.ls_tolerance * snorm / scale;
            float dp, hp, lo = 0, hi = -1;
            ls_eval(0.f, dp, hp);
            const float dp0abs = fabsf(dp);
            // fp32 termination: the Newton decrement -dp/2 predicts the cost decrease without the
            // cancellation of (cost - newcost) between two large fp32 costs
            if (dp >= 0 || hp <= 0 || scale * 0.5f * (-dp) < tol) active = false;
            else {
                alpha = -dp / hp;
                for (int k = 0; k < m.ls_iterations; k++) {
                    ls_eval(alpha, dp, hp);
                    // fp32: the slope cannot be resolved below ~1e-6 of its initial value; MuJoCo's absolute
                    // gtol (tolerance * ls_tolerance * |search| / scale) is kept as the primary criterion
                    if (fabsf(dp) < fmaxf(gtol, 1e-5f * dp0abs)) break;
                    if (dp < 0) lo = alpha; else hi = alpha;
                    float nxt = alpha - dp / hp;
                    if (!(nxt > lo) || (hi > 0 && !(nxt < hi))) nxt = hi > 0 ? 0.5f * (lo + hi) : 2 * alpha;
                    if (nxt == alpha) break;
                    alpha = nxt;
                }
                if (!(alpha > 0)) { active = false; alpha = 0; }
            }
        }
        PHASE(10);
        if (active) vQacc[c] += alpha * search_c;
        __syncthreads();
        const float newcost = eval_at(vQacc);
        if (active) { iter++; cost = newcost; if (!zones_changed) active = false; }
        PHASE(11);
    }
    float qacc_c = vQacc[c], qfc_c = 0;
    if (nefc == 0) qacc_c = qas_c;
    else if (isdof) for (int r = 0; r < nefc; r++) qfc_c -= J[r * RS + c] * rGr[r];
    if (valid && isdof) {
        s.qacc[(size_t)c * N + e] = qacc_c;
        if (debug) { s.qacc_smooth[(size_t)c * N + e] = qas_c; s.qfrc_smooth[(size_t)c * N + e] = qfs_c; s.qfrc_constraint[(size_t)c * N + e] = qfc_c; }
    }
    if (valid && c == 0) { s.ncon[e] = ncon; s.nefc[e] = nefc; s.niter[e] = iter; }
    if (mode == 0) {
        PHASE_FLUSH();
        const float bsum = gsum<G>((float)bad);
        if (valid && c == 0 && bsum > 0) s.bad[e] = 1;
        return;
    }

    PHASE(12);
    // ---------------- phase G: mj_Euler (a-2.7): implicit joint damping, semi-implicit update
    const float h = m.timestep;
    float acc_c = qacc_c;
    if (m.any_damping) {
        float Ar[G];
#pragma unroll
        for (int k = 0; k < G; k++) Ar[k] = Mrow[k] + ((k == c) ? h * damp_c : 0.f);
        float ainvd;
        if (!chol_g<G>(Ar, ainvd, nv, m.ndense, c)) bad = 1;
        acc_c = chol_solve_g<G>(Ar, ainvd, qfs_c + qfc_c, nv, c, tile);
    }
    const float vnew = qvel_c + h * acc_c;
    if (!(fabsf(vnew) <= 1e10f) && isdof) bad = 1;
    __syncthreads();
    vSearch[c] = isdof ? vnew : 0.f;
    __syncthreads();
    if (valid && isdof) {
        s.qvel[(size_t)c * N + e] = vnew;
        s.warm[(size_t)c * N + e] = qacc_c;
        const int t = my_type;
        const int adr = my_qadr;
        if (t == DOF_SLIDE || t == DOF_HINGE || t == DOF_FREE_LIN) s.qpos[(size_t)adr * N + e] = my_q + h * vnew;
        else if (c == my_quat_lane) {
            // first rotational dof of a free joint integrates the quaternion (mju_quatIntegrate)
            const v3 w = mk3(vSearch[c], vSearch[c + 1], vSearch[c + 2]);
            const float wn = norm(w), angle = wn * h;
            if (angle > 0) {
                const v3 ax = w * (1.0f / wn);
                float sn, cs;
                sincosf(0.5f * angle, &sn, &cs);
                q4 q = quat0, qr;
                qr.w = cs; qr.x = ax.x * sn; qr.y = ax.y * sn; qr.z = ax.z * sn;
                q = qnormalized(qmul(q, qr));
                s.qpos[(size_t)adr * N + e] = q.w; s.qpos[(size_t)(adr + 1) * N + e] = q.x; s.qpos[(size_t)(adr + 2) * N + e] = q.y; s.qpos[(size_t)(adr + 3) * N + e] = q.z;
            }
        }
    }
    PHASE(13);
    const float bsum = gsum<G>((float)bad);
    if (valid && c == 0) {
        s.time[e] = time0 + h;
        s.nsteps[e] = nsteps0 + 1;
        if (bsum > 0) s.bad[e] = 1;
        if (reach) s.done[e] = 1;          // a-4: latch; this env skips the remaining substeps of the env-step
    }
    PHASE(14);
    PHASE_FLUSH();
}
