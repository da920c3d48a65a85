// Lane-group primitives of the cooperative solver: one group of G lanes (16 or 32) per env, 64/G envs per single-wave workgroup.
// DPP group reductions / scans / broadcasts, the hand-written v_fmac_f32_dpp, the in-register cooperative Cholesky and the
// elliptic-cone cost / derivatives used by solve_body.inc (mj_makeConstraint, mj_fwdConstraint Newton with elliptic cones, mj_Euler
// behind `self.sim.step()`, hsr/env.py:123; SURVEY.md 8 a-2.2 .. a-2.7), plus the in-kernel phase stamps of the diagnostic build.
#pragma once
#include "devmath.h"
#include "model.h"
#include "solve.h"
#include <type_traits>

#ifndef HSR_LAST_DEC
#define HSR_LAST_DEC 1e-6f      // 100 x the solver tolerance of 1e-8 (solve_body.inc, Newton loop)
#endif
#ifndef HSR_LS_FAR
#define HSR_LS_FAR 1e-4f        // scaled Newton decrement above which the line search stops at HSR_LS_REL_FAR instead of HSR_LS_REL
#endif
#ifndef HSR_LS_REL_FAR
#define HSR_LS_REL_FAR 0.05f
#endif
#ifndef HSR_LS_REL
#define HSR_LS_REL 1e-2f        // relative stop of the exact line search: |phi'(alpha)| < HSR_LS_REL |phi'(0)|  (1e-3: same iteration counts, launch 2.6 % longer)
#endif
#ifndef HSR_SKIN
#define HSR_SKIN 0.01f          // persist.h: the narrowphase item list is built for "closer than this" and kept until a geom may have moved half of it (0: culls every substep)
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
#define ENV_STAMP(slot, e, val) do {} while (0)
// the 32 phase sums live in ONE vector register (lane i = phase i; 32-bit: a launch is < 2^32 cycles) - as scalars they took 64
// SGPRs, which the register allocator spilled into the hot loops
#define PHASE_T0() unsigned long long dc_[4] = {0, 0, 0, 0}; unsigned int pt_v_ = 0; const unsigned int pt_lane_ = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); unsigned long long t_prev_ = stamp_(); const unsigned long long t_first_ = t_prev_, r_first_ = __builtin_amdgcn_s_memrealtime()
// (v_readlane / v_writelane ignore EXEC: a stamp inside divergent code counts whichever lanes are active)
#define PHASE(idx) do { const unsigned long long t_now_ = stamp_(); const unsigned int acc_ = (unsigned int)__builtin_amdgcn_readfirstlane((int)((unsigned int)__builtin_amdgcn_readlane((int)pt_v_, (idx)) + (unsigned int)(t_now_ - t_prev_))); asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(pt_v_) : "s"(acc_), "i"(idx)); t_prev_ = t_now_; } while (0)
#define PHASE_FLUSH() do { if (blockIdx.x < 8192 && pt_lane_ < 32) { unsigned long long *bt_ = s.phase_cyc + 32 + 40 * blockIdx.x; bt_[8 + pt_lane_] = (unsigned long long)pt_v_; if (pt_lane_ < 26) atomicAdd(&s.phase_cyc[pt_lane_], (unsigned long long)pt_v_); } if (tid == 0) { if (blockIdx.x < 8192) { unsigned long long *bt_ = s.phase_cyc + 32 + 40 * blockIdx.x; bt_[4] = dc_[0]; bt_[5] = dc_[1]; bt_[6] = dc_[2]; bt_[7] = dc_[3]; bt_[0] = r_first_; bt_[1] = __builtin_amdgcn_s_memrealtime(); bt_[2] = __builtin_amdgcn_s_getreg(63492); bt_[3] = __builtin_amdgcn_s_getreg(63508); } atomicAdd(&s.phase_cyc[26], stamp_() - t_first_); atomicAdd(&s.phase_cyc[27], __builtin_amdgcn_s_memrealtime() - r_first_); } } while (0)
#elif defined(HSR_BLOCK_LIFE)
// second diagnostic build (libhsrsim_life.so, tools/block_life.py): only the (start, end) stamps of every workgroup and the event
// counters, so the register allocation and the timing of the product kernel are kept
#define PHASE_T0() unsigned long long dc_[4] = {0, 0, 0, 0}; const unsigned long long r_first_ = __builtin_amdgcn_s_memrealtime()
#define DBGCNT(i, v) do { dc_[i] += (v); } while (0)
#define PHASE(idx) do {} while (0)
#define PHASE_FLUSH() do { if (tid == 0 && blockIdx.x < 8192) { unsigned long long *bt_ = s.phase_cyc + 32 + 40 * blockIdx.x; bt_[0] = r_first_; bt_[1] = __builtin_amdgcn_s_memrealtime(); bt_[4] = dc_[0]; bt_[5] = dc_[1]; bt_[6] = dc_[2]; bt_[7] = dc_[3]; } } while (0)
// per-env stamps of the lifetime build (batches of up to 8192 envs, behind the records of 4096 workgroups): [0] when the env's env-step was
// complete (100 MHz clock), [1] the substep at which it was handed over to a solo server (0: never) with the time of the hand-over above bit 16
#define ENV_STAMP(slot, e, val) do { if ((e) < 8192) s.phase_cyc[32 + 40 * 4096 + 8192 * (slot) + (e)] = (val); } while (0)
#else
#define PHASE_T0() do {} while (0)
#define ENV_STAMP(slot, e, val) do {} while (0)
#define DBGCNT(i, v) do {} while (0)
#define PHASE(idx) do {} while (0)
#define PHASE_FLUSH() do {} while (0)
#endif
// the six spare stamps (26..31): the kinematics stages by default; a timing build with -DHSR_SUBPROF=n moves them inside ONE other phase
// (1 the MPR section - the old -DHSR_MPR_PROFILE -, 2 the sphere / box culls, 3 box-box, 4 inertia + bias, 5 constraint assembly,
// 6 the warm-start choice, 7 one Newton iteration's Hessian); tools/block_times.py names them by HSR_SUBPROF
#ifdef HSR_MPR_PROFILE
#define HSR_SUBPROF 1
#endif
#ifndef HSR_SUBPROF
#define HSR_SUBPROF 0
#endif
#define PHASE_S(n, idx) do { if constexpr (HSR_SUBPROF == (n)) { PHASE(idx); } } while (0)
#define PHASE_K(idx) PHASE_S(0, idx)
#define PHASE_M(idx) PHASE_S(1, idx)
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
// Sum over the lanes of the group, bit-identical in every lane: the rotate-and-add butterfly is symmetric (IEEE addition
// commutes), PROVIDED every lane adds the same rounded operands - so the argument is pinned to a register first: without that the
// compiler contracts a product in the argument into the first addition (fma(a, b, neighbour) in one lane, fma(a', b', own) in the
// other), the lanes disagree in the last bit, and a threshold test on the sum can send the lanes of one env down different branches.
template <int G> __device__ __forceinline__ float gsum(float v) {
    asm volatile("" : "+v"(v));
    v += dpp_f<0x128, true>(v); v += dpp_f<0x124, true>(v); v += dpp_f<0x122, true>(v); v += dpp_f<0x121, true>(v);      // every lane: sum of its row
    if constexpr (G == 16) return v;
    else {
        const unsigned u = __builtin_bit_cast(unsigned, v);
        const auto sw = __builtin_amdgcn_permlane16_swap(u, u, false, false);      // [first row sum, second row sum] in every row of the group
        return __builtin_bit_cast(float, (unsigned)sw[0]) + __builtin_bit_cast(float, (unsigned)sw[1]);
    }
}
// Six group sums at once, each delivered to ONE lane: after every halving step two half-reduced quantities share a register (the
// upper half of the lanes carries the second), so the tree costs 6 + 3 + 2 + 1 DPP adds and 5 selects instead of 6 x 4 adds - and no
// lane has to pick its value out of six afterwards.  Lane c of the group's first DPP row ends up with the sum of p[q] over the group,
// q = gsum6_index(c), for c in {0, 8, 4, 12, 2, 10}; the other lanes hold duplicates or partial sums.
__device__ __forceinline__ int gsum6_index(int c) { return ((c >> 3) & 1) + 2 * ((c >> 2) & 1) + 4 * ((c >> 1) & 1); }
template <int G> __device__ __forceinline__ float gsum6_packed(const float (&p)[6], int c) {
    float x[6];
#pragma unroll
    for (int q = 0; q < 6; q++) x[q] = p[q] + dpp_f<0x128, true>(p[q]);                       // lanes i and i + 8 (row_ror:8)
    const bool h8 = c & 8, h4 = c & 4, h2 = c & 2;
    float r0 = h8 ? x[1] : x[0], r1 = h8 ? x[3] : x[2], r2 = h8 ? x[5] : x[4];
    r0 += dpp_f<0x141, true>(r0); r1 += dpp_f<0x141, true>(r1); r2 += dpp_f<0x141, true>(r2);      // lanes i and 7 - i of each half (row_half_mirror)
    float z0 = h4 ? r1 : r0, z1 = r2;
    z0 += dpp_f<0x1B, true>(z0); z1 += dpp_f<0x1B, true>(z1);                                   // lanes i and 3 - i of each quad (quad_perm 3 2 1 0)
    float w = h2 ? z1 : z0;
    w += dpp_f<0xB1, true>(w);                                                                   // neighbours (quad_perm 1 0 3 2)
    if constexpr (G == 32) {
        const unsigned u = __builtin_bit_cast(unsigned, w);
        const auto sw = __builtin_amdgcn_permlane16_swap(u, u, false, false);                    // both DPP rows of the group
        w = __builtin_bit_cast(float, (unsigned)sw[0]) + __builtin_bit_cast(float, (unsigned)sw[1]);
    }
    return w;
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
// A failed pivot (not positive) is not tested for where it happens - the test and its select would sit in the dependent chain of every
// step: rsq turns it into a NaN / infinity that reaches the invd of its own and of every later lane, and the group looks at all invd
// once at the end.
template <int G> __device__ __forceinline__ bool chol_pivots_ok(float invd) {
    const bool badp = !(invd > 0.f && invd < 3.2e7f);          // rsq(1e-15) = 3.16e7: HSR_MINVAL as the smallest pivot accepted
    return gmax<G>(badp ? 1 : 0) == 0;
}
template <int G, int NK = G> __device__ __forceinline__ bool chol_g(float (&row)[G], float &invd, int nv, int ndense, int c) {
    asm volatile("" : "+v"(c));          // lane masks formed where they are used, not hoisted out of the caller's loops as spilled SGPR pairs (chol_g_fwd)
    invd = 1.f;
    static_for<0, NK>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if (j < nv) {
            const float ajj = gbcast_after_asm<G, j>(row[j]);
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
    return chol_pivots_ok<G>(invd);
}
// Factorisation and forward substitution in one sweep: step j also forms y_j = (b_j - sum_{k < j} L[j][k] y_k) / L[j][j] in lane j and
// subtracts L[c][j] y_j from the running right-hand side of every lane c > j - one more multiply and DPP-FMA per step, hidden behind
// the step's trailing updates, instead of a separate 13-step dependent chain afterwards.  Returns the pivot check; y is left in lane c.
template <int G, int NK = G> __device__ __forceinline__ bool chol_g_fwd(float (&row)[G], float &invd, int nv, int c, float b, float &y) {
    invd = 1.f;
    float sacc = b;
    y = 0.f;
    // the lane id is laundered per call: the thirteen `c == j` masks are then formed where they are used (one v_cmp each) instead of being hoisted out of
    // the Newton loop as SGPR pairs, spilled into VGPR lanes and read back with two v_readlane per step
    asm volatile("" : "+v"(c));
    static_for<0, NK>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if (j < nv) {
            const float ajj = gbcast_after_asm<G, j>(row[j]);
            const float inv = __builtin_amdgcn_rsqf(ajj);
            const float lcj = row[j] * inv;
            const float t = sacc * inv;                               // lane j: y_j
            if (c == j) { invd = inv; y = t; }
            row[j] = lcj;
            const float nl = -lcj;
            const BcSrc<G> bl = bc_prepare<G>(lcj);
            static_for<j + 1, NK>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                fmac_bcast<G, i, bc_first<G, i, j + 1>()>(row[i], nl, bl);
            });
            fmac_bcast<G, j, true>(sacc, nl, bc_prepare<G>(t));      // sacc -= L[c][j] y_j (a lane c <= j has taken its y already: what lands in its sacc is never read)
        }
    });
    return chol_pivots_ok<G>(invd);
}
// SPARSE factorisation (round 5, 32-lane instances with several free bodies): H of a robot (dofs 0 .. ND - 1) followed by NB free bodies of six dofs.  `merged` (wave-uniform): every
// env of the wave couples AT MOST ONE body to the robot and no body to another (checked per substep by the caller; 99 % of the env-substeps of cfg4, 95 % of those of its hardest
// tasks); otherwise the bodies' columns are eliminated one by one with all their updates - the dense algorithm, sharing the robot's steps.  Eliminating the robot's columns first then fills nothing outside (robot + that body)^2: the three bodies' blocks stay independent of each other, so after the ND dense steps
// (whose updates still cover every body column - exact zeros for the bodies the robot does not touch; which body it touches is not known at compile time) dof t of EVERY body is
// eliminated in one step: a lane works on its own body's pivot column (its entry selected out of the NB column registers, the pivot fetched with one ds_bpermute whose source lane
// depends on the lane's body), one rsq, and the updates of the body's remaining columns with the multiplier masked per body (a lane of another body must not pick up a product of
// two different bodies' columns in what is ITS lower triangle).  ND + 6 dependent steps and ND (12 + ..) + 6 (19 + 3 t') instructions instead of NK steps of 12 + (NK - 1 - j):
// about 400 instead of 650 for ND = 7, NB = 3 - the factorisation is issue-bound (the arrow experiment), so it is the instruction count that pays.
// Same contract as chol_g_fwd: lane c ends with row c of L in row[0 .. c], invd = 1 / L[c][c], y = the forward substitution L y = b.  fb / kk: the lane's body (-1: robot) and dof in it.
template <int G, int NK, int ND> __device__ __forceinline__ bool chol_sparse_fwd(float (&row)[G], float &invd, int c, float b, float &y, int fb, int kk, bool merged) {
    constexpr int NB = (NK - ND) / 6;
    static_assert(G == 32, "two DPP rows per env: the own-body broadcasts go through ds_bpermute");
    asm volatile("" : "+v"(c));
    asm volatile("" : "+v"(fb));
    invd = 1.f; y = 0.f;
    float sacc = b;
    static_for<0, ND>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const float ajj = gbcast_after_asm<G, j>(row[j]);
        const float inv = __builtin_amdgcn_rsqf(ajj);
        const float lcj = row[j] * inv;
        const float t = sacc * inv;
        if (c == j) { invd = inv; y = t; }
        row[j] = lcj;
        const float nl = -lcj;
        const BcSrc<G> bl = bc_prepare<G>(lcj);
        static_for<j + 1, NK>([&](auto ic) { constexpr int i = decltype(ic)::value; fmac_bcast<G, i, bc_first<G, i, j + 1>()>(row[i], nl, bl); });
        fmac_bcast<G, j, true>(sacc, nl, bc_prepare<G>(t));
    });
    if (!merged) {          // wave-uniform: some env of the wave couples two bodies (to each other, or both to the robot) - the bodies' columns one by one, all updates, as chol_g_fwd does
        static_for<ND, NK>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const float ajj = gbcast_after_asm<G, j>(row[j]);
            const float inv = __builtin_amdgcn_rsqf(ajj);
            const float lcj = row[j] * inv;
            const float t = sacc * inv;
            if (c == j) { invd = inv; y = t; }
            row[j] = lcj;
            const float nl = -lcj;
            const BcSrc<G> bl = bc_prepare<G>(lcj);
            static_for<j + 1, NK>([&](auto ic) { constexpr int i = decltype(ic)::value; fmac_bcast<G, i, bc_first<G, i, j + 1>()>(row[i], nl, bl); });
            fmac_bcast<G, j, true>(sacc, nl, bc_prepare<G>(t));
        });
        return chol_pivots_ok<G>(invd);
    }
    const int fbc = fb < 0 ? 0 : fb;                                     // (a robot lane follows body 0: its registers above the diagonal are never read)
    const int pl0 = 4 * ((threadIdx.x & 32) + ND + 6 * fbc);            // byte address of the lane that holds dof 0 of this lane's body
    static_for<0, 6>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        float mine = row[ND + t];
        static_for<1, NB>([&](auto bc_) { constexpr int bb = decltype(bc_)::value; mine = fb == bb ? row[ND + 6 * bb + t] : mine; });
        const float ajj = __int_as_float(__builtin_amdgcn_ds_bpermute(pl0 + 4 * t, __float_as_int(mine)));
        const float inv = __builtin_amdgcn_rsqf(ajj);
        const float l = mine * inv;
        const float tz = sacc * inv;
        if (kk == t && fb >= 0) { invd = inv; y = tz; }
        static_for<0, NB>([&](auto bc_) { constexpr int bb = decltype(bc_)::value; row[ND + 6 * bb + t] = fbc == bb ? l : row[ND + 6 * bb + t]; });
        if constexpr (t < 5) {
            float nlb[NB];
            static_for<0, NB>([&](auto bc_) { constexpr int bb = decltype(bc_)::value; nlb[bb] = fb == bb ? -l : 0.f; });
            const BcSrc<G> bl = bc_prepare<G>(l);
            static_for<0, NB>([&](auto bc_) {
                constexpr int bb = decltype(bc_)::value;
                static_for<t + 1, 6>([&](auto sc) { constexpr int k = ND + 6 * bb + decltype(sc)::value; fmac_bcast<G, k, (bb == 0 && k == ND + t + 1) || k == 16>(row[k], nlb[bb], bl); });
            });
        }
        const float zt = __int_as_float(__builtin_amdgcn_ds_bpermute(pl0 + 4 * t, __float_as_int(tz)));
        sacc = __builtin_fmaf(-l, zt, sacc);                             // sacc -= L[c][j] y_j, j = this lane's body's dof t
    });
    return chol_pivots_ok<G>(invd);
}
// ... and L^T x = y for such a factor, last dof first: the bodies side by side, two dofs per reduction latency (a body's column sums over the body's own later rows only - every other lane
// holds an exact zero there), then the robot's columns as in chol_back_mf (they sum over the robot's later rows and the coupled body's).
template <int G, int NK, int ND> __device__ __forceinline__ float chol_sparse_back(const float (&row)[G], float invd, float y, int c, bool merged) {
    constexpr int NB = (NK - ND) / 6;
    asm volatile("" : "+v"(c));
    float nlo[G];
#pragma unroll
    for (int k = 0; k < NK; k++) nlo[k] = (k < c) ? -row[k] : 0.f;
    float x = 0.f;
    if (!merged) {          // the bodies' columns pair by pair, last first (chol_back_mf's order)
        static_for<0, (NK - ND) / 2>([&](auto jc) {
            constexpr int j = NK - 1 - 2 * decltype(jc)::value;
            const float A = gsum<G>(nlo[j] * x), B = gsum<G>(nlo[j - 1] * x);
            const float xj = (y + A) * invd;
            if (c == j) x = xj;
            const float t = gbcast<G, j>(nlo[j - 1] * xj);
            if (c == j - 1) x = (y + B + t) * invd;
        });
    } else
    static_for<0, 3>([&](auto sc) {
        constexpr int s1 = 5 - 2 * decltype(sc)::value;                  // dofs s1 and s1 - 1 of every body
        float A[NB > 0 ? NB : 1], B[NB > 0 ? NB : 1];
        static_for<0, NB>([&](auto bc_) { constexpr int bb = decltype(bc_)::value, j = ND + 6 * bb + s1; A[bb] = gsum<G>(nlo[j] * x); B[bb] = gsum<G>(nlo[j - 1] * x); });
        float xn = x;
        static_for<0, NB>([&](auto bc_) {
            constexpr int bb = decltype(bc_)::value, j = ND + 6 * bb + s1;
            const float xj = (y + A[bb]) * invd;
            if (c == j) xn = xj;
            const float t = gbcast<G, j>(nlo[j - 1] * xj);
            if (c == j - 1) xn = (y + B[bb] + t) * invd;
        });
        x = xn;
    });
    static_for<0, ND / 2>([&](auto jc) {
        constexpr int j = ND - 1 - 2 * decltype(jc)::value;              // columns j and j - 1
        const float A = gsum<G>(nlo[j] * x), B = gsum<G>(nlo[j - 1] * x);
        const float xj = (y + A) * invd;
        if (c == j) x = xj;
        const float t = gbcast<G, j>(nlo[j - 1] * xj);
        if (c == j - 1) x = (y + B + t) * invd;
    });
    if constexpr (ND % 2 == 1) { const float tot = gsum<G>(nlo[0] * x); if (c == 0) x = (y + tot) * invd; }
    return x;
}
// Accumulator of v_mfma_f32_16x16x1f32 (4 blocks of 16x16, K = 1; layouts measured with tools/micro/mfma_layout.hip): operand A / B of lane l
// is row / column l % 16 of block l / 16; result register v of lane l is block v / 4, row 4 (l / 16) + v % 4, column l % 16.  add_rows:
// the 4x4 transposition of (lane row, register group) - eight v_permlane32_swap, eight v_permlane16_swap - leaves block b in the 16
// lanes of row b with register k = matrix row k, lane = column: lane c then holds sum_r A_r[k] B_r[c], k = 0..15.
typedef float hess_v16f __attribute__((ext_vector_type(16)));
struct HessAcc {
    hess_v16f v;
    __device__ __forceinline__ void clear() { v = hess_v16f{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; }
    template <int NK, int G> __device__ __forceinline__ void add_rows(float (&row)[G]) {
        unsigned r[16];
#pragma unroll
        for (int i = 0; i < 16; i++) { const float f = v[i]; r[i] = __float_as_uint(f); }      // (bit_cast of a vector element miscompiles: all lanes of element 0)
#pragma unroll
        for (int g = 0; g < 2; g++)
#pragma unroll
            for (int i = 0; i < 4; i++) { const auto sw = __builtin_amdgcn_permlane32_swap(r[4 * g + i], r[4 * (g + 2) + i], false, false); r[4 * g + i] = sw[0]; r[4 * (g + 2) + i] = sw[1]; }
#pragma unroll
        for (int g = 0; g < 4; g += 2)
#pragma unroll
            for (int i = 0; i < 4; i++) { const auto sw = __builtin_amdgcn_permlane16_swap(r[4 * g + i], r[4 * (g + 1) + i], false, false); r[4 * g + i] = sw[0]; r[4 * (g + 1) + i] = sw[1]; }
#pragma unroll
        for (int k = 0; k < NK; k++) row[k] += __uint_as_float(r[k]);
    }
};
// The 32-lane twin: v_mfma_f32_32x32x1f32 (2 blocks of 32x32; tools/micro/mfma_layout32.hip): operand A / B of lane l is row / column l % 32 of
// block l / 32; result register v of lane l is block v / 16, row 8 ((v % 16) / 4) + 4 (l / 32) + v % 4, column l % 32.  Sixteen
// v_permlane32_swap (register t of the upper half <-> register 16 + t of the lower half) leave block b in the 32 lanes of half b, matrix
// row i in register ((i / 4) % 2 ? 16 : 0) + 4 (i / 8) + i % 4.
typedef float hess_v32f __attribute__((ext_vector_type(32)));
struct HessAcc32 {
    hess_v32f v;
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int i = 0; i < 32; i++) v[i] = 0.f;
    }
    template <int NK, int G> __device__ __forceinline__ void add_rows(float (&row)[G]) {
        unsigned r[32];
#pragma unroll
        for (int i = 0; i < 32; i++) { const float f = v[i]; r[i] = __float_as_uint(f); }
#pragma unroll
        for (int t = 0; t < 16; t++) { const auto sw = __builtin_amdgcn_permlane32_swap(r[t], r[16 + t], false, false); r[t] = sw[0]; r[16 + t] = sw[1]; }
#pragma unroll
        for (int k = 0; k < NK; k++) row[k] += __uint_as_float(r[((k / 4) % 2 ? 16 : 0) + 4 * (k / 8) + k % 4]);
    }
};
// The same factorisation for a matrix whose columns j >= ND (compile time) have no off-diagonal entries at all - the inertia
// matrix M and M + h D of a robot followed by free bodies with principal-axis inertia: ND pivot steps with updates of the first ND
// rows only, and every tail lane takes the reciprocal root of its own diagonal entry (diag: lane c's M[c][c]; 1 for the padding lanes).
template <int G, int NK, int ND> __device__ __forceinline__ bool chol_g_tail(float (&row)[G], float &invd, float diag, int c) {
    asm volatile("" : "+v"(c));          // lane masks formed where they are used, not hoisted out of the caller's loops as spilled SGPR pairs (chol_g_fwd)
    invd = 1.f;
    static_for<0, ND>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const float ajj = gbcast_after_asm<G, j>(row[j]);
        const float inv = __builtin_amdgcn_rsqf(ajj);
        const float lcj = row[j] * inv;
        if (c == j) invd = inv;
        row[j] = lcj;
        const float nl = -lcj;
        const BcSrc<G> bl = bc_prepare<G>(lcj);
        static_for<j + 1, ND>([&](auto ic) { constexpr int i = decltype(ic)::value; fmac_bcast<G, i, bc_first<G, i, j + 1>()>(row[i], nl, bl); });
    });
    if (c >= ND) invd = __builtin_amdgcn_rsqf(diag);
    return chol_pivots_ok<G>(invd);
}
// elliptic cone at residual x: cost, gradient g, and the Hessian in the form
//   H = diag(dw) + Dm gn gn^T - k3 u u^T      (zone 0 top: all zero; 1 bottom: dw = D; 2 middle)
// The three cone functions take their six rows ZERO-PADDED: D[j] = x[j] = v[j] = 0 and fri[j - 1] = 0 for the rows j >= dim a contact
// does not have (cone_rows / cone_fri in solve_body.inc) - every formula then yields 0 for them by itself, and no per-row test on dim
// (six exec-mask branches per call) is needed.
struct ConeOut { float cost, Dm, k3; int zone; float g[6], dw[6], gn[6], u[6]; };
__device__ __forceinline__ void cone_eval2(int /*dim*/, float mu, const float *fri, const float *D, const float *x, ConeOut &o) {
    float U[6], T2 = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) { o.g[j] = 0; o.dw[j] = 0; o.gn[j] = 0; o.u[j] = 0; }
    o.cost = 0; o.Dm = 0; o.k3 = 0; o.zone = 0;
    U[0] = x[0] * mu;
    const float Nn = U[0];
#pragma unroll
    for (int j = 1; j < 6; j++) { U[j] = x[j] * fri[j - 1]; T2 += U[j] * U[j]; }
    const float T = fsqrt(T2);
    if (Nn >= mu * T || (T <= 0 && Nn >= 0)) return;
    if (mu * Nn + T <= 0 || (T <= 0 && Nn < 0)) {
        o.zone = 1;
#pragma unroll
        for (int j = 0; j < 6; j++) { o.cost += 0.5f * D[j] * x[j] * x[j]; o.g[j] = D[j] * x[j]; o.dw[j] = D[j]; }
        return;
    }
    o.zone = 2;
    const float Dm = D[0] * frcp(mu * mu * (1 + mu * mu)), NT = Nn - mu * T, invT = frcp(T);
    const float kappa = -Dm * NT * mu;
    o.Dm = Dm; o.k3 = kappa * invT * frcp(T2);
    o.gn[0] = mu;
#pragma unroll
    for (int j = 1; j < 6; j++) {
        o.gn[j] = -mu * U[j] * fri[j - 1] * invT;
        o.u[j] = fri[j - 1] * U[j];
        o.dw[j] = kappa * fri[j - 1] * fri[j - 1] * invT;
    }
#pragma unroll
    for (int j = 0; j < 6; j++) o.g[j] = Dm * NT * o.gn[j];
    o.cost = 0.5f * Dm * NT * NT;
}

// cost only (the warm-start comparison needs nothing else at the unconstrained point)
__device__ __forceinline__ float cone_cost(int /*dim*/, float mu, const float *fri, const float *D, const float *x) {
    const float Nn = x[0] * mu;
    float T2 = 0;
#pragma unroll
    for (int j = 1; j < 6; j++) { const float Uj = x[j] * fri[j - 1]; T2 += Uj * Uj; }
    const float T = fsqrt(T2);
    if (Nn >= mu * T || (T <= 0 && Nn >= 0)) return 0.f;
    if (mu * Nn + T <= 0 || (T <= 0 && Nn < 0)) {
        float cst = 0;
#pragma unroll
        for (int j = 0; j < 6; j++) cst += 0.5f * D[j] * x[j] * x[j];
        return cst;
    }
    const float Dm = D[0] * frcp(mu * mu * (1 + mu * mu)), NT = Nn - mu * T;
    return 0.5f * Dm * NT * NT;
}
// first and second derivative along v of the elliptic-cone cost at residual x (what the line search needs): the same
// zones and formulas as cone_eval2 contracted with v analytically, d1 = g . v, d2 = v^T (diag(dw) + Dm gn gn^T - k3 u u^T) v
__device__ __forceinline__ void cone_dd(int /*dim*/, float mu, const float *fri, const float *D, const float *x, const float *v, float &d1, float &d2) {
    d1 = 0; d2 = 0;
    const float Nn = x[0] * mu;
    float T2 = 0, S1 = 0, S2 = 0;
#pragma unroll
    for (int j = 1; j < 6; j++) {
        const float f = fri[j - 1], Uj = x[j] * f, fv = f * v[j];
        T2 += Uj * Uj; S1 += Uj * fv; S2 += fv * fv;
    }
    const float T = fsqrt(T2);
    if (Nn >= mu * T || (T <= 0 && Nn >= 0)) return;
    if (mu * Nn + T <= 0 || (T <= 0 && Nn < 0)) {
#pragma unroll
        for (int j = 0; j < 6; j++) { d1 += D[j] * x[j] * v[j]; d2 += D[j] * v[j] * v[j]; }
        return;
    }
    const float Dm = D[0] * frcp(mu * mu * (1 + mu * mu)), NT = Nn - mu * T, invT = frcp(T);
    const float kappa = -Dm * NT * mu, gnv = mu * (v[0] - invT * S1);
    d1 = Dm * NT * gnv;
    d2 = kappa * invT * S2 + Dm * gnv * gnv - (kappa * invT * frcp(T2)) * S1 * S1;
}

enum { NLMAX = 16 };
// does the kernel instance of model type T need mj_makeImpedance's powf arms?  The run-time DevModel: yes; a constant instance says (cfg_consts.h)
template <class T, class = void> struct SolimpGeneral { static constexpr bool value = true; };
template <class T> struct SolimpGeneral<T, std::enable_if_t<!std::is_same<T, DevModel>::value>> { static constexpr bool value = T::solimp_general != 0; };
