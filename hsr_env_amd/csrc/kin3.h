// Kinematics, velocity recursion, joint-space inertia and bias force of the persistent kernel with the TREE known at compile time
// (mj_kinematics, mj_comPos, mj_crb, mj_rne behind `self.sim.step()`, hsr/env.py:123): the instances of the
// reference configurations carry their kinematic tree as constexpr tables (cfg_consts.h: Kin3_<cfg>, generated from the blobs and matched
// value for value against the loaded model at hsr_batch_create).  What kin2.h + the inertia phase of solve_body.inc do with table-driven
// lane roles (lane = link, lane = dof; ancestor lists, dof-type selects, 60-float link records and 15-float link results in LDS, ~2.7 k
// instructions and a dozen LDS round trips per substep) is here ONE straight-line pass that every lane of the env runs on the same values:
//   * the scalar-joint chain (the robot): link poses, dof axes, link velocities and bias accelerations (the recursion of kin_link_pose /
//     kin_link_dyn, collide.h, formula for formula), the link wrenches, and - while a link's quantities are in registers - their
//     contributions to V, U, T and the bias force of every dof above it (the sums of solve_body.inc's inertia phase); then the entries
//     M[c][k] of the robot's block.  Constants of the tree are literals: a multiplication by a literal 0 or +-1 and an addition of a literal 0
//     are dropped where they are written (cm / ca / cs below), so the identity orientations, axis-aligned joint axes, the zero angular
//     velocity of the sliding base ... cost nothing.  One lane stores the poses, the dof records and the matrix rows to LDS, every dof lane
//     reads its own.
//   * the free bodies (the blocks) lane-parallel: a lane computes the body its dof belongs to (pose from its seven coordinates, axes, the
//     gyroscopic bias in the body frame, the constant diagonal of M).
// Same quantities as kin2.h + solve_body.inc produce, different rounding only (tests: forward stages against the oracle; the constant
// instances against the generic ones to tolerance).
#pragma once
#include "kin2.h"
#include "cfg_consts.h"

namespace k3 {
// products and sums that vanish at compile time: __builtin_constant_p is resolved after inlining, when the tree constants have arrived
__device__ __forceinline__ float cm(float k, float x) {
    if (__builtin_constant_p(k)) { if (k == 0.f) return 0.f; if (k == 1.f) return x; if (k == -1.f) return -x; }
    if (__builtin_constant_p(x)) { if (x == 0.f) return 0.f; if (x == 1.f) return k; if (x == -1.f) return -k; }
    return k * x;
}
__device__ __forceinline__ float ca(float a, float b) {
    if (__builtin_constant_p(a) && a == 0.f) return b;
    if (__builtin_constant_p(b) && b == 0.f) return a;
    return a + b;
}
__device__ __forceinline__ float cs(float a, float b) {
    if (__builtin_constant_p(b) && b == 0.f) return a;
    if (__builtin_constant_p(a) && a == 0.f) return -b;
    return a - b;
}
__device__ __forceinline__ v3 k_add(v3 a, v3 b) { return mk3(ca(a.x, b.x), ca(a.y, b.y), ca(a.z, b.z)); }
__device__ __forceinline__ v3 k_sub(v3 a, v3 b) { return mk3(cs(a.x, b.x), cs(a.y, b.y), cs(a.z, b.z)); }
__device__ __forceinline__ v3 k_scl(v3 a, float s) { return mk3(cm(a.x, s), cm(a.y, s), cm(a.z, s)); }
__device__ __forceinline__ float k_dot(v3 a, v3 b) { return ca(ca(cm(a.x, b.x), cm(a.y, b.y)), cm(a.z, b.z)); }
__device__ __forceinline__ v3 k_cross(v3 a, v3 b) {
    return mk3(cs(cm(a.y, b.z), cm(a.z, b.y)), cs(cm(a.z, b.x), cm(a.x, b.z)), cs(cm(a.x, b.y), cm(a.y, b.x)));
}
__device__ __forceinline__ v3 k_mulmv(const m3 &m, v3 v) {
    return mk3(ca(ca(cm(m.a[0], v.x), cm(m.a[1], v.y)), cm(m.a[2], v.z)), ca(ca(cm(m.a[3], v.x), cm(m.a[4], v.y)), cm(m.a[5], v.z)),
               ca(ca(cm(m.a[6], v.x), cm(m.a[7], v.y)), cm(m.a[8], v.z)));
}
__device__ __forceinline__ m3 k_mulmm(const m3 &a, const m3 &b) {
    m3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.a[3 * i + j] = ca(ca(cm(a.a[3 * i], b.a[j]), cm(a.a[3 * i + 1], b.a[3 + j])), cm(a.a[3 * i + 2], b.a[6 + j]));
    return r;
}
__device__ __forceinline__ m3 k_transpose(const m3 &a) {
    m3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.a[3 * i + j] = a.a[3 * j + i];
    return r;
}
__device__ __forceinline__ m3 k_q2m(float w, float x, float y, float z) {
    m3 m;
    m.a[0] = cs(1.f, cm(2.f, ca(cm(y, y), cm(z, z)))); m.a[1] = cm(2.f, cs(cm(x, y), cm(w, z))); m.a[2] = cm(2.f, ca(cm(x, z), cm(w, y)));
    m.a[3] = cm(2.f, ca(cm(x, y), cm(w, z))); m.a[4] = cs(1.f, cm(2.f, ca(cm(x, x), cm(z, z)))); m.a[5] = cm(2.f, cs(cm(y, z), cm(w, x)));
    m.a[6] = cm(2.f, cs(cm(x, z), cm(w, y))); m.a[7] = cm(2.f, ca(cm(y, z), cm(w, x))); m.a[8] = cs(1.f, cm(2.f, ca(cm(x, x), cm(y, y))));
    return m;
}
template <class KD> constexpr bool link_anc_or_self(int a, int l) {      // is link a an ancestor of link l, or l itself
    while (l > 0 && l != a) l = KD::parent[l];
    return l == a && a > 0;
}
template <class KD> constexpr bool dof_above_link(int k, int l) { return link_anc_or_self<KD>(KD::dlink[k], l); }      // dof k moves link l
template <class KD> constexpr bool dof_anc_or_self(int k, int c) {      // dof k is dof c or acts above it in the tree (the entries M[c][k] that exist)
    if (KD::dlink[k] == KD::dlink[c]) return k <= c;
    return link_anc_or_self<KD>(KD::dlink[k], KD::dlink[c]);
}
}  // namespace k3

// the tree tables of a model type (DevModel itself has none)
template <class T, class = void> struct Kin3Of { using type = Kin3_none; };
template <class T> struct Kin3Of<T, std::void_t<typename T::kin3>> { using type = typename T::kin3; };

// staging area in LDS (the row-scalar region of the env: free from the end of a substep until the solver forms its rows, which is after it
// has fetched this): 12 floats per dof (axis 3, slide direction 3, anchor 3, bias force, M[c][c], 1 unused), then the robot's block of M, one
// row of RS floats per robot dof.  It is written during the kinematics and read at the top of the solve (kin3_fetch) - as registers the 24
// values per lane would stay alive across the collision phase, where the register pressure peaks
template <class KD> struct Kin3Stage { static constexpr int RS = (KD::NRD + 3) & ~3, oM = 12 * KD::NV, total = oM + RS * KD::NRD; };

// what lane c needs of it all: its dof's world axis a, slide direction l, anchor n, bias force, row c of M (k < NK) and M[c][c]
template <int G> struct Kin3Lane { v3 a, l, n; float bias, Mdiag; float Mrow[G]; };

template <class KD, int G, int NK>
__device__ __forceinline__ void kin3_run(int c, float gravz, float *qposL, const float *qvelL, float *poseL, float *recL, float *st, float *lkL, const float *link_com, const float *link_inertia, const float *link_mass) {
    using namespace k3;
    constexpr int NL = KD::NL, NV = KD::NV, RD0 = KD::RD0, NRD = KD::NRD;
    typedef Kin3Stage<KD> ST;
#define K3C3(T, I) mk3(KD::T[I][0], KD::T[I][1], KD::T[I][2])      /* element reads of constexpr tables: constant expressions, no device-side copy of the table */
    // ---- free bodies, lane-parallel: the body of this lane's dof (a lane without one follows the first body; its results are not stored)
    if constexpr (KD::NFREE > 0) {
        bool isfree_lane = false;
        int fl_ = KD::F0, fq = KD::qadr[KD::F0], fd = KD::dofadr[KD::F0];
        static_for<1, NL>([&](auto lc) {
            constexpr int l = decltype(lc)::value;
            if constexpr (KD::isfree[l]) {
                const bool mine = c >= KD::dofadr[l] && c < KD::dofadr[l] + 6;
                fl_ = mine ? l : fl_; fq = mine ? KD::qadr[l] : fq; fd = mine ? KD::dofadr[l] : fd;
                isfree_lane = isfree_lane || mine;
            }
        });
        const int j = c - fd;
        float q7[7], v6[6];
#pragma unroll
        for (int i = 0; i < 7; i++) q7[i] = qposL[fq + i];
#pragma unroll
        for (int i = 0; i < 6; i++) v6[i] = qvelL[fd + i];
        q4 q;
        q.w = q7[3]; q.x = q7[4]; q.y = q7[5]; q.z = q7[6];
        q = qnormalized(q);                                      // mj_kinematics normalises in place
        const m3 Rf = ::q2m(q);
        const v3 pf = mk3(q7[0], q7[1], q7[2]);
        const v3 om = mk3(v6[3], v6[4], v6[5]);                  // angular velocity in the body frame
        const v3 wf = ::mulmv(Rf, om);
        constexpr float I0 = KD::linr[KD::F0][0], I1 = KD::linr[KD::F0][1], I2 = KD::linr[KD::F0][2], fm = KD::lmass[KD::F0];
        // bias force of the body's dofs: its weight on the translations; on the rotations (about the body axes) the gyroscopic torque
        // a_j . (w x I w) = (om x I_body om)_j
        const v3 gy = ::cross(om, mk3(I0 * om.x, I1 * om.y, I2 * om.z));
        const int jj = j < 3 ? j : j - 3;
        // picks by the lane's dof index: every candidate is an opaque register copy - a chain of selects over the entries of one object comes back
        // from the optimiser as ONE load with a computed index, and an object indexed that way lives in scratch memory (the box-box lesson, round 4)
        auto opq = [](float x) { asm volatile("" : "+v"(x)); return x; };
        const bool j0 = jj == 0, j1 = jj == 1;
        auto pick = [&](float a, float b, float cc) { const float a_ = opq(a), b_ = opq(b), c_ = opq(cc); return j0 ? a_ : (j1 ? b_ : c_); };
        const v3 fl = j < 3 ? mk3(j0 ? 1.f : 0.f, j1 ? 1.f : 0.f, (!j0 && !j1) ? 1.f : 0.f) : mk3(0, 0, 0);
        const v3 fa_ = mk3(pick(Rf.a[0], Rf.a[1], Rf.a[2]), pick(Rf.a[3], Rf.a[4], Rf.a[5]), pick(Rf.a[6], Rf.a[7], Rf.a[8]));
        const v3 fa = j < 3 ? mk3(0, 0, 0) : fa_;
        const float fbias = j < 3 ? ((!j0 && !j1) ? -gravz * fm : 0.f) : pick(gy.x, gy.y, gy.z);
        const float fdiag = j < 3 ? fm : pick(I0, I1, I2);
        if (isfree_lane && j == 0) {
            qposL[fq + 3] = q.w; qposL[fq + 4] = q.x; qposL[fq + 5] = q.y; qposL[fq + 6] = q.z;
            pose_store(poseL + 12 * fl_, Rf, pf);
            ks4(recL + 12 * fl_, make_float4(wf.x, wf.y, wf.z, v6[0])); ks4(recL + 12 * fl_ + 4, make_float4(v6[1], v6[2], 0.f, 0.f));
        }
        if (isfree_lane) {
            float *o = st + 12 * c;
            ks4(o, make_float4(fa.x, fa.y, fa.z, fl.x)); ks4(o + 4, make_float4(fl.y, fl.z, pf.x, pf.y)); ks4(o + 8, make_float4(pf.z, fbias, fdiag, 0.f));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- robot: coordinates and velocities of its scalar joints (every lane reads all of them: broadcast reads)
    const bool lane0 = c == 0;
    // lane = link for the link dynamics below: the constants of this lane's link are fetched now (global memory, L2-resident) and used after the recursion
    const int lc = c < NL ? c : 0;          // (the fetch below is dead code in the instances without LANE_DYN)
    typedef const __attribute__((address_space(1))) float *gcf_;
    const gcf_ gli = (gcf_)link_inertia + 6 * lc, gcm = (gcf_)link_com + 3 * lc;
    const float kI[6] = {gli[0], gli[1], gli[2], gli[3], gli[4], gli[5]};
    const float kcom[3] = {gcm[0], gcm[1], gcm[2]}, kmass = ((gcf_)link_mass)[lc];
    float Q[NRD], QD[NRD];
    static_for<0, NRD>([&](auto kc) { constexpr int k = decltype(kc)::value; Q[k] = qposL[KD::dqadr[RD0 + k]]; QD[k] = qvelL[RD0 + k]; });
    m3 Rw[NL];
    v3 Pw[NL], Ww[NL], VOw[NL], ALw[NL], AOw[NL];
    v3 ANG[NRD], LIN[NRD], ANC[NRD], Vc[NRD], TUc[NRD];
    float Bc[NRD];
    static_for<0, NRD>([&](auto kc) { constexpr int k = decltype(kc)::value; Vc[k] = mk3(0, 0, 0); TUc[k] = mk3(0, 0, 0); Bc[k] = 0.f; });
#pragma unroll
    for (int i = 0; i < 9; i++) Rw[0].a[i] = (i % 4 == 0) ? 1.f : 0.f;
    Pw[0] = mk3(0, 0, 0); Ww[0] = mk3(0, 0, 0); VOw[0] = mk3(0, 0, 0); ALw[0] = mk3(0, 0, 0); AOw[0] = mk3(0, 0, 0);
    if (lane0) { pose_store(poseL, Rw[0], Pw[0]); ks4(recL, make_float4(0, 0, 0, 0)); ks4(recL + 4, make_float4(0, 0, 0, 0)); }
    // LANE_DYN: the chain has hinges - the links' world inertias and wrenches are worth a lane-parallel pass (lane = link, through LDS); a chain of slides only (cfg1 / cfg2:
    // one link, identity orientation) folds to constants in the redundant form, which is then cheaper than the detour
    constexpr bool LANE_DYN = [] { for (int k = 0; k < NRD; k++) if (KD::dtype[RD0 + k] == DOF_HINGE) return true; return false; }();
    // what link l adds to the sums of every dof that moves it (solve_body.inc, inertia phase):
    //   V = sum m jp_c(l), TU = sum I a_c + (com - n_c) x (m jp_c(l)), bias = sum jp_c(l) . F + a_c . N,   jp_c(l) = lin_c + a_c x (com - n_c)
    auto accumulate = [&](auto lc_, const v3 com, const m3 &I, const v3 F, const v3 Nt) {
        constexpr int l = decltype(lc_)::value;
        static_for<0, NRD>([&](auto kc) {
            constexpr int kr = decltype(kc)::value;
            if constexpr (dof_above_link<KD>(RD0 + kr, l)) {
                const v3 rr = k_sub(com, ANC[kr]);
                const v3 jpc = k_add(LIN[kr], k_cross(ANG[kr], rr));
                const v3 v = k_scl(jpc, KD::lmass[l]);
                Bc[kr] = ca(Bc[kr], ca(k_dot(jpc, F), k_dot(ANG[kr], Nt)));
                Vc[kr] = k_add(Vc[kr], v); TUc[kr] = k_add(TUc[kr], k_add(k_mulmv(I, ANG[kr]), k_cross(rr, v)));
            }
        });
    };
    static_for<1, NL>([&](auto lc) {
        constexpr int l = decltype(lc)::value;
        if constexpr (!KD::isfree[l]) {
            constexpr int p = KD::parent[l], d0 = KD::dofadr[l], dn = KD::dofnum[l];
            // pose (kin_link_pose)
            m3 lm;
#pragma unroll
            for (int i = 0; i < 9; i++) lm.a[i] = KD::lmat[l][i];
            v3 pos = k_add(Pw[p], k_mulmv(Rw[p], K3C3(lpos, l)));
            m3 mat = k_mulmm(Rw[p], lm);
            static_for<0, dn>([&](auto jc) {
                constexpr int k = d0 + decltype(jc)::value;
                const float q = Q[k - RD0];
                const v3 ax = K3C3(daxis, k);
                if constexpr (KD::dtype[k] == DOF_SLIDE) pos = k_add(pos, k_scl(k_mulmv(mat, ax), q));
                else {
                    const v3 jp = K3C3(dpos, k);
                    const v3 anchor = k_add(pos, k_mulmv(mat, jp));
                    float sn, cs_;
                    fast_sincos(0.5f * q, &sn, &cs_);
                    mat = k_mulmm(mat, k_q2m(cs_, cm(ax.x, sn), cm(ax.y, sn), cm(ax.z, sn)));
                    pos = k_sub(anchor, k_mulmv(mat, jp));
                }
            });
            static_for<0, dn>([&](auto jc) {
                constexpr int k = d0 + decltype(jc)::value, kr = k - RD0;
                const v3 ax = k_mulmv(mat, K3C3(daxis, k));
                if constexpr (KD::dtype[k] == DOF_SLIDE) { LIN[kr] = ax; ANG[kr] = mk3(0, 0, 0); ANC[kr] = pos; }
                else { ANG[kr] = ax; LIN[kr] = mk3(0, 0, 0); ANC[kr] = k_add(pos, k_mulmv(mat, K3C3(dpos, k))); }
            });
            Rw[l] = mat; Pw[l] = pos;
            // velocity and bias acceleration of the link origin (kin_link_dyn)
            const v3 wp = Ww[p], vop = VOw[p], alp = ALw[p], aop = AOw[p];
            const v3 r = k_sub(pos, Pw[p]);
            v3 w = wp, al = alp;
            v3 vo = k_add(vop, k_cross(wp, r));
            v3 ao = k_add(k_add(aop, k_cross(alp, r)), k_cross(wp, k_cross(wp, r)));
            static_for<0, dn>([&](auto jc) {
                constexpr int k = d0 + decltype(jc)::value, kr = k - RD0;
                const float qd = QD[kr];
                if constexpr (KD::dtype[k] == DOF_SLIDE) {
                    vo = k_add(vo, k_scl(LIN[kr], qd));
                    ao = k_add(ao, k_scl(k_cross(wp, LIN[kr]), cm(2.f, qd)));
                } else {
                    const v3 a = ANG[kr];
                    const v3 rho = k_sub(pos, ANC[kr]), rc = k_sub(r, rho);
                    const v3 wl = k_add(w, k_scl(a, qd)), all = k_add(al, k_scl(k_cross(w, a), qd));
                    const v3 ac = k_add(k_add(aop, k_cross(alp, rc)), k_cross(wp, k_cross(wp, rc)));
                    const v3 vc = k_add(vop, k_cross(wp, rc));
                    ao = k_add(k_add(ac, k_cross(all, rho)), k_cross(wl, k_cross(wl, rho)));
                    vo = k_add(vc, k_cross(wl, rho));
                    w = wl; al = all;
                }
            });
            Ww[l] = w; VOw[l] = vo; ALw[l] = al; AOw[l] = ao;
            if (lane0) {      // one lane of the env stores the link's pose and velocity state (geom placement, goal test, outputs; the link dynamics below)
                pose_store(poseL + 12 * l, mat, pos);
                ks4(recL + 12 * l, make_float4(w.x, w.y, w.z, vo.x)); ks4(recL + 12 * l + 4, make_float4(vo.y, vo.z, al.x, al.y)); ks4(recL + 12 * l + 8, make_float4(al.z, ao.x, ao.y, ao.z));
            }
            if constexpr (!LANE_DYN) {      // world inertia, com and the wrench of mj_rne, F = m (a(com) - gravity), N = I al + w x I w: every lane, constants folded
                m3 Il;
                Il.a[0] = KD::linr[l][0]; Il.a[1] = KD::linr[l][3]; Il.a[2] = KD::linr[l][4]; Il.a[3] = KD::linr[l][3]; Il.a[4] = KD::linr[l][1]; Il.a[5] = KD::linr[l][5];
                Il.a[6] = KD::linr[l][4]; Il.a[7] = KD::linr[l][5]; Il.a[8] = KD::linr[l][2];
                const m3 I = k_mulmm(k_mulmm(mat, Il), k_transpose(mat));
                const v3 rcm = k_mulmv(mat, K3C3(lcom, l)), com = k_add(pos, rcm);
                const v3 acom = k_add(k_add(ao, k_cross(al, rcm)), k_cross(w, k_cross(w, rcm)));
                const v3 F = k_scl(k_sub(acom, mk3(0.f, 0.f, gravz)), KD::lmass[l]);
                const v3 Nt = k_add(k_mulmv(I, al), k_cross(w, k_mulmv(I, w)));
                accumulate(lc, com, I, F, Nt);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    });
    if constexpr (LANE_DYN) {
    wave_sync();
    // ---- link dynamics, lane = link (every link at once): world inertia, com and the wrench of mj_rne, F = m (a(com) - gravity), N = I al + w x I w, from the
    // link's pose and velocity state (LDS) and its constants (fetched above); 16 floats per link: com 3, I (xx yy zz xy xz yz), F 3, N 3, -
    {
        m3 Rl; v3 pl;
        pose_load(poseL + 12 * lc, Rl, pl);
        const float4 r0 = kl4(recL + 12 * lc), r1 = kl4(recL + 12 * lc + 4), r2 = kl4(recL + 12 * lc + 8);
        const v3 wl = mk3(r0.x, r0.y, r0.z), all = mk3(r1.z, r1.w, r2.x), aol = mk3(r2.y, r2.z, r2.w);
        m3 Il;
        Il.a[0] = kI[0]; Il.a[1] = kI[3]; Il.a[2] = kI[4]; Il.a[3] = kI[3]; Il.a[4] = kI[1]; Il.a[5] = kI[5]; Il.a[6] = kI[4]; Il.a[7] = kI[5]; Il.a[8] = kI[2];
        const m3 A = ::mulmm(Rl, Il);
        float Iw[6];          // the six distinct entries of A R^T
        Iw[0] = A.a[0] * Rl.a[0] + A.a[1] * Rl.a[1] + A.a[2] * Rl.a[2]; Iw[1] = A.a[3] * Rl.a[3] + A.a[4] * Rl.a[4] + A.a[5] * Rl.a[5]; Iw[2] = A.a[6] * Rl.a[6] + A.a[7] * Rl.a[7] + A.a[8] * Rl.a[8];
        Iw[3] = A.a[0] * Rl.a[3] + A.a[1] * Rl.a[4] + A.a[2] * Rl.a[5]; Iw[4] = A.a[0] * Rl.a[6] + A.a[1] * Rl.a[7] + A.a[2] * Rl.a[8]; Iw[5] = A.a[3] * Rl.a[6] + A.a[4] * Rl.a[7] + A.a[5] * Rl.a[8];
        auto Imul = [&](v3 x) { return mk3(Iw[0] * x.x + Iw[3] * x.y + Iw[4] * x.z, Iw[3] * x.x + Iw[1] * x.y + Iw[5] * x.z, Iw[4] * x.x + Iw[5] * x.y + Iw[2] * x.z); };
        const v3 rcm = ::mulmv(Rl, mk3(kcom[0], kcom[1], kcom[2])), com = pl + rcm;
        const v3 acom = aol + ::cross(all, rcm) + ::cross(wl, ::cross(wl, rcm));
        const v3 F = (acom - mk3(0.f, 0.f, gravz)) * kmass;
        const v3 Nt = Imul(all) + ::cross(wl, Imul(wl));
        if (c > 0 && c < NL) {
            float *o = lkL + 16 * c;
            ks4(o, make_float4(com.x, com.y, com.z, Iw[0])); ks4(o + 4, make_float4(Iw[1], Iw[2], Iw[3], Iw[4])); ks4(o + 8, make_float4(Iw[5], F.x, F.y, F.z)); ks4(o + 12, make_float4(Nt.x, Nt.y, Nt.z, 0.f));
        }
    }
    wave_sync();
    // ---- what every link adds to the sums of every dof that moves it (every lane, same values)
    static_for<1, NL>([&](auto lc_) {
        constexpr int l = decltype(lc_)::value;
        if constexpr (!KD::isfree[l]) {
            const float4 q0 = kl4(lkL + 16 * l), q1 = kl4(lkL + 16 * l + 4), q2 = kl4(lkL + 16 * l + 8), q3 = kl4(lkL + 16 * l + 12);
            const v3 com = mk3(q0.x, q0.y, q0.z), F = mk3(q2.y, q2.z, q2.w), Nt = mk3(q3.x, q3.y, q3.z);
            m3 I;
            I.a[0] = q0.w; I.a[4] = q1.x; I.a[8] = q1.y; I.a[1] = I.a[3] = q1.z; I.a[2] = I.a[6] = q1.w; I.a[5] = I.a[7] = q2.x;
            accumulate(lc_, com, I, F, Nt);
            __builtin_amdgcn_sched_barrier(0);          // (one link at a time: a scheduler that interleaves the links keeps all their temporaries alive)
        }
    });
    }
    // ---- the dof records and the robot's block of M: M[c][k] = lin_k . V_c + a_k . (TU_c + (n_c - n_k) x V_c) for k = c and the dofs above it;
    // row c goes out as whole rows (entries of unrelated dofs are zero), the mirrored entries M[k][c] one by one
    static_for<0, NRD>([&](auto cc) {
        constexpr int cr = decltype(cc)::value;
        float row[ST::RS];
#pragma unroll
        for (int i = 0; i < ST::RS; i++) row[i] = 0.f;
        static_for<0, NRD>([&](auto kc) {
            constexpr int kr = decltype(kc)::value;
            if constexpr (dof_anc_or_self<KD>(RD0 + kr, RD0 + cr)) {
                const v3 dk = k_sub(ANC[cr], ANC[kr]);
                row[kr] = ca(k_dot(LIN[kr], Vc[cr]), k_dot(ANG[kr], k_add(TUc[cr], k_cross(dk, Vc[cr]))));
            }
        });
        if (lane0) {
            float *o = st + 12 * (RD0 + cr);
            ks4(o, make_float4(ANG[cr].x, ANG[cr].y, ANG[cr].z, LIN[cr].x)); ks4(o + 4, make_float4(LIN[cr].y, LIN[cr].z, ANC[cr].x, ANC[cr].y));
            ks4(o + 8, make_float4(ANC[cr].z, Bc[cr], row[cr], 0.f));
            float *mr = st + ST::oM + ST::RS * cr;
            // row c: its lower part now; what lies right of the diagonal is written by the rows below it (the entries of dofs that neither
            // act above c nor below it are zero: written here, so that no row depends on what the region held before)
            static_for<0, NRD>([&](auto kc) {
                constexpr int kr = decltype(kc)::value;
                if constexpr (dof_anc_or_self<KD>(RD0 + kr, RD0 + cr)) { mr[kr] = row[kr]; if constexpr (kr != cr) st[ST::oM + ST::RS * kr + cr] = row[kr]; }
                else if constexpr (!dof_anc_or_self<KD>(RD0 + cr, RD0 + kr)) mr[kr] = 0.f;
            });
        }
    });
    (void)NV;
#undef K3C3
}

// top of the solve: every dof lane takes its own record and its row of M out of the staging area (after a wave_sync behind kin3_run)
template <class KD, int G, int NK>
__device__ __forceinline__ void kin3_fetch(int c, const float *st, Kin3Lane<G> &out) {
    constexpr int RD0 = KD::RD0, NRD = KD::NRD, NV = KD::NV;
    typedef Kin3Stage<KD> ST;
    const bool isdof = c < NV, isrobot = c >= RD0 && c < RD0 + NRD;
    const int cd = isdof ? c : 0, kr = isrobot ? c - RD0 : 0;
    const float4 r0 = kl4(st + 12 * cd), r1 = kl4(st + 12 * cd + 4), r2 = kl4(st + 12 * cd + 8);
    float mr[ST::RS];
    static_for<0, ST::RS / 4>([&](auto qc) {
        constexpr int q0 = 4 * decltype(qc)::value;
        const float4 t = kl4(st + ST::oM + ST::RS * kr + q0);
        mr[q0] = t.x; mr[q0 + 1] = t.y; mr[q0 + 2] = t.z; mr[q0 + 3] = t.w;
    });
    out.a = isdof ? mk3(r0.x, r0.y, r0.z) : mk3(0, 0, 0);
    out.l = isdof ? mk3(r0.w, r1.x, r1.y) : mk3(0, 0, 0);
    out.n = isdof ? mk3(r1.z, r1.w, r2.x) : mk3(0, 0, 0);
    out.bias = isdof ? r2.y : 0.f;
    out.Mdiag = isdof ? r2.z : 1.f;
    static_for<0, NK>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        float v = (k == c) ? out.Mdiag : 0.f;                       // free-body and padding lanes: their own diagonal entry
        if constexpr (k >= RD0 && k < RD0 + NRD) v = isrobot ? mr[k - RD0] : 0.f;
        out.Mrow[k] = v;
    });
}
