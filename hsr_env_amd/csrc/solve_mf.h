// Matrix-free cooperative solver kernel: same algorithm and lane-group geometry as solve_g.h (G lanes per env),
// but the constraint Jacobian is never stored.  A contact row is  J[adr+j][c] = ax_j . P_c  (j < 3) or ax_{j-3} . Q_c
// with P_c = sg_c (lin_c + ang_c x (pos - anchor_c)), Q_c = sg_c ang_c, so
//   J v      = per DISTINCT link: its velocity field A + W x p under v (six width-G DPP reductions), then lane = contact:
//              rows = ax . (V_l2(pos) - V_l1(pos)), ax . (W_l2 - W_l1)       (per contact when too many links are involved)
//   J^T g    = per lane:    P_c . (g0 n + g1 t1 + g2 t2) + Q_c . (g3 n + g4 t1 + g5 t2)
//   J^T w J  = per row: t = w * Jrc ; Hrow[k] += t * bcast_k(Jrc)   (DPP row_newbcast fused into the FMA)
// Joint-limit rows (J = +-e_dof) live entirely in the registers of their dof lane.  nv-vectors (qacc, search, ...) are
// one register per lane and are broadcast by DPP, so the per-env LDS footprint is ~4 KB (contact records + row
// scalars): 8 single-wave workgroups fit a CU, i.e. 2 waves per SIMD instead of 1 with the stored-Jacobian kernel.
#pragma once
#include "solve_g.h"
#include "kin3.h"

// contact record in LDS: 11 float4 (16-B aligned), grouped so that each consumer needs few ds_read_b128:
//   q0 pos.xyz L1 | q1 n.xyz L2 | q2 t1.xyz ADR | q3 t2.xyz DIM | q4 MU F0 F2 F3 | q5 ZONE DM K3 B | q6 fw.xyz KD | q7 tw.xyz - |
//   q8..q10 GN[6] U[6].   PAIR / SLOT (written by E2, consumed at the top of E3) alias q7.
enum { C2_POS = 0, C2_L1 = 3, C2_N = 4, C2_L2 = 7, C2_T1 = 8, C2_ADR = 11, C2_T2 = 12, C2_DIM = 15, C2_MU = 16, C2_F0 = 17,
       C2_F2 = 18, C2_F3 = 19, C2_ZONE = 20, C2_DM = 21, C2_K3 = 22, C2_B = 23, C2_FW = 24, C2_KD = 27, C2_TW = 28, C2_SLOT = 30, C2_PAIR = 31,
       C2_GN = 32, C2_U = 38, C2_SIZE = 44, C2_AX = 4, C2_AXS = 4 };
__device__ __forceinline__ float4 lds4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

template <int G> struct MfLayout {
    int R, MS, oRows, oB, oLv, total, envf;
    __host__ __device__ MfLayout(int rows, int kstride) {
        R = rows; MS = G + 1;
        const int a = 5 * R > kstride ? 5 * R : kstride;                    // row scalars; the kin record aliases them early on
        const int b = G * MS > C2_SIZE * G ? G * MS : C2_SIZE * G;          // inertia matrix, then contact records
        oRows = 0; oB = (a + 3) & ~3;
        oLv = (oB + b + 3) & ~3;                                            // per-link velocity fields of jmul: 5 links x 6
        total = oLv + 32;
        envf = total;                                                       // LDS floats per env (the name the shared solver body uses)
    }
};

// tile-free triangular solves: lane c holds lo[k] = L[c][k] (k < c, else 0) and invd = 1 / L[c][c]
template <int G, int NK = G> __device__ __forceinline__ float chol_solve_mf(const float (&row)[G], float invd, float b, int nv, int c) {
    asm volatile("" : "+v"(c));          // lane masks formed where they are used, not hoisted out of the caller's loops as spilled SGPR pairs (chol_g_fwd)
    float nlo[G];                                                // minus the strictly lower part of row c of L, 0 elsewhere
#pragma unroll
    for (int k = 0; k < NK; k++) nlo[k] = (k < c) ? -row[k] : 0.f;
    float sacc = b, y = 0.f;
    static_for<0, NK>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if (j < nv) {
            const float t = sacc * invd;                         // lane j: y_j
            if (c == j) y = t;
            fmac_bcast<G, j, true>(sacc, nlo[j], bc_prepare<G>(t));   // sacc -= L[c][j] y_j
        }
    });
    // L^T x = y: x_j = (y_j - sum_{i > j} L[i][j] x_i) / L[j][j]; the sum runs over lanes (nlo[j] is 0 for lanes i <= j).  Two columns
    // per step: the two group sums over the lanes solved so far run side by side, x_j follows, and column j - 1 only lacks the term of
    // lane j itself, which lane j forms and broadcasts - one reduction latency per two columns of the dependent chain
    float x = 0.f;
    static_for<0, NK / 2>([&](auto jc) {
        constexpr int j = NK - 1 - 2 * decltype(jc)::value;           // columns j and j - 1
        if (j - 1 < nv) {
            const float A = gsum<G>(nlo[j] * x), B = gsum<G>(nlo[j - 1] * x);
            const float xj = (j < nv) ? (y + A) * invd : 0.f;
            if (c == j) x = xj;
            const float t = gbcast<G, j>(nlo[j - 1] * xj);
            if (c == j - 1) x = (y + B + t) * invd;
        }
    });
    if constexpr (NK % 2 == 1) {
        if (0 < nv) { const float tot = gsum<G>(nlo[0] * x); if (c == 0) x = (y + tot) * invd; }
    }
    return x;
}

// back substitution alone (the forward half was done by chol_g_fwd): lane c holds y_c, the strictly lower part of row c of L and invd
template <int G, int NK = G> __device__ __forceinline__ float chol_back_mf(const float (&row)[G], float invd, float y, int nv, int c) {
    asm volatile("" : "+v"(c));          // lane masks formed where they are used, not hoisted out of the caller's loops as spilled SGPR pairs (chol_g_fwd)
    float nlo[G];
#pragma unroll
    for (int k = 0; k < NK; k++) nlo[k] = (k < c) ? -row[k] : 0.f;
    float x = 0.f;
    static_for<0, NK / 2>([&](auto jc) {
        constexpr int j = NK - 1 - 2 * decltype(jc)::value;
        if (j - 1 < nv) {
            const float A = gsum<G>(nlo[j] * x), B = gsum<G>(nlo[j - 1] * x);
            const float xj = (j < nv) ? (y + A) * invd : 0.f;
            if (c == j) x = xj;
            const float t = gbcast<G, j>(nlo[j - 1] * xj);
            if (c == j - 1) x = (y + B + t) * invd;
        }
    });
    if constexpr (NK % 2 == 1) {
        if (0 < nv) { const float tot = gsum<G>(nlo[0] * x); if (c == 0) x = (y + tot) * invd; }
    }
    return x;
}

// ... and the two substitutions with such a factor (chol_g_tail): ND dependent steps each; a tail lane solves its own equation
template <int G, int NK, int ND> __device__ __forceinline__ float chol_solve_tail(const float (&row)[G], float invd, float b, int c) {
    asm volatile("" : "+v"(c));          // lane masks formed where they are used, not hoisted out of the caller's loops as spilled SGPR pairs (chol_g_fwd)
    float nlo[ND > 0 ? ND : 1];
#pragma unroll
    for (int k = 0; k < ND; k++) nlo[k] = (k < c && c < ND) ? -row[k] : 0.f;
    float sacc = b, y = 0.f;
    static_for<0, ND>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const float t = sacc * invd;
        if (c == j) y = t;
        fmac_bcast<G, j, true>(sacc, nlo[j], bc_prepare<G>(t));
    });
    float x = 0.f;
    if (c >= ND) x = b * invd * invd;
    static_for<0, ND>([&](auto jc) {
        constexpr int j = ND - 1 - decltype(jc)::value;
        const float tot = (ND <= 16) ? gsum<16>(nlo[j] * x) : gsum<G>(nlo[j] * x);      // the coupled dofs sit in the first DPP row
        if (c == j) x = (y + tot) * invd;
    });
    return x;
}

// the contact counts of eight consecutive candidate pairs as one byte each (the chain keeps them as ints in global memory)
__device__ __forceinline__ unsigned long long pair_cnt8_(const int *p) {
    unsigned long long r = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) r |= (unsigned long long)(p[k] & 0xff) << (8 * k);
    return r;
}

template <int G>
__global__ void __launch_bounds__(64, 2) k_solve_mf(DevModel m, DevState s, int mode, int goal_body, float geofence, int debug) {
    extern __shared__ __align__(16) float lds[];
    constexpr int EPB = 64 / G, NK = G;
    const MfLayout<G> L(m.njmax, s.kstride);
    const int tid = threadIdx.x, g = tid / G, c = tid % G;
    const int e_raw = blockIdx.x * EPB + g;
    const bool valid = e_raw < s.N && !s.done[e_raw < s.N ? e_raw : 0];
    if (!__syncthreads_or(valid)) return;
    const int e = valid ? e_raw : 0;
    const int N = s.N, nv = m.nv, R = L.R, MS = L.MS;
    float *E = lds + (size_t)g * L.total;
    float *rD = E + L.oRows, *rAref = rD + R, *rJar = rAref + R, *rJv = rJar + R, *rDw = rJv + R;
    float *kAng = E + L.oRows, *kLin = kAng + 3 * nv, *kAnc = kAng + 6 * nv, *lk = kAng + 9 * nv;     // kin_aos record (phases A-C)
    float *M = E + L.oB, *con = E + L.oB;
    float *lvbuf = E + L.oLv;
    const int lvcap = 5;
    const bool isdof = c < nv;
    int bad = 0;
    __shared__ int sParent[32], sMask[NLMAX];
    __shared__ float sMass[NLMAX];
    if (tid < nv) sParent[tid] = m.dof_parent[tid];
    if (tid < m.nlink && tid < NLMAX) { sMask[tid] = m.link_dofmask[tid]; sMass[tid] = m.link_mass[tid]; }

    PHASE_T0();
    // ---------------- phase A: all first-level global loads, back to back
    const int *pair_cnt_ = s.ncon_pair + (size_t)e * m.npair_pad;
    float qvel_c = 0, warm_c = 0, my_q = 0, my_ctrl = 0, damp_c = 0;
    int my_type = -1, my_qadr = 0, my_quat_lane = -1, my_limited = 0, my_act = -1;
    float lim_lo = 0, lim_hi = 0, lim_sr0 = 1, lim_sr1 = 1, lim_iw = 0, lim_si[5] = {0, 0, 0, 0, 0}, lim_B = 0, lim_K = 0;
    float act_p[6] = {0, 0, 0, 0, 0, 0};
    if (isdof) {
        my_type = m.dof_type[c]; my_qadr = m.dof_qposadr[c]; my_limited = m.dof_limited[c]; my_act = m.dof_act[c];
        my_quat_lane = m.link_dofadr[m.dof_link[c]] + 3;
        damp_c = m.dof_damping[c];
        qvel_c = s.qvel[(size_t)c * N + e]; warm_c = s.warm[(size_t)c * N + e];
        lim_lo = m.dof_range[2 * c]; lim_hi = m.dof_range[2 * c + 1]; lim_sr0 = m.dof_solref[2 * c]; lim_sr1 = m.dof_solref[2 * c + 1];
        lim_iw = m.dof_invweight0[c];
#pragma unroll
        for (int j = 0; j < 5; j++) lim_si[j] = m.dof_solimp[5 * c + j];
        { const float dmax = fminf(fmaxf(lim_si[1], HSR_MINIMP), HSR_MAXIMP); lim_B = 2.0f / (dmax * lim_sr0); lim_K = 1.0f / (dmax * dmax * lim_sr0 * lim_sr0 * lim_sr1 * lim_sr1); }
        my_q = s.qpos[(size_t)my_qadr * N + e];
        if (my_act >= 0) {
            my_ctrl = s.ctrl[(size_t)my_act * N + e];
            act_p[0] = m.act_kp[my_act]; act_p[1] = m.act_gear[my_act];
            act_p[2] = m.act_ctrlrange[2 * my_act]; act_p[3] = m.act_ctrlrange[2 * my_act + 1];
            act_p[4] = m.act_forcerange[2 * my_act]; act_p[5] = m.act_forcerange[2 * my_act + 1];
        }
        if (!(fabsf(qvel_c) <= 1e10f) || !(fabsf(my_q) <= 1e10f)) bad = 1;
    }
    {
        const float4 *src = reinterpret_cast<const float4 *>(s.kin_aos + (size_t)e * s.kstride);
        float4 *dst = reinterpret_cast<float4 *>(kAng);
        const int n4 = s.kstride / 4;
        float4 tmp[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { const int idx = c + i * G; tmp[i] = idx < n4 ? src[idx] : make_float4(0, 0, 0, 0); }
#pragma unroll
        for (int i = 0; i < 8; i++) { const int idx = c + i * G; if (idx < n4) dst[idx] = tmp[i]; }
    }
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
        if ((goal_body >= 0 || s.ngoal > 0) && mode != 0) {
            const v3 goal = mk3(s.mocap[e], s.mocap[N + e], s.mocap[2 * N + e]);
            View xpos{s.xpos + e, N}, xmat{s.xmat + e, N};
            auto body_point = [&](int body) -> v3 {
                if (m.body_mocap[body]) return goal;
                const int l = m.body_link[body];
                return xpos.get3(l) + mulmv(xmat.getm(l), ld3(m.body_pos, body));
            };
            reach = true;
            if (goal_body >= 0) reach = norm(body_point(goal_body) - goal) < geofence;
            for (int k = 0; k < s.ngoal; k++) reach = reach && norm(body_point(s.goal_a[k]) - body_point(s.goal_b[k])) < s.goal_d[k];
        }
    }
    constexpr int NDK = -1;                // ... and the run-time ndense
    constexpr bool KIN3 = false;           // (kin3.h, compile-time tree: persistent kernel only)
    constexpr bool REP = false;            // (solo-server replicas: persistent kernel only)
    using MT = DevModel;                  // (the chain kernel reads every model field at run time)
    constexpr bool hook_jv_per_contact = false, hook_majorant = false, hook_dense_chol = false, EXACT_CT = false;
    constexpr int nfb = 0;                 // the per-substep chain keeps the per-contact Hessian assembly (no scratch for the per-body one)
    float *fbK = nullptr;
#define SOLVE_STORE_DIAG true
#define PAIR_CNT8(p) pair_cnt8_(pair_cnt_ + (p))
#include "solve_body.inc"
#undef PAIR_CNT8
#undef SOLVE_STORE_DIAG
    const float v1 = __shfl_down(vnew, 1, G), v2 = __shfl_down(vnew, 2, G);
    PHASE(16);
    if (valid && isdof) {
        s.qvel[(size_t)c * N + e] = vnew;
        s.warm[(size_t)c * N + e] = qacc_c;
        if (my_type == DOF_SLIDE || my_type == DOF_HINGE || my_type == DOF_FREE_LIN) s.qpos[(size_t)my_qadr * N + e] = my_q + h * vnew;
    }
    PHASE(17);
    if (valid && isdof) {
        if (my_type == DOF_FREE_ANG && c == my_quat_lane) {
            const v3 w = mk3(vnew, v1, v2);
            const float wn = norm(w), angle = wn * h;
            if (angle > 0) {
                const v3 ax = w * (1.0f / wn);
                float sn, cs;
                fast_sincos(0.5f * angle, &sn, &cs);
                q4 qr;
                qr.w = cs; qr.x = ax.x * sn; qr.y = ax.y * sn; qr.z = ax.z * sn;
                const q4 q = qnormalized(qmul(quat0, qr));
                s.qpos[(size_t)my_qadr * N + e] = q.w; s.qpos[(size_t)(my_qadr + 1) * N + e] = q.x;
                s.qpos[(size_t)(my_qadr + 2) * N + e] = q.y; s.qpos[(size_t)(my_qadr + 3) * N + e] = q.z;
            }
        }
    }
    PHASE(13);
    const float bsum = gsum<G>((float)bad);
    if (valid && c == 0) {
        s.time[e] = time0 + h;
        s.nsteps[e] = nsteps0 + 1;
        if (bsum > 0) s.bad[e] = 1;
        if (reach) s.done[e] = 1;
    }
    PHASE(13);
    PHASE_FLUSH();
}
