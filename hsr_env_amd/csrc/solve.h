// Dynamics + constraint assembly + Newton solver + Euler integration kernel (fp32, one lane per env).
//
// Reference path replaced: the rest of mj_step behind `self.sim.step()` (hsr/env.py:123) - mj_crb,
// mj_rne, passive/actuator forces, mj_makeConstraint, mj_fwdConstraint (Newton, elliptic cones),
// mj_Euler - plus the per-substep goal test and early exit of HSREnv.step (hsr/env.py:124-131).
// SURVEY.md section 8 a-2.2, a-2.4 ... a-2.7, a-3, a-4.
#pragma once
#include "devmath.h"
#include "model.h"

__device__ __forceinline__ int tri(int i, int j) { return i * (i + 1) / 2 + j; }   // j <= i

// in-place Cholesky of a packed lower-triangular SPD matrix; returns false when not SPD
__device__ bool chol_packed(View A, int n) {
    for (int j = 0; j < n; j++) {
        float sdiag = A[tri(j, j)];
        for (int k = 0; k < j; k++) { const float t = A[tri(j, k)]; sdiag -= t * t; }
        if (!(sdiag >= HSR_MINVAL)) return false;
        const float dj = sqrtf(sdiag), inv = 1.0f / dj;
        A[tri(j, j)] = dj;
        for (int i = j + 1; i < n; i++) {
            float t = A[tri(i, j)];
            for (int k = 0; k < j; k++) t -= A[tri(i, k)] * A[tri(j, k)];
            A[tri(i, j)] = t * inv;
        }
    }
    return true;
}
__device__ void chol_solve_packed(View L, int n, View x) {
    for (int i = 0; i < n; i++) { float sacc = x[i]; for (int k = 0; k < i; k++) sacc -= L[tri(i, k)] * x[k]; x[i] = sacc / L[tri(i, i)]; }
    for (int i = n - 1; i >= 0; i--) { float sacc = x[i]; for (int k = i + 1; k < n; k++) sacc -= L[tri(k, i)] * x[k]; x[i] = sacc / L[tri(i, i)]; }
}
// y = M x for packed symmetric M
__device__ void symv_packed(View M, int n, View x, View y) {
    for (int i = 0; i < n; i++) {
        float sacc = 0;
        for (int k = 0; k <= i; k++) sacc += M[tri(i, k)] * x[k];
        for (int k = i + 1; k < n; k++) sacc += M[tri(k, i)] * x[k];
        y[i] = sacc;
    }
}

struct Kin {   // per-env kinematics views
    View xpos, xmat, ang, lin, anc;
};
__device__ __forceinline__ v3 dof_point_vel(const Kin &k, int d, v3 p) {
    return k.lin.get3(d) + cross(k.ang.get3(d), p - k.anc.get3(d));
}
__device__ __forceinline__ void link_world_inertia(const DevModel &m, const Kin &k, int l, v3 &com, m3 &I) {
    const m3 R = k.xmat.getm(l);
    const float *li = m.link_inertia + 6 * l;
    m3 Il;
    Il.a[0] = li[0]; Il.a[1] = li[3]; Il.a[2] = li[4];
    Il.a[3] = li[3]; Il.a[4] = li[1]; Il.a[5] = li[5];
    Il.a[6] = li[4]; Il.a[7] = li[5]; Il.a[8] = li[2];
    com = k.xpos.get3(l) + mulmv(R, ld3(m.link_com, l));
    m3 Rt;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Rt.a[3 * i + j] = R.a[3 * j + i];
    I = mulmm(mulmm(R, Il), Rt);
}

__device__ __forceinline__ float impedance(const float *solimp, float pos) {
    float dmin = fminf(fmaxf(solimp[0], HSR_MINIMP), HSR_MAXIMP), dmax = fminf(fmaxf(solimp[1], HSR_MINIMP), HSR_MAXIMP);
    const float width = fmaxf(solimp[2], HSR_MINVAL), mid = fminf(fmaxf(solimp[3], HSR_MINIMP), HSR_MAXIMP);
    const float power = fmaxf(solimp[4], 1.f);
    if (dmin == dmax || width <= HSR_MINVAL) return 0.5f * (dmin + dmax);
    const float x = fabsf(pos) / width;
    if (x >= 1) return dmax;
    if (x <= 0) return dmin;
    float y;
    if (power == 1.f) y = x;
    else if (power == 2.f) y = x <= mid ? x * x * frcp(mid) : 1 - (1 - x) * (1 - x) * frcp(1 - mid);   // MuJoCo's default power: no powf
    else if (x <= mid) y = powf(x, power) / powf(mid, power - 1);
    else y = 1 - powf(1 - x, power) / powf(1 - mid, power - 1);
    return dmin + y * (dmax - dmin);
}

// row += sign * (dirp . jacp + dirr . jacr) for a point attached to `link`
__device__ __forceinline__ void point_jac_row(const DevModel &m, const Kin &k, int link, v3 pt, bool usep, v3 dirv,
                                              float sign, View row) {
    if (link == 0) return;
    for (int d = m.link_dofadr[link] + m.link_dofnum[link] - 1; d >= 0; d = m.dof_parent[d]) {
        float acc;
        if (usep) acc = dot(dof_point_vel(k, d, pt), dirv);
        else acc = dot(k.ang.get3(d), dirv);
        row[d] += sign * acc;
    }
}

// cost / gradient / Hessian of one elliptic contact at residual x (dim entries; arrays padded to 6)
template <bool WANT_H>
__device__ __forceinline__ float cone_eval(int dim, float mu, const float *fri, const float *D, const float *x, float *g, float *H) {
    float U[6], T2 = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) g[j] = 0;
    if (WANT_H) {
#pragma unroll
        for (int j = 0; j < 36; j++) H[j] = 0;
    }
    U[0] = x[0] * mu;
    const float Nn = U[0];
#pragma unroll
    for (int j = 1; j < 6; j++) { U[j] = (j < dim) ? x[j] * fri[j - 1] : 0.f; T2 += U[j] * U[j]; }
    const float T = sqrtf(T2);
    if (Nn >= mu * T || (T <= 0 && Nn >= 0)) return 0.f;                       // top zone
    if (mu * Nn + T <= 0 || (T <= 0 && Nn < 0)) {                               // bottom zone
        float c = 0;
#pragma unroll
        for (int j = 0; j < 6; j++) if (j < dim) { c += 0.5f * D[j] * x[j] * x[j]; g[j] = D[j] * x[j]; if (WANT_H) H[6 * j + j] = D[j]; }
        return c;
    }
    const float Dm = D[0] / (mu * mu * (1 + mu * mu)), NT = Nn - mu * T, invT = 1.0f / T;   // middle zone
    float gn[6];
    gn[0] = mu;
#pragma unroll
    for (int j = 1; j < 6; j++) gn[j] = (j < dim) ? -mu * U[j] * fri[j - 1] * invT : 0.f;
#pragma unroll
    for (int j = 0; j < 6; j++) g[j] = Dm * NT * gn[j];
    if (WANT_H) {
        const float invT3 = invT / T2;
#pragma unroll
        for (int j = 0; j < 6; j++)
#pragma unroll
            for (int k = 0; k < 6; k++) if (j < dim && k < dim) {
                float h = Dm * gn[j] * gn[k];
                if (j >= 1 && k >= 1) h += Dm * NT * (-mu * fri[j - 1] * fri[k - 1] * ((j == k ? invT : 0.f) - U[j] * U[k] * invT3));
                H[6 * j + k] = h;
            }
    }
    return 0.5f * Dm * NT * NT;
}

struct Efc {   // per-env constraint views + counts
    View J, D, aref, jar, jv, gr, cpair, cmu;
    int nv, nlim, ncon, nefc;
};

__device__ __forceinline__ void load_contact_params(const DevModel &m, const Efc &c, int ci, int &dim, float &mu, float *fri) {
    const int p = (int)c.cpair[ci];
    dim = m.pair_condim[p];
    mu = c.cmu[ci];
#pragma unroll
    for (int j = 0; j < 5; j++) fri[j] = m.pair_friction[5 * p + j];
}

// constraint cost at residual jar; writes gradient wrt jar (= -force) into gr
__device__ float constraint_cost(const DevModel &m, const Efc &c) {
    float cost = 0;
    for (int i = 0; i < c.nlim; i++) {
        const float x = c.jar[i];
        if (x < 0) { cost += 0.5f * c.D[i] * x * x; c.gr[i] = c.D[i] * x; } else c.gr[i] = 0;
    }
    int a = c.nlim;
    for (int ci = 0; ci < c.ncon; ci++) {
        int dim; float mu, fri[5], D[6], x[6], g[6];
        load_contact_params(m, c, ci, dim, mu, fri);
#pragma unroll
        for (int j = 0; j < 6; j++) if (j < dim) { D[j] = c.D[a + j]; x[j] = c.jar[a + j]; } else { D[j] = 0; x[j] = 0; }
        cost += cone_eval<false>(dim, mu, fri, D, x, g, nullptr);
#pragma unroll
        for (int j = 0; j < 6; j++) if (j < dim) c.gr[a + j] = g[j];
        a += dim;
    }
    return cost;
}

// 1-D derivatives of the total cost along the search direction at step alpha
__device__ void ls_eval(const DevModel &m, const Efc &c, float alpha, float g1, float g2, float &dphi, float &ddphi) {
    float dp = g1 + alpha * g2, hp = g2;
    for (int i = 0; i < c.nlim; i++) {
        const float jvi = c.jv[i], x = c.jar[i] + alpha * jvi;
        if (x < 0) { dp += c.D[i] * x * jvi; hp += c.D[i] * jvi * jvi; }
    }
    int a = c.nlim;
    for (int ci = 0; ci < c.ncon; ci++) {
        int dim; float mu, fri[5], D[6], x[6], v[6], g[6], H[36];
        load_contact_params(m, c, ci, dim, mu, fri);
#pragma unroll
        for (int j = 0; j < 6; j++) if (j < dim) { D[j] = c.D[a + j]; v[j] = c.jv[a + j]; x[j] = c.jar[a + j] + alpha * v[j]; } else { D[j] = 0; x[j] = 0; v[j] = 0; }
        cone_eval<true>(dim, mu, fri, D, x, g, H);
#pragma unroll
        for (int j = 0; j < 6; j++) {
            dp += g[j] * v[j];
#pragma unroll
            for (int k = 0; k < 6; k++) hp += v[j] * H[6 * j + k] * v[k];
        }
        a += dim;
    }
    dphi = dp; ddphi = hp;
}

struct Hot {   // hot per-thread vectors (LDS when they fit)
    View M, H, qfs, qas, qacc, Ma, grad, search, Mv, qfc;
};

// Ma = M a ; jar = J a - aref ; returns total cost, leaves the constraint gradient in gr
__device__ float eval_at(const DevModel &m, const Efc &c, const Hot &h, View a) {
    const int nv = c.nv;
    symv_packed(h.M, nv, a, h.Ma);
    float gauss = 0;
    for (int i = 0; i < nv; i++) gauss += 0.5f * (a[i] - h.qas[i]) * (h.Ma[i] - h.qfs[i]);
    for (int r = 0; r < c.nefc; r++) {
        float sacc = -c.aref[r];
        const View row = c.J.sub(r * nv);
        for (int k = 0; k < nv; k++) sacc += row[k] * a[k];
        c.jar[r] = sacc;
    }
    return gauss + constraint_cost(m, c);
}

// mode: 0 = forward only (no integration, no goal test), 1 = full substep
__global__ void __launch_bounds__(64) k_solve(DevModel m, DevState s, int mode, int goal_body, float geofence, int debug) {
    extern __shared__ float lds_hot[];
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= s.N) return;
    if (s.done[e]) return;
    const int N = s.N, nv = m.nv, nM = m.nM;
    View qpos{s.qpos + e, N}, qvel{s.qvel + e, N}, ctrl{s.ctrl + e, N}, warm{s.warm + e, N};
    Kin kin{View{s.xpos + e, N}, View{s.xmat + e, N}, View{s.dof_ang + e, N}, View{s.dof_lin + e, N}, View{s.dof_anchor + e, N}};
    View ws{s.ws + e, N};
    View hot = s.hot_in_lds ? View{lds_hot + threadIdx.x, (int)blockDim.x} : ws.sub(s.o_hot);
    Hot h;
    h.M = hot; h.H = hot.sub(nM); h.qfs = hot.sub(2 * nM); h.qas = hot.sub(2 * nM + nv); h.qacc = hot.sub(2 * nM + 2 * nv);
    h.Ma = hot.sub(2 * nM + 3 * nv); h.grad = hot.sub(2 * nM + 4 * nv); h.search = hot.sub(2 * nM + 5 * nv);
    h.Mv = hot.sub(2 * nM + 6 * nv); h.qfc = hot.sub(2 * nM + 7 * nv);
    int bad = 0;

    // ---------------- state check (mj_checkPos / mj_checkVel)
    for (int i = 0; i < m.nq; i++) { const float q = qpos[i]; if (!(fabsf(q) <= 1e10f)) bad = 1; }
    for (int i = 0; i < nv; i++) { const float q = qvel[i]; if (!(fabsf(q) <= 1e10f)) bad = 1; }

    // ---------------- a-2.2 joint-space inertia (direct composite form, packed lower triangle)
    for (int i = 0; i < nM; i++) h.M[i] = 0;
    for (int l = 1; l < m.nlink; l++) {
        v3 com; m3 I;
        link_world_inertia(m, kin, l, com, I);
        const float mass = m.link_mass[l];
        for (int a = m.link_dofadr[l] + m.link_dofnum[l] - 1; a >= 0; a = m.dof_parent[a]) {
            const v3 jpa = dof_point_vel(kin, a, com) * mass, Ijr = mulmv(I, kin.ang.get3(a));
            for (int b = a; b >= 0; b = m.dof_parent[b])
                h.M[tri(a, b)] += dot(jpa, dof_point_vel(kin, b, com)) + dot(Ijr, kin.ang.get3(b));
        }
    }
    if (debug) { View Mo{s.M + e, N}; for (int i = 0; i < nM; i++) Mo[i] = h.M[i]; }

    // ---------------- a-2.5 bias forces: link velocities / bias accelerations, then J^T [F; N]
    View lw = ws.sub(s.o_lw), lvo = ws.sub(s.o_lvo), lal = ws.sub(s.o_lal), lao = ws.sub(s.o_lao);
    lw.set3(0, mk3(0, 0, 0)); lvo.set3(0, mk3(0, 0, 0)); lal.set3(0, mk3(0, 0, 0)); lao.set3(0, mk3(0, 0, 0));
    for (int l = 1; l < m.nlink; l++) {
        const int d0 = m.link_dofadr[l];
        if (m.link_free[l]) {
            lvo.set3(l, mk3(qvel[d0], qvel[d0 + 1], qvel[d0 + 2]));
            lw.set3(l, mulmv(kin.xmat.getm(l), mk3(qvel[d0 + 3], qvel[d0 + 4], qvel[d0 + 5])));
            lal.set3(l, mk3(0, 0, 0)); lao.set3(l, mk3(0, 0, 0));
            continue;
        }
        const int p = m.link_parent[l];
        const v3 wp = lw.get3(p), vop = lvo.get3(p), alp = lal.get3(p), aop = lao.get3(p);
        const v3 xl = kin.xpos.get3(l), r = xl - kin.xpos.get3(p);
        v3 w = wp, al = alp;
        v3 vo = vop + cross(wp, r);
        v3 ao = aop + cross(alp, r) + cross(wp, cross(wp, r));
        for (int k = d0; k < d0 + m.link_dofnum[l]; k++) {
            const float qd = qvel[k];
            if (m.dof_type[k] == DOF_SLIDE) {
                const v3 sx = kin.lin.get3(k);
                vo = vo + sx * qd;
                ao = ao + cross(wp, sx) * (2 * qd);
            } else {
                const v3 a = kin.ang.get3(k);
                const v3 rho = xl - kin.anc.get3(k), rc = r - rho;
                const v3 wl = w + a * qd, all = al + cross(w, a) * qd;
                const v3 ac = aop + cross(alp, rc) + cross(wp, cross(wp, rc));
                const v3 vc = vop + cross(wp, rc);
                ao = ac + cross(all, rho) + cross(wl, cross(wl, rho));
                vo = vc + cross(wl, rho);
                w = wl; al = all;
            }
        }
        lw.set3(l, w); lvo.set3(l, vo); lal.set3(l, al); lao.set3(l, ao);
    }
    for (int k = 0; k < nv; k++) h.qfs[k] = -m.dof_damping[k] * qvel[k];       // passive
    for (int l = 1; l < m.nlink; l++) {
        v3 com; m3 I;
        link_world_inertia(m, kin, l, com, I);
        const v3 w = lw.get3(l), al = lal.get3(l), rc = com - kin.xpos.get3(l);
        const v3 acom = lao.get3(l) + cross(al, rc) + cross(w, cross(w, rc));
        const v3 F = (acom - mk3(0, 0, m.gravz)) * m.link_mass[l];
        const v3 Nt = mulmv(I, al) + cross(w, mulmv(I, w));
        for (int k = m.link_dofadr[l] + m.link_dofnum[l] - 1; k >= 0; k = m.dof_parent[k])
            h.qfs[k] -= dot(dof_point_vel(kin, k, com), F) + dot(kin.ang.get3(k), Nt);
    }
    // position actuators: force = kp*clamp(ctrl) - kp*gear*q, clamped to forcerange; qfrc = gear*force
    for (int a = 0; a < m.nu; a++) {
        const int k = m.act_dof[a];
        const float c = fminf(fmaxf(ctrl[a], m.act_ctrlrange[2 * a]), m.act_ctrlrange[2 * a + 1]);
        float f = m.act_kp[a] * c - m.act_kp[a] * m.act_gear[a] * qpos[m.dof_qposadr[k]];
        f = fminf(fmaxf(f, m.act_forcerange[2 * a]), m.act_forcerange[2 * a + 1]);
        h.qfs[k] += m.act_gear[a] * f;
    }
    // qacc_smooth = M^-1 qfrc_smooth (factor a copy of M in the H buffer)
    for (int i = 0; i < nM; i++) h.H[i] = h.M[i];
    for (int k = 0; k < nv; k++) h.qas[k] = h.qfs[k];
    if (chol_packed(h.H, nv)) chol_solve_packed(h.H, nv, h.qas); else bad = 1;

    // ---------------- a-2.4 constraint assembly
    Efc c;
    c.J = ws.sub(s.o_J); c.D = ws.sub(s.o_D); c.aref = ws.sub(s.o_aref); c.jar = ws.sub(s.o_jar); c.jv = ws.sub(s.o_jv);
    c.gr = ws.sub(s.o_gr); c.cpair = ws.sub(s.o_cpair); c.cmu = ws.sub(s.o_cmu);
    c.nv = nv;
    int ne = 0;
    for (int k = 0; k < nv && ne < m.njmax; k++) {
        if (!m.dof_limited[k]) continue;
        const float q = qpos[m.dof_qposadr[k]];
#pragma unroll
        for (int side = 0; side < 2; side++) {
            const float dist = side == 0 ? q - m.dof_range[2 * k] : m.dof_range[2 * k + 1] - q;
            if (dist < 0 && ne < m.njmax) {
                const View row = c.J.sub(ne * nv);
                for (int i = 0; i < nv; i++) row[i] = 0;
                row[k] = side == 0 ? 1.f : -1.f;
                const float imp = impedance(m.dof_solimp + 5 * k, dist);
                const float dmax = fminf(fmaxf(m.dof_solimp[5 * k + 1], HSR_MINIMP), HSR_MAXIMP);
                const float tc = m.dof_solref[2 * k], dr = m.dof_solref[2 * k + 1];
                const float Kimp = imp / (dmax * dmax * tc * tc * dr * dr), B = 2.0f / (dmax * tc);
                const float R = fmaxf((1 - imp) / imp * m.dof_invweight0[k], HSR_MINVAL);
                const float vel = (side == 0 ? 1.f : -1.f) * qvel[k];
                c.aref[ne] = -B * vel - Kimp * dist;
                c.D[ne] = 1.0f / R;
                ne++;
            }
        }
    }
    c.nlim = ne;
    int ncon = 0;
    {
        const float *con = s.con + (size_t)e * m.nslot * 8;
        for (int p = 0; p < m.npair; p++) {
            const int cnt = s.ncon_pair[(size_t)e * m.npair_pad + p];
            for (int i = 0; i < cnt; i++) {
                const int dim = m.pair_condim[p];
                if (ncon >= m.nconmax || ne + dim > m.njmax) break;
                const int b = (m.pair_slot[p] + i) * 8;
                const v3 pos = mk3(con[b], con[b + 1], con[b + 2]), nrm = mk3(con[b + 3], con[b + 4], con[b + 5]);
                const float dist = con[b + 6];
                // mju_makeFrame
                v3 t1 = (nrm.y > -0.5f && nrm.y < 0.5f) ? mk3(0, 1, 0) : mk3(0, 0, 1);
                t1 = normalized(t1 - nrm * dot(nrm, t1));
                const v3 t2 = cross(nrm, t1);
                const int g1 = m.pair_geom1[p], g2 = m.pair_geom2[p], l1 = m.geom_link[g1], l2 = m.geom_link[g2];
                const float *solref = m.pair_solref + 2 * p, *solimp = m.pair_solimp + 5 * p, *fri = m.pair_friction + 5 * p;
                const float imp = impedance(solimp, dist), dmax = fminf(fmaxf(solimp[1], HSR_MINIMP), HSR_MAXIMP);
                const float tc = solref[0], dr = solref[1];
                const float tran = m.geom_invweight[2 * g1] + m.geom_invweight[2 * g2];
                const float rot = m.geom_invweight[2 * g1 + 1] + m.geom_invweight[2 * g2 + 1];
                const float B = 2.0f / (dmax * tc), Kimp = imp / (dmax * dmax * tc * tc * dr * dr);
                const float R0 = fmaxf((1 - imp) / imp * tran, HSR_MINVAL);
                const float R1 = R0 / fmaxf(m.impratio, HSR_MINVAL);
                for (int j = 0; j < dim; j++) {
                    const View row = c.J.sub((ne + j) * nv);
                    for (int i2 = 0; i2 < nv; i2++) row[i2] = 0;
                    const v3 ax = (j % 3) == 0 ? nrm : ((j % 3) == 1 ? t1 : t2);
                    point_jac_row(m, kin, l2, pos, j < 3, ax, 1.f, row);
                    point_jac_row(m, kin, l1, pos, j < 3, ax, -1.f, row);
                    float vel = 0;
                    for (int i2 = 0; i2 < nv; i2++) vel += row[i2] * qvel[i2];
                    c.aref[ne + j] = -B * vel - (j == 0 ? Kimp * dist : 0.f);
                    float R = j == 0 ? R0 : (j == 1 ? R1 : R1 * fri[0] * fri[0] / (fri[j - 1] * fri[j - 1]));
                    c.D[ne + j] = 1.0f / R;
                }
                (void)rot;
                c.cpair[ncon] = (float)p;
                c.cmu[ncon] = dim > 1 ? fri[0] * sqrtf(R1 / R0) : fri[0];
                ncon++;
                ne += dim;
            }
        }
    }
    c.ncon = ncon; c.nefc = ne;

    // ---------------- a-2.6 Newton solver
    int iter = 0;
    if (ne == 0) {
        for (int k = 0; k < nv; k++) { h.qacc[k] = h.qas[k]; h.qfc[k] = 0; }
    } else {
        const float tol = m.tolerance, scale = 1.0f / (m.meaninertia * (nv > 1 ? nv : 1));
        // warm start: cheaper of qacc_warmstart and qacc_smooth
        for (int k = 0; k < nv; k++) h.qacc[k] = warm[k];
        const float cost_w = eval_at(m, c, h, h.qacc);
        const float cost_s = eval_at(m, c, h, h.qas);
        float cost;
        if (cost_w < cost_s) cost = eval_at(m, c, h, h.qacc);
        else { for (int k = 0; k < nv; k++) h.qacc[k] = h.qas[k]; cost = cost_s; }
        View T = ws.sub(s.o_T);
        for (; iter < m.iterations; iter++) {
            // gradient
            float gnorm = 0;
            for (int i = 0; i < nv; i++) {
                float sacc = h.Ma[i] - h.qfs[i];
                for (int r = 0; r < ne; r++) sacc += c.J[r * nv + i] * c.gr[r];
                h.grad[i] = sacc; gnorm += sacc * sacc;
            }
            gnorm = sqrtf(gnorm);
            if (scale * gnorm < tol) break;
            // Hessian H = M + J^T (d2s) J
            for (int i = 0; i < nM; i++) h.H[i] = h.M[i];
            for (int r = 0; r < c.nlim; r++) if (c.jar[r] < 0) {
                const View row = c.J.sub(r * nv);
                const float Dr = c.D[r];
                for (int i = 0; i < nv; i++) { const float ri = row[i]; if (ri != 0) for (int k = 0; k <= i; k++) h.H[tri(i, k)] += Dr * ri * row[k]; }
            }
            {
                int a = c.nlim;
                for (int ci = 0; ci < c.ncon; ci++) {
                    int dim; float mu, fri[5], D[6], x[6], g[6], Hc[36];
                    load_contact_params(m, c, ci, dim, mu, fri);
#pragma unroll
                    for (int j = 0; j < 6; j++) if (j < dim) { D[j] = c.D[a + j]; x[j] = c.jar[a + j]; } else { D[j] = 0; x[j] = 0; }
                    cone_eval<true>(dim, mu, fri, D, x, g, Hc);
                    float hsum = 0;
#pragma unroll
                    for (int j = 0; j < 36; j++) hsum += fabsf(Hc[j]);
                    if (hsum > 0) {
                        // T = Hc * Jc  (dim x nv), then H += Jc^T T
                        for (int i = 0; i < nv; i++) {
                            float col[6];
#pragma unroll
                            for (int j = 0; j < 6; j++) col[j] = (j < dim) ? c.J[(a + j) * nv + i] : 0.f;
#pragma unroll
                            for (int j = 0; j < 6; j++) if (j < dim) {
                                float t = 0;
#pragma unroll
                                for (int k = 0; k < 6; k++) t += Hc[6 * j + k] * col[k];
                                T[j * nv + i] = t;
                            }
                        }
                        for (int i = 0; i < nv; i++) {
                            float col[6];
                            bool any = false;
#pragma unroll
                            for (int j = 0; j < 6; j++) { col[j] = (j < dim) ? c.J[(a + j) * nv + i] : 0.f; any |= (col[j] != 0); }
                            if (!any) continue;
                            for (int k = 0; k <= i; k++) {
                                float t = 0;
#pragma unroll
                                for (int j = 0; j < 6; j++) if (j < dim) t += col[j] * T[j * nv + k];
                                h.H[tri(i, k)] += t;
                            }
                        }
                    }
                    a += dim;
                }
            }
            if (!chol_packed(h.H, nv)) { bad = 1; break; }
            for (int i = 0; i < nv; i++) h.search[i] = -h.grad[i];
            chol_solve_packed(h.H, nv, h.search);
            // exact line search (safeguarded 1-D Newton on phi')
            symv_packed(h.M, nv, h.search, h.Mv);
            float g1 = 0, g2 = 0, snorm = 0;
            for (int i = 0; i < nv; i++) { const float si = h.search[i]; g1 += si * (h.Ma[i] - h.qfs[i]); g2 += si * h.Mv[i]; snorm += si * si; }
            snorm = sqrtf(snorm);
            for (int r = 0; r < ne; r++) {
                float sacc = 0;
                const View row = c.J.sub(r * nv);
                for (int k = 0; k < nv; k++) sacc += row[k] * h.search[k];
                c.jv[r] = sacc;
            }
            const float gtol = tol * m.ls_tolerance * snorm / scale;
            float dp, hp, lo = 0, hi = -1;
            ls_eval(m, c, 0.f, g1, g2, dp, hp);
            // fp32 termination on the Newton decrement (predicted decrease), see solve_g.h
            if (dp >= 0 || hp <= 0 || scale * 0.5f * (-dp) < tol) break;
            float alpha = -dp / hp;
            for (int it = 0; it < m.ls_iterations; it++) {
                ls_eval(m, c, alpha, g1, g2, dp, hp);
                if (fabsf(dp) < gtol) break;
                if (dp < 0) lo = alpha; else hi = alpha;
                float nxt = alpha - dp / hp;
                if (!(nxt > lo) || (hi > 0 && !(nxt < hi))) nxt = hi > 0 ? 0.5f * (lo + hi) : 2 * alpha;
                if (nxt == alpha) break;                       // fp32 resolution reached
                alpha = nxt;
            }
            if (!(alpha > 0)) break;
            for (int i = 0; i < nv; i++) h.qacc[i] += alpha * h.search[i];
            cost = eval_at(m, c, h, h.qacc);
        }
        for (int i = 0; i < nv; i++) {
            float sacc = 0;
            for (int r = 0; r < ne; r++) sacc -= c.J[r * nv + i] * c.gr[r];
            h.qfc[i] = sacc;
        }
    }
    {
        View qa{s.qacc + e, N};
        for (int k = 0; k < nv; k++) qa[k] = h.qacc[k];
        s.ncon[e] = ncon; s.nefc[e] = ne; s.niter[e] = iter;
        if (debug) {
            View o1{s.qacc_smooth + e, N}, o2{s.qfrc_smooth + e, N}, o3{s.qfrc_constraint + e, N};
            for (int k = 0; k < nv; k++) { o1[k] = h.qas[k]; o2[k] = h.qfs[k]; o3[k] = h.qfc[k]; }
        }
    }
    if (mode == 0) { if (bad) s.bad[e] = 1; return; }

    // ---------------- a-2.7 mj_Euler: implicit joint damping, semi-implicit position update
    const float hstep = m.timestep;
    if (m.any_damping) {
        for (int i = 0; i < nM; i++) h.H[i] = h.M[i];
        for (int k = 0; k < nv; k++) { h.H[tri(k, k)] += hstep * m.dof_damping[k]; h.search[k] = h.qfs[k] + h.qfc[k]; }
        if (chol_packed(h.H, nv)) chol_solve_packed(h.H, nv, h.search); else bad = 1;
    } else {
        for (int k = 0; k < nv; k++) h.search[k] = h.qacc[k];
    }
    for (int k = 0; k < nv; k++) { const float v = qvel[k] + hstep * h.search[k]; qvel[k] = v; h.Mv[k] = v; if (!(fabsf(v) <= 1e10f)) bad = 1; }
    for (int l = 1; l < m.nlink; l++) {
        const int d0 = m.link_dofadr[l];
        if (m.link_free[l]) {
            const int a = m.link_qposadr[l];
            for (int k = 0; k < 3; k++) qpos[a + k] += hstep * h.Mv[d0 + k];
            const v3 w = mk3(h.Mv[d0 + 3], h.Mv[d0 + 4], h.Mv[d0 + 5]);
            const float wn = norm(w), angle = wn * hstep;
            if (angle > 0) {
                const v3 ax = w * (1.0f / wn);
                float sn, cs;
                sincosf(0.5f * angle, &sn, &cs);
                q4 q, qr;
                q.w = qpos[a + 3]; q.x = qpos[a + 4]; q.y = qpos[a + 5]; q.z = qpos[a + 6];
                qr.w = cs; qr.x = ax.x * sn; qr.y = ax.y * sn; qr.z = ax.z * sn;
                q = qnormalized(qmul(q, qr));
                qpos[a + 3] = q.w; qpos[a + 4] = q.x; qpos[a + 5] = q.y; qpos[a + 6] = q.z;
            }
        } else {
            for (int k = d0; k < d0 + m.link_dofnum[l]; k++) qpos[m.dof_qposadr[k]] += hstep * h.Mv[k];
        }
    }
    for (int k = 0; k < nv; k++) warm[k] = h.qacc[k];
    s.time[e] += hstep;
    s.nsteps[e] += 1;
    if (bad) s.bad[e] = 1;

    // ---------------- a-3 / a-4 goal test on the xpos of this substep's forward pass, latch done
    if (goal_body >= 0) {
        v3 bp;
        if (m.body_mocap[goal_body]) bp = mk3(s.mocap[e], s.mocap[N + e], s.mocap[2 * N + e]);
        else { const int l = m.body_link[goal_body]; bp = kin.xpos.get3(l) + mulmv(kin.xmat.getm(l), ld3(m.body_pos, goal_body)); }
        const v3 r = bp - mk3(s.mocap[e], s.mocap[N + e], s.mocap[2 * N + e]);
        if (norm(r) < geofence) s.done[e] = 1;
    }
}
