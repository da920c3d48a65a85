// Matrix-free cooperative solver kernel: same algorithm and lane-group geometry as solve_g.h (G lanes per env),
// but the constraint Jacobian is never stored.  A contact row is  J[adr+j][c] = ax_j . P_c  (j < 3) or ax_{j-3} . Q_c
// with P_c = sg_c (lin_c + ang_c x (pos - anchor_c)), Q_c = sg_c ang_c, so
//   J v      = per contact: V = sum_c v_c P_c, W = sum_c v_c Q_c (six width-G DPP reductions), rows = ax . V, ax . W
//   J^T g    = per lane:    P_c . (g0 n + g1 t1 + g2 t2) + Q_c . (g3 n + g4 t1 + g5 t2)
//   J^T w J  = per row: t = w * Jrc ; Hrow[k] += t * bcast_k(Jrc)   (DPP row_newbcast fused into the FMA)
// Joint-limit rows (J = +-e_dof) live entirely in the registers of their dof lane.  nv-vectors (qacc, search, ...) are
// one register per lane and are broadcast by DPP, so the per-env LDS footprint is ~4 KB (contact records + row
// scalars): 8 single-wave workgroups fit a CU, i.e. 2 waves per SIMD instead of 1 with the stored-Jacobian kernel.
#pragma once
#include "solve_g.h"

enum { C2_POS = 0, C2_N = 3, C2_T1 = 6, C2_T2 = 9, C2_L1 = 12, C2_L2 = 13, C2_ADR = 14, C2_DIM = 15, C2_MU = 16, C2_F0 = 17,
       C2_F2 = 18, C2_F3 = 19, C2_ZONE = 20, C2_DM = 21, C2_K3 = 22, C2_B = 23, C2_KD = 24, C2_FW = 25, C2_TW = 28, C2_GN = 31,
       C2_U = 37, C2_PAIR = 43, C2_SLOT = 44, C2_SIZE = 45 };

template <int G> struct MfLayout {
    int R, MS, oRows, oB, total;
    __host__ __device__ MfLayout(int rows, int kstride) {
        R = rows; MS = G + 1;
        const int a = 6 * R > kstride ? 6 * R : kstride;                    // row scalars; the kin record aliases them early on
        const int b = G * MS > C2_SIZE * G ? G * MS : C2_SIZE * G;          // inertia matrix, then contact records
        oRows = 0; oB = (a + 3) & ~3;
        total = (oB + b + 3) & ~3;
    }
};

// tile-free triangular solves: lane c holds lo[k] = L[c][k] (k < c, else 0) and invd = 1 / L[c][c]
template <int G> __device__ __forceinline__ float chol_solve_mf(const float (&row)[G], float invd, float b, int nv, int c) {
    float lo[G];
#pragma unroll
    for (int k = 0; k < G; k++) lo[k] = (k < c) ? row[k] : 0.f;
    float sacc = b, y = 0.f;
    static_for<0, G>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if (j < nv) {
            const float yj = gbcast<G, j>(sacc * invd);
            if (c == j) y = yj;
            sacc -= lo[j] * yj;
        }
    });
    // L^T x = y: x_j = (y_j - sum_{i > j} L[i][j] x_i) / L[j][j]; the sum runs over lanes (lo[j] is 0 for lanes i <= j)
    float x = 0.f;
    static_for<0, G>([&](auto jc) {
        constexpr int j = G - 1 - decltype(jc)::value;
        if (j < nv) {
            const float tot = gsum<G>(lo[j] * x);
            if (c == j) x = (y - tot) * invd;
        }
    });
    return x;
}

template <int G>
__global__ void __launch_bounds__(64, 2) k_solve_mf(DevModel m, DevState s, int mode, int goal_body, float geofence, int debug) {
    extern __shared__ __align__(16) float lds[];
    constexpr int EPB = 64 / G;
    const MfLayout<G> L(m.njmax, s.kstride);
    const int tid = threadIdx.x, g = tid / G, c = tid % G;
    const int e_raw = blockIdx.x * EPB + g;
    const bool valid = e_raw < s.N && !s.done[e_raw < s.N ? e_raw : 0];
    if (!__syncthreads_or(valid)) return;
    const int e = valid ? e_raw : 0;
    const int N = s.N, nv = m.nv, R = L.R, MS = L.MS;
    float *E = lds + (size_t)g * L.total;
    float *rD = E + L.oRows, *rAref = rD + R, *rJar = rAref + R, *rJv = rJar + R, *rGr = rJv + R, *rDw = rGr + R;
    float *kAng = E + L.oRows, *kLin = kAng + 3 * nv, *kAnc = kAng + 6 * nv, *lk = kAng + 9 * nv;     // kin_aos record (phases A-C)
    float *M = E + L.oB, *con = E + L.oB;
    const bool isdof = c < nv;
    int bad = 0;
    __shared__ int sParent[32], sMask[NLMAX];
    __shared__ float sMass[NLMAX];
    if (tid < nv) sParent[tid] = m.dof_parent[tid];
    if (tid < m.nlink && tid < NLMAX) { sMask[tid] = m.link_dofmask[tid]; sMass[tid] = m.link_mass[tid]; }

    PHASE_T0();
    // ---------------- phase A: all first-level global loads, back to back
    constexpr int MAXCH = 256 / G;
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
        if (goal_body >= 0 && mode != 0) {
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
#pragma unroll
    for (int k = 0; k < G + 1; k++) M[c * MS + k] = 0.f;
    __syncthreads();
    if (isdof) {
        a_c = mk3(kAng[3 * c], kAng[3 * c + 1], kAng[3 * c + 2]);
        l_c = mk3(kLin[3 * c], kLin[3 * c + 1], kLin[3 * c + 2]);
        n_c = mk3(kAnc[3 * c], kAnc[3 * c + 1], kAnc[3 * c + 2]);
    }

    PHASE(0);
    // ---------------- phase B/C: inertia rows (a-2.2), bias force (a-2.5), lane = dof
    float bias_c = 0;
    if (isdof) {
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
            const float ct = fminf(fmaxf(my_ctrl, act_p[2]), act_p[3]);
            float f = act_p[0] * ct - act_p[0] * act_p[1] * my_q;
            f = fminf(fmaxf(f, act_p[4]), act_p[5]);
            qfs_c += act_p[1] * f;
        }
    }
    __syncthreads();
    if (isdof) for (int k = sParent[c]; k >= 0; k = sParent[k]) M[k * MS + c] = M[c * MS + k];   // mirror
    if (!isdof) M[c * MS + c] = 1.f;
    __syncthreads();
    if (debug && valid && isdof) for (int k = 0; k <= c; k++) s.M[(size_t)(c * (c + 1) / 2 + k) * N + e] = M[c * MS + k];
    float Mrow[G];
#pragma unroll
    for (int k = 0; k < G; k++) Mrow[k] = M[c * MS + k];

    PHASE(1);
    // ---------------- qacc_smooth = M^-1 qfrc_smooth
    float qas_c;
    {
        float Lr[G];
#pragma unroll
        for (int k = 0; k < G; k++) Lr[k] = Mrow[k];
        float invd;
        if (!chol_g<G>(Lr, invd, nv, m.ndense, c)) bad = 1;
        qas_c = chol_solve_mf<G>(Lr, invd, qfs_c, nv, c);
        if (!isdof) qas_c = 0;
    }
    __syncthreads();          // M (LDS) is dead from here: its region becomes the contact records

    PHASE(2);
    // ---------------- phase E: constraint assembly (a-2.4)
    // E1 joint limits: rows in dof order (lower side then upper side); everything about a limit row stays in the
    // registers of its dof lane (index 0 = lower, 1 = upper)
    int nlim;
    bool lim_on[2] = {false, false};
    float lim_D[2] = {0, 0}, lim_aref[2] = {0, 0}, lim_gr[2] = {0, 0}, lim_dw[2] = {0, 0};
    {
        float dlo = 0, dhi = 0;
        if (valid && my_limited) { dlo = my_q - lim_lo; dhi = lim_hi - my_q; lim_on[0] = dlo < 0; lim_on[1] = dhi < 0; }
        const int cnt = (int)lim_on[0] + (int)lim_on[1];
        const int incl = gscan_incl<G>(cnt, c);
        nlim = glast<G>(incl);
        int r = incl - cnt;
#pragma unroll
        for (int side = 0; side < 2; side++) {
            if (lim_on[side]) {
                if (r < R) {
                    const float dist = side == 0 ? dlo : dhi, sg = side == 0 ? 1.f : -1.f;
                    const float imp = impedance(lim_si, dist);
                    const float dmax = fminf(fmaxf(lim_si[1], HSR_MINIMP), HSR_MAXIMP);
                    const float Kimp = imp / (dmax * dmax * lim_sr0 * lim_sr0 * lim_sr1 * lim_sr1), B = 2.0f / (dmax * lim_sr0);
                    lim_aref[side] = -B * sg * qvel_c - Kimp * dist;
                    lim_D[side] = 1.0f / fmaxf((1 - imp) / imp * lim_iw, HSR_MINVAL);
                } else lim_on[side] = false;
                r++;
            }
        }
        if (nlim > R) nlim = R;
    }
    // E2 contact compaction in (pair, index) order
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
                        if (ci < G) { float *cr = con + C2_SIZE * ci; cr[C2_PAIR] = (float)p; cr[C2_SLOT] = (float)(slot0 + i); }
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
    int nefc = nlim;
    {
        int dim = 0;
        float *cr = con + C2_SIZE * c;
        float4 pr[4] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)}, cd0 = make_float4(0, 0, 0, 0), cd1 = cd0;
        if (c < ncon) {
            const int p = (int)cr[C2_PAIR], slot = (int)cr[C2_SLOT];
            const float4 *prp = reinterpret_cast<const float4 *>(m.pair_rec + 16 * p);
            const float4 *cdp = reinterpret_cast<const float4 *>(s.con + ((size_t)e * m.nslot + slot) * 8);
            pr[0] = prp[0]; pr[1] = prp[1]; pr[2] = prp[2]; pr[3] = prp[3]; cd0 = cdp[0]; cd1 = cdp[1];
            dim = (int)pr[0].x;
        }
        const int incl = gscan_incl<G>(dim, c);
        const int adr = nlim + incl - dim;
        const bool ovf = c < ncon && adr + dim > R;
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
            cr[C2_POS] = pos.x; cr[C2_POS + 1] = pos.y; cr[C2_POS + 2] = pos.z;
            cr[C2_N] = nrm.x; cr[C2_N + 1] = nrm.y; cr[C2_N + 2] = nrm.z;
            cr[C2_T1] = t1.x; cr[C2_T1 + 1] = t1.y; cr[C2_T1 + 2] = t1.z;
            cr[C2_T2] = t2.x; cr[C2_T2 + 1] = t2.y; cr[C2_T2 + 2] = t2.z;
            cr[C2_L1] = pr[0].y; cr[C2_L2] = pr[0].z;
            cr[C2_ADR] = (float)adr; cr[C2_DIM] = (float)dim;
            cr[C2_MU] = dim > 1 ? fri[0] * sqrtf(R1 / R0) : fri[0];
            cr[C2_F0] = fri[0]; cr[C2_F2] = fri[2]; cr[C2_F3] = fri[3];
            cr[C2_ZONE] = -1.f;
            cr[C2_B] = B; cr[C2_KD] = Kimp * dist;
            for (int j = 0; j < dim; j++) {
                const float Rj = j == 0 ? R0 : (j == 1 ? R1 : R1 * fri[0] * fri[0] / (fri[j - 1] * fri[j - 1]));
                rD[adr + j] = 1.0f / Rj;
            }
        }
        // total row count = address past the last kept contact
        const int endrow = (c < ncon) ? adr + dim : nlim;
        int mx = endrow;
#pragma unroll
        for (int off = 1; off < G; off <<= 1) { const int t = __shfl_xor(mx, off, G); mx = t > mx ? t : mx; }
        nefc = mx;
    }
    __syncthreads();
    PHASE(4);

    // (J v) for every contact row into out[] (no aref); lane c contributes v_c.  Group-uniform loop, no barriers.
    auto jmul = [&](float vc, float *out) {
        for (int ci = 0; ci < ncon; ci++) {
            const float *cr = con + C2_SIZE * ci;
            const int l1 = (int)cr[C2_L1], l2 = (int)cr[C2_L2];
            const float sg = (float)(((sMask[l2] >> c) & 1) - ((sMask[l1] >> c) & 1)) * vc;
            const v3 pos = mk3(cr[C2_POS], cr[C2_POS + 1], cr[C2_POS + 2]);
            const v3 P = (l_c + cross(a_c, pos - n_c)) * sg, Q = a_c * sg;
            const float Vx = gsum<G>(P.x), Vy = gsum<G>(P.y), Vz = gsum<G>(P.z);
            const float Wx = gsum<G>(Q.x), Wy = gsum<G>(Q.y), Wz = gsum<G>(Q.z);
            const int adr = (int)cr[C2_ADR], dim = (int)cr[C2_DIM];
            if (c < dim) {
                const int jj = c % 3;
                const v3 ax = mk3(cr[3 + 3 * jj], cr[4 + 3 * jj], cr[5 + 3 * jj]);
                out[adr + c] = c < 3 ? (ax.x * Vx + ax.y * Vy + ax.z * Vz) : (ax.x * Wx + ax.y * Wy + ax.z * Wz);
            }
        }
    };
    // lane c's entry of J^T (world wrench stored per contact in C2_FW / C2_TW) + its own limit rows
    auto jt_force = [&]() -> float {
        float acc = 0.f;
        if (lim_on[0]) acc += lim_gr[0];
        if (lim_on[1]) acc -= lim_gr[1];
        for (int ci = 0; ci < ncon; ci++) {
            const float *cr = con + C2_SIZE * ci;
            const int l1 = (int)cr[C2_L1], l2 = (int)cr[C2_L2];
            const float sg = (float)(((sMask[l2] >> c) & 1) - ((sMask[l1] >> c) & 1));
            const v3 pos = mk3(cr[C2_POS], cr[C2_POS + 1], cr[C2_POS + 2]);
            const v3 P = l_c + cross(a_c, pos - n_c);
            acc += sg * (P.x * cr[C2_FW] + P.y * cr[C2_FW + 1] + P.z * cr[C2_FW + 2] + a_c.x * cr[C2_TW] + a_c.y * cr[C2_TW + 1] + a_c.z * cr[C2_TW + 2]);
        }
        return isdof ? acc : 0.f;
    };

    // E5 reference acceleration of contact rows: aref = -B (J qvel) - K imp dist (row 0 only)
    jmul(qvel_c, rJv);
    __syncthreads();
    if (c < ncon) {
        const float *cr = con + C2_SIZE * c;
        const int adr = (int)cr[C2_ADR], dim = (int)cr[C2_DIM];
        for (int j = 0; j < dim; j++) rAref[adr + j] = -cr[C2_B] * rJv[adr + j] - (j == 0 ? cr[C2_KD] : 0.f);
    }
    __syncthreads();

    PHASE(5);
    // ---------------- phase F: Newton solver (a-2.6)
    const float tol = m.tolerance, scale = 1.0f / (m.meaninertia * (nv > 1 ? nv : 1));
    float cost = 0, Ma_c = 0;
    bool zones_changed = true;
    // cost at acceleration a (one register per lane); leaves Ma_c, jar, gr, dw, zones and contact wrenches behind
    auto eval_at = [&](float ac) -> float {
        float ma = 0, unstable = 0.f;
        static_for<0, G>([&](auto kc) { constexpr int k = decltype(kc)::value; ma += Mrow[k] * gbcast<G, k>(ac); });
        Ma_c = isdof ? ma : 0.f;
        float part = isdof ? 0.5f * (ac - qas_c) * (Ma_c - qfs_c) : 0.f;
#pragma unroll
        for (int side = 0; side < 2; side++) {
            if (lim_on[side]) {
                const float x = (side == 0 ? ac : -ac) - lim_aref[side];
                const bool was = lim_dw[side] != 0.f;
                if (x < 0) { part += 0.5f * lim_D[side] * x * x; lim_gr[side] = lim_D[side] * x; lim_dw[side] = lim_D[side]; unstable += was ? 0.f : 1.f; }
                else { lim_gr[side] = 0; lim_dw[side] = 0; unstable += was ? 1.f : 0.f; }
            }
        }
        jmul(ac, rJar);
        __syncthreads();
        if (c < ncon) {
            float *cr = con + C2_SIZE * c;
            const int adr = (int)cr[C2_ADR], dim = (int)cr[C2_DIM];
            float D[6], x[6];
            const float fri[5] = {cr[C2_F0], cr[C2_F0], cr[C2_F2], cr[C2_F3], cr[C2_F3]};
#pragma unroll
            for (int j = 0; j < 6; j++) if (j < dim) { D[j] = rD[adr + j]; x[j] = rJar[adr + j] - rAref[adr + j]; rJar[adr + j] = x[j]; } else { D[j] = 0; x[j] = 0; }
            ConeOut o;
            cone_eval2(dim, cr[C2_MU], fri, D, x, o);
            part += o.cost;
            if ((int)cr[C2_ZONE] != o.zone || o.zone == 2) unstable += 1.f;
            cr[C2_ZONE] = (float)o.zone; cr[C2_DM] = o.Dm; cr[C2_K3] = o.k3;
            const v3 nn = mk3(cr[C2_N], cr[C2_N + 1], cr[C2_N + 2]), t1 = mk3(cr[C2_T1], cr[C2_T1 + 1], cr[C2_T1 + 2]), t2 = mk3(cr[C2_T2], cr[C2_T2 + 1], cr[C2_T2 + 2]);
            const v3 fw = nn * o.g[0] + t1 * o.g[1] + t2 * o.g[2], tw = nn * o.g[3] + t1 * o.g[4] + t2 * o.g[5];
            cr[C2_FW] = fw.x; cr[C2_FW + 1] = fw.y; cr[C2_FW + 2] = fw.z; cr[C2_TW] = tw.x; cr[C2_TW + 1] = tw.y; cr[C2_TW + 2] = tw.z;
#pragma unroll
            for (int j = 0; j < 6; j++) { cr[C2_GN + j] = o.gn[j]; cr[C2_U + j] = o.u[j]; if (j < dim) { rGr[adr + j] = o.g[j]; rDw[adr + j] = o.dw[j]; } }
        }
        const float tot = gsum<G>(part);
        zones_changed = gsum<G>(unstable) > 0.f;
        __syncthreads();
        return tot;
    };

    bool active = valid && nefc > 0;
    int iter = 0;
    float qacc_c;
    {
        const float cost_s = eval_at(qas_c);
        const float cost_w = eval_at(warm_c);
        const bool use_warm = cost_w < cost_s;
        qacc_c = use_warm ? warm_c : qas_c;
        cost = use_warm ? cost_w : cost_s;
        if (__syncthreads_or(!use_warm)) cost = eval_at(qacc_c);
    }
    PHASE(6);
    for (int it = 0; it < m.iterations; it++) {
        if (!__syncthreads_or(active)) break;
        const float jtf = jt_force();
        const float grad_c = isdof ? Ma_c - qfs_c + jtf : 0.f;
        const float gnorm = sqrtf(gsum<G>(grad_c * grad_c));
        if (scale * gnorm < tol) active = false;
        PHASE(7);
        // Hessian rows H = M + J^T (d2s) J, lane = row c of H
        float Hrow[G];
        auto build_H = [&](bool with_neg) {
#pragma unroll
            for (int k = 0; k < G; k++) Hrow[k] = Mrow[k];
            if (active) {
                const float dl = (lim_on[0] ? lim_dw[0] : 0.f) + (lim_on[1] ? lim_dw[1] : 0.f);
#pragma unroll
                for (int k = 0; k < G; k++) Hrow[k] += (k == c) ? dl : 0.f;
                for (int ci = 0; ci < ncon; ci++) {
                    const float *cr = con + C2_SIZE * ci;
                    const int zone = (int)cr[C2_ZONE];
                    if (zone == 0) continue;
                    const int l1 = (int)cr[C2_L1], l2 = (int)cr[C2_L2], adr = (int)cr[C2_ADR], dim = (int)cr[C2_DIM];
                    const float sg = isdof ? (float)(((sMask[l2] >> c) & 1) - ((sMask[l1] >> c) & 1)) : 0.f;
                    const v3 pos = mk3(cr[C2_POS], cr[C2_POS + 1], cr[C2_POS + 2]);
                    const v3 P = (l_c + cross(a_c, pos - n_c)) * sg, Q = a_c * sg;
                    float pc = 0.f, wc = 0.f;
                    for (int j = 0; j < dim; j++) {
                        const int jj = j % 3;
                        const v3 ax = mk3(cr[3 + 3 * jj], cr[4 + 3 * jj], cr[5 + 3 * jj]);
                        const float jc = j < 3 ? dot(ax, P) : dot(ax, Q);
                        pc += jc * cr[C2_GN + j]; wc += jc * cr[C2_U + j];
                        const float w = rDw[adr + j];
                        if (w != 0.f) {
                            const float t = w * jc;
                            static_for<0, G>([&](auto kc) { constexpr int k = decltype(kc)::value; Hrow[k] += t * gbcast<G, k>(jc); });
                        }
                    }
                    if (zone == 2) {
                        const float Dm = cr[C2_DM], k3 = with_neg ? cr[C2_K3] : 0.f;
                        static_for<0, G>([&](auto kc) {
                            constexpr int k = decltype(kc)::value;
                            Hrow[k] += Dm * pc * gbcast<G, k>(pc) - k3 * wc * gbcast<G, k>(wc);
                        });
                    }
                }
            }
            if (!isdof) {
#pragma unroll
                for (int k = 0; k < G; k++) Hrow[k] = (k == c) ? 1.f : 0.f;
            }
        };
        build_H(true);
        PHASE(8);
        float hinvd;
        bool hfail = !chol_g<G>(Hrow, hinvd, nv, nv, c) && active;
        if (__syncthreads_or(hfail)) {
            build_H(!hfail);
            hfail = !chol_g<G>(Hrow, hinvd, nv, nv, c) && active;
            if (hfail) active = false;
        }
        float search_c = chol_solve_mf<G>(Hrow, hinvd, -grad_c, nv, c);
        if (!isdof) search_c = 0;
        PHASE(9);
        // exact line search (safeguarded 1-D Newton on phi')
        float mv = 0;
        static_for<0, G>([&](auto kc) { constexpr int k = decltype(kc)::value; mv += Mrow[k] * gbcast<G, k>(search_c); });
        const float g1 = gsum<G>(search_c * (Ma_c - qfs_c)), g2 = gsum<G>(isdof ? search_c * mv : 0.f), snorm = sqrtf(gsum<G>(search_c * search_c));
        jmul(search_c, rJv);
        __syncthreads();
        float alpha = 0;
        if (active) {
            float cD[6], cx[6], cv[6], cfri[5] = {0, 0, 0, 0, 0}, cmu = 0;
            int cdim = 0;
            if (c < ncon) {
                const float *cr = con + C2_SIZE * c;
                const int adr = (int)cr[C2_ADR];
                cdim = (int)cr[C2_DIM]; cmu = cr[C2_MU];
                cfri[0] = cr[C2_F0]; cfri[1] = cr[C2_F0]; cfri[2] = cr[C2_F2]; cfri[3] = cr[C2_F3]; cfri[4] = cr[C2_F3];
#pragma unroll
                for (int j = 0; j < 6; j++) if (j < cdim) { cD[j] = rD[adr + j]; cx[j] = rJar[adr + j]; cv[j] = rJv[adr + j]; } else { cD[j] = 0; cx[j] = 0; cv[j] = 0; }
            }
            // limit rows of this dof lane: residual at alpha = 0 and slope
            float lx[2], lv[2];
#pragma unroll
            for (int side = 0; side < 2; side++) { lx[side] = (side == 0 ? qacc_c : -qacc_c) - lim_aref[side]; lv[side] = side == 0 ? search_c : -search_c; }
            auto ls_eval = [&](float al, float &dphi, float &ddphi) {
                float dp = 0, hp = 0;
#pragma unroll
                for (int side = 0; side < 2; side++) {
                    if (lim_on[side]) {
                        const float x = lx[side] + al * lv[side];
                        if (x < 0) { dp += lim_D[side] * x * lv[side]; hp += lim_D[side] * lv[side] * lv[side]; }
                    }
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
            if (dp >= 0 || hp <= 0 || scale * 0.5f * (-dp) < tol) active = false;
            else {
                alpha = -dp / hp;
                for (int k = 0; k < m.ls_iterations; k++) {
                    ls_eval(alpha, dp, hp);
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
        if (active) qacc_c += alpha * search_c;
        const float newcost = eval_at(qacc_c);
        if (active) { iter++; cost = newcost; if (!zones_changed) active = false; }
        PHASE(11);
    }
    float qfc_c = 0;
    if (nefc == 0) qacc_c = qas_c;
    else qfc_c = -jt_force();
    PHASE(14);
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
        PHASE(15);
        acc_c = chol_solve_mf<G>(Ar, ainvd, qfs_c + qfc_c, nv, c);
    }
    PHASE(12);
    const float vnew = isdof ? qvel_c + h * acc_c : 0.f;
    if (!(fabsf(vnew) <= 1e10f)) bad = 1;
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
                sincosf(0.5f * angle, &sn, &cs);
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
