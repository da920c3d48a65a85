// Kinematics + collision kernels (fp32, one lane per env; collision: one wave per (64 envs, pair)).
//
// Reference path replaced: the position stage of mj_step reached through `self.sim.step()`
// (hsr/env.py:123): mj_kinematics and mj_collision.
#pragma once
#include "devmath.h"
#include "model.h"

// Kinematics + RNE velocity recursion of ONE env, serial over its links (mj_kinematics, mj_comPos, mj_comVel/mj_rne
// velocity part).  All arrays are Views so that the same code serves the one-lane-per-env kernel (LDS tile per lane)
// and the persistent per-env-group kernel (LDS record of the group).  Outputs: link poses, world dof axes/anchors,
// per-link com(3) Iworld(6) F(3) N(3) in ld; lw/lvo/lal/lao are scratch (12 floats per link).
// model-constant provider of the per-link kinematics (the persistent kernel has its own staged form in kin2.h)
struct KinGlobal {
    const DevModel &m;
    __device__ __forceinline__ int link_dofadr(int l) const { return m.link_dofadr[l]; }
    __device__ __forceinline__ int link_dofnum(int l) const { return m.link_dofnum[l]; }
    __device__ __forceinline__ int link_free(int l) const { return m.link_free[l]; }
    __device__ __forceinline__ int link_qposadr(int l) const { return m.link_qposadr[l]; }
    __device__ __forceinline__ int link_parent(int l) const { return m.link_parent[l]; }
    __device__ __forceinline__ v3 link_pos(int l) const { return ld3(m.link_pos, l); }
    __device__ __forceinline__ m3 link_mat(int l) const { return ldm(m.link_mat, l); }
    __device__ __forceinline__ v3 link_com(int l) const { return ld3(m.link_com, l); }
    __device__ __forceinline__ float link_mass(int l) const { return m.link_mass[l]; }
    __device__ __forceinline__ void link_inertia(int l, float *o) const {
#pragma unroll
        for (int i = 0; i < 6; i++) o[i] = m.link_inertia[6 * l + i]; }
    __device__ __forceinline__ int dof_qposadr(int k) const { return m.dof_qposadr[k]; }
    __device__ __forceinline__ int dof_type(int k) const { return m.dof_type[k]; }
    __device__ __forceinline__ v3 dof_axis(int k) const { return ld3(m.dof_axis, k); }
    __device__ __forceinline__ v3 dof_pos(int k) const { return ld3(m.dof_pos, k); }
};
__device__ __forceinline__ v3 sel3(bool c, v3 a, v3 b) { return mk3(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }

// pose of link l from its (already computed) parent and its joint coordinates; also the world axes of its dofs
template <class KC>
__device__ __forceinline__ void kin_link_pose(const KC &K, int l, View qpos, View xpos, View xmat, View ang, View lin, View anc) {
    v3 pos;
    m3 mat;
    const int d0 = K.link_dofadr(l), dn = K.link_dofnum(l);
    if (K.link_free(l)) {
        const int a = K.link_qposadr(l);
        pos = mk3(qpos[a], qpos[a + 1], qpos[a + 2]);
        q4 q;
        q.w = qpos[a + 3]; q.x = qpos[a + 4]; q.y = qpos[a + 5]; q.z = qpos[a + 6];
        q = qnormalized(q);                                  // mj_kinematics normalises in place
        qpos[a + 3] = q.w; qpos[a + 4] = q.x; qpos[a + 5] = q.y; qpos[a + 6] = q.z;
        mat = q2m(q);
        for (int k = 0; k < 3; k++) {
            lin.set3(d0 + k, mk3(k == 0, k == 1, k == 2));
            ang.set3(d0 + k, mk3(0, 0, 0));
            anc.set3(d0 + k, pos);
            ang.set3(d0 + 3 + k, col(mat, k));
            lin.set3(d0 + 3 + k, mk3(0, 0, 0));
            anc.set3(d0 + 3 + k, pos);
        }
    } else {
        const int p = K.link_parent(l);
        const m3 Rp = xmat.getm(p);
        pos = xpos.get3(p) + mulmv(Rp, K.link_pos(l));
        mat = mulmm(Rp, K.link_mat(l));
        for (int k = d0; k < d0 + dn; k++) {
            const float q = qpos[K.dof_qposadr(k)];
            const v3 ax = K.dof_axis(k);
            if (K.dof_type(k) == DOF_SLIDE) {
                pos = pos + mulmv(mat, ax) * q;
            } else {
                const v3 jp = K.dof_pos(k);
                const v3 anchor = pos + mulmv(mat, jp);
                float sn, cs;
                fast_sincos(0.5f * q, &sn, &cs);
                q4 qr;
                qr.w = cs; qr.x = ax.x * sn; qr.y = ax.y * sn; qr.z = ax.z * sn;
                mat = mulmm(mat, q2m(qr));
                pos = anchor - mulmv(mat, jp);
            }
        }
        for (int k = d0; k < d0 + dn; k++) {
            const v3 ax = mulmv(mat, K.dof_axis(k));
            if (K.dof_type(k) == DOF_SLIDE) {
                lin.set3(k, ax); ang.set3(k, mk3(0, 0, 0)); anc.set3(k, pos);
            } else {
                ang.set3(k, ax); lin.set3(k, mk3(0, 0, 0)); anc.set3(k, pos + mulmv(mat, K.dof_pos(k)));
            }
        }
    }
    xpos.set3(l, pos);
    xmat.setm(l, mat);
}

// velocity / bias acceleration (qacc = 0) of link l from its parent's, and the per-link wrench of mj_rne:
// F = m (a_com - g), N = I alpha + w x I w  (consumed by the solver as bias = J^T [F; N])
template <class KC>
__device__ __forceinline__ void kin_link_dyn(const KC &K, float gravz, int l, View qvel, View xpos, View xmat, View ang, View lin, View anc, View ld,
                                             View lw, View lvo, View lal, View lao) {
    const int d0 = K.link_dofadr(l);
    const m3 R = xmat.getm(l);
    const v3 xl = xpos.get3(l);
    v3 w, vo, al, ao;
    if (K.link_free(l)) {
        vo = mk3(qvel[d0], qvel[d0 + 1], qvel[d0 + 2]);
        w = mulmv(R, mk3(qvel[d0 + 3], qvel[d0 + 4], qvel[d0 + 5]));
        al = mk3(0, 0, 0); ao = mk3(0, 0, 0);
    } else {
        const int p = K.link_parent(l);
        const v3 wp = lw.get3(p), vop = lvo.get3(p), alp = lal.get3(p), aop = lao.get3(p);
        const v3 r = xl - xpos.get3(p);
        w = wp; al = alp;
        vo = vop + cross(wp, r);
        ao = aop + cross(alp, r) + cross(wp, cross(wp, r));
        for (int k = d0; k < d0 + K.link_dofnum(l); k++) {
            const float qd = qvel[k];
            if (K.dof_type(k) == DOF_SLIDE) {
                const v3 sx = lin.get3(k);
                vo = vo + sx * qd;
                ao = ao + cross(wp, sx) * (2 * qd);
            } else {
                const v3 a = ang.get3(k);
                const v3 rho = xl - anc.get3(k), rc = r - rho;
                const v3 wl = w + a * qd, all = al + cross(w, a) * qd;
                const v3 ac = aop + cross(alp, rc) + cross(wp, cross(wp, rc));
                const v3 vc = vop + cross(wp, rc);
                ao = ac + cross(all, rho) + cross(wl, cross(wl, rho));
                vo = vc + cross(wl, rho);
                w = wl; al = all;
            }
        }
    }
    lw.set3(l, w); lvo.set3(l, vo); lal.set3(l, al); lao.set3(l, ao);
    float li[6];
    K.link_inertia(l, li);
    m3 Il, Rt;
    Il.a[0] = li[0]; Il.a[1] = li[3]; Il.a[2] = li[4]; Il.a[3] = li[3]; Il.a[4] = li[1]; Il.a[5] = li[5]; Il.a[6] = li[4]; Il.a[7] = li[5]; Il.a[8] = li[2];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Rt.a[3 * i + j] = R.a[3 * j + i];
    const m3 I = mulmm(mulmm(R, Il), Rt);
    const v3 com = xl + mulmv(R, K.link_com(l)), rc = com - xl;
    const v3 acom = ao + cross(al, rc) + cross(w, cross(w, rc));
    const v3 F = (acom - mk3(0, 0, gravz)) * K.link_mass(l);
    const v3 Nt = mulmv(I, al) + cross(w, mulmv(I, w));
    const int b = 15 * l;
    ld[b] = com.x; ld[b + 1] = com.y; ld[b + 2] = com.z;
    ld[b + 3] = I.a[0]; ld[b + 4] = I.a[4]; ld[b + 5] = I.a[8]; ld[b + 6] = I.a[1]; ld[b + 7] = I.a[2]; ld[b + 8] = I.a[5];
    ld[b + 9] = F.x; ld[b + 10] = F.y; ld[b + 11] = F.z; ld[b + 12] = Nt.x; ld[b + 13] = Nt.y; ld[b + 14] = Nt.z;
}

__device__ __forceinline__ void kin_link0(View xpos, View xmat, View ld, View lw, View lvo, View lal, View lao) {
    m3 I;
#pragma unroll
    for (int k = 0; k < 9; k++) I.a[k] = (k % 4 == 0) ? 1.f : 0.f;
    xpos.set3(0, mk3(0, 0, 0));
    xmat.setm(0, I);
    lw.set3(0, mk3(0, 0, 0)); lvo.set3(0, mk3(0, 0, 0)); lal.set3(0, mk3(0, 0, 0)); lao.set3(0, mk3(0, 0, 0));
    for (int k = 0; k < 15; k++) ld[k] = 0;
}

// Kinematics + RNE velocity recursion of ONE env, serial over its links (mj_kinematics, mj_comPos, mj_comVel / velocity
// part of mj_rne).  All arrays are Views: the one-lane-per-env kernel passes its LDS tile, the persistent kernel runs the
// same per-link functions with lane = link, one tree level at a time.  Outputs: link poses, world dof axes / anchors,
// per-link com(3) Iworld(6) F(3) N(3) in ld; lw/lvo/lal/lao are scratch (3 floats per link each).
__device__ void kin_env(const DevModel &m, View qpos, View qvel, View xpos, View xmat, View ang, View lin, View anc, View ld,
                        View lw, View lvo, View lal, View lao) {
    kin_link0(xpos, xmat, ld, lw, lvo, lal, lao);
    const KinGlobal K{m};
    for (int l = 1; l < m.nlink; l++) kin_link_pose(K, l, qpos, xpos, xmat, ang, lin, anc);
    for (int l = 1; l < m.nlink; l++) kin_link_dyn(K, m.gravz, l, qvel, xpos, xmat, ang, lin, anc, ld, lw, lvo, lal, lao);
}

// ------------------------------------------------------------------ kinematics (a-2.1)
__global__ void __launch_bounds__(64) k_kinematics(DevModel m, DevState s) {
    // The solver's inputs (dof axes, per-link wrenches) are produced per lane in an LDS tile [64][kstride+1] and
    // written out env-major with coalesced stores, so that the solver's 16-lane groups read whole 256-B lines
    // instead of 4-byte pieces of 64 different lines.
    extern __shared__ float ktile[];
    if (blockIdx.x == 0) for (int i = threadIdx.x; i < m.npair_pad; i += 64) s.pair_count[i] = 0;   // work lists of k_cull
    const int e_raw = blockIdx.x * 64 + threadIdx.x;
    const bool live = e_raw < s.N && !s.done[e_raw < s.N ? e_raw : 0];
    const int e = live ? e_raw : 0;
    // per-lane LDS record: [kstride: solver inputs | 3 nlink xpos | 9 nlink xmat | 12 nlink link velocity state] (+1 pad)
    const int N = s.N, TS = s.kstride + 24 * m.nlink + 1;
    float *tl = ktile + threadIdx.x * TS;
    if (live) {
    View qpos{s.qpos + e, N}, xpos{tl + s.kstride, 1}, xmat{tl + s.kstride + 3 * m.nlink, 1};
    View ang{tl, 1}, lin{tl + 3 * m.nv, 1}, anc{tl + 6 * m.nv, 1};
    View qvel{s.qvel + e, N}, ld{tl + 9 * m.nv, 1};
    View lw{tl + s.kstride + 12 * m.nlink, 1}, lvo{tl + s.kstride + 15 * m.nlink, 1}, lal{tl + s.kstride + 18 * m.nlink, 1}, lao{tl + s.kstride + 21 * m.nlink, 1};
    kin_env(m, qpos, qvel, xpos, xmat, ang, lin, anc, ld, lw, lvo, lal, lao);
    for (int i = 9 * m.nv + 15 * m.nlink; i < s.kstride; i++) tl[i] = 0.f;
    }
    __syncthreads();
    // coalesced env-major write of the 64 records of this workgroup
    const int e0 = blockIdx.x * 64, KS = s.kstride;
    const unsigned long long livemask = __ballot(live);
    for (int le = 0; le < 64; le++) {
        if (!((livemask >> le) & 1ull)) continue;
        float *dst = s.kin_aos + (size_t)(e0 + le) * KS;
        const float *src = ktile + le * TS;
        for (int i = threadIdx.x; i < KS; i += 64) dst[i] = src[i];
    }
    if (live) {   // link poses for k_collide / body_xpos: struct-of-arrays, one coalesced store per row
        for (int i = 0; i < 3 * m.nlink; i++) s.xpos[(size_t)i * N + e] = tl[s.kstride + i];
        for (int i = 0; i < 9 * m.nlink; i++) s.xmat[(size_t)i * N + e] = tl[s.kstride + 3 * m.nlink + i];
        for (int i = 0; i < 6 * m.nlink; i++) s.lvel[(size_t)i * N + e] = tl[s.kstride + 12 * m.nlink + i];
    }
}

// ------------------------------------------------------------------ collision (a-2.3)
struct Geom {
    int type, nvert;
    v3 pos, size;
    v3 bc, bh;                // bounding box in the geom frame: world centre, half extents (tight for hulls)
    m3 mat;
    const float4 *verts;      // hull vertices staged in LDS (xyz, w unused); all lanes of the wave share the mesh
};

__device__ __forceinline__ Geom load_geom_v(const DevModel &m, View xpos, View xmat, int g);
__device__ __forceinline__ Geom load_geom(const DevModel &m, const DevState &s, int g, int e) {
    return load_geom_v(m, View{s.xpos + e, s.N}, View{s.xmat + e, s.N}, g);
}
__device__ __forceinline__ Geom load_geom_v(const DevModel &m, View xpos, View xmat, int g) {
    Geom G;
    const int l = m.geom_link[g];
    const m3 R = xmat.getm(l);
    G.pos = xpos.get3(l) + mulmv(R, ld3(m.geom_pos, g));
    G.mat = mulmm(R, ldm(m.geom_mat, g));
    G.type = m.geom_type[g];
    G.size = ld3(m.geom_size, g);
    G.bc = G.pos + mulmv(G.mat, mk3(m.geom_aabb[6 * g], m.geom_aabb[6 * g + 1], m.geom_aabb[6 * g + 2]));
    G.bh = mk3(m.geom_aabb[6 * g + 3], m.geom_aabb[6 * g + 4], m.geom_aabb[6 * g + 5]);
    G.verts = nullptr;
    G.nvert = m.geom_meshnum[g];
    return G;
}

// support point of a convex geom in world direction dir (ties resolved as in the oracle)
// W > 1: the W consecutive lanes of a sub-group hold the SAME geom and direction and split the hull scan between them
// (lane i takes vertices i, i + W, ...); a DPP butterfly picks the maximum, lowest index on ties - the same vertex the
// sequential scan returns.
enum { VID_BITS = 12, VID_NONE = (1 << VID_BITS) - 1 };      // a portal vertex id: three of them (+ 1) fit the three float rows of DevState::sepax as bit patterns
template <int W> __device__ __forceinline__ void sup_merge(float &b, int &idx, float pb, int pi) {
    // bitwise, not short-circuit: `a || (b && c)` came back from the compiler as two nested exec-mask branches per merge step
    const bool take = (pb > b) | ((pb == b) & (pi < idx));
    b = take ? pb : b; idx = take ? pi : idx;
}
// vid (optional): which vertex it was - mesh: its index; box: the three sign bits; VID_NONE for the shapes without vertices (12 bits per shape: hulls of up to 4094 vertices; MAXMESHV bounds them far below).  A vertex id
// names the same material point of the geom on the next substep (support_vertex), which is what the portal warm start needs.
__device__ __forceinline__ v3 support_vertex(const Geom &G, int vid) {
    v3 loc;
    if (G.type == GEOM_BOX) loc = mk3((vid & 1) ? G.size.x : -G.size.x, (vid & 2) ? G.size.y : -G.size.y, (vid & 4) ? G.size.z : -G.size.z);
    else { const float4 w = G.verts[vid]; loc = mk3(w.x, w.y, w.z); }
    return mulmv(G.mat, loc) + G.pos;
}
template <int W = 1> __device__ __forceinline__ v3 support(const Geom &G, v3 dir, int *vid = nullptr) {
    const v3 dl = mulmtv(G.mat, dir);
    v3 loc;
    if (vid) *vid = VID_NONE;
    if (G.type == GEOM_BOX) {
        loc = mk3(dl.x > 0 ? G.size.x : -G.size.x, dl.y > 0 ? G.size.y : -G.size.y, dl.z > 0 ? G.size.z : -G.size.z);
        if (vid) *vid = (dl.x > 0 ? 1 : 0) | (dl.y > 0 ? 2 : 0) | (dl.z > 0 ? 4 : 0);
    } else if (G.type == GEOM_CYLINDER) {
        const float r2 = dl.x * dl.x + dl.y * dl.y, ir = frsq(r2);
        if (r2 > HSR_MINVAL * HSR_MINVAL) { loc.x = dl.x * ir * G.size.x; loc.y = dl.y * ir * G.size.x; } else { loc.x = 0; loc.y = 0; }
        loc.z = dl.z > 0 ? G.size.y : -G.size.y;
    } else if (G.type == GEOM_SPHERE) {
        loc = normalized(dl) * G.size.x;
    } else {
        // mesh hull: exhaustive search, first maximum wins (same tie-break as a sequential scan).  Two independent
        // (value, index) chains over even/odd vertices give the VALU two dependency chains; the winning vertex is
        // fetched once at the end.  Vertices are LDS broadcast reads.
      if constexpr (W > 1) {
        const int sub = threadIdx.x & (W - 1);
        float b = -3.0e38f;
        int idx = 0x7fffffff;
        // the sub-group scans are only used by the persistent kernel: the hulls of the deepest links (the fingers) are staged in LDS, the others stay in
        // global memory (kin2.h: geom_cached3), so these are generic (flat) loads - a global_load here measured no faster (round 4)
        const float *gverts = (const float *)G.verts;
        auto gvert = [&](int i) { return make_float4(gverts[4 * i], gverts[4 * i + 1], gverts[4 * i + 2], 0.f); };
        for (int i0 = 0; i0 < G.nvert; i0 += 4 * W) {
            float4 q[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const int i = i0 + u * W + sub; q[u] = gvert(i < G.nvert ? i : 0); }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = i0 + u * W + sub;
                const float t = q[u].x * dl.x + q[u].y * dl.y + q[u].z * dl.z;
                if (i < G.nvert && t > b) { b = t; idx = i; }
            }
        }
        sup_merge<W>(b, idx, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b), 0xB1, 0xf, 0xf, false)), __builtin_amdgcn_update_dpp(0, idx, 0xB1, 0xf, 0xf, false));
        if constexpr (W >= 4) sup_merge<W>(b, idx, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b), 0x4E, 0xf, 0xf, false)), __builtin_amdgcn_update_dpp(0, idx, 0x4E, 0xf, 0xf, false));
        if constexpr (W >= 8) sup_merge<W>(b, idx, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b), 0x141, 0xf, 0xf, false)), __builtin_amdgcn_update_dpp(0, idx, 0x141, 0xf, 0xf, false));
        if constexpr (W >= 16) sup_merge<W>(b, idx, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b), 0x140, 0xf, 0xf, false)), __builtin_amdgcn_update_dpp(0, idx, 0x140, 0xf, 0xf, false));
        const float4 w = gvert(G.nvert > 0 ? idx : 0);
        loc = mk3(w.x, w.y, w.z);
        if (vid) *vid = G.nvert > 0 ? idx : 0;
      } else {
        float bA = -3.0e38f, bB = -3.0e38f;
        int iA = 0, iB = 1;
        int i = 0;
        for (; i + 8 <= G.nvert; i += 8) {
            // 8 vertex loads in flight before the first use (the table may live in global memory / L1)
            float4 q[8];
#pragma unroll
            for (int u = 0; u < 8; u++) q[u] = G.verts[i + u];
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                const float ta = q[u].x * dl.x + q[u].y * dl.y + q[u].z * dl.z, tb = q[u + 1].x * dl.x + q[u + 1].y * dl.y + q[u + 1].z * dl.z;
                iA = ta > bA ? i + u : iA; bA = fmaxf(bA, ta);
                iB = tb > bB ? i + u + 1 : iB; bB = fmaxf(bB, tb);
            }
        }
        for (; i < G.nvert; i++) {
            const float4 a = G.verts[i];
            const float t = a.x * dl.x + a.y * dl.y + a.z * dl.z;
            if (i & 1) { iB = t > bB ? i : iB; bB = fmaxf(bB, t); } else { iA = t > bA ? i : iA; bA = fmaxf(bA, t); }
        }
        const int best = (bB > bA || (bB == bA && iB < iA)) ? iB : iA;
        const float4 w = G.verts[G.nvert > 0 ? best : 0];
        loc = mk3(w.x, w.y, w.z);
        if (vid) *vid = G.nvert > 0 ? best : 0;
      }
    }
    return mulmv(G.mat, loc) + G.pos;
}

struct ContactOut {
    float *con;    // contact records of this env: [slot][8]
    int slot, cnt, maxcnt;
    __device__ __forceinline__ void add(v3 pos, v3 n, float dist) {
        if (cnt >= maxcnt) return;
        float4 *r = reinterpret_cast<float4 *>(con + (size_t)(slot + cnt) * 8);
        r[0] = make_float4(pos.x, pos.y, pos.z, n.x);
        r[1] = make_float4(n.y, n.z, dist, 0.f);
        cnt++;
    }
};

// --- mjc_PlaneBox: corners below the plane, at most 4
__device__ __forceinline__ void collide_plane_box(const Geom &P, const Geom &B, ContactOut &out) {
    const v3 n = col(P.mat, 2);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const v3 loc = mk3((i & 1) ? B.size.x : -B.size.x, (i & 2) ? B.size.y : -B.size.y, (i & 4) ? B.size.z : -B.size.z);
        const v3 c = mulmv(B.mat, loc) + B.pos;
        const float dist = dot(c - P.pos, n);
        if (dist < 0 && out.cnt < 4) out.add(c - n * (0.5f * dist), n, dist);
    }
}

// --- mjc_PlaneConvex: the deepest support point; when the pair has room for more (ContactOut::maxcnt: the compiler's plane_convex_points
// = 4) up to three more - support points along -n tilted towards three tangent directions 120 degrees apart, kept when they lie below the
// plane and are not a point already kept.  Restated from MuJoCo's behaviour; tilt and duplicate distance as in the oracle (oracle/hsr_oracle.c).
__device__ __forceinline__ void collide_plane_convex(const Geom &P, const Geom &Cx, ContactOut &out) {
    const v3 n = col(P.mat, 2);
    const v3 p = support(Cx, -n);
    const float dist = dot(p - P.pos, n);
    if (!(dist < 0)) return;
    out.add(p - n * (0.5f * dist), n, dist);
    if (out.maxcnt <= 1) return;
    v3 t1 = (n.y > -0.5f && n.y < 0.5f) ? mk3(0, 1, 0) : mk3(0, 0, 1);       // mju_makeFrame
    t1 = normalized(t1 - n * dot(n, t1));
    const v3 t2 = cross(n, t1);
    v3 kept[4];
    kept[0] = p;
    int nk = 1;
    const float cs[3][2] = {{1.f, 0.f}, {-0.5f, 0.8660254f}, {-0.5f, -0.8660254f}};
#pragma unroll
    for (int i = 0; i < 3; i++) {
        if (nk >= out.maxcnt) break;
        const v3 q = support(Cx, (t1 * cs[i][0] + t2 * cs[i][1]) * 0.1f - n);
        const float dq = dot(q - P.pos, n);
        bool take = dq < 0;
#pragma unroll
        for (int j = 0; j < 4; j++) if (j < nk) { const v3 e = q - kept[j]; take = take && !(dot(e, e) < 1e-10f); }
        if (take) {
            out.add(q - n * (0.5f * dq), n, dq);
#pragma unroll
            for (int j = 1; j < 4; j++) if (j == nk) kept[j] = q;
            nk++;
        }
    }
}

// --- box-box: SAT + reference-face clipping; polygon scratch lives in LDS ([buf][vertex][xyz][lane])
#define POLY(buf, i, k) poly[(((buf) * 8 + (i)) * 3 + (k)) * pstride + poff]
__device__ __forceinline__ int clip_poly(float *poly, int pstride, int poff, int src, int n, v3 axis, float lim, v3 origin) {
    const int dst = src ^ 1;
    int no = 0;
    for (int i = 0; i < n; i++) {
        const int i2 = (i + 1 == n) ? 0 : i + 1;
        const v3 a = mk3(POLY(src, i, 0), POLY(src, i, 1), POLY(src, i, 2));
        const v3 b = mk3(POLY(src, i2, 0), POLY(src, i2, 1), POLY(src, i2, 2));
        const float da = dot(a - origin, axis) - lim, db = dot(b - origin, axis) - lim;
        if (da <= 0 && no < 8) { POLY(dst, no, 0) = a.x; POLY(dst, no, 1) = a.y; POLY(dst, no, 2) = a.z; no++; }
        if (((da < 0 && db > 0) || (da > 0 && db < 0)) && no < 8) {
            const float t = da * frcp(da - db);
            POLY(dst, no, 0) = a.x + t * (b.x - a.x); POLY(dst, no, 1) = a.y + t * (b.y - a.y); POLY(dst, no, 2) = a.z + t * (b.z - a.z);
            no++;
        }
    }
    return no;
}

// polygon scratch addressing: element (buf, vertex, xyz) lives at poly[((buf*8+vertex)*3+xyz)*pstride + poff]
// (k_narrow: lane-interleaved pstride 64, poff lane; persistent kernel: private 48-float slot, pstride 1, poff 0)
__device__ __forceinline__ void collide_box_box_p(const Geom &G1, const Geom &G2, ContactOut &out, float *poly, int pstride, int poff) {
    v3 A[3], B[3];
    float s1[3] = {G1.size.x, G1.size.y, G1.size.z}, s2[3] = {G2.size.x, G2.size.y, G2.size.z};
#pragma unroll
    for (int i = 0; i < 3; i++) { A[i] = col(G1.mat, i); B[i] = col(G2.mat, i); }
    const v3 dv = G2.pos - G1.pos;
    float C[3][3], aC[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) { C[i][j] = dot(A[i], B[j]); aC[i][j] = fabsf(C[i][j]); }
    float best = -3.0e38f;
    int code = -1;
    v3 bestn = mk3(0, 0, 0);
    bool sepfound = false;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float t = dot(dv, A[i]);
        const float sep = fabsf(t) - (s1[i] + s2[0] * aC[i][0] + s2[1] * aC[i][1] + s2[2] * aC[i][2]);
        if (sep > 0) sepfound = true;
        if (sep > best) { best = sep; code = i; bestn = A[i] * (t < 0 ? -1.f : 1.f); }
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const float t = dot(dv, B[j]);
        const float sep = fabsf(t) - (s2[j] + s1[0] * aC[0][j] + s1[1] * aC[1][j] + s1[2] * aC[2][j]);
        if (sep > 0) sepfound = true;
        if (sep > best) { best = sep; code = 3 + j; bestn = B[j] * (t < 0 ? -1.f : 1.f); }
    }
    if (sepfound) return;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            v3 L = cross(A[i], B[j]);
            const float ln = norm(L);
            if (ln < 1e-6f) continue;
            L = L * frcp(ln);
            const float t = dot(dv, L);
            float ra = 0, rb = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) { ra += s1[k] * fabsf(dot(A[k], L)); rb += s2[k] * fabsf(dot(B[k], L)); }
            const float sep = fabsf(t) - (ra + rb);
            if (sep > 0) sepfound = true;
            if (sep * 1.05f > best + 1e-9f) { best = sep; code = 6 + 3 * i + j; bestn = L * (t < 0 ? -1.f : 1.f); }
        }
    if (sepfound) return;
    if (code < 6) {
        const bool ref1 = code < 3;
        const int ax = ref1 ? code : code - 3;
        const v3 pr = ref1 ? G1.pos : G2.pos, pi = ref1 ? G2.pos : G1.pos;
        v3 Ar[3], Ai[3];
        float sr[3], si[3];
#pragma unroll
        for (int k = 0; k < 3; k++) { Ar[k] = ref1 ? A[k] : B[k]; Ai[k] = ref1 ? B[k] : A[k]; sr[k] = ref1 ? s1[k] : s2[k]; si[k] = ref1 ? s2[k] : s1[k]; }
        const v3 nref = bestn * (ref1 ? 1.f : -1.f);
        int jx = 0;
        float bd = -1.f;
#pragma unroll
        for (int j = 0; j < 3; j++) { const float t = fabsf(dot(nref, Ai[j])); if (t > bd) { bd = t; jx = j; } }
        // select axes without dynamic register indexing
        const v3 Aj = jx == 0 ? Ai[0] : (jx == 1 ? Ai[1] : Ai[2]);
        const v3 Aj1 = jx == 0 ? Ai[1] : (jx == 1 ? Ai[2] : Ai[0]);
        const v3 Aj2 = jx == 0 ? Ai[2] : (jx == 1 ? Ai[0] : Ai[1]);
        const float sj = jx == 0 ? si[0] : (jx == 1 ? si[1] : si[2]);
        const float sj1 = jx == 0 ? si[1] : (jx == 1 ? si[2] : si[0]);
        const float sj2 = jx == 0 ? si[2] : (jx == 1 ? si[0] : si[1]);
        const float sgn = dot(nref, Aj) > 0 ? -1.f : 1.f;
        const v3 fc = pi + Aj * (sgn * sj);
        const float sg[4][2] = {{1, 1}, {-1, 1}, {-1, -1}, {1, -1}};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const v3 v = fc + Aj1 * (sg[k][0] * sj1) + Aj2 * (sg[k][1] * sj2);
            POLY(0, k, 0) = v.x; POLY(0, k, 1) = v.y; POLY(0, k, 2) = v.z;
        }
        const v3 Au = ax == 0 ? Ar[1] : (ax == 1 ? Ar[2] : Ar[0]);
        const v3 Av = ax == 0 ? Ar[2] : (ax == 1 ? Ar[0] : Ar[1]);
        const float su = ax == 0 ? sr[1] : (ax == 1 ? sr[2] : sr[0]);
        const float sv = ax == 0 ? sr[2] : (ax == 1 ? sr[0] : sr[1]);
        const float sa = ax == 0 ? sr[0] : (ax == 1 ? sr[1] : sr[2]);
        int np = 4, buf = 0;
        np = clip_poly(poly, pstride, poff, buf, np, Au, su, pr); buf ^= 1;
        if (np) { np = clip_poly(poly, pstride, poff, buf, np, -Au, su, pr); buf ^= 1; }
        if (np) { np = clip_poly(poly, pstride, poff, buf, np, Av, sv, pr); buf ^= 1; }
        if (np) { np = clip_poly(poly, pstride, poff, buf, np, -Av, sv, pr); buf ^= 1; }
        for (int k = 0; k < np; k++) {
            const v3 v = mk3(POLY(buf, k, 0), POLY(buf, k, 1), POLY(buf, k, 2));
            const float dist = dot(v - pr, nref) - sa;
            if (dist < 0) out.add(v - nref * (0.5f * dist), bestn, dist);
        }
    } else {
        const int i = (code - 6) / 3, j = (code - 6) % 3;
        v3 pa = G1.pos, pb = G2.pos;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (k != i) pa = pa + A[k] * ((dot(bestn, A[k]) > 0 ? 1.f : -1.f) * s1[k]);
            if (k != j) pb = pb + B[k] * ((dot(bestn, B[k]) > 0 ? -1.f : 1.f) * s2[k]);
        }
        const v3 Ai_ = i == 0 ? A[0] : (i == 1 ? A[1] : A[2]);
        const v3 Bj_ = j == 0 ? B[0] : (j == 1 ? B[1] : B[2]);
        const v3 w = pa - pb;
        const float b = dot(Ai_, Bj_), dd = dot(Ai_, w), ee = dot(Bj_, w), den = 1.f - b * b;
        const float t = den > 1e-12f ? (b * ee - dd) / den : 0.f, uu = den > 1e-12f ? (ee - b * dd) / den : 0.f;
        const v3 qa = pa + Ai_ * t, qb = pb + Bj_ * uu;
        out.add((qa + qb) * 0.5f, bestn, best);
    }
}
#undef POLY
__device__ __forceinline__ void collide_box_box(const Geom &G1, const Geom &G2, ContactOut &out, float *poly, int lane) { collide_box_box_p(G1, G2, out, poly, 64, lane); }

// --- box-box for an 8-lane sub-group (all 8 lanes hold the same pair): the same routine as collide_box_box_p, element for
// element, with the expensive parts spread over the lanes: the nine edge-edge axes are evaluated one per lane and folded in
// the sequential order afterwards; the clipped polygon lives one vertex per lane, a clip is a neighbour exchange, a width-8
// scan of the emitted counts and a scatter through 24 floats of LDS.  Returns the number of contacts written.
__device__ __forceinline__ v3 shfl3_8(v3 a, int src) { return mk3(__shfl(a.x, src, 8), __shfl(a.y, src, 8), __shfl(a.z, src, 8)); }
// DPP moves inside an 8-lane sub-group (two sub-groups per 16-lane DPP row): broadcast of sub-group lane Q, value of the next
// lane, value D lanes below (callers guard the lanes that would read across the sub-group boundary)
template <int CTRL> __device__ __forceinline__ int dppi_(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL> __device__ __forceinline__ float dppf_(float v) { return __builtin_bit_cast(float, dppi_<CTRL>(__builtin_bit_cast(int, v))); }
template <int Q> __device__ __forceinline__ int sg8_bcast(int v) { const int lo = dppi_<0x150 + Q>(v), hi = dppi_<0x158 + Q>(v); return (threadIdx.x & 8) ? hi : lo; }
template <int Q> __device__ __forceinline__ float sg8_bcast(float v) { return __builtin_bit_cast(float, sg8_bcast<Q>(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ float sg8_next(float v) { return dppf_<0x101>(v); }
struct NoStamp { __device__ __forceinline__ void operator()(int) const {} };
template <class ST = NoStamp> __device__ __forceinline__ int collide_box_box_w8(const Geom &G1, const Geom &G2, float *con, int slot, int maxcnt, float *scr, ST stamp = ST()) {
    const int sub = threadIdx.x & 7;
    v3 A[3], B[3];
    float s1[3] = {G1.size.x, G1.size.y, G1.size.z}, s2[3] = {G2.size.x, G2.size.y, G2.size.z};
#pragma unroll
    for (int i = 0; i < 3; i++) { A[i] = col(G1.mat, i); B[i] = col(G2.mat, i); }
    // opaque copies for the lane-dependent picks (`i == 0 ? A0 : ...`): written on the arrays themselves the picks come back from the optimiser as ONE
    // load with a computed index - and an array indexed that way cannot stay in registers: both rotation matrices went to scratch memory and every
    // substep of every workgroup paid a scratch store + dependent scratch loads for the block resting on the table
    v3 A0 = A[0], A1 = A[1], A2 = A[2], B0 = B[0], B1 = B[1], B2 = B[2];
    asm volatile("" : "+v"(A0.x), "+v"(A0.y), "+v"(A0.z), "+v"(A1.x), "+v"(A1.y), "+v"(A1.z), "+v"(A2.x), "+v"(A2.y), "+v"(A2.z));
    asm volatile("" : "+v"(B0.x), "+v"(B0.y), "+v"(B0.z), "+v"(B1.x), "+v"(B1.y), "+v"(B1.z), "+v"(B2.x), "+v"(B2.y), "+v"(B2.z));
    const v3 dv = G2.pos - G1.pos;
    float aC[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) aC[i][j] = fabsf(dot(A[i], B[j]));
    float best = -3.0e38f;
    int code = -1;
    v3 bestn = mk3(0, 0, 0);
    bool sepfound = false;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float t = dot(dv, A[i]);
        const float sep = fabsf(t) - (s1[i] + s2[0] * aC[i][0] + s2[1] * aC[i][1] + s2[2] * aC[i][2]);
        if (sep > 0) sepfound = true;
        if (sep > best) { best = sep; code = i; bestn = A[i] * (t < 0 ? -1.f : 1.f); }
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const float t = dot(dv, B[j]);
        const float sep = fabsf(t) - (s2[j] + s1[0] * aC[0][j] + s1[1] * aC[1][j] + s1[2] * aC[2][j]);
        if (sep > 0) sepfound = true;
        if (sep > best) { best = sep; code = 3 + j; bestn = B[j] * (t < 0 ? -1.f : 1.f); }
    }
    if (sepfound) return 0;
    stamp(28);
    // edge axes q = 3 i + j: lane q evaluates axis q, lane 0 also axis 8
    float esep[2];
    int eok[2];
    v3 eL[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int q = r == 0 ? sub : 8, i = q / 3, j = q - 3 * i;
        const v3 Ai = i == 0 ? A0 : (i == 1 ? A1 : A2), Bj = j == 0 ? B0 : (j == 1 ? B1 : B2);
        v3 L = cross(Ai, Bj);
        const float ln = norm(L);
        eok[r] = ln < 1e-6f ? 0 : 1;
        L = L * frcp(ln);
        const float t = dot(dv, L);
        float ra = 0, rb = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) { ra += s1[k] * fabsf(dot(A[k], L)); rb += s2[k] * fabsf(dot(B[k], L)); }
        esep[r] = fabsf(t) - (ra + rb);
        eL[r] = L * (t < 0 ? -1.f : 1.f);
    }
    // The fold over the nine edge axes in their sequential order only ever RAISES `best`, so an axis that cannot beat the face axes' best
    // cannot beat any later one either: when no lane of the sub-group holds
    // an axis that separates or passes the first test (a face contact - the block resting on the table) the fold changes nothing and is skipped.
    const bool cand = (eok[0] && (esep[0] > 0 || esep[0] * 1.05f > best + 1e-9f)) || ((threadIdx.x & 7) == 0 && eok[1] && (esep[1] > 0 || esep[1] * 1.05f > best + 1e-9f));
    if ((((unsigned)(__ballot(cand) >> (threadIdx.x & 56))) & 0xffu) != 0u) {
    static_for<0, 9>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const float sep = q < 8 ? sg8_bcast<q & 7>(esep[0]) : sg8_bcast<0>(esep[1]);
        const int ok = q < 8 ? sg8_bcast<q & 7>(eok[0]) : sg8_bcast<0>(eok[1]);
        if (ok) {
            if (sep > 0) sepfound = true;
            if (sep * 1.05f > best + 1e-9f) { best = sep; code = 6 + q; }
        }
    });
    }
    if (sepfound) return 0;
    if (code >= 6) bestn = code < 14 ? shfl3_8(eL[0], code - 6) : shfl3_8(eL[1], 0);
    stamp(29);
    if (code < 6) {
        const bool ref1 = code < 3;
        const int ax = ref1 ? code : code - 3;
        const v3 pr = ref1 ? G1.pos : G2.pos, pi = ref1 ? G2.pos : G1.pos;
        v3 Ar[3], Ai[3];
        float sr[3], si[3];
#pragma unroll
        for (int k = 0; k < 3; k++) { Ar[k] = ref1 ? A[k] : B[k]; Ai[k] = ref1 ? B[k] : A[k]; sr[k] = ref1 ? s1[k] : s2[k]; si[k] = ref1 ? s2[k] : s1[k]; }
        const v3 nref = bestn * (ref1 ? 1.f : -1.f);
        int jx = 0;
        float bd = -1.f;
#pragma unroll
        for (int j = 0; j < 3; j++) { const float t = fabsf(dot(nref, Ai[j])); if (t > bd) { bd = t; jx = j; } }
        const v3 Aj = jx == 0 ? Ai[0] : (jx == 1 ? Ai[1] : Ai[2]);
        const v3 Aj1 = jx == 0 ? Ai[1] : (jx == 1 ? Ai[2] : Ai[0]);
        const v3 Aj2 = jx == 0 ? Ai[2] : (jx == 1 ? Ai[0] : Ai[1]);
        const float sj = jx == 0 ? si[0] : (jx == 1 ? si[1] : si[2]);
        const float sj1 = jx == 0 ? si[1] : (jx == 1 ? si[2] : si[0]);
        const float sj2 = jx == 0 ? si[2] : (jx == 1 ? si[0] : si[1]);
        const float sgn = dot(nref, Aj) > 0 ? -1.f : 1.f;
        const v3 fc = pi + Aj * (sgn * sj);
        // incident face, vertex k on lane k: signs (1,1) (-1,1) (-1,-1) (1,-1)
        const float sgx = (sub == 0 || sub == 3) ? 1.f : -1.f, sgy = sub < 2 ? 1.f : -1.f;
        v3 P = fc + Aj1 * (sgx * sj1) + Aj2 * (sgy * sj2);
        const v3 Au = ax == 0 ? Ar[1] : (ax == 1 ? Ar[2] : Ar[0]);
        const v3 Av = ax == 0 ? Ar[2] : (ax == 1 ? Ar[0] : Ar[1]);
        const float su = ax == 0 ? sr[1] : (ax == 1 ? sr[2] : sr[0]);
        const float sv = ax == 0 ? sr[2] : (ax == 1 ? sr[0] : sr[1]);
        const float sa = ax == 0 ? sr[0] : (ax == 1 ? sr[1] : sr[2]);
        int np = 4;
        auto clip = [&](v3 axis, float lim) {
            const bool wrap = sub + 1 >= np;
            // both moves are executed by every lane of the sub-group, then selected: a DPP move under a per-lane condition
            // would read lanes that sit in the other branch
            const v3 nx = mk3(sg8_next(P.x), sg8_next(P.y), sg8_next(P.z)), fx = mk3(sg8_bcast<0>(P.x), sg8_bcast<0>(P.y), sg8_bcast<0>(P.z));
            const v3 b = mk3(wrap ? fx.x : nx.x, wrap ? fx.y : nx.y, wrap ? fx.z : nx.z);
            const float da = dot(P - pr, axis) - lim, db = dot(b - pr, axis) - lim;
            const bool have = sub < np;
            const int in = (have && da <= 0) ? 1 : 0, cr = (have && ((da < 0 && db > 0) || (da > 0 && db < 0))) ? 1 : 0;
            int incl = in + cr;
            { const int t1 = dppi_<0x111>(incl); if (sub >= 1) incl += t1;
              const int t2 = dppi_<0x112>(incl); if (sub >= 2) incl += t2;
              const int t4 = dppi_<0x114>(incl); if (sub >= 4) incl += t4; }
            const int base = incl - in - cr, total = sg8_bcast<7>(incl);
            if (in && base < 8) { scr[3 * base] = P.x; scr[3 * base + 1] = P.y; scr[3 * base + 2] = P.z; }
            if (cr && base + in < 8) {
                const float t = da * frcp(da - db);
                const int o = 3 * (base + in);
                scr[o] = P.x + t * (b.x - P.x); scr[o + 1] = P.y + t * (b.y - P.y); scr[o + 2] = P.z + t * (b.z - P.z);
            }
            np = total < 8 ? total : 8;
            __builtin_amdgcn_wave_barrier();
            P = mk3(scr[3 * sub], scr[3 * sub + 1], scr[3 * sub + 2]);
            __builtin_amdgcn_wave_barrier();
        };
        // Resting contact (the block on the table): the whole incident face lies inside the four side planes of the reference face.  The four
        // clips then copy the polygon unchanged - no vertex is outside (da <= 0 everywhere), so none is dropped and no edge is cut - and are
        // skipped; `side` is the very expression the clip evaluates.
        auto side = [&](v3 axis, float lim) { return dot(P - pr, axis) - lim; };
        const bool in4 = sub >= 4 || (side(Au, su) <= 0 && side(-Au, su) <= 0 && side(Av, sv) <= 0 && side(-Av, sv) <= 0);
        const bool all_in = (((unsigned)(__ballot(in4) >> (threadIdx.x & 56))) & 0xffu) == 0xffu;
        if (!all_in)
        {
        clip(Au, su);
        if (np) clip(-Au, su);
        if (np) clip(Av, sv);
        if (np) clip(-Av, sv);
        }
        stamp(30);
        const float dist = dot(P - pr, nref) - sa;
        const bool hit = sub < np && dist < 0;
        const unsigned int hm = (unsigned int)((__ballot(hit) >> (threadIdx.x & 56)) & 0xffull);
        const int rank = __popc(hm & ((1u << sub) - 1u)), nh = __popc(hm);
        if (hit && rank < maxcnt) {
            const v3 pos = P - nref * (0.5f * dist);
            float4 *r = reinterpret_cast<float4 *>(con + (size_t)(slot + rank) * 8);
            r[0] = make_float4(pos.x, pos.y, pos.z, bestn.x);
            r[1] = make_float4(bestn.y, bestn.z, dist, 0.f);
        }
        return nh < maxcnt ? nh : maxcnt;
    }
    {
        const int i = (code - 6) / 3, j = (code - 6) % 3;
        v3 pa = G1.pos, pb = G2.pos;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (k != i) pa = pa + A[k] * ((dot(bestn, A[k]) > 0 ? 1.f : -1.f) * s1[k]);
            if (k != j) pb = pb + B[k] * ((dot(bestn, B[k]) > 0 ? -1.f : 1.f) * s2[k]);
        }
        const v3 Ai_ = i == 0 ? A0 : (i == 1 ? A1 : A2);
        const v3 Bj_ = j == 0 ? B0 : (j == 1 ? B1 : B2);
        const v3 w = pa - pb;
        const float b = dot(Ai_, Bj_), dd = dot(Ai_, w), ee = dot(Bj_, w), den = 1.f - b * b;
        const float t = den > 1e-12f ? (b * ee - dd) / den : 0.f, uu = den > 1e-12f ? (ee - b * dd) / den : 0.f;
        const v3 qa = pa + Ai_ * t, qb = pb + Bj_ * uu, pos = (qa + qb) * 0.5f;
        if (sub == 0 && maxcnt > 0) {
            float4 *r = reinterpret_cast<float4 *>(con + (size_t)slot * 8);
            r[0] = make_float4(pos.x, pos.y, pos.z, bestn.x);
            r[1] = make_float4(bestn.y, bestn.z, best, 0.f);
        }
        return maxcnt > 0 ? 1 : 0;
    }
}

// --- convex-convex: Minkowski Portal Refinement (libccd ccdMPRPenetration as used by mjc_Convex)
struct Sup { v3 v, v1, v2; int id; };      // id: vertex ids of the two shapes (support()), id1 | id2 << VID_BITS
#ifdef HSR_PHASE_TIMING
__device__ int g_dbg_nsup_lane;   // unused placeholder to keep the symbol table stable
#define DBG_COUNT_SUPPORT(ctr) (ctr)++
#else
#define DBG_COUNT_SUPPORT(ctr) do {} while (0)
#endif
template <int W = 1> __device__ __forceinline__ Sup mpr_support(const Geom &G1, const Geom &G2, v3 dir) {
    Sup s;
    int i1, i2;
    s.v1 = support<W>(G1, dir, &i1);
    s.v2 = support<W>(G2, -dir, &i2);
    s.id = i1 | (i2 << VID_BITS);
    s.v = s.v1 - s.v2;
    return s;
}
__device__ __forceinline__ float point_seg_dist2(v3 x0, v3 b, v3 &wit) {   // P = origin
    const v3 d = b - x0;
    const float t = -dot(x0, d) / dot(d, d);
    if (t < 0 || fabsf(t) < 1e-30f) wit = x0;
    else if (t >= 1) wit = b;
    else wit = x0 + d * t;
    return dot(wit, wit);
}
// libccd's ccdVec3PointTriDist2 for P = origin.  fp32 note: libccd's expanded quadratic form
// s^2 v + t^2 w + 2 s t r + 2 s p + 2 t q + u cancels catastrophically in fp32 (u ~ 0.1, result ~ 1e-8),
// so the interior case only reports `interior` and the caller measures the depth along the portal normal.
__device__ __forceinline__ float point_tri_dist2(v3 x0, v3 B, v3 Cc, v3 &wit, bool &interior) {
    const v3 d1 = B - x0, d2 = Cc - x0, a = x0;
    const float v = dot(d1, d1), w = dot(d2, d2), p = dot(a, d1), q = dot(a, d2), r = dot(d1, d2);
    const float den = w * v - r * r;
    float sx = -1, t = -1;
    interior = false;
    if (fabsf(den) > 0) { sx = (q * r - w * p) / den; t = (-sx * r - q) / w; }
    if (sx >= 0 && sx <= 1 && t >= 0 && t <= 1 && t + sx <= 1) {
        wit = x0 + d1 * sx + d2 * t;
        interior = true;
        return dot(wit, wit);
    }
    v3 w2;
    float best = point_seg_dist2(x0, B, wit);
    float dist = point_seg_dist2(x0, Cc, w2);
    if (dist < best) { best = dist; wit = w2; }
    dist = point_seg_dist2(B, Cc, w2);
    if (dist < best) { best = dist; wit = w2; }
    return best;
}
__device__ __forceinline__ v3 portal_dir(const Sup &p1, const Sup &p2, const Sup &p3) {
    return normalized(cross(p2.v - p1.v, p3.v - p1.v));
}
// branch-free selects keep the portal in registers (struct assignment under divergent ifs made the compiler
// place the three portal points in a scratch array with dynamic indexing: every MPR step went through memory)
__device__ __forceinline__ Sup selS(bool c, const Sup &a, const Sup &b) {
    Sup r;
    r.v = sel3(c, a.v, b.v); r.v1 = sel3(c, a.v1, b.v1); r.v2 = sel3(c, a.v2, b.v2); r.id = c ? a.id : b.id;
    return r;
}
__device__ __forceinline__ void expand_portal(const Sup &p0, Sup &p1, Sup &p2, Sup &p3, const Sup &v4) {
    const v3 v4v0 = cross(v4.v, p0.v);
    const bool d1 = dot(p1.v, v4v0) > 0, d2 = dot(p2.v, v4v0) > 0, d3 = dot(p3.v, v4v0) > 0;
    // d1: (d2 ? p1 : p3) <- v4 ; !d1: (d3 ? p2 : p1) <- v4
    const bool w1 = (d1 && d2) || (!d1 && !d3), w2 = !d1 && d3, w3 = d1 && !d2;
    p1 = selS(w1, v4, p1); p2 = selS(w2, v4, p2); p3 = selS(w3, v4, p3);
}
__device__ __forceinline__ bool reach_tol(const Sup &p1, const Sup &p2, const Sup &p3, const Sup &v4, v3 dir, float tol) {
    const float dv4 = dot(v4.v, dir);
    const float mn = fminf(fminf(dv4 - dot(p1.v, dir), dv4 - dot(p2.v, dir)), dv4 - dot(p3.v, dir));
    return mn < tol;
}

// returns true on penetration; otherwise `sep` is a proven separating direction of the Minkowski difference
// (support(A-B, sep) . sep < 0) or zero when MPR gave up without one
// warm: in: the vertex ids of last substep's final portal of this pair (three Sup::id, 0 = none); out: this run's.
// The same six material points, placed with the present poses, are a portal again whenever the origin ray still passes through their
// triangle (checked; else the search starts from scratch): the refinement then confirms the face it ended on last time with one or
// two support calls instead of rediscovering it with eight.
template <int W = 1> __device__ __forceinline__ bool mpr_penetration(const Geom &G1, const Geom &G2, float tol, int maxit, float &depth, v3 &dirout, v3 &pos, v3 &sep, int &nsup, int *warm = nullptr) {
    const float eps = HSR_EPS;
    Sup p0, p1, p2, p3, v4;
    p0.v1 = G1.pos; p0.v2 = G2.pos; p0.v = p0.v1 - p0.v2; p0.id = 0;
    if (fabsf(p0.v.x) < eps && fabsf(p0.v.y) < eps && fabsf(p0.v.z) < eps) p0.v.x += 1e-5f;
    v3 dir = normalized(-p0.v);
    sep = mk3(0, 0, 0);
    nsup = 0;
    bool warm_ok = false;
    if (warm && warm[0] > 0) {
        auto rebuild = [&](int id, Sup &p) { p.id = id; p.v1 = support_vertex(G1, id & VID_NONE); p.v2 = support_vertex(G2, (id >> VID_BITS) & VID_NONE); p.v = p.v1 - p.v2; };
        rebuild(warm[0] - 1, p1); rebuild(warm[1] - 1, p2); rebuild(warm[2] - 1, p3);
        // the ray from p0 through the origin crosses the triangle, with the winding the refinement expects (the three tests of the
        // portal discovery below), by a clear margin
        const float m1 = 1e-4f * norm(p0.v);
        const float s13 = dot(cross(p1.v, p3.v), p0.v), s32 = dot(cross(p3.v, p2.v), p0.v), s21 = dot(cross(p2.v, p1.v), p0.v);
        const float sc = fmaxf(fmaxf(dot(p1.v, p1.v), dot(p2.v, p2.v)), dot(p3.v, p3.v)) * m1;
        warm_ok = s13 > sc && s32 > sc && s21 > sc;
    }
    if (warm) { warm[0] = warm[1] = warm[2] = 0; }
  cold_start:
  if (!warm_ok) {
    p1 = mpr_support<W>(G1, G2, dir); nsup++;
    if (dot(p1.v, dir) < eps) { sep = dir; return false; }
    dir = cross(p0.v, p1.v);
    if (dot(dir, dir) < eps * eps) {
        const float l1 = norm(p1.v);
        if (l1 < eps) return false;
        depth = l1; dirout = normalized(p1.v); pos = (p1.v1 + p1.v2) * 0.5f;
        return true;
    }
    dir = normalized(dir);
    p2 = mpr_support<W>(G1, G2, dir); nsup++;
    if (dot(p2.v, dir) < eps) { sep = dir; return false; }
    dir = normalized(cross(p1.v - p0.v, p2.v - p0.v));
    {
        const bool sw = dot(dir, p0.v) > 0;
        const Sup t1 = selS(sw, p2, p1), t2 = selS(sw, p1, p2);
        p1 = t1; p2 = t2; dir = sel3(sw, -dir, dir);
    }
    for (int it = 0;; it++) {
        if (it > 100) return false;
        p3 = mpr_support<W>(G1, G2, dir); nsup++;
        if (dot(p3.v, dir) < eps) { sep = dir; return false; }
        const bool c1 = dot(cross(p1.v, p3.v), p0.v) < -eps;
        const bool c2 = !c1 && dot(cross(p3.v, p2.v), p0.v) < -eps;
        p2 = selS(c1, p3, p2);
        p1 = selS(c2, p3, p1);
        if (!(c1 || c2)) break;
        dir = normalized(cross(p1.v - p0.v, p2.v - p0.v));
    }
  }
    for (int it = 0;; it++) {
        dir = portal_dir(p1, p2, p3);
        if (dot(dir, p1.v) >= -eps) break;
        v4 = mpr_support<W>(G1, G2, dir); nsup++;
        if (dot(v4.v, dir) < -eps) { sep = dir; return false; }
        if (reach_tol(p1, p2, p3, v4, dir, tol) || it > maxit) return false;
        expand_portal(p0, p1, p2, p3, v4);
    }
    for (int it = 0;; it++) {
        dir = portal_dir(p1, p2, p3);
        v4 = mpr_support<W>(G1, G2, dir); nsup++;
        if (reach_tol(p1, p2, p3, v4, dir, tol) || it > maxit) {
            v3 pdir;
            bool interior;
            depth = fsqrt(point_tri_dist2(p1.v, p2.v, p3.v, pdir, interior));
            // Where the origin projects into the final triangle, depth and direction are those of the face's plane - the same for every
            // triangle of that face, however the run got there.  Where it does not, libccd measures to the triangle's edge and the answer
            // depends on the triangle: a warm-started run then starts over from scratch, so that it ends where libccd's own search does
            // (as far as single precision follows it).
            if (!interior && warm_ok) { warm_ok = false; dir = normalized(-p0.v); goto cold_start; }
            if (interior) {
                // the witness is the foot of the perpendicular: depth = |n . v1|, direction = +-n
                const float dn = dot(dir, p1.v);
                depth = fabsf(dn);
                dirout = dn >= 0 ? dir : -dir;
            } else {
                if (fabsf(pdir.x) < eps && fabsf(pdir.y) < eps && fabsf(pdir.z) < eps) pdir = dir;
                dirout = normalized(pdir);
            }
            // mpr_find_pos
            float b0 = dot(cross(p1.v, p2.v), p3.v), b1 = dot(cross(p3.v, p2.v), p0.v);
            float b2 = dot(cross(p0.v, p1.v), p3.v), b3 = dot(cross(p2.v, p1.v), p0.v);
            float sum = b0 + b1 + b2 + b3;
            if (sum <= 0) {
                b0 = 0; b1 = dot(cross(p2.v, p3.v), dir); b2 = dot(cross(p3.v, p1.v), dir); b3 = dot(cross(p1.v, p2.v), dir);
                sum = b1 + b2 + b3;
            }
            const float inv = 0.5f * frcp(sum);
            pos = (p0.v1 * b0 + p1.v1 * b1 + p2.v1 * b2 + p3.v1 * b3 + p0.v2 * b0 + p1.v2 * b1 + p2.v2 * b2 + p3.v2 * b3) * inv;
            // the portal this run ended on, for the next substep - only vertices carry over (mesh, box)
            auto vertex_ids = [](int id) { return (id & VID_NONE) != VID_NONE && ((id >> VID_BITS) & VID_NONE) != VID_NONE; };
            // (only from a run whose witness was interior: one that ended on a triangle edge would be started over next time anyway)
            if (warm && interior && vertex_ids(p1.id) && vertex_ids(p2.id) && vertex_ids(p3.id)) { warm[0] = p1.id + 1; warm[1] = p2.id + 1; warm[2] = p3.id + 1; }
            return true;
        }
        expand_portal(p0, p1, p2, p3, v4);
    }
}

// conservative culls on the oriented bounding boxes (centre G.bc, axes G.mat, half extents G.bh)
__device__ __forceinline__ bool sphere_hits_obb(v3 c, float r, const Geom &B) {
    const v3 l = mulmtv(B.mat, c - B.bc);
    const float dx = fmaxf(fabsf(l.x) - B.bh.x, 0.f), dy = fmaxf(fabsf(l.y) - B.bh.y, 0.f), dz = fmaxf(fabsf(l.z) - B.bh.z, 0.f);
    return dx * dx + dy * dy + dz * dz <= r * r;
}
// separating-axis test of two oriented boxes (15 axes); false = certainly disjoint
__device__ __forceinline__ bool obb_overlap(const Geom &A, const Geom &B) {
    const v3 d = B.bc - A.bc;
    const float ha[3] = {A.bh.x, A.bh.y, A.bh.z}, hb[3] = {B.bh.x, B.bh.y, B.bh.z};
    v3 Aa[3], Ba[3];
    float C[3][3], aC[3][3], ta[3];
#pragma unroll
    for (int i = 0; i < 3; i++) { Aa[i] = col(A.mat, i); Ba[i] = col(B.mat, i); }
#pragma unroll
    for (int i = 0; i < 3; i++) {
        ta[i] = dot(d, Aa[i]);
#pragma unroll
        for (int j = 0; j < 3; j++) { C[i][j] = dot(Aa[i], Ba[j]); aC[i][j] = fabsf(C[i][j]) + 1e-6f; }
    }
    bool sep = false;
#pragma unroll
    for (int i = 0; i < 3; i++) sep |= fabsf(ta[i]) > ha[i] + hb[0] * aC[i][0] + hb[1] * aC[i][1] + hb[2] * aC[i][2];
#pragma unroll
    for (int j = 0; j < 3; j++) sep |= fabsf(ta[0] * C[0][j] + ta[1] * C[1][j] + ta[2] * C[2][j]) > hb[j] + ha[0] * aC[0][j] + ha[1] * aC[1][j] + ha[2] * aC[2][j];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            const float ra = ha[i1] * aC[i2][j] + ha[i2] * aC[i1][j], rb = hb[j1] * aC[i][j2] + hb[j2] * aC[i][j1];
            sep |= fabsf(ta[i2] * C[i1][j] - ta[i1] * C[i2][j]) > ra + rb;
        }
    return !sep;
}

// ---- two-stage collision: cull + compaction, then dense narrowphase ------------------------------------------------
// Stage 1 (k_cull): one wave = 64 consecutive envs x one pair at a time.  Lanes whose pair survives the bounding tests are
// compacted with a wave ballot and appended (one atomic per wave) to the pair's work list; every (env, pair) count is
// reset to 0.  Stage 2 (k_narrow): one wave = 64 work items of ONE pair (so the narrowphase function, geom constants and
// the LDS-staged hull vertices stay wave-uniform) with all lanes busy, instead of a few active lanes per wave.
// List order depends on atomic arrival order, results do not: every item writes only its own (env, slot) records.
#define MAXMESHV 256

__device__ __forceinline__ bool pair_cull(const DevModel &m, const Geom &G1, const Geom &G2, int g1, int g2) {
    // mj_collideGeoms bounding-sphere test (margin 0), then tighter conservative culls on oriented bounding boxes
    // (sphere vs box, box vs box SAT).  A contact needs the geoms to intersect, which implies every test passes, so
    // the contact set is unchanged; the culls only spare the narrowphase.
    if (G1.type == GEOM_PLANE) {
        const v3 n = col(G1.mat, 2);
        if (!(dot(G2.pos - G1.pos, n) <= m.geom_rbound[g2])) return false;
        return dot(G2.bc - G1.pos, n) - (fabsf(dot(n, col(G2.mat, 0))) * G2.bh.x + fabsf(dot(n, col(G2.mat, 1))) * G2.bh.y + fabsf(dot(n, col(G2.mat, 2))) * G2.bh.z) <= 0.f;
    }
    const v3 r = G2.pos - G1.pos;
    const float r1 = m.geom_rbound[g1], r2 = m.geom_rbound[g2], b = r1 + r2;
    if (!(dot(r, r) <= b * b)) return false;
    if (!(sphere_hits_obb(G2.pos, r2, G1) && sphere_hits_obb(G1.pos, r1, G2))) return false;
    return obb_overlap(G1, G2);
}

__global__ void __launch_bounds__(64) k_cull(DevModel m, DevState s) {
    const int lane = threadIdx.x;
    const int e = blockIdx.x * 64 + lane;
    const bool live = e < s.N && !s.done[e < s.N ? e : 0];
    if (!__any(live)) return;
    const int es = live ? e : 0;
    const int N = s.N;
    for (int pv = blockIdx.y; pv < m.npair; pv += gridDim.y) {
        const int p = __builtin_amdgcn_readfirstlane(pv);
        const int g1 = m.pair_geom1[p], g2 = m.pair_geom2[p];
        const Geom G1 = load_geom(m, s, g1, es), G2 = load_geom(m, s, g2, es);
        bool pass = live && pair_cull(m, G1, G2, g1, g2);
        const unsigned long long bal = __ballot(pass);
        if (bal) {
            const int first = __ffsll((long long)bal) - 1;
            int base = 0;
            if (lane == first) base = atomicAdd(&s.pair_count[p], __popcll(bal));
            base = __shfl(base, first);
            if (pass) s.pair_list[(size_t)p * N + base + __popcll(bal & ((1ull << lane) - 1ull))] = e;
        }
        if (live) s.ncon_pair[(size_t)e * m.npair_pad + p] = 0;
    }
}

__global__ void __launch_bounds__(64) k_narrow(DevModel m, DevState s) {
    __shared__ float poly[2 * 8 * 3 * 64];
    __shared__ float4 vbuf[2][MAXMESHV];
    __shared__ int sPre[384];
    const int lane = threadIdx.x;
    const int N = s.N;
    // inclusive prefix of 64-item chunks per pair (npair <= 384), every wave computes the same table
    int run = 0;
    for (int base = 0; base < m.npair; base += 64) {
        const int p = base + lane;
        int ch = p < m.npair ? (s.pair_count[p] + 63) >> 6 : 0;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(ch, off); if (lane >= off) ch += t; }
        if (p < m.npair) sPre[p] = run + ch;
        run += __shfl(ch, 63);
    }
    __syncthreads();
    const int total = run;
    for (int cid = blockIdx.x; cid < total; cid += gridDim.x) {
        // pair of this chunk: first p with sPre[p] > cid (binary search over the shared table; wave-uniform)
        int lo = 0, hi = m.npair - 1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (sPre[mid] > cid) hi = mid; else lo = mid + 1; }
        const int p = __builtin_amdgcn_readfirstlane(lo);
        const int k = (cid - (p > 0 ? sPre[p - 1] : 0)) * 64 + lane;
        const int cnt = s.pair_count[p];
        const bool act = k < cnt;
        const int e = s.pair_list[(size_t)p * N + (act ? k : 0)];
        const int g1 = m.pair_geom1[p], g2 = m.pair_geom2[p];
        Geom G1 = load_geom(m, s, g1, e), G2 = load_geom(m, s, g2, e);
        ContactOut out;
        out.con = s.con + (size_t)e * m.nslot * 8;
        out.slot = m.pair_slot[p];
        out.maxcnt = m.pair_slot[p + 1] - m.pair_slot[p];
        out.cnt = 0;
        const int fn = m.pair_fn[p];
        __syncthreads();
        if (fn == FN_PLANE_CONVEX || fn == FN_CONVEX) {
            if (G1.type == GEOM_MESH) { const float4 *src = m.mesh_vert4 + m.geom_meshadr[g1]; for (int i = lane; i < G1.nvert; i += 64) vbuf[0][i] = src[i]; G1.verts = vbuf[0]; }
            if (G2.type == GEOM_MESH) { const float4 *src = m.mesh_vert4 + m.geom_meshadr[g2]; for (int i = lane; i < G2.nvert; i += 64) vbuf[1][i] = src[i]; G2.verts = vbuf[1]; }
        }
        __syncthreads();
        if (act) {
            if (fn == FN_PLANE_BOX) collide_plane_box(G1, G2, out);
            else if (fn == FN_PLANE_CONVEX) collide_plane_convex(G1, G2, out);
            else if (fn == FN_BOX_BOX) collide_box_box(G1, G2, out, poly, lane);
            else {
                // temporal coherence: the separating direction MPR proved last time is tried first; if it still
                // separates (two support calls), MPR would again report "no intersection" - identical result
                float *sx = s.sepax + (size_t)(4 * p) * N + e;          // rows 4 p .. 4 p + 2: direction; row 4 p + 3: margin cache of the persistent kernel
                const v3 d = mk3(sx[0], sx[N], sx[2 * (size_t)N]);
                bool still = false;
                const bool is_dir = sx[3 * (size_t)N] >= 0.f;          // -1: the rows hold the portal vertex ids of the persistent kernel's warm start, not a direction
                if (is_dir && (d.x != 0.f || d.y != 0.f || d.z != 0.f)) still = dot(support(G1, d) - support(G2, -d), d) < -1e-7f;
                if (!still) {
                    float depth; v3 dir, pos, sep;
                    int nsup = 0;
                    if (mpr_penetration(G1, G2, m.mpr_tolerance, m.mpr_iterations, depth, dir, pos, sep, nsup)) { out.add(pos, dir, -depth); sep = mk3(0, 0, 0); }
                    sx[0] = sep.x; sx[N] = sep.y; sx[2 * (size_t)N] = sep.z;
                    if (!is_dir) sx[3 * (size_t)N] = 0.f;
#ifdef HSR_PHASE_TIMING
                    atomicAdd(&s.phase_cyc[20], (unsigned long long)nsup); atomicMax(&s.phase_cyc[21], (unsigned long long)nsup); atomicAdd(&s.phase_cyc[22], 1ull);
                    if (nsup > 20) atomicAdd(&s.phase_cyc[24], 1ull);
#endif
                }
#ifdef HSR_PHASE_TIMING
                else atomicAdd(&s.phase_cyc[23], 1ull);
#endif
            }
            s.ncon_pair[(size_t)e * m.npair_pad + p] = out.cnt;
        }
    }
}
