// Kinematics + RNE recursion of the persistent kernel (mj_kinematics, mj_comPos, velocity part of mj_rne behind
// hsr/env.py:123), lane = link / lane = dof, restructured so that only what really depends on the parent runs inside the
// tree-level loops:
//   A  lane = link   transform of the link in its parent's frame from its own joint coordinates (no parent data)
//   B  lane = link   world pose = product of the local transforms of the link's ancestors, root first: every lane walks its own
//                    (compile-time packed) ancestor list, all loads independent of the running product - no tree-level barriers
//   C  lane = dof    world axis / anchor of every dof from its link's pose; its velocity-field increment qd (a, lin - a x anchor)
//   D  lane = link   spatial velocity and bias acceleration (qacc = 0) of the link as FIELDS about the world origin,
//                    v(x) = u + w x x,  a(x) = g + al x x + w x (w x x), accumulated over the dofs of the link's chain, root first:
//                    w' = w + dw, al' = al + w x dw, u' = u + du, g' = g + (w + w') x du        (hinge and slide alike)
//   E  lane = link   world inertia, com, the per-link wrench F = m (a(com) - gravity), N = I al + w x I w of mj_rne
// Same quantities as kin_link_pose / kin_link_dyn (collide.h, per-substep chain kernel), which recurse parent -> child with the
// link origin as reference point; the two forms differ in rounding only.  Free bodies hang off the world and carry no children
// (checked on the host): w = R qvel_ang, al = 0, zero acceleration of the origin.
#pragma once
#include "collide.h"

// per-link constants in LDS, 16-B aligned groups:
//   0 dofadr dofnum free qposadr | 4 parent depth mass ancestors (4 bits each, nearest first, 0 = world: stop) | 8 lpos.xyz - | 12 lmat[9] - - - | 24 com.xyz - | 28 inertia[6] - - |
//   36 + 8 j (j < 3): qposadr type axis.xyz pos.xyz of the link's j-th scalar dof
enum { KIN2_FLOATS = 60 };
__device__ __forceinline__ void kin2_store(const DevModel &m, int l, float *o) {
    const int d0 = m.link_dofadr[l], dn = m.link_dofnum[l];
    o[0] = (float)d0; o[1] = (float)dn; o[2] = (float)m.link_free[l]; o[3] = (float)m.link_qposadr[l];
    o[4] = (float)m.link_parent[l]; o[5] = (float)m.link_depth[l]; o[6] = m.link_mass[l];
    unsigned pack = 0;
    { int a = m.link_parent[l]; for (int j = 0; j < 8 && a > 0; j++) { pack |= (unsigned)a << (4 * j); a = m.link_parent[a]; } }
    o[7] = __uint_as_float(pack);
    o[8] = m.link_pos[3 * l]; o[9] = m.link_pos[3 * l + 1]; o[10] = m.link_pos[3 * l + 2]; o[11] = 0.f;
#pragma unroll
    for (int i = 0; i < 9; i++) o[12 + i] = m.link_mat[9 * l + i];
    o[21] = o[22] = o[23] = 0.f;
    o[24] = m.link_com[3 * l]; o[25] = m.link_com[3 * l + 1]; o[26] = m.link_com[3 * l + 2]; o[27] = 0.f;
#pragma unroll
    for (int i = 0; i < 6; i++) o[28 + i] = m.link_inertia[6 * l + i];
    o[34] = o[35] = 0.f;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int k = d0 + (j < dn ? j : 0);
        float *q = o + 36 + 8 * j;
        q[0] = (float)m.dof_qposadr[k]; q[1] = (float)m.dof_type[k];
        q[2] = m.dof_axis[3 * k]; q[3] = m.dof_axis[3 * k + 1]; q[4] = m.dof_axis[3 * k + 2];
        q[5] = m.dof_pos[3 * k]; q[6] = m.dof_pos[3 * k + 1]; q[7] = m.dof_pos[3 * k + 2];
    }
}

__device__ __forceinline__ float4 kl4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void ks4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
// pose record of a link: R (row-major 9) then p (3), 16-B aligned
__device__ __forceinline__ void pose_load(const float *P, m3 &R, v3 &p) {
    const float4 a = kl4(P), b = kl4(P + 4), c = kl4(P + 8);
    R.a[0] = a.x; R.a[1] = a.y; R.a[2] = a.z; R.a[3] = a.w; R.a[4] = b.x; R.a[5] = b.y; R.a[6] = b.z; R.a[7] = b.w; R.a[8] = c.x;
    p = mk3(c.y, c.z, c.w);
}
__device__ __forceinline__ void pose_store(float *P, const m3 &R, v3 p) {
    ks4(P, make_float4(R.a[0], R.a[1], R.a[2], R.a[3])); ks4(P + 4, make_float4(R.a[4], R.a[5], R.a[6], R.a[7]));
    ks4(P + 8, make_float4(R.a[8], p.x, p.y, p.z));
}

// Per-lane state of the kinematics of one env (lane c of its G-lane group); the stages are separate calls so that the diagnostic
// build can stamp them.  kc: the workgroup's link constants (KIN2_FLOATS per link); dof_link: LDS table dof -> link; masks: chain
// dof masks per link.  poseL [12 nlink], recL [12 nlink] (local transforms, then w / v(origin) per link), dwL [8 nv] (scratch),
// qposL [nq], qvelL [nv] and the solver's kin record (kAng kLin kAnc [3 nv] each, lk [15 nlink]) are this env's LDS arrays.
struct Kin2 {
    const float *K;
    bool isl;
    int c, d0, dn, free_;
    unsigned anc;
    float mass;
    m3 R;
    v3 p, w, al, u, g;

    // A: the link in its parent's frame -> recL; levels <= 1 are already world poses.
    // Two rounds of LDS reads, both issued whole before anything is computed - the link's record with its three joint records, then the free joint's
    // seven coordinates and the three joint coordinates (a record beyond the link's dofnum repeats joint 0: a valid address, an unused value) - instead of
    // a loop whose every turn read a joint record and then the coordinate it names.  The joint loop is unrolled over the three joint slots a link can
    // have; a slot no link of the wave uses, and the hinge arm of a slot without a hinge, are skipped wave-uniformly; the slide arm (twelve
    // instructions) and the selects run unconditionally.  Same arithmetic as before, term by term: results are bit-identical.
    __device__ __forceinline__ void stageA(const float *kc, int nlink, int c_, float *qposL, float *recL) {
        c = c_; isl = c < nlink;
        K = kc + KIN2_FLOATS * (isl ? c : 0);
        const float4 h0 = kl4(K), h1 = kl4(K + 4), lp = kl4(K + 8), m0 = kl4(K + 12), m1 = kl4(K + 16), m2 = kl4(K + 20);
        float4 ja[3], jb[3];
#pragma unroll
        for (int j = 0; j < 3; j++) { ja[j] = kl4(K + 36 + 8 * j); jb[j] = kl4(K + 40 + 8 * j); }
        d0 = (int)h0.x; dn = (int)h0.y; free_ = (int)h0.z;
        const int qadr = (int)h0.w;
        mass = h1.z; anc = __float_as_uint(h1.w);
        float qf[7], qj[3];
#pragma unroll
        for (int i = 0; i < 7; i++) qf[i] = qposL[qadr + i];          // (a link without a free joint reads its neighbours' coordinates: in bounds, not used)
#pragma unroll
        for (int j = 0; j < 3; j++) qj[j] = qposL[(int)ja[j].x];
        // the chain of scalar joints
        v3 pc = mk3(lp.x, lp.y, lp.z);
        m3 Rc;
        Rc.a[0] = m0.x; Rc.a[1] = m0.y; Rc.a[2] = m0.z; Rc.a[3] = m0.w; Rc.a[4] = m1.x; Rc.a[5] = m1.y; Rc.a[6] = m1.z; Rc.a[7] = m1.w; Rc.a[8] = m2.x;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const bool on = !free_ && j < dn;
            if (!wave_any(on)) continue;
            const bool hinge = on && (int)ja[j].y != DOF_SLIDE;
            const v3 ax = mk3(ja[j].z, ja[j].w, jb[j].x);
            const v3 ps = pc + mulmv(Rc, ax) * qj[j];
            v3 ph = pc;
            m3 Rh = Rc;
            if (wave_any(hinge)) {
                const v3 jp = mk3(jb[j].y, jb[j].z, jb[j].w);
                const v3 anchor = pc + mulmv(Rc, jp);
                float sn, cs;
                fast_sincos(0.5f * qj[j], &sn, &cs);
                q4 qr;
                qr.w = cs; qr.x = ax.x * sn; qr.y = ax.y * sn; qr.z = ax.z * sn;
                Rh = mulmm(Rc, q2m(qr));
                ph = anchor - mulmv(Rh, jp);
            }
            pc = sel3(hinge, ph, sel3(on, ps, pc));
#pragma unroll
            for (int k = 0; k < 9; k++) Rc.a[k] = hinge ? Rh.a[k] : Rc.a[k];
        }
        p = pc; R = Rc;
        if (wave_any(free_ != 0)) {
            q4 q;
            q.w = qf[3]; q.x = qf[4]; q.y = qf[5]; q.z = qf[6];
            q = qnormalized(q);                                      // mj_kinematics normalises in place
            if (isl && free_) { qposL[qadr + 3] = q.w; qposL[qadr + 4] = q.x; qposL[qadr + 5] = q.y; qposL[qadr + 6] = q.z; }
            const m3 Rf = q2m(q);
            p = sel3(free_ != 0, mk3(qf[0], qf[1], qf[2]), pc);
#pragma unroll
            for (int k = 0; k < 9; k++) R.a[k] = free_ ? Rf.a[k] : Rc.a[k];
        }
        if (isl) pose_store(recL + 12 * c, R, p);
    }
    // B: world pose = T(root) ... T(parent) T(self); the world's own transform is the identity, so lanes with fewer ancestors
    // multiply by it (exactly) instead of leaving the loop
    __device__ __forceinline__ void stageB(int maxdepth, const float *recL, float *poseL) {
        for (int j = 0; j + 1 < maxdepth; j++) {
            const int a = (anc >> (4 * j)) & 15;
            m3 Ra; v3 pa;
            pose_load(recL + 12 * a, Ra, pa);
            p = pa + mulmv(Ra, p);
            R = mulmm(Ra, R);
        }
        if (isl) pose_store(poseL + 12 * c, R, p);
    }
    // C: lane = dof
    __device__ __forceinline__ void stageC(const float *kc, const unsigned char *dof_link, int nv, const float *poseL, const float *qvelL,
                                           float *kAng, float *kLin, float *kAnc, float *dwL) const {
        if (c < nv) {
            const int l = dof_link[c];
            const float *Kd = kc + KIN2_FLOATS * l;
            const float4 g0 = kl4(Kd);
            const int j = c - (int)g0.x;
            m3 Rl; v3 pl;
            pose_load(poseL + 12 * l, Rl, pl);
            v3 ang = mk3(0, 0, 0), lin = mk3(0, 0, 0), an = pl;
            if ((int)g0.z) {
                if (j < 3) lin = mk3(j == 0, j == 1, j == 2);
                else ang = mk3(j == 3 ? Rl.a[0] : (j == 4 ? Rl.a[1] : Rl.a[2]), j == 3 ? Rl.a[3] : (j == 4 ? Rl.a[4] : Rl.a[5]), j == 3 ? Rl.a[6] : (j == 4 ? Rl.a[7] : Rl.a[8]));
            } else {
                const float4 j0 = kl4(Kd + 36 + 8 * j), j1 = kl4(Kd + 40 + 8 * j);
                const v3 ax = mulmv(Rl, mk3(j0.z, j0.w, j1.x));
                if ((int)j0.y == DOF_SLIDE) lin = ax;
                else { ang = ax; an = pl + mulmv(Rl, mk3(j1.y, j1.z, j1.w)); }
            }
            kAng[3 * c] = ang.x; kAng[3 * c + 1] = ang.y; kAng[3 * c + 2] = ang.z;
            kLin[3 * c] = lin.x; kLin[3 * c + 1] = lin.y; kLin[3 * c + 2] = lin.z;
            kAnc[3 * c] = an.x; kAnc[3 * c + 1] = an.y; kAnc[3 * c + 2] = an.z;
            const float qd = qvelL[c];
            const v3 dw = ang * qd, du = (lin - cross(ang, an)) * qd;
            ks4(dwL + 8 * c, make_float4(dw.x, dw.y, dw.z, du.x)); ks4(dwL + 8 * c + 4, make_float4(du.y, du.z, 0.f, 0.f));
        }
    }
    // D: velocity / bias-acceleration fields; also leaves w and v(link origin) in recL for the 'openai' observation
    __device__ __forceinline__ void stageD(int mask, const float *qvelL, const float *dwL, float *recL) {
        w = mk3(0, 0, 0); al = w; u = w; g = w;
        if (free_) {
            const v3 vo = mk3(qvelL[d0], qvelL[d0 + 1], qvelL[d0 + 2]);
            w = mulmv(R, mk3(qvelL[d0 + 3], qvelL[d0 + 4], qvelL[d0 + 5]));
            u = vo - cross(w, p);
            g = -cross(w, cross(w, p));
        } else {
            for (int mm = isl ? mask : 0; mm; mm &= mm - 1) {
                const int k = __ffs(mm) - 1;
                const float4 a0 = kl4(dwL + 8 * k), a1 = kl4(dwL + 8 * k + 4);
                const v3 dw = mk3(a0.x, a0.y, a0.z), du = mk3(a0.w, a1.x, a1.y);
                al = al + cross(w, dw);
                g = g + cross(w + w + dw, du);
                w = w + dw; u = u + du;
            }
        }
        if (isl) {
            const v3 vo = u + cross(w, p);
            ks4(recL + 12 * c, make_float4(w.x, w.y, w.z, vo.x)); recL[12 * c + 4] = vo.y; recL[12 * c + 5] = vo.z;
        }
    }
    // E: world inertia, com and the link wrench
    __device__ __forceinline__ void stageE(float gravz, float *lk) const {
        if (!isl) return;
        float *o = lk + 15 * c;
        if (c == 0) {
#pragma unroll
            for (int i = 0; i < 15; i++) o[i] = 0.f;
            return;
        }
        const float4 cm = kl4(K + 24), i0 = kl4(K + 28), i1 = kl4(K + 32);
        m3 Il, Rt;
        Il.a[0] = i0.x; Il.a[1] = i0.w; Il.a[2] = i1.x; Il.a[3] = i0.w; Il.a[4] = i0.y; Il.a[5] = i1.y; Il.a[6] = i1.x; Il.a[7] = i1.y; Il.a[8] = i0.z;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) Rt.a[3 * i + j] = R.a[3 * j + i];
        const m3 I = mulmm(mulmm(R, Il), Rt);
        const v3 com = p + mulmv(R, mk3(cm.x, cm.y, cm.z));
        const v3 acom = g + cross(al, com) + cross(w, cross(w, com));
        const v3 F = (acom - mk3(0, 0, gravz)) * mass;
        const v3 Nt = mulmv(I, al) + cross(w, mulmv(I, w));
        o[0] = com.x; o[1] = com.y; o[2] = com.z;
        o[3] = I.a[0]; o[4] = I.a[4]; o[5] = I.a[8]; o[6] = I.a[1]; o[7] = I.a[2]; o[8] = I.a[5];
        o[9] = F.x; o[10] = F.y; o[11] = F.z; o[12] = Nt.x; o[13] = Nt.y; o[14] = Nt.z;
    }
};


// Geom cache of the persistent kernel.  World placement of a geom = 16 floats (pos3 mat9, then the box centre - or, for a
// plane, its normal): static geoms (link 0, a prefix of the geom list) are placed once per launch into a table shared by the
// envs of the workgroup, moving geoms once per substep into their env's LDS.  The constants the culls / narrowphases need
// (8 floats per geom: size3, type | nvert << 4 | meshadr << 13; box half extents3, bounding radius) live in a second shared
// table, so that nothing of the collision front end goes to global memory but the prefetched placement constants.
struct GeomPlaceC { float4 q1, q2, q3, q4, q5; int link, type; };     // lpos | lmat[0..3] | lmat[4..7] | lmat[8] size | aabb centre
__device__ __forceinline__ GeomPlaceC geom_place_consts(const float *rec) {
    const float4 *r4 = reinterpret_cast<const float4 *>(rec);
    GeomPlaceC k;
    const float4 q0 = r4[0];
    k.q1 = r4[1]; k.q2 = r4[2]; k.q3 = r4[3]; k.q4 = r4[4]; k.q5 = r4[5];
    k.link = (int)q0.x; k.type = (int)q0.y;
    return k;
}
// step_move: upper bound of how far any point of the geom moved since the previous substep (0 for a static geom); it feeds the
// separation-margin cache of the convex pairs (persist.h)
__device__ __forceinline__ void geom_place3(const GeomPlaceC &k, const m3 &R, v3 pl, float *out, float step_move = 0.f) {
    m3 lm;
    lm.a[0] = k.q2.x; lm.a[1] = k.q2.y; lm.a[2] = k.q2.z; lm.a[3] = k.q2.w; lm.a[4] = k.q3.x; lm.a[5] = k.q3.y; lm.a[6] = k.q3.z; lm.a[7] = k.q3.w; lm.a[8] = k.q4.x;
    const v3 pos = pl + mulmv(R, mk3(k.q1.x, k.q1.y, k.q1.z));
    const m3 mat = mulmm(R, lm);
    v3 bc = pos + mulmv(mat, mk3(k.q5.x, k.q5.y, k.q5.z));
    if (k.type == GEOM_PLANE) bc = col(mat, 2);
    ks4(out, make_float4(pos.x, pos.y, pos.z, mat.a[0])); ks4(out + 4, make_float4(mat.a[1], mat.a[2], mat.a[3], mat.a[4]));
    ks4(out + 8, make_float4(mat.a[5], mat.a[6], mat.a[7], mat.a[8])); ks4(out + 12, make_float4(bc.x, bc.y, bc.z, step_move));
}
__device__ __forceinline__ void geom_consts_store(const float *rec, float *out) {
    const float4 *r4 = reinterpret_cast<const float4 *>(rec);
    const float4 q0 = r4[0], q1 = r4[1], q4 = r4[4], q6 = r4[6];
    const unsigned pk = (unsigned)(int)q0.y | ((unsigned)(int)q0.z << 4) | ((unsigned)(int)q0.w << 13);
    ks4(out, make_float4(q4.y, q4.z, q4.w, __uint_as_float(pk))); ks4(out + 4, make_float4(q6.x, q6.y, q6.z, q1.w));
}
// Geom from its cached world placement w (16 floats) and its constants cc (8 floats), both in LDS
// hull_lds_patch: a geom whose hull the persistent kernel stages in LDS (DevModel::geom_ldsv) carries the float4 slot of its first vertex in place
// of its mesh address, and bit 31 says so
__device__ __forceinline__ void hull_lds_patch(float *out, int slot) {
    const unsigned pk = __float_as_uint(out[3]);
    out[3] = __uint_as_float((pk & 0x1fffu) | ((unsigned)slot << 13) | 0x80000000u);
}
__device__ __forceinline__ Geom geom_cached3(const float *w, const float *cc, const float4 *mesh_vert4, float &rbound, const float4 *ldsv = nullptr) {
    const float4 w0 = kl4(w), w1 = kl4(w + 4), w2 = kl4(w + 8), w3 = kl4(w + 12), c0 = kl4(cc), c1 = kl4(cc + 4);
    Geom G;
    G.pos = mk3(w0.x, w0.y, w0.z);
    G.mat.a[0] = w0.w; G.mat.a[1] = w1.x; G.mat.a[2] = w1.y; G.mat.a[3] = w1.z; G.mat.a[4] = w1.w; G.mat.a[5] = w2.x; G.mat.a[6] = w2.y; G.mat.a[7] = w2.z; G.mat.a[8] = w2.w;
    G.bc = mk3(w3.x, w3.y, w3.z);
    const unsigned pk = __float_as_uint(c0.w);
    G.type = pk & 15;
    G.size = mk3(c0.x, c0.y, c0.z);
    G.bh = mk3(c1.x, c1.y, c1.z);
    G.nvert = (pk >> 4) & 511;
    G.verts = (pk >> 31) ? ldsv + ((pk >> 13) & 0x3ffffu) : mesh_vert4 + (pk >> 13);          // (generic pointers: LDS or global)
    rbound = c1.w;
    return G;
}
// the oriented-box cull of pair_cull_box (collide.h) without data-dependent branches: both arms are evaluated and selected
// skin > 0: the test of "closer than skin" instead of "touching" (the item list of the persistent kernel is kept for several substeps,
// persist.h): spheres and the first box grow by skin - a box grown by skin along its axes contains every point within skin of it
__device__ __forceinline__ bool pair_cull_box_nb(const Geom &G1, const Geom &G2, float rb1, float rb2, float skin = 0.f) {
    const v3 n = col(G1.mat, 2);
    const bool plane_pass = dot(G2.bc - G1.pos, n) - (fabsf(dot(n, col(G2.mat, 0))) * G2.bh.x + fabsf(dot(n, col(G2.mat, 1))) * G2.bh.y + fabsf(dot(n, col(G2.mat, 2))) * G2.bh.z) <= skin;
    Geom G1s = G1;
    G1s.bh = mk3(G1.bh.x + skin, G1.bh.y + skin, G1.bh.z + skin);
    const bool s1 = sphere_hits_obb(G2.pos, rb2 + skin, G1), s2 = sphere_hits_obb(G1.pos, rb1 + skin, G2), ov = obb_overlap(G1s, G2);
    return G1.type == GEOM_PLANE ? plane_pass : (s1 & s2 & ov);
}
// sphere-cull record of a candidate pair, one dword: geom1 (6 bits) | geom2 (6 bits) | narrowphase function (2 bits; 0 and 1 mean
// geom1 is a plane) | - | upper half of the fp32 radius bound (sum of the bounding radii; geom2's alone against a plane), rounded
// up: a slightly larger bound only passes a few more candidates on to the oriented-box cull
__device__ __forceinline__ unsigned pair_pack(int g1, int g2, int fn, float rad) {
    return (unsigned)g1 | ((unsigned)g2 << 6) | ((unsigned)fn << 12) | (((__float_as_uint(rad) + 0xffffu) >> 16) << 16);
}
__device__ __forceinline__ int pk_g1(unsigned pk) { return pk & 63; }
__device__ __forceinline__ int pk_g2(unsigned pk) { return (pk >> 6) & 63; }
__device__ __forceinline__ int pk_fn(unsigned pk) { return (pk >> 12) & 3; }
