/*
 * hsr_oracle.c - CPU fp64 (HO_REAL = double; float on request) restatement of one MuJoCo substep (mj_step) for the HSR scene.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under hsr_env_amd/ (the product) may import, link or call
 * this file; it is used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * PARITY UNPINNED: the reference (ethanabrooks/hsr-env) delegates all arithmetic to
 * mujoco-py / MuJoCo (call site: hsr/env.py:123 `self.sim.step()`, hsr/mujoco_env.py:103),
 * an un-vendored, un-versioned third-party dependency that is absent from /root/reference and
 * from this container, and the reference ships no tests or golden vectors.  This file restates
 * the published MuJoCo 2.0 computation pipeline (Computation chapter: kinematics, CRB inertia,
 * RNE bias, position actuators, collision, soft-constraint model with elliptic cones, Newton
 * solver, semi-implicit Euler with implicit joint damping) for the model tables produced by
 * hsr_env_amd/compiler.py.  It is pinned only by analytic known-answer tests (tests/test_kat.py).
 *
 * Stage map:
 *   ho_kinematics      a-2.1  mj_kinematics / mj_comPos
 *   ho_inertia         a-2.2  mj_crb / mj_factorM (dense Cholesky; nv <= 32)
 *   ho_collision       a-2.3  mj_collision (plane-box, box-box, plane-convex, convex-convex MPR)
 *   ho_make_constraint a-2.4  mj_makeConstraint / mj_makeImpedance / mj_referenceConstraint
 *   ho_smooth          a-2.5  mj_fwdVelocity / mj_fwdActuation / mj_fwdAcceleration
 *   ho_solve           a-2.6  mj_fwdConstraint (Newton, exact line search)
 *   ho_euler           a-2.7  mj_Euler
 *   ho_env_step        a-1/a-3/a-4  HSREnv.step loop, goal test, early exit (hsr/env.py:115-135)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* the arithmetic type: double (the oracle); `make libhsr_oracle_f32.so` builds the same restatement with -DHO_REAL=float for the tests that ask
 * whether a difference between the fp32 kernels and this oracle is PRECISION (tests/test_gpu_hotpath.py: the pinch regime): test infrastructure only */
#ifndef HO_REAL
#define HO_REAL double
#endif
typedef HO_REAL real;

#define NV_MAX 32
#define NCON_MAX 128
#define NEFC_MAX 512
#define MINVAL 1e-15
#define MINIMP 0.0001
#define MAXIMP 0.9999

enum { SZ_NQ, SZ_NV, SZ_NU, SZ_NLINK, SZ_NBODY, SZ_NGEOM, SZ_NPAIR, SZ_NMESHVERT, SZ_NSLOT,
       SZ_NLIMIT, SZ_NCONMAX, SZ_NJMAX, SZ_NMOCAP };
enum { OPT_TIMESTEP, OPT_IMPRATIO, OPT_GRAV_Z, OPT_TOLERANCE, OPT_ITERATIONS, OPT_LS_ITERATIONS,
       OPT_LS_TOLERANCE, OPT_MPR_TOLERANCE, OPT_MPR_ITERATIONS, OPT_MEANINERTIA };
enum { DOF_SLIDE = 0, DOF_HINGE = 1, DOF_FREE_LIN = 2, DOF_FREE_ANG = 3 };
enum { GEOM_PLANE = 0, GEOM_SPHERE = 2, GEOM_CYLINDER = 5, GEOM_BOX = 6, GEOM_MESH = 7 };
enum { FN_PLANE_BOX = 0, FN_PLANE_CONVEX = 1, FN_BOX_BOX = 2, FN_CONVEX = 3 };

typedef struct {
    void *blob;
    const int *sizes; const real *opt;
    int nq, nv, nu, nlink, nbody, ngeom, npair, nmeshvert, nconmax, njmax;
    const real *qpos0;
    const int *link_parent; const real *link_pos, *link_quat;
    const int *link_dofadr, *link_dofnum, *link_qposadr, *link_free;
    const real *link_mass, *link_com, *link_inertia;
    const int *dof_link, *dof_type; const real *dof_axis, *dof_pos; const int *dof_parent;
    const real *dof_damping; const int *dof_qposadr; const real *dof_invweight0;
    const int *dof_limited; const real *dof_range, *dof_solref, *dof_solimp;
    const int *body_link; const real *body_pos, *body_quat; const int *body_mocap;
    const int *geom_type, *geom_link, *geom_body; const real *geom_pos, *geom_quat, *geom_size,
        *geom_rbound; const int *geom_condim, *geom_meshadr, *geom_meshnum;
    const real *geom_invweight, *mesh_vert;
    const int *pair_geom1, *pair_geom2, *pair_fn, *pair_condim, *pair_slot;
    const real *pair_friction, *pair_solref, *pair_solimp;
    const int *act_dof; const real *act_gear, *act_kp, *act_ctrlrange, *act_forcerange;
    real *geom_lmat;  /* [ngeom*9] geom rotation in link frame */
    void *conv[64]; int nconv;   /* converted tables of a float build */
} ho_model;

typedef struct {
    real pos[3], frame[9], dist;
    int geom1, geom2, link1, link2, dim, pair;
    real friction[5], solref[2], solimp[5], mu;
    int efc_address;
} ho_contact;

typedef struct {
    /* state */
    real *qpos, *qvel, *ctrl, *qacc_warmstart, mocap_pos[3], time;
    /* kinematics */
    real *xpos, *xquat, *xmat, *dof_ang, *dof_lin, *dof_anchor;
    real *link_w, *link_vo, *link_alpha, *link_ao;
    real *gpos, *gmat;
    /* dynamics */
    real *M, *L, *qfrc_bias, *qfrc_passive, *qfrc_actuator, *qfrc_smooth, *qacc_smooth;
    /* contacts / constraints */
    int ncon, nefc, nlimit_active;
    ho_contact *contact;
    real *efc_J, *efc_pos, *efc_D, *efc_R, *efc_aref, *efc_force, *efc_vel, *efc_B, *efc_K;
    int *efc_type; /* 0 limit, 1 contact-first-row, 2 contact-other-row */
    real *qacc, *qfrc_constraint;
    int solver_niter, bad;
    real solver_cost;
    /* introspection for tests/test_oracle_optimality.py: accepted cost after the warm-start choice ([0]) and after every Newton iteration, the largest
     * number of line-search evaluations one iteration needed, and whether a step that would have raised the cost was refused */
    real solver_trace[104];
    int solver_ntrace, solver_ls_max, solver_refused;
    int euler_rhs_macc;   /* test option (ho_set_euler_rhs): mj_Euler's damped solve takes M qacc as its right-hand side, as the HIP path does */
} ho_data;

/* ------------------------------------------------------------------ tiny vector helpers */
static inline real dot3(const real *a, const real *b) { return a[0]*b[0] + a[1]*b[1] + a[2]*b[2]; }
static inline void cross3(real *r, const real *a, const real *b) {
    real x = a[1]*b[2] - a[2]*b[1], y = a[2]*b[0] - a[0]*b[2], z = a[0]*b[1] - a[1]*b[0];
    r[0] = x; r[1] = y; r[2] = z;
}
static inline void sub3(real *r, const real *a, const real *b) { r[0]=a[0]-b[0]; r[1]=a[1]-b[1]; r[2]=a[2]-b[2]; }
static inline void add3(real *r, const real *a, const real *b) { r[0]=a[0]+b[0]; r[1]=a[1]+b[1]; r[2]=a[2]+b[2]; }
static inline void copy3(real *r, const real *a) { r[0]=a[0]; r[1]=a[1]; r[2]=a[2]; }
static inline void scl3(real *r, const real *a, real s) { r[0]=a[0]*s; r[1]=a[1]*s; r[2]=a[2]*s; }
static inline void addscl3(real *r, const real *a, real s) { r[0]+=a[0]*s; r[1]+=a[1]*s; r[2]+=a[2]*s; }
static inline real norm3(const real *a) { return sqrt(dot3(a, a)); }
static inline real normalize3(real *a) {
    real n = norm3(a);
    if (n < MINVAL) { a[0] = 1; a[1] = 0; a[2] = 0; return 0; }
    a[0] /= n; a[1] /= n; a[2] /= n; return n;
}
/* r = M v (row-major 3x3) */
static inline void mulmv3(real *r, const real *m, const real *v) {
    real x = m[0]*v[0]+m[1]*v[1]+m[2]*v[2], y = m[3]*v[0]+m[4]*v[1]+m[5]*v[2], z = m[6]*v[0]+m[7]*v[1]+m[8]*v[2];
    r[0]=x; r[1]=y; r[2]=z;
}
/* r = M^T v */
static inline void mulmtv3(real *r, const real *m, const real *v) {
    real x = m[0]*v[0]+m[3]*v[1]+m[6]*v[2], y = m[1]*v[0]+m[4]*v[1]+m[7]*v[2], z = m[2]*v[0]+m[5]*v[1]+m[8]*v[2];
    r[0]=x; r[1]=y; r[2]=z;
}
static inline void mulmm3(real *r, const real *a, const real *b) {
    real t[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
        t[3*i+j] = a[3*i]*b[j] + a[3*i+1]*b[3+j] + a[3*i+2]*b[6+j];
    memcpy(r, t, sizeof t);
}
static inline void quat2mat(real *m, const real *q) {
    real w=q[0], x=q[1], y=q[2], z=q[3];
    m[0]=1-2*(y*y+z*z); m[1]=2*(x*y-w*z); m[2]=2*(x*z+w*y);
    m[3]=2*(x*y+w*z); m[4]=1-2*(x*x+z*z); m[5]=2*(y*z-w*x);
    m[6]=2*(x*z-w*y); m[7]=2*(y*z+w*x); m[8]=1-2*(x*x+y*y);
}
static inline void quatmul(real *r, const real *a, const real *b) {
    real w = a[0]*b[0]-a[1]*b[1]-a[2]*b[2]-a[3]*b[3];
    real x = a[0]*b[1]+a[1]*b[0]+a[2]*b[3]-a[3]*b[2];
    real y = a[0]*b[2]-a[1]*b[3]+a[2]*b[0]+a[3]*b[1];
    real z = a[0]*b[3]+a[1]*b[2]-a[2]*b[1]+a[3]*b[0];
    r[0]=w; r[1]=x; r[2]=y; r[3]=z;
}
static inline void quatnorm(real *q) {
    real n = sqrt(q[0]*q[0]+q[1]*q[1]+q[2]*q[2]+q[3]*q[3]);
    if (n < MINVAL) { q[0]=1; q[1]=q[2]=q[3]=0; return; }
    q[0]/=n; q[1]/=n; q[2]/=n; q[3]/=n;
}

/* ------------------------------------------------------------------ blob loader */
typedef struct { char name[32]; uint32_t dtype, ndim, shape[4]; uint64_t off, nbytes; } blob_entry;

static const void *blob_find(const uint8_t *raw, const char *name, const uint8_t *data, uint32_t n);
/* an f64 table of the blob as reals: the blob's own memory for real = double, a converted copy (owned by the model: m->conv) otherwise */
static const void *blob_reals(ho_model *m, const uint8_t *raw, const char *name, const uint8_t *data, uint32_t n) {
    const blob_entry *e = (const blob_entry *)(raw + 16);
    for (uint32_t i = 0; i < n; i++)
        if (strncmp(e[i].name, name, 32) == 0) {
            const double *src = (const double *)(data + e[i].off);          /* (spelled out: this one IS the blob's f64) */
            if (sizeof(real) == sizeof(double)) return src;
            size_t cnt = (size_t)(e[i].nbytes / 8);
            real *dst = (real *)malloc(sizeof(real) * (cnt ? cnt : 1));
            for (size_t k = 0; k < cnt; k++) dst[k] = (real)src[k];
            if (m->nconv < 64) m->conv[m->nconv++] = dst;
            return dst;
        }
    return blob_find(raw, name, data, n);
}
static const void *blob_find(const uint8_t *raw, const char *name, const uint8_t *data, uint32_t n) {
    const blob_entry *e = (const blob_entry *)(raw + 16);
    for (uint32_t i = 0; i < n; i++)
        if (strncmp(e[i].name, name, 32) == 0) return data + e[i].off;
    fprintf(stderr, "hsr_oracle: blob entry '%s' missing\n", name);
    return NULL;
}

ho_model *ho_model_load(const void *blob, size_t len) {
    if (len < 16 || memcmp(blob, "HSRM0001", 8) != 0) return NULL;
    ho_model *m = (ho_model *)calloc(1, sizeof *m);
    m->blob = malloc(len);
    memcpy(m->blob, blob, len);
    const uint8_t *raw = (const uint8_t *)m->blob;
    uint32_t n = *(const uint32_t *)(raw + 8);
    const uint8_t *p = raw + 16 + (size_t)n * sizeof(blob_entry);
    uint64_t jl = *(const uint64_t *)p;
    const uint8_t *data = p + 8 + jl;
    /* int tables point into the blob; real tables too when real is the blob's f64, else they are converted once (FD) */
#define F(field) m->field = blob_find(raw, #field, data, n)
#define FD(field) m->field = (const real *)blob_reals(m, raw, #field, data, n)
    F(sizes); FD(opt); FD(qpos0);
    F(link_parent); FD(link_pos); FD(link_quat); F(link_dofadr); F(link_dofnum); F(link_qposadr);
    F(link_free); FD(link_mass); FD(link_com); FD(link_inertia);
    F(dof_link); F(dof_type); FD(dof_axis); FD(dof_pos); F(dof_parent); FD(dof_damping); F(dof_qposadr);
    FD(dof_invweight0); F(dof_limited); FD(dof_range); FD(dof_solref); FD(dof_solimp);
    F(body_link); FD(body_pos); FD(body_quat); F(body_mocap);
    F(geom_type); F(geom_link); F(geom_body); FD(geom_pos); FD(geom_quat); FD(geom_size); FD(geom_rbound);
    F(geom_condim); F(geom_meshadr); F(geom_meshnum); FD(geom_invweight); FD(mesh_vert);
    F(pair_geom1); F(pair_geom2); F(pair_fn); F(pair_condim); F(pair_slot); FD(pair_friction);
    FD(pair_solref); FD(pair_solimp);
    F(act_dof); FD(act_gear); FD(act_kp); FD(act_ctrlrange); FD(act_forcerange);
#undef FD
#undef F
    m->nq = m->sizes[SZ_NQ]; m->nv = m->sizes[SZ_NV]; m->nu = m->sizes[SZ_NU];
    m->nlink = m->sizes[SZ_NLINK]; m->nbody = m->sizes[SZ_NBODY]; m->ngeom = m->sizes[SZ_NGEOM];
    m->npair = m->sizes[SZ_NPAIR]; m->nmeshvert = m->sizes[SZ_NMESHVERT];
    m->nconmax = m->sizes[SZ_NCONMAX] < NCON_MAX ? m->sizes[SZ_NCONMAX] : NCON_MAX;
    m->njmax = m->sizes[SZ_NJMAX] < NEFC_MAX ? m->sizes[SZ_NJMAX] : NEFC_MAX;
    if (m->nv > NV_MAX) { free(m->blob); free(m); return NULL; }
    m->geom_lmat = (real *)malloc(sizeof(real) * 9 * (size_t)m->ngeom);
    for (int g = 0; g < m->ngeom; g++) quat2mat(m->geom_lmat + 9*g, m->geom_quat + 4*g);
    return m;
}
void ho_model_free(ho_model *m) { if (m) { for (int i = 0; i < m->nconv; i++) free(m->conv[i]); free(m->geom_lmat); free(m->blob); free(m); } }
int ho_real_bytes(void) { return (int)sizeof(real); }
int ho_model_size(const ho_model *m, int which) { return m->sizes[which]; }

#define ALLOC(field, n) d->field = (real *)calloc((size_t)(n) > 0 ? (size_t)(n) : 1, sizeof(real))
ho_data *ho_data_new(const ho_model *m) {
    ho_data *d = (ho_data *)calloc(1, sizeof *d);
    int nv = m->nv, nl = m->nlink;
    ALLOC(qpos, m->nq); ALLOC(qvel, nv); ALLOC(ctrl, m->nu); ALLOC(qacc_warmstart, nv);
    ALLOC(xpos, 3*nl); ALLOC(xquat, 4*nl); ALLOC(xmat, 9*nl);
    ALLOC(dof_ang, 3*nv); ALLOC(dof_lin, 3*nv); ALLOC(dof_anchor, 3*nv);
    ALLOC(link_w, 3*nl); ALLOC(link_vo, 3*nl); ALLOC(link_alpha, 3*nl); ALLOC(link_ao, 3*nl);
    ALLOC(gpos, 3*m->ngeom); ALLOC(gmat, 9*m->ngeom);
    ALLOC(M, nv*nv); ALLOC(L, nv*nv); ALLOC(qfrc_bias, nv); ALLOC(qfrc_passive, nv);
    ALLOC(qfrc_actuator, nv); ALLOC(qfrc_smooth, nv); ALLOC(qacc_smooth, nv);
    d->contact = (ho_contact *)calloc(NCON_MAX, sizeof(ho_contact));
    ALLOC(efc_J, NEFC_MAX*nv); ALLOC(efc_pos, NEFC_MAX); ALLOC(efc_D, NEFC_MAX); ALLOC(efc_R, NEFC_MAX);
    ALLOC(efc_aref, NEFC_MAX); ALLOC(efc_force, NEFC_MAX); ALLOC(efc_vel, NEFC_MAX);
    ALLOC(efc_B, NEFC_MAX); ALLOC(efc_K, NEFC_MAX);
    d->efc_type = (int *)calloc(NEFC_MAX, sizeof(int));
    ALLOC(qacc, nv); ALLOC(qfrc_constraint, nv);
    return d;
}
void ho_data_free(ho_data *d) {
    if (!d) return;
    free(d->qpos); free(d->qvel); free(d->ctrl); free(d->qacc_warmstart);
    free(d->xpos); free(d->xquat); free(d->xmat); free(d->dof_ang); free(d->dof_lin); free(d->dof_anchor);
    free(d->link_w); free(d->link_vo); free(d->link_alpha); free(d->link_ao); free(d->gpos); free(d->gmat);
    free(d->M); free(d->L); free(d->qfrc_bias); free(d->qfrc_passive); free(d->qfrc_actuator);
    free(d->qfrc_smooth); free(d->qacc_smooth); free(d->contact);
    free(d->efc_J); free(d->efc_pos); free(d->efc_D); free(d->efc_R); free(d->efc_aref); free(d->efc_force);
    free(d->efc_vel); free(d->efc_B); free(d->efc_K); free(d->efc_type); free(d->qacc); free(d->qfrc_constraint);
    free(d);
}

/* mj_resetData: qpos<-qpos0, qvel<-0, ctrl<-0, mocap_pos<-body pos (0 0 0), warmstart<-0, time<-0 */
void ho_reset(const ho_model *m, ho_data *d) {
    memcpy(d->qpos, m->qpos0, sizeof(real) * (size_t)m->nq);
    memset(d->qvel, 0, sizeof(real) * (size_t)m->nv);
    memset(d->ctrl, 0, sizeof(real) * (size_t)m->nu);
    memset(d->qacc_warmstart, 0, sizeof(real) * (size_t)m->nv);
    d->mocap_pos[0] = d->mocap_pos[1] = d->mocap_pos[2] = 0;
    d->time = 0; d->bad = 0; d->ncon = 0; d->nefc = 0;
}

/* ------------------------------------------------------------------ a-2.1 kinematics */
static void ho_kinematics(const ho_model *m, ho_data *d) {
    real *xpos = d->xpos, *xquat = d->xquat, *xmat = d->xmat;
    xpos[0]=xpos[1]=xpos[2]=0; xquat[0]=1; xquat[1]=xquat[2]=xquat[3]=0; quat2mat(xmat, xquat);
    for (int l = 1; l < m->nlink; l++) {
        real *pos = xpos + 3*l, *quat = xquat + 4*l, *mat = xmat + 9*l;
        if (m->link_free[l]) {
            int a = m->link_qposadr[l];
            quatnorm(d->qpos + a + 3);             /* mj_kinematics normalises free-joint quats in place */
            copy3(pos, d->qpos + a);
            memcpy(quat, d->qpos + a + 3, 4 * sizeof(real));
            quat2mat(mat, quat);
        } else {
            int p = m->link_parent[l];
            real t[3];
            mulmv3(t, xmat + 9*p, m->link_pos + 3*l);
            add3(pos, xpos + 3*p, t);
            quatmul(quat, xquat + 4*p, m->link_quat + 4*l);
            quat2mat(mat, quat);
            for (int k = m->link_dofadr[l]; k < m->link_dofadr[l] + m->link_dofnum[l]; k++) {
                real q = d->qpos[m->dof_qposadr[k]];
                if (m->dof_type[k] == DOF_SLIDE) {
                    mulmv3(t, mat, m->dof_axis + 3*k);
                    addscl3(pos, t, q);
                } else {
                    real anchor[3], qr[4], s = sin(0.5*q);
                    mulmv3(t, mat, m->dof_pos + 3*k); add3(anchor, pos, t);
                    qr[0] = cos(0.5*q); qr[1] = m->dof_axis[3*k]*s; qr[2] = m->dof_axis[3*k+1]*s; qr[3] = m->dof_axis[3*k+2]*s;
                    quatmul(quat, quat, qr);
                    quat2mat(mat, quat);
                    mulmv3(t, mat, m->dof_pos + 3*k); sub3(pos, anchor, t);
                }
            }
            quatnorm(quat); quat2mat(mat, quat);
        }
    }
    /* world-frame motion axes per dof */
    for (int l = 1; l < m->nlink; l++) {
        const real *mat = xmat + 9*l;
        int d0 = m->link_dofadr[l];
        if (m->link_free[l]) {
            for (int k = 0; k < 3; k++) {
                real *lin = d->dof_lin + 3*(d0+k), *ang = d->dof_ang + 3*(d0+k);
                lin[0]=lin[1]=lin[2]=0; lin[k]=1; ang[0]=ang[1]=ang[2]=0;
                copy3(d->dof_anchor + 3*(d0+k), xpos + 3*l);
                ang = d->dof_ang + 3*(d0+3+k); lin = d->dof_lin + 3*(d0+3+k);
                ang[0]=mat[k]; ang[1]=mat[3+k]; ang[2]=mat[6+k]; lin[0]=lin[1]=lin[2]=0;
                copy3(d->dof_anchor + 3*(d0+3+k), xpos + 3*l);
            }
        } else {
            for (int k = d0; k < d0 + m->link_dofnum[l]; k++) {
                real ax[3], t[3];
                mulmv3(ax, mat, m->dof_axis + 3*k);
                if (m->dof_type[k] == DOF_SLIDE) {
                    copy3(d->dof_lin + 3*k, ax); d->dof_ang[3*k]=d->dof_ang[3*k+1]=d->dof_ang[3*k+2]=0;
                    copy3(d->dof_anchor + 3*k, xpos + 3*l);
                } else {
                    copy3(d->dof_ang + 3*k, ax); d->dof_lin[3*k]=d->dof_lin[3*k+1]=d->dof_lin[3*k+2]=0;
                    mulmv3(t, mat, m->dof_pos + 3*k); add3(d->dof_anchor + 3*k, xpos + 3*l, t);
                }
            }
        }
    }
    /* geom world frames */
    for (int g = 0; g < m->ngeom; g++) {
        int l = m->geom_link[g];
        real t[3];
        mulmv3(t, xmat + 9*l, m->geom_pos + 3*g);
        add3(d->gpos + 3*g, xpos + 3*l, t);
        mulmm3(d->gmat + 9*g, xmat + 9*l, m->geom_lmat + 9*g);
    }
}

/* column k of the point Jacobian: velocity of world point p (attached below dof k) per unit qvel[k] */
static inline void dof_point_vel(const ho_data *d, int k, const real *p, real *v) {
    real r[3], c[3];
    sub3(r, p, d->dof_anchor + 3*k);
    cross3(c, d->dof_ang + 3*k, r);
    add3(v, d->dof_lin + 3*k, c);
}

/* ------------------------------------------------------------------ a-2.2 inertia */
static int cholesky(real *L, const real *A, int n) {
    memcpy(L, A, sizeof(real) * (size_t)n * (size_t)n);
    for (int j = 0; j < n; j++) {
        real s = L[j*n+j];
        for (int k = 0; k < j; k++) s -= L[j*n+k]*L[j*n+k];
        if (s < MINVAL) return -1;
        s = sqrt(s); L[j*n+j] = s;
        for (int i = j+1; i < n; i++) {
            real t = L[i*n+j];
            for (int k = 0; k < j; k++) t -= L[i*n+k]*L[j*n+k];
            L[i*n+j] = t / s;
        }
    }
    return 0;
}
static void chol_solve(const real *L, int n, real *x) {
    for (int i = 0; i < n; i++) { real s = x[i]; for (int k = 0; k < i; k++) s -= L[i*n+k]*x[k]; x[i] = s / L[i*n+i]; }
    for (int i = n-1; i >= 0; i--) { real s = x[i]; for (int k = i+1; k < n; k++) s -= L[k*n+i]*x[k]; x[i] = s / L[i*n+i]; }
}

static void link_world_inertia(const ho_model *m, const ho_data *d, int l, real *com, real *I) {
    const real *mat = d->xmat + 9*l, *li = m->link_inertia + 6*l;
    real t[3], Il[9] = { li[0], li[3], li[4], li[3], li[1], li[5], li[4], li[5], li[2] }, tmp[9], matT[9];
    mulmv3(t, mat, m->link_com + 3*l); add3(com, d->xpos + 3*l, t);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) matT[3*i+j] = mat[3*j+i];
    mulmm3(tmp, mat, Il); mulmm3(I, tmp, matT);
}

static void ho_inertia(const ho_model *m, ho_data *d) {
    int nv = m->nv;
    memset(d->M, 0, sizeof(real) * (size_t)nv * (size_t)nv);
    for (int l = 1; l < m->nlink; l++) {
        real com[3], I[9], jp[NV_MAX][3], jr[NV_MAX][3], Ijr[3];
        int chain[NV_MAX], nc = 0;
        link_world_inertia(m, d, l, com, I);
        for (int k = m->link_dofadr[l] + m->link_dofnum[l] - 1; k >= 0; k = m->dof_parent[k]) chain[nc++] = k;
        for (int a = 0; a < nc; a++) { dof_point_vel(d, chain[a], com, jp[a]); copy3(jr[a], d->dof_ang + 3*chain[a]); }
        for (int a = 0; a < nc; a++) {
            mulmv3(Ijr, I, jr[a]);
            for (int b = 0; b < nc; b++)
                d->M[chain[a]*nv + chain[b]] += m->link_mass[l] * dot3(jp[a], jp[b]) + dot3(Ijr, jr[b]);
        }
    }
    if (cholesky(d->L, d->M, nv) != 0) d->bad = 1;
}

/* ------------------------------------------------------------------ a-2.5 smooth dynamics */
static void ho_smooth(const ho_model *m, ho_data *d) {
    int nv = m->nv;
    const real grav[3] = { 0, 0, m->opt[OPT_GRAV_Z] };
    /* link velocities and bias accelerations (qacc = 0) */
    memset(d->link_w, 0, sizeof(real)*3*(size_t)m->nlink); memset(d->link_vo, 0, sizeof(real)*3*(size_t)m->nlink);
    memset(d->link_alpha, 0, sizeof(real)*3*(size_t)m->nlink); memset(d->link_ao, 0, sizeof(real)*3*(size_t)m->nlink);
    for (int l = 1; l < m->nlink; l++) {
        real *w = d->link_w + 3*l, *vo = d->link_vo + 3*l, *al = d->link_alpha + 3*l, *ao = d->link_ao + 3*l;
        int d0 = m->link_dofadr[l];
        if (m->link_free[l]) {
            copy3(vo, d->qvel + d0);
            mulmv3(w, d->xmat + 9*l, d->qvel + d0 + 3);     /* local -> world angular velocity */
            continue;                                        /* bias acceleration of the frame is zero */
        }
        int p = m->link_parent[l];
        const real *wp = d->link_w + 3*p;
        real r[3], t[3], t2[3];
        /* the parent point that coincides with this link's origin *before* this link's joints act is
           not needed explicitly: use r = xpos_l - xpos_p (current), add joint-relative terms */
        sub3(r, d->xpos + 3*l, d->xpos + 3*p);
        copy3(w, wp); copy3(al, d->link_alpha + 3*p);
        cross3(t, wp, r); add3(vo, d->link_vo + 3*p, t);
        cross3(t, d->link_alpha + 3*p, r); add3(ao, d->link_ao + 3*p, t);
        cross3(t, wp, r); cross3(t2, wp, t); add3(ao, ao, t2);
        for (int k = d0; k < d0 + m->link_dofnum[l]; k++) {
            real qd = d->qvel[k];
            if (m->dof_type[k] == DOF_SLIDE) {
                const real *s = d->dof_lin + 3*k;
                addscl3(vo, s, qd);
                cross3(t, wp, s); addscl3(ao, t, 2*qd);      /* Coriolis 2 w x (s qd) */
            } else {
                const real *a = d->dof_ang + 3*k;
                real rho[3];                               /* origin relative to anchor */
                sub3(rho, d->xpos + 3*l, d->dof_anchor + 3*k);
                /* acceleration of origin: a_c + alpha x rho + w x (w x rho), with a_c the parent-point
                   acceleration at the anchor; rewrite relative to what is already in ao (computed at r) */
                real wl[3], all[3], rc[3];
                copy3(wl, w); addscl3(wl, a, qd);
                cross3(t, w, a); copy3(all, al); addscl3(all, t, qd);   /* alpha += (w x a) qd */
                /* parent-point acceleration at anchor c = origin - rho */
                sub3(rc, r, rho);
                real ac[3];
                cross3(t, d->link_alpha + 3*p, rc); add3(ac, d->link_ao + 3*p, t);
                cross3(t, wp, rc); cross3(t2, wp, t); add3(ac, ac, t2);
                real vc[3];
                cross3(t, wp, rc); add3(vc, d->link_vo + 3*p, t);
                cross3(t, all, rho); add3(ao, ac, t);
                cross3(t, wl, rho); cross3(t2, wl, t); add3(ao, ao, t2);
                cross3(t, wl, rho); add3(vo, vc, t);
                copy3(w, wl); copy3(al, all);
            }
        }
    }
    /* bias = sum over links of J^T [m (a_com - g); I alpha + w x I w] */
    memset(d->qfrc_bias, 0, sizeof(real) * (size_t)nv);
    for (int l = 1; l < m->nlink; l++) {
        real com[3], I[9], rc[3], t[3], t2[3], acom[3], F[3], N[3], Iw[3];
        const real *w = d->link_w + 3*l, *al = d->link_alpha + 3*l;
        link_world_inertia(m, d, l, com, I);
        sub3(rc, com, d->xpos + 3*l);
        cross3(t, al, rc); add3(acom, d->link_ao + 3*l, t);
        cross3(t, w, rc); cross3(t2, w, t); add3(acom, acom, t2);
        sub3(t, acom, grav); scl3(F, t, m->link_mass[l]);
        mulmv3(N, I, al); mulmv3(Iw, I, w); cross3(t, w, Iw); add3(N, N, t);
        for (int k = m->link_dofadr[l] + m->link_dofnum[l] - 1; k >= 0; k = m->dof_parent[k]) {
            real jp[3];
            dof_point_vel(d, k, com, jp);
            d->qfrc_bias[k] += dot3(jp, F) + dot3(d->dof_ang + 3*k, N);
        }
    }
    for (int k = 0; k < nv; k++) { d->qfrc_passive[k] = -m->dof_damping[k] * d->qvel[k]; d->qfrc_actuator[k] = 0; }
    /* position actuators: force = kp*clamp(ctrl) - kp*gear*q, clamped to forcerange; qfrc = gear*force */
    for (int a = 0; a < m->nu; a++) {
        int k = m->act_dof[a];
        real c = d->ctrl[a], lo = m->act_ctrlrange[2*a], hi = m->act_ctrlrange[2*a+1];
        c = c < lo ? lo : (c > hi ? hi : c);
        real f = m->act_kp[a] * c - m->act_kp[a] * m->act_gear[a] * d->qpos[m->dof_qposadr[k]];
        lo = m->act_forcerange[2*a]; hi = m->act_forcerange[2*a+1];
        f = f < lo ? lo : (f > hi ? hi : f);
        d->qfrc_actuator[k] += m->act_gear[a] * f;
    }
    for (int k = 0; k < nv; k++) {
        d->qfrc_smooth[k] = d->qfrc_passive[k] - d->qfrc_bias[k] + d->qfrc_actuator[k];
        d->qacc_smooth[k] = d->qfrc_smooth[k];
    }
    if (!d->bad) chol_solve(d->L, nv, d->qacc_smooth);
}

/* ------------------------------------------------------------------ a-2.3 collision */
/* mju_makeFrame: complete frame[0:3] (normal) with two tangents */
static void make_frame(real *f) {
    real *x = f, *y = f + 3, *z = f + 6;
    if (x[1] > -0.5 && x[1] < 0.5) { y[0]=0; y[1]=1; y[2]=0; } else { y[0]=0; y[1]=0; y[2]=1; }
    real dd = dot3(x, y);
    addscl3(y, x, -dd); normalize3(y);
    cross3(z, x, y);
}

static int add_contact(const ho_model *m, ho_data *d, int pair, const real *pos, const real *n, real dist) {
    if (d->ncon >= m->nconmax) return 0;
    ho_contact *c = d->contact + d->ncon++;
    copy3(c->pos, pos); copy3(c->frame, n); make_frame(c->frame); c->dist = dist;
    c->pair = pair; c->geom1 = m->pair_geom1[pair]; c->geom2 = m->pair_geom2[pair];
    c->link1 = m->geom_link[c->geom1]; c->link2 = m->geom_link[c->geom2];
    c->dim = m->pair_condim[pair];
    memcpy(c->friction, m->pair_friction + 5*pair, 5*sizeof(real));
    memcpy(c->solref, m->pair_solref + 2*pair, 2*sizeof(real));
    memcpy(c->solimp, m->pair_solimp + 5*pair, 5*sizeof(real));
    return 1;
}

/* mjc_PlaneBox: corners below the plane, at most 4 */
static void collide_plane_box(const ho_model *m, ho_data *d, int pair, int g1, int g2) {
    const real *pp = d->gpos + 3*g1, *pm = d->gmat + 9*g1, *bp = d->gpos + 3*g2, *bm = d->gmat + 9*g2;
    const real *s = m->geom_size + 3*g2;
    real n[3] = { pm[2], pm[5], pm[8] };
    int cnt = 0;
    for (int i = 0; i < 8 && cnt < 4; i++) {
        real loc[3] = { (i & 1 ? s[0] : -s[0]), (i & 2 ? s[1] : -s[1]), (i & 4 ? s[2] : -s[2]) }, c[3], r[3];
        mulmv3(c, bm, loc); add3(c, c, bp);
        sub3(r, c, pp);
        real dist = dot3(r, n);
        if (dist < 0) {
            real pos[3]; copy3(pos, c); addscl3(pos, n, -0.5*dist);
            cnt += add_contact(m, d, pair, pos, n, dist);
        }
    }
}

/* support point of a convex geom (box / cylinder / mesh hull) in world direction dir */
static void support(const ho_model *m, const ho_data *d, int g, const real *dir, real *out) {
    const real *mat = d->gmat + 9*g, *pos = d->gpos + 3*g, *s = m->geom_size + 3*g;
    real dl[3], loc[3];
    mulmtv3(dl, mat, dir);
    switch (m->geom_type[g]) {
    case GEOM_BOX:
        loc[0] = dl[0] > 0 ? s[0] : -s[0]; loc[1] = dl[1] > 0 ? s[1] : -s[1]; loc[2] = dl[2] > 0 ? s[2] : -s[2];
        break;
    case GEOM_CYLINDER: {
        real rr = sqrt(dl[0]*dl[0] + dl[1]*dl[1]);
        if (rr > MINVAL) { loc[0] = dl[0]/rr*s[0]; loc[1] = dl[1]/rr*s[0]; } else { loc[0] = loc[1] = 0; }
        loc[2] = dl[2] > 0 ? s[1] : -s[1];
        break; }
    case GEOM_SPHERE:
        { real t[3]; copy3(t, dl); normalize3(t); scl3(loc, t, s[0]); }
        break;
    default: { /* mesh: exhaustive search over hull vertices, first maximum wins */
        const real *v = m->mesh_vert + 3*m->geom_meshadr[g];
        int best = 0; real bd = -1e300;
        for (int i = 0; i < m->geom_meshnum[g]; i++) {
            real t = dot3(v + 3*i, dl);
            if (t > bd) { bd = t; best = i; }
        }
        copy3(loc, v + 3*best);
        break; }
    }
    mulmv3(out, mat, loc); add3(out, out, pos);
}

/* mjc_PlaneConvex.  The deepest support point (direction -n) always; when the pair has room for more than one contact (pair_slot: the
 * compiler's plane_convex_points = 4) up to three more, as MuJoCo adds them: support points along -n tilted towards three tangent
 * directions 120 degrees apart, kept when they lie below the plane and are not one of the points already kept.  MuJoCo's source is not at
 * hand: the scheme is restated from its behaviour, the tilt (0.1) and the duplicate distance (1e-5 m) are this restatement's own constants
 * (deviations).  A hull resting on a flat face gets the extreme vertices of that face; a hull tilted by more than the tilt
 * angle, or a rounded one, keeps the single point. */
#define PLANE_CONVEX_TILT 0.1
#define PLANE_CONVEX_DUP 1e-5
static void collide_plane_convex(const ho_model *m, ho_data *d, int pair, int g1, int g2) {
    const real *pp = d->gpos + 3*g1, *pm = d->gmat + 9*g1;
    real n[3] = { pm[2], pm[5], pm[8] }, nn[3] = { -pm[2], -pm[5], -pm[8] }, p[3], r[3];
    support(m, d, g2, nn, p);
    sub3(r, p, pp);
    real dist = dot3(r, n);
    if (!(dist < 0)) return;
    { real pos[3]; copy3(pos, p); addscl3(pos, n, -0.5*dist); add_contact(m, d, pair, pos, n, dist); }
    const int maxcnt = m->pair_slot[pair+1] - m->pair_slot[pair];
    if (maxcnt <= 1) return;
    real fr[9], kept[4][3]; int nk = 1;
    copy3(fr, n); make_frame(fr);
    copy3(kept[0], p);
    static const real cs[3][2] = { {1.0, 0.0}, {-0.5, 0.8660254037844386}, {-0.5, -0.8660254037844386} };
    for (int i = 0; i < 3 && nk < maxcnt; i++) {
        real dir[3], q[3];
        for (int k = 0; k < 3; k++) dir[k] = nn[k] + PLANE_CONVEX_TILT * (cs[i][0]*fr[3+k] + cs[i][1]*fr[6+k]);
        support(m, d, g2, dir, q);
        sub3(r, q, pp);
        const real dq = dot3(r, n);
        if (!(dq < 0)) continue;
        int dup = 0;
        for (int j = 0; j < nk; j++) { real e[3]; sub3(e, q, kept[j]); if (dot3(e, e) < PLANE_CONVEX_DUP*PLANE_CONVEX_DUP) dup = 1; }
        if (dup) continue;
        real pos[3]; copy3(pos, q); addscl3(pos, n, -0.5*dq);
        add_contact(m, d, pair, pos, n, dq);
        copy3(kept[nk++], q);
    }
}

/* --- box-box: separating-axis test + reference-face clipping (up to 8 points) */
static int clip_poly(real (*poly)[3], int n, const real *axis, real lim, const real *origin) {
    /* keep the part with (p-origin).axis <= lim ; Sutherland-Hodgman */
    real out[16][3]; int no = 0;
    for (int i = 0; i < n; i++) {
        const real *a = poly[i], *b = poly[(i+1) % n];
        real ra[3], rb[3]; sub3(ra, a, origin); sub3(rb, b, origin);
        real da = dot3(ra, axis) - lim, db = dot3(rb, axis) - lim;
        if (da <= 0) { copy3(out[no++], a); }
        if ((da < 0 && db > 0) || (da > 0 && db < 0)) {
            real t = da / (da - db);
            out[no][0] = a[0] + t*(b[0]-a[0]); out[no][1] = a[1] + t*(b[1]-a[1]); out[no][2] = a[2] + t*(b[2]-a[2]); no++;
        }
    }
    for (int i = 0; i < no; i++) copy3(poly[i], out[i]);
    return no;
}

static void collide_box_box(const ho_model *m, ho_data *d, int pair, int g1, int g2) {
    const real *p1 = d->gpos + 3*g1, *R1 = d->gmat + 9*g1, *s1 = m->geom_size + 3*g1;
    const real *p2 = d->gpos + 3*g2, *R2 = d->gmat + 9*g2, *s2 = m->geom_size + 3*g2;
    real A[3][3], B[3][3], C[3][3], aC[3][3], dv[3];
    for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) { A[i][k] = R1[3*k+i]; B[i][k] = R2[3*k+i]; }
    sub3(dv, p2, p1);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { C[i][j] = dot3(A[i], B[j]); aC[i][j] = fabs(C[i][j]); }
    real best = -1e300; int code = -1; real bestn[3] = {0,0,0};
    /* face axes of box 1, then box 2: first strictly-better axis wins (box 1 preferred on ties) */
    for (int i = 0; i < 3; i++) {
        real t = dot3(dv, A[i]);
        real sep = fabs(t) - (s1[i] + s2[0]*aC[i][0] + s2[1]*aC[i][1] + s2[2]*aC[i][2]);
        if (sep > 0) return;
        if (sep > best) { best = sep; code = i; scl3(bestn, A[i], t < 0 ? -1.0 : 1.0); }
    }
    for (int j = 0; j < 3; j++) {
        real t = dot3(dv, B[j]);
        real sep = fabs(t) - (s2[j] + s1[0]*aC[0][j] + s1[1]*aC[1][j] + s1[2]*aC[2][j]);
        if (sep > 0) return;
        if (sep > best) { best = sep; code = 3 + j; scl3(bestn, B[j], t < 0 ? -1.0 : 1.0); }
    }
    /* edge x edge axes; must beat the best face axis by 5 % + 1e-9 to be chosen */
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        real L[3]; cross3(L, A[i], B[j]);
        real ln = norm3(L);
        if (ln < 1e-6) continue;
        scl3(L, L, 1.0/ln);
        real t = dot3(dv, L), ra = 0, rb = 0;
        for (int k = 0; k < 3; k++) { ra += s1[k]*fabs(dot3(A[k], L)); rb += s2[k]*fabs(dot3(B[k], L)); }
        real sep = fabs(t) - (ra + rb);
        if (sep > 0) return;
        if (sep * 1.05 > best + 1e-9) { best = sep / 1.0; code = 6 + 3*i + j; scl3(bestn, L, t < 0 ? -1.0 : 1.0); }
    }
    if (code < 6) {
        /* reference face on box `ref`, incident face on the other */
        int ref1 = code < 3;
        const real *pr = ref1 ? p1 : p2, *pi = ref1 ? p2 : p1, *sr = ref1 ? s1 : s2, *si = ref1 ? s2 : s1;
        real (*Ar)[3] = ref1 ? A : B, (*Ai)[3] = ref1 ? B : A;
        int ax = ref1 ? code : code - 3;
        real nref[3]; scl3(nref, bestn, ref1 ? 1.0 : -1.0);   /* outward normal of the reference face */
        /* incident face: most anti-parallel to nref */
        int jx = 0; real bd = -1;
        for (int j = 0; j < 3; j++) { real t = fabs(dot3(nref, Ai[j])); if (t > bd) { bd = t; jx = j; } }
        real sgn = dot3(nref, Ai[jx]) > 0 ? -1.0 : 1.0;
        int j1 = (jx + 1) % 3, j2 = (jx + 2) % 3;
        real fc[3]; copy3(fc, pi); addscl3(fc, Ai[jx], sgn * si[jx]);
        real poly[16][3];
        const real sg[4][2] = { {1,1}, {-1,1}, {-1,-1}, {1,-1} };
        for (int k = 0; k < 4; k++) { copy3(poly[k], fc); addscl3(poly[k], Ai[j1], sg[k][0]*si[j1]); addscl3(poly[k], Ai[j2], sg[k][1]*si[j2]); }
        int np = 4, u = (ax + 1) % 3, v = (ax + 2) % 3;
        real neg[3];
        np = clip_poly(poly, np, Ar[u], sr[u], pr); if (np) { scl3(neg, Ar[u], -1); np = clip_poly(poly, np, neg, sr[u], pr); }
        if (np) np = clip_poly(poly, np, Ar[v], sr[v], pr);
        if (np) { scl3(neg, Ar[v], -1); np = clip_poly(poly, np, neg, sr[v], pr); }
        real n12[3]; copy3(n12, bestn);                      /* from geom1 to geom2 */
        int cnt = 0;
        for (int k = 0; k < np && cnt < 8; k++) {
            real r[3]; sub3(r, poly[k], pr);
            real dist = dot3(r, nref) - sr[ax];               /* signed distance to the reference face */
            if (dist < 0) {
                real pos[3]; copy3(pos, poly[k]); addscl3(pos, nref, -0.5*dist);
                cnt += add_contact(m, d, pair, pos, n12, dist);
            }
        }
    } else {
        int i = (code - 6) / 3, j = (code - 6) % 3;
        real pa[3], pb[3];
        copy3(pa, p1); copy3(pb, p2);
        for (int k = 0; k < 3; k++) {
            if (k != i) addscl3(pa, A[k], (dot3(bestn, A[k]) > 0 ? 1.0 : -1.0) * s1[k]);
            if (k != j) addscl3(pb, B[k], (dot3(bestn, B[k]) > 0 ? -1.0 : 1.0) * s2[k]);
        }
        /* closest points of lines pa + t A_i and pb + u B_j */
        real w[3]; sub3(w, pa, pb);
        real b = C[i][j], dd = dot3(A[i], w), e = dot3(B[j], w), den = 1 - b*b;
        real t = den > 1e-12 ? (b*e - dd) / den : 0, uu = den > 1e-12 ? (e - b*dd) / den : 0;
        real qa[3], qb[3], pos[3];
        copy3(qa, pa); addscl3(qa, A[i], t); copy3(qb, pb); addscl3(qb, B[j], uu);
        add3(pos, qa, qb); scl3(pos, pos, 0.5);
        add_contact(m, d, pair, pos, bestn, best);
    }
}

/* --- convex-convex: Minkowski Portal Refinement (restates libccd's ccdMPRPenetration, used by
 *     MuJoCo's mjc_Convex); one contact per pair */
typedef struct { real v[3], v1[3], v2[3]; } mpr_sup;
/* diagnostics: support-pair evaluations of the last / all MPR calls (not thread safe; tests only) */
static long ho_dbg_mpr_supports_ = 0, ho_dbg_mpr_calls_ = 0, ho_dbg_mpr_max_ = 0, ho_dbg_mpr_cur_ = 0;
long ho_dbg_mpr_supports(void) { return ho_dbg_mpr_supports_; }
long ho_dbg_mpr_calls(void) { return ho_dbg_mpr_calls_; }
long ho_dbg_mpr_max(void) { return ho_dbg_mpr_max_; }
static void mpr_support(const ho_model *m, const ho_data *d, int g1, int g2, const real *dir, mpr_sup *s) {
    ho_dbg_mpr_supports_++; ho_dbg_mpr_cur_++;
    real nd[3] = { -dir[0], -dir[1], -dir[2] };
    support(m, d, g1, dir, s->v1); support(m, d, g2, nd, s->v2); sub3(s->v, s->v1, s->v2);
}
static real point_seg_dist2(const real *P, const real *x0, const real *b, real *wit) {
    real dd[3], a[3]; sub3(dd, b, x0); sub3(a, x0, P);
    real t = -dot3(a, dd) / dot3(dd, dd), w[3];
    if (t < 0 || fabs(t) < 1e-300) { copy3(w, x0); }
    else if (t > 1 || t == 1) { copy3(w, b); }
    else { copy3(w, x0); addscl3(w, dd, t); }
    if (wit) copy3(wit, w);
    real r[3]; sub3(r, w, P); return dot3(r, r);
}
static real point_tri_dist2(const real *P, const real *x0, const real *B, const real *Cc, real *wit) {
    real d1[3], d2[3], a[3];
    sub3(d1, B, x0); sub3(d2, Cc, x0); sub3(a, x0, P);
    real u = dot3(a, a), v = dot3(d1, d1), w = dot3(d2, d2), p = dot3(a, d1), q = dot3(a, d2), r = dot3(d1, d2);
    real den = w*v - r*r, s = -1, t = -1;
    if (fabs(den) > 0) { s = (q*r - w*p) / den; t = (-s*r - q) / w; }
    if (s >= 0 && s <= 1 && t >= 0 && t <= 1 && t + s <= 1) {
        if (wit) { copy3(wit, x0); addscl3(wit, d1, s); addscl3(wit, d2, t); }
        real dist = s*s*v + t*t*w + 2*s*t*r + 2*s*p + 2*t*q + u;
        return dist < 0 ? 0 : dist;
    }
    real w2[3], best = point_seg_dist2(P, x0, B, wit), dist;
    dist = point_seg_dist2(P, x0, Cc, w2); if (dist < best) { best = dist; if (wit) copy3(wit, w2); }
    dist = point_seg_dist2(P, B, Cc, w2); if (dist < best) { best = dist; if (wit) copy3(wit, w2); }
    return best;
}
static void portal_dir(const mpr_sup *p, real *dir) {
    real a[3], b[3]; sub3(a, p[2].v, p[1].v); sub3(b, p[3].v, p[1].v); cross3(dir, a, b); normalize3(dir);
}
static void expand_portal(mpr_sup *p, const mpr_sup *v4) {
    real v4v0[3]; cross3(v4v0, v4->v, p[0].v);
    if (dot3(p[1].v, v4v0) > 0) { if (dot3(p[2].v, v4v0) > 0) p[1] = *v4; else p[3] = *v4; }
    else { if (dot3(p[3].v, v4v0) > 0) p[2] = *v4; else p[1] = *v4; }
}
static int portal_reach_tol(const mpr_sup *p, const mpr_sup *v4, const real *dir, real tol) {
    real dv4 = dot3(v4->v, dir), d1 = dv4 - dot3(p[1].v, dir), d2 = dv4 - dot3(p[2].v, dir), d3 = dv4 - dot3(p[3].v, dir);
    real mn = d1 < d2 ? d1 : d2; mn = mn < d3 ? mn : d3;
    return mn < tol;
}
static void mpr_find_pos(const mpr_sup *p, real *pos) {
    real dir[3], b[4], t[3], sum;
    portal_dir(p, dir);
    cross3(t, p[1].v, p[2].v); b[0] = dot3(t, p[3].v);
    cross3(t, p[3].v, p[2].v); b[1] = dot3(t, p[0].v);
    cross3(t, p[0].v, p[1].v); b[2] = dot3(t, p[3].v);
    cross3(t, p[2].v, p[1].v); b[3] = dot3(t, p[0].v);
    sum = b[0] + b[1] + b[2] + b[3];
    if (sum <= 0) {
        b[0] = 0;
        cross3(t, p[2].v, p[3].v); b[1] = dot3(t, dir);
        cross3(t, p[3].v, p[1].v); b[2] = dot3(t, dir);
        cross3(t, p[1].v, p[2].v); b[3] = dot3(t, dir);
        sum = b[1] + b[2] + b[3];
    }
    real inv = 1.0 / sum, p1[3] = {0,0,0}, p2[3] = {0,0,0};
    for (int i = 0; i < 4; i++) { addscl3(p1, p[i].v1, b[i]); addscl3(p2, p[i].v2, b[i]); }
    for (int k = 0; k < 3; k++) pos[k] = 0.5 * inv * (p1[k] + p2[k]);
}
/* returns 1 and fills depth/dir/pos when the geoms penetrate */
static int mpr_penetration(const ho_model *m, const ho_data *d, int g1, int g2, real *depth, real *dirout, real *pos) {
    const real eps = 2.220446049250313e-16, tol = m->opt[OPT_MPR_TOLERANCE];
    const int maxit = (int)m->opt[OPT_MPR_ITERATIONS];
    mpr_sup p[4], v4;
    real dir[3], va[3], vb[3], dt;
    /* discover portal */
    copy3(p[0].v1, d->gpos + 3*g1); copy3(p[0].v2, d->gpos + 3*g2); sub3(p[0].v, p[0].v1, p[0].v2);
    if (fabs(p[0].v[0]) < eps && fabs(p[0].v[1]) < eps && fabs(p[0].v[2]) < eps) p[0].v[0] += 1e-5;
    scl3(dir, p[0].v, -1); normalize3(dir);
    mpr_support(m, d, g1, g2, dir, &p[1]);
    dt = dot3(p[1].v, dir);
    if (dt < eps) return 0;
    cross3(dir, p[0].v, p[1].v);
    if (dot3(dir, dir) < eps*eps) {
        /* origin on the v0-v1 segment: touching (depth 0) or segment penetration */
        real l1 = norm3(p[1].v);
        if (l1 < eps) return 0;
        *depth = l1; copy3(dirout, p[1].v); normalize3(dirout);
        for (int k = 0; k < 3; k++) pos[k] = 0.5 * (p[1].v1[k] + p[1].v2[k]);
        return 1;
    }
    normalize3(dir);
    mpr_support(m, d, g1, g2, dir, &p[2]);
    if (dot3(p[2].v, dir) < eps) return 0;
    sub3(va, p[1].v, p[0].v); sub3(vb, p[2].v, p[0].v); cross3(dir, va, vb); normalize3(dir);
    if (dot3(dir, p[0].v) > 0) { mpr_sup t = p[1]; p[1] = p[2]; p[2] = t; scl3(dir, dir, -1); }
    for (int it = 0; ; it++) {
        if (it > 100) return 0;
        mpr_support(m, d, g1, g2, dir, &p[3]);
        if (dot3(p[3].v, dir) < eps) return 0;
        int cont = 0;
        cross3(va, p[1].v, p[3].v);
        dt = dot3(va, p[0].v);
        if (dt < -eps) { p[2] = p[3]; cont = 1; }
        if (!cont) {
            cross3(va, p[3].v, p[2].v);
            dt = dot3(va, p[0].v);
            if (dt < -eps) { p[1] = p[3]; cont = 1; }
        }
        if (!cont) break;
        sub3(va, p[1].v, p[0].v); sub3(vb, p[2].v, p[0].v); cross3(dir, va, vb); normalize3(dir);
    }
    /* refine portal */
    for (int it = 0; ; it++) {
        portal_dir(p, dir);
        if (dot3(dir, p[1].v) >= -eps) break;                 /* portal encapsulates the origin */
        mpr_support(m, d, g1, g2, dir, &v4);
        if (dot3(v4.v, dir) < -eps || portal_reach_tol(p, &v4, dir, tol) || it > maxit) return 0;
        expand_portal(p, &v4);
    }
    /* find penetration */
    for (int it = 0; ; it++) {
        portal_dir(p, dir);
        mpr_support(m, d, g1, g2, dir, &v4);
        if (portal_reach_tol(p, &v4, dir, tol) || it > maxit) {
            real org[3] = {0,0,0}, pdir[3];
            *depth = sqrt(point_tri_dist2(org, p[1].v, p[2].v, p[3].v, pdir));
            if (fabs(pdir[0]) < eps && fabs(pdir[1]) < eps && fabs(pdir[2]) < eps) copy3(pdir, dir);
            normalize3(pdir); copy3(dirout, pdir);
            mpr_find_pos(p, pos);
            return 1;
        }
        expand_portal(p, &v4);
    }
}
static void collide_convex(const ho_model *m, ho_data *d, int pair, int g1, int g2) {
    real depth, dir[3], pos[3];
    ho_dbg_mpr_calls_++; ho_dbg_mpr_cur_ = 0;
    if (mpr_penetration(m, d, g1, g2, &depth, dir, pos)) add_contact(m, d, pair, pos, dir, -depth);
    if (ho_dbg_mpr_cur_ > ho_dbg_mpr_max_) ho_dbg_mpr_max_ = ho_dbg_mpr_cur_;
}

static void ho_collision(const ho_model *m, ho_data *d) {
    d->ncon = 0;
    for (int p = 0; p < m->npair; p++) {
        int g1 = m->pair_geom1[p], g2 = m->pair_geom2[p];
        /* mj_collideGeoms bounding test (margin 0) */
        if (m->geom_type[g1] == GEOM_PLANE) {
            const real *pm = d->gmat + 9*g1; real n[3] = { pm[2], pm[5], pm[8] }, r[3];
            sub3(r, d->gpos + 3*g2, d->gpos + 3*g1);
            if (dot3(r, n) > m->geom_rbound[g2]) continue;
        } else {
            real r[3]; sub3(r, d->gpos + 3*g2, d->gpos + 3*g1);
            real bound = m->geom_rbound[g1] + m->geom_rbound[g2];
            if (dot3(r, r) > bound*bound) continue;
        }
        switch (m->pair_fn[p]) {
        case FN_PLANE_BOX: collide_plane_box(m, d, p, g1, g2); break;
        case FN_PLANE_CONVEX: collide_plane_convex(m, d, p, g1, g2); break;
        case FN_BOX_BOX: collide_box_box(m, d, p, g1, g2); break;
        default: collide_convex(m, d, p, g1, g2); break;
        }
    }
}

/* ------------------------------------------------------------------ a-2.4 constraints */
static real impedance(const real *solimp, real pos) {
    real dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
    dmin = dmin < MINIMP ? MINIMP : (dmin > MAXIMP ? MAXIMP : dmin);
    dmax = dmax < MINIMP ? MINIMP : (dmax > MAXIMP ? MAXIMP : dmax);
    width = width < MINVAL ? MINVAL : width;
    mid = mid < MINIMP ? MINIMP : (mid > MAXIMP ? MAXIMP : mid);
    power = power < 1 ? 1 : power;
    if (dmin == dmax || width <= MINVAL) return 0.5*(dmin + dmax);
    real x = fabs(pos) / width, y;
    if (x >= 1) return dmax;
    if (x <= 0) return dmin;
    if (power == 1) y = x;
    else if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
    else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
    return dmin + y*(dmax - dmin);
}

static void point_jac_row(const ho_model *m, const ho_data *d, int link, const real *pt,
                          const real *dirp, const real *dirr, real sign, real *row) {
    /* row += sign * (dirp . jacp + dirr . jacr) for a point on `link` */
    if (link == 0) return;
    for (int k = m->link_dofadr[link] + m->link_dofnum[link] - 1; k >= 0; k = m->dof_parent[k]) {
        real v[3], acc = 0;
        if (dirp) { dof_point_vel(d, k, pt, v); acc += dot3(v, dirp); }
        if (dirr) acc += dot3(d->dof_ang + 3*k, dirr);
        row[k] += sign * acc;
    }
}

static void ho_make_constraint(const ho_model *m, ho_data *d) {
    int nv = m->nv, ne = 0;
    real impratio = m->opt[OPT_IMPRATIO];
    /* joint limits (mj_instantiateLimit): lower side then upper side */
    for (int k = 0; k < nv && ne < m->njmax; k++) {
        if (!m->dof_limited[k]) continue;
        real q = d->qpos[m->dof_qposadr[k]];
        for (int side = 0; side < 2; side++) {
            real dist = side == 0 ? q - m->dof_range[2*k] : m->dof_range[2*k+1] - q;
            if (dist < 0 && ne < m->njmax) {
                real *row = d->efc_J + (size_t)ne*nv;
                memset(row, 0, sizeof(real)*(size_t)nv);
                row[k] = side == 0 ? 1.0 : -1.0;
                real imp = impedance(m->dof_solimp + 5*k, dist), dmax = m->dof_solimp[5*k+1];
                real tc = m->dof_solref[2*k], dr = m->dof_solref[2*k+1];
                dmax = dmax < MINIMP ? MINIMP : (dmax > MAXIMP ? MAXIMP : dmax);
                d->efc_pos[ne] = dist; d->efc_type[ne] = 0;
                d->efc_K[ne] = imp / (dmax*dmax*tc*tc*dr*dr);  /* K * imp folded together */
                d->efc_B[ne] = 2.0 / (dmax*tc);
                real R = (1 - imp)/imp * m->dof_invweight0[k];
                d->efc_R[ne] = R < MINVAL ? MINVAL : R;
                ne++;
            }
        }
    }
    d->nlimit_active = ne;
    /* contacts (mj_instantiateContact, elliptic cones) */
    for (int c = 0; c < d->ncon; c++) {
        ho_contact *con = d->contact + c;
        int dim = con->dim;
        if (ne + dim > m->njmax) { d->ncon = c; break; }
        con->efc_address = ne;
        real imp = impedance(con->solimp, con->dist), dmax = con->solimp[1];
        dmax = dmax < MINIMP ? MINIMP : (dmax > MAXIMP ? MAXIMP : dmax);
        real tc = con->solref[0], dr = con->solref[1];
        real tran = m->geom_invweight[2*con->geom1] + m->geom_invweight[2*con->geom2];
        real rot = m->geom_invweight[2*con->geom1+1] + m->geom_invweight[2*con->geom2+1];
        for (int j = 0; j < dim; j++) {
            real *row = d->efc_J + (size_t)(ne+j)*nv;
            memset(row, 0, sizeof(real)*(size_t)nv);
            const real *ax = con->frame + 3*(j % 3);
            if (j < 3) { point_jac_row(m, d, con->link2, con->pos, ax, NULL, 1.0, row); point_jac_row(m, d, con->link1, con->pos, ax, NULL, -1.0, row); }
            else { point_jac_row(m, d, con->link2, con->pos, NULL, ax, 1.0, row); point_jac_row(m, d, con->link1, con->pos, NULL, ax, -1.0, row); }
            d->efc_pos[ne+j] = j == 0 ? con->dist : 0.0;
            d->efc_type[ne+j] = j == 0 ? 1 : 2;
            d->efc_K[ne+j] = j == 0 ? imp / (dmax*dmax*tc*tc*dr*dr) : 0.0;   /* friction rows: K = 0 */
            d->efc_B[ne+j] = 2.0 / (dmax*tc);
            real R = (1 - imp)/imp * (j < 3 ? tran : rot);
            d->efc_R[ne+j] = R < MINVAL ? MINVAL : R;
        }
        /* elliptic-cone regulariser adjustment */
        if (dim > 1) {
            d->efc_R[ne+1] = d->efc_R[ne] / (impratio < MINVAL ? MINVAL : impratio);
            for (int j = 2; j < dim; j++)
                d->efc_R[ne+j] = d->efc_R[ne+1] * con->friction[0]*con->friction[0] / (con->friction[j-1]*con->friction[j-1]);
            con->mu = con->friction[0] * sqrt(d->efc_R[ne+1] / d->efc_R[ne]);
        } else con->mu = con->friction[0];
        ne += dim;
    }
    d->nefc = ne;
    /* mj_referenceConstraint: aref = -B*(J qvel) - K*imp*(pos - margin) */
    for (int i = 0; i < ne; i++) {
        real v = 0; const real *row = d->efc_J + (size_t)i*nv;
        for (int k = 0; k < nv; k++) v += row[k]*d->qvel[k];
        d->efc_vel[i] = v;
        d->efc_aref[i] = -d->efc_B[i]*v - d->efc_K[i]*d->efc_pos[i];
        d->efc_D[i] = 1.0 / d->efc_R[i];
    }
}

/* ------------------------------------------------------------------ a-2.6 solver */
/* cost / gradient / Hessian of one elliptic contact at residual x = jar (dim entries) */
static real cone_eval(const ho_contact *con, const real *D, const real *x, real *g, real *H) {
    int dim = con->dim; real mu = con->mu, U[6], N, T2 = 0, T;
    for (int j = 0; j < dim; j++) g[j] = 0;
    if (H) memset(H, 0, sizeof(real)*36);
    U[0] = x[0]*mu; N = U[0];
    for (int j = 1; j < dim; j++) { U[j] = x[j]*con->friction[j-1]; T2 += U[j]*U[j]; }
    T = sqrt(T2);
    if (N >= mu*T || (T <= 0 && N >= 0)) return 0;                                   /* top zone */
    if (mu*N + T <= 0 || (T <= 0 && N < 0)) {                                       /* bottom zone */
        real c = 0;
        for (int j = 0; j < dim; j++) { c += 0.5*D[j]*x[j]*x[j]; g[j] = D[j]*x[j]; if (H) H[6*j+j] = D[j]; }
        return c;
    }
    real Dm = D[0] / (mu*mu*(1 + mu*mu)), NT = N - mu*T, gn[6];                     /* middle zone */
    gn[0] = mu;
    for (int j = 1; j < dim; j++) gn[j] = -mu * U[j] * con->friction[j-1] / T;
    for (int j = 0; j < dim; j++) g[j] = Dm*NT*gn[j];
    if (H) {
        for (int j = 0; j < dim; j++) for (int k = 0; k < dim; k++) H[6*j+k] = Dm*gn[j]*gn[k];
        for (int j = 1; j < dim; j++) for (int k = 1; k < dim; k++) {
            real fj = con->friction[j-1], fk = con->friction[k-1];
            real h = -mu*fj*fk*((j == k ? 1.0/T : 0.0) - U[j]*U[k]/(T*T2));
            H[6*j+k] += Dm*NT*h;
        }
    }
    return 0.5*Dm*NT*NT;
}

/* evaluate constraint cost at residual jar; optional gradient wrt jar (= -force) */
static real constraint_cost(const ho_data *d, const real *jar, real *gout) {
    real cost = 0;
    for (int i = 0; i < d->nlimit_active; i++) {
        if (jar[i] < 0) { cost += 0.5*d->efc_D[i]*jar[i]*jar[i]; if (gout) gout[i] = d->efc_D[i]*jar[i]; }
        else if (gout) gout[i] = 0;
    }
    for (int c = 0; c < d->ncon; c++) {
        const ho_contact *con = d->contact + c; int a = con->efc_address; real g[6];
        cost += cone_eval(con, d->efc_D + a, jar + a, g, NULL);
        if (gout) for (int j = 0; j < con->dim; j++) gout[a+j] = g[j];
    }
    return cost;
}

/* 1-D derivatives of the total cost along the search direction at step alpha */
static void ls_eval(const ho_data *d, const real *jar, const real *jv, real alpha,
                    real g1, real g2, real *dphi, real *ddphi) {
    real dp = g1 + alpha*g2, hp = g2;
    for (int i = 0; i < d->nlimit_active; i++) {
        real x = jar[i] + alpha*jv[i];
        if (x < 0) { dp += d->efc_D[i]*x*jv[i]; hp += d->efc_D[i]*jv[i]*jv[i]; }
    }
    for (int c = 0; c < d->ncon; c++) {
        const ho_contact *con = d->contact + c; int a = con->efc_address, dim = con->dim;
        real x[6], g[6], H[36];
        for (int j = 0; j < dim; j++) x[j] = jar[a+j] + alpha*jv[a+j];
        cone_eval(con, d->efc_D + a, x, g, H);
        for (int j = 0; j < dim; j++) {
            dp += g[j]*jv[a+j];
            for (int k = 0; k < dim; k++) hp += jv[a+j]*H[6*j+k]*jv[a+k];
        }
    }
    *dphi = dp; *ddphi = hp;
}

static void ho_solve(const ho_model *m, ho_data *d) {
    int nv = m->nv, ne = d->nefc;
    real *qacc = d->qacc;
    if (ne == 0) {
        memcpy(qacc, d->qacc_smooth, sizeof(real)*(size_t)nv);
        memset(d->qfrc_constraint, 0, sizeof(real)*(size_t)nv);
        d->solver_niter = 0; d->solver_cost = 0;
        return;
    }
    static __thread real jar[NEFC_MAX], jv[NEFC_MAX], gr[NEFC_MAX];
    real Ma[NV_MAX], grad[NV_MAX], search[NV_MAX], Mv[NV_MAX], H[NV_MAX*NV_MAX], Lh[NV_MAX*NV_MAX];
    const real tol = m->opt[OPT_TOLERANCE], ls_tol = m->opt[OPT_LS_TOLERANCE];
    const int maxit = (int)m->opt[OPT_ITERATIONS], ls_maxit = (int)m->opt[OPT_LS_ITERATIONS];
    const real scale = 1.0 / (m->opt[OPT_MEANINERTIA] * (nv > 1 ? nv : 1));
#define EVAL_AT(a, costvar) do { \
        real gauss_ = 0; \
        for (int i_ = 0; i_ < nv; i_++) { real s_ = 0; for (int k_ = 0; k_ < nv; k_++) s_ += d->M[i_*nv+k_]*(a)[k_]; Ma[i_] = s_; } \
        for (int i_ = 0; i_ < nv; i_++) gauss_ += 0.5*((a)[i_] - d->qacc_smooth[i_])*(Ma[i_] - d->qfrc_smooth[i_]); \
        for (int i_ = 0; i_ < ne; i_++) { real s_ = -d->efc_aref[i_]; const real *r_ = d->efc_J + (size_t)i_*nv; \
            for (int k_ = 0; k_ < nv; k_++) s_ += r_[k_]*(a)[k_]; jar[i_] = s_; } \
        costvar = gauss_ + constraint_cost(d, jar, gr); } while (0)
    /* warm start: the cheaper of qacc_warmstart and qacc_smooth */
    real cost_w, cost_s, cost;
    EVAL_AT(d->qacc_warmstart, cost_w);
    EVAL_AT(d->qacc_smooth, cost_s);
    if (cost_w < cost_s) { memcpy(qacc, d->qacc_warmstart, sizeof(real)*(size_t)nv); EVAL_AT(qacc, cost); }
    else { memcpy(qacc, d->qacc_smooth, sizeof(real)*(size_t)nv); cost = cost_s; }
    int iter = 0;
    d->solver_ntrace = 0; d->solver_ls_max = 0; d->solver_refused = 0;
    d->solver_trace[d->solver_ntrace++] = cost;
    for (; iter < maxit; iter++) {
        /* gradient and Hessian */
        for (int i = 0; i < nv; i++) {
            real s = Ma[i] - d->qfrc_smooth[i];
            for (int r = 0; r < ne; r++) s += d->efc_J[(size_t)r*nv+i]*gr[r];
            grad[i] = s;
        }
        memcpy(H, d->M, sizeof(real)*(size_t)nv*(size_t)nv);
        for (int r = 0; r < d->nlimit_active; r++) if (jar[r] < 0) {
            const real *row = d->efc_J + (size_t)r*nv;
            for (int i = 0; i < nv; i++) if (row[i] != 0) for (int k = 0; k < nv; k++) H[i*nv+k] += d->efc_D[r]*row[i]*row[k];
        }
        for (int c = 0; c < d->ncon; c++) {
            const ho_contact *con = d->contact + c; int a = con->efc_address, dim = con->dim;
            real g[6], Hc[36];
            cone_eval(con, d->efc_D + a, jar + a, g, Hc);
            for (int j = 0; j < dim; j++) for (int k2 = 0; k2 < dim; k2++) {
                real h = Hc[6*j+k2]; if (h == 0) continue;
                const real *rj = d->efc_J + (size_t)(a+j)*nv, *rk = d->efc_J + (size_t)(a+k2)*nv;
                for (int i = 0; i < nv; i++) if (rj[i] != 0) for (int k = 0; k < nv; k++) H[i*nv+k] += h*rj[i]*rk[k];
            }
        }
        if (cholesky(Lh, H, nv) != 0) { d->bad = 1; break; }
        real gnorm = 0;
        for (int i = 0; i < nv; i++) { search[i] = -grad[i]; gnorm += grad[i]*grad[i]; }
        gnorm = sqrt(gnorm);
        if (scale*gnorm < tol) break;
        chol_solve(Lh, nv, search);
        /* exact line search along `search` (safeguarded 1-D Newton on phi') */
        real g1 = 0, g2 = 0, snorm = 0;
        for (int i = 0; i < nv; i++) { real s = 0; for (int k = 0; k < nv; k++) s += d->M[i*nv+k]*search[k]; Mv[i] = s; }
        for (int i = 0; i < nv; i++) { g1 += search[i]*(Ma[i] - d->qfrc_smooth[i]); g2 += search[i]*Mv[i]; snorm += search[i]*search[i]; }
        snorm = sqrt(snorm);
        for (int r = 0; r < ne; r++) { real s = 0; const real *row = d->efc_J + (size_t)r*nv; for (int k = 0; k < nv; k++) s += row[k]*search[k]; jv[r] = s; }
        real gtol = tol * ls_tol * snorm / scale, dp, hp, alpha, lo = 0, hi = -1;
        ls_eval(d, jar, jv, 0.0, g1, g2, &dp, &hp);
        if (dp >= 0 || hp <= 0) break;
        alpha = -dp / hp;
        /* phi' has kinks where a cone changes its zone: the 1-D Newton iteration can then hop between the two ends of its bracket for ever
         * (found in round 4 on the cupboard scene: 0.2575 <-> 0.4224 for all 50 iterations, the step accepted at the end RAISED the cost and
         * the improvement test ended the solve unconverged).  From the sixth evaluation on every second step bisects the bracket (the HIP
         * line search carries the same rule). */
        for (int it = 0; it < ls_maxit; it++) {
            ls_eval(d, jar, jv, alpha, g1, g2, &dp, &hp);
            if (it + 1 > d->solver_ls_max) d->solver_ls_max = it + 1;
            if (fabs(dp) < gtol) break;
            if (dp < 0) lo = alpha; else hi = alpha;
            real nxt = alpha - dp / hp;
            if (!(nxt > lo) || (hi > 0 && !(nxt < hi))) nxt = hi > 0 ? 0.5*(lo + hi) : 2*alpha;
            else if (it >= 5 && (it & 1) && hi > 0) nxt = 0.5*(lo + hi);
            alpha = nxt;
        }
        if (alpha <= 0) break;
        for (int i = 0; i < nv; i++) qacc[i] += alpha*search[i];
        real oldcost = cost;
        EVAL_AT(qacc, cost);
        if (cost > oldcost) {          /* a step that raises the cost is never taken (MuJoCo's search returns a point no worse than alpha = 0) */
            for (int i = 0; i < nv; i++) qacc[i] -= alpha*search[i];
            EVAL_AT(qacc, cost);
            d->solver_refused = 1;
            break;
        }
        if (d->solver_ntrace < 104) d->solver_trace[d->solver_ntrace++] = cost;
        if (scale*(oldcost - cost) < tol) { iter++; break; }
    }
    d->solver_niter = iter; d->solver_cost = cost;
    for (int r = 0; r < ne; r++) d->efc_force[r] = -gr[r];
    for (int i = 0; i < nv; i++) { real s = 0; for (int r = 0; r < ne; r++) s += d->efc_J[(size_t)r*nv+i]*d->efc_force[r]; d->qfrc_constraint[i] = s; }
#undef EVAL_AT
}

/* ------------------------------------------------------------------ forward / step */
void ho_forward(const ho_model *m, ho_data *d) {
    ho_kinematics(m, d);
    ho_inertia(m, d);
    ho_collision(m, d);
    ho_make_constraint(m, d);   /* needs qvel only for aref, as in mj_referenceConstraint */
    ho_smooth(m, d);
    ho_solve(m, d);
}

/* a-2.7 mj_Euler: implicit joint damping, semi-implicit position update */
static void ho_euler(const ho_model *m, ho_data *d) {
    int nv = m->nv; real h = m->opt[OPT_TIMESTEP];
    real qacc[NV_MAX], A[NV_MAX*NV_MAX], La[NV_MAX*NV_MAX];
    int damped = 0;
    for (int k = 0; k < nv; k++) if (m->dof_damping[k] > 0) damped = 1;
    if (damped) {
        memcpy(A, d->M, sizeof(real)*(size_t)nv*(size_t)nv);
        /* MuJoCo: (M + h B) a = qfrc_smooth + qfrc_constraint.  With the option the right-hand side is M qacc - the same vector at the
         * solver's fixed point (its gradient vanishes there), what the HIP path integrates (deviations): the parity tests
         * report the difference from both forms */
        for (int k = 0; k < nv; k++) {
            A[k*nv+k] += h*m->dof_damping[k];
            if (d->euler_rhs_macc) { real t = 0; for (int j = 0; j < nv; j++) t += d->M[k*nv+j]*d->qacc[j]; qacc[k] = t; }
            else qacc[k] = d->qfrc_smooth[k] + d->qfrc_constraint[k];
        }
        if (cholesky(La, A, nv) == 0) chol_solve(La, nv, qacc); else d->bad = 1;
    } else memcpy(qacc, d->qacc, sizeof(real)*(size_t)nv);
    for (int k = 0; k < nv; k++) d->qvel[k] += h*qacc[k];
    for (int l = 1; l < m->nlink; l++) {
        int d0 = m->link_dofadr[l];
        if (m->link_free[l]) {
            int a = m->link_qposadr[l];
            for (int k = 0; k < 3; k++) d->qpos[a+k] += h*d->qvel[d0+k];
            /* mju_quatIntegrate: rotate by |w| h about w (local frame) */
            real w[3] = { d->qvel[d0+3], d->qvel[d0+4], d->qvel[d0+5] }, ang = norm3(w)*h;
            if (ang > 0) {
                real ax[3]; copy3(ax, w); normalize3(ax);
                real s = sin(0.5*ang), qr[4] = { cos(0.5*ang), ax[0]*s, ax[1]*s, ax[2]*s }, qn[4];
                quatmul(qn, d->qpos + a + 3, qr); quatnorm(qn);
                memcpy(d->qpos + a + 3, qn, sizeof qn);
            }
        } else {
            for (int k = d0; k < d0 + m->link_dofnum[l]; k++) d->qpos[m->dof_qposadr[k]] += h*d->qvel[k];
        }
    }
    memcpy(d->qacc_warmstart, d->qacc, sizeof(real)*(size_t)nv);
    d->time += h;
}

static void check_state(const ho_model *m, ho_data *d) {
    for (int i = 0; i < m->nq; i++) if (!isfinite(d->qpos[i]) || fabs(d->qpos[i]) > 1e10) d->bad = 1;
    for (int i = 0; i < m->nv; i++) if (!isfinite(d->qvel[i]) || fabs(d->qvel[i]) > 1e10) d->bad = 1;
}

void ho_step(const ho_model *m, ho_data *d) {
    check_state(m, d);
    ho_forward(m, d);
    ho_euler(m, d);
    check_state(m, d);
}

/* world position of an (original, pre-folding) body; mocap bodies read mocap_pos */
void ho_body_xpos(const ho_model *m, const ho_data *d, int body, real *out) {
    if (m->body_mocap[body]) { copy3(out, d->mocap_pos); return; }
    int l = m->body_link[body]; real t[3];
    mulmv3(t, d->xmat + 9*l, m->body_pos + 3*body); add3(out, d->xpos + 3*l, t);
}

/* HSREnv.step (hsr/env.py:115-135): write ctrl, run <= nsub substeps, test the goal after each,
 * break on success.  goal_body < 0 means `goals is None` (done stays False).  xpos is the one
 * computed by the substep's forward pass (i.e. at the pre-integration qpos), exactly what
 * sim.data.get_body_xpos returns after sim.step().  Returns substeps executed; *done set. */
int ho_env_step(const ho_model *m, ho_data *d, const real *ctrl, int nsub, int goal_body,
                const real *goal, real geofence, int *done) {
    memcpy(d->ctrl, ctrl, sizeof(real)*(size_t)m->nu);
    *done = 0;
    int i = 0;
    for (; i < nsub; i++) {
        ho_step(m, d);
        if (goal_body >= 0) {
            real p[3], r[3]; ho_body_xpos(m, d, goal_body, p); sub3(r, p, goal);
            if (sqrt(dot3(r, r)) < geofence) { *done = 1; i++; break; }
        }
    }
    return i;
}

/* ---- accessors for ctypes */
real *ho_qpos(ho_data *d) { return d->qpos; }
real *ho_qvel(ho_data *d) { return d->qvel; }
real *ho_ctrl(ho_data *d) { return d->ctrl; }
real *ho_qacc(ho_data *d) { return d->qacc; }
real *ho_qacc_warmstart(ho_data *d) { return d->qacc_warmstart; }
real *ho_qacc_smooth(ho_data *d) { return d->qacc_smooth; }
real *ho_qfrc_smooth(ho_data *d) { return d->qfrc_smooth; }
real *ho_qfrc_bias(ho_data *d) { return d->qfrc_bias; }
real *ho_qfrc_constraint(ho_data *d) { return d->qfrc_constraint; }
real *ho_mocap_pos(ho_data *d) { return d->mocap_pos; }
real *ho_xpos(ho_data *d) { return d->xpos; }
real *ho_xquat(ho_data *d) { return d->xquat; }
real *ho_xmat(ho_data *d) { return d->xmat; }
real *ho_M(ho_data *d) { return d->M; }
real *ho_efc_J(ho_data *d) { return d->efc_J; }
real *ho_efc_force(ho_data *d) { return d->efc_force; }
real *ho_efc_aref(ho_data *d) { return d->efc_aref; }
real *ho_efc_R(ho_data *d) { return d->efc_R; }
real *ho_efc_pos(ho_data *d) { return d->efc_pos; }
real *ho_time(ho_data *d) { return &d->time; }
int ho_ncon(const ho_data *d) { return d->ncon; }
int ho_nefc(const ho_data *d) { return d->nefc; }
int ho_bad(const ho_data *d) { return d->bad; }
int ho_solver_niter(const ho_data *d) { return d->solver_niter; }
void ho_set_euler_rhs(ho_data *d, int m_qacc) { d->euler_rhs_macc = m_qacc != 0; }
/* solver introspection (tests only): which = 0 number of trace entries, 1 largest line-search evaluation count of an iteration, 2 a cost-raising step was refused,
 * 3 number of active limit rows (they precede the contact rows) */
int ho_solver_stat(const ho_data *d, int which) { return which == 0 ? d->solver_ntrace : which == 1 ? d->solver_ls_max : which == 2 ? d->solver_refused : d->nlimit_active; }
real *ho_solver_trace(ho_data *d) { return d->solver_trace; }
/* contact i -> out[0:5] friction, [5] first constraint row */
void ho_contact_get2(const ho_data *d, int i, real *out) {
    const ho_contact *c = d->contact + i;
    memcpy(out, c->friction, 5*sizeof(real)); out[5] = c->efc_address;
}
/* contact i -> out[0:3] pos, [3:12] frame, [12] dist, [13] geom1, [14] geom2, [15] dim, [16] mu */
void ho_contact_get(const ho_data *d, int i, real *out) {
    const ho_contact *c = d->contact + i;
    copy3(out, c->pos); memcpy(out + 3, c->frame, 9*sizeof(real)); out[12] = c->dist;
    out[13] = c->geom1; out[14] = c->geom2; out[15] = c->dim; out[16] = c->mu;
}

/* ---- batched env-step for the cpu_baseline leg: n independent envs, OpenMP over envs.
 * state arrays are [n, nq] / [n, nv] / [n, nu] row-major fp64 and are updated in place. */
int ho_batch_env_step(const ho_model *m, int n, real *qpos, real *qvel, real *warm, const real *ctrl,
                      const real *mocap, int nsub, int goal_body, real geofence, int *done, int nthreads) {
    int total = 0;
#pragma omp parallel for num_threads(nthreads) reduction(+:total) schedule(dynamic, 1)
    for (int e = 0; e < n; e++) {
        ho_data *d = ho_data_new(m);
        ho_reset(m, d);
        memcpy(d->qpos, qpos + (size_t)e*m->nq, sizeof(real)*(size_t)m->nq);
        memcpy(d->qvel, qvel + (size_t)e*m->nv, sizeof(real)*(size_t)m->nv);
        memcpy(d->qacc_warmstart, warm + (size_t)e*m->nv, sizeof(real)*(size_t)m->nv);
        copy3(d->mocap_pos, mocap + 3*(size_t)e);
        int dn = 0;
        total += ho_env_step(m, d, ctrl + (size_t)e*m->nu, nsub, goal_body, mocap + 3*(size_t)e, geofence, &dn);
        done[e] = dn;
        memcpy(qpos + (size_t)e*m->nq, d->qpos, sizeof(real)*(size_t)m->nq);
        memcpy(qvel + (size_t)e*m->nv, d->qvel, sizeof(real)*(size_t)m->nv);
        memcpy(warm + (size_t)e*m->nv, d->qacc_warmstart, sizeof(real)*(size_t)m->nv);
        ho_data_free(d);
    }
    return total;
}
