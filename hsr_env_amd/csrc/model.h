// Device-side model tables and per-batch state (struct-of-arrays, [field][env]).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

enum { DOF_SLIDE = 0, DOF_HINGE = 1, DOF_FREE_LIN = 2, DOF_FREE_ANG = 3 };
enum { GEOM_PLANE = 0, GEOM_SPHERE = 2, GEOM_CYLINDER = 5, GEOM_BOX = 6, GEOM_MESH = 7 };
enum { FN_PLANE_BOX = 0, FN_PLANE_CONVEX = 1, FN_BOX_BOX = 2, FN_CONVEX = 3 };

// constant tables (device pointers; fp64 blob values converted to fp32 once at load)
struct DevModel {
    int nq, nv, nu, nlink, nbody, ngeom, npair, nslot, nconmax, njmax, nM, ndense;
    float timestep, impratio, gravz, tolerance, ls_tolerance, mpr_tolerance, meaninertia;
    int iterations, ls_iterations, mpr_iterations, any_damping;
    int solimp_general;              // some solimp power is neither 1 nor 2 (mj_makeImpedance's powf arms are needed); 0 for every committed model
    const int *link_parent, *link_dofadr, *link_dofnum, *link_qposadr, *link_free;
    const float *link_pos, *link_mat, *link_mass, *link_com, *link_inertia;
    const int *link_dofmask, *link_depth;   // depth of a link in the kinematic tree (world = 0)
    int maxdepth;
    const int *dof_link, *dof_type, *dof_parent, *dof_qposadr, *dof_limited;
    const float *dof_axis, *dof_pos, *dof_damping, *dof_invweight0, *dof_range, *dof_solref, *dof_solimp;
    const int *body_link, *body_mocap;
    const float *body_pos, *body_mat;  // body frame in its link's frame
    const int *geom_type, *geom_link, *geom_meshadr, *geom_meshnum;
    const float *geom_pos, *geom_mat, *geom_size, *geom_rbound, *geom_invweight, *mesh_vert;
    const float *geom_aabb;            // [ngeom][6] bounding box in the geom frame: centre, half extents
    const float4 *mesh_vert4;          // hull vertices padded to float4 (LDS staging in k_collide)
    const int *pair_geom1, *pair_geom2, *pair_fn, *pair_condim, *pair_slot;
    const float *pair_friction, *pair_solref, *pair_solimp;
    const int *act_dof, *dof_act;     // dof_act[k]: actuator driving dof k or -1
    const float *pair_rec;            // [npair][16]: dim l1 l2 tran fri[5] B K solimp[5]  (B = 2 / (dmax timeconst), K = 1 / (dmax^2 timeconst^2 dampratio^2) of solref)
    const float *pair_geo;            // [npair][8]  g1 + 256 (geom1 is a plane)  g2  rbound1  rbound2 | fn slot maxcnt type1 (two float4 loads)
    const float *geom_rec;            // [ngeom][32] seven float4: link type nvert meshadr | lpos rbound | lmat[0..3] | lmat[4..7] | lmat[8] size | aabb centre - | aabb half -
    int npair_pad;                    // npair rounded up to 8: row length of the per-env pair-count table
    int nstatic_geom;                 // leading geoms attached to the world link (placed once per launch by the persistent kernel)
    int nfb;                          // free bodies whose six dofs (3 lin, 3 ang) form the tail of the dof vector, one after the other: their
                                      // contacts with the world are assembled per body in the Newton Hessian (solve_body.inc); 0 = none / not applicable
    const float *act_gear, *act_kp, *act_ctrlrange, *act_forcerange;
    const int *geom_ldsv;             // [ngeom] hulls staged in LDS by the persistent kernel (persist.h: hull area): first float4 slot, or -1 (the hull stays in global memory)
    const int *ldsv_src;              // [nldsv] vertex index in mesh_vert4 of every staged slot
    int nldsv;
    int kin3_match;                   // host side: row + 1 of the constant instance (cfg_consts.h) whose compile-time tree tables equal this model's, 0 = none
};

// env-step inputs / outputs of the caller (hsr_batch_step_dev), env-major as the C-ABI hands them over: the persistent kernel reads ctrl
// and writes obs / reward / done / nsteps itself (ctrl == NULL: the state arrays s.ctrl / s.done are used as they are)
struct StepIO { const float *ctrl; float *obs, *reward; uint8_t *done; int32_t *nsteps; };

// per-batch buffers; every array is [rows][N] with the env index fastest (coalesced across lanes)
struct DevState {
    int N;
    // persistent simulation state
    float *qpos, *qvel, *ctrl, *mocap, *warm, *time;
    int *done, *bad, *nsteps;
    // kinematics outputs
    float *xpos, *xmat;
    float *lvel;                      // [6 nlink][N] angular velocity, linear velocity of the link origin: same forward pass as xpos (obs kernel)
    // env-major copy of the solver's kinematic inputs: kin_aos[e][kstride] = ang[3nv] lin[3nv] anchor[3nv] link_dyn[15 nlink]
    float *kin_aos;
    int kstride;
    // collision outputs, env-major so that one env's lane group reads them coalesced:
    //   con[(e * nslot + slot) * 8 + k]  (pos 0-2, normal 3-5, dist 6), ncon_pair[e * npair_pad + p]
    float *con;
    int *ncon_pair;
    int *pair_count, *pair_list;   // per-pair work lists of the narrowphase: count[npair_pad], list[npair][N] env ids
    float *sepax;             // [4 npair][N] per (pair, env) of the MPR pairs: cached separating direction (0 = none), then the separation left along it
    int npair_sep;            // rows of sepax / 4
    int *septick, *tick;      // septick[npair][N]: the env's substep count (tick[N]) when the margin of (pair, env) was last brought up to date;
                              // a margin is valid on the very next substep of its env only (a pair culled in between moves unaccounted)
    unsigned *pair_pack;      // [npair padded to 8] packed sphere-cull records and [ngeom][8] narrowphase constants: global copies of the two
    float *geom_c;            // LDS tables of the persistent kernel, read instead of them by its TG instances (models whose LDS would not fit 8 workgroups per CU)
    // dynamics / solver outputs kept for introspection
    float *M, *qacc, *qacc_smooth, *qfrc_smooth, *qfrc_constraint;
    int *ncon, *nefc, *niter;
    // further goal terms AND-ed with the main one (hsr/env.py:124-126 `all(in_range(*g) for g in goals)`): term k holds when
    // |point(goal_a[k]) - point(goal_b[k])| < goal_d[k], point(body) = its xpos, or the env's mocap point for a mocap body
    int ngoal, goal_a[4], goal_b[4];
    float goal_d[4];
    // wave packing of the persistent kernel: slot_env[workgroup * envs_per_workgroup + group] = env index or -1 (NULL: identity);
    // trips[e] = Newton iterations env e ran in the last substeps of its last launch (what the packing is derived from)
    int *slot_env, *trips;
    // work queue of the persistent kernel (persist.h: q_claim / q_push); q_chunk = 0: off (one task per workgroup, the whole env-step)
    int q_chunk;
    int *q_head, *q_wpos, *q_items, *q_err;
    // solo servers of the persistent kernel (persist.h): the first solo_servers workgroups of a queued launch take no tasks; they wait for
    // envs that a worker has found hard (solo_trips or more Newton iterations per substep over a round) and run each of them ALONE in a wave
    // to the end of the env-step.  sq_items[ticket] = env | first substep << 20 (-1 until pushed); sq_ctl = { tickets taken, items
    // reserved, servers without an env, tasks that finished their last round }
    int solo_servers, solo_trips_x4, solo_min_left, sq_cap;      // sq_cap = entries of sq_items: every env can be handed over once, every server holds one more ticket
    int *sq_items, *sq_ctl;
    unsigned long long *capstat;     // [4] cap statistics (include/hsrsim.h: hsr_batch_cap_counts)
    unsigned long long *phase_cyc;   // diagnostic build only (HSR_PHASE_TIMING): per-phase cycle sums
};
