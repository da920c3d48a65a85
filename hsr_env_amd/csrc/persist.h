// Persistent per-env-group kernel: one launch advances every env through a whole env-step (up to n_substeps substeps of
// kinematics -> collision -> constraint solve -> Euler, with the per-substep goal test and early exit of HSREnv.step,
// hsr/env.py:115-135).  Envs never interact, so there is no grid-wide synchronisation between substeps: each single-wave
// workgroup (64/G envs, G lanes per env) loops on its own, state stays in registers/LDS, and the spread of Newton iteration
// counts between envs averages out over the substeps instead of stalling the whole GPU at every substep boundary
// (with per-substep kernels the mean workgroup lifetime was 110 us but the kernel lasted 244 us).
//
// Phases per substep and lane roles (DESIGN.md has the cycle table):
//   K  kinematics (kin2.h): A local transform of every link (lane = link, all at once), B world pose = product over the link's packed
//      ancestor list, C dof axes / anchors / velocity-field increments (lane = dof), D spatial velocity and bias acceleration as fields
//      accumulated over the chain dofs, E world inertia, com, RNE wrench; per-link constants from a 60-float LDS record built once per launch
//   C  collision: lane = moving geom places it (LDS geom cache; static geoms once per launch); lane = eight consecutive candidate pairs does
//      the bounding-sphere cull; survivors of ALL envs of the workgroup are compacted with wave ballots and box-culled one per lane; the
//      items that remain run the narrowphase side by side - one lane per plane item, one 8-lane sub-group per MPR or box-box item
//      (convex pairs: separating axis + stamped separation margin while apart, portal of the previous substep while penetrating);
//      contacts go to the env's fixed (pair, index) slots, counts stay in LDS
//   S  solve: the shared body solve_body.inc (lane = dof / contact / row), then the Euler update in registers
// The env-step inputs and outputs of the caller (ctrl; obs, reward, done, nsteps) are read and written by this kernel (StepIO); with more
// tasks than resident workgroups the launch runs as persistent workgroups over a work queue (q_claim / q_push below).
#pragma once
#include "collide.h"
#include "kin2.h"
#include "kin3.h"
#include "solve_mf.h"

template <int G> struct PersistLayout {
    int R, MS, oRows, oCnt, oB, envf, oPoly, oKin, oPair, oGeomC, oGeomS, oLinkTab, total;
    // tables_global: the packed pair records and the geom constants stay in global memory (DevState::pair_pack / geom_c)
    __host__ __device__ PersistLayout(int rows, int kstride, int npair_pad, int nlink, int ngeom, int nstatic, bool tables_global = false) {
        R = rows; MS = G + 1;
        int a = 5 * R > kstride ? 5 * R : kstride;                            // row scalars / kin record
        oRows = 0; oCnt = (a + 3) & ~3;                                       // pair counts survive phases A-D next to the kin record
        a = oCnt + (npair_pad + 3) / 4;                                        // one byte per pair
        int b = G * MS > C2_SIZE * G ? G * MS : C2_SIZE * G;                  // inertia matrix / contact records ...
        const int nmov = ngeom - nstatic;
        const int kin_tmp = 24 * nlink + 2 * G + (16 * nmov > 8 * G ? 16 * nmov : 8 * G);   // ... or link poses + local transforms + qpos/qvel staging + moving-geom placements (dof velocity increments before them)
        if (kin_tmp > b) b = kin_tmp;
        oB = (a + 3) & ~3;
        envf = (oB + b + 3) & ~3;
        oPoly = envf * (64 / G);                                               // box-box polygon scratch: 24 floats per 8-lane sub-group
        oKin = oPoly + 4 * 48;                                                 // per-link kinematic constants (kin2.h), shared by the envs of the workgroup
        oPair = oKin + KIN2_FLOATS * nlink;                                     // packed sphere-cull record per candidate pair (one dword), rows of 8
        oGeomC = oPair + (tables_global ? 0 : ((npair_pad + 7) & ~7));          // narrowphase constants of every geom (8 floats each)
        oGeomS = oGeomC + (tables_global ? 0 : 8 * ngeom);                                            // world placements of the static geoms (16 floats each)
        oLinkTab = oGeomS + 16 * nstatic;                                       // chain dof mask and mass of every link
        total = (oLinkTab + 2 * nlink + 3) & ~3;
    }
};

// ---- work queue of the persistent kernel (DevState::q_*; hsr_batch_step_dev turns it on when there are more tasks than workgroups
// the GPU holds at once, e.g. 4096 two-env tasks of cfg4 at 8192 envs on 2048 resident workgroups).  Static assignment makes such a
// launch end with whatever hard task happened to start in the second dispatch round (cfg4: 63 ms for 36 ms of work per slot); and
// hardness cannot be predicted.  So the env-step is cut into rounds of q_chunk substeps, round r holds one ticket per task,
// and a workgroup always takes the next ticket of the LOWEST round that has one left: a task that is behind (a hard one) is picked up
// the moment it finishes its previous round - it runs back to back, and everything else fills the machine around it.
//   q_head[r]  tickets of round r handed out so far (atomic counter; ticket h >= T: none left in this round)
//   q_wpos[r]  items of round r pushed so far;  q_items[r * T + i]: task id, -1 until pushed (round 0 is filled by the host)
// Every task passes through every round exactly once, so every ticket < T is served eventually: its holder waits for the item
// (the workgroup that runs the task's previous round is resident - tickets are only taken by running workgroups, lowest round first, so
// no cycle of waiting workgroups can form).  State handover: release fence + item store by the producer, item load + acquire fence
// by the consumer, both at agent scope (the eight XCDs have separate L2 caches).  A spin that exceeds its bound sets q_err and
// leaves: the launch then drains instead of hanging.
__device__ __forceinline__ bool q_claim(const DevState &s, int T, int R, int &rlo, int &task, int &round) {
    int t = -1, r = rlo;
    if (threadIdx.x == 0) {
        int h = 0;
        for (; r < R; r++) {
            if (__hip_atomic_load(&s.q_head[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= T) continue;
            h = __hip_atomic_fetch_add(&s.q_head[r], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (h < T) break;
        }
        if (r < R) {
            const int *slot = s.q_items + (size_t)r * T + h;
            int spins = 0;
            while ((t = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < 0) {
                __builtin_amdgcn_s_sleep(16);
                if (++spins > (1 << 22) || __hip_atomic_load(s.q_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {      // seconds: something is broken
                    __hip_atomic_store(s.q_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    t = -1; r = R;
                    break;
                }
            }
        }
    }
    t = __builtin_amdgcn_readfirstlane(t); r = __builtin_amdgcn_readfirstlane(r);
    rlo = r;
    if (t < 0) return false;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    task = t; round = r;
    return true;
}
__device__ __forceinline__ void q_push(const DevState &s, int T, int task, int round, bool last) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // every lane: its state stores are visible before the item is
    __builtin_amdgcn_wave_barrier();
    if (threadIdx.x == 0 && !last) {
        const int p = __hip_atomic_fetch_add(&s.q_wpos[round + 1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(s.q_items + (size_t)(round + 1) * T + p, task, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0 && last && s.sq_ctl) __hip_atomic_fetch_add(&s.sq_ctl[3], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);      // the solo servers leave when every task is through
}

// NVT: compile-time bound on nv (nv <= NVT <= G); the matrix loops of the solver run to NVT instead of G
// EXACT: nv == NVT, known at compile time (the `j < nv` guards of the unrolled matrix loops fold away)
// NDT (EXACT only, else -1): ndense at compile time - the dofs from NDT on never couple to another dof in M (free bodies)
// TG: pair records / geom constants read from global memory instead of LDS (for models with many pairs: 8 workgroups per CU)
// MT: DevModel, or a type derived from it whose static constexpr members HIDE the scalar fields of the model (sizes, solver
// options) with the values of one compiled configuration (cfg_consts.h): `m.nlink` is then a literal, the LDS layout a set of
// immediates, and no scalar load / SGPR is spent on any of them; hsr_batch_create checks the values against the loaded model
// SV: the instance carries the solo-server path (hsr_batch_set_solo): a second copy of the substep loop with its own register allocation; the
// default instances are compiled without it, a launch with servers picks the SV one
template <int G, int NVT, bool EXACT, int NDT = -1, bool TG = false, class MT = DevModel, bool SV = false>
__global__ void __launch_bounds__(64, 2) k_env_step_mf(const DevModel *__restrict__ mp, DevState s, int n_substeps, int goal_body, float geofence, int flags, StepIO io) {
    // the ~90 model fields stay in (constant-cached) memory and are read where they are used, instead of sitting in - and
    // spilling from - SGPRs for the whole launch
    const MT &m = *static_cast<const MT *>(mp);
    extern __shared__ __align__(16) float lds[];
    constexpr int EPB = 64 / G, NK = NVT, NDK = EXACT ? NDT : -1;
    const PersistLayout<G> L(m.njmax, (9 * (EXACT ? NVT : m.nv) + 15 * m.nlink + 15) & ~15, m.npair_pad, m.nlink, m.ngeom, m.nstatic_geom, TG);      // second argument = DevState::kstride
    const int tid0 = threadIdx.x;
    const int N = s.N, nv = EXACT ? NVT : m.nv, nq = m.nq, R = L.R, MS = L.MS;
    const int mode = 1, debug = 0;
    const bool dbg_store = flags & 1;          // introspection: contact counts / solver counters of each env's last substep go to global memory
    const bool hook_jv_per_contact = flags & 2, hook_majorant = flags & 4;      // tests: force the J v per contact / the PSD-majorant Newton step
    const bool hook_dense_chol = flags & 128;         // tests: the dense factorisation of the Newton Hessian even where the sparse one applies
    constexpr bool EXACT_CT = EXACT && !std::is_same<MT, DevModel>::value;      // sizes AND structure (nfb, ndense) known at compile time
    const bool hook_no_item_list = flags & 64;        // tests: cull every substep (the behaviour before the item list was kept over substeps)
    const bool hook_ignore_stamps = flags & 16;      // tests: trust a separation margin whatever its stamp (the behaviour before the stamps existed)
    const bool mpr_warm = !(flags & 8);       // the portal of a penetrating convex pair is carried to its next substep (hsr_batch_set_mpr_warm)
    int cap_con = 0, cap_row = 0, cap_item = 0, nsub_run = 0;      // cap statistics of this lane's env (lane c == 0 reports)
    int own_trips = 0;                                              // Newton iterations of this lane's env over its last (up to) 100 substeps
    // A task = the envs of one lane-group set (EPB envs) over a run of substeps.  Without the work queue (s.q_chunk == 0) there is one
    // task per workgroup: its blockIdx.x and the whole env-step.  With it (more tasks than resident workgroups: see q_claim) a
    // workgroup takes (task, round) tickets until none is left; the env state travels through the state arrays in between.
    int task = s.q_chunk ? 0 : blockIdx.x, q_round = 0, q_rlo = 0;      // queued launches: tasks come from q_claim (the grid may hold more workgroups than there are tasks)
    // SOLO (shadowed inside run_task): this workgroup runs ONE env, handed over by a worker that found it hard, in lane group 0 (solo server)
    // REP: the other lane groups of a solo server run the same env as REPLICAS - the same instructions on the same values, each group
    // with its own LDS region - and split what can be split: the contacts of the Newton Hessian / gradient go round the groups (one pass of
    // the matrix core covers EPB contacts, the partial sums are added across the rows); the collision front end runs once (group 0's
    // candidate pairs, one item list, the contact counts copied to the other groups)
    constexpr bool SOLO = false, REP = false;
    const int solo_env = -1;
    (void)solo_env;
    // everything derived from the lane id is declared through this macro: once for the prologue, once per substep from a
    // laundered copy of the lane id (so that LLVM does not hoist ~100 loop-invariant addresses out of the substep loop and
    // then spill them), once for the epilogue
#define PERSIST_LANE_VIEW(TID)                                                                                              \
    const int tid = (TID), g = tid / G, c = tid % G;                                                                         \
    const int e_raw = SOLO ? ((REP || g == 0) ? solo_env : -1) : (s.slot_env ? s.slot_env[task * EPB + g] : task * EPB + g); \
    const bool in_range = e_raw >= 0 && e_raw < N;                                                                           \
    const int e = in_range ? e_raw : 0;                                                                                      \
    float *E = lds + (size_t)g * L.envf;                                                                                     \
    float *rD = E + L.oRows, *rAref = rD + R, *rJar = rAref + R, *rJv = rJar + R, *rDw = rJv + R;            \
    float *kAng = E + L.oRows, *kLin = kAng + 3 * nv, *kAnc = kAng + 6 * nv, *lk = kAng + 9 * nv;                            \
    unsigned char *pcnt = reinterpret_cast<unsigned char *>(E + L.oCnt);                                                                       \
    float *M = E + L.oB, *con = E + L.oB;                                                                                    \
    /* kinematics scratch inside region B (dead before the solver writes M there) */                                        \
    float *poseL = E + L.oB, *recL = poseL + 12 * m.nlink, *qposL = recL + 12 * m.nlink, *qvelL = qposL + G; \
    float *poly = lds + L.oPoly;                                                                                             \
    const bool isdof = c < nv;                                                                                               \
    (void)rD; (void)rAref; (void)rJar; (void)rJv; (void)rDw; (void)kLin; (void)kAnc; (void)lk; (void)pcnt; (void)M; (void)con; \
    (void)poseL; (void)recL; (void)qvelL; (void)poly; (void)isdof; (void)e; (void)in_range;
    int bad_acc = 0, trips_acc = 0;
    (void)q_rlo;
    // every byte counts: cfg4's 20.6 KB per workgroup were 144 B above an eighth of the CU's LDS (7 instead of 8 workgroups per CU)
    __shared__ signed char sParent[32];
    int *sMask = reinterpret_cast<int *>(lds + L.oLinkTab);
    float *sMass = lds + L.oLinkTab + m.nlink;
    __shared__ unsigned short sItems[64 * (64 / G)];   // narrowphase work items: at most 64 per env, processed 64 at a time
    __shared__ unsigned char sMpr[64];
#ifdef HSR_PHASE_TIMING
    __shared__ int sDbg[4];
    if (tid0 < 4) sDbg[tid0] = 0;
#endif
    if (tid0 < nv) sParent[tid0] = (signed char)m.dof_parent[tid0];
    if (tid0 < m.nlink && tid0 < NLMAX) { sMask[tid0] = m.link_dofmask[tid0]; sMass[tid0] = m.link_mass[tid0]; }
    __shared__ unsigned char sDofLink[32];
    __shared__ int sEnv[64 / G];                       // env index of every lane group of this workgroup (wave packing: DevState::slot_env)
    __shared__ int sTick[64 / G];                      // substeps every env of this workgroup has run since its batch was created (DevState::tick): stamps of the separation margins
    if (tid0 < nv) sDofLink[tid0] = (unsigned char)m.dof_link[tid0];
    // the area of the per-link kinematic constants (kin2.h) is free in the instances that know their tree at compile time (kin3.h): it holds the hull
    // vertices of the deepest links - the finger hulls that the hard envs run MPR on - so that a support scan is an LDS round trip, not a global one
    constexpr bool KIN3_INST = Kin3Of<MT>::type::ok && EXACT;
    const float4 *ldsv = reinterpret_cast<const float4 *>(lds + L.oKin);
    if constexpr (KIN3_INST) { for (int i = tid0; i < m.nldsv; i += 64) reinterpret_cast<float4 *>(lds + L.oKin)[i] = m.mesh_vert4[m.ldsv_src[i]]; }
    else if (tid0 < m.nlink) kin2_store(m, tid0, lds + L.oKin + KIN2_FLOATS * tid0);
    // geom cache: constants of every geom, placements of the static ones (world link: identity pose)
    for (int gi = tid0; gi < m.ngeom; gi += 64) {
        if constexpr (!TG) { geom_consts_store(m.geom_rec + 32 * gi, lds + L.oGeomC + 8 * gi); if (KIN3_INST && m.geom_ldsv[gi] >= 0) hull_lds_patch(lds + L.oGeomC + 8 * gi, m.geom_ldsv[gi]); }
        if (gi < m.nstatic_geom) {
            m3 I3;
#pragma unroll
            for (int k = 0; k < 9; k++) I3.a[k] = (k % 4 == 0) ? 1.f : 0.f;
            geom_place3(geom_place_consts(m.geom_rec + 32 * gi), I3, mk3(0, 0, 0), lds + L.oGeomS + 16 * gi);
        }
    }
    // sphere-cull records of the candidate pairs (kin2.h: pair_pack)
    if constexpr (!TG) for (int p = tid0; p < ((m.npair_pad + 7) & ~7); p += 64) {
        unsigned pk = 0;
        if (p < m.npair) {
            const float4 a = reinterpret_cast<const float4 *>(m.pair_geo)[2 * p], b = reinterpret_cast<const float4 *>(m.pair_geo)[2 * p + 1];
            const int code = (int)a.x;
            pk = pair_pack(code & 255, (int)a.y, (int)b.x, (code >> 8) ? a.w : a.z + a.w);
        }
        reinterpret_cast<unsigned *>(lds + L.oPair)[p] = pk;
    }

    float qpos_c = 0, qvel_c = 0, warm_c = 0;          // qpos_c: lane = qpos index; qvel_c / warm_c: lane = dof
    bool done;
    v3 goal;
    float time_e;
    int nsteps_e = 0;
    // per-lane model constants of the solver: loaded once per launch (the registers are there since the exact-nv build)
    float my_ctrl = 0, damp_c = 0;
    int my_type = -1, my_qadr = 0, my_quat_lane = -1, my_limited = 0, my_act = -1;
    float act_p[6] = {0, 0, 0, 0, 0, 0};
    // the dof tree as this lane sees it, in two registers (SOLVE_LANE_TABLES, solve_body.inc): bit l of sub_c = link l moves with dof c; anc_c = the
    // dofs from c up to the root, 6 bits each (dof + 1, 0 ends the list) - so that the inertia rows walk registers, not a chain of LDS reads
    unsigned sub_c = 0;
    unsigned long long anc_c = 0;
    {
        PERSIST_LANE_VIEW(tid0)
        if (isdof) {
            for (int l = 1; l < m.nlink && l < 32; l++) sub_c |= ((unsigned)(m.link_dofmask[l] >> c) & 1u) << l;
            int sh = 0;
            for (int k = c; k >= 0 && sh < 60; k = m.dof_parent[k], sh += 6) anc_c |= (unsigned long long)(k + 1) << sh;
            my_type = m.dof_type[c]; my_qadr = m.dof_qposadr[c]; my_limited = m.dof_limited[c]; my_act = m.dof_act[c];
            my_quat_lane = m.link_dofadr[m.dof_link[c]] + 3;
            damp_c = m.dof_damping[c];
            if (my_act >= 0) {
                act_p[0] = m.act_kp[my_act]; act_p[1] = m.act_gear[my_act];
                act_p[2] = m.act_ctrlrange[2 * my_act]; act_p[3] = m.act_ctrlrange[2 * my_act + 1];
                act_p[4] = m.act_forcerange[2 * my_act]; act_p[5] = m.act_forcerange[2 * my_act + 1];
            }
        }
    }
    const int goal_link = (goal_body >= 0 && !m.body_mocap[goal_body]) ? m.body_link[goal_body] : -1;
    const v3 goal_off = goal_body >= 0 ? ld3(m.body_pos, goal_body) : mk3(0, 0, 0);
    PHASE_T0();

  constexpr bool HAS_SOLO = SV && EXACT && G == 16 && !std::is_same<MT, DevModel>::value;      // server instances of the reference configurations
  // one run of substeps [sub0, sub1) of a task (the envs of one lane-group set) - or, SOLO, of the single env solo_env in lane group 0
  auto run_task = [&](auto solo_c, const int sub0, const int sub1, const int solo_env) {
    constexpr bool SOLO = decltype(solo_c)::value;
    constexpr bool REP = SOLO;
    (void)solo_env;
    const bool first_run = sub0 == 0, last_run = sub1 >= n_substeps;
    const bool fresh = io.ctrl != nullptr && first_run;       // HSREnv.step begins here: ctrl[:] = action, a fresh done flag (hsr/env.py:116,124)
    cap_con = cap_row = cap_item = nsub_run = own_trips = 0; bad_acc = trips_acc = 0; nsteps_e = 0;
    int run_trips = 0;                                     // Newton iterations of this lane's env over this run (hand-over to a solo server)
    // Item list of the narrowphase, kept over several substeps (HSR_SKIN > 0): it is built from culls that ask "closer than HSR_SKIN"
    // instead of "touching", and holds until some geom of some env of the workgroup may have moved half of that since (upper bounds of
    // the moves, the ones the separation margins of the convex pairs use).  A pair that is not on the list was farther apart than the
    // skin when the list was built and has closed in by less since: no contact.  A pair on it runs its narrowphase every substep -
    // which reports a contact only where there is one - so the contact set is the one of culling every substep.
    int nitems = 0;                                        // wave-uniform
    int trunc_mask = 0;                                    // wave-uniform: bit j = the list holds only the first 64 items of env j (counted in capstat[2] on every substep the list is used)
    bool list_ok = false;                                  // wave-uniform: sItems[0 .. nitems) is a valid list
    float acc_move = 0.f;                                  // this lane's env: bound on the move of any of its geoms since the list was built
    // ---------------- load the env state; it lives in registers for the whole run of substeps
    {
        PERSIST_LANE_VIEW(tid0)
        if (c == 0) { sEnv[g] = e; sTick[g] = s.tick[e]; }
        done = !in_range || (fresh ? false : s.done[e] != 0);
        qpos_c = 0; qvel_c = 0; warm_c = 0;
        if (c < nq) qpos_c = s.qpos[(size_t)c * N + e];
        if (isdof) { qvel_c = s.qvel[(size_t)c * N + e]; warm_c = s.warm[(size_t)c * N + e]; }
        goal = mk3(s.mocap[e], s.mocap[N + e], s.mocap[2 * N + e]);
        time_e = s.time[e];
        if (isdof && my_act >= 0) {
            // with the caller's arrays at hand (io.ctrl) the kernel reads the action itself instead of a k_begin_step launch in front of it
            if (fresh) { my_ctrl = in_range ? io.ctrl[(size_t)e * m.nu + my_act] : 0.f; if (in_range) s.ctrl[(size_t)my_act * N + e] = my_ctrl; }
            else my_ctrl = s.ctrl[(size_t)my_act * N + e];
        }
    }
    wave_sync();

    for (int sub = sub0; sub < sub1; sub++) {
        // a workgroup that carries a hard env (many Newton iterations per substep so far) sets the pace of the launch: it
        // gets issue priority over the wave it shares its SIMD with, which has slack
        if (sub - sub0 >= 4 && 2 * trips_acc > 5 * (sub - sub0)) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);      // A/B on the bench: +4 %
        int tid_l = tid0;
        asm volatile("" : "+v"(tid_l));
        PERSIST_LANE_VIEW(tid_l)
        const bool valid = !done;
        if (!wave_any(valid)) break;
        if (valid && c == 0) sTick[g] += 1;      // read by the convex-pair section of ANY lane group, several wave_syncs further down
        int bad = 0;                 // per substep; only a live env's flag is kept
        asm volatile("" ::: "memory");   // model constants are re-read (L2 hits) every substep instead of living in - and spilling from - registers
        // placement constants of this lane's moving geom: fetched now, consumed after the kinematics (the L2 latency hides behind it)
        const int nstat = m.nstatic_geom, nmov = m.ngeom - nstat;
        const GeomPlaceC gpc = geom_place_consts(m.geom_rec + 32 * (nstat + (c < nmov ? c : 0)));
        // ---------------- K: kinematics + RNE recursion (kin2.h): lane = link / dof, only the parent-dependent part per tree level
        qposL[c] = qpos_c; qvelL[c] = qvel_c;
        wave_sync();
        // KIN3: the instance knows its kinematic tree at compile time (kin3.h): poses, velocities, inertia rows and bias force in one straight-line pass
        typedef typename Kin3Of<MT>::type KD;
        constexpr bool KIN3 = KD::ok && EXACT;
        if constexpr (KIN3) {
#ifndef HSR_K3_NORUN
            kin3_run<KD, G, NK>(c, m.gravz, qposL, qvelL, poseL, recL, kAng, qvelL + G, m.link_com, m.link_inertia, m.link_mass);          // (link records where the geom placements go afterwards)
#endif
            wave_sync();
            PHASE_K(26);
        } else {
            float *dwL = qvelL + G;                              // the geom placements are written after the kinematics
            Kin2 kin;
            kin.stageA(lds + L.oKin, m.nlink, c, qposL, recL);
            wave_sync();
            PHASE_K(26);
            kin.stageB(m.maxdepth, recL, poseL);
            wave_sync();
            PHASE_K(27);
            kin.stageC(lds + L.oKin, sDofLink, nv, poseL, qvelL, kAng, kLin, kAnc, dwL);
            wave_sync();
            PHASE_K(28);
            kin.stageD((c < m.nlink && c < NLMAX) ? sMask[c] : 0, qvelL, dwL, recL);
            PHASE_K(29);
            kin.stageE(m.gravz, lk);
            wave_sync();
        }
        PHASE(24);
        PHASE(16);
        if (valid) qpos_c = qposL[c];                        // mj_kinematics normalises free-joint quaternions in place
        if (!(fabsf(qpos_c) <= 1e10f) || !(fabsf(qvel_c) <= 1e10f)) bad = 1;
        // a-3 goal test uses the xpos of this substep's forward pass
        bool reach = false;
        if (goal_body >= 0 || s.ngoal > 0) {
            reach = true;
            if (goal_body >= 0) {
                v3 bp = goal;
                if (goal_link >= 0) { m3 Rg; v3 pg; pose_load(poseL + 12 * goal_link, Rg, pg); bp = pg + mulmv(Rg, goal_off); }
                reach = norm(bp - goal) < geofence;
            }
            for (int k = 0; k < s.ngoal; k++) {
                v3 pt[2];
#pragma unroll
                for (int w = 0; w < 2; w++) {
                    const int body = w == 0 ? s.goal_a[k] : s.goal_b[k];
                    pt[w] = goal;
                    if (!m.body_mocap[body]) { m3 Rg; v3 pg; pose_load(poseL + 12 * m.body_link[body], Rg, pg); pt[w] = pg + mulmv(Rg, ld3(m.body_pos, body)); }
                }
                reach = reach && norm(pt[0] - pt[1]) < s.goal_d[k];
            }
        }
        // link poses of an env's last substep go to global memory (sim.data.get_body_xpos after step(), hsr/env.py:144)
        if (valid && (sub == n_substeps - 1 || reach)) {
            for (int i = c; i < 3 * m.nlink; i += G) s.xpos[(size_t)i * N + e] = poseL[12 * (i / 3) + 9 + i % 3];
            for (int i = c; i < 9 * m.nlink; i += G) s.xmat[(size_t)i * N + e] = poseL[12 * (i / 9) + i % 9];
            for (int i = c; i < 6 * m.nlink; i += G) { const int l = (i % (3 * m.nlink)) / 3; s.lvel[(size_t)i * N + e] = recL[12 * l + (i < 3 * m.nlink ? 0 : 3) + i % 3]; }   // link w, then v(origin)
        }
        PHASE_K(31);
        // ---------------- C: collision
        // G: lane = geom, world placement of every geom once per substep (LDS, next to the link poses)
        // 1: lane = candidate pair of its own env, bounding spheres; survivors -> workgroup candidate list (wave ballots)
        // 2: lane = candidate of ANY env, oriented-box culls; survivors -> work items
        // 3: lane (or 8-lane sub-group for MPR) = work item: the narrowphases of all envs of the workgroup side by side
        {
            float *gw = qvelL + G;
            const int oGw = (int)(gw - E);
            // placement of geom gi: static ones in the shared table, moving ones in this env's cache
            auto gaddr = [&](int gi, const float *Ei) -> const float * { return gi < nstat ? lds + L.oGeomS + 16 * gi : Ei + oGw + 16 * (gi - nstat); };
            const float *gcc;
            if constexpr (TG) gcc = s.geom_c; else gcc = lds + L.oGeomC;
            float mymove = 0.f;
            if (valid) {
                for (int gm = c; gm < nmov; gm += G) {
                    const GeomPlaceC k2 = gm < G ? gpc : geom_place_consts(m.geom_rec + 32 * (nstat + gm));
                    m3 Rl; v3 pl;
                    pose_load(poseL + 12 * k2.link, Rl, pl);
                    // the geom moved since the previous substep by at most h (|v(geom origin)| + |w| rbound), v and w being this
                    // substep's link velocities (qpos advanced with exactly this qvel); 25 % cover the curvature of the path
                    const float4 lv0 = kl4(recL + 12 * k2.link);
                    const v3 lw = mk3(lv0.x, lv0.y, lv0.z), lvo = mk3(lv0.w, recL[12 * k2.link + 4], recL[12 * k2.link + 5]);
                    const v3 vg = lvo + cross(lw, mulmv(Rl, mk3(k2.q1.x, k2.q1.y, k2.q1.z)));
                    const float smove = 1.25f * m.timestep * (norm(vg) + norm(lw) * k2.q1.w) + 1e-7f;
                    geom_place3(k2, Rl, pl, gw + 16 * gm, smove);
                    mymove = fmaxf(mymove, smove);
                }
            }
            acc_move += __int_as_float(gmax<G>(__float_as_int(mymove)));      // non-negative floats order like their bit patterns
            const float skin = hook_no_item_list ? 0.f : HSR_SKIN;
            const bool rebuild = skin <= 0.f || !list_ok || wave_any(valid && 2.f * acc_move >= skin);
            for (int p0 = 0; p0 < m.npair_pad / 4; p0 += G) { const int p = p0 + c; if (p < m.npair_pad / 4) reinterpret_cast<int *>(pcnt)[p] = 0; }
            wave_sync();
            PHASE_K(30);
            const float4 *pg4 = reinterpret_cast<const float4 *>(m.pair_geo);
            const unsigned *sPair;
            if constexpr (TG) sPair = s.pair_pack; else sPair = reinterpret_cast<const unsigned *>(lds + L.oPair);
            unsigned short *sCand = reinterpret_cast<unsigned short *>(poly);      // 128 entries in the box-box polygon scratch (dead during the culls)
            if (rebuild) {
            int ncand = 0;                                         // wave-uniform
            nitems = 0;
            int nit_env[EPB];                                      // items per env (wave-uniform): an env keeps at most 64, whatever its neighbours do
#pragma unroll
            for (int j = 0; j < EPB; j++) nit_env[j] = 0;
            // level 2 over the first min(ncand, 64) candidates of the list; the rest moves to the front
            auto box_round = [&]() {
                const int take = ncand < 64 ? ncand : 64;
                bool pass = false;
                int it = 0;
                if (tid < take) {
                    it = sCand[tid];
                    const float *Ei = lds + (size_t)(it >> 14) * L.envf;
                    const unsigned pk = sPair[it & 0x3fff];
                    const int g1 = pk_g1(pk), g2 = pk_g2(pk);
                    float rb1, rb2;
                    const Geom A = geom_cached3(gaddr(g1, Ei), gcc + 8 * g1, m.mesh_vert4, rb1, ldsv), B = geom_cached3(gaddr(g2, Ei), gcc + 8 * g2, m.mesh_vert4, rb2, ldsv);
                    pass = pair_cull_box_nb(A, B, rb1, rb2, skin);
                }
                PHASE_S(2, 28);
                const unsigned long long lower = (1ull << tid) - 1ull;
                const int ig = it >> 14;
                bool accept = false;
#pragma unroll
                for (int j = 0; j < EPB; j++) {
                    const unsigned long long bj = __ballot(pass && ig == j);
                    if (pass && ig == j) accept = nit_env[j] + __popcll(bj & lower) < 64;
                    nit_env[j] += __popcll(bj);
                }
                const unsigned long long bal = __ballot(accept);
                if (accept) sItems[nitems + __popcll(bal & lower)] = (unsigned short)it;
                nitems += __popcll(bal);
                const int rest = ncand - take;
                unsigned short mv = 0;
                if (tid < rest) mv = sCand[take + tid];
                wave_sync();
                if (tid < rest) sCand[tid] = mv;
                wave_sync();
                ncand = rest;
                PHASE_S(2, 29);
            };
            // level 1: lane c of every env tests eight consecutive pairs per pass (their packed records are two LDS reads, the
            // positions / plane normals 24 more, all issued before the first test); the survivors of all envs are appended to the
            // workgroup's candidate list, one result bit at a time (wave ballots), and the list is box-culled whenever 64
            // candidates are pending and once at the end - a single call site
            {
                const int npair8 = (m.npair_pad + 7) & ~7;
                int pb = 0, u = 8, p0 = 0;
                unsigned mybits = 0;
                for (;;) {
                    if (u == 8 && pb < m.npair) {
                        p0 = pb + 8 * c;
                        uint4 k0 = make_uint4(0, 0, 0, 0), k1 = k0;
                        if (p0 < npair8) { k0 = *reinterpret_cast<const uint4 *>(sPair + p0); k1 = *reinterpret_cast<const uint4 *>(sPair + p0 + 4); }
                        const unsigned pk[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
                        float4 a1[8], a2[8], nn[8];
#pragma unroll
                        for (int q = 0; q < 8; q++) { const float *w1 = gaddr(pk_g1(pk[q]), E), *w2 = gaddr(pk_g2(pk[q]), E); a1[q] = kl4(w1); a2[q] = kl4(w2); nn[q] = kl4(w1 + 12); }
                        mybits = 0;
#pragma unroll
                        for (int q = 0; q < 8; q++) {
                            const float rad = __uint_as_float(pk[q] & 0xffff0000u) + skin;
                            const v3 r = mk3(a2[q].x - a1[q].x, a2[q].y - a1[q].y, a2[q].z - a1[q].z);
                            const float dpl = dot(r, mk3(nn[q].x, nn[q].y, nn[q].z)), dsq = dot(r, r);
                            const bool hit = pk_fn(pk[q]) <= FN_PLANE_CONVEX ? dpl <= rad : dsq <= rad * rad;
                            mybits |= (valid && (!REP || g == 0) && p0 + q < m.npair && hit) ? 1u << q : 0u;
                        }
                        u = 0; pb += 8 * G;
                        PHASE_S(2, 26);
                    }
                    while (u < 8 && ncand < 64) {
                        const bool pass = (mybits >> u) & 1u;
                        const unsigned long long bal = __ballot(pass);
                        if (pass) sCand[ncand + __popcll(bal & ((1ull << tid) - 1ull))] = (unsigned short)((g << 14) | (p0 + u));
                        ncand += __popcll(bal);
                        u++;
                    }
                    PHASE_S(2, 27);
                    const bool more = u < 8 || pb < m.npair;
                    if (ncand >= 64 || (!more && ncand > 0)) { wave_sync(); box_round(); }
                    else if (!more) break;
                }
            }
            wave_sync();
            trunc_mask = 0;
#pragma unroll
            for (int j = 0; j < EPB; j++) if (nit_env[j] > 64) trunc_mask |= 1 << j;      // more than 64 surviving pairs in ONE env: its contacts beyond them are dropped (a capacity event; NOT a diverged state)
            acc_move = 0.f; list_ok = true;
            }
            if (trunc_mask && ((trunc_mask >> g) & 1) && valid) cap_item++;      // capstat[2]: (env, substep) pairs that ran on a truncated list - every substep the list is used, not only the one that built it
            PHASE(19);
            DBGCNT(2, nitems);
            wave_sync();
            for (int ib = 0; ib < nitems; ib += 64) {
                // every section rebuilds the geoms it needs from the placement cache, so that nothing but the item id
                // stays live across the register-hungry narrowphases
                const bool act = ib + tid < nitems;
                const int it = act ? sItems[ib + tid] : 0;
                const int fn = act ? pk_fn(sPair[it & 0x3fff]) : -1;
                auto item_geoms = [&](int item, Geom &A, Geom &B, ContactOut &o, unsigned char *&cntp) {
                    const int ig = item >> 14, p = item & 0x3fff;
                    float *Ei = lds + (size_t)ig * L.envf;
                    const unsigned pk = sPair[p];
                    const float4 pb = pg4[2 * p + 1];
                    float rb;
                    A = geom_cached3(gaddr(pk_g1(pk), Ei), gcc + 8 * pk_g1(pk), m.mesh_vert4, rb, ldsv);
                    B = geom_cached3(gaddr(pk_g2(pk), Ei), gcc + 8 * pk_g2(pk), m.mesh_vert4, rb, ldsv);
                    o.con = s.con + (size_t)sEnv[ig] * m.nslot * 8; o.slot = (int)pb.y; o.maxcnt = (int)pb.z; o.cnt = 0;
                    cntp = reinterpret_cast<unsigned char *>(Ei + L.oCnt) + p;
                };
                PHASE(20);
                if (fn == FN_PLANE_BOX || fn == FN_PLANE_CONVEX) {
                    Geom G1, G2; ContactOut out; unsigned char *cntp;
                    item_geoms(it, G1, G2, out, cntp);
                    if (fn == FN_PLANE_BOX) collide_plane_box(G1, G2, out); else collide_plane_convex(G1, G2, out);
                    *cntp = (unsigned char)out.cnt;
                }
                PHASE(21);
                // convex pairs (mesh / cylinder against box / mesh)
                // Pass 1, lane = item: temporal coherence.  The pair keeps the direction d that separated it last time and how much of that
                // separation is left after the geoms' moves since (upper bounds, geom cache slot 15): while it is positive the pair is
                // still separated along d and that is all there is to do - one lane, no geoms, every such item of the wave at once.
                // The margin is only worth anything if it was brought up to date on the env's PREVIOUS substep: a pair that was off the
                // item list in between has moved by amounts nobody subtracted - its stamp (the env's substep count at the last visit)
                // then differs from tick - 1 and the margin counts as 0.
                // Pass 2, one MW-lane sub-group per item that needs its hulls (the lanes share the scans of the support function): both
                // hulls are scanned along d and the margin is refreshed; MPR runs only when d no longer separates - from the portal of the
                // previous substep if the pair penetrated then (margin row = -1: rows 0-2 hold its vertex ids).
                {
                    constexpr int MW = 8;          // lanes per convex-pair sub-group (4: -3 %, 16: -1 %)
                    const bool cv = fn == FN_CONVEX;
                    float cdx = 0.f, cdy = 0.f, cdz = 0.f, cmg = 0.f;
                    int cfresh = 0;
                    bool need = false;
                    if (cv) {
                        const int p1 = it & 0x3fff, ig1 = it >> 14;
                        float *sx1 = s.sepax + (size_t)(4 * p1) * N + sEnv[ig1];
                        int *stamp1 = s.septick + (size_t)p1 * N + sEnv[ig1];
                        cdx = sx1[0]; cdy = sx1[N]; cdz = sx1[2 * (size_t)N]; cmg = sx1[3 * (size_t)N];
                        const int tick1 = sTick[ig1];
                        cfresh = (*stamp1 == tick1 - 1 || hook_ignore_stamps) ? 1 : 0;
                        need = true;
                        if (cmg >= 0.f && (cdx != 0.f || cdy != 0.f || cdz != 0.f)) {
                            const unsigned pk1 = sPair[p1];
                            const float *Ei1 = lds + (size_t)ig1 * L.envf;
                            const float left = (cfresh ? cmg : 0.f) - (gaddr(pk_g1(pk1), Ei1)[15] + gaddr(pk_g2(pk1), Ei1)[15]);
                            if (left > 0.f) { sx1[3 * (size_t)N] = left; *stamp1 = tick1; need = false; }      // still separated along d (its contact count stays 0)
                        }
#ifdef HSR_PAIR_HIST
                        atomicAdd(&s.phase_cyc[32 + 40 * 4096 + p1], 1ull);       // tools/exp_pairs.py: convex items per pair
#endif
                    }
                    const unsigned long long cbal = __ballot(need);
                    if (need) sMpr[__popcll(cbal & ((1ull << tid) - 1ull))] = (unsigned char)tid;
                    const int ncv = __popcll(cbal);
                    wave_sync();
                    PHASE_M(26);
                    for (int r0 = 0; r0 < ncv; r0 += 64 / MW) {
                        const int k = r0 + tid / MW;
                        // what pass 1 read for the item, out of the registers of the lane that read it
                        const int srcl = 4 * (int)sMpr[k < ncv ? k : 0];
                        v3 d = mk3(__int_as_float(__builtin_amdgcn_ds_bpermute(srcl, __float_as_int(cdx))), __int_as_float(__builtin_amdgcn_ds_bpermute(srcl, __float_as_int(cdy))),
                                   __int_as_float(__builtin_amdgcn_ds_bpermute(srcl, __float_as_int(cdz))));
                        const float mg0 = __int_as_float(__builtin_amdgcn_ds_bpermute(srcl, __float_as_int(cmg)));
                        const bool fresh_cache = __builtin_amdgcn_ds_bpermute(srcl, cfresh) != 0;
                        if (k < ncv) {
                            const int it2 = sItems[ib + sMpr[k]];
                            Geom H1, H2; ContactOut o2; unsigned char *cntp;
                            item_geoms(it2, H1, H2, o2, cntp);
                            float *sx = s.sepax + (size_t)(4 * (it2 & 0x3fff)) * N + sEnv[it2 >> 14];
                            int *stampp = s.septick + (size_t)(it2 & 0x3fff) * N + sEnv[it2 >> 14];
                            const int tick_now = sTick[it2 >> 14];
                            float mg = 0.f;
                            int wid[3] = {0, 0, 0};
                            // a pair that penetrated on the previous substep left the vertex ids of its final portal instead of a
                            // separating direction: the portal warm start of mpr_penetration
                            if (mg0 < 0.f) {
                                if (fresh_cache && mpr_warm) { wid[0] = __float_as_int(d.x); wid[1] = __float_as_int(d.y); wid[2] = __float_as_int(d.z); }
                                d = mk3(0, 0, 0);
                            }
                            const bool have = d.x != 0.f || d.y != 0.f || d.z != 0.f;
                            bool still = false;                   // pass 1 found no margin left
                            PHASE_M(27);
                            if (have) {
                                const float gap = -dot(support<MW>(H1, d) - support<MW>(H2, -d), d);
                                still = gap > 1e-7f;
                                mg = still ? 0.98f * gap : 0.f;
                            }
                            PHASE_M(28);
                            if (!still) {
                                float depth; v3 dir, pos, sep;
                                int nsup = 0;
                                const bool hit = mpr_penetration<MW>(H1, H2, m.mpr_tolerance, m.mpr_iterations, depth, dir, pos, sep, nsup, wid);
#ifdef HSR_PHASE_TIMING
                                if ((tid & (MW - 1)) == 0) { atomicAdd(&sDbg[0], nsup); atomicAdd(&sDbg[1], 1); atomicMax(&sDbg[2], nsup); }
#endif
#ifdef HSR_PAIR_HIST
                                if ((tid & (MW - 1)) == 0) atomicAdd(&s.phase_cyc[32 + 40 * 4096 + 512 + (it2 & 0x3fff)], 1ull);
#endif
                                PHASE_M(29);
                                mg = 0.f;
                                if ((tid & (MW - 1)) == 0) {
                                    if (hit) { o2.add(pos, dir, -depth); sep = mk3(0, 0, 0); }
                                    if (hit && wid[0] > 0 && mpr_warm) sep = mk3(__int_as_float(wid[0]), __int_as_float(wid[1]), __int_as_float(wid[2]));
                                    sx[0] = sep.x; sx[N] = sep.y; sx[2 * (size_t)N] = sep.z;
                                }
                                if (hit && wid[0] > 0 && mpr_warm) mg = -1.f;
                            }
                            if ((tid & (MW - 1)) == 0) { sx[3 * (size_t)N] = mg; *stampp = tick_now; }
                            if ((tid & (MW - 1)) == 0) *cntp = (unsigned char)o2.cnt;
                            PHASE_M(30);
                        }
                    }
                }
                PHASE(22);
                // box-box: one work item per 8-lane sub-group as well (edge axes and polygon vertices spread over the lanes)
                {
                    const bool bb = fn == FN_BOX_BOX;
                    const unsigned long long bbal = __ballot(bb);
                    wave_sync();                                 // sMpr is reused as the box-box item list
                    if (bb) sMpr[__popcll(bbal & ((1ull << tid) - 1ull))] = (unsigned char)tid;
                    const int nbb = __popcll(bbal);
                    wave_sync();
                    PHASE_S(3, 26);
                    for (int r0 = 0; r0 < nbb; r0 += 8) {
                        const int k = r0 + tid / 8;
                        if (k < nbb) {
                            Geom G1, G2; ContactOut out; unsigned char *cntp;
                            item_geoms(sItems[ib + sMpr[k]], G1, G2, out, cntp);
                            PHASE_S(3, 27);
                            const int cnt = collide_box_box_w8(G1, G2, out.con, out.slot, out.maxcnt, poly + 24 * (tid / 8), [&](int idx) { (void)idx; if (idx == 28) { PHASE_S(3, 28); } else if (idx == 29) { PHASE_S(3, 29); } else if (idx == 30) { PHASE_S(3, 30); } else { PHASE_S(3, 31); } });
                            if ((tid & 7) == 0) *cntp = (unsigned char)cnt;
                        }
                    }
                }
                PHASE(23);
            }
        }
        __threadfence_block();       // contact records written to global by other lanes of this workgroup
        __syncthreads();
        if constexpr (REP) {         // the narrowphase left the contact counts with group 0: every replica gets its copy
            if (g > 0) for (int i = c; i < m.npair_pad / 4; i += G) reinterpret_cast<int *>(pcnt)[i] = reinterpret_cast<const int *>(lds + L.oCnt)[i];
            wave_sync();
        }
        if (dbg_store && valid) for (int p = c; p < m.npair; p += G) s.ncon_pair[(size_t)e * m.npair_pad + p] = pcnt[p];
        PHASE(17);
        // ---------------- S: dynamics + constraint solve + Euler (shared body)
        // per-dof view of qpos (scalar joints: their own coordinate; free joints: lin dofs their coordinate)
        float my_q = 0;
        q4 quat0; quat0.w = 1; quat0.x = quat0.y = quat0.z = 0;
        if (isdof) {
            my_q = qposL[my_qadr];
            if (c == my_quat_lane && my_type == DOF_FREE_ANG) { quat0.w = qposL[my_qadr]; quat0.x = qposL[my_qadr + 1]; quat0.y = qposL[my_qadr + 2]; quat0.z = qposL[my_qadr + 3]; }
        }
        wave_sync();             // qposL / link poses are read; region B may now be reused by the solver
        // per-body Hessian accumulators of the solver: the box-box polygon scratch is dead from here to the next substep's collision
        // body b's columns are static tiles counted from nv = NK; with a single block (G = 16) the per-contact assembly is as
        // fast (measured: -DHSR_FB_ALL), so it stays
        const int nfb = (EXACT && G == 32) ? m.nfb : 0;
        float *fbK = poly + (size_t)g * 28 * nfb;
        // per-link velocity fields of J v: the rows of the reference-acceleration / residual arrays (both live in the contact lanes' registers
        // since round 3; 2 R floats) - three vectors side by side in the setup, one in the Newton loop.  (Until round 4 the scratch was the
        // pair-count bytes: two link slots at cfg2's 47 pairs, so every robot <-> block contact there fell back to the per-contact products.)
        float *lvbuf = rAref;
        const int lvcap = (2 * R) / 6 < 16 ? (2 * R) / 6 : 16;          // link slots of a single J v (one is the world's); the setup's three at once: lvcap3
        {
#define SOLVE_STORE_DIAG dbg_store
#define SOLVE_COUNT_CAPS 1
#define PAIR_CNT8(p) (*reinterpret_cast<const unsigned long long *>(pcnt + (p)))
#define SOLVE_LANE_TABLES 1
#define SOLVE_LIMITS_FROM_MODEL 1
#include "solve_body.inc"
#undef SOLVE_LANE_TABLES
#undef SOLVE_LIMITS_FROM_MODEL
#undef PAIR_CNT8
#undef SOLVE_COUNT_CAPS
#undef SOLVE_STORE_DIAG
            // ---------------- integrate (a-2.7) in registers; qpos is redistributed through LDS (lane = qpos index)
            const float v1 = __shfl_down(vnew, 1, G), v2 = __shfl_down(vnew, 2, G);
            wave_sync();
            float *qn = rJv;             // any free row array: the solver is finished with it
            if (valid && isdof) {
                if (my_type == DOF_SLIDE || my_type == DOF_HINGE || my_type == DOF_FREE_LIN) qn[my_qadr] = my_q + h * vnew;
                else if (c == my_quat_lane) {
                    const v3 w = mk3(vnew, v1, v2);
                    const float wn = norm(w), angle = wn * h;
                    q4 q = quat0;
                    if (angle > 0) {
                        const v3 ax = w * (1.0f / wn);
                        float sn, cs;
                        fast_sincos(0.5f * angle, &sn, &cs);
                        q4 qr;
                        qr.w = cs; qr.x = ax.x * sn; qr.y = ax.y * sn; qr.z = ax.z * sn;
                        q = qnormalized(qmul(quat0, qr));
                    }
                    qn[my_qadr] = q.w; qn[my_qadr + 1] = q.x; qn[my_qadr + 2] = q.y; qn[my_qadr + 3] = q.z;
                }
            }
            wave_sync();
            if (valid) {
                if (c < nq) qpos_c = qn[c];
                qvel_c = vnew; warm_c = qacc_c;
                time_e += h; nsteps_e += 1; nsub_run += 1;
                if (bad) bad_acc = 1;
                if (reach) done = true;          // a-4: latch; the env skips the remaining substeps
            }
            if (wave_any(valid && reach)) list_ok = false;      // the items of an env that just finished leave the list
            trips_acc += newton_trips;
            if (valid && n_substeps - sub <= 100) own_trips += iter;
            if (valid) run_trips += iter;
            wave_sync();
            PHASE(18);
        }
    }
    // ---------------- write the state back (struct-of-arrays)
    {
    PERSIST_LANE_VIEW(tid0)
    const int was_done = in_range ? s.done[e] : 0;
    if (in_range && (fresh || !(was_done != 0 && nsteps_e == 0))) {
        if (c < nq) s.qpos[(size_t)c * N + e] = qpos_c;
        if (isdof) { s.qvel[(size_t)c * N + e] = qvel_c; s.warm[(size_t)c * N + e] = warm_c; }
        const float bsum = gsum<G>((float)bad_acc);
        if (c == 0 && (!REP || g == 0)) {
            if (cap_con) atomicAdd(&s.capstat[0], (unsigned long long)cap_con);
            if (cap_row) atomicAdd(&s.capstat[1], (unsigned long long)cap_row);
            if (cap_item) atomicAdd(&s.capstat[2], (unsigned long long)cap_item);
            atomicAdd(&s.capstat[3], (unsigned long long)nsub_run);
            // hardness of the env for the next packing: Newton iterations over the last (up to) 100 substeps of the env-step
            s.trips[e] = (sub0 <= n_substeps - 100 || first_run) ? own_trips : s.trips[e] + own_trips;
            s.tick[e] = sTick[g];
            s.time[e] = time_e;
            if (bsum > 0) s.bad[e] = 1;
            s.nsteps[e] = (fresh ? 0 : s.nsteps[e]) + nsteps_e;
            if (fresh) s.done[e] = done ? 1 : 0; else if (done) s.done[e] = 1;
        }
    }
    // a-5 observation = concat(qpos, qvel) (hsr/env.py:111-113), reward = float(success) (hsr/env.py:133), env-major, straight from
    // the registers, when the env-step is complete
    if (in_range && io.ctrl && last_run) {
        if (io.obs) {
            float *o = io.obs + (size_t)e * (nq + nv);
            if (c < nq) o[c] = qpos_c;
            if (isdof) o[nq + c] = qvel_c;
        }
        if (c == 0) {
            if (io.reward) io.reward[e] = done ? 1.f : 0.f;
            if (io.done) io.done[e] = done ? 1 : 0;
            if (io.nsteps) io.nsteps[e] = s.nsteps[e];
            ENV_STAMP(0, e, __builtin_amdgcn_s_memrealtime());
        }
    }
    // Hand-over of a hard env to a solo server (queued launches with servers only): an env that needed solo_trips or more Newton iterations
    // per substep over this round, is not finished and has enough substeps left leaves its task if a server is free - its state is in the
    // state arrays (written above), its slot of the task is emptied, and the server's ticket gets the env and the substep to go on from.
    if constexpr (!SOLO && HAS_SOLO) {
        if (s.q_chunk && s.solo_servers > 0 && !last_run) {
            const bool hard = in_range && !done && n_substeps - sub1 >= s.solo_min_left && 4 * run_trips >= s.solo_trips_x4 * (sub1 - sub0);
            if (wave_any(hard)) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // every lane: what it wrote of the state is out before the ticket is filled
                __builtin_amdgcn_wave_barrier();
                if (hard && c == 0) {
                    const int old = __hip_atomic_fetch_sub(&s.sq_ctl[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (old > 0) {
                        const int pos = __hip_atomic_fetch_add(&s.sq_ctl[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        s.slot_env[task * EPB + g] = -1;
                        ENV_STAMP(1, e, (unsigned long long)sub1 | (__builtin_amdgcn_s_memrealtime() << 16));
                        __hip_atomic_store(&s.sq_items[pos], e | (sub1 << 20), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    } else __hip_atomic_fetch_add(&s.sq_ctl[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
    }
  };      // run_task
  if (HAS_SOLO && s.q_chunk && s.solo_servers > 0 && (int)blockIdx.x < s.solo_servers) {
    // ---------------- solo server: no tasks; one hard env at a time, alone in this wave, from the substep its worker left it at to the
    // end of the env-step.  A server always holds exactly one ticket of the hand-over list; it leaves when every task has finished its
    // last round (no worker can hand anything over any more) and its ticket is still empty.
    if constexpr (HAS_SOLO) {
      const int T = (N + EPB - 1) / EPB;
      for (;;) {
        int item = -1;
        if (threadIdx.x == 0) {
            const int h = __hip_atomic_fetch_add(&s.sq_ctl[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            for (; h < s.sq_cap;) {          // (tickets never reach the capacity: one per hand-over - at most one per env - and one per server)
                item = __hip_atomic_load(&s.sq_items[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (item >= 0) break;
                if (__hip_atomic_load(&s.sq_ctl[3], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= T) {      // all tasks done: a last look, then leave
                    item = __hip_atomic_load(&s.sq_items[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(32);
                if (++spins > (1 << 22) || __hip_atomic_load(s.q_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                    __hip_atomic_store(s.q_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        item = __builtin_amdgcn_readfirstlane(item);
        if (item < 0) break;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        run_task(std::true_type{}, item >> 20, n_substeps, item & 0xfffff);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_wave_barrier();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(&s.sq_ctl[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // free again
      }
    }
  } else
  for (;;) {      // task loop (a single pass without the work queue)
    if (s.q_chunk) {
        if (!q_claim(s, (N + EPB - 1) / EPB, (n_substeps + s.q_chunk - 1) / s.q_chunk, q_rlo, task, q_round)) break;
    }
    const int sub0 = s.q_chunk ? q_round * s.q_chunk : 0, sub1 = s.q_chunk ? min(n_substeps, sub0 + s.q_chunk) : n_substeps;
    run_task(std::false_type{}, sub0, sub1, -1);
    if (!s.q_chunk) break;
    q_push(s, (N + EPB - 1) / EPB, task, q_round, sub1 >= n_substeps);
  }
    const int tid = tid0; (void)tid;
#ifdef HSR_PHASE_TIMING
    wave_sync();
    dc_[1] = (unsigned long long)sDbg[0] | ((unsigned long long)sDbg[1] << 24) | ((unsigned long long)sDbg[2] << 48);
#endif
    PHASE_FLUSH();
#undef PERSIST_LANE_VIEW
}
