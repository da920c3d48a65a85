/*
 * hsrsim.h - C-ABI of libhsrsim.so: the MI355X-native batched replacement for the slice of
 * mujoco_py that ethanabrooks/hsr-env touches on its hot path (HSREnv.step -> sim.step()).
 *
 * Every entry point cites the reference interface it replaces (paths relative to the reference
 * repository root).  All functions return 0 on success or a negative HSR_E* code; they never
 * throw across the ABI.  hsr_last_error() returns a thread-local message for the last failure.
 * Host arrays are caller-owned, row-major, float32 unless stated; "dev" variants take device
 * pointers (same layout) and do not synchronise the stream.
 *
 * A batch is NOT thread-safe; it owns one HIP stream and all device buffers (SoA [field][env]).
 */
#ifndef HSRSIM_H
#define HSRSIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HSR_OK 0
#define HSR_EINVAL (-1)   /* bad argument / shape (reference: AssertionError, hsr/mujoco_env.py:88-89) */
#define HSR_EBLOB (-2)    /* not a model blob (reference: IOError on a missing XML, hsr/mujoco_env.py:30-31) */
#define HSR_EDEVICE (-3)  /* HIP runtime failure / no GPU (reference: DependencyNotInstalled, hsr/mujoco_env.py:12-15) */
#define HSR_ENAME (-4)    /* unknown body / joint name */
#define HSR_EBADSTATE (-5) /* some env hit a non-finite state (reference: mujoco_py.MujocoException) */

typedef struct hsr_model hsr_model;
typedef struct hsr_batch hsr_batch;

/* indices for hsr_model_size() */
enum hsr_size { HSR_NQ = 0, HSR_NV, HSR_NU, HSR_NLINK, HSR_NBODY, HSR_NGEOM, HSR_NPAIR, HSR_NMESHVERT,
                HSR_NSLOT, HSR_NLIMIT, HSR_NCONMAX, HSR_NJMAX, HSR_NMOCAP };

const char *hsr_last_error(void);

/* ---- model: replaces mujoco_py.load_model_from_path (hsr/mujoco_env.py:33) on the XML produced by
 *      hsr/util.py:87-182; the blob is emitted offline by hsr_env_amd/compiler.py ---------------- */
int hsr_model_load(const void *blob, size_t len, hsr_model **out);
void hsr_model_destroy(hsr_model *m);
int hsr_model_size(const hsr_model *m, int which);                 /* sim.model.nq / nv / nu (hsr/mujoco_env.py:88-89) */
double hsr_model_timestep(const hsr_model *m);                     /* sim.model.opt.timestep (hsr/mujoco_env.py:98) */
int hsr_model_ctrlrange(const hsr_model *m, float *out /*[nu,2]*/); /* model.actuator_ctrlrange (hsr/mujoco_env.py:44) */
int hsr_model_qpos0(const hsr_model *m, float *out /*[nq]*/);       /* sim.data.qpos after MjSim() (hsr/mujoco_env.py:49) */
int hsr_model_body_id(const hsr_model *m, const char *name);       /* name lookup behind data.get_body_xpos (hsr/env.py:144,180,184) */
int hsr_model_joint_qpos_addr(const hsr_model *m, const char *name, int *start, int *end); /* model.get_joint_qpos_addr (hsr/env.py:153) */

/* ---- batch of N independent envs on one GPU: replaces N x mujoco_py.MjSim(model) (hsr/mujoco_env.py:34) */
int hsr_batch_create(const hsr_model *m, int n_envs, int device_id, hsr_batch **out);
void hsr_batch_destroy(hsr_batch *b);                              /* sim.__exit__ (hsr/env.py:209) */
int hsr_batch_size(const hsr_batch *b);
void *hsr_batch_stream(const hsr_batch *b);                        /* hipStream_t the batch launches on */
int hsr_batch_sync(hsr_batch *b);

/* sim.reset() + reset_model() writes (hsr/mujoco_env.py:83-85, hsr/env.py:158-177): for envs with
 * mask[e] != 0 (all if mask == NULL): qpos <- qpos0[e] (model qpos0 if NULL), qvel <- 0, ctrl <- 0,
 * warm start <- 0, time <- 0, mocap_pos <- mocap[e] (0 if NULL); then sim.forward(). */
int hsr_batch_reset(hsr_batch *b, const uint8_t *mask, const float *qpos0 /*[N,nq]*/, const float *mocap /*[N,3]*/);

/* same for device pointers, asynchronous; d_mask == NULL means "the envs whose done flag the last step
 * latched" - the batched form of `if done: env.reset()` (hsr/control.py:73-75) */
int hsr_batch_reset_dev(hsr_batch *b, const uint8_t *d_mask, const float *d_qpos0, const float *d_mocap);

/* sim.get_state() / sim.set_state(MjSimState(time,qpos,qvel,act,udd_state)) + forward
 * (hsr/mujoco_env.py:87-94, hsr/env.py:69,150,175); act and udd_state are empty for this model */
int hsr_batch_get_state(hsr_batch *b, float *time /*[N]*/, float *qpos /*[N,nq]*/, float *qvel /*[N,nv]*/);
int hsr_batch_set_state(hsr_batch *b, const float *time, const float *qpos, const float *qvel);
int hsr_batch_set_mocap(hsr_batch *b, const float *mocap /*[N,3]*/);   /* sim.data.mocap_pos[:] = ... (hsr/env.py:169) */
int hsr_batch_set_warmstart(hsr_batch *b, const float *qacc_warmstart /*[N,nv]*/);
int hsr_batch_get_warmstart(hsr_batch *b, float *qacc_warmstart /*[N,nv]*/);
int hsr_batch_forward(hsr_batch *b);                               /* sim.forward() (hsr/env.py:176) */

/* HSREnv.step (hsr/env.py:115-135) for all envs: ctrl[:] = action; up to n_substeps x { sim.step();
 * done = |xpos(goal_body) - mocap_pos| < geofence; break if done }; obs = concat(qpos, qvel);
 * reward = float(done).  goal_body < 0 reproduces `goals is None` (done stays 0).
 * Outputs may be NULL.  nsteps (optional) receives the substeps each env actually ran. */
int hsr_batch_step(hsr_batch *b, const float *ctrl /*[N,nu]*/, int n_substeps, int goal_body, float geofence,
                   float *obs /*[N,nq+nv]*/, float *reward /*[N]*/, uint8_t *done /*[N]*/, int32_t *nsteps /*[N]*/);
/* same with device pointers, asynchronous on hsr_batch_stream() */
int hsr_batch_step_dev(hsr_batch *b, const float *d_ctrl, int n_substeps, int goal_body, float geofence,
                       float *d_obs, float *d_reward, uint8_t *d_done, int32_t *d_nsteps);

/* Further goal terms, AND-ed per substep with the main term of hsr_batch_step* - `all(in_range(*g) for g in goals)`,
 * hsr/env.py:124-126,137-147: term k holds when |p(body_a[k]) - p(body_b[k])| < dist[k], p(body) = its xpos, or the env's mocap
 * point when the body is the mocap body.  n = 0..4 (0 clears); with goal_body < 0 in hsr_batch_step* the terms alone decide. */
int hsr_batch_set_goals(hsr_batch *b, int n, const int *body_a, const int *body_b, const float *dist);

/* sim.data.get_body_xpos(name) for every env (hsr/env.py:144,180,184); valid after forward/step */
int hsr_batch_body_xpos(hsr_batch *b, int body_id, float *out /*[N,3]*/);


/* the reference's 'openai' observation (hsr/env.py:72-110, restored to its evident intent), fused in
 * one kernel: out[N,25] = grip_pos 3 | object_pos 3 | object_rel_pos 3 | gripper_state 2 | object_rot 3 (mat2euler,
 * hsr/env.py:256-272) | object_velp 3 | object_velr 3 | grip_velp 3 | gripper_vel 2.
 * ids[7] = {finger body l, finger body r (hsr/env.py:59), object body (hsr/env.py:58), qpos address of the two finger
 * joints, dof address of the two finger joints}.  Positions / velocities are those of the last forward pass. */
int hsr_batch_obs_openai(hsr_batch *b, const int *ids, float *out /*[N,25]*/);
int hsr_batch_obs_openai_dev(hsr_batch *b, const int *ids, float *d_out);
/* per-env error flags (non-finite or |q| > 1e10), the batched form of MuJoCo's mj_checkPos/Vel */
int hsr_batch_bad_state(hsr_batch *b, uint8_t *out /*[N]*/);

/* ---- introspection used by the parity tests (stage-by-stage comparison with the oracle) ------ */
enum hsr_field { HSR_F_XPOS = 0 /*[N,nlink,3]*/, HSR_F_XMAT /*[N,nlink,9]*/, HSR_F_M /*[N,nv,nv]*/,
                 HSR_F_QACC /*[N,nv]*/, HSR_F_QACC_SMOOTH /*[N,nv]*/, HSR_F_QFRC_SMOOTH /*[N,nv]*/,
                 HSR_F_QFRC_CONSTRAINT /*[N,nv]*/, HSR_F_NCON /*[N] (as float)*/, HSR_F_NEFC /*[N]*/,
                 HSR_F_CONTACT /*[N,nslot,7] pos3 normal3 dist; dist=+1 marks an empty slot*/,
                 HSR_F_NITER /*[N]*/ };
int hsr_batch_get_field(hsr_batch *b, int field, float *out);
/* timing of the last hsr_batch_step*: total ms and per-kernel ms (HIP events on the batch stream);
 * enable with hsr_batch_set_profiling(b, 1).  kernel order: kinematics, collide, solve */
int hsr_batch_set_profiling(hsr_batch *b, int on);
int hsr_batch_last_timing(hsr_batch *b, float *total_ms, float *kernel_ms /*[3]*/, int *launches /*[3]*/);
/* hsr_batch_set_profiling(b, 2): log an event pair around EVERY launch of the persistent kernel without synchronising (what bench.py
 * times its roofline with).  hsr_batch_kernel_times synchronises, writes the durations (ms) of the launches logged since the last
 * call into out_ms[0..cap) and returns how many there were. */
int hsr_batch_kernel_times(hsr_batch *b, float *out_ms, int cap);
/* use a captured hipGraph for the substep loop of the per-substep-kernel path (default on) */
int hsr_batch_set_graph(hsr_batch *b, int on);
/* whole env-step in ONE persistent kernel (default on when the model fits: nv <= 32, LDS budget); returns the
 * resulting setting.  With it on, hsr_batch_last_timing() reports the persistent kernel in slot 2 (slots 0,1 = 0). */
int hsr_batch_set_persistent(hsr_batch *b, int on);
int hsr_batch_is_persistent(const hsr_batch *b);                  /* 0 = per-substep chain; else bit 0 set, bit 1: the kernel instance carries the model's scalars as
                                                                    * compile-time constants, bit 2: and its kinematic tree (straight-line kinematics / inertia) */
/* introspection after hsr_batch_step*: with it on, the persistent kernel also stores what hsr_batch_forward stores - the
 * per-pair contact counts (HSR_F_CONTACT), HSR_F_NCON / NEFC / NITER and HSR_F_QACC of every env's LAST substep (default off:
 * the fields then describe the last hsr_batch_forward).  Parity tests of the hot path's own narrowphase use it.
 * `on` is a bit mask: 1 = the above; 2 and 4 are test hooks that force rarely taken branches of the Newton solver of the persistent
 * kernel (2: J v per contact instead of per link; 4: every iteration takes the PSD-majorant Hessian) - same minimiser, other path;
 * 16 makes the convex-pair section trust a cached separation margin whatever its stamp - the round-2 behaviour, kept so that the test of
 * the stamps can show what they prevent; 64 makes the persistent kernel cull every substep instead of keeping its narrowphase item list
 * until a geom may have moved half a skin (the behaviour up to round 3; the contact sets are the same either way); 128 keeps the dense factorisation of the Newton Hessian where the sparse one applies (32-lane instances with several free bodies, envs whose contacts couple at most one body to the robot: same minimiser, fewer instructions); 32 raises the work queue's watchdog flag after every persistent launch (what a ticket that is
 * never served does): the flag is sticky on the device, the next synchronising call (hsr_batch_sync, _step, _kernel_times, _cap_counts,
 * _bad_state, _get_state) returns HSR_EDEVICE once and clears it. */
int hsr_batch_set_debug(hsr_batch *b, int on);
/* Solo servers of the persistent kernel (queued launches; the 16-lane instances of the reference configurations).  `servers` workgroups
 * of the launch take no tasks: they wait for envs that a worker has found hard - `trips` or more Newton iterations per substep over a
 * round of the work queue (0 keeps the current threshold) - and run each of them alone in a wave, from the substep its worker left it at to
 * the end of the env-step, while the task it came from goes on without it.  The launch ends with its slowest env's chain (hsr/env.py:118-131
 * is one serial loop per env); this shortens that chain.  0 servers = off.  A server's replicas sum the contact terms of the Hessian and
 * of J^T f in another order than a worker does, and which envs get a server depends on timing: with servers on, results are reproducible to
 * rounding (median |dobs| 1e-5 after an env-step, tests/test_gpu_hotpath.py::test_solo_servers_follow_the_plain_run), not bit for bit.  Launches of
 * 2048 or more substeps or of more than 2^20 envs run without servers (the hand-over ticket is one int).  Returns 1 (not an error) when the
 * model's kernel instance has no server path. */
int hsr_batch_set_solo(hsr_batch *b, int servers, float trips);
/* envs handed over to solo servers by the last persistent launch (synchronises) */
int hsr_batch_solo_handovers(hsr_batch *b, int *out);
/* wave packing of the persistent kernel (default on; HSR_SCHEDULE=0 turns it off at creation): before every launch the envs are re-distributed over the waves by the
 * Newton iterations they needed at the end of their previous launch (hard envs one per wave, with the easiest as neighbours).
 * A pure scheduling decision: every env's result is bit-identical with it on or off. */
int hsr_batch_set_schedule(hsr_batch *b, int on);
/* Convex pairs (mesh / cylinder vs box / mesh: MPR, libccd's ccdMPRPenetration behind mjc_Convex).  A pair that penetrates keeps the
 * vertex ids of the portal its run ended on; placed with the next substep's poses they are a portal again (checked: the origin ray
 * still crosses their triangle, else the run starts from scratch), so the refinement confirms the face it found last time with 2-3
 * support calls instead of rediscovering it with 8.  The refinement itself is libccd's.  Where the origin projects into the final
 * triangle the result (the face's plane) does not depend on how the run got there; where it does not (libccd then measures to the
 * triangle's edge) a warm-started run is repeated from scratch and nothing is kept for the next substep - the warm start is as close
 * to the cold-started fp64 oracle as a cold fp32 run is (tests/test_gpu_hotpath.py::test_pinched_block_contacts_follow_the_oracle).
 * Default on (HSR_MPR_WARM=0 at creation, or this call, turns it off). */
int hsr_batch_set_mpr_warm(hsr_batch *b, int on);
/* Work queue of the persistent kernel.  When a batch has more tasks (groups of 4 or 2 envs) than the GPU holds workgroups at once
 * (BASELINE configs 4 / 5: 4096 two-env tasks on 2048 resident workgroups), the env-step is cut into rounds of `chunk` substeps and
 * persistent workgroups take (task, round) tickets, lowest round first, so a task that lags is always resumed at once and no
 * second dispatch round starts a hard task late.  mode: -1 automatic (default; HSR_QUEUE overrides at creation), 0 off, 1 on whenever the env-step
 * has at least two rounds; chunk: substeps per round (0 keeps it; default 20, HSR_QUEUE_CHUNK).  Scheduling only: results are bit-identical. */
int hsr_batch_set_queue(hsr_batch *b, int mode, int chunk);
/* How often the device buffer caps bit since the last call (the counters are cleared): out[0] = (env, substep) pairs that
 * dropped contacts beyond nconmax, out[1] = dropped constraint rows beyond njmax, out[2] = dropped narrowphase work items
 * beyond 64 per env, out[3] = (env, substep) pairs executed.  MuJoCo's own caps are nconmax=100 njmax=500
 * (hsr/models/world.xml:44); the compiled models carry smaller ones. */
int hsr_batch_cap_counts(hsr_batch *b, unsigned long long *out /*[4]*/);
/* of the row-cap events above: how many rows beyond njmax the env would have needed - out[k] counts the events with
 * njmax + 8 k < rows wanted <= njmax + 8 (k + 1) (the last bin is open-ended); cleared by the call.  What the compiled njmax is sized by. */
int hsr_batch_cap_histogram(hsr_batch *b, unsigned long long *out /*[8]*/);
/* Newton iterations every env ran over the last (up to) 100 substeps of its previous env-step launch: the hardness measure
 * hsr_batch_set_schedule packs by (the solver's iteration count MuJoCo reports as mjData.solver_iter, summed). */
int hsr_batch_newton_trips(hsr_batch *b, int32_t *out /*[n_envs]*/);
/* the packing the last persistent launch ran with (k_schedule, hsrsim.hip): out[slot] = env that lane group `slot % (64 / lanes per env)` of task
 * `slot / (64 / lanes per env)` held, -1 = empty; ceil(n_envs / envs per wave) * envs per wave entries.  Contract (tests/test_gpu_hotpath.py): every env exactly
 * once; per chunk of 8192 envs the first lane group of task w holds the env with the w-th most iterations (ties: lower index), the other groups are
 * filled from the easy end.  Results never depend on it. */
int hsr_batch_packing(hsr_batch *b, int32_t *out /*[ceil(n_envs / epw) * epw]*/);

/* diagnostics (meaningful only in the -DHSR_PHASE_TIMING build, libhsrsim_timing.so; tools/phase_timing.py,
 * tools/block_times.py): per-phase cycle sums of the last launches, and per-workgroup
 * {start, end (s_memrealtime), HW_ID, XCC_ID, Newton trips, sphere-cull candidates, work items, rows} */
int hsr_batch_phase_cycles(hsr_batch *b, unsigned long long *out /*[32]*/);
int hsr_batch_block_times(hsr_batch *b, unsigned long long *out /*[nblocks,40]: 8 header words + 26 phase cycle sums*/, int nblocks);

#ifdef __cplusplus
}
#endif
#endif /* HSRSIM_H */
