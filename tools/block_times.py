"""Diagnostic: when does each workgroup of the persistent kernel start and end, and where (needs libhsrsim_timing.so)."""
import ctypes as C, sys, numpy as np
sys.path.insert(0, '.')
from hsr_env_amd.compiler import load_config
from hsr_env_amd import sim as hs
from bench import sample_inputs
import os
cfg = os.environ.get('HSR_CFG', 'cfg3')
m = load_config(cfg); n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
q0, goal = sample_inputs(m, n, 0, 0)
sim = hs.BatchSim(m, n); sim.set_graph(False)
if os.environ.get('HSR_LIFE_QUEUE') == '0':
    sim.set_queue(0)          # static assignment: a workgroup's stamps and counters are those of ONE task (cfg4 at 8192 envs would take the work queue)
sim.reset(qpos0=q0, mocap=goal)
rng = np.random.default_rng(1)
for k in range(3):
    ctrl = rng.uniform(m.act_ctrlrange[:, 0], m.act_ctrlrange[:, 1], (n, m.nu)).astype(np.float32)
    sim.step(ctrl, 300, m.body_id(m.block_body()) if m.block_body() else -1, 0.05)
epb = 4 if m.nv <= 16 else 2
nb = n // epb
L = sim._L
L.hsr_batch_block_times.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * (40 * nb))()
L.hsr_batch_block_times(sim._b, buf, nb)
a = np.array(list(buf), dtype=np.int64).reshape(nb, 40)
t0 = a[:, 0].min()
st = (a[:, 0] - t0) / 100.0; en = (a[:, 1] - t0) / 100.0       # us
print('blocks', nb, 'kernel span %.1f ms' % (en.max() / 1e3))
print('start  percentiles us (0,25,50,75,90,100):', np.percentile(st, [0, 25, 50, 75, 90, 100]).round(0))
print('end    percentiles us:', np.percentile(en, [0, 25, 50, 75, 90, 100]).round(0))
print('life   percentiles us:', np.percentile(en - st, [0, 25, 50, 75, 90, 100]).round(0))
print('blocks started within 100 us of kernel start:', int((st < 100).sum()))
hw = a[:, 2]; xcc = a[:, 3] & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; simd = (hw >> 4) & 3; wv = hw & 0xf
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
first = st < 100
print('distinct CUs', len(np.unique(cuid)), ' first-round blocks per CU: min %d max %d' % (np.bincount(cuid[first]).min(), np.bincount(cuid[first]).max()))
print('first-round waves per (CU, SIMD):', np.bincount(np.bincount(cuid[first] * 4 + simd[first])))
print('wave slot ids used:', np.unique(wv))

life = en - st
o = np.argsort(life)
print('newton trips / substep: mean %.2f, slowest 10 blocks %s, fastest 10 %s' % (a[:, 4].mean() / 300, (a[o[-10:], 4] / 300).round(2), (a[o[:10], 4] / 300).round(2)))
nsup = a[:, 5] & 0xffffff; ncall = (a[:, 5] >> 24) & 0xffffff; nmax = a[:, 5] >> 48
print('MPR runs / substep: mean %.2f, slowest 10 %s; supports per run: mean %.1f, slowest 10 %s; max supports in a run %d' % (
    ncall.mean() / 300, (ncall[o[-10:]] / 300).round(2), nsup.sum() / max(ncall.sum(), 1), (nsup[o[-10:]] / np.maximum(ncall[o[-10:]], 1)).round(1), nmax.max()))
print('narrowphase items / substep: mean %.2f, slowest 10 %s' % (a[:, 6].mean() / 300, (a[o[-10:], 6] / 300).round(2)))
if os.environ.get('HSR_LSCOUNT'):      # timing build with -DHSR_LSCOUNT: counter 3 = line-search evaluations | own Newton iterations << 32 of lane 0's env
    ev = a[:, 7] & 0xffffffff; itn = a[:, 7] >> 32
    print('line-search evaluations per Newton iteration (lane 0 env): all %.2f, slowest 20 %.2f (its iterations / substep %.2f), median 20 %.2f' % (
        ev.sum() / max(itn.sum(), 1), ev[o[-20:]].sum() / max(itn[o[-20:]].sum(), 1), itn[o[-20:]].mean() / 300, ev[o[nb // 2 - 10: nb // 2 + 10]].sum() / max(itn[o[nb // 2 - 10: nb // 2 + 10]].sum(), 1)))
if os.environ.get('HSR_NOSTEPCOUNT'):      # timing build with -DHSR_NOSTEPCOUNT
    ns_, tot_ = a[:, 7] & 0xffffffff, a[:, 7] >> 32
    for nm, idx in (('all', o), ('slowest 20', o[-20:]), ('median 20', o[nb // 2 - 10: nb // 2 + 10]), ('fastest 20', o[:20])):
        print('substeps whose first Newton iteration took no step in any env of the wave, %s: %.3f of %.1f substeps with an iteration' % (nm, ns_[idx].sum() / max(tot_[idx].sum(), 1), tot_[idx].mean()))
if os.environ.get('HSR_CPLSTAT'):      # timing build with -DHSR_CPLSTAT: counter 3 = substeps of lane 0's env by what its contacts couple
    c0, c1, c2 = a[:, 7] & 0x1fffff, (a[:, 7] >> 21) & 0x1fffff, (a[:, 7] >> 42) & 0x1fffff
    for nm, idx in (('all', o), ('slowest 20', o[-20:]), ('median 20', o[nb // 2 - 10: nb // 2 + 10])):
        tot = max((c0[idx] + c1[idx] + c2[idx]).sum(), 1)
        print('coupling of the contact list (lane 0 env), %s: no robot <-> body contact %.2f, one body coupled to the robot and no body <-> body %.2f, more %.2f' % (nm, c0[idx].sum() / tot, c1[idx].sum() / tot, c2[idx].sum() / tot))
if os.environ.get('HSR_ACTSTAT'):      # timing build with -DHSR_ACTSTAT: counter 3 = Newton trips of the wave by the number of envs still iterating
    c1, c2, c3 = a[:, 7] & 0x1fffff, (a[:, 7] >> 21) & 0x1fffff, (a[:, 7] >> 42) & 0x1fffff
    for nm, idx in (('all', o), ('slowest 20', o[-20:]), ('median 20', o[nb // 2 - 10: nb // 2 + 10])):
        tot = max((c1[idx] + c2[idx] + c3[idx]).sum(), 1)
        print('Newton trips of a wave by envs still iterating, %s: one %.2f, two %.2f, three or four %.2f (trips / substep %.2f)' % (nm, c1[idx].sum() / tot, c2[idx].sum() / tot, c3[idx].sum() / tot, tot / len(idx) / 300))
print('nefc sum / substep: mean %.2f, slowest 10 %s' % (a[:, 7].mean() / 300, (a[o[-10:], 7] / 300).round(2)))
print('corr(life, newton) %.3f  corr(life, items) %.3f corr(life, nefc) %.3f' % (np.corrcoef(life, a[:, 4])[0, 1], np.corrcoef(life, a[:, 6])[0, 1], np.corrcoef(life, a[:, 7])[0, 1]))
A = np.stack([np.ones(nb), a[:, 4], a[:, 6], a[:, 7]], 1).astype(np.float64)
coef = np.linalg.lstsq(A, life * 300 / 300, rcond=None)[0]
print('life us ~ %.0f + %.2f * newton_trips + %.2f * items + %.2f * nefc' % tuple(coef))

names = ['A load', 'B/C M+bias', 'D chol M', 'E1-2', 'E3', 'E4-5', 'F0 warm', 'F grad', 'F hess', 'F chol', 'F ls loop', 'F eval', 'G(12)', 'F ls mv+sums(13)', 'exit+qfc(14)', 'G chol(15)',
         'K kin(16)', 'C(17)', 'integrate(18)', 'C cull1(19)', 'C cull2(20)', 'C plane(21)', 'C mpr(22)', 'C boxbox(23)', 'K pose(24)', 'F ls jmul(25)']
# the six spare stamps 26..31 (solve_g.h: HSR_SUBPROF of the timing build; say which with the environment variable of the same name)
SUB = {0: ['K A local transforms', 'K B world poses', 'K C dof axes', 'K D velocity fields', 'C geom placement', 'K goal test + xpos out'],
       1: ['M item setup', 'M margin read', 'M hull rescans', 'M mpr call', 'M result stores', '(x31)'],
       2: ['c1 pair reads + sphere tests', 'c1 ballot compaction', 'c2 box cull', 'c2 item compaction', 'C geom placement*', '(x31)'],
       3: ['bb item list', 'bb item geoms', 'bb face axes', 'bb edge axes', 'bb clips', '(x31)'],
       4: ['B link loop', 'B ancestors -> M', 'B qfrc + sync', 'B mirror + rows', '(x30)', '(x31)'],
       5: ['E1 limits', 'E3 global reads', 'E3 frames / impedance', 'E3 link masks + sync / E5 aref', 'E5 cached Jacobian', '(x31)'],
       6: ['F0 cost at qacc_smooth', 'F0 eval at warm start', '(x28)', '(x29)', '(x30)', '(x31)'],
       7: ['H init', 'H cached contacts', 'H further contacts', 'H transposition', '(x30)', '(x31)']}
sub = int(os.environ.get('HSR_SUBPROF', '0'))
names += [f'x{26 + i} {n}' for i, n in enumerate(SUB[sub])]
if sub:
    print(f'HSR_SUBPROF={sub}: stamps 26..31 sit inside another phase - the time up to the first of them is charged to it, the parent phase keeps what follows the last')
ph = a[:, 8:40].astype(np.float64) / 300.0
slow, med = o[-20:], o[nb // 2 - 10: nb // 2 + 10]
print('phase cycles per substep: slowest 20 blocks vs 20 median blocks')
for i, nm in enumerate(names):
    if ph[:, i].sum() > 0:
        print(f'  {nm:14s} {ph[slow, i].mean():9.0f} {ph[med, i].mean():9.0f}')
print(f'  {"total":14s} {ph[slow].sum(1).mean():9.0f} {ph[med].sum(1).mean():9.0f}')
