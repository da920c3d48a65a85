"""Condense the rocprofv3 outputs of tools/profile_round.sh into profiles/: the kernel-trace stats CSV of the bench command and
pmc_summary.json (HBM-side bytes per launch from FETCH_SIZE / WRITE_SIZE with the gfx950 correction of MI355X_MICROARCH.md -
FETCH_SIZE counts half of a wide streaming read - and the SQ counters of the persistent kernel per wave and substep).
usage: python tools/pmc_collect.py gpurun_out/prof_<tag> <tag> [substeps=300] [waves=2048]
The summary is rebuilt from scratch (kernels that no longer exist do not linger); its _note names the commit the numbers are from."""
import csv, glob, json, shutil, sys
from collections import defaultdict
from pathlib import Path

src, tag = Path(sys.argv[1]), sys.argv[2]
nsub = int(sys.argv[3]) if len(sys.argv) > 3 else 300
waves = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
root = Path(__file__).resolve().parents[1]
prof = root / "profiles"


def counters(d):
    out = defaultdict(lambda: defaultdict(float)); launches = defaultdict(set)
    files = sorted(glob.glob(str(src / d / "*" / "*counter_collection.csv")), key=lambda f: Path(f).stat().st_mtime)
    for f in files[-1:]:            # gpurun merges every call's outputs into the same directory: the newest run only
        rows = list(csv.DictReader(open(f)))
        # the persistent kernel: its LAST launch only (the PMC passes run three warm-up env-steps first: the launches right after a reset are short -
        # nothing is in contact yet - and would flatter every per-launch figure)
        last = max((int(r["Dispatch_Id"]) for r in rows if "k_env_step_mf" in r["Kernel_Name"]), default=-1)
        for r in rows:
            k = r["Kernel_Name"].split("(")[0].split("<")[0]
            k = k[k.find("k_"):] if "k_" in k else k
            if "k_env_step_mf" in k and int(r["Dispatch_Id"]) != last: continue
            out[k][r["Counter_Name"]] += float(r["Counter_Value"]); launches[k].add(r["Dispatch_Id"])
    return out, {k: len(v) for k, v in launches.items()}


stats = sorted(glob.glob(str(src / "trace" / "*" / "*kernel_stats.csv")), key=lambda f: Path(f).stat().st_mtime)
if stats:
    shutil.copy(stats[-1], prof / f"{tag}_kernel_stats.csv")
fetch, nf = counters("pmc_fetch"); write, nw = counters("pmc_write"); sq, ns = counters("pmc_sq")
import subprocess
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=root).stdout.strip()
dirty = bool(subprocess.run(["git", "status", "--porcelain", "--", "hsr_env_amd", "bench.py"], capture_output=True, text=True, cwd=root).stdout.strip())
summary = {}
for k in sorted(set(fetch) | set(write)):
    f = fetch[k].get("FETCH_SIZE", 0.0) / max(nf.get(k, 1), 1); w = write[k].get("WRITE_SIZE", 0.0) / max(nw.get(k, 1), 1)
    summary[k] = {"fetch_kb_per_launch": f, "write_kb_per_launch": w, "hbm_bytes_per_launch": (2 * f + w) * 1024, "launches": nf.get(k, 0)}
for k, v in sq.items():
    if "k_env_step_mf" in k:
        n = max(ns.get(k, 1), 1) * waves * nsub
        summary["_sq_per_wave_per_substep"] = {c: round(x / n, 1) for c, x in sorted(v.items())}
        summary["_sq_per_wave_per_substep"]["note"] = (f"{tag}: SQ counters of k_env_step_mf divided by (launches x {waves} waves x {nsub} substeps); "
                                                       "*_CYCLES / ACTIVE / WAIT in units of 4 clocks")
mix = {}
for d in ("pmc_mixa", "pmc_mixb"):          # tools/pmc_mix.sh: instruction mix and issue cycles per class
    cm, _ = counters(d)
    for k, v in cm.items():
        if "k_env_step_mf" in k:
            mix.update({c: round(x / (waves * nsub), 1) for c, x in v.items()})
if mix:
    mix["note"] = (f"{tag}: per wave and substep, last launch of k_env_step_mf (tools/pmc_mix.sh); SQ_ACTIVE_INST_* / SQ_INST_CYCLES_* / SQ_WAVE_CYCLES / SQ_WAIT_* in units of 4 clocks: "
                   "the share of a wave's lifetime spent issuing each instruction class - the rest is waiting (s_waitcnt, issue arbitration)")
    summary["_mix_per_wave_per_substep"] = dict(sorted(mix.items()))
wr, nwr = counters("pmc_wr")
for k, v in wr.items():
    if "k_env_step_mf" in k and v.get("TCC_EA0_WRREQ_sum"):
        n64, nall, l2w = v.get("TCC_EA0_WRREQ_64B_sum", 0.0), v["TCC_EA0_WRREQ_sum"], v.get("TCC_WRITE_sum", 0.0)
        summary["_hbm_writes_of_k_env_step_mf"] = {
            "write_requests_leaving_L2": nall, "of_them_full_64B_lines": n64, "bytes_full_lines": 64 * n64, "bytes_32B_partial": 32 * (nall - n64),
            "write_requests_reaching_L2": l2w,
            "reading": "one launch (the fourth after a reset).  The counters cannot name buffers; they separate two kinds.  Full 64 B lines are wave-wide rows: the "
                       "register spills (scratch_store: 64 lanes x 4..16 B contiguous; round 6: 12 B of scratch per lane in the cfg3 instance = 1.6 MB for the 2048 waves; "
                       "68 B in round 5, 208 B in round 3) and the once-per-launch SoA outputs (obs / qpos / qvel / warm start: 3 MB) - the rest of the full lines are runs of "
                       "partial writes that L2 merged before they left.  "
                       "32 B partial writes are single words of env-strided arrays - DevState::sepax / septick (separation margin + stamp of every convex item, "
                       "two words per item and substep, 19 MB footprint), the per-pair contact counts - and the 32 B contact records (DevState::con, 2 x float4 per contact).  "
                       "Requests reaching L2 against requests leaving it = how much of the store stream L2 absorbs."}
# the kernel-stats CSV averages every launch of the trace run, warm-up included (the first launches after a reset are short: nothing is in
# contact yet); what bench.py times are the last `--steps` launches: their mean from the same trace, next to the bench line's own figure
try:
    import re
    tr = sorted(glob.glob(str(src / "trace" / "*" / "*kernel_trace.csv")), key=lambda f: Path(f).stat().st_mtime)[-1]
    rows = sorted((r for r in csv.DictReader(open(tr)) if "k_env_step_mf" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    line = [l for l in open(src / "trace.log") if l.startswith("{")][-1]
    b = json.loads(line)
    k = b["steps"]
    summary["_timed_launches"] = {"all_launches_ms": [round(x, 2) for x in dur], "timed": k, "mean_ms_of_the_timed_launches_rocprofv3": round(sum(dur[-k:]) / k, 3),
                                  "mean_ms_bench_hip_events_same_run": round(b["roofline"]["kernel_ms_mean"], 3), "bench_value_same_run": round(b["value"], 1)}
except Exception as ex:           # a missing trace is not an error of the PMC summary
    summary["_timed_launches"] = {"error": str(ex)}
summary["_note"] = (f"{tag}, commit {commit}{' + uncommitted changes' if dirty else ''}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* in separate passes (with --kernel-trace only), python3 bench.py --steps 1 --no-capacity "
                    "--warmup 3 --no-cpu-baseline (cfg3, 8192 envs; k_env_step_mf: the fourth, timed launch only); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per the gfx950 correction of "
                    "MI355X_MICROARCH.md; k_env_step_mf: one launch = 300 substeps of 8192 envs (algorithmic 630 MB) including ctrl in and obs / reward / done out; narrow accesses are uncalibrated")
(prof / "pmc_summary.json").write_text(json.dumps(summary, indent=1))
print(json.dumps({k: summary[k] for k in summary if "env_step" in k or k.startswith("_sq")}, indent=1))
