"""Static instruction counts of one kernel per source region: python tools/isa_map.py KERNEL.s [--top persist.h,solve_body.inc] [--bucket 10]
KERNEL.s = the kernel's part of a `hipcc -gline-tables-only -save-temps` assembly (the .loc directives map instructions to source lines
without changing the code).  Inlined helpers (devmath.h, solve_g.h, collide.h ...) are attributed to the last line of a TOP file seen before
them, so the listing follows the phases of the persistent kernel; per region: VALU / SALU / LDS / VMEM / branches / waitcnt / scratch."""
import re, sys, collections
path = sys.argv[1]
top = ("persist.h", "solve_body.inc")
bucket = 1
for a in sys.argv[2:]:
    if a.startswith("--top="): top = tuple(a[6:].split(","))
    if a.startswith("--bucket="): bucket = int(a[9:])
files = {}
cur = None; curtop = ("?", 0)
rows = collections.OrderedDict()
order = []
for ln in open(path):
    s = ln.strip()
    m = re.match(r"\.file\s+(\d+)\s+\"[^\"]*\"\s+\"([^\"]+)\"", s)
    if m: files[int(m.group(1))] = m.group(2); continue
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
    if m:
        f = files.get(int(m.group(1)), "?"); l = int(m.group(2))
        if f in top and l > 0: curtop = (f, l // bucket * bucket)
        continue
    if not s or s.startswith((";", ".", "//")) or s.endswith(":"): continue
    op = s.split()[0]
    k = curtop
    if k not in rows: rows[k] = collections.Counter()
    c = rows[k]
    if op.startswith("v_"): c["valu"] += 1
    if op.startswith("v_") and "dpp" in s: c["dpp"] += 1
    if op.startswith("v_mfma"): c["mfma"] += 1
    if op.startswith("s_") and not op.startswith(("s_waitcnt", "s_cbranch", "s_branch", "s_nop")): c["salu"] += 1
    if op.startswith("ds_"): c["lds"] += 1
    if op.startswith(("global_", "flat_", "buffer_")): c["vmem"] += 1
    if op.startswith("scratch_"): c["scratch"] += 1
    if op.startswith(("s_cbranch", "s_branch")): c["br"] += 1
    if op.startswith("s_waitcnt"): c["wait"] += 1
    if op.startswith("s_nop"): c["nop"] += 1
    if op in ("v_readlane_b32", "v_writelane_b32"): c["lane"] += 1
    c["all"] += 1
keys = sorted(rows, key=lambda k: (top.index(k[0]) if k[0] in top else 9, k[1]))
print(f"{'file:line':28s} {'all':>6s} {'valu':>6s} {'dpp':>5s} {'salu':>5s} {'lds':>5s} {'vmem':>5s} {'scr':>4s} {'br':>4s} {'wait':>5s} {'nop':>4s} {'lane':>5s}")
tot = collections.Counter()
for k in keys:
    c = rows[k]; tot.update(c)
    print(f"{k[0] + ':' + str(k[1]):28s} {c['all']:6d} {c['valu']:6d} {c['dpp']:5d} {c['salu']:5d} {c['lds']:5d} {c['vmem']:5d} {c['scratch']:4d} {c['br']:4d} {c['wait']:5d} {c['nop']:4d} {c['lane']:5d}")
c = tot
print(f"{'total':28s} {c['all']:6d} {c['valu']:6d} {c['dpp']:5d} {c['salu']:5d} {c['lds']:5d} {c['vmem']:5d} {c['scratch']:4d} {c['br']:4d} {c['wait']:5d} {c['nop']:4d} {c['lane']:5d}")
