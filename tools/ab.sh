#!/bin/bash
# A/B of build variants (tools/build_variants.py) on the GPU box: tools/ab.sh [-t "pytest -k expr"] NAME...
# per variant: optional parity tests, bench line (cfg3), workgroup lifetimes (var_NAME_l.so), phase profile (var_NAME_t.so)
OUT=${AB_OUT:-gpurun_out/r6}; mkdir -p $OUT
KEXPR=""
if [ "$1" == "-t" ]; then KEXPR="$2"; shift; shift; fi
for v in "$@"; do
  if [ -n "$KEXPR" ]; then HSR_LIB=$PWD/hsr_env_amd/var_${v}.so python -m pytest tests -m gpu -x -q -k "$KEXPR" > $OUT/ab_${v}_tests.log 2>&1; echo "rc=$?" >> $OUT/ab_${v}_tests.log; fi
  HSR_LIB=$PWD/hsr_env_amd/var_${v}.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-capacity --config ${HSR_CFG:-cfg3} > $OUT/ab_${v}_bench.log 2>&1 || exit 1
  [ -f hsr_env_amd/var_${v}_l.so ] && HSR_LIB=$PWD/hsr_env_amd/var_${v}_l.so python tools/block_life.py > $OUT/ab_${v}_life.log 2>&1
  [ -f hsr_env_amd/var_${v}_t.so ] && HSR_LIB=$PWD/hsr_env_amd/var_${v}_t.so python tools/block_times.py > $OUT/ab_${v}_bt.log 2>&1
  python - $v <<'PY'
import json,sys,re,os
v=sys.argv[1]; out=os.environ.get("AB_OUT","gpurun_out/r6")+"/"
d=json.loads(open(out+f'ab_{v}_bench.log').read().strip().splitlines()[-1])
r=d['roofline']
line=f"{v:12s} {d['value']/1e3:7.1f}k  kernel mean {r.get('kernel_ms_mean',0):.2f} min {r.get('kernel_ms_min',0):.2f} max {r.get('kernel_ms_max',0):.2f}"
try:
    t=open(out+f'ab_{v}_life.log').read()
    m=re.search(r'life percentiles us.*\] \[(.*)\]',t)
    line+='  life '+m.group(1)
except Exception: pass
try:
    t=open(out+f'ab_{v}_tests.log').read().strip().splitlines()
    line+='  tests: '+t[-2][:60]+' '+t[-1]
except Exception: pass
print(line)
PY
done
