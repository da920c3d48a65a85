#!/bin/bash
# instruction-fetch counters of the persistent kernel for one config: tools/pmc_icache.sh CFG
CFG=${1:-cfg3}
OUT=gpurun_out/pmc_ic_$CFG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_IFETCH SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU -d $OUT/sq -- python3 bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-capacity > $OUT/sq.log 2>&1 || exit 1
rocprofv3 --kernel-trace --output-format csv --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $OUT/sqc -- python3 bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-capacity > $OUT/sqc.log 2>&1 || exit 1
python3 - <<PY
import csv, glob
from collections import defaultdict
for d in ('sq', 'sqc'):
    acc = defaultdict(float); n = set()
    for f in glob.glob('$OUT/' + d + '/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if 'k_env_step_mf' in r['Kernel_Name']:
                acc[r['Counter_Name']] += float(r['Counter_Value']); n.add(r['Dispatch_Id'])
    print(d, 'launches', len(n), {k: v / max(len(n), 1) for k, v in acc.items()})
PY
