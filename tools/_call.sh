set -x
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r4/c9_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r4/c9_tests.log
timeout -k 10 600 bash tools/ab.sh v8 > gpurun_out/r4/c9_ab.log 2>&1
for c in cfg1 cfg2 cfg4 cupboard; do timeout -k 10 200 python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4/c9_bench_$c.log 2>&1; done
timeout -k 10 300 python bench.py --envs-per-gpu 65536 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r4/c9_bench_64k.log 2>&1
tail -3 gpurun_out/r4/c9_tests.log; cat gpurun_out/r4/c9_ab.log
for f in cfg1 cfg2 cfg4 cupboard 64k; do python - $f <<'PY'
import json,sys
f=sys.argv[1]
try:
    d=json.loads(open(f'gpurun_out/r4/c9_bench_{f}.log').read().strip().splitlines()[-1]); print(f, round(d['value']/1e3,1),'k', round(d['ms_per_step'],2),'ms')
except Exception as e: print(f,'failed',e)
PY
done
