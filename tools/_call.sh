set -x
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r4/c21_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r4/c21_tests.log
tail -4 gpurun_out/r4/c21_tests.log
for c in cfg2 cfg3 cfg4 cupboard cfg1; do timeout -k 10 200 python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4/c21_bench_$c.log 2>&1; tail -1 gpurun_out/r4/c21_bench_$c.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c', round(d['value']/1e3,1), d['roofline']['kernel_ms_mean'])"; done
timeout -k 10 300 python bench.py --envs-per-gpu 65536 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r4/c21_bench_64k.log 2>&1; tail -1 gpurun_out/r4/c21_bench_64k.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('64k', round(d['value']/1e3,1))"
