set -x
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q -s > gpurun_out/r4/c19_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r4/c19_tests.log
tail -6 gpurun_out/r4/c19_tests.log; grep "hulls touch\|solo servers vs" gpurun_out/r4/c19_tests.log
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4/c19_bench.log 2>&1; tail -1 gpurun_out/r4/c19_bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', round(d['value']/1e3,1))"
