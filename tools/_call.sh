set -x
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q -s > gpurun_out/r4/c8_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r4/c8_tests.log
timeout -k 10 600 bash tools/ab.sh v7 > gpurun_out/r4/c8_ab.log 2>&1
tail -5 gpurun_out/r4/c8_tests.log; cat gpurun_out/r4/c8_ab.log; grep "beyond 2e-3" gpurun_out/r4/c8_tests.log | cut -c1-600
