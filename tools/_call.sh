mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/r4/c31_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r4/c31_tests.log; tail -3 gpurun_out/r4/c31_tests.log
timeout -k 10 600 bash tools/ab.sh v14 v13 > gpurun_out/r4/c31_ab.log 2>&1; cat gpurun_out/r4/c31_ab.log; grep "E1-2\|E3 \|total" gpurun_out/r4/ab_v14_bt.log
