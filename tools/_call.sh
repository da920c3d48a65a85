mkdir -p gpurun_out/r4
timeout -k 10 300 python -m pytest tests/test_gpu_hotpath.py tests/test_gpu_parity.py -m gpu -q -x -s -k "solo" > gpurun_out/r4/c25_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r4/c25_tests.log
tail -3 gpurun_out/r4/c25_tests.log; grep "solo servers vs" gpurun_out/r4/c25_tests.log
timeout -k 10 200 python tools/experiments/solo_chain.py cfg3 512 > gpurun_out/r4/c25_chain.log 2>&1; tail -5 gpurun_out/r4/c25_chain.log
timeout -k 10 200 python tools/experiments/solo_chain.py cfg3 64 > gpurun_out/r4/c25_chain64.log 2>&1; tail -4 gpurun_out/r4/c25_chain64.log
