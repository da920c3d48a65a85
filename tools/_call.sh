mkdir -p gpurun_out/r4
timeout -k 10 600 bash tools/ab.sh v12 > gpurun_out/r4/c23_ab.log 2>&1; cat gpurun_out/r4/c23_ab.log; grep "F ls loop\|total" gpurun_out/r4/ab_v12_bt.log
