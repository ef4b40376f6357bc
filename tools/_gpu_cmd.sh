mkdir -p gpurun_out/r2
bash tools/bench_sweep.sh > gpurun_out/r2/sweep.log 2>&1; tail -22 gpurun_out/r2/sweep.log
