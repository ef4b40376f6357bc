mkdir -p gpurun_out/r2
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2/t_all.log 2>&1; tail -3 gpurun_out/r2/t_all.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/bench_sweep.sh > gpurun_out/r2/sweep.log 2>&1; tail -20 gpurun_out/r2/sweep.log
