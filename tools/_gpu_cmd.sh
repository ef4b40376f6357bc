# scratch: what the last GPU call of the round ran (gpurun -- 'bash tools/_gpu_cmd.sh')
mkdir -p gpurun_out/r2
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2/t_all.log 2>&1; tail -3 gpurun_out/r2/t_all.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2/bench_driver_form.json 2> gpurun_out/r2/bench_driver_form.err; tail -c 400 gpurun_out/r2/bench_driver_form.json
