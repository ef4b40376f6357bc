mkdir -p gpurun_out/r2
( time timeout -k 10 800 python -m pytest tests/test_gpu_scale_c5_shard.py -m gpu -x -q -s ) > gpurun_out/r2/t_c5.log 2>&1; tail -15 gpurun_out/r2/t_c5.log
