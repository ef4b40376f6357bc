mkdir -p gpurun_out/r2
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "four_wave" > gpurun_out/r2/t4w.log 2>&1; tail -15 gpurun_out/r2/t4w.log
