mkdir -p gpurun_out/r2
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -x -q > gpurun_out/r2/t4.log 2>&1; tail -2 gpurun_out/r2/t4.log
timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --rows-per-gpu 1000000 --batch 256 --steps 20 --warmup 3 > gpurun_out/r2/s4_1.json 2>gpurun_out/r2/s4_1.err || echo FAIL
python - <<EOF
import json
d=json.load(open("gpurun_out/r2/s4_1.json"))
print(round(d["value"]), round(d["ms_per_step"],3), {n: round(x["avg_ms"],3) for n,x in d["kernels"].items() if "screen" in n})
EOF
rm -f /tmp/st_1.bin
ORR_SCREEN_STAMPS=/tmp/st_1.bin timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --rows-per-gpu 1000000 --batch 256 --steps 3 --warmup 1 > /dev/null 2>&1 || echo FAIL
python tools/analyze_stamps.py /tmp/st_1.bin
