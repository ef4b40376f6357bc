mkdir -p gpurun_out/r2
export ORR_TILE16=1
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "four_wave" > gpurun_out/r2/t16.log 2>&1; tail -12 gpurun_out/r2/t16.log
for v in 0 1; do
ORR_TILE16=$v timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --rows-per-gpu 1000000 --batch 256 --steps 20 --warmup 3 > gpurun_out/r2/s16_$v.json 2>gpurun_out/r2/s16_$v.err || echo FAIL $v
python - <<EOF
import json
d=json.load(open("gpurun_out/r2/s16_$v.json"))
print("TILE16=$v", round(d["value"]), round(d["ms_per_step"],3), {n: round(x["avg_ms"],3) for n,x in d["kernels"].items() if "screen" in n}, d["rank1_is_planted_row"])
EOF
rm -f /tmp/st_1.bin
ORR_TILE16=$v ORR_SCREEN_STAMPS=/tmp/st_1.bin timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --rows-per-gpu 1000000 --batch 256 --steps 3 --warmup 1 > /dev/null 2>&1 || echo FAIL
python tools/analyze_stamps.py /tmp/st_1.bin | grep -E "K loop|epilogue cycles|clock|row constants|first block|blocks 2"
done
