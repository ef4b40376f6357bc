mkdir -p gpurun_out/r2
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2/t_all.log 2>&1; tail -3 gpurun_out/r2/t_all.log
timeout -k 10 600 python bench.py --no-cpu-baseline > gpurun_out/r2/bench_d.json 2>gpurun_out/r2/bench_d.err || tail -5 gpurun_out/r2/bench_d.err
python - <<EOF
import json
d=json.load(open("gpurun_out/r2/bench_d.json"))
print(round(d["value"]), d["ms_per_step"], d["roofline"]["frac"], {n: (x["launches"], round(x["avg_ms"],3)) for n,x in d["kernels"].items()})
for k,v in d["legs"].items(): print(" ", k, round(v["value"]), round(v["ms_per_step"],3))
EOF
timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --rows-per-gpu 12500000 --batch 1024 --steps 6 --warmup 2 > gpurun_out/r2/b125.json 2>/dev/null
python - <<EOF
import json
d=json.load(open("gpurun_out/r2/b125.json"))
print("12.5M x 1024:", round(d["value"]), d["ms_per_step"], {n: (x["launches"], round(x["avg_ms"],3)) for n,x in d["kernels"].items() if n in ("screen_i8_fused","count_planes","expand_hits","fuse_select")})
EOF
