mkdir -p gpurun_out/r2
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2/t_all.log 2>&1; tail -3 gpurun_out/r2/t_all.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2/bench_driver_form.json 2> gpurun_out/r2/bench_driver_form.err
python - <<EOF
import json
d=json.loads(open("gpurun_out/r2/bench_driver_form.json").read().strip().splitlines()[-1])
print(round(d["value"]), d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["mfma"]["frac_of_sustained"], d["parity"]["rank_identical"], d["parity"]["max_abs_score_delta"], {n: (x["launches"], round(x["avg_ms"],3)) for n,x in d["kernels"].items() if "screen" in n})
for k,v in d["legs"].items(): print(" ", k, round(v["value"]), round(v["ms_per_step"],3))
EOF
