mkdir -p gpurun_out/r2
( time timeout -k 10 300 python bench.py --gpus 2 --backend gloo --rows-per-gpu 1000000 --batch 64 --steps 5 --warmup 2 ) > gpurun_out/r2/bench_2rank_gloo.json 2> gpurun_out/r2/bench_2rank_gloo.err; tail -c 1500 gpurun_out/r2/bench_2rank_gloo.err
python - <<EOF
import json
d=json.loads(open("gpurun_out/r2/bench_2rank_gloo.json").read().strip().splitlines()[-1])
print(d["config"]["workload"]); print(round(d["value"]), d["ms_per_step"], d["n_gpus"], d.get("backend"), d.get("rccl_ranks_seen"), d.get("collectives_per_step"), d["rank1_is_planted_row"], d["roofline"]["kernel"])
for k,v in d["legs"].items(): print(" ", k, round(v["value"]), round(v["ms_per_step"],3), v["rank1_is_planted_row"], v["search_stats"]["survivors_per_query"], v["search_stats"]["passes"], v["search_stats"]["searches"])
EOF
