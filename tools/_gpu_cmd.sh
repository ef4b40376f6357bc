mkdir -p gpurun_out/r2
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batcher.py tests/test_recall_search_service.py -m gpu -x -q > gpurun_out/r2/t1.log 2>&1; tail -5 gpurun_out/r2/t1.log
( time timeout -k 10 400 python bench.py ) > gpurun_out/r2/bench_default_2.json 2> gpurun_out/r2/bench_default_2.err; tail -c 600 gpurun_out/r2/bench_default_2.err
python - <<EOF
import json
d=json.load(open("gpurun_out/r2/bench_default_2.json"))
print(round(d["value"]), d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["parity"]["rank_identical"], d["parity"]["max_abs_score_delta"], d["setup_s"], d["total_s"])
for k,v in d["legs"].items():
    ss=v["search_stats"]; print(" ", k, round(v["value"]), round(v["ms_per_step"],3), v["roofline"]["kernel"], round(v["roofline"]["avg_launch_ms"],3), "surv/q", ss["survivors_per_query"], "max", ss["survivors_max"], "passes", ss["passes"], "/", ss["searches"], "cap", ss["survivor_capacity"], "tokens", ss["vocab_tokens"], "hits/pass", ss["kw_hits_total"]/max(1,ss["kw_passes"]), v["rank1_is_planted_row"])
    print("     ", {n: round(x["avg_ms"],3) for n,x in v["kernels"].items()})
EOF
