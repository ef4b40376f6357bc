#!/bin/bash
# Runs on the GPU box (gpurun): the round's bench lines beside the default one -- batch sweep at 1M rows, the per-GPU
# shape of the 100M-row configs (12.5M rows), each as the single JSON line bench.py prints.  Output: gpurun_out/sweep/.
set -u
cd "$(dirname "$0")/.." ; OUT=gpurun_out/sweep ; rm -rf $OUT ; mkdir -p $OUT
( time timeout -k 10 500 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err
for B in 1 4 8 32 64 128 256 1024; do
  timeout -k 10 120 python bench.py --no-legs --no-cpu-baseline --rows-per-gpu 1000000 --batch $B > $OUT/bench_1m_b$B.json 2>/dev/null
done
for B in 1 256 1024; do
  timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --rows-per-gpu 12500000 --batch $B --steps 10 --warmup 3 > $OUT/bench_12p5m_b$B.json 2>/dev/null
done
timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --rows-per-gpu 12500000 --batch 256 --no-terms --steps 10 --warmup 4 > $OUT/bench_12p5m_b256_cosine_only.json 2>/dev/null
timeout -k 10 120 python bench.py --no-legs --no-cpu-baseline --rows-per-gpu 1000000 --batch 1 --set-option two_stage=0 > $OUT/bench_1m_b1_exact_kernel.json 2>/dev/null
python - <<EOF
import json, glob, os
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(os.path.basename(f), round(d["value"]), "q/s", round(d["ms_per_step"], 3), "ms", r["kernel"], round(r["avg_launch_ms"], 3), "ms", r["bound"], round(r["frac"], 3),
              "surv/q", d["search_stats"]["survivors_per_query"], "passes/searches", d["search_stats"]["passes"], d["search_stats"]["searches"], d["rank1_is_planted_row"])
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
EOF
tail -4 $OUT/bench_default.err
