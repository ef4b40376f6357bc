mkdir -p gpurun_out/r2/tl3
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2/t_all.log 2>&1; tail -3 gpurun_out/r2/t_all.log
ORR_HOST_TIMING=1 timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --rows-per-gpu 1000000 --batch 1 --steps 256 --warmup 10 > gpurun_out/r2/b1.json 2> gpurun_out/r2/b1.err
tail -1 gpurun_out/r2/b1.err
python - <<EOF
import json
d=json.load(open("gpurun_out/r2/b1.json"))
print(round(d["value"]), d["ms_per_step"], {n: round(x["avg_ms"],4) for n,x in d["kernels"].items()})
EOF
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2/tl3 -o b1 -- python3 $GRAFT_REPO_ROOT/bench.py --no-legs --no-cpu-baseline --rows-per-gpu 1000000 --batch 1 --steps 20 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/r2/tl3/bench.json 2>$GRAFT_REPO_ROOT/gpurun_out/r2/tl3/err.log
