#!/bin/bash
# Socket power and shader clock (rocm-smi) while the default bench workload (C3) runs: is the screening GEMM power-limited?
# Usage (GPU box): bash tools/power_sample.sh > gpurun_out/power.txt
out=${1:-/dev/stdout}
rocm-smi --showmaxpower 2>&1 | grep -E "Max Graphics" 
(timeout -k 10 200 python bench.py --no-legs --no-cpu-baseline --steps 2500 --warmup 5 > /tmp/power_bench.json 2>/tmp/power_bench.err) &
BP=$!
for i in $(seq 1 90); do
  if ! kill -0 $BP 2>/dev/null; then break; fi
  echo "t=$(date +%s.%N | cut -c1-14) $(rocm-smi --showpower --showclocks 2>&1 | grep -E "Power \(W\)|sclk" | sed 's/.*: //' | tr '\n' ' ')"
  sleep 0.4
done
wait $BP
python - <<PY
import json
d=json.loads(open('/tmp/power_bench.json').read().strip().splitlines()[-1])
print("bench: %.0f queries/s, %.3f ms per batch, screening kernel %.3f ms per launch x %d" % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['launches_per_step']))
PY
