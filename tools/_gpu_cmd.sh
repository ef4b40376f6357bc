mkdir -p gpurun_out/r2
for v in "" _v34 _v25 _v24 _v35; do
  export ORR_HIP_LIB=$PWD/omni-recall-rag_amd/libomnirecall_hip$v.so
  for rep in 1 2; do
  timeout -k 10 100 python bench.py --no-legs --no-cpu-baseline --rows-per-gpu 1000000 --batch 256 > gpurun_out/r2/bv$v.json 2>/dev/null
  python - <<EOF
import json
d=json.load(open("gpurun_out/r2/bv$v.json")); print("variant '$v'", round(d["value"]), round(d["ms_per_step"],4), round(d["roofline"]["avg_launch_ms"],4), d["rank1_is_planted_row"])
EOF
  done
done
unset ORR_HIP_LIB
