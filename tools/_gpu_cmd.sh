mkdir -p gpurun_out/r2
for rep in 1 2 3; do for v in "" _vns; do
  export ORR_HIP_LIB=$PWD/omni-recall-rag_amd/libomnirecall_hip$v.so
  timeout -k 10 100 python bench.py --no-legs --no-cpu-baseline --rows-per-gpu 1000000 --batch 256 > gpurun_out/r2/bv$v.json 2>/dev/null
  python - <<EOF
import json
d=json.load(open("gpurun_out/r2/bv$v.json")); print("variant '$v'", round(d["value"]), round(d["ms_per_step"],4), round(d["roofline"]["avg_launch_ms"],4), d["rank1_is_planted_row"])
EOF
done; done
unset ORR_HIP_LIB
timeout -k 10 100 python bench.py --no-legs --no-cpu-baseline --rows-per-gpu 1000000 --batch 1024 > gpurun_out/r2/b1024.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r2/b1024.json')); print('b1024', round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_ms'])"
timeout -k 10 600 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r2/t1.log 2>&1; tail -3 gpurun_out/r2/t1.log
