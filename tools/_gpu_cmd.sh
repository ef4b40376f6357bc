mkdir -p gpurun_out/r2; rm -f gpurun_out/r2/stamps_*.bin
timeout -k 10 600 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r2/t1.log 2>&1; tail -3 gpurun_out/r2/t1.log
ORR_EPI_GATE=0 ORR_SCREEN_STAMPS=gpurun_out/r2/stamps_nogate.bin timeout -k 10 100 python bench.py --no-legs --no-cpu-baseline --rows-per-gpu 1000000 --batch 256 --steps 3 --warmup 1 > gpurun_out/r2/b2.json 2>/dev/null
python tools/analyze_stamps.py gpurun_out/r2/stamps_nogate.bin
rm -f gpurun_out/r2/stamps_*.bin
for v in gate nogate; do
  if [ $v = nogate ]; then export ORR_EPI_GATE=0; else unset ORR_EPI_GATE; fi
  timeout -k 10 100 python bench.py --no-legs --no-cpu-baseline --rows-per-gpu 1000000 --batch 256 > gpurun_out/r2/bench_1m_b256_$v.json 2>/dev/null
  timeout -k 10 100 python bench.py --no-legs --no-cpu-baseline --rows-per-gpu 1000000 --batch 256 --no-terms > gpurun_out/r2/bench_1m_b256_noterms_$v.json 2>/dev/null
done
unset ORR_EPI_GATE
timeout -k 10 200 python bench.py --no-legs --no-cpu-baseline > gpurun_out/r2/bench_c3.json 2>/dev/null
python - <<EOF
import json
for f in ("bench_1m_b256_gate","bench_1m_b256_nogate","bench_1m_b256_noterms_gate","bench_1m_b256_noterms_nogate","bench_c3"):
    try:
        d=json.load(open("gpurun_out/r2/%s.json"%f))
        print(f, round(d["value"]), d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], d["search_stats"]["survivors_per_query"], d["kernels"].get("count_planes"))
    except Exception as e: print(f,"ERR",e)
EOF
