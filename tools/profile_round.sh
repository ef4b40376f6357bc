#!/bin/bash
# Runs on the GPU box (gpurun): the rocprofv3 evidence of the round.  Kernel trace and counters in SEPARATE runs
# (a --pmc run carries no trace domain but --kernel-trace); summaries land under gpurun_out/prof/.
set -u
cd "$(dirname "$0")/.." ; export TMPDIR=/tmp
OUT=gpurun_out/prof ; rm -rf $OUT ; mkdir -p $OUT
C3="bench.py --no-legs --no-overlap-leg --no-cpu-baseline --no-pmc --steps 10 --warmup 3"
C2="bench.py --no-legs --no-overlap-leg --no-cpu-baseline --no-pmc --rows-per-gpu 1000000 --batch 1 --steps 20 --warmup 3"
P256="bench.py --no-legs --no-overlap-leg --no-cpu-baseline --no-pmc --rows-per-gpu 262144 --batch 256 --steps 3 --warmup 1"
run() { name=$1; shift; echo "== $name: rocprofv3 $*" ; timeout -k 10 280 rocprofv3 --output-format csv "$@" > $OUT/$name.stdout 2> $OUT/$name.stderr; echo "rc=$?"; }
run kt_c3   --kernel-trace --stats -d $OUT/kt_c3 -- python3 $C3
tail -1 $OUT/kt_c3.stdout > $OUT/bench_c3_under_trace.json
f=$(ls $OUT/kt_c3/*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -40 "$f" > $OUT/kernel_stats_c3.csv
run kt_c2   --kernel-trace --stats -d $OUT/kt_c2 -- python3 $C2
tail -1 $OUT/kt_c2.stdout > $OUT/bench_c2_under_trace.json
f=$(ls $OUT/kt_c2/*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -40 "$f" > $OUT/kernel_stats_c2.csv
run fetch_c3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch_c3 -- python3 $C3
run write_c3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write_c3 -- python3 $C3
run fetch_c2 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch_c2 -- python3 $C2
run write_c2 --kernel-trace --pmc WRITE_SIZE -d $OUT/write_c2 -- python3 $C2
run sq_a --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/sq_a -- python3 $P256
run sq_b --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/sq_b -- python3 $P256
run sq_c --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/sq_c -- python3 $P256
for d in fetch_c3 write_c3 fetch_c2 write_c2 sq_a sq_b sq_c; do python3 tools/summarize_pmc.py $OUT/$d "screen" > $OUT/$d.json 2>/dev/null; done
# the raw traces are large: keep the summaries only
rm -rf $OUT/kt_c3 $OUT/kt_c2 $OUT/fetch_c3 $OUT/write_c3 $OUT/fetch_c2 $OUT/write_c2 $OUT/sq_a $OUT/sq_b $OUT/sq_c
ls -la $OUT
