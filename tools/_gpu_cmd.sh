ls -la tools/x_m16.so
for X in BASE M16; do
rm -f /tmp/st_x.bin
if [ $X = BASE ]; then unset ORR_HIP_LIB; else export ORR_HIP_LIB=$PWD/tools/x_m16.so; fi
ORR_SCREEN_STAMPS=/tmp/st_x.bin timeout -k 10 120 python bench.py --no-legs --no-cpu-baseline --rows-per-gpu 1000000 --batch 256 --steps 2 --warmup 1 > /tmp/x.json 2>/tmp/x.err || echo "bench rc!=0 ($X)"
echo "== $X"; python tools/analyze_stamps.py /tmp/st_x.bin 0 | grep -E "K loop|epilogue cycles|clock"
python tools/analyze_stamps.py /tmp/st_x.bin 1 | grep -E "K loop|epilogue cycles|clock"
done
