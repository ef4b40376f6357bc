mkdir -p gpurun_out/r2
( time timeout -k 10 1000 python -m pytest tests -m gpu -x -q ) > gpurun_out/r2/t_all.log 2>&1; tail -6 gpurun_out/r2/t_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
