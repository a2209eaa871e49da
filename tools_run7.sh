cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
free -g | head -2 > gpurun_out/r2g/mem.txt; nproc >> gpurun_out/r2g/mem.txt; df -h /dev/shm | tail -1 >> gpurun_out/r2g/mem.txt
( time timeout 600 python bench.py ) > gpurun_out/r2g/bench_default.json 2> gpurun_out/r2g/bench_default.err
cat gpurun_out/r2g/mem.txt
