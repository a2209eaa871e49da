cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2h
for f in tests/test_gpu_hints.py tests/test_gpu_hostcpp.py tests/test_gpu_scale.py tests/test_gpu_query.py; do
  b=$(basename $f .py)
  ( time timeout 600 python -m pytest $f -m gpu -x -q ) > gpurun_out/r2h/$b.log 2>&1
  echo "$b rc=$?" >> gpurun_out/r2h/summary.txt
done
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-overlap --steps 4 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/r2h/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2h/prof.err
find /tmp/p1 -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r2h/kernel_stats.csv \;
cat $GRAFT_REPO_ROOT/gpurun_out/r2h/summary.txt
