cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
for f in tests/test_gpu_hints.py tests/test_gpu_hostcpp.py tests/test_gpu_ingest.py tests/test_gpu_properties.py tests/test_gpu_query.py; do
  b=$(basename $f .py)
  timeout 400 python -m pytest $f -m gpu -x -q > gpurun_out/r2c/$b.log 2>&1
  echo "$b rc=$?" >> gpurun_out/r2c/summary.txt
done
timeout 400 python bench.py --no-cpu --stats > gpurun_out/r2c/bench_sparse.json 2> gpurun_out/r2c/bench_sparse.err
timeout 500 python bench.py --no-cpu --stats --workload dense --steps 4 --warmup 1 > gpurun_out/r2c/bench_dense.json 2> gpurun_out/r2c/bench_dense.err
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-overlap --steps 4 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/r2c/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2c/prof.err
find /tmp/p1 -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r2c/kernel_stats.csv \;
cat $GRAFT_REPO_ROOT/gpurun_out/r2c/summary.txt
