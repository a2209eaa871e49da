cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
for f in tests/test_gpu_hints.py tests/test_gpu_hostcpp.py tests/test_gpu_ingest.py tests/test_gpu_properties.py tests/test_gpu_query.py; do
  b=$(basename $f .py)
  timeout 400 python -m pytest $f -m gpu -x -v > gpurun_out/r2b/$b.log 2>&1
  echo "$b rc=$?" >> gpurun_out/r2b/summary.txt
done
( time timeout 400 python bench.py --no-cpu --stats --workload dense --steps 4 --warmup 1 ) > gpurun_out/r2b/bench_dense.json 2> gpurun_out/r2b/bench_dense.err
cat gpurun_out/r2b/summary.txt
