cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
timeout 300 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_query.py tests/test_gpu_properties.py -m gpu -x -q > gpurun_out/r2f/pytest.log 2>&1
tail -3 gpurun_out/r2f/pytest.log
CC_K2_PHASES=1 timeout 300 python bench.py --no-cpu --db-scans 256 --steps 2 --warmup 1 > /dev/null 2> gpurun_out/r2f/k2ph_sparse.err
CC_K2_PHASES=1 timeout 300 python bench.py --no-cpu --db-scans 256 --steps 2 --warmup 1 --workload dense > /dev/null 2> gpurun_out/r2f/k2ph_dense.err
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-overlap --steps 4 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/r2f/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2f/prof.err
find /tmp/p1 -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r2f/kernel_stats.csv \;
cd $GRAFT_REPO_ROOT
timeout 400 python bench.py --no-cpu > gpurun_out/r2f/bench_sparse.json 2> gpurun_out/r2f/bench_sparse.err
