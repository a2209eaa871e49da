cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
true
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-overlap --steps 4 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/r2d/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2d/prof.err
find /tmp/p1 -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r2d/kernel_stats.csv \;
