# usage: bash profiles/collect.sh <tag> [quick|std|full]   (run on the GPU box from the repo root; results under
# gpurun_out/<tag>/; std = without the 20 k-scan DB bench; the files are then copied to profiles/<tag>_*)
TAG=${1:-r2}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
HEAD=$(cat .git_head 2>/dev/null || echo unknown)
( time timeout 900 python -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.log 2>&1
( time timeout 600 python bench.py ) > $OUT/bench_default.out 2> $OUT/bench_default.err
grep '^{' $OUT/bench_default.out > $OUT/bench_line_default.json
if [ "$2" != "quick" ]; then
  timeout 600 python bench.py --no-cpu --workload dense --steps 6 --warmup 2 2> $OUT/bench_dense.err | grep '^{' > $OUT/bench_line_dense.json
  timeout 900 python bench.py --no-cpu --db-scans 50000 --steps 4 --warmup 1 2> $OUT/bench_db50k.err | grep '^{' > $OUT/bench_line_db50k.json
  if [ "$2" = "full" ]; then
    timeout 600 python bench.py --no-cpu --db-scans 20000 --steps 4 --warmup 1 2> $OUT/bench_db20k.err | grep '^{' > $OUT/bench_line_db20k.json
  fi
fi
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p1 /tmp/p2 /tmp/p3
timeout 600 rocprofv3 --kernel-include-regex "cc_k_" --kernel-trace --stats --output-format csv -d /tmp/p1 -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu 2> $OUT/prof_trace.err | grep '^{' > $OUT/bench_line_under_rocprof.json
timeout 600 rocprofv3 --kernel-include-regex "cc_k_" --pmc FETCH_SIZE --output-format csv -d /tmp/p2 -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 2 --warmup 1 > /dev/null 2> $OUT/prof_fetch.err
timeout 600 rocprofv3 --kernel-include-regex "cc_k_" --pmc WRITE_SIZE --output-format csv -d /tmp/p3 -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 2 --warmup 1 > /dev/null 2> $OUT/prof_write.err
S=$(find /tmp/p1 -name "*kernel_stats.csv" | head -1); T=$(find /tmp/p1 -name "*kernel_trace.csv" | head -1)
F=$(find /tmp/p2 -name "*counter_collection.csv" | head -1); W=$(find /tmp/p3 -name "*counter_collection.csv" | head -1)
cd $GRAFT_REPO_ROOT
python profiles/summarize.py ${TAG} $OUT/prof "$S" "$T" "$F" "$W" 1024 5000 sparse $HEAD > $OUT/summarize.log 2>&1
# last, if the budget allows: three lanes on eight hardware queues (no stream shares a queue)
GPU_MAX_HW_QUEUES=8 timeout 120 python bench.py --no-cpu --lanes 3 --steps 10 --warmup 3 2> /dev/null | grep '^{' > $OUT/bench_line_lanes3_hwq8.json
GPU_MAX_HW_QUEUES=8 timeout 120 python bench.py --no-cpu --steps 10 --warmup 3 2> /dev/null | grep '^{' > $OUT/bench_line_lanes2_hwq8.json
tail -3 $OUT/pytest_gpu.log; cat $OUT/bench_line_default.json | head -c 300; ls $OUT $OUT/prof
