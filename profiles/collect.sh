# usage: bash profiles/collect.sh <tag> [quick|std|full]   (run on the GPU box from the repo root; results under
# gpurun_out/<tag>/; the files are then copied to profiles/<tag>_*)
#   quick: pytest + the default bench line            std: + seq / 50k lines + kernel trace + FETCH/WRITE PMC at the headline config
#   full : + FETCH/WRITE PMC at the 50 000-scan DB (the tiled K3)
TAG=${1:-r3}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
HEAD=$(cat .git_head 2>/dev/null || echo unknown)
( time timeout 1200 python -m pytest tests -m gpu -q -s ) > $OUT/pytest_gpu.log 2>&1
( time timeout 900 python bench.py ) > $OUT/bench_default.out 2> $OUT/bench_default.err
grep '^{' $OUT/bench_default.out > $OUT/bench_line_default.json
if [ "$2" != "quick" ]; then
  timeout 600 python bench.py --workload seq 2> $OUT/bench_seq.err | grep '^{' > $OUT/bench_line_seq.json
  timeout 600 python bench.py --no-cpu --no-extra --workload sparse --db-scans 50000 --steps 8 --warmup 2 2> $OUT/bench_db50k.err | grep '^{' > $OUT/bench_line_db50k.json
  timeout 600 python bench.py --no-cpu --no-extra --workload sparse --db-scans 20000 --steps 8 --warmup 2 2> $OUT/bench_db20k.err | grep '^{' > $OUT/bench_line_db20k.json
  timeout 600 python bench.py --no-cpu --no-extra --workload dense --steps 6 --warmup 2 2> $OUT/bench_dense.err | grep '^{' > $OUT/bench_line_dense.json
  timeout 600 python bench.py --no-cpu --no-extra --workload sparse --steps 16 --warmup 2 2> $OUT/bench_sparse.err | grep '^{' > $OUT/bench_line_sparse.json
  # a long timed region (100 steps = 102 400 scans, ~0.23 s): the figure that does not depend on where eight steps happen to end
  timeout 600 python bench.py --no-cpu --no-extra --steps 100 --warmup 4 2> $OUT/bench_steps100.err | grep '^{' > $OUT/bench_line_steps100.json
  timeout 600 python bench.py --no-cpu --no-extra --workload sparse --steps 100 --warmup 4 2> $OUT/bench_sparse100.err | grep '^{' > $OUT/bench_line_sparse_steps100.json
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/p1 /tmp/p2 /tmp/p3 /tmp/p4 /tmp/p5
  timeout 600 rocprofv3 --kernel-include-regex "cc_k_" --kernel-trace --stats --output-format csv -d /tmp/p1 -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extra 2> $OUT/prof_trace.err | grep '^{' > $OUT/bench_line_under_rocprof.json
  timeout 600 rocprofv3 --kernel-include-regex "cc_k_" --pmc FETCH_SIZE --output-format csv -d /tmp/p2 -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extra --steps 2 --warmup 1 > /dev/null 2> $OUT/prof_fetch.err
  timeout 600 rocprofv3 --kernel-include-regex "cc_k_" --pmc WRITE_SIZE --output-format csv -d /tmp/p3 -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extra --steps 2 --warmup 1 > /dev/null 2> $OUT/prof_write.err
  S=$(find /tmp/p1 -name "*kernel_stats.csv" | head -1); T=$(find /tmp/p1 -name "*kernel_trace.csv" | head -1)
  F=$(find /tmp/p2 -name "*counter_collection.csv" | head -1); W=$(find /tmp/p3 -name "*counter_collection.csv" | head -1)
  cd $GRAFT_REPO_ROOT
  python profiles/summarize.py ${TAG} $OUT/prof "$S" "$T" "$F" "$W" 1024 5000 kitti $HEAD > $OUT/summarize.log 2>&1
  # the same three passes on the sparse world of rounds 1-5 (5 k-scan DB): its kernel mix is a different one (K4 leads)
  cd /tmp
  rm -rf /tmp/k1 /tmp/k2 /tmp/k3
  timeout 600 rocprofv3 --kernel-include-regex "cc_k_" --kernel-trace --stats --output-format csv -d /tmp/k1 -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extra --workload sparse > /dev/null 2> $OUT/prof_trace_sparse.err
  timeout 600 rocprofv3 --kernel-include-regex "cc_k_" --pmc FETCH_SIZE --output-format csv -d /tmp/k2 -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extra --workload sparse --steps 2 --warmup 1 > /dev/null 2> $OUT/prof_fetch_sparse.err
  timeout 600 rocprofv3 --kernel-include-regex "cc_k_" --pmc WRITE_SIZE --output-format csv -d /tmp/k3 -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extra --workload sparse --steps 2 --warmup 1 > /dev/null 2> $OUT/prof_write_sparse.err
  S=$(find /tmp/k1 -name "*kernel_stats.csv" | head -1); T=$(find /tmp/k1 -name "*kernel_trace.csv" | head -1)
  F=$(find /tmp/k2 -name "*counter_collection.csv" | head -1); W=$(find /tmp/k3 -name "*counter_collection.csv" | head -1)
  cd $GRAFT_REPO_ROOT
  python profiles/summarize.py ${TAG}_sparse $OUT/prof_sparse "$S" "$T" "$F" "$W" 1024 5000 sparse $HEAD > $OUT/summarize_sparse.log 2>&1
  if [ "$2" = "full" ]; then
    cd /tmp
    timeout 600 rocprofv3 --kernel-include-regex "cc_k_knn" --kernel-trace --stats --output-format csv -d /tmp/p6 -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extra --workload sparse --db-scans 50000 --steps 2 --warmup 1 > /dev/null 2> $OUT/prof_trace50k.err
    timeout 600 rocprofv3 --kernel-include-regex "cc_k_knn" --pmc FETCH_SIZE --output-format csv -d /tmp/p4 -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extra --workload sparse --db-scans 50000 --steps 2 --warmup 1 > /dev/null 2> $OUT/prof_fetch50k.err
    timeout 600 rocprofv3 --kernel-include-regex "cc_k_knn" --pmc WRITE_SIZE --output-format csv -d /tmp/p5 -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extra --workload sparse --db-scans 50000 --steps 2 --warmup 1 > /dev/null 2> $OUT/prof_write50k.err
    S=$(find /tmp/p6 -name "*kernel_stats.csv" | head -1); T=$(find /tmp/p6 -name "*kernel_trace.csv" | head -1)
    F=$(find /tmp/p4 -name "*counter_collection.csv" | head -1); W=$(find /tmp/p5 -name "*counter_collection.csv" | head -1)
    cd $GRAFT_REPO_ROOT
    python profiles/summarize.py ${TAG}_db50k $OUT/prof50k "$S" "$T" "$F" "$W" 1024 50000 sparse $HEAD > $OUT/summarize50k.log 2>&1
  fi
fi
tail -3 $OUT/pytest_gpu.log; cat $OUT/bench_line_default.json | head -c 300; ls $OUT $OUT/prof 2>/dev/null
