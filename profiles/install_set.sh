# usage: bash profiles/install_set.sh <tag>  -- copy the files `collect.sh <tag> full` left under gpurun_out/<tag>/ to profiles/r6_* (the committed set)
T=$1
for f in gpurun_out/$T/prof/${T}_*; do b=$(basename $f); cp $f profiles/${b/${T}_/r6_}; done
for f in gpurun_out/$T/prof_sparse/*; do b=$(basename $f); cp $f profiles/${b/${T}_sparse_/r6_sparse_}; done
for f in gpurun_out/$T/prof50k/*; do b=$(basename $f); cp $f profiles/${b/${T}_db50k_/r6_db50k_}; done
for n in db20k db50k default dense seq sparse sparse_steps100 steps100 under_rocprof; do cp gpurun_out/$T/bench_line_$n.json profiles/r6_bench_line_$n.json; done
cp gpurun_out/$T/bench_line_default.json profiles/r6_bench_line_steps20_driver_args.json
cp gpurun_out/$T/pytest_gpu.log profiles/r6_pytest_gpu.log
grep -n "passed\|failed" profiles/r6_pytest_gpu.log | tail -1
for f in default seq db50k db20k dense sparse steps100 sparse_steps100 under_rocprof; do python - <<PY
import json
d = json.load(open("profiles/r6_bench_line_$f.json"))
print("$f", round(d["value"]), round(d["ms_per_step"], 3))
PY
done
