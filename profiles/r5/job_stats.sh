# usage: bash profiles/r5/job_stats.sh <tag> [workload]  -- rocprofv3 kernel stats of a NON-overlapped bench run (every kernel alone on the GPU)
TAG=${1:-r5s}; WL=${2:-kitti}
mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps_$WL
timeout 600 rocprofv3 --kernel-include-regex "cc_k_" --kernel-trace --stats --output-format csv -d /tmp/ps_$WL -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extra --no-overlap --workload $WL --steps 8 --warmup 2 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/$TAG/stats_$WL.err
S=$(find /tmp/ps_$WL -name "*kernel_stats.csv" | head -1)
cp "$S" $GRAFT_REPO_ROOT/gpurun_out/$TAG/kernel_stats_${WL}_no_overlap.csv
cd $GRAFT_REPO_ROOT
python - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/$TAG/kernel_stats_${WL}_no_overlap.csv")))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:26]:
    print("%-60s calls %5s  total %9.1f us  avg %8.1f us  %5.1f %%" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
