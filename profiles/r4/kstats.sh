# usage (GPU box): bash profiles/r4/kstats.sh <tag> <bench args...>: rocprofv3 kernel stats of one bench run, cc_k_* kernels only
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pk_$TAG
timeout 400 rocprofv3 --kernel-include-regex "cc_k_" --kernel-trace --stats --output-format csv -d /tmp/pk_$TAG -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extra "$@" > /dev/null 2> /tmp/pk_$TAG.err
S=$(find /tmp/pk_$TAG -name "*kernel_stats.csv" | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/kstats
cp $S $GRAFT_REPO_ROOT/gpurun_out/kstats/$TAG.csv
python - "$S" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith(("cc_k", "void cc_k")):
        print("%-44s calls %4s avg %9.1f us max %9.1f us" % (r["Name"].replace("void ", "")[:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
