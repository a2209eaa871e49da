# usage: bash profiles/r6/job_k5_trace.sh <tag> [world] [lib]  -- kernel trace of profiles/k5_probe.py (one chunk's query chain, 5 repeats): per-kernel average durations
TAG=$1; W=${2:-kitti}; LIB=$3
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
if [ -n "$LIB" ]; then export CC_AMD_LIB=$PWD/$LIB; fi
R=$PWD
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o r -- python $R/profiles/k5_probe.py $W 4 > $OUT/probe_$W.json 2> $OUT/trace.err
F=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open("$F")):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    acc[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open("$OUT/kernels_$W.txt", "w") as out:
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1][-4:])):
        if not k.startswith("cc_k"): continue
        last = v[-4:] if len(v) >= 4 else v     # the probe's 4 timed repeats are the last launches
        line = "%-40s n %5d  last4 avg %9.1f us  max %9.1f us" % (k[:40], len(v), sum(last) / len(last), max(last))
        print(line); print(line, file=out)
PY
