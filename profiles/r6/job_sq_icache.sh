# usage: bash profiles/r6/job_sq_icache.sh <tag> [world]  -- instruction-cache counters of the ingest kernels on one probe launch
TAG=${1:-r6ic}; W=${2:-kitti}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/qi
CC_PROBE_NOPHASES=1 timeout 300 rocprofv3 --kernel-include-regex "cc_k_contours|cc_k_rasterize" --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/qi -o r -- python $GRAFT_REPO_ROOT/profiles/k2_probe.py $W 1024 2 > /dev/null 2> $OUT/pmc.err
F=$(find /tmp/qi -name "*counter_collection.csv" | head -1)
python - <<PY
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$F")):
    k=r["Kernel_Name"].split("(")[0].replace("void ","")
    if int(r["Grid_Size"]) < 1024*512: continue
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/icache.txt","w") as out:
    for k,d in sorted(acc.items()):
        line = k[:40] + "  " + "  ".join("%s=%.4g" % (n, sum(v)/len(v)) for n,v in sorted(d.items()))
        print(line); print(line, file=out)
PY
tail -2 $OUT/pmc.err
