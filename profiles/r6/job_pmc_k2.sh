# usage: bash profiles/r6/job_pmc_k2.sh <tag> [world]  -- FETCH_SIZE / WRITE_SIZE of the ingest kernels on one 1 024-scan probe launch (two passes)
TAG=${1:-r6pmc}; W=${2:-kitti}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/q$C
  CC_PROBE_NOPHASES=1 timeout 300 rocprofv3 --kernel-include-regex "cc_k_contours|cc_k_rasterize" --pmc $C --output-format csv -d /tmp/q$C -o r -- python $GRAFT_REPO_ROOT/profiles/k2_probe.py $W 1024 2 > /dev/null 2> $OUT/pmc_$C.err
  F=$(find /tmp/q$C -name "*counter_collection.csv" | head -1)
  python - <<PY
import csv, collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open("$F")):
    k=r["Kernel_Name"].split("(")[0].replace("void ","")
    if int(r["Grid_Size"]) < 1024*512 and "mid" not in k and "big" not in k: continue
    acc[k].append(float(r["Counter_Value"]))
with open("$OUT/pmc_$C.txt","w") as out:
    for k,v in sorted(acc.items()):
        line = "%-40s $C mean %.1f KB per launch over %d launches (counter unit: KB; FETCH_SIZE x2 on gfx950 for wide streaming reads)" % (k[:40], sum(v)/len(v), len(v))
        print(line); print(line, file=out)
PY
done
