# usage: bash profiles/r6/job_sq_k2.sh <tag> [world]  -- SQ counters of the ingest kernels on one 1 024-scan probe launch (two passes)
TAG=${1:-r6sq}; W=${2:-kitti}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P=1
for CTRS in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES" \
            "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  rm -rf /tmp/q$P
  CC_PROBE_NOPHASES=1 timeout 300 rocprofv3 --kernel-include-regex "cc_k_contours|cc_k_rasterize" --pmc $CTRS --output-format csv -d /tmp/q$P -o r -- python $GRAFT_REPO_ROOT/profiles/k2_probe.py $W 1024 2 > /dev/null 2> $OUT/pmc$P.err
  F=$(find /tmp/q$P -name "*counter_collection.csv" | head -1)
  python - <<PY
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
rows=list(csv.DictReader(open("$F")))
for r in rows:
    k=r["Kernel_Name"].split("(")[0].replace("void ","")
    if int(r["Grid_Size"]) < 1024*512 and "mid" not in k and "big" not in k: continue
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/sq_pass$P.txt","w") as out:
    for k,d in sorted(acc.items()):
        line = k[:40] + "  " + "  ".join("%s=%.4g" % (n, sum(v)/len(v)) for n,v in sorted(d.items()))
        print(line); print(line, file=out)
PY
  P=$((P+1))
done
