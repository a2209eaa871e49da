TAG=${1:-pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/q1
timeout 300 rocprofv3 --kernel-include-regex "cc_k_" --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d /tmp/q1 -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-overlap --steps 2 --warmup 1 --db-scans 5000 --workload ${WL:-kitti} > /dev/null 2> $OUT/pmc.err
F=$(find /tmp/q1 -name "*counter_collection.csv" | head -1)
python - <<PY
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
grid={}
rows=list(csv.DictReader(open("$F")))
for r in rows:
    k=r["Kernel_Name"].split("(")[0].replace("void ","").split("<")[0]
    g=int(r["Grid_Size"]); grid[k]=max(grid.get(k,0),g)
for r in rows:
    k=r["Kernel_Name"].split("(")[0].replace("void ","").split("<")[0]
    if int(r["Grid_Size"])!=grid[k]: continue
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out=open("$OUT/sq_summary.txt","w")
names=["SQ_WAVE_CYCLES","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_LDS","SQ_INSTS_VALU","SQ_BUSY_CYCLES"]
print("kernel,launches,"+",".join(names),file=out)
for k,d in sorted(acc.items()):
    print(k+","+str(len(d.get(names[0],[])))+","+",".join("%.4g"%(sum(d[n])/len(d[n])) if d.get(n) else "" for n in names),file=out)
out.close()
print(open("$OUT/sq_summary.txt").read())
PY
tail -3 $OUT/pmc.err
