# usage (GPU box): bash profiles/r4/collect_sq2.sh <tag>: dynamic instruction mix per kernel (second SQ pass: executed VALU / SALU / LDS /
# VMEM instruction counts), single-stream run of the headline configuration; one --pmc pass, kernel trace only
TAG=${1:-sq2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/q2
timeout 300 rocprofv3 --kernel-include-regex "cc_k_" --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d /tmp/q2 -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extra --no-overlap --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc.err
F=$(find /tmp/q2 -name "*counter_collection.csv" | head -1)
python - <<PY
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
grid={}
rows=list(csv.DictReader(open("$F")))
def nm(r): return r["Kernel_Name"].split("(")[0].replace("void ","")
for r in rows:
    k=nm(r); g=int(r["Grid_Size"]); grid[k]=max(grid.get(k,0),g)
for r in rows:
    k=nm(r)
    if int(r["Grid_Size"])!=grid[k]: continue
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names=["SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_LDS","SQ_INSTS_VMEM_RD","SQ_INSTS_VMEM_WR","SQ_WAVE_CYCLES","SQ_WAIT_INST_ANY","SQ_BUSY_CYCLES"]
out=open("$OUT/sq2_summary.csv","w")
print("kernel,launches,"+",".join(names)+",salu_share_of_executed",file=out)
for k,d in sorted(acc.items()):
    v=[sum(d[n])/len(d[n]) if d.get(n) else 0.0 for n in names]
    tot=v[0]+v[1]+v[2]+v[3]+v[4]
    print(k+","+str(len(d.get(names[0],[])))+","+",".join("%.4g"%x for x in v)+",%.3f"%(v[1]/tot if tot else 0),file=out)
out.close()
print(open("$OUT/sq2_summary.csv").read())
PY
tail -3 $OUT/pmc.err
