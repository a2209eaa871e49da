# usage: bash profiles/r5/job_sq.sh <tag> <workload>  -- SQ counters per kernel (non-overlapped run): how busy the SIMDs are and what the waves wait for
TAG=${1:-r5sq}; WL=${2:-sparse}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/q_$WL
timeout 400 rocprofv3 --kernel-include-regex "cc_k_" --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d /tmp/q_$WL -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extra --no-overlap --workload $WL --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_$WL.err
F=$(find /tmp/q_$WL -name "*counter_collection.csv" | head -1)
python - <<PY
import csv, collections, re
acc=collections.defaultdict(lambda: collections.defaultdict(list))
grid={}
rows=list(csv.DictReader(open("$F")))
def name(r):
    k=r["Kernel_Name"].replace("void ","")
    return re.sub(r"\(.*","",k)
for r in rows:
    k=name(r); g=int(r["Grid_Size"]); grid[k]=max(grid.get(k,0),g)
for r in rows:
    k=name(r)
    if int(r["Grid_Size"])!=grid[k]: continue
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out=open("$OUT/sq_summary_$WL.csv","w")
names=["SQ_WAVE_CYCLES","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_LDS","SQ_INSTS_VALU","SQ_BUSY_CYCLES"]
print("kernel,launches,"+",".join(names)+",valu_share_of_issue_slots,wait_share_of_wave_cycles",file=out)
for k,d in sorted(acc.items()):
    m={n:(sum(d[n])/len(d[n]) if d.get(n) else float('nan')) for n in names}
    print(k+","+str(len(d.get(names[0],[])))+","+",".join("%.4g"%m[n] for n in names)+",%.3f,%.3f"%(m["SQ_ACTIVE_INST_VALU"]/(m["SQ_BUSY_CYCLES"]*8) if m["SQ_BUSY_CYCLES"] else 0, m["SQ_WAIT_ANY"]/m["SQ_WAVE_CYCLES"] if m["SQ_WAVE_CYCLES"] else 0),file=out)
out.close()
print(open("$OUT/sq_summary_$WL.csv").read())
PY
tail -2 $OUT/pmc_$WL.err
