# usage: bash profiles/r6/job_step_trace.sh <tag> [world] -- kernel trace of a short bench run; the raw trace of the timed steps goes to gpurun_out/<tag>/trace.csv.gz
TAG=$1; W=${2:-kitti}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
R=$PWD
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o r -- python $R/bench.py --no-cpu --no-extra --workload $W --steps 8 --warmup 3 > $OUT/bench.json 2> $OUT/err.txt
F=$(ls -S $(find /tmp/kt -name "*kernel_trace.csv") | head -1)
python - <<PY
import csv, gzip
rows = list(csv.DictReader(open("$F")))
rows = [r for r in rows if r["Kernel_Name"].startswith("cc_k") or r["Kernel_Name"].startswith("void cc_k")]
keep = rows[-1400:]
with gzip.open("$OUT/trace.csv.gz", "wt") as f:
    w = csv.writer(f)
    w.writerow(["name", "queue", "start_ns", "end_ns", "grid", "wg"])
    for r in keep:
        w.writerow([r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Queue_Id", ""), r["Start_Timestamp"], r["End_Timestamp"], r["Grid_Size_X"], r["Workgroup_Size_X"]])
print(len(rows), "kernel records;", len(keep), "kept")
PY
grep '^{' $OUT/bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'])"
