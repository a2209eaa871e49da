#!/bin/bash
# tuning aid: the drop-in loop under read-ahead depth / batch / evaluator look-ahead settings (4 laps = 4 096 scans, median of three runs)
#   each line: DB depth, evaluator look-ahead, DB batch, evaluator ingest batch, readers
out=gpurun_out/dropin_sweep8.txt; : > $out
for cfg in "16 32 8 8 4" "32 64 16 16 4" "16 48 8 16 4" "24 40 8 8 4" "32 64 16 8 4" "16 32 8 8 6"; do
  set -- $cfg
  CC_DB_READ_AHEAD_BATCH=$3 CC_EVAL_INGEST_BATCH=$4 CC_EVAL_READERS=$5 timeout 200 python profiles/r5/dropin_probe.py $1 $2 4 2>&1 | grep -v amdgpu.ids | sed "s/^/dbbatch=$3 ingestbatch=$4 readers=$5 /" >> $out
done
python - <<'PY'
import json,re
for line in open("gpurun_out/dropin_sweep8.txt"):
    i = line.find("{")
    if i < 0: continue
    d = json.loads(line[i:])
    print(line[:i].strip(), d["scans_per_s_runs"], round(d.get("scans_per_s_with_construction", 0)), d["seconds_per_call"])
PY
