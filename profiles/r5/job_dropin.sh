#!/bin/bash
# tuning aid: the drop-in loop under read-ahead depth / batch / evaluator look-ahead settings
#   each line: DB depth, evaluator look-ahead, DB batch, evaluator ingest batch, readers, spin us
out=gpurun_out/dropin_sweep7.txt; : > $out
for cfg in "0 4 1 1 1 0" "16 32 8 8 4 4000" "16 32 8 8 4 0" "16 32 8 8 3 0" "16 32 8 8 6 0" "24 40 8 8 4 0" "16 32 8 8 4 4000"; do
  set -- $cfg
  CC_EVAL_SPIN_US=$6 CC_DB_READ_AHEAD_BATCH=$3 CC_EVAL_INGEST_BATCH=$4 CC_EVAL_READERS=$5 timeout 120 python profiles/r5/dropin_probe.py $1 $2 2>&1 | grep -v amdgpu.ids | sed "s/^/dbbatch=$3 ingestbatch=$4 readers=$5 spin=$6 /" >> $out
done
cut -c1-1500 $out
