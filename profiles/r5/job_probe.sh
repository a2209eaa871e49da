# usage: bash profiles/r5/job_probe.sh <tag> [worlds] [ab]  -- K2 probe of the current build (with and without the phase clocks),
# with `ab` also round 4's kernels (profiles/r5/libcont2_r4.so)
TAG=${1:-r5p}; W=${2:-sparse,dense,kitti}
mkdir -p gpurun_out/$TAG
CC_PROBE_NOPHASES=1 python profiles/k2_probe.py $W 1024 8 > gpurun_out/$TAG/k2_probe_noclk.json 2> gpurun_out/$TAG/k2_probe_noclk.err
python profiles/k2_probe.py $W 1024 5 > gpurun_out/$TAG/k2_probe.json 2> gpurun_out/$TAG/k2_probe.err
if [ -f profiles/r5/libcont2_r4.so ] && [ "$3" = "ab" ]; then
  CC_PROBE_NOPHASES=1 CC_PROBE_LIB=profiles/r5/libcont2_r4.so python profiles/k2_probe.py $W 1024 8 > gpurun_out/$TAG/k2_probe_r4.json 2> gpurun_out/$TAG/k2_probe_r4.err
fi
grep -v amdgpu.ids gpurun_out/$TAG/k2_probe.err
python - <<PY
import json, os
for f in ("k2_probe_r4", "k2_probe_noclk", "k2_probe"):
    p = "gpurun_out/$TAG/%s.json" % f
    if os.path.exists(p):
        for l in open(p):
            d = json.loads(l)
            print("%-15s %-7s k1 %.4f  k2 %.4f ms  %s %s %s flagged %d" % (f, d["workload"], d["k1_ms"], d["k2_ms"], d["digest"], d["digest_keys"], d["digest_without_keys"], d["flagged"]))
PY
