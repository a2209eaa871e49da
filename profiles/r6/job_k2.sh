# usage: bash profiles/r6/job_k2.sh <tag> [worlds]  -- K2 probe (with and without phase clocks) + KITTI-shaped bench line
TAG=${1:-r6k2}
W=${2:-kitti,sparse}
mkdir -p gpurun_out/$TAG
CC_PROBE_NOPHASES=1 timeout 600 python profiles/k2_probe.py $W 1024 5 > gpurun_out/$TAG/probe_nophases.json 2> gpurun_out/$TAG/probe_nophases.err
timeout 600 python profiles/k2_probe.py $W 1024 3 > gpurun_out/$TAG/probe_phases.json 2> gpurun_out/$TAG/probe_phases.err
timeout 600 python bench.py --no-cpu --no-extra --workload kitti --steps 30 --warmup 3 2> gpurun_out/$TAG/bench_kitti.err | grep '^{' > gpurun_out/$TAG/bench_kitti.json
python - <<PY
import json
for f in ("probe_nophases", "probe_phases"):
    for l in open("gpurun_out/$TAG/%s.json" % f):
        d = json.loads(l)
        print(f, d["workload"], "k1 %.3f k2 %.3f" % (d["k1_ms"], d["k2_ms"]), "flagged", d["flagged"], "n_act", d["n_act_mean"], d["n_act_max"], "n_cont", [round(v) for v in d["n_cont_mean"]], d["digest"], d["digest_keys"], d["digest_without_keys"])
try:
    d = json.load(open("gpurun_out/$TAG/bench_kitti.json"))
    print("kitti", round(d["value"]), "scans/s", d["ms_per_step"], "ms/step; isolated:", {k: round(v, 3) for k, v in d["roofline"]["kernels_ms_per_launch_isolated"].items()})
except Exception as e:
    print("bench failed", e)
PY
grep -h "cc_k_contours" gpurun_out/$TAG/probe_phases.err | tail -12
