# usage: bash profiles/r5/job_quick.sh <tag>  -- pytest -m gpu + the headline and KITTI-shaped bench lines (no CPU leg, no extras)
TAG=${1:-r5q}
mkdir -p gpurun_out/$TAG
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > gpurun_out/$TAG/pytest_gpu.log 2>&1
timeout 600 python bench.py --no-cpu --no-extra --workload sparse --steps 30 --warmup 3 2> gpurun_out/$TAG/bench_sparse.err | grep '^{' > gpurun_out/$TAG/bench_sparse.json
timeout 600 python bench.py --no-cpu --no-extra --workload kitti --steps 30 --warmup 3 2> gpurun_out/$TAG/bench_kitti.err | grep '^{' > gpurun_out/$TAG/bench_kitti.json
tail -5 gpurun_out/$TAG/pytest_gpu.log
python - <<PY
import json
for w in ("sparse", "kitti"):
    try:
        d = json.load(open("gpurun_out/$TAG/bench_%s.json" % w))
        print(w, round(d["value"]), "scans/s", d["ms_per_step"], "ms/step; isolated:", {k: round(v, 3) for k, v in d["roofline"]["kernels_ms_per_launch_isolated"].items()})
    except Exception as e:
        print(w, "failed", e)
PY
