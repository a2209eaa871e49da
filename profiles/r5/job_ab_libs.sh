# usage: bash profiles/r5/job_ab_libs.sh <lib>...  -- headline and KITTI-shaped bench lines: the product's library, then each other build
for LIB in product "$@"; do
  for w in sparse kitti; do
    if [ $LIB = product ]; then unset CC_BENCH_LIB; else export CC_BENCH_LIB=$LIB; fi
    timeout 600 python bench.py --no-cpu --no-extra --workload $w --steps 30 --warmup 3 2>/dev/null | grep '^{' > /tmp/ab.json
    python - <<PY
import json
d = json.load(open("/tmp/ab.json"))
k = d["roofline"]["kernels_ms_per_launch_isolated"]
print("$LIB $w %d scans/s  %.3f ms/step  isolated: gmm %.3f check %.3f contours %.3f knn %.3f" % (round(d["value"]), d["ms_per_step"], k["cc_k_gmm"], k["cc_k_check"], k["cc_k_contours"], k["cc_k_knn"]))
PY
  done
done
