# usage: bash profiles/r6/job_ab_lib.sh <other lib> [reps] [workloads] -- bench lines with the product's library and with another build, alternating
LIB=$1; REPS=${2:-2}; WL=${3:-kitti}
for r in $(seq $REPS); do
  for which in product other; do
    for w in $WL; do
      if [ $which = other ]; then export CC_BENCH_LIB=$LIB; else unset CC_BENCH_LIB; fi
      timeout 600 python bench.py --no-cpu --no-extra --workload $w --steps 30 --warmup 3 2>/dev/null | grep '^{' > /tmp/ab.json
      python - <<PY
import json
d = json.load(open("/tmp/ab.json"))
k = d["roofline"]["kernels_ms_per_launch_isolated"]
print("$which $w %d scans/s  %.3f ms/step  isolated: K1 %.3f K2 %.3f knn %.3f check %.3f gmm %.3f" % (round(d["value"]), d["ms_per_step"], k["cc_k_rasterize"], k["cc_k_contours"], k["cc_k_knn"], k["cc_k_check"], k["cc_k_gmm"]))
PY
    done
  done
done
