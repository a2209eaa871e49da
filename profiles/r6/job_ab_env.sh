# usage: bash profiles/r6/job_ab_env.sh "<VAR=val ...>" [reps] [workloads]  -- bench lines with and without an environment setting, alternating
SET="$1"; REPS=${2:-2}; WL=${3:-kitti}
for r in $(seq $REPS); do
  for which in base with; do
    for w in $WL; do
      if [ $which = with ]; then E="env $SET"; else E="env"; fi
      $E timeout 600 python bench.py --no-cpu --no-extra --workload $w --steps 30 --warmup 3 2>/dev/null | grep '^{' > /tmp/ab.json
      python - <<PY
import json
d = json.load(open("/tmp/ab.json"))
k = d["roofline"]["kernels_ms_per_launch_isolated"]
print("$which [$SET] $w %d scans/s  %.3f ms/step  isolated: K1 %.3f K2 %.3f knn %.3f check %.3f gmm %.3f" % (round(d["value"]), d["ms_per_step"], k["cc_k_rasterize"], k["cc_k_contours"], k["cc_k_knn"], k["cc_k_check"], k["cc_k_gmm"]))
PY
    done
  done
done
