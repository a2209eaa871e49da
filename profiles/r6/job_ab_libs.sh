# usage: bash profiles/r6/job_ab_libs.sh "<lib> <lib> ..." [reps] [bench args] -- bench lines with the product's library and other builds, alternating
LIBS=$1; REPS=${2:-2}; shift; shift
for r in $(seq $REPS); do
  for LIB in product $LIBS; do
    if [ $LIB = product ]; then unset CC_BENCH_LIB; else export CC_BENCH_LIB=$LIB; fi
    timeout 600 python bench.py --no-cpu --no-extra --steps 30 --warmup 3 "$@" 2>/dev/null | grep '^{' > /tmp/ab.json
    python - <<PY
import json
d = json.load(open("/tmp/ab.json"))
k = d["roofline"]["kernels_ms_per_launch_isolated"]
print("$LIB %d scans/s  %.3f ms/step  isolated: K1 %.3f K2 %.3f knn %.3f check %.3f merge %.3f gmm %.3f" % (round(d["value"]), d["ms_per_step"], k["cc_k_rasterize"], k["cc_k_contours"], k["cc_k_knn"], k["cc_k_check"], k["cc_k_merge"], k["cc_k_gmm"]))
PY
  done
done
