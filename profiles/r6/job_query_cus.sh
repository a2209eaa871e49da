# usage: bash profiles/r6/job_query_cus.sh "<spec> <spec> ..." [bench args] -- bench lines with the query lanes' streams masked to a subset of the CUs (CC_QUERY_CUS), "-" = unmasked
SPECS=$1; shift
for S in $SPECS; do
  CC_QUERY_CUS=$S timeout 600 python bench.py --no-cpu --no-extra --steps 30 --warmup 3 "$@" 2>/dev/null | grep '^{' > /tmp/ab.json
  python - <<PY
import json
d = json.load(open("/tmp/ab.json"))
k = d["roofline"]["kernels_ms_per_launch"]
print("CC_QUERY_CUS=$S %d scans/s  %.3f ms/step  in-step: K1 %.3f K2 %.3f knn %.3f check %.3f merge %.3f gmm %.3f" % (round(d["value"]), d["ms_per_step"], k["cc_k_rasterize"], k["cc_k_contours"], k["cc_k_knn"], k["cc_k_check"], k["cc_k_merge"], k["cc_k_gmm"]))
PY
done
