# usage: bash profiles/r6/job_env_sweep.sh VAR "<value> <value> ..." [bench args] -- bench lines with an environment variable set to each value ("-" = unset), two rounds
VAR=$1; VALS=$2; shift; shift
for r in 1 2; do
for V in $VALS; do
  if [ "$V" = "-" ]; then unset $VAR; else export $VAR=$V; fi
  timeout 600 python bench.py --no-cpu --no-extra --steps 30 --warmup 3 "$@" 2>/dev/null | grep '^{' > /tmp/ab.json
  python - <<PY
import json
d = json.load(open("/tmp/ab.json"))
k = d["roofline"]["kernels_ms_per_launch"]; i = d["roofline"]["kernels_ms_per_launch_isolated"]
print("$VAR=$V %d scans/s  %.3f ms/step  in-step: K1 %.3f K2 %.3f knn %.3f check %.3f gmm %.3f | isolated gmm %.3f check %.3f knn %.3f" % (round(d["value"]), d["ms_per_step"], k["cc_k_rasterize"], k["cc_k_contours"], k["cc_k_knn"], k["cc_k_check"], k["cc_k_gmm"], i["cc_k_gmm"], i["cc_k_check"], i["cc_k_knn"]))
PY
done
done
