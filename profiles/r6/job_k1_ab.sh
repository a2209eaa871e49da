# usage: bash profiles/r6/job_k1_ab.sh <other lib> -- K1/K2 probe (times + descriptor digests) with the product's library and another build, then bench A/B
LIB=$1
mkdir -p gpurun_out/k1ab
for r in 1 2; do
  for which in product other; do
    if [ $which = other ]; then export CC_PROBE_LIB=$LIB; else unset CC_PROBE_LIB; fi
    CC_PROBE_NOPHASES=1 timeout 600 python profiles/k2_probe.py kitti,sparse 1024 5 2>/dev/null > /tmp/probe.json
    python - <<PY
import json
for l in open("/tmp/probe.json"):
    d = json.loads(l)
    print("$which", d["workload"], "k1 %.3f k2 %.3f" % (d["k1_ms"], d["k2_ms"]), d["digest"], d["digest_keys"])
PY
  done
done
unset CC_PROBE_LIB
bash profiles/r6/job_ab_lib.sh $LIB 2 kitti
