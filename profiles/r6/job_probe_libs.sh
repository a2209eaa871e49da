# usage: bash profiles/r6/job_probe_libs.sh <lib>... -- K1/K2 probe (isolated times + descriptor digests) per library, two alternating rounds; "product" = the tree's own
for r in 1 2; do
  for LIB in product "$@"; do
    if [ $LIB = product ]; then unset CC_PROBE_LIB; else export CC_PROBE_LIB=$LIB; fi
    CC_PROBE_NOPHASES=1 timeout 600 python profiles/k2_probe.py kitti,sparse 1024 5 2>/dev/null > /tmp/probe.json
    python - <<PY
import json
for l in open("/tmp/probe.json"):
    d = json.loads(l)
    print("$LIB", d["workload"], "k1 %.3f k2 %.3f" % (d["k1_ms"], d["k2_ms"]), d["digest"], d["digest_keys"])
PY
  done
done
