# usage: bash profiles/r6/job_ab_k2.sh <lib> [<lib> ...]  -- isolated K2 time + descriptor digests of the product library and of other builds
for LIB in product "$@"; do
  if [ $LIB = product ]; then unset CC_PROBE_LIB CC_AMD_LIB; else export CC_PROBE_LIB=$LIB CC_AMD_LIB=$LIB; fi
  CC_PROBE_NOPHASES=1 timeout 600 python profiles/k2_probe.py ${WORLDS:-kitti,sparse} 1024 5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('$LIB', d['workload'], 'k1 %.3f k2 %.3f' % (d['k1_ms'], d['k2_ms']), 'flagged', d['flagged'], d['digest'], d['digest_keys'])
"
done
