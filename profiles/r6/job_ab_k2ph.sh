# usage: bash profiles/r6/job_ab_k2ph.sh <lib> ...  -- K2 phase clocks of the product library and of other builds (sparse world: the fixed costs)
for LIB in product "$@"; do
  if [ $LIB = product ]; then unset CC_PROBE_LIB CC_AMD_LIB; else export CC_PROBE_LIB=$LIB CC_AMD_LIB=$LIB; fi
  echo "== $LIB"
  timeout 600 python profiles/k2_probe.py ${WORLDS:-sparse} 1024 3 2>&1 | grep -E "stage A|sub-phases|level loop|total"
done
