# usage: bash profiles/r5/job_ab4.sh <other lib> [reps] -- four workloads, the product's library against another build, alternating
LIB=$1; REPS=${2:-2}
for r in $(seq $REPS); do
  for which in product other; do
    if [ $which = other ]; then export CC_BENCH_LIB=$LIB; else unset CC_BENCH_LIB; fi
    for w in sparse kitti seq dense; do
      st=30; [ $w = seq ] && st=6; [ $w = dense ] && st=8
      timeout 600 python bench.py --no-cpu --no-extra --workload $w --steps $st --warmup 3 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline'].get('kernels_ms_per_launch_isolated',{})
print('$which $w', round(d['value']), round(d['ms_per_step'],3), 'gmm', round(k.get('cc_k_gmm',0),3))"
    done
  done
done
