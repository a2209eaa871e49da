# usage: bash profiles/r6/job_k5.sh <tag> <lib or ""> [worlds]  -- profiles/k5_probe.py per world, output under gpurun_out/<tag>/
TAG=$1; LIB=$2; WL=${3:-kitti sparse}
mkdir -p gpurun_out/$TAG
for w in $WL; do
  if [ -n "$LIB" ]; then export CC_AMD_LIB=$PWD/$LIB; fi
  timeout 300 python profiles/k5_probe.py $w 4 > gpurun_out/$TAG/$w.json 2>> gpurun_out/$TAG/err.txt
done
tail -3 gpurun_out/$TAG/err.txt
