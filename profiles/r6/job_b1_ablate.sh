# usage: bash profiles/r6/job_b1_ablate.sh <tag> [world] -- stage B1's time with the kernel cut off after its n-th part (a -DCC_TUNE build, CC_ABLATE=n)
TAG=$1; W=${2:-kitti}
for a in 0 1 2 3 4 5 6; do
  export CC_ABLATE=$a
  bash profiles/r6/job_k5_trace.sh ${TAG}_$a $W profiles/r6/_libs/lib_tune.so | grep "check_b1<64" | sed "s/^/ablate $a $W: /"
done
