# usage: bash profiles/r5/job_probe.sh <tag> [worlds]   -- K2 probe of the current build next to round 4's kernels (profiles/r5/libcont2_r4.so)
TAG=${1:-r5p}; W=${2:-sparse,dense,kitti}
mkdir -p gpurun_out/$TAG
python profiles/k2_probe.py $W 1024 5 > gpurun_out/$TAG/k2_probe.json 2> gpurun_out/$TAG/k2_probe.err
if [ -f profiles/r5/libcont2_r4.so ] && [ "$3" = "ab" ]; then
  CC_PROBE_LIB=profiles/r5/libcont2_r4.so python profiles/k2_probe.py $W 1024 5 > gpurun_out/$TAG/k2_probe_r4.json 2> gpurun_out/$TAG/k2_probe_r4.err
  echo "--- round-4 kernels"; cat gpurun_out/$TAG/k2_probe_r4.json
fi
echo "--- current"; cat gpurun_out/$TAG/k2_probe.json; grep -v amdgpu.ids gpurun_out/$TAG/k2_probe.err
