cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2i
timeout 400 python bench.py --no-cpu --tune-sweep "CC_B1_GRID=2048,6144,8192;CC_B2_GRID=2048,8192;CC_GMM_GRID=2048,8192" > gpurun_out/r2i/sweep.json 2> gpurun_out/r2i/sweep.err
GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --no-cpu --lanes 4 > gpurun_out/r2i/lanes4_q8.json 2> gpurun_out/r2i/lanes4_q8.err
GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --no-cpu --lanes 3 > gpurun_out/r2i/lanes3_q8.json 2> gpurun_out/r2i/lanes3_q8.err
GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --no-cpu > gpurun_out/r2i/lanes2_q8.json 2> gpurun_out/r2i/lanes2_q8.err
grep tune-sweep gpurun_out/r2i/sweep.err
