cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-include-regex "cc_k_" --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/p2 -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-overlap --steps 2 --warmup 1 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r2e/pmc1.err
find /tmp/p2 -name "*counter_collection.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r2e/pmc1.csv \;
timeout 400 rocprofv3 --kernel-include-regex "cc_k_" --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES --output-format csv -d /tmp/p3 -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-overlap --steps 2 --warmup 1 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r2e/pmc2.err
find /tmp/p3 -name "*counter_collection.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r2e/pmc2.csv \;
ls -la $GRAFT_REPO_ROOT/gpurun_out/r2e
