mkdir -p gpurun_out/r5b
python profiles/k2_probe.py sparse,kitti 1024 5 > gpurun_out/r5b/k2_probe.json 2> gpurun_out/r5b/k2_probe.err
AB="INGEST_CUS=7/8;INGEST_CUS=6/8;INGEST_CUS=5/8;INGEST_CUS=4/8;INGEST_CUS=x7/8;INGEST_CUS=x6/8;INGEST_CUS=x5/8;INGEST_CUS=x4/8;INGEST_CUS=-"
timeout 300 python bench.py --no-cpu --no-extra --steps 30 --warmup 3 --ab-env "$AB" > gpurun_out/r5b/ab_sparse.json 2> gpurun_out/r5b/ab_sparse.err
timeout 300 python bench.py --no-cpu --no-extra --workload kitti --steps 30 --warmup 3 --ab-env "$AB" > gpurun_out/r5b/ab_kitti.json 2> gpurun_out/r5b/ab_kitti.err
cat gpurun_out/r5b/k2_probe.json; grep -v amdgpu.ids gpurun_out/r5b/k2_probe.err; grep "ab-env\|ingest stream\|Error\|error" gpurun_out/r5b/ab_sparse.err gpurun_out/r5b/ab_kitti.err
