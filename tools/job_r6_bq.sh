#!/bin/bash
# round 6: the LDS-tiled ball query and the LDS scatter gradients — parity, then the indexing bench under rocprofv3
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_pointnet2.py tests/test_capi.py -x -q 2>&1 | tail -5
bash tools/job_r6_bq2.sh
timeout 300 python tools/kernel_bench.py c2 --indexing 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_kernel_bench_indexing2.txt
grep -i "grad" gpurun_out/r06_kernel_bench_indexing2.txt
