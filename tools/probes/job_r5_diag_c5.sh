mkdir -p gpurun_out/r5j
python tools/probes/diag_c5_table_grad.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5j/diag_c5_table_grad.txt | cut -c1-700
