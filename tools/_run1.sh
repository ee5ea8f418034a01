python -m pytest tests/test_gpu_attention.py -x -q -k "box or sorted or full_size" 2>&1 | tail -3
for v in 2 5; do VDETR_BWD_BOX=$v timeout 300 python tools/kernel_bench.py c2 2>&1 | grep bwd_us; done
