python -m pytest tests/test_gpu_attention.py -x -q -k "box or sorted or full_size or key_side" 2>&1 | tail -8
for v in 2 4; do VDETR_BWD_BOX=$v timeout 300 python tools/kernel_bench.py c2 2>&1 | grep bwd_us; done
tools/b3_prof.sh
