python -m pytest tests/test_gpu_attention.py -x -q -k "rotated or full_size or box_backward" 2>&1 | tail -12
