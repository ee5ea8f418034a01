mkdir -p gpurun_out/r5j; export TMPDIR=/tmp
python -m pytest tests/test_gpu_model.py -x -q -s -m gpu -k "full_config and c5" 2>&1 | grep "full config\|passed\|failed" | cut -c1-500
