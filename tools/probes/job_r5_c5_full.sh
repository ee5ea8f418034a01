mkdir -p gpurun_out/r5j; export TMPDIR=/tmp
s=$(date +%s)
python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "full_config and c5" 2>&1 | tail -40 > gpurun_out/r5j/fullcfg_c5.txt
echo "$(( $(date +%s) - s )) s"; tail -25 gpurun_out/r5j/fullcfg_c5.txt | cut -c1-400
