#!/bin/bash
# Runs on the GPU box: every bench configuration once, so that TunableOp has measured every GEMM shape of the current tree, then
# writes the accumulated results to gpurun_out/gfx950_tunableop.csv (copy to tuning/ and commit: fresh processes then start
# from these picks instead of tuning new shapes inside their warm-up).
F="--no-cpu-baseline --steps 6 --warmup 3"
python bench.py $F > /dev/null 2>&1
for c in c1 c4 c5; do python bench.py $F --config $c --no-criterion-leg --no-exact-leg --no-backbone-leg > /dev/null 2>&1; done
python bench.py $F --force-dist --no-criterion-leg --no-exact-leg --no-backbone-leg > /dev/null 2>&1
mkdir -p gpurun_out
VDETR_TUNABLEOP_SAVE=$PWD/gpurun_out/gfx950_tunableop.csv python bench.py $F --no-criterion-leg --no-exact-leg --no-backbone-leg --no-roofline > /dev/null 2>&1
wc -l gpurun_out/gfx950_tunableop.csv
