for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-200; done
echo "== box2"; for i in 1 2; do VDETR_BWD_BOX=2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-200; done
