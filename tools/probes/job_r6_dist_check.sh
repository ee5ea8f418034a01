#!/bin/bash
# the N > 1 paths on the 1-GPU box with the round's final tree: a 1-rank RCCL communicator through the launcher, 2 gloo ranks, the other configs
out=gpurun_out/r6_dist; mkdir -p $out
F="--steps 20 --warmup 3 --no-cpu-baseline --no-criterion-leg --no-exact-leg --no-backbone-leg --no-roofline"
timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-dist $F > $out/torchrun_1rank.json 2> $out/torchrun_1rank.err; echo "torchrun 1 rank: exit $? $(tail -1 $out/torchrun_1rank.json | cut -c1-200)"
timeout 900 python3 bench.py --gpus 2 --backend gloo --steps 5 --warmup 3 --no-cpu-baseline --no-criterion-leg --no-exact-leg --no-backbone-leg --no-roofline > $out/gloo_2ranks.json 2> $out/gloo_2ranks.log; echo "gloo 2 ranks: exit $? $(tail -1 $out/gloo_2ranks.json | cut -c1-200)"
for c in c1 c4 c5; do timeout 600 python3 bench.py --config $c --no-cpu-baseline --no-criterion-leg --no-exact-leg --no-backbone-leg > $out/bench_$c.json 2> $out/bench_$c.err; echo "$c: exit $? $(tail -1 $out/bench_$c.json | cut -c1-160)"; done
