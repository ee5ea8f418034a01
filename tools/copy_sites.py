import os, sys
import torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, "/root/repo")
import bench
dev = torch.device("cuda")
model = bench.build_model("c2", dev)
inputs = bench.make_inputs("c2", dev, 0)
tr = bench.Trainer(model, inputs, 1, use_graph=False, overlap=False, fps_prefetch=False)
for _ in range(3):
    tr.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step()
    torch.cuda.synchronize()
from collections import defaultdict
agg = defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.name in ("aten::copy_", "aten::add_", "aten::add", "aten::mul", "aten::fill_", "aten::zero_", "aten::sum", "aten::cat") and e.kernels:
        st = [s for s in (e.stack or []) if "/root/repo" in s or "autograd" in s][:3]
        key = (e.name, " | ".join(s.split("/")[-1][:60] for s in st))
        agg[key][0] += len(e.kernels); agg[key][1] += sum(k.duration for k in e.kernels)
for (n, st), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{c:4d} {t:8.1f}  {n:12s} {st}")
