import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
t = torch.arange(8, dtype=torch.float32, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.AVG)
torch.cuda.synchronize()
print("nccl AVG ok", t.tolist(), dist.get_backend())
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    dist.all_reduce(t, op=dist.ReduceOp.AVG)
torch.cuda.synchronize()
dist.destroy_process_group()
