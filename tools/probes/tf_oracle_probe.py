import sys, os, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from oracle.attention_oracle import fused_attention_reference
from vdetr_amd import attention as A
dev = "cuda"
for dt in (torch.float32, torch.float64):
    for impl in ("grid_sample", "explicit"):
        B, nQ, nK = 1, 1024, 4096
        g = torch.Generator().manual_seed(0)
        xyz = (1 + torch.rand((B, nK, 3), generator=g) * torch.tensor([8.0, 6.0, 3.0])).to(dev, dt)
        verts = (xyz[:, :nQ, None, :] + torch.rand((B, nQ, 8, 3), generator=g).to(dev, dt) - 0.5)
        q = torch.randn((B, nQ, 256), generator=g).to(dev, dt).requires_grad_(True)
        k = torch.randn((B, nK, 64), generator=g).to(dev, dt).requires_grad_(True)
        v = torch.randn((B, nK, 64), generator=g).to(dev, dt).requires_grad_(True)
        tab = torch.randn((8, 10, 10, 10, 4), generator=g).to(dev, dt).requires_grad_(True)
        torch.cuda.synchronize(); t0 = time.time()
        out = fused_attention_reference(q, k, v, num_heads=4, scale=0.125, shared_kv=True, table=tab, rpe=A.RPEConfig(), vertices=verts, xyz=xyz, rpe_impl=impl)
        torch.cuda.synchronize(); t1 = time.time()
        out.sum().backward()
        torch.cuda.synchronize(); t2 = time.time()
        print(dt, impl, f"fwd {t1-t0:.2f}s bwd {t2-t1:.2f}s peak {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
        torch.cuda.reset_peak_memory_stats()
