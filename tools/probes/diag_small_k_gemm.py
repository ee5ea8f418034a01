#!/usr/bin/env python3
"""Is the library GEMM of the query-position embedding's first layer ([256 x 6] x [6 x N], helpers.PointwiseConv1d) right at
every N?  GPU result against the CPU's, for the operand layouts the module produces."""
import torch

torch.manual_seed(0)
for N in (64, 256, 512, 1000, 1024, 2048, 4096):
    xyz = torch.rand(1, N, 6) * 8
    w = torch.randn(256, 6)
    x = xyz.transpose(1, 2)                       # [1, 6, N] view, as PositionEmbeddingLearned.forward builds it
    ref = torch.mm(w, x.reshape(6, N))
    xg, wg = xyz.cuda().transpose(1, 2), w.cuda()
    got = torch.mm(wg, xg.reshape(6, N)).cpu()
    got2 = torch.mm(wg, xg.contiguous().view(6, N)).cpu()
    got3 = torch.matmul(xyz.cuda()[0], wg.t()).t().cpu()
    got4 = torch.nn.functional.conv1d(xg.contiguous(), wg.unsqueeze(-1)).cpu()[0]
    print(f"N={N:5d}: mm(reshape of view) {float((got - ref).abs().max()):.3e}  mm(contiguous) {float((got2 - ref).abs().max()):.3e}  "
          f"x w^T {float((got3 - ref).abs().max()):.3e}  conv1d {float((got4 - ref).abs().max()):.3e}   |ref| {float(ref.abs().max()):.2f}")
