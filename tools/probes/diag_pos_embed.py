#!/usr/bin/env python3
"""PositionEmbeddingLearned (helpers.py) alone, GPU vs CPU, on the inputs the full-size model feeds it: is it the module or its
input that differs?"""
import copy
import os
import sys

os.environ["VDETR_ROWBLOCK"] = "0"
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_model as TM  # noqa: E402

train = "--train" in sys.argv
model = TM._make_model(nq=1024, npre=4096, nl=3).train(train)
TM._zero_dropout(model)
gpu = copy.deepcopy(model).to("cuda")
inp_cpu = TM._inputs(40000, 3, "cpu", 1)
inp = {k: ([t.detach().to("cuda") for t in v] if isinstance(v, list) else v.to("cuda")) for k, v in inp_cpu.items()}
cap = {}


def grab(side):
    def f(mod, args, out):
        cap[side] = (args[0].detach().cpu().clone(), out.detach().float().cpu().clone())
    return f


hg = gpu.decoder.query_pos_projection[0].register_forward_hook(grab("gpu"))
with torch.no_grad():
    out_g = gpu(inp)
hg.remove()
xin, yg = cap["gpu"]
print("input to query_pos_projection[0]:", tuple(xin.shape), "range", float(xin.min()), float(xin.max()))
# the same module, same input, on the CPU copy and again on the GPU copy
mc = model.decoder.query_pos_projection[0]
mg = gpu.decoder.query_pos_projection[0]
with torch.no_grad():
    yc = mc(xin)
    yg2 = mg(xin.cuda()).float().cpu()
print("GPU (in model) vs CPU module on the GPU's input:", float((yg - yc).abs().max()), "of", float(yc.abs().max()))
print("GPU module alone vs CPU module, same input    :", float((yg2 - yc).abs().max()))
bn = mc.position_embedding_head[1]
print("BN running_var min / max:", float(bn.running_var.min()), float(bn.running_var.max()), " weight max", float(mc.position_embedding_head[0].weight.abs().max()),
      float(mc.position_embedding_head[3].weight.abs().max()))
# sensitivity: perturb the input by 1e-6 relative
with torch.no_grad():
    yc2 = mc(xin * (1 + 1e-6))
print("CPU module, input x (1 + 1e-6): output moves by", float((yc2 - yc).abs().max()))

# ---- the same hook on the CPU oracle side: where do the two inputs differ? ------------------------------------------------
import vdetr_amd.attention as A  # noqa: E402
import vdetr_amd.pointnet2_utils as PU  # noqa: E402
from conftest import _OracleExt  # noqa: E402
from oracle.attention_oracle import fused_attention_reference  # noqa: E402
import vdetr_amd.box_decode as BD  # noqa: E402
from oracle.box_oracle import decode_boxes_reference  # noqa: E402
import vdetr_amd.add_ln as ALN  # noqa: E402
from oracle import add_ln_oracle  # noqa: E402
A.fused_attention, A.begin_step, A.current_rng = fused_attention_reference, (lambda d: None), (lambda d: None)
PU._ext = _OracleExt()
BD.decode_boxes = decode_boxes_reference
ALN.layer_norm, ALN.add_dropout_layer_norm = add_ln_oracle.layer_norm, add_ln_oracle.add_dropout_layer_norm
hc = mc.register_forward_hook(grab("cpu"))
with torch.no_grad():
    out_c = model(inp_cpu)
hc.remove()
xc = cap["cpu"][0]
d = (xin - xc).abs()
rows = (d.amax(-1)[0] > 1e-4).nonzero().flatten()
print("inputs differ in", int(rows.numel()), "of", xin.shape[1], "rows; max", float(d.max()))
for r in rows[:6].tolist():
    print("  rank", r, "gpu", [round(v, 4) for v in xin[0, r].tolist()], "cpu", [round(v, 4) for v in xc[0, r].tolist()])
og = out_g["aux_outputs"][0]
oc = out_c["aux_outputs"][0]
tg = torch.topk(og["objectness_prob"], 1024, dim=1)[1].cpu()
tc = torch.topk(oc["objectness_prob"], 1024, dim=1)[1]
print("top-k on each side's own device: same at", int((tg == tc).sum()), "ranks")
for k in ("center_unnormalized", "size_unnormalized"):
    g_ = torch.gather(og[k].cpu(), 1, tg[..., None].expand(-1, -1, 3))
    c_ = torch.gather(oc[k], 1, tc[..., None].expand(-1, -1, 3))
    print(k, "gathered by each side's top-k: max diff", float((g_ - c_).abs().max()), "; vs the hooked input:",
          float((g_ - xin[..., :3] if k.startswith("center") else g_ - xin[..., 3:]).abs().max()))
