#!/usr/bin/env python3
"""Full-size eval forward on the GPU (one launch per op: VDETR_ROWBLOCK=0) and on the CPU oracle with forward hooks on the
decoder's modules: the first module whose output differs."""
import copy
import os
import sys

os.environ["VDETR_ROWBLOCK"] = "0"
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_model as TM  # noqa: E402

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
nq, npre, nl, npts = (int(x) for x in (argv[:4] + ["1024", "4096", "3", "40000"][len(argv):]))
model = TM._make_model(nq=nq, npre=npre, nl=nl).train("--train" in sys.argv)
TM._zero_dropout(model)
inp_cpu = TM._inputs(npts, 3, "cpu", 1)
gpu = copy.deepcopy(model).to("cuda")
inp = {k: ([t.detach().to("cuda") for t in v] if isinstance(v, list) else v.to("cuda")) for k, v in inp_cpu.items()}


def hook_all(m, store):
    hs = []
    for name, mod in m.named_modules():
        if name and name.count(".") <= 3 and ("decoder" in name):
            def f(mod_, args, out, name=name):
                o = out[0] if isinstance(out, (tuple, list)) else out
                if isinstance(o, torch.Tensor):
                    store.setdefault(name, []).append(o.detach().float().cpu())
                elif isinstance(o, dict):
                    for k, v in o.items():
                        if isinstance(v, torch.Tensor) and v.is_floating_point():
                            store.setdefault(name + "/" + k, []).append(v.detach().float().cpu())
            hs.append(mod.register_forward_hook(f))
    return hs


sg, sc = {}, {}
hook_all(gpu, sg)
with torch.no_grad():
    out_g = gpu(inp)
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
hook_all(model, sc)
with torch.no_grad():
    out_c = model(inp_cpu)
print("modules hooked:", len(sc), len(sg))
sg0 = out_g["aux_outputs"][0]["objectness_prob"].detach().cpu()
sc0 = out_c["aux_outputs"][0]["objectness_prob"].detach()
tg, tc = torch.topk(sg0, nq, dim=1)[1], torch.topk(sc0, nq, dim=1)[1]
print("ranks holding the same token:", int((tg == tc).sum()), "of", nq, "; distinct objectness values on the CPU side:", int(sc0.unique().numel()), "of", sc0.numel())
for name in sc:
    if name not in sg:
        print(f"{name}: only on the CPU side")
        continue
    for i, (a, b) in enumerate(zip(sg[name], sc[name])):
        if a.shape != b.shape:
            print(f"{name}[{i}]: shapes {tuple(a.shape)} vs {tuple(b.shape)}")
            continue
        d = float((a - b).abs().max())
        s = float(b.abs().max())
        flag = "  <<<<" if d > 1e-3 * max(s, 1e-3) else ""
        print(f"{name}[{i}]: max |diff| {d:.3e} of {s:.3e}{flag}")
