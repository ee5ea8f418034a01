"""Shared builders for the parity tests: the SAME module construction + deterministic fill that
oracle/make_golden.py applied to the reference's modules, applied to this repo's modules."""
from argparse import Namespace

import numpy as np
import torch

from oracle.param_fill import fill_module


def args_ns(**kw):
    a = dict(log_scale=512.0, rpe_quant="bilinear_4_10", angle_type="", rpe_dim=128, share_selfattn=False)
    a.update(kw)
    return Namespace(**a)


def t(a, device="cpu", grad=False):
    x = torch.from_numpy(np.asarray(a).copy()).to(device)
    return x.requires_grad_(True) if grad else x


def build_cross_attention(angle_type="", device="cpu"):
    from vdetr_amd.vdetr_transformer import GlobalShareCrossAttention
    mod = GlobalShareCrossAttention(256, 4, attn_drop=0.1, proj_drop=0.1, args=args_ns(angle_type=angle_type))
    fill_module(mod)
    with torch.no_grad():
        for m in mod.cpb_mlps:
            m[0].weight.mul_(2.0)
            m[2].weight.mul_(1.5)
    return mod.eval().to(device)


def build_share_self_attention(device="cpu"):
    from vdetr_amd.vdetr_transformer import ShareSelfAttention
    return fill_module(ShareSelfAttention(256, 4, dropout=0.1)).eval().to(device)


def build_decoder(dec_nlayers, share=False, device="cpu", num_queries=64):
    from vdetr_amd.dataset_config import ScannetDatasetConfig
    from vdetr_amd.vdetr_transformer import FFNLayer, GlobalDecoderLayer, TransformerDecoder
    a = args_ns(share_selfattn=share)
    first = FFNLayer(d_model=256, dim_feedforward=256, dropout=0.1)
    layer = GlobalDecoderLayer(d_model=256, nhead=4, dim_feedforward=256, dropout=0.1, pos_for_key=False, args=a)
    dec = TransformerDecoder(first, layer, ScannetDatasetConfig(), num_layers=dec_nlayers - 1, decoder_dim=256,
                             mlp_dropout=0.3, mlp_norm="bn1d", mlp_act="relu", mlp_sep=True, pos_for_key=False,
                             num_queries=num_queries, cls_loss="focalloss_0.25", is_bilable=True, q_content="random",
                             return_intermediate=True, args=a)
    fill_module(dec)
    with torch.no_grad():
        for l in dec.layers:
            for m in l.multihead_attn.cpb_mlps:
                m[0].weight.mul_(2.0)
                m[2].weight.mul_(1.5)
        for h in dec.mlp_heads:
            for k in ("center_head", "size_head"):
                h[k].layers[-1].weight.mul_(0.2)
    return dec.eval().to(device)


def grad_atol(g, key, frac, floor=2e-6):
    """Absolute tolerance of a fixture gradient: `frac` of its largest entry, at least `floor`.  A bias whose gradient is
    zero in exact arithmetic (the key projection's: every row of dS sums to zero, so sum_k dK[k] = 0) has no scale of its
    own — the reference's 3e-7 there is its fp32 summation noise.  The device sums the same terms from split-bf16 products
    (2^-17 each instead of 2^-24, attn_bwd_kv.hip): for `x.bias` the floor is 1e-5 of the largest `x.weight` gradient."""
    tol = max(frac * float(np.abs(g[key]).max()), floor)
    if key.endswith(".bias") and key[:-5] + ".weight" in g.files:
        tol = max(tol, 1e-5 * float(np.abs(g[key[:-5] + ".weight"]).max()))
    return tol


def assert_close(actual, expected, rtol, atol, what=""):
    actual = actual.detach().cpu().double().numpy() if isinstance(actual, torch.Tensor) else np.asarray(actual, np.float64)
    expected = np.asarray(expected, np.float64)
    assert actual.shape == expected.shape, f"{what}: shape {actual.shape} vs {expected.shape}"
    err = np.abs(actual - expected)
    tol = atol + rtol * np.abs(expected)
    if not (err <= tol).all():
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError(f"{what}: max abs err {err.max():.3e} (|ref| max {np.abs(expected).max():.3e}); worst at {i}: "
                             f"{actual[i]:.6e} vs {expected[i]:.6e}")


def run_cross_attention_case(g, mod, device):
    """Forward + backward of one golden cross-attention case; returns dict of results to compare."""
    angle_type = str(g["angle_type"])
    query, key = t(g["query"], device, True), t(g["key"], device, True)
    ref_pts, xyz = t(g["reference_point"], device), t(g["xyz"], device)
    angle = t(g["reference_angle"], device) if angle_type else None
    mod.return_attn = True
    x, attn = mod(query, key, ref_pts, angle, xyz)
    (x * t(g["wout"], device)).sum().backward()
    res = {"x": x, "attn": attn, "grad_query": query.grad, "grad_key": key.grad}
    for pname, p in mod.named_parameters():
        res["grad_param:" + pname] = p.grad
    return res


def run_decoder_case(g, dec, device):
    feats = t(g["feats"], device, True)
    xyz = t(g["xyz"], device)
    dims = [t(g["dims_min"], device), t(g["dims_max"], device)]
    enc = {"center_normalized": t(g["center_normalized"], device), "size_normalized": t(g["size_normalized"], device)}
    out, _ = dec(None, feats, xyz, xyz, dims, query_pos=xyz, enc_box_predictions=enc, enc_box_features=feats)
    stages = out["aux_outputs"] + [out["outputs"]]
    loss = 0
    for s, st in enumerate(stages):
        loss = loss + (st["sem_cls_logits"] * t(g[f"s{s}:w"], device)).sum() + st["center_normalized"].sum() \
            + st["size_normalized"].sum()
    loss.backward()
    return stages, loss, feats.grad


def fixed_salts(monkeypatch, base=0x7E570000):
    """Restart the three module-level counters that hand every dropout site its stream (add_ln / bn_act salts, the attention modules'
    salt) for the duration of a test.  A whole-step comparison of two evaluation orders is chaotic where a ReLU gate sits within
    rounding of zero; with the process's history in the salts, WHICH gates those are depended on the tests that ran before."""
    import itertools
    from vdetr_amd import add_ln, bn_act, vdetr_transformer
    monkeypatch.setattr(add_ln, "_salts", itertools.count(base + 0x1000))
    monkeypatch.setattr(bn_act, "_salts", itertools.count(base + 0x2000))
    monkeypatch.setattr(vdetr_transformer, "_salt_counter", itertools.count(base + 0x3000))
