"""CPU restatement (plain torch ops, any dtype) of the attention math on the hot path.

TEST INFRASTRUCTURE ONLY — nothing under ``v-detr_amd/`` may import this module.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it, as the checker.

Pinned against the reference itself: ``oracle/make_golden.py`` imports the reference's
``GlobalShareCrossAttention`` / ``ShareSelfAttention`` / ``TransformerDecoder`` (models/vdetr_transformer.py) in the
build container and stores inputs + outputs + gradients under ``tests/golden/``; ``tests/test_host_vs_reference.py``
checks this restatement (and the host modules built on it) against those vectors.  It deliberately does NOT call F.grid_sample: the trilinear
lookup is written out so that it is an independent statement of vdetr_transformer.py:710-731.
"""
import math

import torch


def rpe_bias_reference(tables, vertices, xyz, log_scale=512.0, max_value=4.0, cos_sin=None):
    """rpe[B,H,nQ,nK] = sum_i trilinear(T_i, g(P_i[q] - X[k]))          (vdetr_transformer.py:710-731)

    tables [8,T,T,T,H] (= cpb_mlps[i](relative_coords_table)), vertices [B,nQ,8,3], xyz [B,nK,3],
    cos_sin [B,nQ,2] for angle_type == "object_coords" (:712-720), else None.
    grid_sample semantics (bilinear, zeros padding, align_corners=False): pix = ((g+1)*T - 1)/2; the x
    component of the delta indexes the LAST table axis, y the middle one, z the first one.
    """
    T, H = tables.shape[1], tables.shape[-1]
    B, nQ = vertices.shape[:2]
    nK = xyz.shape[1]
    out = tables.new_zeros((B, nQ, nK, H))
    for i in range(8):
        d = vertices[:, :, None, i, :] - xyz[:, None, :, :]  # B,nQ,nK,3  (:711)
        if cos_sin is not None:
            # :713-720 (axis swap, right-multiply by roty(angle), swap back) is a yaw rotation of (dx, dy)
            c, s = cos_sin[..., 0][:, :, None], cos_sin[..., 1][:, :, None]
            d = torch.stack((d[..., 0] * c - d[..., 1] * s, d[..., 0] * s + d[..., 1] * c, d[..., 2]), dim=-1)
        g = torch.sign(d) * torch.log2(torch.abs(d) * log_scale + 1.0) / math.log2(8) / max_value  # :722-723
        pix = ((g + 1.0) * T - 1.0) / 2.0
        base = torch.floor(pix)
        frac = pix - base
        base = base.long()
        flat = tables[i].reshape(T * T * T, H)
        for cz in (0, 1):
            for cy in (0, 1):
                for cx in (0, 1):
                    ix, iy, iz = base[..., 0] + cx, base[..., 1] + cy, base[..., 2] + cz
                    w = ((frac[..., 0] if cx else 1 - frac[..., 0]) * (frac[..., 1] if cy else 1 - frac[..., 1])
                         * (frac[..., 2] if cz else 1 - frac[..., 2]))
                    ok = (ix >= 0) & (ix < T) & (iy >= 0) & (iy < T) & (iz >= 0) & (iz < T)
                    cell = (iz.clamp(0, T - 1) * T + iy.clamp(0, T - 1)) * T + ix.clamp(0, T - 1)
                    out = out + (w * ok)[..., None] * flat[cell]
    return out.permute(0, 3, 1, 2)


def rpe_bias_grid_sample(tables, vertices, xyz, log_scale=512.0, max_value=4.0, cos_sin=None):
    """The same bias composed exactly as the reference composes it on CPU — eight F.grid_sample passes
    (vdetr_transformer.py:710-731) — used where the CPU path is TIMED (bench.py cpu_baseline), so that the baseline
    costs what the reference's own CPU path costs.  tests/ check it equals rpe_bias_reference."""
    import torch.nn.functional as F
    B, nQ = vertices.shape[:2]
    nK = xyz.shape[1]
    rpe = 0
    for i in range(8):
        d = vertices[:, :, None, i, :] - xyz[:, None, :, :]
        if cos_sin is not None:
            c, s = cos_sin[..., 0][:, :, None], cos_sin[..., 1][:, :, None]
            d = torch.stack((d[..., 0] * c - d[..., 1] * s, d[..., 0] * s + d[..., 1] * c, d[..., 2]), dim=-1)
        d = torch.sign(d) * torch.log2(torch.abs(d) * log_scale + 1.0) / math.log2(8) / max_value
        tab = tables[i][None].permute(0, 4, 1, 2, 3)
        rpe = rpe + F.grid_sample(tab, d.reshape(1, 1, 1, -1, 3).to(tab.dtype), mode="bilinear", align_corners=False) \
            .reshape(-1, B, nQ, nK).permute(1, 0, 2, 3)
    return rpe


def fused_attention_reference(q, k, v, *, num_heads, scale, shared_kv, table=None, rpe=None, vertices=None,
                              xyz=None, cos_sin=None, attn_mask=None, dropout_p=0.0, rng_state=None, salt=0,
                              keep_mask=None, return_probs=False, rpe_impl="explicit", table_grad_async=False, kv_img=None, vertices_are_boxes=False):
    """Same contract as ``vdetr_amd.attention.fused_attention`` (q [B,nQ,H*64]; k,v [B,nK,64] or [B,nK,H*64]).

    shared_kv: vdetr_transformer.py:733-753 (cross attention) / :638-648 (ShareSelfAttention);
    per-head : the scaled-dot-product core of nn.MultiheadAttention.
    ``keep_mask`` [B,H,nQ,nK] (0/1) applies dropout with a GIVEN mask (the HIP kernel's, dumped through the test
    hook) so that dropout runs can be compared element-wise.
    """
    B, nQ, C = q.shape
    nK = k.shape[1]
    H = num_heads
    dh = C // H
    qh = (q.view(B, nQ, H, dh).permute(0, 2, 1, 3)) * scale
    if shared_kv:
        kh = k.view(B, 1, nK, dh)
        vh = v.view(B, 1, nK, dh)
    else:
        kh = k.view(B, nK, H, dh).permute(0, 2, 1, 3)
        vh = v.view(B, nK, H, dh).permute(0, 2, 1, 3)
    attn = qh @ kh.transpose(-2, -1)  # B,H,nQ,nK
    if table is not None:
        bias_fn = rpe_bias_reference if rpe_impl == "explicit" else rpe_bias_grid_sample
        attn = attn + bias_fn(table, vertices, xyz, rpe.log_scale, rpe.max_value, cos_sin)
    if attn_mask is not None:
        m = attn_mask if attn_mask.dim() == 3 else attn_mask.unsqueeze(0)
        m = m.unsqueeze(1).expand(B, H, nQ, nK)
        if m.dtype == torch.bool or m.dtype == torch.uint8:
            attn = attn.masked_fill(m.bool(), -100.0)  # :746-747
        else:
            attn = attn + m
    probs = torch.softmax(attn, dim=-1)
    if keep_mask is not None:
        probs = probs * keep_mask.to(probs.dtype) / (1.0 - dropout_p)
    elif dropout_p > 0.0:
        probs = torch.nn.functional.dropout(probs, dropout_p, training=True)
    out = (probs @ vh).transpose(1, 2).reshape(B, nQ, C)
    return (out, probs) if return_probs else out
