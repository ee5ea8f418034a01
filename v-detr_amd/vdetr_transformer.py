"""V-DETR decoder on the MI355X kernels — module / argument / state-dict compatible with the reference's
``models/vdetr_transformer.py``.

Only the attention cores differ from a plain PyTorch module: ``GlobalShareCrossAttention`` (3DV-RPE cross
attention, reference :656-758), ``ShareSelfAttention`` (:609-653) and ``MultiheadSelfAttention`` (the
``nn.MultiheadAttention(256, 4)`` of :468) call ``attention.fused_attention`` (HIP).  Projections, norms, FFNs and
the box heads are library GEMMs / element-wise ops and stay in PyTorch with the reference's parameter names:

  decoder.first_layer.{linear1,linear2,norm}            decoder.layers.N.self_attn.{in_proj_weight,in_proj_bias,out_proj}
  decoder.layers.N.multihead_attn.{relative_coords_table,cpb_mlps.I.{0,2},q,k,v,proj}
  decoder.layers.N.{norm1,norm2,norm3,linear1,linear2}  decoder.norm   decoder.query_embed
  decoder.query_pos_projection.N.position_embedding_head.{0,1,3}
  decoder.mlp_heads.S.{sem_cls_head,center_head,size_head,angle_cls_head,angle_residual_head}.layers.{0,1,4,5,8}
  decoder.pointcls_heads.layers.*
"""
import copy
import itertools
import math
import os
from functools import partial
from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor, nn

from . import add_ln as ALN
from . import bn_act as BNA
from . import attention as A
from . import box_decode
from . import heads as HD
from . import rowblock as RB
from .helpers import (ACTIVATION_DICT, NORM_DICT, WEIGHT_INIT_DICT, GenericMLP, PointwiseConv1d,
                      PositionEmbeddingLearned, buffers_alias, cat_params, get_clones, linear, linear_pair, slot_stack_params,
                      stack_params)
from .pc_util import morton_argsort, scale_points, shift_scale_points

_salt_counter = itertools.count(1)


class BoxProcessor(object):
    """MLP head outputs -> box parameters (reference :20-90)."""

    def __init__(self, dataset_config, cls_loss="celoss"):
        self.dataset_config = dataset_config
        self.cls_loss = cls_loss

    def compute_predicted_center(self, center_offset, query_xyz, point_cloud_dims):
        center_unnormalized = query_xyz + center_offset
        return shift_scale_points(center_unnormalized, src_range=point_cloud_dims), center_unnormalized

    def compute_predicted_class_size(self, size_normalized_offset, logits):
        class_idx = logits.sigmoid().max(dim=-1)[1]
        size_per_class = torch.tensor(self.dataset_config.mean_size_arr, device=logits.device).float()
        class_size = size_per_class[class_idx]
        return class_size + size_normalized_offset * class_size, size_normalized_offset * class_size

    def compute_predicted_size(self, size_normalized, point_cloud_dims):
        scene_scale = torch.clamp(point_cloud_dims[1] - point_cloud_dims[0], min=1e-1)
        return scale_points(size_normalized, mult_factor=scene_scale)

    def compute_predicted_angle(self, angle_logits, angle_residual, zero_angle=False):
        """:48-71.  Datasets without rotation (one angle bin) still route the head outputs into the result (x0) so
        that every parameter receives a gradient, as the reference does for DDP."""
        nbin = angle_logits.shape[-1]
        if nbin == 1 or zero_angle:
            if nbin == 1:
                angle = (angle_logits * 0 + angle_residual * 0).squeeze(-1).clamp(min=0)
            else:
                angle = (angle_logits.sum(-1) * 0 + angle_residual.sum(-1) * 0).squeeze(-1).clamp(min=0)
            return angle, angle
        angle_per_cls = 2 * np.pi / self.dataset_config.num_angle_bin
        angle_prob, pred_angle_class = F.softmax(angle_logits, dim=-1).max(dim=-1)
        pred_angle_class = pred_angle_class.detach()
        angle = angle_per_cls * pred_angle_class + angle_residual.gather(2, pred_angle_class.unsqueeze(-1)).squeeze(-1)
        angle = torch.where(angle > np.pi, angle - 2 * np.pi, angle)
        return angle, angle_prob

    def compute_objectness_and_cls_prob(self, cls_logits):
        if self.cls_loss.split("_")[0] == "focalloss":
            return cls_logits, cls_logits.sigmoid().max(dim=-1)[0]
        assert cls_logits.shape[-1] == self.dataset_config.num_semcls + 1
        cls_prob = F.softmax(cls_logits, dim=-1)
        return cls_prob[..., :-1], 1 - cls_prob[..., -1]

    def box_parametrization_to_corners(self, box_center_unnorm, box_size_unnorm, box_angle):
        return self.dataset_config.box_parametrization_to_corners(box_center_unnorm, box_size_unnorm, box_angle)


def inverse_sigmoid(x, eps=1e-5):
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def convert_corners_camera2lidar(corners_camera):
    """camera (x, y, z) -> lidar (x, z, -y)   (reference :98-102, without mutating the argument)"""
    return torch.stack((corners_camera[..., 0], corners_camera[..., 2], -corners_camera[..., 1]), dim=-1)


def roty_batch_tensor(t):
    c, s = torch.cos(t), torch.sin(t)
    zero, one = torch.zeros_like(c), torch.ones_like(c)
    return torch.stack((torch.stack((c, zero, s), -1), torch.stack((zero, one, zero), -1),
                        torch.stack((-s, zero, c), -1)), -2)


def rotz_batch_tensor(t):
    c, s = torch.cos(t), torch.sin(t)
    zero, one = torch.zeros_like(c), torch.ones_like(c)
    return torch.stack((torch.stack((c, -s, zero), -1), torch.stack((s, c, zero), -1),
                        torch.stack((zero, zero, one), -1)), -2)


# =====================================================================================================
# attention modules
# =====================================================================================================
def set_attention_dtype(module, dtype):
    """storage type of q / k / v in every 3DV-RPE cross attention under `module` (torch.float32 | torch.bfloat16)"""
    assert dtype in (torch.float32, torch.bfloat16)
    n = 0
    for m in module.modules():
        if isinstance(m, GlobalShareCrossAttention):
            m.core_dtype = dtype
            n += 1
    return n


class GlobalShareCrossAttention(nn.Module):
    """3D-vertex-RPE cross attention (reference :656-758).

    forward(query [nQ,B,C], key [nK,B,C], reference_point [B,nQ,8,3], reference_angle [B,nQ], xyz [B,nK,3],
            attn_mask=None, key_padding_mask=None) -> (x [nQ,B,C], attn or None)

    K and V are 64-wide and shared by the heads.  The bias rpe[b,h,q,k] = sum_i trilinear(T_i, g(P_i[q] - X[k])) is
    evaluated inside the fused HIP kernel from the eight tables T_i = cpb_mlps[i](relative_coords_table); the
    [B,H,nQ,nK] bias / probability tensors of the reference never exist.  ``attn`` (post-dropout probabilities,
    :751-752) is only produced when ``self.return_attn`` is set.  ``key_padding_mask`` is accepted and ignored,
    exactly as in the reference.
    """

    def __init__(self, dim, num_heads, qkv_bias=True, attn_drop=0.0, proj_drop=0.0, args=None):
        super().__init__()
        assert dim % num_heads == 0, "dim should be divisible by num_heads"
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = head_dim ** -0.5
        self.log_scale = args.log_scale
        self.rpe_quant = args.rpe_quant
        self.angle_type = args.angle_type
        self.interp_method, max_value, num_points = self.rpe_quant.split("_")
        max_value, num_points = float(max_value), int(num_points)
        if self.interp_method != "bilinear":
            raise NotImplementedError(f"rpe_quant interpolation '{self.interp_method}' (only 'bilinear' is built)")
        lin = torch.linspace(-max_value, max_value, num_points, dtype=torch.float32)
        table = torch.stack(torch.meshgrid(lin, lin, lin, indexing="ij"), dim=-1).unsqueeze(0)
        self.register_buffer("relative_coords_table", table)  # [1,T,T,T,3]
        self.max_value = max_value
        self.cpb_mlps = get_clones(self.build_cpb_mlp(3, args.rpe_dim, num_heads), 8)
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.k = nn.Linear(dim, dim // num_heads, bias=qkv_bias)
        self.v = nn.Linear(dim, dim // num_heads, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)  # p is read by the fused kernel
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.rpe_cfg = A.RPEConfig(num_points, self.log_scale, max_value)
        self.return_attn = False
        # set by a caller whose reference points are box corners out of its own box decode (TransformerDecoder): the table gradient
        # then launches the box kernel alone (attention.fused_attention: vertices_are_boxes)
        self.vertices_are_boxes = False
        self.defer_proj_drop = False
        # storage type of the projected q / k / v handed to the attention core: torch.bfloat16 = BASELINE config 4 (the bf16
        # matrix instructions for QK^T / PV; scores, RPE, softmax, accumulators, output fp32); set_attention_dtype() below
        self.core_dtype = torch.float32
        self._salt = next(_salt_counter)

    def __deepcopy__(self, memo):
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            setattr(new, k, copy.deepcopy(v, memo))
        new._salt = next(_salt_counter)  # every clone draws its own dropout stream
        return new

    def build_cpb_mlp(self, in_dim, hidden_dim, out_dim):
        return nn.Sequential(nn.Linear(in_dim, hidden_dim, bias=True), nn.ReLU(inplace=False),
                             nn.Linear(hidden_dim, out_dim, bias=False))

    def rpe_tables(self):
        """[8,T,T,T,H]: the eight cpb MLPs evaluated on the coordinate grid (reference :725), as two batched GEMMs."""
        w1 = torch.stack([m[0].weight for m in self.cpb_mlps])  # [8,hid,3]
        b1 = torch.stack([m[0].bias for m in self.cpb_mlps])    # [8,hid]
        w2 = torch.stack([m[2].weight for m in self.cpb_mlps])  # [8,H,hid]
        T = self.relative_coords_table.shape[1]
        hid = self._cpb_hidden(self, w1, b1, 8)
        return torch.bmm(hid, w2.transpose(1, 2)).view(8, T, T, T, self.num_heads)

    @staticmethod
    def _cpb_hidden(mod, w1, b1, n):
        """relu(coords W1^T + b1) of n cpb MLPs ([n,hid,3], [n,hid]) on the coordinate grid.  The bias rides in the GEMM as a
        fourth contraction row against a column of ones: `baddbmm` first copies the broadcast bias into the [n, T^3, hid]
        output (22 us for 32 MB at n = 64) and its backward reduces that tensor again for the bias gradient (21 us); here the
        weight gradient's GEMM yields the bias gradient as its fourth row."""
        tab = mod.relative_coords_table
        c1 = mod.__dict__.get("_coords1")
        if c1 is None or c1.device != tab.device or c1.dtype != tab.dtype:
            flat = tab.reshape(-1, 3)
            c1 = mod.__dict__["_coords1"] = torch.cat((flat, torch.ones_like(flat[:, :1])), dim=1)  # [T^3, 4]
        w1b = torch.cat((w1, b1.unsqueeze(-1)), dim=2)                                           # [n, hid, 4]
        return torch.relu(torch.bmm(c1.unsqueeze(0).expand(n, -1, -1), w1b.transpose(1, 2)))

    @staticmethod
    def _cpb_fused(mod, w1, b1, w2, n):
        """(relu(coords W1^T + b1) [n, T^3, hid], tables [n, T^3, H]) of n cpb MLPs by vdetr_cpb_tables_f32, or None where the launch
        does not apply (not on the GPU, other widths); leaves mod._coords1 (the tables' backward reads it) as _cpb_hidden does"""
        if not (_CPB_FUSED and w1.is_cuda and w1.dtype == torch.float32 and w2.shape[1] == 4 and w1.shape[1] % 16 == 0 and w1.shape[1] <= 256
                and w1.is_contiguous() and b1.is_contiguous() and w2.is_contiguous()):
            return None
        tab = mod.relative_coords_table
        c1 = mod.__dict__.get("_coords1")
        if c1 is None or c1.device != tab.device or c1.dtype != tab.dtype:
            flat = tab.reshape(-1, 3)
            c1 = mod.__dict__["_coords1"] = torch.cat((flat, torch.ones_like(flat[:, :1])), dim=1)  # [T^3, 4]
        coords = mod.__dict__.get("_coords3")
        if coords is None or coords.device != tab.device:
            coords = mod.__dict__["_coords3"] = tab.reshape(-1, 3).contiguous()
        P, hidden = coords.shape[0], w1.shape[1]
        hid = torch.empty((n, P, hidden), dtype=torch.float32, device=w1.device)
        tables = torch.empty((n, P, 4), dtype=torch.float32, device=w1.device)
        from . import _lib as L
        L.check(L.lib().vdetr_cpb_tables_f32(L.ptr(coords), L.ptr(w1), L.ptr(b1), L.ptr(w2), n, P, hidden, 4, L.ptr(hid), L.ptr(tables),
                                             L.stream_ptr()), "cpb_tables")
        return hid, tables

    @staticmethod
    def precompute(mods, key):
        """K, V and RPE tables of SEVERAL cross-attention modules that see the same ``key`` [nK,B,C] (the decoder layers
        all attend to the same encoder features): one [C -> n*128] projection GEMM instead of 2n small ones, and the
        8n cpb MLPs as two batched GEMMs.  Returns a list of (k, v, tables) for ``forward(..., cache=)``; k / v are
        64-wide column blocks of the joint projection (row-strided views, read in place by the kernel)."""
        n = len(mods)
        key_b = key.permute(1, 0, 2)
        w = cat_params([t for m in mods for t in (m.k.weight, m.v.weight)])
        b = cat_params([t for m in mods for t in (m.k.bias, m.v.bias)]) if mods[0].k.bias is not None else None
        kv = linear(key_b, w, b)                                                   # [B,nK,n*128]
        imgs = None
        if kv.is_cuda and mods[0].rpe_cfg.table_size == 10:
            # the forward kernels' K / V operand images of all layers: one launch (core_dtype bf16: one rounded part per operand)
            imgs = A.pack_kv_images(kv.detach(), n, parts=1 if mods[0].core_dtype == torch.bfloat16 else 3)
        if imgs is None and mods[0].core_dtype != torch.float32 and kv.is_cuda:
            kv = kv.to(mods[0].core_dtype)  # (no image path: bf16 tensors, one cast for the K / V of every layer)
        parts = kv.view(kv.shape[0], kv.shape[1], 2 * n, -1).unbind(2)
        mlps = [mm for m in mods for mm in m.cpb_mlps]
        w1 = stack_params([mm[0].weight for mm in mlps])
        b1 = stack_params([mm[0].bias for mm in mlps])
        w2 = stack_params([mm[2].weight for mm in mlps])
        T, H = mods[0].relative_coords_table.shape[1], mods[0].num_heads
        # the table gradients are computed on a side stream (attention.py: ASYNC_TABLE_GRAD); autograd reaches the joins —
        # they were created before any decoder layer — only after every layer's backward, and waits there; with parked weight
        # gradients the tables' own backward (three GEMMs, written out in DeferredTableGrads.begin_flush) follows the last
        # table kernel ON that stream, next to the flush
        if A.table_grads_parkable(w1) and os.environ.get("VDETR_BWD_ASYNC_MLP", "1") != "0":
            with torch.no_grad():
                fused = GlobalShareCrossAttention._cpb_fused(mods[0], w1, b1, w2, 8 * n)
                if fused is not None:  # hidden activations + tables of all 8 n MLPs in one launch (csrc/cpb_tables.hip)
                    hid, tables = fused
                    tables = tables.view(n, 8, T, T, T, H)
                else:
                    hid = GlobalShareCrossAttention._cpb_hidden(mods[0], w1, b1, 8 * n)
                    tables = torch.bmm(hid, w2.transpose(1, 2)).view(n, 8, T, T, T, H)
            tables = A.park_table_grads(tables, mlp=(mods[0].__dict__["_coords1"], w1, b1, w2, hid))
        else:
            hid = GlobalShareCrossAttention._cpb_hidden(mods[0], w1, b1, 8 * n)
            tables = A.park_table_grads(torch.bmm(hid, w2.transpose(1, 2)).view(n, 8, T, T, T, H))
        return [(parts[2 * i], parts[2 * i + 1], tables[i], imgs[i] if imgs is not None else None) for i in range(n)]

    def core(self, q, key, reference_point, reference_angle, xyz, attn_mask=None, cache=None, defer_combine=False):
        """The attention between the query projection and the output projection (:733-753): q [B,nQ,C] projected queries ->
        (x [B,nQ,C], attn or None).  `key` [nK,B,C] is projected here unless `cache` = (k, v, tables) carries the layer's share
        of TransformerDecoder's joint projection."""
        if cache is None:
            key_b = key.permute(1, 0, 2)
            k = linear(key_b, self.k.weight, self.k.bias)     # [B,nK,C/H]
            v = linear(key_b, self.v.weight, self.v.bias)
        else:
            k, v = cache[0], cache[1]
        cos_sin = None
        if self.angle_type == "object_coords" and reference_angle is not None:
            ang = reference_angle.detach()
            cos_sin = torch.stack((torch.cos(ang), torch.sin(ang)), dim=-1)
        p = self.attn_drop.p if self.training else 0.0
        rng = A.current_rng(q.device) if p > 0 else None
        if p > 0 and rng is None:
            rng = A.begin_step(q.device)
        tables = self.rpe_tables() if cache is None else cache[2]
        q32, k32 = q, k
        # bf16 arithmetic (BASELINE config 4): where the persistent forward takes the call, f32 tensors whose values are rounded to
        # bf16 inside the kernels (no cast launches, no bf16 copies: measured 0.37 ms of the C4 step); bf16 tensors otherwise
        rounded = (self.core_dtype == torch.bfloat16 and q.is_cuda and A.FWD_KERNEL == 0 and attn_mask is None
                   and self.rpe_cfg.table_size == 10 and k.dtype == torch.float32)
        if self.core_dtype != torch.float32 and q.is_cuda and not rounded:
            q, k, v = (t if t.dtype == self.core_dtype else t.to(self.core_dtype) for t in (q, k, v))
        x = A.fused_attention(q, k, v, num_heads=self.num_heads, scale=self.scale, shared_kv=True, table=tables,
                              rpe=self.rpe_cfg, vertices=reference_point, xyz=xyz, cos_sin=cos_sin,
                              attn_mask=attn_mask, dropout_p=p, rng_state=rng, salt=self._salt,
                              table_grad_async=cache is not None, vertices_are_boxes=self.vertices_are_boxes, **({"operand_bf16": True} if rounded else {}),
                              **({"kv_img": cache[3]} if (cache is not None and len(cache) > 3 and cache[3] is not None) else {}),
                              **({"defer_combine": True} if (defer_combine and not self.return_attn) else {}))
        attn = None
        if self.return_attn:
            attn = A.attention_probabilities(q32.float(), k32.float(), num_heads=self.num_heads, scale=self.scale, shared_kv=True,
                                             table=tables, rpe=self.rpe_cfg, vertices=reference_point, xyz=xyz,
                                             cos_sin=cos_sin, attn_mask=attn_mask, dropout_p=p, rng_state=rng,
                                             salt=self._salt)
        return x, attn

    def forward(self, query, key, reference_point, reference_angle, xyz, attn_mask=None, key_padding_mask=None,
                cache=None):
        query_b = query.permute(1, 0, 2)
        q = linear(query_b, self.q.weight, self.q.bias)   # [B,nQ,C]
        x, attn = self.core(q, key, reference_point, reference_angle, xyz, attn_mask, cache)
        x = linear(x, self.proj.weight, self.proj.bias)
        if not self.defer_proj_drop:  # (the caller applies it together with its own residual dropout: one mask)
            x = self.proj_drop(x)
        return x.permute(1, 0, 2), attn


class ShareSelfAttention(nn.Module):
    """Self attention with head-shared 64-wide K/V (reference :609-653), enabled by --share_selfattn."""

    def __init__(self, dim, num_heads, qkv_bias=True, dropout=0.0, args=None):
        super().__init__()
        assert dim % num_heads == 0, "dim should be divisible by num_heads"
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.k = nn.Linear(dim, dim // num_heads, bias=qkv_bias)
        self.v = nn.Linear(dim, dim // num_heads, bias=qkv_bias)
        self.attn_drop = nn.Dropout(dropout)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(dropout)
        self._salt = next(_salt_counter)

    __deepcopy__ = GlobalShareCrossAttention.__deepcopy__

    def forward(self, query, key, value=None, attn_mask=None, key_padding_mask=None):
        assert attn_mask is None and key_padding_mask is None
        key_b, query_b = key.permute(1, 0, 2), query.permute(1, 0, 2)
        k = linear(key_b, self.k.weight, self.k.bias)
        # The reference projects `value` WITHOUT the (1,0,2) permute and then reshapes the sequence-first
        # [N,B,64] result as (B,N,64) (:639).  Identical for B == 1, a batch/sequence mix-up for B > 1; kept as
        # is so that outputs match the reference on the same inputs.
        v = linear(value, self.v.weight, self.v.bias).reshape(key_b.shape[0], key_b.shape[1], -1)
        q = linear(query_b, self.q.weight, self.q.bias)
        p = self.attn_drop.p if self.training else 0.0
        x = A.fused_attention(q, k, v, num_heads=self.num_heads, scale=self.scale, shared_kv=True, dropout_p=p,
                              salt=self._salt)
        return self.proj_drop(linear(x, self.proj.weight, self.proj.bias)).permute(1, 0, 2), None


class _OutProj(nn.Linear):
    """Plain Linear under the name ``out_proj`` (nn.MultiheadAttention's NonDynamicallyQuantizableLinear)."""


class MultiheadSelfAttention(nn.Module):
    """Drop-in for the ``nn.MultiheadAttention(d_model, nhead, dropout)`` the reference builds at :468: same
    parameters (in_proj_weight [3E,E], in_proj_bias [3E], out_proj.{weight,bias}), same call
    ``(query, key, value=..., attn_mask=..., key_padding_mask=...) -> (out, None)`` on sequence-first tensors."""

    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        assert embed_dim % num_heads == 0
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.head_dim = embed_dim // num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.empty(3 * embed_dim))
        self.out_proj = _OutProj(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.in_proj_bias, 0.0)
        nn.init.constant_(self.out_proj.bias, 0.0)
        self._salt = next(_salt_counter)

    __deepcopy__ = GlobalShareCrossAttention.__deepcopy__

    def forward(self, query, key, value=None, attn_mask=None, key_padding_mask=None, need_weights=False):
        E = self.embed_dim
        value = key if value is None else value
        # three projections from the packed weight.  unbind (backward = one stack) instead of slicing the parameter
        # or a joint q/k output (each slice's backward is a zero-filled full-size buffer + copy)
        wq, wk, wv = self.in_proj_weight.view(3, E, E).unbind(0)
        bq, bk, bv = self.in_proj_bias.view(3, E).unbind(0)
        if query is key:  # the decoder's call (q = k = tgt + pos): both projections in one batched GEMM
            q, k = linear_pair(query, wq, bq, wk, bk)
        else:
            q, k = linear(query, wq, bq), linear(key, wk, bk)
        v = linear(value, wv, bv)
        L_, B = query.shape[0], query.shape[1]
        S = key.shape[0]
        mask = None
        if attn_mask is not None or key_padding_mask is not None:
            mask = torch.zeros((B, L_, S), dtype=torch.float32, device=query.device)
            if attn_mask is not None:
                am = attn_mask
                am = torch.zeros_like(am, dtype=torch.float32).masked_fill_(am, -1e30) if am.dtype == torch.bool else am.float()
                mask = mask + (am if am.dim() == 3 else am.unsqueeze(0))
            if key_padding_mask is not None:
                mask = mask.masked_fill(key_padding_mask.view(B, 1, S).bool(), -1e30)
        p = self.dropout if self.training else 0.0
        x = A.fused_attention(q.transpose(0, 1), k.transpose(0, 1), v.transpose(0, 1), num_heads=self.num_heads,
                              scale=self.head_dim ** -0.5, shared_kv=False, attn_mask=mask, dropout_p=p,
                              salt=self._salt)
        return linear(x, self.out_proj.weight, self.out_proj.bias).transpose(0, 1), None


# =====================================================================================================
# decoder layers
# =====================================================================================================
def _proposal_order(objectness, n):
    """indices [B, n] of the n highest objectness values, ties to the lower token index (see the proposals in
    GlobalTransformer.forward: a stable sort is one of the orders torch.topk may return, and the same one on every device)"""
    if objectness.is_cuda and objectness.dtype == torch.float32 and objectness.dim() == 2:
        from . import _lib as L
        if objectness.shape[1] <= L.lib().vdetr_morton_sort_max():  # one launch (csrc/morton.hip: topk_order_kernel), the same order
            vals = objectness.contiguous()
            order = torch.empty((vals.shape[0], n), dtype=torch.int64, device=vals.device)
            L.check(L.lib().vdetr_topk_order_f32(L.ptr(vals), vals.shape[0], vals.shape[1], n, L.ptr(order), L.stream_ptr()), "topk_order")
            return order
    return torch.sort(objectness, dim=1, descending=True, stable=True)[1][:, :n]


# the glue between a layer's attention cores as three launches (rowblock.hip); VDETR_ROWBLOCK=0: one launch per op (A/B, parity)
_ROWBLOCK = os.environ.get("VDETR_ROWBLOCK", "1") != "0"


def _act_drop(mod, h):
    """``mod.dropout(mod.activation(h))`` of an FFN block (:566, :604): one launch for relu + dropout on the GPU."""
    if type(mod.activation) is nn.ReLU and h.is_cuda and h.dtype == torch.float32 and h.numel() % 4 == 0:
        if getattr(mod, "_act_salt", None) is None:
            mod._act_salt = BNA.new_salt()
        return BNA.relu_dropout(h, mod.dropout, salt=mod._act_salt)
    return mod.dropout(mod.activation(h))


class GlobalDecoderLayer(nn.Module):
    """self-attn -> 3DV-RPE cross-attn -> FFN with residuals (reference :455-582; pre-norm by default)."""

    def __init__(self, d_model, nhead=4, dim_feedforward=256, dropout=0.1, dropout_attn=None, activation="relu",
                 normalize_before=True, norm_fn_name="ln", pos_for_key=False, args=None):
        super().__init__()
        if dropout_attn is None:
            dropout_attn = dropout
        self.pos_for_key = pos_for_key
        self.cross_cache = None  # (k, v, tables) handed over by TransformerDecoder for the current forward
        # fused residual + dropout + LayerNorm plumbing, set by TransformerDecoder around a call:
        self.pre_normed = None   # norm1(tgt), already computed by the previous block's launch
        self.post_norms = None   # LayerNorm modules to apply to the layer output in the same launch as dropout3 + add
        self.post_normed = None  # their results
        self._aln_salts = None   # dropout streams of the three residual blocks (created on first use: clones differ)
        if args.share_selfattn:
            self.self_attn = ShareSelfAttention(d_model, nhead, dropout=dropout)
        else:
            self.self_attn = MultiheadSelfAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = GlobalShareCrossAttention(d_model, nhead, attn_drop=dropout, proj_drop=dropout, args=args)
        self.norm1 = NORM_DICT[norm_fn_name](d_model)
        self.norm2 = NORM_DICT[norm_fn_name](d_model)
        self.norm3 = NORM_DICT[norm_fn_name](d_model)
        self.dropout1 = nn.Dropout(dropout, inplace=False)
        self.dropout2 = nn.Dropout(dropout, inplace=False)
        self.dropout3 = nn.Dropout(dropout, inplace=False)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout, inplace=False)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.activation = ACTIVATION_DICT[activation]()
        self.normalize_before = normalize_before

    @staticmethod
    def with_pos_embed(tensor, pos: Optional[Tensor]):
        return tensor if pos is None else tensor + pos

    def _cross(self, tgt_in, memory, reference_point, reference_angle, enc_xyz, memory_mask,
               memory_key_padding_mask, pos, query_pos):
        key = self.with_pos_embed(memory, pos) if self.pos_for_key else memory
        extra = {"cache": self.cross_cache} if self.cross_cache is not None else {}
        return self.multihead_attn(query=self.with_pos_embed(tgt_in, query_pos), key=key,
                                   reference_point=reference_point, reference_angle=reference_angle, xyz=enc_xyz,
                                   attn_mask=memory_mask, key_padding_mask=memory_key_padding_mask, **extra)

    def forward_pre(self, tgt, memory, reference_point, reference_angle, enc_xyz, point_cloud_dims, tgt_mask=None,
                    memory_mask=None, tgt_key_padding_mask=None, memory_key_padding_mask=None, pos=None,
                    query_pos=None, return_attn_weights=False):
        if not ALN.supported(self.norm1, self.norm2, self.norm3):  # other norm types: the plain composition
            HD.materialize_pos(query_pos)
            tgt2 = self.norm1(tgt)
            q = k = self.with_pos_embed(tgt2, query_pos)
            tgt2 = self.self_attn(q, k, value=tgt2, attn_mask=tgt_mask, key_padding_mask=tgt_key_padding_mask)[0]
            tgt = tgt + self.dropout1(tgt2)
            tgt2, attn = self._cross(self.norm2(tgt), memory, reference_point, reference_angle, enc_xyz, memory_mask,
                                     memory_key_padding_mask, pos, query_pos)
            tgt = tgt + self.dropout2(tgt2)
            tgt2 = linear(self.dropout(self.activation(linear(self.norm3(tgt), self.linear1.weight, self.linear1.bias))), self.linear2.weight, self.linear2.bias)
            tgt = tgt + self.dropout3(tgt2)
            return tgt, (attn if return_attn_weights else None)
        # `tgt = tgt + dropoutN(branch); normed = norm(tgt)` (:541-560) is one launch per residual block
        if self._aln_salts is None:
            self._aln_salts = [ALN.new_salt() for _ in range(3)]
        tgt2 = self.pre_normed if self.pre_normed is not None else ALN.layer_norm(tgt, self.norm1)
        use_rb = (_ROWBLOCK and self.post_norms and not return_attn_weights and not self.pos_for_key and
                  RB.usable(self, tgt, query_pos, (tgt_mask, memory_mask, tgt_key_padding_mask, memory_key_padding_mask)))
        if not use_rb:
            HD.materialize_pos(query_pos)  # (a position MLP whose launch the decoder left to rowblock.qkv: here instead)
        if use_rb:
            # everything between the attention cores as three launches (rowblock.hip): the same values, the same dropout
            # streams (salts) as the composition below
            B = tgt.shape[1]
            sa, ca = self.self_attn, self.multihead_attn
            img = RB.images(self)   # the projections' W^T images: the decoder rewrites all layers' in one launch per forward
            q, k, v, pos2 = RB.qkv(tgt2, query_pos, sa, B, img)  # (pos2 = query_pos, routed through qkv for its gradient)
            p = sa.dropout if sa.training else 0.0
            core = A.fused_attention(q, k, v, num_heads=sa.num_heads, scale=sa.head_dim ** -0.5, shared_kv=False, dropout_p=p,
                                     salt=sa._salt)
            tgt, qc = RB.proj_q(core, tgt, pos2, sa.out_proj, ca.q, self.dropout1, self.norm2, self._aln_salts[0], B, img)
            # (defer_combine: the merge of the forward's key-split partials is done by rb_ffn, the next launch, on its way in)
            core, _ = ca.core(qc, memory, reference_point, reference_angle, enc_xyz, None, self.cross_cache, defer_combine=True)
            if getattr(self, "_act_salt", None) is None:
                self._act_salt = BNA.new_salt()
            res = RB.ffn(core, tgt, self, self.post_norms, self._aln_salts, self._act_salt, B, img)
            self.post_normed = tuple(r for r in res[1:] if r is not None)
            return res[0], None
        q = k = self.with_pos_embed(tgt2, query_pos)
        branch = self.self_attn(q, k, value=tgt2, attn_mask=tgt_mask, key_padding_mask=tgt_key_padding_mask)[0]
        tgt, tgt2 = ALN.add_dropout_layer_norm(tgt, branch, self.dropout1, self.norm2, salt=self._aln_salts[0])
        fold = isinstance(self.multihead_attn, GlobalShareCrossAttention)
        if fold:  # proj_drop and dropout2 are two masks on the same tensor: applied as one inside the fused launch
            self.multihead_attn.defer_proj_drop = True
        branch, attn = self._cross(tgt2, memory, reference_point, reference_angle, enc_xyz, memory_mask,
                                   memory_key_padding_mask, pos, query_pos)
        if fold:
            self.multihead_attn.defer_proj_drop = False
        tgt, tgt2 = ALN.add_dropout_layer_norm(tgt, branch, self.dropout2, self.norm3, salt=self._aln_salts[1],
                                               also_drop=self.multihead_attn.proj_drop if fold else None)
        branch = linear(_act_drop(self, linear(tgt2, self.linear1.weight, self.linear1.bias)), self.linear2.weight,
                        self.linear2.bias)
        if self.post_norms:  # the decoder's output norm (+ the next layer's norm1) ride in the same launch
            res = ALN.add_dropout_layer_norm(tgt, branch, self.dropout3, *self.post_norms, salt=self._aln_salts[2])
            tgt, self.post_normed = res[0], res[1:]
        else:
            tgt = tgt + self.dropout3(branch)
        return tgt, (attn if return_attn_weights else None)

    def forward_post(self, tgt, memory, reference_point, reference_angle, enc_xyz, point_cloud_dims, tgt_mask=None,
                     memory_mask=None, tgt_key_padding_mask=None, memory_key_padding_mask=None, pos=None,
                     query_pos=None, return_attn_weights=False):
        HD.materialize_pos(query_pos)  # (never through rowblock.py)
        q = k = self.with_pos_embed(tgt, query_pos)
        tgt2 = self.self_attn(q, k, value=tgt, attn_mask=tgt_mask, key_padding_mask=tgt_key_padding_mask)[0]
        tgt = self.norm1(tgt + self.dropout1(tgt2))
        tgt2, attn = self._cross(tgt, memory, reference_point, reference_angle, enc_xyz, memory_mask,
                                 memory_key_padding_mask, pos, query_pos)
        tgt = self.norm2(tgt + self.dropout2(tgt2))
        tgt2 = linear(self.dropout(self.activation(linear(tgt, self.linear1.weight, self.linear1.bias))), self.linear2.weight, self.linear2.bias)
        tgt = self.norm3(tgt + self.dropout3(tgt2))
        return tgt, (attn if return_attn_weights else None)

    def forward(self, tgt, memory, reference_point, reference_angle, enc_xyz, point_cloud_dims, tgt_mask=None,
                memory_mask=None, tgt_key_padding_mask=None, memory_key_padding_mask=None, pos=None, query_pos=None,
                return_attn_weights=False):
        self.multihead_attn.return_attn = bool(return_attn_weights)
        fn = self.forward_pre if self.normalize_before else self.forward_post
        return fn(tgt, memory, reference_point, reference_angle, enc_xyz, point_cloud_dims, tgt_mask, memory_mask,
                  tgt_key_padding_mask, memory_key_padding_mask, pos, query_pos, return_attn_weights)


class FFNLayer(nn.Module):
    """The light first 'decoder layer' applied to all encoder tokens (reference :585-606)."""

    def __init__(self, d_model, dim_feedforward=256, dropout=0.1, norm_fn_name="ln", activation="relu",
                 normalize_before=True):
        super().__init__()
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout, inplace=False)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm = NORM_DICT[norm_fn_name](d_model)
        self.activation = ACTIVATION_DICT[activation]()
        self.normalize_before = normalize_before
        self.post_norm = None    # LayerNorm the caller applies to the output: fused into the residual launch
        self.post_normed = None
        self._aln_salt = None

    def forward_pre(self, memory):
        if not ALN.supported(self.norm):
            memory = self.norm(memory)
            return memory + self.dropout(linear(self.dropout(self.activation(linear(memory, self.linear1.weight, self.linear1.bias))), self.linear2.weight, self.linear2.bias))
        if RB.ffn0_usable(self, memory):  # the whole layer + the caller's output norm as ONE launch (csrc/rowblock.hip: rb_ffn_kernel<-2>)
            if getattr(self, "_act_salt", None) is None:
                self._act_salt = BNA.new_salt()
            if self._aln_salt is None:
                self._aln_salt = ALN.new_salt()
            out, self.post_normed = RB.ffn0(self, memory, self._act_salt, self._aln_salt)
            return out
        memory = ALN.layer_norm(memory, self.norm)
        branch = linear(_act_drop(self, linear(memory, self.linear1.weight, self.linear1.bias)), self.linear2.weight,
                        self.linear2.bias)
        if self.post_norm is not None:
            if self._aln_salt is None:
                self._aln_salt = ALN.new_salt()
            memory, self.post_normed = ALN.add_dropout_layer_norm(memory, branch, self.dropout, self.post_norm,
                                                                  salt=self._aln_salt)
            return memory
        return memory + self.dropout(branch)

    def forward(self, memory):
        return self.forward_pre(memory)


# =====================================================================================================
# decoder
# =====================================================================================================
_DEFER_HEADS = os.environ.get("VDETR_DEFER_HEADS", "1") != "0"  # A/B switch (read once)
_CPB_FUSED = os.environ.get("VDETR_CPB_FUSED", "1") != "0"       # the RPE tables' MLPs as one launch (csrc/cpb_tables.hip)
_STAGE0_WG_SIDE = os.environ.get("VDETR_STAGE0_WG", "inline") == "side"
_DEFER_STAGE0 = os.environ.get("VDETR_DEFER_STAGE0", "1") != "0"  # the first stage's heads recorded too (round 6; A/B switch)
# the heads' weight gradients on the side branch (VDETR_HEADS_SIDE): 1 = at once, at the BEGINNING of the backward, where the branch
# is idle - measured C2 8.376 -> 8.33 ms, but C5 (4 scenes: the GEMMs are 4 x larger and hold the first table kernels up) 26.15 ->
# 26.61 ms; 2 (default) = at the END of the backward with the layers' weight gradients (attention.flush_layer_params_on_side):
# C2 8.33 -> 8.26 ms (two runs each), C5 25.01 -> 24.80; 0 = on the main stream inside the heads' backward
_HEADS_SIDE = os.environ.get("VDETR_HEADS_SIDE", "2") != "0"
_HEADS_SIDE_LATE = os.environ.get("VDETR_HEADS_SIDE", "2") == "2"


class _DeferredHeads(torch.autograd.Function):
    """Backward of the box heads of SEVERAL decoder stages as one batched pass.

    Forward: nothing to compute -- the stages' heads and box decodes already ran, launch for launch as before but without
    autograd nodes (their boxes feed the next layer detached, :408-415, so only the loss needs their graph); this node just
    hands their results to autograd.  It is created after the last decoder layer, hence it is the first thing the backward
    pass reaches: all stages' box-decode backward launches, then ONE batched GEMM per (layer of the MLP, operand) for all
    stages and heads (S x 5 problems each) instead of S separate chains -- 8 stages: ~40 launches instead of ~100, and the
    GEMMs are 8 x larger.  inputs: S feature tensors [nQ,B,C] followed by 8 parameter aliases per stage
    (w1, g1, b1, w2, g2, b2, w3, b3)."""

    @staticmethod
    def forward(ctx, records, names, *tensors):
        ctx.records, ctx.names = records, names
        S = len(records)
        ctx.set_materialize_grads(False)
        outs, nondiff = [], []
        for rec in records:
            o = rec["outs"]
            for k in names:
                outs.append(o[k])
                if k in box_decode._JOINT_NONDIFF:
                    nondiff.append(o[k])
        ctx.mark_non_differentiable(*nondiff)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        recs, names = ctx.records, ctx.names
        S, K = len(recs), len(names)
        r0 = recs[0]
        Bsz, G, rows, N = r0["y"].shape
        C = r0["f"].shape[1]
        dev = r0["y"].device
        # ---- box decode backward of every stage, straight into one stacked buffer
        dY = torch.empty((S,) + tuple(r0["y"].shape), dtype=torch.float32, device=dev)
        box_decode.joint_backward_batch([(rec["meta"], rec["saved"], dict(zip(names, grads[s * K:(s + 1) * K])), dY[s])
                                         for s, rec in enumerate(recs)])  # (one launch per 8 stages)
        one = Bsz == 1

        def fold(t, ch):  # [S,B,G*ch or G,ch..,N] -> [S*G, ch, B*N]: the batch joins the contraction
            t = t.reshape(S, Bsz, G, ch, N)
            return t.reshape(S * G, ch, N) if one else t.permute(0, 2, 3, 1, 4).reshape(S * G, ch, Bsz * N)

        def stack(k):  # the forward wrote h1 / h2 of stage s into slice s of one buffer when all stages were deferred
            st = [r["stack"] for r in recs]
            if k != "f" and all(x is not None and x[0] is st[0][0] and x[1] == i for i, x in enumerate(st)) and \
                    st[0][0][k].shape[0] == S:
                return st[0][0][k]
            return recs[0][k].unsqueeze(0) if S == 1 else torch.stack([r[k] for r in recs])

        h2, h1, f = stack("h2"), stack("h1"), stack("f")                      # [S,B,G*C,N] x2, [S,B,C,N]
        # (aliases of the parameter memory when dist.FlatParams laid the stages out next to each other, copies otherwise)
        w3 = stack_params([r["w3"].detach().reshape(G, rows, C) for r in recs])  # [S,G,rows,C]
        w2 = stack_params([r["w2"].detach().reshape(G, C, C) for r in recs])
        w1 = stack_params([r["w1"].detach().reshape(G * C, C) for r in recs])    # [S,G*C,C]
        # The three weight-gradient GEMMs and the output bias sum feed parameters only.  Where this step's table gradients run on
        # a side branch (attention.side_branch_in_use) and parameter gradients are delivered at the flush anyway, they go to that
        # branch - it is idle until the first decoder layer's key-side pass - behind ONE fork after the last input-gradient
        # operand exists: 180 us of GEMMs leave the chain (VDETR_HEADS_SIDE=1; off by default, see _HEADS_SIDE)
        from .helpers import DeferredParamGrads
        # (the first stage's node is the LAST thing the backward pass reaches: its weight gradients in line, under the side branch's
        #  tail, not behind it)
        on_side = (_HEADS_SIDE and dY.is_cuda and DeferredParamGrads.enabled and DeferredParamGrads.direct
                   and A.side_branch_in_use(dev) and not r0.get("inline_wg", False))
        late = _HEADS_SIDE_LATE and not r0.get("wg_side_now", False)

        def weight_grads(dx2, dx1):
            db3 = dY.sum(dim=(1, 4))                                                               # [S,G,rows]
            dw3 = torch.bmm(fold(dY, rows), fold(h2, C).transpose(1, 2)).view(S, G, rows, C)
            dw2 = torch.bmm(fold(dx2, C), fold(h1, C).transpose(1, 2)).view(S, G, C, C)
            if one:
                dw1 = torch.bmm(dx1.view(S, G * C, N), f.view(S, C, N).transpose(1, 2))           # [S,G*C,C]
            else:
                dw1 = torch.bmm(dx1.permute(0, 2, 1, 3).reshape(S, G * C, Bsz * N),
                                f.permute(0, 2, 1, 3).reshape(S, C, Bsz * N).transpose(1, 2))
            return db3, dw3, dw2, dw1

        # ---- output layer
        dh2 = torch.matmul(w3.transpose(2, 3).unsqueeze(1), dY).view(S, Bsz, G * C, N)             # [S,B,G*C,N]
        # ---- second hidden block
        dx2 = torch.empty_like(dh2)
        dbn2 = BNA.backward_from_records([r["bn2"] for r in recs], [dh2[s] for s in range(S)], [dx2[s] for s in range(S)])
        dh1 = torch.matmul(w2.transpose(2, 3).unsqueeze(1), dx2.view(S, Bsz, G, C, N)).view(S, Bsz, G * C, N)
        # ---- first hidden block
        dx1 = torch.empty_like(dh1)
        dbn1 = BNA.backward_from_records([r["bn1"] for r in recs], [dh1[s] for s in range(S)], [dx1[s] for s in range(S)])
        fork = None
        if on_side and not late:  # (recorded here, waited for behind the chain's next launch: the chain keeps its queue)
            fork = torch.cuda.Event()
            fork.record(torch.cuda.current_stream(dev))
        if one:  # the transposed product: rows = queries, i.e. the [nQ,B,C] layout the layers want (no permuted view to copy)
            dft = torch.bmm(dx1.view(S, G * C, N).transpose(1, 2), w1)                             # [S,N,C]
            out = [dft[s].view(N, 1, C) for s in range(S)]
        else:
            df = torch.matmul(w1.transpose(1, 2).unsqueeze(1), dx1)                                # [S,B,C,N]
            out = [df[s].permute(2, 0, 1) for s in range(S)]                                       # as [nQ,B,C]
        if on_side:
            def pairs_of(db3, dw3, dw2, dw1):
                pairs = []
                for s, r in enumerate(recs):
                    pairs += [(r["w1"], dw1[s].reshape(r["w1"].shape)), (r["w2"], dw2[s].reshape(r["w2"].shape)),
                              (r["w3"], dw3[s].reshape(r["w3"].shape)), (r["b3"], db3[s].reshape(r["b3"].shape))]
                return pairs
            for s in range(S):
                out += [None, dbn1[s][1], dbn1[s][2], None, dbn2[s][1], dbn2[s][2], None, None]
            if late:  # at the END of the backward, behind the last table kernel (attention.flush_layer_params_on_side)
                A.side_late.append((lambda: pairs_of(*weight_grads(dx2, dx1)), (dY, h2, h1, f, dx2, dx1)))
                return (None, None, *out)
            side = A._side_stream(dev)
            side.wait_event(fork)
            with torch.cuda.stream(side):
                pairs = pairs_of(*weight_grads(dx2, dx1))
            A.SideResults.pending.append((dev, pairs, (dY, h2, h1, f, dx2, dx1)))
            return (None, None, *out)
        db3, dw3, dw2, dw1 = weight_grads(dx2, dx1)
        for s, r in enumerate(recs):
            out += [dw1[s].reshape(r["w1"].shape), dbn1[s][1], dbn1[s][2], dw2[s].reshape(r["w2"].shape), dbn2[s][1],
                    dbn2[s][2], dw3[s].reshape(r["w3"].shape), db3[s].reshape(r["b3"].shape)]
        return (None, None, *out)



class _LayersDone(torch.autograd.Function):
    """identity on the decoder layers' input; its backward runs when the backward of every layer has (all their parked weight
    gradients exist then) and hands them to the side branch (attention.flush_layer_params_on_side)"""

    @staticmethod
    def forward(ctx, x):
        ctx.rows = x.shape[0] * x.shape[1]
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        A.flush_layer_params_on_side(g, ctx.rows)
        return g


class TransformerDecoder(nn.Module):
    """FFN stage on all tokens -> top-k proposals -> num_layers x GlobalDecoderLayer with per-stage box heads and
    iterative box refinement (reference :105-452).  Constructor arguments as in the reference."""

    def __init__(self, first_layer, decoder_layer, dataset_config, num_layers, decoder_dim=256, mlp_dropout=0.3,
                 mlp_norm="bn1d", mlp_act="relu", mlp_sep=False, pos_for_key=False, num_queries=256,
                 cls_loss="celoss", norm_fn_name="ln", is_bilable=False, q_content="sample",
                 return_intermediate=False, weight_init_name="xavier_uniform", args=None):
        super().__init__()
        self.first_layer = first_layer
        self.layers = get_clones(decoder_layer, num_layers)
        self.num_layers = num_layers
        self.dec_output_dim = self.layers[0].linear2.out_features
        self.norm = NORM_DICT[norm_fn_name](self.dec_output_dim) if norm_fn_name is not None else None
        self.is_bilable = is_bilable
        self.pos_for_key = pos_for_key
        self.num_queries = num_queries
        self.q_content = q_content
        self.query_pos_projection = nn.ModuleList(
            PositionEmbeddingLearned(6, self.dec_output_dim) for _ in range(num_layers))
        if pos_for_key:
            self.key_pos_projection = nn.ModuleList(
                PositionEmbeddingLearned(3, self.dec_output_dim) for _ in range(num_layers))
        if q_content in ("random", "random_add"):
            self.query_embed = nn.Embedding(num_queries, self.dec_output_dim)
        self.return_intermediate = return_intermediate
        # keys of the cross attention are visited in Morton order (performance only: attention is invariant to it)
        self.sort_keys = True
        self._reset_parameters(weight_init_name)  # before the heads are built, as in the reference (:151-159)

        self.mlp_norm, self.mlp_act, self.mlp_sep, self.cls_loss = mlp_norm, mlp_act, mlp_sep, cls_loss
        self.build_mlp_heads(dataset_config, decoder_dim, mlp_dropout)
        self.build_pointcls_heads(dataset_config, decoder_dim, mlp_dropout)

        heads = self.mlp_heads if mlp_sep else [self.mlp_heads] * (num_layers + 1)
        if cls_loss.split("_")[0] == "focalloss":  # focal prior: p = 0.01 (:161-167)
            bias_value = -math.log((1 - 0.01) / 0.01)
            for h in heads:
                last = h["sem_cls_head"].layers[-1]
                last.bias.data = torch.ones(last.bias.shape[0]) * bias_value
        for h in heads:  # centre / size regressors start at zero offset (:169-173)
            for name in ("center_head", "size_head"):
                nn.init.constant_(h[name].layers[-1].weight.data, 0.0)
                nn.init.constant_(h[name].layers[-1].bias.data, 0.0)
        self.box_processor = BoxProcessor(dataset_config, cls_loss=cls_loss)

    def _mlp(self, decoder_dim, mlp_dropout, output_dim):
        return GenericMLP(input_dim=decoder_dim, hidden_dims=[decoder_dim, decoder_dim], output_dim=output_dim,
                          norm_fn_name=self.mlp_norm, activation=self.mlp_act, use_conv=True, dropout=mlp_dropout)

    def build_pointcls_heads(self, dataset_config, decoder_dim, mlp_dropout):
        ncls = dataset_config.num_semcls + (0 if self.cls_loss.split("_")[0] == "focalloss" else 1)
        self.pointcls_heads = self._mlp(decoder_dim, mlp_dropout, ncls)

    def build_mlp_heads(self, dataset_config, decoder_dim, mlp_dropout):
        mk = partial(self._mlp, decoder_dim, mlp_dropout)
        ncls = dataset_config.num_semcls + (0 if self.cls_loss.split("_")[0] == "focalloss" else 1)
        heads = [("sem_cls_head", mk(ncls)), ("center_head", mk(3)), ("size_head", mk(3)),
                 ("angle_cls_head", mk(dataset_config.num_angle_bin)),
                 ("angle_residual_head", mk(dataset_config.num_angle_bin))]
        if not self.mlp_sep:
            self.mlp_heads = nn.ModuleDict(heads)
        elif self.is_bilable:  # stage 0 is a binary objectness head (:226-230)
            self.mlp_heads = get_clones(nn.ModuleDict(heads), self.num_layers)
            first = copy.deepcopy(heads)
            first[0] = ("sem_cls_head", mk(1))
            self.mlp_heads.insert(0, nn.ModuleDict(first))
        else:
            self.mlp_heads = get_clones(nn.ModuleDict(heads), self.num_layers + 1)

    # ---- the five head MLPs of a stage as ONE batched MLP ---------------------------------------------------------
    _HEAD_NAMES = ("sem_cls_head", "center_head", "size_head", "angle_cls_head", "angle_residual_head")

    @staticmethod
    def _batchable(heads):
        """True when the five heads are the Conv1d-BatchNorm1d-ReLU-Dropout x2 -> Conv1d stacks `_mlp` builds (SyncBN
        conversion or another norm / activation falls back to the per-head modules)."""
        for n in TransformerDecoder._HEAD_NAMES:
            l = heads[n].layers
            if len(l) != 9:
                return False
            kinds = (PointwiseConv1d, nn.BatchNorm1d, nn.ReLU, nn.Dropout, PointwiseConv1d, nn.BatchNorm1d, nn.ReLU,
                     nn.Dropout, PointwiseConv1d)
            if any(type(m) is not k for m, k in zip(l, kinds)) or l[0].bias is not None or l[4].bias is not None:
                return False
        return True

    @staticmethod
    def _bn_group(x, bns, training):
        """BatchNorm1d of G side-by-side channel groups in one call (per-channel statistics: identical numbers).
        Parameters / running statistics that are adjacent in memory are used in place (no cat, no copy-back)."""
        w = cat_params([b.weight for b in bns])
        bias = cat_params([b.bias for b in bns])
        rm, rv = buffers_alias([b.running_mean for b in bns]), buffers_alias([b.running_var for b in bns])
        if rm is None or rv is None:
            if training and x.is_cuda:  # first call on this device: put the statistics next to each other, once
                with torch.no_grad():
                    for name in ("running_mean", "running_var"):
                        flat = torch.cat([getattr(b, name) for b in bns])
                        c = bns[0].num_features
                        for i, b in enumerate(bns):
                            setattr(b, name, flat[i * c:(i + 1) * c])
                rm, rv = buffers_alias([b.running_mean for b in bns]), buffers_alias([b.running_var for b in bns])
        inplace = rm is not None and rv is not None
        if not inplace:
            rm = torch.cat([b.running_mean for b in bns])
            rv = torch.cat([b.running_var for b in bns])
        if training and BNA.sync_active():  # cross-replica statistics on this path too (incl. the first call of a device)
            y = BNA.sync_batch_norm(x, w, bias, rm, rv, None, bns[0].momentum, bns[0].eps)
        else:
            y = F.batch_norm(x, rm, rv, w, bias, training, bns[0].momentum, bns[0].eps)
        if training:
            with torch.no_grad():
                if not inplace:  # hand the updated statistics back to the modules that own them
                    c = bns[0].num_features
                    torch._foreach_copy_([b.running_mean for b in bns], list(rm.split(c)))
                    torch._foreach_copy_([b.running_var for b in bns], list(rv.split(c)))
                torch._foreach_add_([b.num_batches_tracked for b in bns], 1)
        return y

    @staticmethod
    def _bn_salt(heads, i):
        """dropout stream of hidden block i of a group of heads; kept ON the module group so that copies of a model that
        has run draw the same masks (a dictionary keyed by id() would not survive copy.deepcopy)"""
        salts = heads.__dict__.setdefault("_vdetr_bn_salts", {})
        if i not in salts:
            salts[i] = BNA.new_salt()
        return salts[i]

    def _bn_relu_drop(self, x, bns, p, key):
        """dropout(relu(batch_norm(x))) of G side-by-side channel groups: one HIP launch when training on the GPU with
        the statistics laid out adjacently, the ATen composition otherwise."""
        if not (self.training and x.is_cuda and bns[0].momentum is not None and bns[0].track_running_stats):
            return F.dropout(F.relu(self._bn_group(x, bns, self.training)), p, self.training)
        rm, rv = buffers_alias([b.running_mean for b in bns]), buffers_alias([b.running_var for b in bns])
        if rm is None or rv is None:
            return F.dropout(F.relu(self._bn_group(x, bns, self.training)), p, self.training)  # (lays them out, once)
        salt = self._bn_salt(*key)
        return BNA.bn_act(x, cat_params([b.weight for b in bns]), cat_params([b.bias for b in bns]), rm, rv, True,
                          bns[0].eps, bns[0].momentum, relu=True, dropout_p=p, salt=salt,
                          counters=[b.num_batches_tracked for b in bns])

    def _head_layers(self, stage):
        heads = self.mlp_heads[stage] if self.mlp_sep else self.mlp_heads
        return [heads[n].layers for n in self._HEAD_NAMES]

    def flat_param_groups(self):
        """Parameter groups the batched GEMMs read as ONE tensor (dist.FlatParams lays each group out contiguously,
        helpers.cat_params / stack_params / slot_stack_params then alias the memory instead of copying it)."""
        groups = []
        stages = list(range(len(self.mlp_heads))) if self.mlp_sep else [0]
        stages = [st for st in stages if self._batchable(self.mlp_heads[st] if self.mlp_sep else self.mlp_heads)]
        # stages whose heads have the same shapes share one group per parameter kind, stage after stage: the five heads of a
        # stage stay adjacent (the per-stage batched GEMMs) AND the stages do (the stacked operands of _DeferredHeads)
        by_shape = {}
        for st in stages:
            L = self._head_layers(st)
            by_shape.setdefault(tuple(tuple(l[8].weight.shape) for l in L), []).append(st)
        for sts in by_shape.values():
            Ls = [self._head_layers(st) for st in sts]
            C = Ls[0][0][0].weight.shape[1]
            rows = max(l[8].weight.shape[0] for l in Ls[0])
            pick = lambda f: [f(l) for L in Ls for l in L]  # noqa: E731
            groups += [(pick(lambda l: l[0].weight), None), (pick(lambda l: l[1].weight), None),
                       (pick(lambda l: l[1].bias), None), (pick(lambda l: l[4].weight), None),
                       (pick(lambda l: l[5].weight), None), (pick(lambda l: l[5].bias), None),
                       (pick(lambda l: l[8].weight), rows * C), (pick(lambda l: l[8].bias), rows)]
        cross = [l.multihead_attn for l in self.layers]
        if all(type(m) is GlobalShareCrossAttention for m in cross):
            groups.append(([t for m in cross for t in (m.k.weight, m.v.weight)], None))
            if cross[0].k.bias is not None:
                groups.append(([t for m in cross for t in (m.k.bias, m.v.bias)], None))
            mlps = [mm for m in cross for mm in m.cpb_mlps]
            groups += [([mm[0].weight for mm in mlps], None), ([mm[0].bias for mm in mlps], None),
                       ([mm[2].weight for mm in mlps], None)]
        return groups

    def _run_heads(self, heads, feats):
        """{head name: [B, out, N]} for feats [B, C, N].  The reference runs five independent GenericMLPs on the
        same input (:261-285); their first layers are one [5C x C] GEMM, their second layers one batched GEMM, the
        BatchNorms one call over 5C channels, the output layers one zero-padded batched GEMM when the parameters
        are laid out for it — a fraction of the launches and better-shaped GEMMs, with the parameters still living (and
        checkpointing) in the per-head modules."""
        names = self._HEAD_NAMES
        if not self._batchable(heads):
            return {n: heads[n](feats) for n in names}
        L = [heads[n].layers for n in names]
        G, C = len(L), feats.shape[1]
        Bsz, _, N = feats.shape
        w1 = cat_params([l[0].weight for l in L]).squeeze(-1)                            # [G*C, C]
        x = torch.mm(w1, feats.reshape(C, N)).unsqueeze(0) if Bsz == 1 else torch.matmul(w1, feats)  # [B, G*C, N]
        x = self._bn_relu_drop(x, [l[1] for l in L], L[0][3].p, (heads, 1))
        w2 = stack_params([l[4].weight for l in L]).squeeze(-1)                          # [G, C, C]
        x = torch.matmul(w2.unsqueeze(0), x.view(Bsz, G, C, N)).view(Bsz, G * C, N)
        x = self._bn_relu_drop(x, [l[5] for l in L], L[0][7].p, (heads, 2))
        outs = [l[8].weight.shape[0] for l in L]
        rows = max(outs)
        w3 = slot_stack_params([l[8].weight for l in L], rows)                            # [G, rows, C, 1] or None
        b3 = slot_stack_params([l[8].bias for l in L], rows) if w3 is not None and L[0][8].bias is not None else None
        if w3 is not None and (b3 is not None or L[0][8].bias is None):
            # output layers of the five heads as one batched GEMM over zero-padded [rows, C] weight slabs
            y = torch.matmul(w3.squeeze(-1).unsqueeze(0), x.view(Bsz, G, C, N))           # [B, G, rows, N]
            if b3 is not None:
                y = y + b3.view(1, G, rows, 1)
            return {"_joint": (y, outs)}  # the box decode reads the slabs in place (and writes their gradient whole)
        # unbind (one stack kernel in backward) rather than five slices (five zero-fills + copies + adds)
        xs = x.view(Bsz, G, C, N).unbind(1)
        return {n: L[g][8](xs[g]) for g, n in enumerate(names)}

    def _heads_recordable(self, heads, feats):
        """the conditions under which a stage's heads can run without autograd nodes and be differentiated later by
        _DeferredHeads: the batched layout with joint output slabs, fused BatchNorm launches, training on the GPU"""
        if not (self.training and feats.is_cuda and torch.is_grad_enabled() and self._batchable(heads)):
            return False
        L = [heads[n].layers for n in self._HEAD_NAMES]
        bns = [l[1] for l in L] + [l[5] for l in L]
        if any(b.momentum is None or not b.track_running_stats for b in bns) or L[0][8].bias is None:
            return False
        for grp in ([l[1] for l in L], [l[5] for l in L]):
            if buffers_alias([b.running_mean for b in grp]) is None or buffers_alias([b.running_var for b in grp]) is None:
                return False
        return True

    def _run_heads_recorded(self, heads, feats, slot=None, seq=None):
        """_run_heads for one stage with every launch outside autograd; returns (y [B,5,rows,N], chans, record).  ``slot`` =
        (shared dictionary, index, count): the hidden activations go straight into slice `index` of buffers stacked over
        the `count` deferred stages, which is how the batched backward wants them.  ``seq``: the same features as the dense
        sequence-first tensor [N,B,C] they are a view of — where given and the shapes fit, the stage's heads are the three
        launches of csrc/heads.hip (same values, same records; heads.py)."""
        names = self._HEAD_NAMES
        L = [heads[n].layers for n in names]
        G, C = len(L), feats.shape[1]
        Bsz, _, N = feats.shape
        outs = [l[8].weight.shape[0] for l in L]
        rows = max(outs)
        # parameter aliases are created WITH autograd (they are inputs of the deferred node)
        w1 = cat_params([l[0].weight for l in L])
        g1, b1 = cat_params([l[1].weight for l in L]), cat_params([l[1].bias for l in L])
        w2 = stack_params([l[4].weight for l in L])
        g2, b2 = cat_params([l[5].weight for l in L]), cat_params([l[5].bias for l in L])
        w3 = slot_stack_params([l[8].weight for l in L], rows)
        b3 = slot_stack_params([l[8].bias for l in L], rows)
        if w3 is None or b3 is None:
            return None
        salts = [self._bn_salt(heads, i) for i in (1, 2)]
        stk = None
        if slot is not None:
            stk, si, ns = slot
            for k in ("h1", "h2"):
                if k not in stk:
                    stk[k] = feats.new_empty((ns, Bsz, G * C, N))
        if seq is not None and HD.heads_usable(seq, L, rows):
            with torch.no_grad():
                f = feats.detach()
                rm1, rv1 = buffers_alias([l[1].running_mean for l in L]), buffers_alias([l[1].running_var for l in L])
                rm2, rv2 = buffers_alias([l[5].running_mean for l in L]), buffers_alias([l[5].running_var for l in L])
                h1 = stk["h1"][si] if stk is not None else feats.new_empty((Bsz, G * C, N))
                h2 = stk["h2"][si] if stk is not None else feats.new_empty((Bsz, G * C, N))
                y, bn1, bn2 = HD.heads_forward(seq.detach(), L, (g1, b1, g2, b2, w3, b3), (rm1, rv1, rm2, rv2), salts, rows, h1, h2)
            record = {"f": f, "h1": h1, "h2": h2, "bn1": bn1, "bn2": bn2, "y": y, "stack": (stk, si) if stk is not None else None,
                      "w1": w1, "g1": g1, "b1": b1, "w2": w2, "g2": g2, "b2": b2, "w3": w3, "b3": b3}
            return y, outs, record
        with torch.no_grad():
            f = feats.detach()  # [B,C,N] view of the [nQ,B,C] layer output: the GEMM reads it transposed, no copy
            w1d = w1.detach().squeeze(-1)
            x = torch.mm(w1d, f.reshape(C, N)).unsqueeze(0) if Bsz == 1 else torch.matmul(w1d, f)
            bns = [l[1] for l in L]
            h1, bn1 = BNA.forward_record(x, g1, b1, buffers_alias([b.running_mean for b in bns]),
                                         buffers_alias([b.running_var for b in bns]), bns[0].eps, bns[0].momentum, L[0][3].p,
                                         salts[0], counters=[b.num_batches_tracked for b in bns],
                                         y_out=stk["h1"][si] if stk is not None else None)
            x = torch.matmul(w2.detach().squeeze(-1).unsqueeze(0), h1.view(Bsz, G, C, N)).view(Bsz, G * C, N)
            bns = [l[5] for l in L]
            h2, bn2 = BNA.forward_record(x, g2, b2, buffers_alias([b.running_mean for b in bns]),
                                         buffers_alias([b.running_var for b in bns]), bns[0].eps, bns[0].momentum, L[0][7].p,
                                         salts[1], counters=[b.num_batches_tracked for b in bns],
                                         y_out=stk["h2"][si] if stk is not None else None)
            y = torch.matmul(w3.detach().squeeze(-1).unsqueeze(0), h2.view(Bsz, G, C, N))
            y = y + b3.detach().view(1, G, rows, 1)
        record = {"f": f, "h1": h1, "h2": h2, "bn1": bn1, "bn2": bn2, "y": y, "stack": (stk, si) if stk is not None else None,
                  "w1": w1, "g1": g1, "b1": b1, "w2": w2, "g2": g2, "b2": b2, "w3": w3, "b3": b3}
        return y, outs, record

    def _stage_recorded(self, idx, point_cloud_dims, box_features, pre_center_normalized, pre_size_normalized, slot=None):
        """get_proposal_box_predictions_refine for a stage whose autograd graph is built later (_DeferredHeads).  Returns
        (box prediction dictionary of plain tensors, record) or None when the stage does not qualify."""
        heads = self.mlp_heads[idx] if self.mlp_sep else self.mlp_heads
        feats = box_features.permute(1, 2, 0)
        if not self._heads_recordable(heads, feats):
            return None
        ran = self._run_heads_recorded(heads, feats, slot, box_features if box_features.is_contiguous() else None)
        if ran is None:
            return None
        y, chans, record = ran
        outs, saved, meta = box_decode.decode_boxes_joint_record(
            y, chans, pre_center_normalized, pre_size_normalized, point_cloud_dims,
            self.box_processor.dataset_config.num_angle_bin, self.box_processor.cls_loss)
        record.update(outs=outs, saved=saved, meta=meta, feats_in=box_features)
        return box_decode.joint_result(outs), record

    @staticmethod
    def _attach_deferred(records, predictions):
        """One autograd node for the recorded stages; the dictionaries get the tracked tensors."""
        names = tuple(k for k in box_decode._BoxDecodeJoint.OUTS if records[0]["outs"][k] is not None)
        tensors = [r["feats_in"] for r in records]
        for r in records:
            tensors += [r["w1"], r["g1"], r["b1"], r["w2"], r["g2"], r["b2"], r["w3"], r["b3"]]
        flat = _DeferredHeads.apply(records, names, *tensors)
        K = len(names)
        for s, (rec, pred) in enumerate(zip(records, predictions)):
            o = dict(rec["outs"])
            o.update(zip(names, flat[s * K:(s + 1) * K]))
            tracked = box_decode.joint_result(o)
            for k in list(pred.keys()):
                if k in tracked and not k.startswith("_"):
                    pred[k] = tracked[k]


    def _reset_parameters(self, weight_init_name):
        init = WEIGHT_INIT_DICT[weight_init_name]
        for _, p in self.named_parameters():
            if p.dim() > 1:
                init(p)

    def get_proposal_box_predictions_refine(self, idx, query_xyz, point_cloud_dims, box_features,
                                            pre_center_normalized=None, pre_size_normalized=None):
        """Stage-`idx` heads on box_features [nQ,B,C]; boxes are decoded RELATIVE to the given (normalised) prior
        centre / size: centre = reg * size_prior + centre_prior, size = exp(reg) * size_prior (reference :244-333)."""
        assert pre_center_normalized is not None and pre_size_normalized is not None
        feats = box_features.permute(1, 2, 0)  # B x C x nQ
        heads = self.mlp_heads[idx] if self.mlp_sep else self.mlp_heads
        raw = self._run_heads(heads, feats)
        if "_joint" in raw:
            y, chans = raw["_joint"]
            return box_decode.decode_boxes_joint(y, chans, pre_center_normalized, pre_size_normalized, point_cloud_dims,
                                                 self.box_processor.dataset_config.num_angle_bin,
                                                 self.box_processor.cls_loss)
        # everything after the heads (:286-333: ~45 ATen launches on [B,nQ,<=24] tensors, as many again in backward)
        # is one HIP launch forward and one backward
        return box_decode.decode_boxes(raw, pre_center_normalized, pre_size_normalized, point_cloud_dims,
                                       self.box_processor.dataset_config.num_angle_bin, self.box_processor.cls_loss)

    def forward(self, tgt, memory, query_xyz, enc_xyz, point_cloud_dims, tgt_mask=None, memory_mask=None,
                tgt_key_padding_mask=None, memory_key_padding_mask=None, pos=None, query_pos=None,
                transpose_swap=False, return_attn_weights=False, enc_box_predictions=None, enc_box_features=None):
        A.begin_step(memory.device)  # one dropout-RNG snapshot per forward, shared by all attention modules
        from .runtime import ts_mark  # (no-ops unless VDETR_TS_PROBE=1: tools/probes/step_timeline.py)
        ts_mark("decoder: start")
        intermediate, attns = [], []
        fuse_ln = ALN.supported(self.norm) and isinstance(self.first_layer, FFNLayer) and \
            ALN.supported(self.first_layer.norm, self.norm)
        self.first_layer.post_norm = self.norm if fuse_ln else None
        defer = _DEFER_HEADS and self.mlp_sep and self.return_intermediate and len(self.layers) > 1
        if defer and self.training and enc_box_features.is_cuda and torch.is_grad_enabled():
            # the W^T images of every stage's heads, of the position MLPs and of the first layer's FFN: one launch (heads.py)
            HD.decoder_refresh(self)
        output = self.first_layer(enc_box_features)
        normed = self.first_layer.post_normed if fuse_ln else self.norm(output)
        self.first_layer.post_norm = self.first_layer.post_normed = None
        # the first stage's heads on all encoder tokens: recorded like the later stages' (the fused launches of csrc/heads.hip where
        # the shapes fit) and differentiated by a _DeferredHeads node of its own — the last thing the backward pass reaches
        recorded = self._stage_recorded(0, point_cloud_dims, normed, enc_box_predictions["center_normalized"],
                                        enc_box_predictions["size_normalized"]) if (defer and _DEFER_STAGE0) else None
        if recorded is not None:
            box_prediction, rec0 = recorded
            # the first stage's node is the LAST the backward reaches: its weight gradients in line (VDETR_STAGE0_WG=side: on the
            # side branch at once, behind whatever that still has queued)
            if _STAGE0_WG_SIDE:
                rec0["wg_side_now"] = True
            else:
                rec0["inline_wg"] = True
            self._attach_deferred([rec0], [box_prediction])
        else:
            box_prediction = self.get_proposal_box_predictions_refine(
                0, query_xyz, point_cloud_dims, normed,
                pre_center_normalized=enc_box_predictions["center_normalized"],
                pre_size_normalized=enc_box_predictions["size_normalized"])
        if self.return_intermediate:
            intermediate.append(box_prediction)

        ts_mark("decoder: first stage done")
        # ---- top-k proposals by objectness (:364-398) ----------------------------------------------------
        objectness = box_prediction["objectness_prob"].detach()
        ntok = objectness.shape[1]
        if ntok >= self.num_queries:
            # the reference takes torch.topk (:364-366), whose order among EQUAL values is the implementation's choice — and at
            # 4096 tokens two fp32 objectness values coincide in about every scene (birthday bound: 4096^2 / 2^24), so a CPU and
            # a GPU run of the same weights rank those two proposals differently and hand them each other's query embedding
            # (measured: ranks 882 / 883 of BASELINE config 2's scene).  A stable descending sort is one of the orders topk may
            # return, and the same one on every device: ties go to the lower token index.
            topk = _proposal_order(objectness, self.num_queries)
        else:
            topk = torch.arange(ntok, device=objectness.device).unsqueeze(0).repeat(objectness.shape[0], 1)

        def take(t):
            t = t.detach()
            index = topk.view(topk.shape + (1,) * (t.dim() - 2)).expand(topk.shape + t.shape[2:])
            return torch.gather(t, 1, index)

        proposals_qref = None
        if box_decode.proposals_fusable(topk, box_prediction):  # the seven gathers, the frame change and the cat as one launch
            (reference_point, reference_center, reference_size, reference_angle, proposal_center_normalized,
             proposal_size_normalized, proposals_qref) = box_decode.gather_proposals(topk, box_prediction)
            box_prediction.pop("_reference_point_lidar", None)
            box_prediction.pop("_query_reference", None)
        else:
            reference_point = convert_corners_camera2lidar(take(box_prediction["box_corners"]))
            reference_center = take(box_prediction["center_unnormalized"])
            reference_size = take(box_prediction["size_unnormalized"])
            reference_angle = take(box_prediction["angle_continuous"])
            proposal_center_normalized = take(box_prediction["center_normalized"])
            proposal_size_normalized = take(box_prediction["size_normalized"])
        query_xyz = reference_center
        batch = output.shape[1]
        if self.q_content == "zero":
            output = output.new_zeros((topk.shape[1], batch, output.shape[-1]))
        elif self.q_content == "random":  # query content is a learned embedding; first-stage features are dropped
            output = self.query_embed.weight.unsqueeze(1).repeat(1, batch, 1)
        else:
            gathered = torch.gather(output.permute(1, 0, 2), 1,
                                    topk.unsqueeze(-1).expand(-1, -1, output.shape[-1])).permute(1, 0, 2).contiguous()
            output = gathered + self.query_embed.weight.unsqueeze(1) if self.q_content == "random_add" else gathered

        if output.requires_grad and output.is_cuda:
            output = _LayersDone.apply(output)
        # ---- keys in Morton order for the RPE kernels (memory / enc_xyz feed only the cross attention) -----------
        key_order = None
        if self.sort_keys and memory_mask is None and memory_key_padding_mask is None and pos is None:
            key_order = morton_argsort(enc_xyz.detach())                                   # B,nK
            enc_xyz = torch.gather(enc_xyz, 1, key_order.unsqueeze(-1).expand(-1, -1, 3))
            memory = torch.gather(memory, 0, key_order.t().unsqueeze(-1).expand(-1, -1, memory.shape[-1]))

        # ---- K / V / RPE tables of all layers in one go: every layer projects the SAME encoder features ----------
        cross = [l.multihead_attn for l in self.layers]
        caches = [None] * len(self.layers)
        if (not self.pos_for_key and all(type(m) is GlobalShareCrossAttention for m in cross) and
                not any(l.pos_for_key for l in self.layers)):
            caches = GlobalShareCrossAttention.precompute(cross, memory)

        # ---- decoder layers with box feedback (:407-436) ---------------------------------------------------
        fuse_ln = ALN.supported(self.norm) and all(
            isinstance(l, GlobalDecoderLayer) and l.normalize_before and ALN.supported(self.norm, l.norm1, l.norm2, l.norm3)
            for l in self.layers)
        carried = None  # norm1 of the next layer, produced by the previous layer's last launch
        rb_layers = bool(fuse_ln and _ROWBLOCK and output.is_cuda and memory_mask is None and
                         all(RB.usable(l, output, None, ()) and not l.pos_for_key for l in self.layers))
        if rb_layers:
            RB.refresh(self.layers)  # the fused glue launches' weight images of all layers: one launch (rowblock.py)
        if not defer and self.training and output.is_cuda and torch.is_grad_enabled():
            HD.decoder_refresh(self)  # (the position MLPs' images; the stages' heads take the batched path)
        deferred, stacked = [], {}
        ts_mark("decoder: proposals, key order, K / V / tables, weight images done")
        for idx, layer in enumerate(self.layers):
            layer.cross_cache = caches[idx]
            if fuse_ln:
                layer.pre_normed = carried
                nxt = self.layers[idx + 1].norm1 if idx + 1 < len(self.layers) else None
                layer.post_norms = (self.norm,) + ((nxt,) if nxt is not None else ())
            own_boxes = False
            if idx > 0:
                reference_point = box_prediction.pop("_reference_point_lidar", None)  # written by the fused box decode
                # corners out of box_decode.hip are AXIS-ALIGNED boxes only for a dataset without angle bins: with num_angle_bin > 1
                # the decode rotates them by the predicted angle, and unless the attention works in the object's frame
                # (angle_type "object_coords": cos / sin travel with the call and the kernels undo the rotation) the general
                # table-gradient kernel must stay in front — vouching here would let the box-only kernel poison dtable with NaN
                own_boxes = reference_point is not None and self.box_processor.dataset_config.num_angle_bin == 1
                if reference_point is None:
                    reference_point = convert_corners_camera2lidar(box_prediction["box_corners"].detach())
                reference_center = box_prediction["center_unnormalized"].detach()
                reference_size = box_prediction["size_unnormalized"].detach()
                reference_angle = box_prediction["angle_continuous"].detach()
            query_reference = box_prediction.pop("_query_reference", None) if idx > 0 else proposals_qref
            if query_reference is None:
                query_reference = torch.cat([reference_center, reference_size], dim=-1)
            # (rb_layers: the layer runs through rowblock.py, whose q / k / v launch computes the position rows on its way in)
            prev_lazy = HD.lazy_pos(rb_layers and not return_attn_weights)
            try:
                query_pos = self.query_pos_projection[idx](query_reference).permute(2, 0, 1)
            finally:
                HD.lazy_pos(prev_lazy)
            if self.pos_for_key:
                pos = self.key_pos_projection[idx](enc_xyz).permute(2, 0, 1)
            if hasattr(layer.multihead_attn, "vertices_are_boxes"):
                layer.multihead_attn.vertices_are_boxes = own_boxes
            output, attn = layer(output, memory, reference_point, reference_angle, enc_xyz, point_cloud_dims,
                                 tgt_mask=tgt_mask, memory_mask=memory_mask,
                                 tgt_key_padding_mask=tgt_key_padding_mask,
                                 memory_key_padding_mask=memory_key_padding_mask, pos=pos, query_pos=query_pos,
                                 return_attn_weights=return_attn_weights)
            layer.cross_cache = None
            if fuse_ln:
                normed = layer.post_normed[0]
                carried = layer.post_normed[1] if len(layer.post_normed) > 1 else None
                layer.pre_normed = layer.post_norms = layer.post_normed = None
            else:
                normed = self.norm(output)
            # stages >= 1 decode relative to the FIXED stage-0 proposal centre / size (:427-431)
            recorded = self._stage_recorded(idx + 1, point_cloud_dims, normed, proposal_center_normalized,
                                            proposal_size_normalized, (stacked, idx, len(self.layers))) if defer else None
            if recorded is not None:  # launches now, autograd graph after the loop (one batched backward for all stages)
                box_prediction, rec = recorded
                deferred.append((rec, box_prediction))
            else:
                box_prediction = self.get_proposal_box_predictions_refine(
                    idx + 1, query_xyz, point_cloud_dims, normed,
                    pre_center_normalized=proposal_center_normalized, pre_size_normalized=proposal_size_normalized)
            if self.return_intermediate:
                intermediate.append(box_prediction)
            ts_mark(f"decoder: layer {idx + 1} + its stage done")
            if return_attn_weights:
                if key_order is not None:  # back to the caller's key order
                    inv = torch.argsort(key_order, dim=1)[:, None, None, :].expand_as(attn)
                    attn = torch.gather(attn, 3, inv)
                attns.append(attn)

        HD._fresh["on"] = False  # (the images are this forward's: the weights change at the next optimiser step)
        if deferred:
            self._attach_deferred([r for r, _ in deferred], [p for _, p in deferred])
        for extra in ("_reference_point_lidar", "_query_reference"):  # helpers of the loop, not part of the result
            box_prediction.pop(extra, None)
            if intermediate:
                intermediate[0].pop(extra, None)
        if return_attn_weights:
            attns = torch.stack(attns)
        if self.return_intermediate:
            return {"outputs": intermediate[-1], "aux_outputs": intermediate[:-1]}, attns
        return {"outputs": box_prediction}, attns
