"""The decoder layer's glue between its attention cores as three launches (csrc/rowblock.hip): autograd bindings.

Reference: GlobalDecoderLayer.forward_pre (models/vdetr_transformer.py:531-568).  ``qkv`` is the self-attention's in-projection
(:540-542 + nn.MultiheadAttention's in_proj), ``proj_q`` its out-projection with residual block 1 and the cross attention's
query projection (:543-545, :733), ``ffn`` the cross attention's output projection with residual blocks 2 and 3, the FFN
(:556-567, :755-757) and the norms the decoder applies to the layer output (:401, :433).

Forward: one launch each.  Backward: the launches the separate modules would run (vdetr_add_ln_bwd_f32,
vdetr_relu_dropout_bwd_f32, the input-gradient GEMMs), on the tensors the fused launch wrote for them, with the weight / bias
gradients parked exactly as helpers._Linear parks them (runtime.defer_weight_grads).  Same dropout streams as add_ln.py /
bn_act.py for the same salts: a layer computes the same values on either path.  No CPU path.
"""
import ctypes
import os

import torch

from . import _lib as L
from . import add_ln as ALN
from . import attention as A
from . import heads as HD
from .helpers import DeferredParamGrads, colsum

C = 256
# the input-gradient chains as one launch each (rb_*_bwd_kernel); VDETR_ROWBLOCK_BWD=0: the separate backward launches (A/B, parity)
FUSED_BWD = os.environ.get("VDETR_ROWBLOCK_BWD", "1") != "0"


def _seq_rows(t, B):
    """batch-first [B, nQ, C] -> the rows (q, b) of the sequence-first layout, [nQ * B, C] (a view for one scene)"""
    if B == 1:
        return t.reshape(-1, t.shape[-1])
    return t.transpose(0, 1).reshape(-1, t.shape[-1])


def _batch_first(rows2d, B):
    """[nQ * B, C] rows (q, b) -> [B, nQ, C]"""
    if B == 1:
        return rows2d.view(1, -1, rows2d.shape[-1])
    return rows2d.view(-1, B, rows2d.shape[-1]).transpose(0, 1).contiguous()


def _rng_for(p, device):
    if p <= 0.0:
        return None
    rng = A.current_rng(device)
    return rng if rng is not None else A.begin_step(device)


def _lin(d, w, b, wt=None):
    d.w = w.data_ptr()
    d.b = b.data_ptr() if b is not None else None
    d.wt = wt.data_ptr() if wt is not None else None


# ---- the forward launches' weight images (W^T, csrc/rowblock.hip) ---------------------------------------------------------
# One [8, 256, 256] buffer per layer: in_proj q | k | v, the self-attention's out_proj, the cross attention's q and proj, linear1,
# linear2.  Weights change in place at every optimiser step, so the images are rewritten at the start of every decoder forward —
# ONE launch for all layers (a node of the captured step like any other); a layer called on its own rewrites its own.
_IMG_SLOTS = 8
_tables = {}   # (layer ids) -> (pointer ints, device table of source pointers, [n_layers, 8, 256, 256] images)


def _sources(layer):
    sa, ca = layer.self_attn, layer.multihead_attn
    wq, wk, wv = sa.in_proj_weight.view(3, C, C).unbind(0)
    return [wq, wk, wv, sa.out_proj.weight, ca.q.weight, ca.proj.weight, layer.linear1.weight, layer.linear2.weight]


def refresh(layers):
    """rewrite the W^T images of `layers` (one launch) and hand each layer its [8, 256, 256] slice (consumed by its next forward)"""
    layers = list(layers)
    srcs = [w for l in layers for w in _sources(l)]
    dev = srcs[0].device
    ptrs = tuple(w.data_ptr() for w in srcs)
    key = tuple(id(l) for l in layers)
    ent = _tables.get(key)
    if ent is None or ent[0] != ptrs or ent[2].device != dev:
        for w in srcs:
            _check(w)
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("rowblock.refresh: the weight pointers changed inside a stream capture (run one step eagerly first)")
        table = torch.tensor(ptrs, dtype=torch.int64).to(dev)
        ent = (ptrs, table, torch.empty((len(layers), _IMG_SLOTS, C, C), dtype=torch.float32, device=dev))
        if len(_tables) > 64:
            _tables.clear()
        _tables[key] = ent
    L.check(L.lib().vdetr_rb_transpose_f32(ent[1].data_ptr(), ent[2].data_ptr(), len(srcs), L.stream_ptr()), "rb_transpose")
    for i, l in enumerate(layers):
        l.__dict__["_rb_images"] = ent[2][i]
    return ent[2]


def images(layer):
    """the layer's fresh images: those a decoder-level refresh() left for this forward, or rewritten now"""
    img = layer.__dict__.pop("_rb_images", None)
    if img is None or img.device != layer.linear1.weight.device:
        refresh([layer])
        img = layer.__dict__.pop("_rb_images")
    return img


def _norm(d, ln_w, ln_b, eps):
    d.gamma, d.beta, d.eps = ln_w.data_ptr(), ln_b.data_ptr(), float(eps)


def _drop(d, p, salt):
    d.p, d.seed = float(p), int(salt) & 0xFFFFFFFFFFFFFFFF


def _park_or_grad(w, b, dy2, x2, need_w, need_b):
    """weight / bias gradient of y = x w^T + b from (dy, x): parked (runtime.defer_weight_grads) or computed here"""
    if not (need_w or need_b):
        return None, None
    if DeferredParamGrads.enabled:
        DeferredParamGrads.pending.append((w if need_w else None, b if need_b else None, dy2, x2))
        return None, None
    dw = torch.mm(dy2.t(), x2) if need_w else None
    db = colsum(dy2 if dy2.stride(1) == 1 else dy2.contiguous()) if need_b else None
    return dw, db


def _ln_sums(part, nblk, ln_params, two):
    """parameter gradients of a LayerNorm from a fused backward launch's per-workgroup partial rows [nblk][4][256]: parked for the
    flush (add_ln.DeferredLnGrads) or summed here.  Returns (d_gamma, d_beta, d_gamma2, d_beta2)."""
    if DeferredParamGrads.enabled and DeferredParamGrads.direct:
        ALN.DeferredLnGrads.pending.append((part, nblk, C, ln_params, two))
        return None, None, None, None
    sums = part.view(nblk, 4, C).sum(0)
    g2 = ln_params[2]
    if two:
        return sums[0], sums[1], sums[2], sums[3]
    return sums[0], sums[1], (torch.zeros_like(g2) if g2 is not None else None), (torch.zeros_like(g2) if g2 is not None else None)


def _opt(t, rows=None):
    if t is None:
        return None
    t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone()
    return t


def _check(*ts):
    for t in ts:
        if t is None:
            continue
        L.require_gpu(t, "rowblock operand")
        L.require_float(t, "rowblock operand")
        if not t.is_contiguous() or t.data_ptr() % 16:
            raise RuntimeError("rowblock operands must be contiguous and 16-B aligned")


FUSED_FFN0 = os.environ.get("VDETR_FFN0_FUSED", "1") != "0"


def _emit_for(rec, a, d_a, rows, B, per_head):
    """the vdetr_rb_attn_emit of a backward launch that produces d_a = the output gradient of the attention call `rec` (whose forward
    output is `a`), or None where the call does not qualify (attention.py: the key-side pass without its packing launch)"""
    if rec is None or d_a is None or B != 1 or rows % 32 or rec.dims[2] != rows or (rec.kind == L.VDETR_ATTN_PER_HEAD) != bool(per_head):
        return None
    A.kv_prepare(rec)
    e = L.RbAttnEmit()
    e.workspace, e.delta, e.out = rec.ws.data_ptr(), rec.delta.data_ptr(), a.data_ptr()
    e.bwd_aux = rec.aux.data_ptr() if rec.aux is not None else None
    e.per_head, e.nQ = int(bool(per_head)), rows
    return e


class _Qkv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, pos, wq, wk, wv, bq, bk, bv, B, wt):
        _check(t, pos, wq, wk, wv, bq, bk, bv, wt)
        rows = t.numel() // C
        if not (wk.data_ptr() == wq.data_ptr() + C * C * 4 and wv.data_ptr() == wk.data_ptr() + C * C * 4 and
                bk.data_ptr() == bq.data_ptr() + C * 4 and bv.data_ptr() == bk.data_ptr() + C * 4):
            raise RuntimeError("rowblock.qkv: the q / k / v blocks must be adjacent rows of one in_proj parameter")
        out = torch.empty((3, B, rows // B, C), dtype=torch.float32, device=t.device)
        x = torch.empty((rows, C), dtype=torch.float32, device=t.device) if pos is not None else None
        d = L.RbQkvDesc()
        d.rows, d.B = rows, B
        d.t, d.pos = t.data_ptr(), (pos.data_ptr() if pos is not None else None)
        d.w, d.b, d.wt = wq.data_ptr(), bq.data_ptr(), wt.data_ptr()
        d.x, d.out = (x.data_ptr() if x is not None else None), out.data_ptr()
        pend = HD.take_pending_pos(pos)
        if pend is not None:
            # `pos` is the output of a position MLP whose launch was left to this one (heads.lazy_pos): the q / k workgroups compute
            # their rows of it on the way in, the q workgroups write it (and what the MLP's backward reads)
            L.check(L.lib().vdetr_rb_qkv_pos_f32(ctypes.byref(d), ctypes.byref(pend[0]), L.stream_ptr()), "rb_qkv_pos")
        else:
            L.check(L.lib().vdetr_rb_qkv_f32(ctypes.byref(d), L.stream_ptr()), "rb_qkv")
        ctx.B, ctx.shape = B, t.shape
        ctx.save_for_backward(t, x, wq, wk, wv, bq, bk, bv)
        ctx.set_materialize_grads(False)
        # The fourth output is `pos` again: the layer hands THIS tensor to the position's second consumer (proj_q), whose gradient
        # then arrives here as d_alias and is added to the q / k share inside the backward launch — autograd would otherwise sum the
        # two [rows, 256] gradients with a launch of its own per layer.
        alias = pos.view_as(pos) if pos is not None else None
        if alias is not None and not pos.requires_grad:
            ctx.mark_non_differentiable(alias)
        return out[0], out[1], out[2], alias

    @staticmethod
    def backward(ctx, dq, dk, dv, d_alias=None):
        t, x, wq, wk, wv, bq, bk, bv = ctx.saved_tensors
        B = ctx.B
        t2 = t.reshape(-1, C)
        x2 = x if x is not None else t2
        need = ctx.needs_input_grad
        if dq is None and dk is None and dv is None:
            return (None, d_alias if need[1] else None) + (None,) * 8
        dq, dk, dv = (g if g is not None else torch.zeros((B, t2.shape[0] // B, C), dtype=torch.float32, device=t.device) for g in (dq, dk, dv))
        if FUSED_BWD:
            rows = t2.shape[0]
            dq, dk, dv = _opt(dq), _opt(dk), _opt(dv)
            d_t = torch.empty_like(t2)
            d_x = torch.empty_like(t2) if (need[1] and x is not None) else None
            if B > 1:
                rws = torch.empty((3, rows, C), dtype=torch.float32, device=t.device)
                dq2, dk2, dv2 = rws[0], rws[1], rws[2]
            else:
                dq2, dk2, dv2 = dq.view(rows, C), dk.view(rows, C), dv.view(rows, C)
            d = L.RbQkvDesc()
            d.rows, d.B, d.w = rows, B, wq.data_ptr()
            g = L.RbQkvGrads()
            g.dq, g.dk, g.dv = dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
            if B > 1:
                g.dq_rows, g.dk_rows, g.dv_rows = dq2.data_ptr(), dk2.data_ptr(), dv2.data_ptr()
            g.d_x = d_x.data_ptr() if d_x is not None else None
            if d_alias is not None and d_x is not None:
                d_alias = _opt(d_alias)
                g.d_x_add = d_alias.data_ptr()
            g.d_t = d_t.data_ptr()
            # This is the LAST launch of the layer's backward, and its own weight-gradient operands (the incoming dq / dk / dv and
            # the saved inputs) exist already — for one scene; several scenes' row-ordered copies are written by the launch — so
            # the layer's parked gradients can leave for the side branch from here (attention.side_flush_begin)
            gr = [None] * 6
            tok = None
            if B == 1:
                for i, (w, b, dy, xx) in enumerate(((wq, bq, dq2, x2), (wk, bk, dk2, x2), (wv, bv, dv2, t2))):
                    gr[i], gr[3 + i] = _park_or_grad(w, b, dy, xx, need[2 + i], need[5 + i])
                tok = A.side_flush_begin(t2)
            L.check(L.lib().vdetr_rb_qkv_bwd_f32(ctypes.byref(d), ctypes.byref(g), L.stream_ptr()), "rb_qkv_bwd")
            if B > 1:
                for i, (w, b, dy, xx) in enumerate(((wq, bq, dq2, x2), (wk, bk, dk2, x2), (wv, bv, dv2, t2))):
                    gr[i], gr[3 + i] = _park_or_grad(w, b, dy, xx, need[2 + i], need[5 + i])
            A.side_flush_end(tok, t2, rows)
            return (d_t.view(ctx.shape) if need[0] else None, d_x.view(ctx.shape) if d_x is not None else None,
                    gr[0], gr[1], gr[2], gr[3], gr[4], gr[5], None, None)
        dq2, dk2, dv2 = (_seq_rows(g.contiguous(), B) for g in (dq, dk, dv))
        d_t = d_pos = None
        if need[0] or need[1]:
            d_x = torch.mm(dq2, wq)
            d_x.addmm_(dk2, wk)                      # gradient of t + pos
            if need[0]:
                d_t = torch.addmm(d_x, dv2, wv).view(ctx.shape)
            if need[1]:
                d_pos = (d_x if d_alias is None else d_x + d_alias.reshape(d_x.shape)).view(ctx.shape)
        g = [None] * 6
        for i, (w, b, dy, xx) in enumerate(((wq, bq, dq2, x2), (wk, bk, dk2, x2), (wv, bv, dv2, t2))):
            g[i], g[3 + i] = _park_or_grad(w, b, dy, xx, need[2 + i], need[5 + i])
        return d_t, d_pos, g[0], g[1], g[2], g[3], g[4], g[5], None, None


class _ProjQ(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, tgt, pos, wo, bo, wq, bq, g2, b2, eps, p, salt, rng, B, wot, wqt):
        _check(a, tgt, pos, wo, bo, wq, bq, g2, b2, wot, wqt)
        ctx.kv_rec = A.kv_record_of(a)  # (the self-attention call that produced a: its backward's operands are emitted here)
        rows = tgt.numel() // C
        dev = tgt.device
        y = torch.empty_like(tgt)
        t2 = torch.empty_like(tgt)
        xq = torch.empty_like(tgt) if pos is not None else None
        mean = torch.empty(rows, dtype=torch.float32, device=dev)
        rstd = torch.empty_like(mean)
        qout = torch.empty((B, rows // B, C), dtype=torch.float32, device=dev)
        d = L.RbProjQDesc()
        d.rows, d.B = rows, B
        d.rng_state = rng.data_ptr() if (rng is not None and p > 0) else None
        d.a, d.tgt, d.pos = a.data_ptr(), tgt.data_ptr(), (pos.data_ptr() if pos is not None else None)
        _lin(d.proj, wo, bo, wot)
        _lin(d.q, wq, bq, wqt)
        _drop(d.drop1, p, salt)
        _norm(d.norm2, g2, b2, eps)
        d.y, d.mean_y, d.rstd_y, d.t2 = y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), t2.data_ptr()
        d.xq, d.qout = (xq.data_ptr() if xq is not None else None), qout.data_ptr()
        L.check(L.lib().vdetr_rb_proj_q_f32(ctypes.byref(d), L.stream_ptr()), "rb_proj_q")
        ctx.cfg = (rows, float(eps), float(p), salt, B, tgt.shape)
        ctx.ln_params = (g2, b2, None, None)
        ctx.save_for_backward(a, y, xq if xq is not None else t2, wo, bo, wq, bq, g2, mean, rstd, rng if p > 0 else None)
        ctx.set_materialize_grads(False)
        return y, qout

    @staticmethod
    def backward(ctx, d_y, d_qout):
        rows, eps, p, salt, B, shape = ctx.cfg
        a, y, xq, wo, bo, wq, bq, g2, mean, rstd, rng = ctx.saved_tensors
        need = ctx.needs_input_grad
        if FUSED_BWD and (d_y is not None or d_qout is not None):
            dev = y.device
            d_y, d_qout = _opt(d_y), _opt(d_qout)
            nblk = (rows + 15) // 16
            d_tgt = torch.empty((rows, C), dtype=torch.float32, device=dev)
            d_proj = torch.empty_like(d_tgt)
            d_t2 = torch.empty_like(d_tgt) if (need[2] and d_qout is not None) else None
            d_a = torch.empty((B, rows // B, C), dtype=torch.float32, device=dev) if need[0] else None
            dq_rows = torch.empty_like(d_tgt) if (B > 1 and d_qout is not None) else None
            part = torch.empty((nblk, 4, C), dtype=torch.float32, device=dev)
            d = L.RbProjQDesc()
            d.rows, d.B = rows, B
            d.rng_state = rng.data_ptr() if (rng is not None and p > 0) else None
            _lin(d.proj, wo, None)
            _lin(d.q, wq, None)
            _drop(d.drop1, p, salt)
            _norm(d.norm2, g2, g2, eps)
            d.y, d.mean_y, d.rstd_y = y.data_ptr(), mean.data_ptr(), rstd.data_ptr()
            g = L.RbProjQGrads()
            g.d_y = d_y.data_ptr() if d_y is not None else None
            g.d_qout = d_qout.data_ptr() if d_qout is not None else None
            g.d_tgt, g.d_proj, g.part_n2 = d_tgt.data_ptr(), d_proj.data_ptr(), part.data_ptr()
            g.d_a = d_a.data_ptr() if d_a is not None else None
            g.d_t2 = d_t2.data_ptr() if d_t2 is not None else None
            g.dq_rows = dq_rows.data_ptr() if dq_rows is not None else None
            emit = _emit_for(getattr(ctx, "kv_rec", None), a, d_a, rows, B, True)
            if emit is not None:
                L.check(L.lib().vdetr_rb_proj_q_bwd_emit_f32(ctypes.byref(d), ctypes.byref(g), ctypes.byref(emit), L.stream_ptr()), "rb_proj_q_bwd_emit")
                ctx.kv_rec.emitted = d_a.data_ptr()
            else:
                L.check(L.lib().vdetr_rb_proj_q_bwd_f32(ctypes.byref(d), ctypes.byref(g), L.stream_ptr()), "rb_proj_q_bwd")
            gwq = gbq = None
            if d_qout is not None:
                gwq, gbq = _park_or_grad(wq, bq, dq_rows if dq_rows is not None else d_qout.view(rows, C), xq.reshape(-1, C), need[5], need[6])
            gwo, gbo = _park_or_grad(wo, bo, d_proj, _seq_rows(a, B), need[3], need[4])
            dg, db, _, _ = _ln_sums(part, nblk, ctx.ln_params, False)
            return (d_a, d_tgt.view(shape), d_t2.view(shape) if d_t2 is not None else None, gwo, gbo, gwq, gbq, dg, db,
                    None, None, None, None, None, None, None)
        d_t2 = None
        gwq = gbq = None
        if d_qout is not None:
            dq2 = _seq_rows(d_qout.contiguous(), B)
            d_t2 = torch.mm(dq2, wq)                 # gradient of t2 + pos
            gwq, gbq = _park_or_grad(wq, bq, dq2, xq.reshape(-1, C), need[5], need[6])
        if d_y is None and d_t2 is None:
            return (None,) * 16
        d_x, d_r, dg, db, _, _ = ALN.backward_core((rows, C, eps, p, salt, True), y, g2, None, mean, rstd, rng, ctx.ln_params,
                                                   d_y.contiguous().view(rows, C) if d_y is not None else None, d_t2, None)
        d_x = d_x.view(rows, C)
        d_r2 = d_r.view(rows, C) if d_r is not None else d_x
        a2 = _seq_rows(a, B)
        d_a = _batch_first(torch.mm(d_r2, wo), B) if need[0] else None
        gwo, gbo = _park_or_grad(wo, bo, d_r2, a2, need[3], need[4])
        d_pos = d_t2.view(shape) if (need[2] and d_t2 is not None) else None
        return d_a, d_x.view(shape), d_pos, gwo, gbo, gwq, gbq, dg, db, None, None, None, None, None, None, None


class _Ffn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, tgt, wp, bp, w1, b1, w2, b2, g3, be3, gp1, bep1, gp2, bep2, eps3, epsp, p2, salt2, pa, salta, p3, salt3,
                rng, B, wpt, w1t, w2t):
        _check(a, tgt, wp, bp, w1, b1, w2, b2, g3, be3, gp1, bep1, gp2, bep2, wpt, w1t, w2t)
        ctx.kv_rec = A.kv_record_of(a)  # (the cross-attention call that produced a)
        rows = tgt.numel() // C
        dev = tgt.device
        new = lambda: torch.empty_like(tgt)
        y, t2, h, z, o1 = new(), new(), new(), new(), new()
        o2 = new() if gp2 is not None else None
        stats = torch.empty((4, rows), dtype=torch.float32, device=dev)
        use_rng = rng is not None and (p2 > 0 or pa > 0 or p3 > 0)
        d = L.RbFfnDesc()
        d.rows, d.B = rows, B
        d.rng_state = rng.data_ptr() if use_rng else None
        d.a, d.tgt = a.data_ptr(), tgt.data_ptr()
        _lin(d.proj, wp, bp, wpt)
        _lin(d.lin1, w1, b1, w1t)
        _lin(d.lin2, w2, b2, w2t)
        _drop(d.drop2, p2, salt2)
        _drop(d.drop_act, pa, salta)
        _drop(d.drop3, p3, salt3)
        _norm(d.norm3, g3, be3, eps3)
        _norm(d.post1, gp1, bep1, epsp)
        if gp2 is not None:
            _norm(d.post2, gp2, bep2, epsp)
        d.y, d.mean_y, d.rstd_y, d.t2 = y.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), t2.data_ptr()
        d.h = h.data_ptr()
        d.z, d.mean_z, d.rstd_z, d.o1 = z.data_ptr(), stats[2].data_ptr(), stats[3].data_ptr(), o1.data_ptr()
        d.o2 = o2.data_ptr() if o2 is not None else None
        pend = A.take_pending_parts(a)
        if pend is not None:
            # `a` is the output of an attention forward that left the merge of its key-split partials to this launch
            # (attention.fused_attention(defer_combine=True)): rb_ffn merges them on its way in and writes `a` and its lse
            parts, _ws, lse, _out = pend
            L.check(L.lib().vdetr_rb_ffn_parts_f32(ctypes.byref(d), ctypes.byref(parts), a.data_ptr(), lse.data_ptr(), L.stream_ptr()),
                    "rb_ffn_parts")
        else:
            L.check(L.lib().vdetr_rb_ffn_f32(ctypes.byref(d), L.stream_ptr()), "rb_ffn")
        ctx.cfg = (rows, float(eps3), float(epsp), float(p2), salt2, float(pa), float(p3), salt3, B, tgt.shape)
        ctx.ln3 = (g3, be3, None, None)
        ctx.lnp = (gp1, bep1, gp2, bep2)
        ctx.save_for_backward(a, y, t2, h, z, stats, wp, bp, w1, b1, w2, b2, g3, gp1, gp2, rng if use_rng else None)
        ctx.set_materialize_grads(False)
        return z, o1, o2

    @staticmethod
    def backward(ctx, d_z, d_o1, d_o2):
        rows, eps3, epsp, p2, salt2, pa, p3, salt3, B, shape = ctx.cfg
        a, y, t2, h, z, stats, wp, bp, w1, b1, w2, b2, g3, gp1, gp2, rng = ctx.saved_tensors
        need = ctx.needs_input_grad
        if d_z is None and d_o1 is None and d_o2 is None:
            return (None,) * 27
        if FUSED_BWD:
            dev = y.device
            d_z, d_o1, d_o2 = _opt(d_z), _opt(d_o1), _opt(d_o2)
            two = d_o2 is not None
            nblk = (rows + 15) // 16
            new = lambda: torch.empty((rows, C), dtype=torch.float32, device=dev)
            d_tgt, d_lin2, d_lin1, d_proj = new(), new(), new(), new()
            d_a = torch.empty((B, rows // B, C), dtype=torch.float32, device=dev) if need[0] else None
            parts = torch.empty((2, nblk, 4, C), dtype=torch.float32, device=dev)
            d = L.RbFfnDesc()
            d.rows, d.B = rows, B
            d.rng_state = rng.data_ptr() if rng is not None else None
            _lin(d.proj, wp, None)
            _lin(d.lin1, w1, None)
            _lin(d.lin2, w2, None)
            _drop(d.drop2, p2, salt2)
            _drop(d.drop_act, pa, 0)
            _drop(d.drop3, p3, salt3)
            _norm(d.norm3, g3, g3, eps3)
            _norm(d.post1, gp1, gp1, epsp)
            if gp2 is not None:
                _norm(d.post2, gp2, gp2, epsp)
            d.y, d.mean_y, d.rstd_y = y.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr()
            d.h, d.z, d.mean_z, d.rstd_z = h.data_ptr(), z.data_ptr(), stats[2].data_ptr(), stats[3].data_ptr()
            g = L.RbFfnGrads()
            g.d_z = d_z.data_ptr() if d_z is not None else None
            g.d_o1 = d_o1.data_ptr() if d_o1 is not None else None
            g.d_o2 = d_o2.data_ptr() if d_o2 is not None else None
            g.d_tgt, g.d_lin2, g.d_lin1, g.d_proj = d_tgt.data_ptr(), d_lin2.data_ptr(), d_lin1.data_ptr(), d_proj.data_ptr()
            g.d_a = d_a.data_ptr() if d_a is not None else None
            g.part_post, g.part_n3 = parts[0].data_ptr(), parts[1].data_ptr()
            emit = _emit_for(getattr(ctx, "kv_rec", None), a, d_a, rows, B, False)
            if emit is not None:
                L.check(L.lib().vdetr_rb_ffn_bwd_emit_f32(ctypes.byref(d), ctypes.byref(g), ctypes.byref(emit), L.stream_ptr()), "rb_ffn_bwd_emit")
                ctx.kv_rec.emitted = d_a.data_ptr()
            else:
                L.check(L.lib().vdetr_rb_ffn_bwd_f32(ctypes.byref(d), ctypes.byref(g), L.stream_ptr()), "rb_ffn_bwd")
            gw2, gb2 = _park_or_grad(w2, b2, d_lin2, h.view(rows, C), need[6], need[7])
            gw1, gb1 = _park_or_grad(w1, b1, d_lin1, t2.view(rows, C), need[4], need[5])
            gwp, gbp = _park_or_grad(wp, bp, d_proj, _seq_rows(a, B), need[2], need[3])
            dgp1, dbp1, dgp2, dbp2 = _ln_sums(parts[0], nblk, ctx.lnp, two)
            dg3, db3, _, _ = _ln_sums(parts[1], nblk, ctx.ln3, False)
            return (d_a, d_tgt.view(shape), gwp, gbp, gw1, gb1, gw2, gb2, dg3, db3, dgp1, dbp1, dgp2, dbp2) + (None,) * 13
        cont = lambda t: t.contiguous().view(rows, C) if t is not None else None
        # block 3: z = y + drop3(lin2 h); o1 = post1(z), o2 = post2(z)
        d_y, d_r3, dgp1, dbp1, dgp2, dbp2 = ALN.backward_core((rows, C, epsp, p3, salt3, True), z, gp1, gp2, stats[2], stats[3], rng,
                                                              ctx.lnp, cont(d_z), cont(d_o1), cont(d_o2))
        d_y = d_y.view(rows, C)
        d_r3 = d_r3.view(rows, C) if d_r3 is not None else d_y
        d_h = torch.mm(d_r3, w2)
        gw2, gb2 = _park_or_grad(w2, b2, d_r3, h.view(rows, C), need[6], need[7])
        d_pre = torch.empty_like(d_h)
        L.check(L.lib().vdetr_relu_dropout_bwd_f32(L.ptr(h), L.ptr(d_h), L.ptr(d_pre), h.numel(), float(pa), L.stream_ptr()),
                "relu_dropout_bwd")
        d_t2 = torch.mm(d_pre, w1)
        gw1, gb1 = _park_or_grad(w1, b1, d_pre, t2.view(rows, C), need[4], need[5])
        # block 2: y = tgt + drop2(proj a); t2 = norm3(y)
        d_tgt, d_r2, dg3, db3, _, _ = ALN.backward_core((rows, C, eps3, p2, salt2, True), y, g3, None, stats[0], stats[1], rng, ctx.ln3,
                                                        d_y, d_t2, None)
        d_tgt = d_tgt.view(rows, C)
        d_r2 = d_r2.view(rows, C) if d_r2 is not None else d_tgt
        d_a = _batch_first(torch.mm(d_r2, wp), B) if need[0] else None
        gwp, gbp = _park_or_grad(wp, bp, d_r2, _seq_rows(a, B), need[2], need[3])
        return (d_a, d_tgt.view(shape), gwp, gbp, gw1, gb1, gw2, gb2, dg3, db3, dgp1, dbp1, dgp2, dbp2) + (None,) * 13


class _Ffn0(torch.autograd.Function):
    """FFNLayer.forward_pre (the light layer in front of the decoder, reference :585-606) as one launch forward and one backward
    (vdetr_rb_ffn0_f32): t2 = norm(x); z = t2 + drop(lin2(drop(relu(lin1 t2)))); o1 = post_norm(z).  Same dropout streams as the
    composition it replaces (bn_act.relu_dropout / add_ln.add_dropout_layer_norm with the same salts)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, g, be, gp, bep, eps, epsp, pa, salta, p3, salt3, rng, w1t, w2t):
        _check(x, w1, b1, w2, b2, g, be, gp, bep, w1t, w2t)
        rows = x.numel() // C
        dev = x.device
        new = lambda: torch.empty_like(x)
        t2, h, z, o1 = new(), new(), new(), new()
        stats = torch.empty((4, rows), dtype=torch.float32, device=dev)
        use_rng = rng is not None and (pa > 0 or p3 > 0)
        d = L.RbFfnDesc()
        d.rows, d.B = rows, x.shape[1] if x.dim() == 3 else 1
        d.rng_state = rng.data_ptr() if use_rng else None
        d.tgt = x.data_ptr()
        _lin(d.lin1, w1, b1, w1t)
        _lin(d.lin2, w2, b2, w2t)
        _drop(d.drop_act, pa, salta)
        _drop(d.drop3, p3, salt3)
        _norm(d.norm3, g, be, eps)
        _norm(d.post1, gp, bep, epsp)
        d.mean_y, d.rstd_y, d.t2 = stats[0].data_ptr(), stats[1].data_ptr(), t2.data_ptr()
        d.h = h.data_ptr()
        d.z, d.mean_z, d.rstd_z, d.o1 = z.data_ptr(), stats[2].data_ptr(), stats[3].data_ptr(), o1.data_ptr()
        L.check(L.lib().vdetr_rb_ffn0_f32(ctypes.byref(d), L.stream_ptr()), "rb_ffn0")
        ctx.cfg = (rows, float(eps), float(epsp), float(pa), float(p3), salt3, d.B, x.shape)
        ctx.ln = (g, be, None, None)
        ctx.lnp = (gp, bep, None, None)
        ctx.save_for_backward(x, t2, h, z, stats, w1, b1, w2, b2, g, gp, rng if use_rng else None)
        ctx.set_materialize_grads(False)
        return z, o1

    @staticmethod
    def backward(ctx, d_z, d_o1):
        rows, eps, epsp, pa, p3, salt3, B, shape = ctx.cfg
        x, t2, h, z, stats, w1, b1, w2, b2, g, gp, rng = ctx.saved_tensors
        need = ctx.needs_input_grad
        if d_z is None and d_o1 is None:
            return (None,) * 18
        dev = x.device
        d_z, d_o1 = _opt(d_z), _opt(d_o1)
        nblk = (rows + 15) // 16
        new = lambda: torch.empty((rows, C), dtype=torch.float32, device=dev)
        d_x, d_lin2, d_lin1 = new(), new(), new()
        parts = torch.empty((2, nblk, 4, C), dtype=torch.float32, device=dev)
        d = L.RbFfnDesc()
        d.rows, d.B = rows, B
        d.rng_state = rng.data_ptr() if rng is not None else None
        _lin(d.lin1, w1, None)
        _lin(d.lin2, w2, None)
        _drop(d.drop_act, pa, 0)
        _drop(d.drop3, p3, salt3)
        _norm(d.norm3, g, g, eps)
        _norm(d.post1, gp, gp, epsp)
        d.y, d.mean_y, d.rstd_y = x.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr()
        d.h, d.z, d.mean_z, d.rstd_z = h.data_ptr(), z.data_ptr(), stats[2].data_ptr(), stats[3].data_ptr()
        gr = L.RbFfnGrads()
        gr.d_z = d_z.data_ptr() if d_z is not None else None
        gr.d_o1 = d_o1.data_ptr() if d_o1 is not None else None
        gr.d_tgt, gr.d_lin2, gr.d_lin1 = d_x.data_ptr(), d_lin2.data_ptr(), d_lin1.data_ptr()
        gr.part_post, gr.part_n3 = parts[0].data_ptr(), parts[1].data_ptr()
        L.check(L.lib().vdetr_rb_ffn0_bwd_f32(ctypes.byref(d), ctypes.byref(gr), L.stream_ptr()), "rb_ffn0_bwd")
        gw2, gb2 = _park_or_grad(w2, b2, d_lin2, h.view(rows, C), need[3], need[4])
        gw1, gb1 = _park_or_grad(w1, b1, d_lin1, t2.view(rows, C), need[1], need[2])
        dgp, dbp, _, _ = _ln_sums(parts[0], nblk, ctx.lnp, False)
        dg, db, _, _ = _ln_sums(parts[1], nblk, ctx.ln, False)
        return (d_x.view(shape) if need[0] else None, gw1, gb1, gw2, gb2, dg, db, dgp, dbp) + (None,) * 9


# ---- module-level entry points -----------------------------------------------------------------------------------------
def ffn0_usable(layer, x):
    """FFNLayer `layer` on x [n, B, 256] through vdetr_rb_ffn0_f32: fp32 on the GPU, 256 -> 256 -> 256 with biases, ReLU, plain LayerNorms
    (the layer's own and the one the caller applies to its output), gradients on"""
    return bool(FUSED_FFN0 and x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.shape[-1] == C and layer.normalize_before
                and layer.linear1.weight.shape == (C, C) and layer.linear2.weight.shape == (C, C) and layer.linear1.bias is not None
                and layer.linear2.bias is not None and type(layer.activation) is torch.nn.ReLU and _plain_ln(layer.norm)
                and layer.post_norm is not None and _plain_ln(layer.post_norm) and torch.is_grad_enabled())


def ffn0(layer, x, act_salt, aln_salt):
    """(z, post_norm(z)) of FFNLayer.forward_pre"""
    pa = layer.dropout.p if layer.dropout.training else 0.0
    w1, w2 = layer.linear1.weight, layer.linear2.weight
    base = HD._images([w1, w2], HD._fresh["on"])  # ([2, 256, 256] freshly written, or the first of two adjacent images of this forward's refresh)
    imgs = (base[0], base[1]) if base.dim() == 3 else (base, HD._current[w2.data_ptr()])
    n, pn = layer.norm, layer.post_norm
    return _Ffn0.apply(x.contiguous(), layer.linear1.weight, layer.linear1.bias, layer.linear2.weight, layer.linear2.bias, n.weight, n.bias,
                       pn.weight, pn.bias, n.eps, pn.eps, pa, act_salt, pa, aln_salt, _rng_for(pa, x.device), imgs[0], imgs[1])



def _plain_ln(m):
    return ALN.supported(m) and m.normalized_shape[0] == C


def usable(layer, tgt, query_pos, masks):
    """True where GlobalDecoderLayer.forward_pre may take the three fused launches: fp32 activations on the GPU, d_model =
    dim_feedforward = 256, nn.MultiheadAttention-style self attention, plain LayerNorms, a ReLU FFN, no masks."""
    from .vdetr_transformer import GlobalShareCrossAttention, MultiheadSelfAttention
    sa, ca = layer.self_attn, layer.multihead_attn
    return bool(tgt.is_cuda and tgt.dtype == torch.float32 and tgt.dim() == 3 and tgt.shape[-1] == C and all(m is None for m in masks)
                and type(sa) is MultiheadSelfAttention and sa.embed_dim == C and type(ca) is GlobalShareCrossAttention
                and ca.q.weight.shape == (C, C) and ca.proj.weight.shape == (C, C) and ca.q.bias is not None
                and ca.proj.bias is not None and sa.in_proj_bias is not None and sa.out_proj.bias is not None
                and layer.linear1.weight.shape == (C, C) and layer.linear2.weight.shape == (C, C)
                and layer.linear1.bias is not None and layer.linear2.bias is not None  # (a bias-free layer: the unfused path)
                and type(layer.activation) is torch.nn.ReLU and all(_plain_ln(m) for m in (layer.norm1, layer.norm2, layer.norm3))
                and (query_pos is None or query_pos.shape == tgt.shape))


def qkv(t, pos, sa, B, img):
    """the self-attention's projected operands, batch-first [B, nQ, 256] each, and `pos` again — hand THAT tensor to proj_q so that
    the position's gradient is summed inside qkv's backward launch; img: the layer's images (images())"""
    E = sa.embed_dim
    wq, wk, wv = sa.in_proj_weight.view(3, E, E).unbind(0)
    bq, bk, bv = sa.in_proj_bias.view(3, E).unbind(0)
    q, k, v, pos_alias = _Qkv.apply(t.contiguous(), pos.contiguous() if pos is not None else None, wq, wk, wv, bq, bk, bv, B, img[0:3])
    return q, k, v, pos_alias


def proj_q(a, tgt, pos, out_proj, q_lin, drop, ln, salt, B, img):
    """(tgt + drop(out_proj(a)), q_lin(ln(.) + pos)): the residual stream [nQ, B, 256] and the cross attention's query [B, nQ, 256]"""
    p = drop.p if (drop is not None and drop.training) else 0.0
    return _ProjQ.apply(a.contiguous(), tgt.contiguous(), pos.contiguous() if pos is not None else None, out_proj.weight, out_proj.bias,
                        q_lin.weight, q_lin.bias, ln.weight, ln.bias, ln.eps, p, salt, _rng_for(p, tgt.device), B, img[3], img[4])


def ffn(a, tgt, layer, post_norms, salts, act_salt, B, img):
    """residual blocks 2 and 3 of the layer around its FFN; returns (z, post_norms[0](z) [, post_norms[1](z)])"""
    ca = layer.multihead_attn
    p2 = layer.dropout2.p if layer.dropout2.training else 0.0
    if ca.proj_drop.training and ca.proj_drop.p > 0.0:   # proj_drop and dropout2: one mask (add_ln.add_dropout_layer_norm)
        p2 = 1.0 - (1.0 - p2) * (1.0 - ca.proj_drop.p)
    pa = layer.dropout.p if layer.dropout.training else 0.0
    p3 = layer.dropout3.p if layer.dropout3.training else 0.0
    n1 = post_norms[0]
    n2 = post_norms[1] if len(post_norms) > 1 else None
    rng = _rng_for(max(p2, pa, p3), tgt.device)
    return _Ffn.apply(a.contiguous(), tgt.contiguous(), ca.proj.weight, ca.proj.bias, layer.linear1.weight, layer.linear1.bias,
                      layer.linear2.weight, layer.linear2.bias, layer.norm3.weight, layer.norm3.bias, n1.weight, n1.bias,
                      n2.weight if n2 is not None else None, n2.bias if n2 is not None else None, layer.norm3.eps, n1.eps,
                      p2, salts[1], pa, act_salt, p3, salts[2], rng, B, img[5], img[6], img[7])
