"""Residual add + dropout + LayerNorm as one HIP launch (and one for its backward).

Host side of ``vdetr_add_ln_{fwd,bwd}_f32``: the pre-norm residual blocks of the decoder layer
(reference models/vdetr_transformer.py:531-568, ``tgt = tgt + self.dropoutN(tgt2); tgt2 = self.norm(tgt)``) and the
decoder's ``self.norm(output)`` next to the following layer's ``norm1(output)`` (:401, :433).  The dropout mask comes
from the same counter-based generator as the attention kernels (attention.begin_step / current_rng), keyed by a per-call
salt; backward regenerates it.  No CPU path: CPU tensors raise.
"""
import ctypes
import itertools

import torch

from . import _lib as L
from . import attention as A
from .helpers import DeferredParamGrads

_salts = itertools.count(0x5EED0001)


def supported(ln, *more):
    """True when the modules are plain affine LayerNorms over a last dimension the kernel handles."""
    for m in (ln,) + more:
        if m is None:
            continue
        if type(m) is not torch.nn.LayerNorm or not m.elementwise_affine or m.bias is None:
            return False
        if len(m.normalized_shape) != 1 or m.normalized_shape[0] % 256 or m.normalized_shape[0] > 1024:
            return False
    return more == () or all(m is None or (m.normalized_shape == ln.normalized_shape and m.eps == ln.eps) for m in more)


def new_salt():
    return next(_salts)


def _desc(rows, C, eps, p, salt, rng, tensors):
    d = L.AddLnDesc()
    d.rows, d.C, d.eps, d.dropout_p = rows, C, float(eps), float(p)
    d.seed = int(salt) & 0xFFFFFFFFFFFFFFFF
    d.rng_state = rng.data_ptr() if rng is not None else None
    for k in ("x", "r", "gamma", "beta", "gamma2", "beta2", "y", "out", "out2", "mean", "rstd"):
        t = tensors.get(k)
        setattr(d, k, t.data_ptr() if t is not None else None)
    return d


class DeferredLnGrads:
    """dgamma / dbeta of the LayerNorms, reduced AFTER the backward: with runtime.defer_weight_grads() the backward kernel only
    leaves its per-workgroup partial sums; `flush()` (runtime.flush_weight_grads) reduces all parked passes in ONE launch
    (27 launches of ~7 us per step otherwise) and hands the sums to the parameters as DeferredParamGrads does."""
    pending = []

    @classmethod
    def flush(cls, collect=None, keepalive=None):
        """``collect`` / ``keepalive``: as DeferredParamGrads.flush (another stream's launches; the caller delivers after the join)"""
        items, cls.pending = cls.pending, []
        if not items:
            return
        if keepalive is not None:
            keepalive.extend(items)
        # passes that share their first LayerNorm (the decoder's output norm closes all 9 stages) next to each other: their
        # sums are then added by ONE reduction over a slice of `out` instead of one accumulation launch per pass and parameter
        items.sort(key=lambda it: id(it[3][0]))
        n = len(items)
        dev = items[0][0].device
        cmax = max(it[2] for it in items)
        out = torch.empty((4, n, cmax), dtype=torch.float32, device=dev)
        descs = (L.AddLnReduce * n)()
        for i, (ws, nparts, C, _, two) in enumerate(items):
            d = descs[i]
            d.partials, d.nparts, d.C = ws.data_ptr(), nparts, C
            d.d_gamma, d.d_beta = out[0, i].data_ptr(), out[1, i].data_ptr()
            d.d_gamma2, d.d_beta2 = (out[2, i].data_ptr(), out[3, i].data_ptr()) if two else (None, None)
        L.check(L.lib().vdetr_add_ln_param_reduce_batch_f32(descs, n, L.stream_ptr()), "add_ln_param_reduce_batch")
        roots, grads = [], []
        with torch.no_grad():
            i = 0
            while i < n:
                C, params = items[i][2], items[i][3]
                j = i + 1
                while j < n and items[j][3][0] is params[0] and items[j][3][1] is params[1] and items[j][2] == C:
                    j += 1
                first = out[0:2, i:j].sum(1) if j - i > 1 else out[0:2, i]  # [2, cmax]: d_gamma, d_beta of the shared norm

                def give(p, g):
                    if collect is not None:
                        collect.append((p, g))
                    else:
                        DeferredParamGrads._deliver(p, g, roots, grads)
                for k in range(2):
                    if params[k] is not None and params[k].requires_grad:
                        give(params[k], first[k, :C])
                for m in range(i, j):
                    if items[m][4]:
                        for k in (2, 3):
                            p = items[m][3][k]
                            if p is not None and p.requires_grad:
                                give(p, out[k, m, :C])
                i = j
        if roots:
            torch.autograd.backward(roots, grads)


class _AddLN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, r, gamma, beta, gamma2, beta2, eps, p, rng, salt):
        for name, t in (("x", x), ("r", r), ("gamma", gamma), ("beta", beta), ("gamma2", gamma2), ("beta2", beta2)):
            if t is not None:
                L.require_gpu(t, name)
                L.require_float(t, name)
        x = x.contiguous()
        r = r.contiguous() if r is not None else None
        C = x.shape[-1]
        rows = x.numel() // C
        y = torch.empty_like(x) if r is not None else None
        out = torch.empty_like(x)
        out2 = torch.empty_like(x) if gamma2 is not None else None
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        use_drop = r is not None and p > 0.0
        d = _desc(rows, C, eps, p if use_drop else 0.0, salt, rng if use_drop else None,
                  dict(x=x, r=r, gamma=gamma.contiguous(), beta=beta.contiguous(),
                       gamma2=gamma2.contiguous() if gamma2 is not None else None,
                       beta2=beta2.contiguous() if beta2 is not None else None, y=y, out=out, out2=out2, mean=mean, rstd=rstd))
        L.check(L.lib().vdetr_add_ln_fwd_f32(ctypes.byref(d), L.stream_ptr()), "add_ln_fwd")
        ctx.cfg = (rows, C, eps, p if use_drop else 0.0, salt, r is not None)
        ctx.ln_params = (gamma, beta, gamma2, beta2)  # the tensors themselves: the deferred sums are delivered to them
        ctx.save_for_backward(y if r is not None else x, gamma, gamma2, mean, rstd, rng if use_drop else None)
        ctx.set_materialize_grads(False)
        return y, out, out2

    @staticmethod
    def backward(ctx, d_y, d_out, d_out2):
        ysrc, gamma, gamma2, mean, rstd, rng = ctx.saved_tensors
        if d_y is None and d_out is None and d_out2 is None:
            return (None,) * 10
        cont = lambda t: t.contiguous() if t is not None else None
        has_r = ctx.cfg[5]
        d_x, d_r, d_gamma, d_beta, d_gamma2, d_beta2 = backward_core(ctx.cfg, ysrc, gamma, gamma2, mean, rstd, rng, ctx.ln_params,
                                                                      cont(d_y), cont(d_out), cont(d_out2))
        # without dropout the branch gradient IS the residual gradient
        return d_x, (d_r if d_r is not None else d_x) if has_r else None, d_gamma, d_beta, d_gamma2, d_beta2, None, None, None, None


def backward_core(cfg, ysrc, gamma, gamma2, mean, rstd, rng, ln_params, d_y, d_out, d_out2):
    """One vdetr_add_ln_bwd_f32 launch on the tensors a forward launch left (this module's, or a fused launch that writes the
    same ones: rowblock.py).  cfg = (rows, C, eps, p, salt, has_r); gradients contiguous or None.  Returns (d_x, d_r, d_gamma,
    d_beta, d_gamma2, d_beta2): d_r is None without dropout (the branch gradient is then d_x); the parameter sums are None when
    they are parked for the flush (DeferredLnGrads, `ln_params` = the four parameter tensors they are delivered to)."""
    rows, C, eps, p, salt, has_r = cfg
    # the backward kernel only tests `r` for NULL; `y` / `x` carry the normalised tensor
    d = _desc(rows, C, eps, p, salt, rng, dict(x=ysrc, r=ysrc if has_r else None, gamma=gamma, beta=gamma, gamma2=gamma2,
                                                beta2=gamma2, y=ysrc if has_r else None, mean=mean, rstd=rstd))
    g = L.AddLnGrads()
    d_x = torch.empty_like(ysrc)
    d_r = torch.empty_like(ysrc) if (has_r and p > 0.0) else None
    nbytes = L.lib().vdetr_add_ln_bwd_workspace_bytes(ctypes.byref(d))
    # parameter sums after the backward, all LayerNorms in one launch (see DeferredLnGrads): the kernel only leaves
    # its per-workgroup partial sums, in a buffer of their own
    defer = DeferredParamGrads.enabled and DeferredParamGrads.direct and (d_out2 is not None or gamma2 is None)
    if defer:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=ysrc.device)
        d_gamma = d_beta = d_gamma2 = d_beta2 = None
        DeferredLnGrads.pending.append((ws, nbytes // (16 * C), C, ln_params, d_out2 is not None))
    else:
        d_gamma, d_beta = torch.empty_like(gamma), torch.empty_like(gamma)
        d_gamma2 = torch.empty_like(gamma2) if d_out2 is not None else None
        d_beta2 = torch.empty_like(gamma2) if d_out2 is not None else None
        ws = L.workspace(nbytes, ysrc.device)
    for k, t in (("d_out", d_out), ("d_out2", d_out2), ("d_y", d_y), ("d_x", d_x), ("d_r", d_r), ("d_gamma", d_gamma),
                 ("d_beta", d_beta), ("d_gamma2", d_gamma2), ("d_beta2", d_beta2), ("partials", ws)):
        setattr(g, k, t.data_ptr() if t is not None else None)
    L.check(L.lib().vdetr_add_ln_bwd_f32(ctypes.byref(d), ctypes.byref(g), L.stream_ptr()), "add_ln_bwd")
    if not defer and d_out2 is None and gamma2 is not None:
        d_gamma2, d_beta2 = torch.zeros_like(gamma2), torch.zeros_like(gamma2)
    return d_x, d_r, d_gamma, d_beta, d_gamma2, d_beta2


def layer_norm(x, ln, ln2=None):
    """``ln(x)`` (and ``ln2(x)``: a second affine map of the same statistics).  Returns out or (out, out2)."""
    _, out, out2 = _AddLN.apply(x, None, ln.weight, ln.bias, ln2.weight if ln2 is not None else None,
                                ln2.bias if ln2 is not None else None, ln.eps, 0.0, None, 0)
    return out if ln2 is None else (out, out2)


def add_dropout_layer_norm(x, r, drop, ln, ln2=None, salt=0, also_drop=None):
    """y = x + drop(r); returns (y, ln(y)) or (y, ln(y), ln2(y)).  ``drop`` is the nn.Dropout module of the block.
    ``also_drop``: a second nn.Dropout the branch output r still has to pass (e.g. the attention module's proj_drop):
    two independent Bernoulli masks are one mask with keep probability (1-p1)(1-p2) and the product of the scales."""
    p = drop.p if (drop is not None and drop.training) else 0.0
    if also_drop is not None and also_drop.training and also_drop.p > 0.0:
        p = 1.0 - (1.0 - p) * (1.0 - also_drop.p)
    rng = None
    if p > 0.0:
        rng = A.current_rng(x.device)
        if rng is None:
            rng = A.begin_step(x.device)
    y, out, out2 = _AddLN.apply(x, r, ln.weight, ln.bias, ln2.weight if ln2 is not None else None,
                                ln2.bias if ln2 is not None else None, ln.eps, p, rng, salt)
    return (y, out) if ln2 is None else (y, out, out2)
