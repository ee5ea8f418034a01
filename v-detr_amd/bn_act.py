"""BatchNorm1d + ReLU + Dropout on [B, C, N] as one HIP launch (and one for its backward).

Host side of ``vdetr_bn_act_{fwd,bwd}_f32``: the hidden blocks of the reference's GenericMLP (models/helpers.py:74-141,
``Conv1d -> BatchNorm1d -> ReLU -> Dropout``).  Running statistics are updated in place by the kernel (training mode), the
dropout mask comes from the attention kernels' counter-based generator.  No CPU path: CPU tensors raise.
"""
import ctypes
import itertools

import torch

from . import _lib as L
from . import attention as A

_salts = itertools.count(0xB0A70001)

# ---- cross-replica statistics (SyncBatchNorm: the reference converts every BatchNorm before wrapping the model in DDP,
# main.py:512-514; at one scene per GPU a per-rank BatchNorm would see 1/8 of the reference's batch) -----------------------
_sync = {"on": False, "group": None, "force": False}


def set_sync(enabled, group=None, force=False):
    """Batch statistics over ALL data-parallel ranks for every training-mode call of this module (the fused launches stay:
    a statistics launch + one all-gather of [3, C] per call forward, one all-reduce of [2, C] backward).  ``force``: also on
    a 1-rank group (tests)."""
    _sync.update(on=bool(enabled), group=group, force=bool(force))


def sync_active():
    import torch.distributed as dist
    return _sync["on"] and dist.is_available() and dist.is_initialized() and (
        dist.get_world_size(_sync["group"]) > 1 or _sync["force"])


def _global_stats(d, x, eps, momentum, rm, rv, pre_bias, counters):
    """local (mean, M2) -> all ranks' -> merged (Chan et al.): returns (mean, invstd, inv_count) device tensors and updates
    the running statistics / counters the way nn.SyncBatchNorm does (unbiased variance of the GLOBAL batch)."""
    import torch.distributed as dist
    C = x.shape[1]
    loc = torch.empty((3, C), dtype=torch.float32, device=x.device)
    L.check(L.lib().vdetr_bn_stats_f32(ctypes.byref(d), L.ptr(loc[0]), L.ptr(loc[1]), L.stream_ptr()), "bn_stats")
    loc[2].fill_(float(x.shape[0] * x.shape[2]))
    world = dist.get_world_size(_sync["group"])
    allr = torch.empty((world, 3, C), dtype=torch.float32, device=x.device)
    dist.all_gather(list(allr.unbind(0)), loc, group=_sync["group"])  # (the list form: gloo's flat variant wants other shapes)
    cnt = allr[:, 2]                                      # [W, C] element counts (ranks may hold different N)
    n = cnt.sum(0)
    mean = (allr[:, 0] * cnt).sum(0) / n
    m2 = (allr[:, 1] + cnt * (allr[:, 0] - mean) ** 2).sum(0)
    var = m2 / n
    invstd = torch.rsqrt(var + eps)
    if rm is not None:
        rm.mul_(1.0 - momentum).add_((mean + pre_bias) if pre_bias is not None else mean, alpha=momentum)
        rv.mul_(1.0 - momentum).add_(m2 / torch.clamp(n - 1.0, min=1.0), alpha=momentum)
    for c in counters:
        c += 1
    return mean, invstd, (1.0 / n[:1]).contiguous()


class _SyncBatchNormFn(torch.autograd.Function):
    """Cross-replica BatchNorm as tensor expressions, for every call the fused launches do not take (channel counts they are
    not built for, momentum=None, CPU tensors in the gloo tests, modules reached through ``nn.Sequential``): x with the channels
    in dimension 1 ([B, C, N]) or last ([N, C]), any C, ANY local element count including zero — a rank with an empty tensor
    still joins the two collectives (count 0), otherwise the others would wait for it forever.  Statistics as
    ``_global_stats`` (Chan's merge of the ranks' (mean, M2, count)); running statistics as nn.SyncBatchNorm."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, nbt, momentum, eps, channel_last):
        import torch.distributed as dist
        grp = _sync["group"]
        xr = x if channel_last else x.transpose(1, -1)
        C = xr.shape[-1]
        rows = xr.reshape(-1, C)
        n_loc = rows.shape[0]
        sdt = torch.float64 if x.dtype == torch.float64 else torch.float32  # statistics in fp32 (fp64 for fp64 inputs: the tests)
        rows = rows.to(sdt)
        if n_loc:
            var_l, mean_l = torch.var_mean(rows, 0, unbiased=False)
        else:
            var_l = mean_l = torch.zeros(C, dtype=sdt, device=x.device)
        loc = torch.stack((mean_l, var_l * n_loc, torch.full_like(mean_l, float(n_loc))))
        world = dist.get_world_size(grp)
        allr = torch.empty((world, 3, C), dtype=sdt, device=x.device)
        dist.all_gather(list(allr.unbind(0)), loc, group=grp)
        cnt = allr[:, 2]
        n = cnt.sum(0).clamp_(min=1.0)
        mean = (allr[:, 0] * cnt).sum(0) / n
        m2 = (allr[:, 1] + cnt * (allr[:, 0] - mean) ** 2).sum(0)
        var = m2 / n
        invstd = torch.rsqrt(var + eps)
        if running_mean is not None:
            f = momentum if momentum is not None else 1.0 / float(int(nbt) + 1 if nbt is not None else 1)  # None: cumulative average
            running_mean.mul_(1.0 - f).add_(mean.to(running_mean.dtype), alpha=f)
            running_var.mul_(1.0 - f).add_((m2 / torch.clamp(n - 1.0, min=1.0)).to(running_var.dtype), alpha=f)
        if nbt is not None:
            nbt += 1
        xhat = (rows - mean) * invstd
        y = xhat if weight is None else xhat * weight + bias
        ctx.save_for_backward(xhat, weight, invstd, (1.0 / n[:1]))
        ctx.cfg = (channel_last, xr.shape, x.dtype)
        y = y.to(x.dtype).reshape(xr.shape)
        return y if channel_last else y.transpose(1, -1)

    @staticmethod
    def backward(ctx, dy):
        import torch.distributed as dist
        xhat, weight, invstd, inv_n = ctx.saved_tensors
        channel_last, shape, dtype = ctx.cfg
        g = (dy if channel_last else dy.transpose(1, -1)).reshape(-1, shape[-1]).to(xhat.dtype)
        loc = torch.stack((g.sum(0), (g * xhat).sum(0)))
        tot = loc.clone()
        dist.all_reduce(tot, group=_sync["group"])
        dx = None
        if ctx.needs_input_grad[0]:
            k = invstd if weight is None else weight * invstd
            dx = (k * (g - tot[0] * inv_n - xhat * (tot[1] * inv_n))).to(dtype).reshape(shape)
            dx = dx if channel_last else dx.transpose(1, -1)
        # parameter gradients stay LOCAL sums (the gradient all-reduce averages them, as for every other parameter)
        return dx, (loc[1] if weight is not None else None), (loc[0] if weight is not None else None), None, None, None, None, None, None


def sync_batch_norm(x, weight, bias, running_mean, running_var, nbt, momentum, eps, channel_last=False):
    """Training-mode BatchNorm with statistics over all data-parallel ranks, as tensor expressions (see _SyncBatchNormFn)."""
    return _SyncBatchNormFn.apply(x, weight, bias, running_mean, running_var, nbt, momentum, eps, channel_last)


def batch_norm_module(bn, x, channel_last=False):
    """``bn(x)`` for an ``nn.BatchNorm1d`` — with cross-replica statistics while ``set_sync`` is on and the module trains.  The
    one door every BatchNorm call that does not go through a fused launch takes (the reference converts EVERY BatchNorm,
    main.py:512-514: a per-rank fallback would let the running statistics drift apart from step 1 on)."""
    if bn.training and sync_active() and isinstance(bn, torch.nn.modules.batchnorm._BatchNorm):
        if not bn.track_running_stats:
            return sync_batch_norm(x, bn.weight, bn.bias, None, None, None, bn.momentum, bn.eps, channel_last)
        return sync_batch_norm(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.momentum,
                               bn.eps, channel_last)
    return bn(x)


def run_sequential(seq, x):
    """``seq(x)`` for an ``nn.Sequential`` whose BatchNorm members go through ``batch_norm_module``."""
    if not sync_active():
        return seq(x)
    for m in seq:
        x = batch_norm_module(m, x) if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) else m(x)
    return x


def new_salt():
    return next(_salts)


def _desc(x, gamma, beta, rm, rv, y, smean, sinv, training, relu, eps, momentum, p, salt, rng, pre_bias=None, counters=()):
    d = L.BnActDesc()
    d.B, d.C, d.N = x.shape
    d.training, d.relu, d.eps, d.momentum, d.dropout_p = int(training), int(relu), float(eps), float(momentum), float(p)
    d.seed = int(salt) & 0xFFFFFFFFFFFFFFFF
    d.rng_state = rng.data_ptr() if rng is not None else None
    for k, t in (("x", x), ("gamma", gamma), ("beta", beta), ("running_mean", rm), ("running_var", rv), ("y", y),
                 ("save_mean", smean), ("save_invstd", sinv), ("pre_bias", pre_bias)):
        setattr(d, k, t.data_ptr() if t is not None else None)
    assert len(counters) <= 8
    for i, t in enumerate(counters):
        assert t.dtype == torch.int64 and t.is_cuda
        d.counters[i] = t.data_ptr()
    d.ncounters = len(counters)
    return d


class _BnAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, rm, rv, training, relu, eps, momentum, p, rng, salt, pre_bias, counters):
        L.require_gpu(x, "x")
        L.require_float(x, "x")
        x = x.contiguous()
        gamma_c, beta_c = (gamma.contiguous(), beta.contiguous()) if gamma is not None else (None, None)
        y = torch.empty_like(x)
        C = x.shape[1]
        smean = torch.empty(C, dtype=torch.float32, device=x.device) if training else None
        sinv = torch.empty_like(smean) if training else None
        p = p if training else 0.0
        pb = pre_bias.detach().contiguous() if pre_bias is not None else None
        inv_n = None
        if training and sync_active():
            d0 = _desc(x, None, None, None, None, None, None, None, True, relu, eps, momentum, 0.0, 0, None)
            smean, sinv, inv_n = _global_stats(d0, x, eps, momentum, rm, rv, pb, counters)
            d = _desc(x, gamma_c, beta_c, None, None, y, smean, sinv, True, relu, eps, momentum, p, salt, rng if p > 0 else None)
            d.stats_given = 1
        else:
            d = _desc(x, gamma_c, beta_c, rm, rv, y, smean, sinv, training, relu, eps, momentum, p, salt, rng if p > 0 else None,
                      pb, counters)
        L.check(L.lib().vdetr_bn_act_fwd_f32(ctypes.byref(d), L.stream_ptr()), "bn_act_fwd")
        ctx.cfg = (training, relu, eps, momentum, p, salt)
        ctx.inv_n = inv_n
        ctx.save_for_backward(x, gamma_c, beta_c, smean, sinv, rng if p > 0 else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        training, relu, eps, momentum, p, salt = ctx.cfg
        x, gamma, beta, smean, sinv, rng = ctx.saved_tensors
        if not training:
            raise RuntimeError("bn_act: backward through the eval-mode (running statistics) path is not built")
        dy = dy.contiguous()
        d = _desc(x, gamma, beta, None, None, None, smean, sinv, True, relu, eps, momentum, p, salt, rng)
        g = L.BnActGrads()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dg = torch.empty_like(gamma) if gamma is not None and ctx.needs_input_grad[1] else None
        db = torch.empty_like(beta) if beta is not None and ctx.needs_input_grad[2] else None
        if ctx.inv_n is not None:
            dx2, dg2, db2 = _sync_backward(d, dy, dx, ctx.inv_n)
            return dx2, (dg2 if dg is not None else None), (db2 if db is not None else None), None, None, None, None, None, \
                None, None, None, None, None, None
        g.dy, g.dx = dy.data_ptr(), dx.data_ptr() if dx is not None else None
        g.d_gamma, g.d_beta = (dg.data_ptr() if dg is not None else None), (db.data_ptr() if db is not None else None)
        L.check(L.lib().vdetr_bn_act_bwd_f32(ctypes.byref(d), ctypes.byref(g), L.stream_ptr()), "bn_act_bwd")
        return dx, dg, db, None, None, None, None, None, None, None, None, None, None, None


def _sync_backward(d, dy, dx, inv_n):
    """backward with statistics over all ranks: this rank's two sums (a launch without dx), their all-reduce, dx with the
    global sums.  Returns (dx, dgamma, dbeta); the parameter gradients are THIS rank's sums, as nn.SyncBatchNorm returns
    them (the gradient all-reduce averages them like every other parameter gradient)."""
    import torch.distributed as dist
    C = d.C
    loc = torch.empty((2, C), dtype=torch.float32, device=dy.device)
    g = L.BnActGrads()
    g.dy, g.dx, g.d_gamma, g.d_beta = dy.data_ptr(), None, loc[0].data_ptr(), loc[1].data_ptr()
    L.check(L.lib().vdetr_bn_act_bwd_f32(ctypes.byref(d), ctypes.byref(g), L.stream_ptr()), "bn_act_bwd")
    if dx is not None:
        tot = loc.clone()
        dist.all_reduce(tot, group=_sync["group"])
        g2 = L.BnActGrads()
        g2.dy, g2.dx = dy.data_ptr(), dx.data_ptr()
        g2.sum_dy_xhat, g2.sum_dy, g2.inv_count = tot[0].data_ptr(), tot[1].data_ptr(), inv_n.data_ptr()
        L.check(L.lib().vdetr_bn_act_bwd_f32(ctypes.byref(d), ctypes.byref(g2), L.stream_ptr()), "bn_act_bwd")
        tot.record_stream(torch.cuda.current_stream())
    return dx, loc[0], loc[1]


def bn_act(x, weight, bias, running_mean, running_var, training, eps, momentum, relu=True, dropout_p=0.0, salt=0,
           pre_bias=None, counters=()):
    """dropout(relu(batch_norm(x))) for x [B, C, N]; running statistics are updated in place when training.
    ``pre_bias``: bias of the 1x1 convolution that produced x, NOT yet added (exactly cancelled by batch statistics; its
    gradient is zero, the running mean accounts for it).  ``counters``: num_batches_tracked buffers to increment."""
    rng = None
    if training and dropout_p > 0.0:
        rng = A.current_rng(x.device)
        if rng is None:
            rng = A.begin_step(x.device)
    return _BnAct.apply(x, weight, bias, running_mean, running_var, bool(training), bool(relu), float(eps), float(momentum),
                        float(dropout_p), rng, int(salt), pre_bias, tuple(counters))


def forward_record(x, gamma, beta, rm, rv, eps, momentum, p, salt, counters=(), y_out=None, pre_bias=None):
    """Training-mode forward WITHOUT an autograd node: returns (y, record); ``backward_from_record(record, dy)`` gives
    (dx, dgamma, dbeta) later.  For callers that batch the backward of several independent blocks themselves
    (vdetr_transformer._DeferredHeads)."""
    rng = None
    if p > 0.0:
        rng = A.current_rng(x.device)
        if rng is None:
            rng = A.begin_step(x.device)
    with torch.no_grad():
        x = x.detach().contiguous()
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        y = y_out if y_out is not None else torch.empty_like(x)  # (a slice of a caller's stacked buffer)
        assert y.is_contiguous() and y.shape == x.shape
        C = x.shape[1]
        smean = torch.empty(C, dtype=torch.float32, device=x.device)
        sinv = torch.empty_like(smean)
        pb = pre_bias.detach().contiguous() if pre_bias is not None else None
        inv_n = None
        if sync_active():
            d0 = _desc(x, None, None, None, None, None, None, None, True, True, eps, momentum, 0.0, 0, None)
            smean, sinv, inv_n = _global_stats(d0, x, eps, momentum, rm, rv, pb, tuple(counters))
            d = _desc(x, g, b, None, None, y, smean, sinv, True, True, eps, momentum, p, salt, rng if p > 0 else None)
            d.stats_given = 1
        else:
            d = _desc(x, g, b, rm, rv, y, smean, sinv, True, True, eps, momentum, p, salt, rng if p > 0 else None, pb, tuple(counters))
        L.check(L.lib().vdetr_bn_act_fwd_f32(ctypes.byref(d), L.stream_ptr()), "bn_act_fwd")
    return y, (x, g, b, smean, sinv, rng if p > 0 else None, (float(eps), float(momentum), float(p), int(salt)), inv_n)


def backward_from_record(record, dy, dx_out=None):
    x, gamma, beta, smean, sinv, rng, (eps, momentum, p, salt), inv_n = record
    dy = dy.contiguous()
    d = _desc(x, gamma, beta, None, None, None, smean, sinv, True, True, eps, momentum, p, salt, rng)
    g = L.BnActGrads()
    dx = dx_out if dx_out is not None else torch.empty_like(x)
    assert dx.is_contiguous() and dx.shape == x.shape
    if inv_n is not None:
        return _sync_backward(d, dy, dx, inv_n)
    dg, db = torch.empty_like(gamma), torch.empty_like(beta)
    g.dy, g.dx, g.d_gamma, g.d_beta = dy.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr()
    L.check(L.lib().vdetr_bn_act_bwd_f32(ctypes.byref(d), ctypes.byref(g), L.stream_ptr()), "bn_act_bwd")
    return dx, dg, db


class _ReluDropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, rng, salt):
        L.require_gpu(x, "x")
        L.require_float(x, "x")
        x = x.contiguous()
        y = torch.empty_like(x)
        L.check(L.lib().vdetr_relu_dropout_fwd_f32(L.ptr(x), L.ptr(y), x.numel(), float(p), int(salt) & 0xFFFFFFFFFFFFFFFF, 0,
                                                   L.ptr(rng) if (rng is not None and p > 0) else None, L.stream_ptr()),
                "relu_dropout_fwd")
        ctx.p = p
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        L.check(L.lib().vdetr_relu_dropout_bwd_f32(L.ptr(y), L.ptr(dy), L.ptr(dx), y.numel(), float(ctx.p), L.stream_ptr()),
                "relu_dropout_bwd")
        return dx, None, None, None


def relu_dropout(x, drop, salt=0):
    """``drop(relu(x))`` (drop: the nn.Dropout module) as one launch; needs numel % 4 == 0."""
    p = drop.p if (drop is not None and drop.training) else 0.0
    rng = None
    if p > 0.0:
        rng = A.current_rng(x.device)
        if rng is None:
            rng = A.begin_step(x.device)
    return _ReluDropout.apply(x, p, rng, salt)


def backward_from_records(records, dys, dx_outs):
    """backward_from_record for several independent blocks of the same B*N in ONE launch.  Returns [(dx, dgamma, dbeta)]."""
    n = len(records)
    if any(rec[7] is not None for rec in records):  # cross-replica statistics: each block has its own collective
        return [backward_from_record(rec, dy, dx) for rec, dy, dx in zip(records, dys, dx_outs)]
    descs, grads = (L.BnActDesc * n)(), (L.BnActGrads * n)()
    keep, out = [], []
    for i, (rec, dy, dx) in enumerate(zip(records, dys, dx_outs)):
        x, gamma, beta, smean, sinv, rng, (eps, momentum, p, salt), _ = rec
        dy = dy.contiguous()
        assert dx.is_contiguous() and dx.shape == x.shape
        descs[i] = _desc(x, gamma, beta, None, None, None, smean, sinv, True, True, eps, momentum, p, salt, rng)
        dg, db = torch.empty_like(gamma), torch.empty_like(beta)
        g = grads[i]
        g.dy, g.dx, g.d_gamma, g.d_beta = dy.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr()
        keep.append(dy)
        out.append((dx, dg, db))
    L.check(L.lib().vdetr_bn_act_bwd_batch_f32(descs, grads, n, L.stream_ptr()), "bn_act_bwd_batch")
    return out
