"""Small building blocks of the decoder with the reference's state-dict layout (models/helpers.py).

Kept in plain PyTorch on purpose: these are 1x1 convolutions / linears + norms, i.e. library GEMMs; only their
parameter NAMES and numerics matter for checkpoint compatibility:
  GenericMLP.layers.{0,1,4,5,8}.*                (helpers.py:74-141, Conv1d-BN-ReLU-Dropout x2 -> Conv1d)
  PositionEmbeddingLearned.position_embedding_head.{0,1,3}.*   (helpers.py:17-33)
"""
import copy
import os
from functools import partial

import torch
import torch.nn as nn


def colsum_batched(x3d):
    """sums over the rows of n [rows, cols] fp32 CUDA matrices (x3d [n, rows, cols], rows / items strided is fine) -> [n, cols], one
    launch — two for tall matrices with few columns (vdetr_colsum_batched_f32; fixed summation order)."""
    import ctypes
    from . import _lib as L
    assert x3d.dim() == 3 and x3d.stride(2) == 1
    n, rows, cols = x3d.shape
    lib = L.lib()
    need = lib.vdetr_colsum_workspace_bytes(n, rows, cols)
    ws = torch.empty(need, dtype=torch.uint8, device=x3d.device) if need else None  # (tall, few columns: partial sums of row ranges)
    out = torch.empty((n, cols), dtype=x3d.dtype, device=x3d.device)
    L.check(lib.vdetr_colsum_batched_f32(L.ptr(x3d), L.ptr(out), n, rows, cols, ctypes.c_long(x3d.stride(1)), ctypes.c_long(x3d.stride(0)),
                                         L.ptr(ws), need, L.stream_ptr()), "colsum_batched")
    return out


PTR_BATCH = os.environ.get("VDETR_PTR_BATCH", "1") != "0"


def _ptr_batchable(group):
    """items (w, b, dY [rows, out], X [rows, in]) whose operands rocblas_sgemm_batched / vdetr_colsum_ptrs_f32 can read in place"""
    if not PTR_BATCH or len(group) > 256 or len(group) < 8:
        return False  # (few items: the copies are small, and the library's pick for a batch of 2 with K = 4096 measured 90 us slower)
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        return False  # (data-parallel steps flush in phases: other batch counts than the ones measured; see the note on 4096 rows below)
    g0, x0 = group[0][2], group[0][3]
    if g0.shape[0] > 1024:
        # MEASURED: with 4096-row items (BASELINE config 5: four scenes) rocblas_sgemm_batched never returns — the GPU hangs in the
        # eager warm-up already (tools/probes/job_r6_c5_bisect.sh); the shapes of the one-scene configurations (1024 rows, 64 / 40
        # items) run in every test and bench of the round.  Longer contractions keep the stacked operands and torch.bmm.
        return False
    if not (g0.is_cuda and g0.dtype == torch.float32 and x0.dtype == torch.float32 and g0.shape[1] % 4 == 0):
        return False
    for it in group:
        g, x = it[2], it[3]
        if not (g.stride(1) == 1 and x.stride(1) == 1 and g.stride(0) == g0.stride(0) and x.stride(0) == x0.stride(0)
                and g.stride(0) % 4 == 0 and g.data_ptr() % 16 == 0 and x.data_ptr() % 4 == 0):
            return False
    return True


def _wgrad_ptrs(group, need_w, need_b):
    """(dW [n, out, in], dB [n, out]) of a group of parked (dY, X) pairs, read where they are"""
    import ctypes
    from . import _lib as L
    n = len(group)
    g0, x0 = group[0][2], group[0][3]
    rows, out_f, in_f = g0.shape[0], g0.shape[1], x0.shape[1]
    dev = g0.device
    dW = torch.empty((n, out_f, in_f), dtype=torch.float32, device=dev) if need_w else None
    dB = None
    ptrs = [it[3].data_ptr() for it in group] + [it[2].data_ptr() for it in group]
    if need_w:
        ptrs += [dW.data_ptr() + i * out_f * in_f * 4 for i in range(n)]
    tab = L.upload_ptrs(ptrs, dev)
    base = tab.data_ptr()
    if need_w:
        # row-major dW [out, in] = dY^T X  ==  column-major [in x out] = X_cm [in x rows] (dY_cm [out x rows])^T
        L.sgemm_batched_ptrs(L.ROCBLAS_OP_N, L.ROCBLAS_OP_T, in_f, out_f, rows, base, x0.stride(0), base + 8 * n, g0.stride(0),
                             base + 16 * n, in_f, n, dev)
    if need_b:
        lib = L.lib()
        dB = torch.empty((n, out_f), dtype=torch.float32, device=dev)
        need = lib.vdetr_colsum_workspace_bytes(n, rows, out_f)
        ws = torch.empty(need, dtype=torch.uint8, device=dev) if need else None
        L.check(lib.vdetr_colsum_ptrs_f32(ctypes.c_void_p(base + 8 * n), L.ptr(dB), n, rows, out_f, ctypes.c_long(g0.stride(0)), L.ptr(ws), need,
                                          L.stream_ptr()), "colsum_ptrs")
    if dW is not None:
        dW._ptr_table = tab  # (the launches read the table after this returns: it lives as long as the result)
    elif dB is not None:
        dB._ptr_table = tab
    return dW, dB


def colsum(x2d):
    """sum over the rows of a [rows, cols] fp32 CUDA matrix (row-strided is fine)."""
    assert x2d.dim() == 2 and x2d.stride(1) == 1
    return colsum_batched(x2d.unsqueeze(0))[0]


class DeferredParamGrads:
    """Weight / bias gradients of the `linear` layers, computed in batches AFTER the backward pass.

    Only dX is on the backward's critical path; dW = dY^T X and db = column sums of dY are ~130 launches of 5-11 us per
    step, each a [256 x 1024] x [1024 x 256] product that fills a quarter of the GPU.  With this enabled
    (runtime.defer_weight_grads) `_Linear.backward` only stores (dY, X) and returns the input gradient; `flush()` -- called
    by the training loop right after `loss.backward()` -- groups the stored pairs by shape, computes each group as ONE
    batched GEMM and ONE reduction, and hands the results to autograd (`torch.autograd.backward` on the weight / bias
    tensors themselves, so parameter aliases and gradient accumulation behave as always; a leaf parameter whose gradient is
    still unset takes its slice of the batched result as `.grad` directly)."""
    enabled = False
    pending = []
    direct = os.environ.get("VDETR_WG_DIRECT", "1") != "0"  # A/B switch (read once)

    @staticmethod
    def _view_of_leaf(p):
        """the leaf parameter `p` is a plain view of (e.g. one of in_proj_weight.view(3, E, E).unbind(0)), or None"""
        if p is None or p.is_leaf or not p._is_view():
            return None
        base = p._base
        return base if base is not None and base.is_leaf and p.is_contiguous() and base.is_contiguous() else None

    @classmethod
    def _deliver(cls, p, g, roots, grads):
        """gradient g of the tensor p a `linear` call used as weight / bias: straight into `.grad` where no accumulation or
        view bookkeeping is needed, through autograd otherwise"""
        if cls.direct and p.is_leaf and p.grad is None:
            # a parameter without a gradient yet simply takes its slice of the batched result:
            # AccumulateGrad would clone every such slice (one copy launch per parameter, ~80 per step)
            p.grad = g
            return
        fn = p.grad_fn
        if cls.direct and type(fn).__name__ == "_AliasBackward":
            # an alias of adjacent parameters (cat_params / stack_params / slot_stack_params): hand each parameter its
            # slice, as _Alias.backward would, without one AccumulateGrad launch per parameter
            parts = _Alias.backward(fn, g)[2:]
            # next_functions holds one edge per TENSOR input of the Function (the two leading non-tensor arguments have none)
            leaves = [nf[0].variable if nf[0] is not None and hasattr(nf[0], "variable") else None for nf in fn.next_functions]
            if len(leaves) == len(parts) and all(l is not None for l in leaves):
                for leaf, part in zip(leaves, parts):
                    cls._deliver(leaf, part, roots, grads)
                return
        roots.append(p)
        grads.append(g)

    @classmethod
    def leaves_of(cls, p):
        """the leaf parameters the gradient of `p` (a weight / bias a `linear` call used) ends up in, or None if that takes
        an autograd walk to find out"""
        if p is None:
            return []
        if p.is_leaf:
            return [p]
        base = cls._view_of_leaf(p)
        if base is not None:
            return [base]
        fn = p.grad_fn
        if type(fn).__name__ == "_AliasBackward":
            leaves = [nf[0].variable if nf[0] is not None and hasattr(nf[0], "variable") else None for nf in fn.next_functions]
            if all(l is not None for l in leaves):
                return leaves
        return None

    @classmethod
    def flush(cls, select=None, collect=None, keepalive=None):
        """``select(item) -> bool``: flush only those parked items now (the others stay parked): gradient buckets of a
        data-parallel step complete one after the other that way (runtime.flush_weight_grads_phased).
        ``collect``: a list that receives the (parameter or view, gradient) pairs INSTEAD of their delivery — a caller that
        computes them on another stream delivers after joining it (attention.SideResults); ``keepalive`` then receives the
        parked operands, which that stream's launches still read after this call returns."""
        if select is None:
            items, cls.pending = cls.pending, []
        else:
            items = [it for it in cls.pending if select(it)]
            cls.pending = [it for it in cls.pending if not select(it)]
        if not items:
            return
        if keepalive is not None:
            keepalive.extend(items)
        groups = {}
        for it in items:
            groups.setdefault((tuple(it[2].shape), tuple(it[3].shape)), []).append(it)
        roots, grads = [], []
        with torch.no_grad():
            for group in groups.values():
                n = len(group)
                if n == 1:
                    w, b, g2, x2 = group[0]
                    dW = torch.mm(g2.t(), x2)[None] if w is not None else None
                    dB = (colsum(g2 if g2.stride(1) == 1 else g2.contiguous()) if g2.is_cuda else g2.sum(0))[None] \
                        if b is not None else None
                else:
                    # views of one leaf (the q / k / v blocks of an in_proj_weight) next to each other, in memory order: their
                    # rows of the batched result then ARE the leaf's gradient (no stack launch in an UnbindBackward)
                    def order(e):
                        i, it = e
                        base = cls._view_of_leaf(it[0]) if cls.direct else None
                        return (0, i, 0) if base is None else (1, id(base), it[0].storage_offset())
                    group = [it for _, it in sorted(enumerate(group), key=order)]
                    need_w, need_b = any(it[0] is not None for it in group), any(it[1] is not None for it in group)
                    if _ptr_batchable(group):
                        # the operands where they are: rocBLAS' pointer-array batched GEMM and a column sum over a pointer table —
                        # the same kernels as below without the two torch.stack copies (64 items: 2 x 64 MB, 60 us of the tail)
                        dW, dB = _wgrad_ptrs(group, need_w, need_b)
                    else:
                        G = torch.stack([it[2] for it in group])                       # [n, rows, out]
                        dW = torch.bmm(G.transpose(1, 2), torch.stack([it[3] for it in group])) if need_w else None
                        dB = (colsum_batched(G) if G.is_cuda and G.dtype == torch.float32 else G.sum(1)) if need_b else None
                for which, R in ((0, dW), (1, dB)):
                    if R is None:
                        continue
                    i = 0
                    while i < n:
                        p = group[i][which]
                        if p is None:
                            i += 1
                            continue
                        base = cls._view_of_leaf(p) if cls.direct else None
                        k = 1
                        if base is not None and base.numel() % p.numel() == 0:
                            k = base.numel() // p.numel()
                            tiles = i + k <= n and all(
                                group[i + j][which] is not None and cls._view_of_leaf(group[i + j][which]) is base and
                                group[i + j][which].shape == p.shape and
                                group[i + j][which].storage_offset() == base.storage_offset() + j * p.numel() for j in range(k))
                            if tiles:
                                if collect is not None:
                                    collect.append((base, R[i:i + k].view(base.shape)))
                                else:
                                    cls._deliver(base, R[i:i + k].view(base.shape), roots, grads)
                                i += k
                                continue
                            k = 1
                        if collect is not None:
                            collect.append((p, R[i]))
                        else:
                            cls._deliver(p, R[i], roots, grads)
                        i += 1
        if roots:
            torch.autograd.backward(roots, grads)


class _Linear(torch.autograd.Function):
    """y = x W^T + b with the bias gradient as ONE launch (ATen: two-pass reduction, 2 launches / 14 us per layer)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w, b)
        return torch.nn.functional.linear(x, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w, b = ctx.saved_tensors
        g2 = g.reshape(-1, g.shape[-1])
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.mm(g2, w).view(x.shape)
        if DeferredParamGrads.enabled and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            DeferredParamGrads.pending.append((w if ctx.needs_input_grad[1] else None, b if ctx.needs_input_grad[2] else None,
                                               g2, x.reshape(-1, x.shape[-1])))
            return dx, None, None
        if ctx.needs_input_grad[1]:
            dw = torch.mm(g2.t(), x.reshape(-1, x.shape[-1]))
        if ctx.needs_input_grad[2]:
            g2c = g2 if g2.stride(1) == 1 else g2.contiguous()
            db = colsum(g2c)
        return dx, dw, db


class _LinearPair(torch.autograd.Function):
    """(x W1^T + b1, x W2^T + b2) for two [E, E] blocks that sit next to each other in one parameter (the q / k blocks of an
    in_proj_weight): ONE batched GEMM forward (both outputs contiguous), and the input gradient as a GEMM plus an
    accumulating GEMM instead of two GEMMs and an add.  Weight / bias gradients as `_Linear` (deferred when that is on)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        ctx.save_for_backward(x, w1, b1, w2, b2)
        E_out, E_in = w1.shape
        x2 = x.reshape(-1, E_in)
        W = torch.as_strided(w1, (2, E_out, E_in), (w2.storage_offset() - w1.storage_offset(), w1.stride(0), w1.stride(1)))
        Bv = torch.as_strided(b1, (2, 1, E_out), (b2.storage_offset() - b1.storage_offset(), 0, b1.stride(0)))
        y = torch.baddbmm(Bv, x2.unsqueeze(0).expand(2, -1, -1), W.transpose(1, 2))
        return y[0].view(x.shape[:-1] + (E_out,)), y[1].view(x.shape[:-1] + (E_out,))

    @staticmethod
    def backward(ctx, g1, g2):
        x, w1, b1, w2, b2 = ctx.saved_tensors
        g1 = g1.reshape(-1, g1.shape[-1])
        g2 = g2.reshape(-1, g2.shape[-1])
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.mm(g1, w1)
            torch.addmm(dx, g2, w2, out=dx)
            dx = dx.view(x.shape)
        x2 = x.reshape(-1, x.shape[-1])
        grads = [None, None, None, None]
        for i, (w, b, g) in enumerate(((w1, b1, g1), (w2, b2, g2))):
            need_w, need_b = ctx.needs_input_grad[1 + 2 * i], ctx.needs_input_grad[2 + 2 * i]
            if DeferredParamGrads.enabled and (need_w or need_b):
                DeferredParamGrads.pending.append((w if need_w else None, b if need_b else None, g, x2))
                continue
            if need_w:
                grads[2 * i] = torch.mm(g.t(), x2)
            if need_b:
                grads[2 * i + 1] = colsum(g if g.stride(1) == 1 else g.contiguous())
        return (dx,) + tuple(grads)


def linear_pair(x, w1, b1, w2, b2):
    """(linear(x, w1, b1), linear(x, w2, b2)); one batched GEMM when the two blocks are equally shaped, equally strided
    views into one storage (e.g. unbind of a packed projection weight) on the GPU."""
    ok = (x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled() and w1.shape == w2.shape and w1.stride() == w2.stride()
          and b1 is not None and b2 is not None and b1.shape == b2.shape and b1.stride() == b2.stride()
          and w1.untyped_storage().data_ptr() == w2.untyped_storage().data_ptr()
          and b1.untyped_storage().data_ptr() == b2.untyped_storage().data_ptr()
          and w2.storage_offset() > w1.storage_offset() and b2.storage_offset() > b1.storage_offset())
    if not ok:
        return linear(x, w1, b1), linear(x, w2, b2)
    if DeferredParamGrads.pending:
        raise RuntimeError("runtime.defer_weight_grads() is on but runtime.flush_weight_grads() was not called after "
                           "the last backward pass")
    return _LinearPair.apply(x, w1, b1, w2, b2)


def linear(x, w, b=None):
    """F.linear; on the GPU with a bias, the backward computes the bias gradient with the one-launch column sum."""
    if b is not None and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled() and (
            x.requires_grad or w.requires_grad or b.requires_grad):
        if DeferredParamGrads.pending:  # a forward pass with gradients of the previous backward still waiting: they would be lost
            raise RuntimeError("runtime.defer_weight_grads() is on but runtime.flush_weight_grads() was not called after "
                               "the last backward pass")
        return _Linear.apply(x, w, b)
    return torch.nn.functional.linear(x, w, b)


class PointwiseConv1d(nn.Conv1d):
    """``nn.Conv1d(cin, cout, kernel_size=1)`` (same parameters, same state-dict entries) evaluated as the GEMM it is.

    MIOpen wraps every such convolution in layout transposes and spends ~4.5 launches on its backward; a [cout, cin] x
    [cin, N] GEMM is one launch forward and two backward, and on MI355X the step pays ~4.5 us per launch whatever its
    size (profiles/)."""

    def __init__(self, in_channels, out_channels, bias=True):
        super().__init__(in_channels, out_channels, 1, bias=bias)

    def forward_no_bias(self, x):
        w = self.weight.squeeze(-1)
        if x.shape[0] == 1:
            return torch.mm(w, x.reshape(x.shape[1], x.shape[2])).unsqueeze(0)
        return torch.bmm(w.unsqueeze(0).expand(x.shape[0], -1, -1), x)

    def forward(self, x):
        w = self.weight.squeeze(-1)
        if x.shape[0] == 1:
            y = torch.mm(w, x.reshape(x.shape[1], x.shape[2])).unsqueeze(0)  # view, not x[0]: SelectBackward is a fill + a copy
        else:
            y = torch.bmm(w.unsqueeze(0).expand(x.shape[0], -1, -1), x)
        return y if self.bias is None else y + self.bias.view(1, -1, 1)


# ---- zero-copy concatenation of parameters that already sit next to each other in memory ---------------------------
# Five heads per stage, eight layers, 64 cpb MLPs: the batched GEMMs want their weights as ONE tensor, and torch.cat /
# torch.stack of the per-module parameters is a launch (~4.5 us on MI355X) per group and step.  dist.FlatParams lays
# the groups a model names in ``flat_param_groups()`` out contiguously; these helpers then return an alias of that memory
# (no launch) whose backward hands each parameter its slice of the joint gradient.  Parameters that are NOT adjacent
# (a model used without FlatParams, after .to(), a loaded checkpoint with its own storages) fall back to cat / stack.
def _adjacent(ts, slot=None):
    t0 = ts[0]
    esize = t0.element_size()
    ptr = t0.data_ptr()
    for t in ts:
        if t.data_ptr() != ptr or not t.is_contiguous() or t.dtype != t0.dtype or t.device != t0.device:
            return False
        ptr += (slot if slot is not None else t.numel()) * esize
    avail = t0.untyped_storage().nbytes() - t0.storage_offset() * esize
    return ptr - t0.data_ptr() <= avail


class _Alias(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mode, shape, *ts):
        ctx.mode, ctx.shapes = mode, [t.shape for t in ts]
        t0 = ts[0]
        out = torch.empty(0, dtype=t0.dtype, device=t0.device)
        out.set_(t0.untyped_storage(), t0.storage_offset(), shape)  # contiguous alias of the parameters' memory
        # no zero gradients for an alias nobody differentiated: with deferred weight gradients `_Linear.backward` returns None
        # for the alias, and a materialised zero would reach every parameter through AccumulateGrad (a fill, and one add_ per
        # parameter when DeferredParamGrads.flush delivers the real gradient: 34 launches per step at C2)
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None, None) + (None,) * len(ctx.shapes)
        g = g.contiguous()
        if ctx.mode == "cat":
            parts, off = [], 0
            for sh in ctx.shapes:
                parts.append(g.narrow(0, off, sh[0]))
                off += sh[0]
        elif ctx.mode == "stack":
            parts = list(g.unbind(0))
        else:  # "slot": [G, S, ...] with parameter i in rows [0, shape_i[0]) of slab i
            parts = [g[i, :sh[0]] for i, sh in enumerate(ctx.shapes)]
        return (None, None) + tuple(parts)


def cat_params(ts):
    """torch.cat(ts, 0), without a copy when the tensors are adjacent in memory."""
    ts = list(ts)
    if ts[0].is_cuda and _adjacent(ts):
        return _Alias.apply("cat", (sum(t.shape[0] for t in ts),) + tuple(ts[0].shape[1:]), *ts)
    return torch.cat(ts, 0)


def stack_params(ts):
    """torch.stack(ts), without a copy when the tensors are adjacent in memory."""
    ts = list(ts)
    if ts[0].is_cuda and _adjacent(ts):
        return _Alias.apply("stack", (len(ts),) + tuple(ts[0].shape), *ts)
    return torch.stack(ts)


def slot_stack_params(ts, rows):
    """[G, rows, ...] alias when parameter i (shape [r_i <= rows, ...]) starts slab i of a zero-padded slab array
    (dist.FlatParams lays such groups out on request); None otherwise."""
    ts = list(ts)
    rest = tuple(ts[0].shape[1:])
    per_row = 1
    for d in rest:
        per_row *= d
    if ts[0].is_cuda and all(tuple(t.shape[1:]) == rest and t.shape[0] <= rows for t in ts) and _adjacent(ts, rows * per_row):
        return _Alias.apply("slot", (len(ts), rows) + rest, *ts)
    return None


def buffers_alias(ts):
    """Adjacent buffers (running statistics) as one tensor that shares their memory, or None."""
    ts = list(ts)
    if ts[0].is_cuda and _adjacent(ts):
        t0 = ts[0]
        out = torch.empty(0, dtype=t0.dtype, device=t0.device)
        return out.set_(t0.untyped_storage(), t0.storage_offset(), (sum(t.shape[0] for t in ts),) + tuple(t0.shape[1:]))
    return None


class BatchNormDim1Swap(nn.BatchNorm1d):
    """BatchNorm over the channel axis of a sequence-first (L, N, C) tensor (helpers.py:36-53)."""

    def forward(self, x):
        return super().forward(x.permute(1, 2, 0)).permute(2, 0, 1)


NORM_DICT = {"bn": BatchNormDim1Swap, "bn1d": nn.BatchNorm1d, "id": nn.Identity, "ln": nn.LayerNorm}
ACTIVATION_DICT = {"relu": nn.ReLU, "gelu": nn.GELU, "leakyrelu": partial(nn.LeakyReLU, negative_slope=0.1)}
WEIGHT_INIT_DICT = {"xavier_uniform": nn.init.xavier_uniform_}


_POS_TOKEN_MAJOR = os.environ.get("VDETR_POS_TOKEN_MAJOR", "1") != "0"  # A/B switch (read once), _PosEmbedDeferred.forward


class DeferredPosEmbedGrads:
    """The learned query-position embeddings (Conv1d -> BatchNorm -> ReLU -> Conv1d on DETACHED box coordinates, one per decoder
    layer) exist only to train their own parameters: nothing upstream waits for their backward.  With
    runtime.defer_weight_grads() the forward runs without autograd nodes inside (one node for the whole block), the backward
    parks the incoming gradient, and `flush()` runs the parameter gradients of ALL layers as batched GEMMs + one batched
    BatchNorm backward (8 layers: ~10 launches instead of 48)."""
    pending = []

    @classmethod
    def flush(cls, collect=None, keepalive=None):
        """``collect`` / ``keepalive``: as DeferredParamGrads.flush (the launches may then run on another stream: the caller
        delivers the pairs after joining it)."""
        items, cls.pending = cls.pending, []
        if not items:
            return
        if keepalive is not None:
            keepalive.extend(items)
        from . import bn_act as BNA
        groups = {}
        for it in items:
            groups.setdefault((tuple(it[1].shape), tuple(it[2].shape)), []).append(it)
        roots, grads = [], []
        with torch.no_grad():
            for group in groups.values():
                n = len(group)
                B, C, N = group[0][2].shape
                flat = (lambda t: t.reshape(t.shape[1], N)) if B == 1 else (lambda t: t.permute(1, 0, 2).reshape(t.shape[1], B * N))
                D = torch.stack([flat(it[4]) for it in group])                       # [n, C, B*N] gradient of the output
                Y = torch.stack([flat(it[2]) for it in group])                       # hidden activations
                X = torch.stack([flat(it[1]) for it in group])                       # [n, Cin, B*N]
                W2 = stack_params([it[0].position_embedding_head[3].weight.squeeze(-1) for it in group])
                dB2 = D.sum(2)
                dW2 = torch.bmm(D, Y.transpose(1, 2))
                dY = torch.bmm(W2.transpose(1, 2), D)                                # [n, C, B*N]
                dys = [dY[i].view(1, C, N) if B == 1 else dY[i].view(C, B, N).permute(1, 0, 2).contiguous() for i in range(n)]
                dH = torch.empty((n, B, C, N), dtype=D.dtype, device=D.device)
                res = BNA.backward_from_records([it[3] for it in group], dys, [dH[i] for i in range(n)])
                dW1 = torch.bmm(torch.stack([flat(dH[i]) for i in range(n)]) if B > 1 else dH.view(n, C, N), X.transpose(1, 2))
                for i, it in enumerate(group):
                    head = it[0].position_embedding_head
                    for p, g in ((head[3].weight, dW2[i].unsqueeze(-1)), (head[3].bias, dB2[i]), (head[1].weight, res[i][1]),
                                 (head[1].bias, res[i][2]), (head[0].weight, dW1[i].unsqueeze(-1))):
                        if p is not None and p.requires_grad:
                            if collect is not None:
                                collect.append((p, g))
                            else:
                                DeferredParamGrads._deliver(p, g, roots, grads)
        if roots:
            torch.autograd.backward(roots, grads)


class _PosEmbedDeferred(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, *params):  # params: listed so that the output requires grad; their gradients come at the flush
        from . import bn_act as BNA
        from . import heads as HD
        conv1, bn, _, conv2 = module.position_embedding_head
        x_tok = x.transpose(1, 2)  # (the caller's [B, N, cin] coordinates)
        if conv2.bias is not None and HD.pos_mlp_usable(module, x_tok):
            # the whole block as ONE launch (csrc/heads.hip: the batch statistics of the first convolution follow from the
            # coordinates' mean and covariance); it writes what DeferredPosEmbedGrads.flush reads
            out, hact, rec, _ = HD.pos_mlp_forward(module, x_tok)
            ctx.rec = (module, x, hact, rec)
            return out.permute(1, 2, 0)  # [B, C, N] view of the dense sequence-first result
        h = PointwiseConv1d.forward_no_bias(conv1, x)
        y, rec = BNA.forward_record(h, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum, 0.0, 0,
                                    counters=[bn.num_batches_tracked], pre_bias=conv1.bias)
        ctx.rec = (module, x, y, rec)
        if _POS_TOKEN_MAJOR and y.shape[0] == 1 and conv2.bias is not None:
            # one scene: the output layer as out^T [N, C] = y^T W^T + b — ONE launch (the bias rides in the GEMM's epilogue; the
            # channel-major form is a GEMM + a broadcast add), and the [N, B, C] view the decoder adds to its queries twice per
            # layer is then dense memory (the channel-major result reached those adds as a transposed view)
            out_t = torch.addmm(conv2.bias, y[0].t(), conv2.weight.squeeze(-1).t())
            return out_t.t().unsqueeze(0)
        return conv2(y)

    @staticmethod
    def backward(ctx, d_out):
        DeferredPosEmbedGrads.pending.append(ctx.rec + (d_out,))  # (a permuted view: the flush's stack is the only copy)
        return (None,) * (2 + 5)


class PositionEmbeddingLearned(nn.Module):
    """(B, N, C_in) coordinates -> (B, num_pos_feats, N) learned embedding (helpers.py:17-33)."""

    def __init__(self, input_channel, num_pos_feats=288):
        super().__init__()
        self.position_embedding_head = nn.Sequential(
            PointwiseConv1d(input_channel, num_pos_feats),
            nn.BatchNorm1d(num_pos_feats),
            nn.ReLU(inplace=True),
            PointwiseConv1d(num_pos_feats, num_pos_feats),
        )

    def forward(self, xyz):
        head = self.position_embedding_head
        x = xyz.transpose(1, 2)  # [B, C_in, N]; the 1x1 convolution reads the transposed operand in place
        bn = head[1]
        if not (self.training and x.is_cuda and type(bn) is nn.BatchNorm1d and bn.momentum is not None and bn.track_running_stats):
            from . import bn_act as BNA  # (cross-replica statistics on this path too while bn_act.set_sync is on)
            return BNA.run_sequential(head, x.contiguous())
        if DeferredParamGrads.enabled and DeferredParamGrads.direct and not x.requires_grad and torch.is_grad_enabled() \
                and head[3].bias is not None:
            return _PosEmbedDeferred.apply(self, x, head[0].weight, head[1].weight, head[1].bias, head[3].weight,
                                           head[3].bias)
        from . import bn_act as BNA  # BatchNorm1d + ReLU as one launch (and one backward)
        conv = head[0]
        # the convolution's bias cancels under batch statistics (zero gradient): it only enters the running mean
        h = conv(x) if conv.bias is None else PointwiseConv1d.forward_no_bias(conv, x)
        y = BNA.bn_act(h, bn.weight, bn.bias, bn.running_mean, bn.running_var, True, bn.eps, bn.momentum, relu=True,
                       dropout_p=0.0, pre_bias=conv.bias, counters=[bn.num_batches_tracked])
        return head[3](y)


class GenericMLP(nn.Module):
    """Stack of (Conv1d|Linear) -> norm -> activation -> dropout blocks followed by an output layer.

    Constructor arguments, defaults and the order of sub-modules inside ``self.layers`` follow helpers.py:74-141,
    so ``state_dict()`` keys line up with reference checkpoints.
    """

    def __init__(self, input_dim, hidden_dims, output_dim, norm_fn_name=None, activation="relu", use_conv=False,
                 dropout=None, hidden_use_bias=False, output_use_bias=True, output_use_activation=False,
                 output_use_norm=False, weight_init_name=None):
        super().__init__()
        act = ACTIVATION_DICT[activation]
        norm = NORM_DICT[norm_fn_name] if norm_fn_name is not None else None
        if norm_fn_name == "ln" and use_conv:
            norm = lambda ch: nn.GroupNorm(1, ch)  # LayerNorm over channels for (B, C, N) tensors
        if dropout is not None and not isinstance(dropout, list):
            dropout = [dropout] * len(hidden_dims)

        def affine(cin, cout, bias):
            return PointwiseConv1d(cin, cout, bias=bias) if use_conv else nn.Linear(cin, cout, bias=bias)

        mods, cin = [], input_dim
        for i, width in enumerate(hidden_dims):
            mods.append(affine(cin, width, hidden_use_bias))
            if norm:
                mods.append(norm(width))
            mods.append(act())
            if dropout is not None:
                mods.append(nn.Dropout(p=dropout[i]))
            cin = width
        mods.append(affine(cin, output_dim, output_use_bias))
        if output_use_norm:
            mods.append(norm(output_dim))
        if output_use_activation:
            mods.append(act())
        self.layers = nn.Sequential(*mods)
        if weight_init_name is not None:
            self.do_weight_init(weight_init_name)

    def do_weight_init(self, weight_init_name):
        init = WEIGHT_INIT_DICT[weight_init_name]
        for _, p in self.named_parameters():
            if p.dim() > 1:
                init(p)

    def forward(self, x):
        mods = list(self.layers)
        if not (self.training and x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and torch.is_grad_enabled()):
            from . import bn_act as BNA
            return BNA.run_sequential(self.layers, x)
        # Conv1d(k=1) -> BatchNorm1d -> ReLU [-> Dropout] blocks as GEMM + ONE fused launch (bn_act.py: statistics, affine,
        # ReLU, dropout and the running-statistics bookkeeping; the convolution's bias only enters the running mean)
        from . import bn_act as BNA
        i = 0
        while i < len(mods):
            m = mods[i]
            nxt = mods[i + 1:i + 4]
            if (type(m) is PointwiseConv1d and len(nxt) >= 2 and type(nxt[0]) is nn.BatchNorm1d and type(nxt[1]) is nn.ReLU
                    and nxt[0].track_running_stats and nxt[0].momentum is not None and nxt[0].affine):
                bn = nxt[0]
                drop = nxt[2] if len(nxt) >= 3 and type(nxt[2]) is nn.Dropout else None
                salts = self.__dict__.setdefault("_bn_salts", {})
                if i not in salts:
                    salts[i] = BNA.new_salt()
                h = m(x) if m.bias is None else PointwiseConv1d.forward_no_bias(m, x)
                x = BNA.bn_act(h, bn.weight, bn.bias, bn.running_mean, bn.running_var, True, bn.eps, bn.momentum, relu=True,
                               dropout_p=drop.p if drop is not None else 0.0, salt=salts[i], pre_bias=m.bias,
                               counters=[bn.num_batches_tracked])
                i += 3 if drop is None else 4
            else:
                x = BNA.batch_norm_module(m, x) if isinstance(m, nn.modules.batchnorm._BatchNorm) else m(x)
                i += 1
        return x


def get_clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])
