"""Autograd bindings of the fused attention kernels (lib/libvdetr_hip.so: vdetr_attn_fwd_f32 /
vdetr_attn_bwd_scores_f32).

One ``torch.autograd.Function`` covers the three attention flavours of the decoder:
  * 3DV-RPE cross attention  (GlobalShareCrossAttention core, vdetr_transformer.py:710-753): shared K/V + RPE table
  * ShareSelfAttention core  (:644-648): shared K/V, no bias
  * nn.MultiheadAttention core (:468): per-head K/V
Everything outside the core (q/k/v/out projections, the cpb MLP that produces the table) stays in PyTorch so
that parameters, their names and their gradients flow exactly as in the reference modules.

Forward saves the biased scores S [rows, nK] and the row log-sum-exp; backward is
  shared K/V, 4 heads: one pass over S (attn_bwd_kv.hip: dP~ = dO V^T, P~, dS, dV, dK) -> dTable from dS -> dQ = dS K (GEMM);
  otherwise: dP~ = dO V^T (GEMM) -> kernel: P~, dS, dTable -> dV = P~^T dO, dK = dS^T q, dQ = dS K (GEMMs).
"""
import contextlib
import ctypes
import os
import weakref

import torch
from torch.autograd import Function

from . import _lib as L

HEAD_DIM = 64


class RPEConfig:
    """Static parameters of the 3DV-RPE lookup (vdetr_transformer.py:671-683,722-723)."""

    def __init__(self, table_size=10, log_scale=512.0, max_value=4.0):
        self.table_size = int(table_size)
        self.log_scale = float(log_scale)
        self.max_value = float(max_value)

    @property
    def inv_log_norm(self):
        # deltas / np.log2(8) / max_value
        return 1.0 / (3.0 * self.max_value)


def new_rng_state(device, seed=None):
    """Device-resident {seed, offset} of the counter-based dropout generator."""
    if seed is None:
        seed = torch.initial_seed()
    return torch.tensor([seed & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=device)


# Per-device master state + the snapshot of the current step.  ``begin_step`` takes a snapshot (a NEW tensor that
# is never written again, so autograd can keep it by reference) and bumps the master offset with a device op:
# the same captured hipGraph therefore draws fresh dropout masks on every replay.  Modules that share a snapshot
# stay independent through their per-module ``salt`` (the by-value seed of the descriptor).
DYNAMIC_BWD = os.environ.get("VDETR_BWD_DYNAMIC", "1") != "0"
# forward kernel of the 3DV-RPE attention (vdetr_attn_desc.fwd_kernel): 0 = persistent workgroups (attn_fwd_pipe.hip), 1 = the
# round-4 grid kernel (A/B runs)
FWD_KERNEL = int(os.environ.get("VDETR_FWD_KERNEL", "0"))
# table-gradient kernel (vdetr_attn_desc.bwd_kernel): 0 = the box kernel where every query's vertices are a box, 1 = the general
# kernel only (the parity tests compare the two); TABLE_GRID: workgroups of an in-line table-gradient launch, 0 = one per CU
BWD_KERNEL = int(os.environ.get("VDETR_BWD_KERNEL", "0"))
TABLE_GRID = 0
# shared-KV backward: dO V^T, the softmax backward, dV and dK in one pass over the scores (attn_bwd_kv.hip) instead of
# three library GEMMs around an element-wise kernel.  VDETR_BWD_FUSED=0 keeps the GEMM path (A/B measurements, parity).
FUSED_KV_BWD = os.environ.get("VDETR_BWD_FUSED", "1") != "0"
# dQ by attn_bwd_dq.hip (row-owner, exact fp32) instead of the library's batched GEMM.  OFF: alone it wins for per-head K/V (the
# query self-attention: 11.7 against 19.2 us) and loses for shared K/V (31 against 26 us: every 16-row workgroup re-reads all of
# K), but in the step these launches run NEXT TO the table-gradient branch on a quarter of the CUs, where a kernel bound by
# matrix time pays 4x: C2 8.34 ms with the library, 8.37-8.41 with the kernel for the self-attention (VDETR_BWD_DQ=1), 8.45-8.48
# for both (=2).
_SELF_FWD8 = os.environ.get("VDETR_SELF_FWD8", "0") != "0"  # A/B switch (read once): see _FusedAttention.forward
SELF_FWD_BODY = os.environ.get("VDETR_SELF_FWD", "") == "body"  # per-head forward: attn_fwd.hip's body instead of attn_fwd_self.hip (A/B; tests flip it)
_DQ_KERNEL = os.environ.get("VDETR_BWD_DQ", "0") != "0"
_DQ_KERNEL_SHARED = os.environ.get("VDETR_BWD_DQ", "0") == "2"


# The RPE-table gradient kernel (2.3 ms of a 9 ms step) feeds parameters only: nothing on the backward's critical path waits
# for it.  `fused_attention(..., table_grad_async=True)` launches it on a side stream; the caller wraps the table with
# `join_table_grad` BEFORE the layers that use it (GlobalShareCrossAttention.precompute), so that autograd reaches the join —
# main stream waits for the side stream — only after those layers' backward.
# Round 3 measured this SLOWER (10.07 vs 9.62 ms): the kernel's persistent workgroups keep every register and 150 KB of LDS of
# every CU for its whole run, so whatever the main chain launched next to it waited for a CU.  Round 4: the side-stream launch
# takes ASYNC_TABLE_GRID of the 256 CUs (vdetr_attn_bwd_table_set_grid) and leaves the others to the main chain, whose ~40
# small launches per layer cannot fill the chip anyway: 9.02 -> 8.60 ms at 192 (224: 8.84, 176: 8.67, 160: 8.71, 128: 9.11),
# C5 27.7 -> 26.7, C4 9.42 -> 9.08; C1 (64 queries) 2.14 -> 2.31, hence the size gate.  DESIGN.md 4.4e.
#   VDETR_BWD_ASYNC_TABLE = auto (default: launches of >= 2^21 query-key pairs) | 1 (always) | 0 (never)
_ASYNC_ENV = os.environ.get("VDETR_BWD_ASYNC_TABLE", "auto")
ASYNC_TABLE_GRAD = _ASYNC_ENV != "0"
ASYNC_MIN_PAIRS = 1 << 21
# workgroups (= CUs) of the table kernel when it runs on the side stream: the remaining CUs are the main chain's.  Round 5: 190,
# not 192 — the row-block launches of the backward chain (csrc/rowblock.hip) are 64 workgroups for 1024 queries, and the next
# scene's sampling kernel holds one CU for the first ~4.7 ms of a step: on 64 - 1 CUs every such launch took two rounds.
# Step 7.31 ms at 192, 7.01 at 190, 7.06 at 188, 7.09 at 184, 7.26 at 176 (profiles/r05_step_bounds.txt).
ASYNC_TABLE_GRID = int(os.environ.get("VDETR_BWD_ASYNC_GRID", "190"))
_ASYNC_KV4 = os.environ.get("VDETR_BWD_ASYNC_KV_WAVES", "8") == "4"
# vdetr_attn_desc.kv_halves = 1 (one workgroup per key tile) while a table kernel is live — 1: the per-head pass only; 2: the
# shared-K/V pass as well; 0: never.  With the side grid at 192 the per-head form won 0.03 ms (7.36 -> 7.33; the
# shared-K/V form lost: 7.41); at 190 the default shape is as fast or faster (7.01 / 7.03-7.05 / 7.18 for 0 / 1 / 2).
# (round 6, on the final step: 0 / 1 / 2 = 6.41-6.42 / 6.39 / 6.58-6.59 ms: the per-head form is the default now)
KV_ONE_WG = int(os.environ.get("VDETR_BWD_KV_ONE_WG", "1"))


_step_side = {}  # device key -> this step's forward ran a cross-attention whose table gradient will go to the side stream


def side_branch_in_use(device):
    """True when the current step's backward will launch table-gradient kernels on the side stream of `device`: other
    parameter-only work (the box heads' weight gradients) may then go there too and is joined at the flush (SideResults)."""
    return bool(_step_side.get(_dev_key(device)))


class SideResults:
    """Parameter gradients other modules computed on the side stream: [(device, [(parameter or alias, gradient)], tensors
    kept alive)]; delivered after the join at runtime.flush_weight_grads (DeferredTableGrads.flush)."""
    pending = []

    @classmethod
    def flush(cls):
        from .helpers import DeferredParamGrads
        items, cls.pending = cls.pending, []
        late, side_late[:] = list(side_late), []
        for fn, alive in late:  # handed over for the end of the backward, but nobody ran them there: here, on this stream
            with torch.no_grad():
                items.append((None, fn(), alive))
        for dev, pairs, _alive in items:
            side = _side_streams.get(_dev_key(dev)) if dev is not None else None
            if side is not None:
                torch.cuda.current_stream(dev).wait_stream(side)
            roots, grads = [], []
            with torch.no_grad():
                for p, g in pairs:
                    if p is not None and p.requires_grad:
                        DeferredParamGrads._deliver(p, g, roots, grads)
            if roots:
                torch.autograd.backward(roots, grads)


_FLUSH_SIDE = os.environ.get("VDETR_FLUSH_SIDE", "1") != "0"
_FLUSH_SIDE_POS = os.environ.get("VDETR_FLUSH_SIDE_POS", "1") != "0"
_FLUSH_SIDE_LN = os.environ.get("VDETR_FLUSH_SIDE_LN", "1") != "0"
_tick = {}
side_late = []  # parameter-only work other modules hand over for the END of the backward: callables -> [(parameter, gradient)], keep-alive


def flush_layer_params_on_side(ref, rows):
    """Called from the backward pass when every decoder layer's backward has run (vdetr_transformer._LayersDone): the parked
    weight / bias gradients of the layers' linear maps (the items with `rows` rows: 64 of them at the model's size) are computed
    on the side branch — behind the last table kernel — while the main stream goes on with the first layer's and the
    projection's backward; runtime.flush_weight_grads joins and delivers (SideResults).  Only where that branch is in use."""
    from .helpers import DeferredParamGrads, DeferredPosEmbedGrads
    if not (_FLUSH_SIDE and ref.is_cuda and DeferredParamGrads.enabled and DeferredParamGrads.direct and side_branch_in_use(ref.device)):
        return
    dev = ref.device
    key = _dev_key(dev)
    cur = torch.cuda.current_stream(dev)
    fork = torch.cuda.Event()
    fork.record(cur)
    # the chain's next captured launch comes BEFORE the branch's: a captured graph keeps a node's first successor on the node's
    # queue (see _FusedAttention.backward); nothing of the chain is launched between here and the return, hence this 2 us kernel
    if key not in _tick:
        _tick[key] = torch.zeros(4, device=dev)
    _tick[key].add_(0.0)
    side = _side_stream(dev)
    side.wait_event(fork)
    pairs, keep = [], []
    late, side_late[:] = list(side_late), []
    from .runtime import ts_mark
    ts_mark("main: all layers' backward done")
    with torch.cuda.stream(side):
        ts_mark("side: parked gradients start")
        DeferredParamGrads.flush(select=lambda it: it[2].shape[0] == rows, collect=pairs, keepalive=keep)
        if _FLUSH_SIDE_POS:  # the layers' learned query-position embeddings: complete as well (one per layer)
            DeferredPosEmbedGrads.flush(collect=pairs, keepalive=keep)
        if _FLUSH_SIDE_LN:   # the LayerNorm parameter sums of the layers' residual blocks (the first layer's follow at the flush)
            from .add_ln import DeferredLnGrads
            DeferredLnGrads.flush(collect=pairs, keepalive=keep)
        for fn, alive in late:  # (e.g. the box heads' weight gradients: vdetr_transformer._DeferredHeads)
            pairs += fn()
            keep.append(alive)
        ts_mark("side: parked gradients end")
    if pairs:
        SideResults.pending.append((dev, pairs, keep))


# Round 6: the parked parameter gradients of a decoder layer leave for the side branch as soon as that layer's backward has
# produced them, instead of all layers' together behind the last table kernel.  The step's real timeline (timestamp kernels in the
# captured graph, tools/probes/step_timeline.py) showed why: during the layers' backward the MAIN chain is the critical path (330-415
# us per layer on the 66 CUs the table kernel leaves it, against ~300 us per table kernel), so the side stream sits idle for
# 20-113 us behind every table kernel — 470 us per step — and then ran 480 us of weight-gradient GEMMs after the last one, with
# the main chain long done.  MEASURED AND NOT KEPT (off by default; VDETR_FLUSH_PER_LAYER=1: the layers' weight gradients, =2: the
# LayerNorm sums and the position MLPs' as well): per layer the work is 4 / 14 launches and takes 100 / 170 us next to the chain —
# 8 x that is more than the one batched pass at the end (480 us), the side branch stays the longer one: 6.85 -> 7.06 / 7.47 ms
# (profiles/r06_step_timeline.txt).
_FLUSH_PER_LAYER = os.environ.get("VDETR_FLUSH_PER_LAYER", "0") != "0"
_FLUSH_PER_LAYER_ALL = os.environ.get("VDETR_FLUSH_PER_LAYER", "0") == "2"


def side_flush_begin(ref):
    """Called inside a layer's backward once its parked operands are complete and BEFORE its last launch of the chain: records
    the fork (the chain's next launch then stays the fork node's first successor: it keeps its hardware queue, see
    _FusedAttention.backward).  Returns a token for side_flush_end, or None where the side branch is not in use."""
    from .helpers import DeferredParamGrads
    if not (_FLUSH_SIDE and _FLUSH_PER_LAYER and ref.is_cuda and DeferredParamGrads.enabled and DeferredParamGrads.direct
            and side_branch_in_use(ref.device) and _side_keep):
        return None
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(ref.device))
    return ev


def side_flush_end(ev, ref, rows):
    """the parked gradients that were complete at side_flush_begin, computed on the side branch (delivered at the flush: SideResults)"""
    if ev is None:
        return
    from .helpers import DeferredParamGrads, DeferredPosEmbedGrads
    from .runtime import ts_mark
    dev = ref.device
    side = _side_stream(dev)
    side.wait_event(ev)
    pairs, keep = [], []
    with torch.cuda.stream(side):
        ts_mark("side: layer's parked gradients start")
        DeferredParamGrads.flush(select=lambda it: it[2].shape[0] == rows, collect=pairs, keepalive=keep)
        if _FLUSH_PER_LAYER_ALL:  # (the position MLPs' and LayerNorms' parameters too: ~14 launches, 170 us per layer — too long)
            if _FLUSH_SIDE_POS:
                DeferredPosEmbedGrads.flush(collect=pairs, keepalive=keep)
            if _FLUSH_SIDE_LN:
                from .add_ln import DeferredLnGrads
                DeferredLnGrads.flush(collect=pairs, keepalive=keep)
        ts_mark("side: layer's parked gradients end")
    if pairs:
        SideResults.pending.append((dev, pairs, keep))


def set_async_table_grad(mode):
    """"auto" | "1" | "0" (the VDETR_BWD_ASYNC_TABLE values) from here on; returns the previous mode.  A caller whose own side
    streams already fill the hardware queues (bench.BackboneTrainer: loader stream + sampling kernel) turns it off."""
    global _ASYNC_ENV, ASYNC_TABLE_GRAD
    prev, _ASYNC_ENV = _ASYNC_ENV, str(mode)
    ASYNC_TABLE_GRAD = _ASYNC_ENV != "0"
    return prev


def _async_wanted(B, nQ, nK):
    return ASYNC_TABLE_GRAD and (_ASYNC_ENV == "1" or B * nQ * nK >= ASYNC_MIN_PAIRS)


_side_streams = {}
_side_keep = []  # tensors the side-stream kernels read, alive until the join


def _dev_key(device):
    device = torch.device(device)
    if device.type != "cuda":
        return (device.type, 0)
    return (device.type, device.index if device.index is not None else torch.cuda.current_device())


def _side_stream(device):
    key = _dev_key(device)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


class _JoinTableGrad(Function):
    @staticmethod
    def forward(ctx, t):
        ctx.dev = t.device
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        side = _side_streams.get(_dev_key(ctx.dev))
        if side is not None and _side_keep:  # (only if something was launched there: a capture must not wait for a stream outside it)
            torch.cuda.current_stream(ctx.dev).wait_stream(side)
        _side_keep.clear()
        return g


class _ParkTableGrad(Function):
    """identity on `table` (already cut from its graph); the gradient that arrives is parked, nothing flows on"""

    @staticmethod
    def forward(ctx, anchor, table, slot):
        ctx.slot = slot
        return table.view_as(table)

    @staticmethod
    def backward(ctx, g):
        ctx.slot[1] = g
        return None, None, None


class DeferredTableGrads:
    """The gradient of the RPE tables' MLPs, run AFTER the backward pass and the other parked parameter gradients
    (runtime.flush_weight_grads): the side stream's last table kernel then also overlaps the weight-gradient GEMMs instead of
    being waited for inside the backward.  Only with runtime.defer_weight_grads() (the training loop promises a flush)."""
    pending = []   # [tables as computed (with their graph), [slot = [index, grad] per layer], stacked accumulator or None, mlp]
    _anchor = {}
    _begun = []    # results of begin_flush(): (parameter alias, gradient) pairs computed on the side stream
    buffers = {}   # data_ptr of a layer's (cut) table -> its slice of the stacked accumulator, for the attention backward
    _outputs = {}      # id(slots) of an entry -> weak references to the tensors park() handed out for it
    _buffer_keys = {}  # id(slots) of an entry -> its keys in `buffers`

    @classmethod
    def park(cls, tables, mlp=None):
        """tables [n, ...] (requires grad) -> n tensors for the n layers, cut from the graph.
        ``mlp`` = (c1 [T^3, 4], w1 [8n, hid, 3], b1 [8n, hid], w2 [8n, H, hid], hidden [8n, T^3, hid]): the tables were computed
        WITHOUT a graph as relu([coords, 1] [w1, b1]^T) w2^T; their backward is then three GEMMs that `begin_flush` puts on
        the side stream right behind the last table kernel, so nothing of it sits on the main stream."""
        key = _dev_key(tables.device)
        if key not in cls._anchor:  # a leaf that makes the outputs require grad; it never receives one
            cls._anchor[key] = torch.zeros(1, device=tables.device, requires_grad=True)
        # Earlier entries stay until the flush: two forward passes followed by one backward (micro-batches with a summed loss, a
        # second model instance) must both deliver.  An entry is dropped here only when it is provably dead: every tensor park()
        # handed out for it is gone and no gradient has arrived (a forward pass that was never differentiated).
        alive = []
        for item in cls.pending:
            _tables, slots, _acc, _mlp = item
            outs = cls._outputs.get(id(slots), ())
            if all(g is None for _, g in slots) and all(r() is None for r in outs):
                cls._outputs.pop(id(slots), None)
                for key_ in cls._buffer_keys.pop(id(slots), ()):
                    cls.buffers.pop(key_, None)
                continue
            alive.append(item)
        cls.pending = alive
        if sum(1 for _, slots, _, _ in alive if any(g is not None for _, g in slots)) > 32:
            raise RuntimeError("runtime.defer_weight_grads() is on but runtime.flush_weight_grads() has not been called for 32 "
                               "backward passes (parked RPE-table gradients pile up)")
        cut = tables.detach()
        slots = [[i, None] for i in range(tables.shape[0])]
        # the layers' accumulators as slices of ONE zeroed buffer in layer order: the flush hands it to the tables' backward
        # as it is (a stack of 8 separately placed tensors is 8 copy nodes of ~9 us each in a captured step)
        acc = None
        if tables.is_cuda and tables.dtype == torch.float32:
            acc = _take_zeros(tables, tuple(tables.shape), tables.dtype)
            for i in range(tables.shape[0]):
                cls.buffers[cut[i].data_ptr()] = acc[i]
            cls._buffer_keys[id(slots)] = [cut[i].data_ptr() for i in range(tables.shape[0])]
        cls.pending.append((tables, slots, acc, mlp))
        outs = [_ParkTableGrad.apply(cls._anchor[key], cut[i], slots[i]) for i in range(tables.shape[0])]
        cls._outputs[id(slots)] = [weakref.ref(o) for o in outs]
        return outs

    @classmethod
    def begin_flush(cls):
        """First thing of runtime.flush_weight_grads: for tables parked with their MLP's operands, the MLPs' backward (three
        batched GEMMs + the ReLU mask) is enqueued on the side stream behind the table kernels it reads; flush() joins."""
        keep = []
        for item in cls.pending:
            tables, slots, acc, mlp = item
            if mlp is None or all(g is None for _, g in slots):
                keep.append(item)
                continue
            dev = tables.device
            # on the side stream only where this step's table kernels ran there (a fork and a join just for these launches
            # cost a small step more than they hide: C1 2.08 -> 2.29 ms)
            side = _side_streams.get(_dev_key(dev)) if (tables.is_cuda and _side_keep) else None
            cur = torch.cuda.current_stream(dev) if tables.is_cuda else None
            if acc is None or not all(g is None or g.data_ptr() == acc[i].data_ptr() for i, g in slots):
                # a layer's backward did not accumulate in place (e.g. the GEMM composition of the attention backward)
                if side is not None and _side_keep:
                    cur.wait_stream(side)
                acc = torch.stack([g if g is not None else torch.zeros_like(tables[i]) for i, g in slots])
                if side is not None:
                    side.wait_stream(cur)
            c1, w1, b1, w2, hid = mlp
            n8 = hid.shape[0]
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()), torch.no_grad():
                dT = acc.reshape(n8, hid.shape[1], -1)                                # [8n, T^3, H]
                dW2 = torch.bmm(dT.transpose(1, 2), hid)                              # [8n, H, hid]
                dhid = torch.ops.aten.threshold_backward(torch.bmm(dT, w2.detach()), hid, 0.0)
                dW1b = torch.bmm(dhid.transpose(1, 2), c1.unsqueeze(0).expand(n8, -1, -1))   # [8n, hid, 4]: weights | bias
                if side is not None:
                    from .runtime import ts_mark
                    ts_mark("side: table MLPs' backward end")
            cls._begun.append((dev if side is not None else None, ((w1, dW1b[..., :3]), (b1, dW1b[..., 3]), (w2, dW2)),
                               (acc, dT, dhid, hid)))
        cls.pending = keep

    @classmethod
    def flush(cls):
        from .helpers import DeferredParamGrads
        SideResults.flush()
        begun, cls._begun = cls._begun, []
        for dev, pairs, _alive in begun:
            side = _side_streams.get(_dev_key(dev)) if dev is not None else None  # (None: computed on the current stream)
            if side is not None:
                torch.cuda.current_stream(dev).wait_stream(side)
            _side_keep.clear()
            roots, grads = [], []
            with torch.no_grad():
                for p, g in pairs:
                    if p.requires_grad:
                        DeferredParamGrads._deliver(p, g, roots, grads)
            if roots:
                torch.autograd.backward(roots, grads)
        items, cls.pending = cls.pending, []
        cls.buffers.clear()
        cls._outputs.clear()
        cls._buffer_keys.clear()
        for tables, slots, acc, _mlp in items:
            if all(g is None for _, g in slots):
                continue
            if not tables.requires_grad:
                raise RuntimeError("tables parked with their MLP's operands can only be flushed through begin_flush()")
            dev = tables.device
            side = _side_streams.get(_dev_key(dev))
            if side is not None and _side_keep:
                torch.cuda.current_stream(dev).wait_stream(side)
            _side_keep.clear()
            if acc is not None and all(g is None or g.data_ptr() == acc[i].data_ptr() for i, g in slots):
                g = acc  # every layer accumulated in place (a layer without a gradient left its slice zero)
            else:
                g = torch.stack([g if g is not None else torch.zeros_like(tables[i]) for i, g in slots])
            torch.autograd.backward([tables], [g])


def table_grads_parkable(ref):
    """True where `park_table_grads` parks: side-stream table gradients + parked weight gradients, on the GPU, under grad mode"""
    from .helpers import DeferredParamGrads
    return bool(ASYNC_TABLE_GRAD and DeferredParamGrads.enabled and ref.is_cuda and ref.requires_grad and torch.is_grad_enabled()
                and os.environ.get("VDETR_BWD_ASYNC_PARK", "1") != "0")


def park_table_grads(tables, mlp=None):
    """[n, 8, T, T, T, H] tables of n layers -> list of n per-layer tables.  With the table gradient on the side stream and
    parked weight gradients (runtime.defer_weight_grads) the tables are cut from their graph and their MLPs' backward runs at
    the flush (`mlp`: see DeferredTableGrads.park — the caller checked `table_grads_parkable` and computed the tables without
    a graph); otherwise each table passes through `join_table_grad`."""
    if mlp is not None or table_grads_parkable(tables):
        _side_stream(tables.device)
        return DeferredTableGrads.park(tables, mlp)
    return [join_table_grad(t) for t in tables.unbind(0)]


def join_table_grad(table):
    """Identity on `table`; in the backward pass the current stream waits for the side stream on which the gradients of the
    tables derived from it (`fused_attention(..., table_grad_async=True)`) were computed."""
    if ASYNC_TABLE_GRAD and table is not None and table.is_cuda and table.requires_grad and torch.is_grad_enabled():
        _side_stream(table.device)
        return _JoinTableGrad.apply(table)
    return table


def _table_accumulator(table):
    """zeroed gradient accumulator of one layer's table: its slice of the parked tables' stacked buffer where there is one"""
    buf = DeferredTableGrads.buffers.pop(table.data_ptr(), None)
    if buf is not None and buf.shape == table.shape:
        return buf
    return _take_zeros(table, tuple(table.shape), table.dtype)


def _launch_table_async(lib, d, q, ds, table, aux, vertices, xyz, mask, fork=None, dtable=None, also=()):
    """the table-gradient launches of one layer on the side stream, behind everything the current stream has queued (dS, the
    zeroed accumulator and bwd_aux are ready then); returns the accumulator, which `join_table_grad`'s backward makes final"""
    if dtable is None:
        dtable = _take_zeros(table, tuple(table.shape), table.dtype)
    nbytes = lib.vdetr_attn_bwd_workspace_bytes(ctypes.byref(d))
    side = _side_stream(q.device)
    if fork is not None:
        side.wait_event(fork)
    else:
        side.wait_stream(torch.cuda.current_stream(q.device))
    from .runtime import ts_mark
    with torch.cuda.stream(side):
        ws2 = L.workspace(nbytes, q.device)
        keep_grid = d.table_grid
        d.table_grid = side_table_grid(q.device)  # the launch's own field: whole CUs stay with the main chain
        ts_mark("side: table kernel start")
        try:
            L.check(lib.vdetr_attn_bwd_table_f32(ctypes.byref(d), L.ptr(ds), L.ptr(dtable), L.ptr(ws2), nbytes, L.stream_ptr()),
                    "attn_bwd_table")
        finally:
            d.table_grid = keep_grid
        ts_mark("side: table kernel end")
    # EVERY tensor the descriptor points at stays alive until the join: the launch runs later than the caller's backward() returns,
    # and a block autograd frees then (the saved cos / sin of the rotated-box kind, the dropout state) is handed to the main stream's
    # next allocation while this kernel still reads it — `also` (found by the cut C5 case of test_full_config_training_step_vs_cpu_oracle)
    _side_keep.append((ds, dtable, ws2, aux, table, vertices, xyz, mask) + tuple(also))
    return dtable


def side_table_grid(device):
    """workgroups of a side-stream table-gradient launch: ASYNC_TABLE_GRID of 256 CUs, scaled to the device's count (even)"""
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    if not 2 <= ASYNC_TABLE_GRID < 256:
        return 0
    return max(2, (ASYNC_TABLE_GRID * cus // 256) & ~1)


def _fused_kv_ok(want_table):
    if not FUSED_KV_BWD:
        return False
    if want_table:  # the table gradient then reads the dS the fused kernel wrote, with the dynamic distribution's counters
        return DYNAMIC_BWD
    return True
_master = {}
_current = {}


def begin_step(device):
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _master:
        _master[key] = new_rng_state(device)
    snap = _master[key].clone()
    _master[key][1] += 1
    _current[key] = snap
    _step_side[key] = False
    _no_pending("begin_step")
    _kv_recs.pop(key, None)
    _kv_by_out.clear()
    _zero_pool[key] = None  # a fresh pool of zeros for this step's backward passes (allocated on first use)
    return snap


# Zero-initialised scratch of one step's backward passes (the table-gradient accumulators and the 4 counters of every
# layer): ONE fill per step instead of two per layer.  Slices are handed out once and never reused within the step.
_zero_pool = {}
_ZERO_POOL_FLOATS = 1 << 19


def _take_zeros(like, shape, dtype):
    numel = 1
    for d in shape:
        numel *= int(d)
    key = (like.device.type, like.device.index if like.device.index is not None else torch.cuda.current_device())
    if key in _zero_pool and numel * 4 <= _ZERO_POOL_FLOATS:
        pool = _zero_pool[key]
        if pool is None:
            pool = _zero_pool[key] = [torch.zeros(_ZERO_POOL_FLOATS, dtype=torch.float32, device=like.device), 0]
        buf, used = pool
        n4 = (numel + 3) // 4 * 4  # 16-byte aligned slices
        if used + n4 <= buf.numel():
            pool[1] = used + n4
            return buf[used:used + numel].view(dtype).view(shape)
    return torch.zeros(shape, dtype=dtype, device=like.device)


def current_rng(device):
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    return _current.get(key)


def reset_rng(seed=None):
    """Forget every per-device state (e.g. after torch.manual_seed)."""
    _master.clear()
    _current.clear()
    _zero_pool.clear()


def _desc(kind, B, H, nQ, nK, scale, table, rpe, vertices, xyz, cos_sin, mask, dropout_p, rng_state, salt=0,
          k_stride=0, v_stride=0):
    d = L.AttnDesc()
    d.k_row_stride, d.v_row_stride = int(k_stride), int(v_stride)
    d.kind, d.B, d.H, d.nQ, d.nK, d.scale = kind, B, H, nQ, nK, float(scale)
    if table is not None:
        d.table = table.data_ptr()
        d.table_size, d.log_scale, d.inv_log_norm = rpe.table_size, rpe.log_scale, rpe.inv_log_norm
        d.vertices, d.xyz = vertices.data_ptr(), xyz.data_ptr()
        d.cos_sin = cos_sin.data_ptr() if cos_sin is not None else None
    if mask is not None:
        d.mask = mask.data_ptr()
        d.mask_kind = L.VDETR_MASK_BOOL if mask.dtype == torch.uint8 else L.VDETR_MASK_FLOAT
    d.dropout_p = float(dropout_p)
    d.seed = int(salt) & 0xFFFFFFFFFFFFFFFF
    if rng_state is not None:
        d.rng_state = rng_state.data_ptr()
    d.fwd_kernel = FWD_KERNEL
    d.bwd_kernel = BWD_KERNEL
    d.table_grid = TABLE_GRID
    return d


def _prep_mask(mask, B, nQ, nK):
    if mask is None:
        return None
    if mask.dtype == torch.bool:
        mask = mask.to(torch.uint8)
    elif mask.dtype != torch.float32:
        mask = mask.float()
    return mask.expand(B, nQ, nK).contiguous()


def _check_inputs(**tensors):
    for name, t in tensors.items():
        if t is None:
            continue
        L.require_gpu(t, name)
        L.require_contiguous(t, name)
        L.require_float(t, name)


# Forward calls whose key-split merge was left to their consumer (vdetr_attn_fwd_parts_f32 -> rowblock._Ffn): data_ptr of `out` ->
# (parts, the workspace that holds the partials, lse, out).  `out` and `lse` are NOT valid until the consumer has run; the decoder
# layer's fused path is the only caller that asks for this, directly in front of rowblock.ffn.  An entry that nobody took is a bug
# (somebody would read an unwritten tensor): begin_step and the next deferred forward raise on it.
_pending_parts = {}
DEFER_COMBINE = os.environ.get("VDETR_DEFER_COMBINE", "1") != "0"


def take_pending_parts(out):
    """the record of a forward that left its merge to the consumer of `out` (None: `out` is final)"""
    return _pending_parts.pop(out.data_ptr(), None) if _pending_parts else None


def _no_pending(where):
    if _pending_parts:
        _pending_parts.clear()
        raise RuntimeError(f"{where}: an attention forward left its key-split merge to a consumer that never ran "
                           "(fused_attention(defer_combine=True) must be followed by rowblock.ffn on its output)")


# ---- round 6: the key-side pass of the backward without its operand-packing launch ---------------------------------------------------
# That launch (8-12 us, one per attention backward, on the backward's critical chain) does two things.  What depends on dO — its
# images, delta, max |dO row|^2 — is now left behind by the row-block kernel that PRODUCES dO (rowblock._Ffn / _ProjQ backward:
# vdetr_rb_*_bwd_emit_f32); what depends on the forward only — the q images, the zeroed dk / dv, bwd_aux words 1, 4, 5 — is done for all
# of a step's attention calls by ONE launch when the first of those backward kernels runs (kv_prepare).  A forward call that
# qualifies leaves a record; the consumer of its output takes the record in ITS forward (kv_record_of) and emits in its backward;
# the attention's backward then calls the pass alone (vdetr_attn_bwd_kv_packed_f32) — or, when nobody emitted, everything as before.
KV_PREPACK = os.environ.get("VDETR_KV_PREPACK", "1") != "0"
_kv_recs = {}    # device key -> this step's records (begin_step clears)
_kv_by_out = {}  # data_ptr of a call's output -> its record, until the output's consumer has taken it


class _KvRec:
    __slots__ = ("q", "v", "vertices", "cos_sin", "kind", "dims", "want_aux", "ws", "nbytes", "dkv", "delta", "aux", "prepared", "emitted")


def _kv_register(out, q, v, table, vertices, cos_sin, kind, B, H, nQ, nK):
    shared = kind == L.VDETR_ATTN_SHARED_KV
    if not (KV_PREPACK and FUSED_KV_BWD and B == 1 and H == 4 and nQ % 32 == 0 and q.dtype == torch.float32 and v.dtype == torch.float32
            and (shared or table is None) and v.stride(1) % 4 == 0 and v.data_ptr() % 16 == 0 and 4 * nQ * nK * 4 < (1 << 31)):
        return None
    r = _KvRec()
    r.q, r.v, r.vertices, r.cos_sin, r.kind, r.dims = q, v, (vertices if table is not None else None), cos_sin, kind, (B, H, nQ, nK)
    r.want_aux = bool(table is not None and table.requires_grad and DYNAMIC_BWD)
    r.prepared, r.emitted = False, None
    r.ws = r.dkv = r.delta = r.aux = None
    lst = _kv_recs.setdefault(_dev_key(q.device), [])
    if len(lst) >= 64:  # (steps that never call begin_step: forget what is stale)
        del lst[:]
        _kv_by_out.clear()
    lst.append(r)
    _kv_by_out[out.data_ptr()] = r
    return r


def kv_record_of(out):
    """the record of the attention call that produced `out`, for the module that consumes `out` (taken once)"""
    return _kv_by_out.pop(out.data_ptr(), None) if _kv_by_out else None


def kv_prepare(rec):
    """Called by the first backward kernel of a step that is about to emit: one launch (per 16 calls) leaves the q images, the
    zeroed dk / dv and the forward-only bwd_aux words of EVERY recorded call of the device."""
    if rec.prepared:
        return
    dev = rec.q.device
    lib = L.lib()
    todo = [r for r in _kv_recs.get(_dev_key(dev), []) if not r.prepared]
    if rec not in todo:
        todo.append(rec)
    for i in range(0, len(todo), 16):
        chunk = todo[i:i + 16]
        items = (L.AttnKvPrep * len(chunk))()
        for it, r in zip(items, chunk):
            B, H, nQ, nK = r.dims
            shared = r.kind == L.VDETR_ATTN_SHARED_KV
            dd = L.AttnDesc()
            dd.kind, dd.B, dd.H, dd.nQ, dd.nK = r.kind, B, H, nQ, nK
            r.nbytes = lib.vdetr_attn_bwd_kv_workspace_bytes(ctypes.byref(dd))
            r.ws = L.workspace(r.nbytes, dev)
            assert r.ws.data_ptr() % 256 == 0
            r.dkv = torch.empty((2, B, nK, 64 if shared else H * 64), dtype=torch.float32, device=dev)
            r.delta = torch.empty((B, nQ, H) if shared else (B, H, nQ), dtype=torch.float32, device=dev)
            r.aux = _take_zeros(r.q, (8,), torch.int32) if r.want_aux else None
            it.kind, it.B, it.H, it.nQ, it.nK, it.v_row_stride = r.kind, B, H, nQ, nK, r.v.stride(1)
            it.q, it.v = r.q.data_ptr(), r.v.data_ptr()
            it.vertices = r.vertices.data_ptr() if (r.vertices is not None and r.aux is not None) else None
            it.cos_sin = r.cos_sin.data_ptr() if (r.cos_sin is not None and it.vertices) else None
            it.workspace, it.dk, it.dv = r.ws.data_ptr(), r.dkv[0].data_ptr(), r.dkv[1].data_ptr()
            it.bwd_aux = r.aux.data_ptr() if r.aux is not None else None
        L.check(lib.vdetr_attn_bwd_kv_prep_f32(items, len(chunk), L.stream_ptr()), "attn_bwd_kv_prep")
        for r in chunk:
            r.prepared = True


class _FusedAttention(Function):
    @staticmethod
    def forward(ctx, q, k, v, table, vertices, xyz, cos_sin, mask, kind, H, scale, rpe, dropout_p, rng_state,
                need_grad, salt, table_async=False, kv_img=None, boxes=False, operand_bf16=False, defer_combine=False):
        B, nQ, C = q.shape
        nK = k.shape[1]
        assert C == H * HEAD_DIM, f"embed dim {C} != {H} heads x {HEAD_DIM}"
        bf16 = q.dtype == torch.bfloat16  # q, k, v stored as bf16: the bf16 matrix instructions, everything else fp32
        if bf16:
            if kind != L.VDETR_ATTN_SHARED_KV or k.dtype != torch.bfloat16 or v.dtype != torch.bfloat16:
                raise RuntimeError("fused_attention: the bf16 path takes bf16 q, k AND v of a shared-KV attention")
            _check_inputs(table=table, vertices=vertices, xyz=xyz, cos_sin=cos_sin)
            L.require_gpu(q, "q")
            L.require_contiguous(q, "q")
        else:
            _check_inputs(q=q, table=table, vertices=vertices, xyz=xyz, cos_sin=cos_sin)
        for name, t in (("k", k), ("v", v)):  # row-strided views are fine (see _kv_layout)
            L.require_gpu(t, name)
            if not bf16:
                L.require_float(t, name)
        ks, vs = k.stride(1), v.stride(1)
        lib = L.lib()
        use_drop = dropout_p > 0.0
        rng = rng_state if use_drop else None  # a per-step snapshot nobody writes again (begin_step)
        d = _desc(kind, B, H, nQ, nK, scale, table, rpe, vertices, xyz, cos_sin, mask, dropout_p if use_drop else 0.0, rng,
                  salt, ks, vs)
        out = torch.empty(q.shape, dtype=torch.float32, device=q.device)
        rows = (B, nQ, H) if kind == L.VDETR_ATTN_SHARED_KV else (B, H, nQ)
        lse = torch.empty(rows, dtype=torch.float32, device=q.device)
        scores = torch.empty(rows + (nK,), dtype=torch.float32, device=q.device) if need_grad else None
        if kind == L.VDETR_ATTN_PER_HEAD and _SELF_FWD8:
            d.fwd_kernel = 1  # (A/B: the eight-wave workgroups also where the four-wave form would be taken)
        elif kind == L.VDETR_ATTN_PER_HEAD and SELF_FWD_BODY:
            d.fwd_kernel = 4  # (A/B, parity tests: the general body also where the lean self-attention kernel would be taken)
        if operand_bf16 and not bf16 and FWD_KERNEL == 0:
            d.fwd_kernel = 3  # f32 tensors, q / k / v rounded to one bf16 part each inside the kernels (vdetr_hip.h)
        if kv_img is not None and not bf16 and FWD_KERNEL == 0:
            # this call's K / V operand images (pack_kv_images): packed for THIS part count and THESE sizes, or the kernel reads
            # a one-part image as three parts (or runs off its end) without any error
            if not isinstance(kv_img, KVImage):
                raise TypeError("fused_attention: kv_img must be an item of pack_kv_images()")
            want = 1 if operand_bf16 else 3
            if (kv_img.parts, kv_img.B, kv_img.nK) != (want, B, nK) or kv_img.data.device != q.device:
                raise RuntimeError(f"fused_attention: kv_img was packed for parts={kv_img.parts}, B={kv_img.B}, nK={kv_img.nK} on "
                                   f"{kv_img.data.device}; this call needs parts={want}, B={B}, nK={nK} on {q.device}")
            d.kv_img = kv_img.data.data_ptr()
        sched = None
        if table is not None:
            # the persistent forward's item counter: a zero word of THIS launch's own (a slice of the step's zero pool, which a
            # captured step re-zeroes at every replay) — a word shared by a stream's launches is baked into every graph captured
            # there, and two such graphs replayed side by side, or one launch that aborts, corrupt each other's item counts
            sched = _take_zeros(q, (4,), torch.int32)
            d.fwd_sched = sched.data_ptr()
        nbytes = lib.vdetr_attn_fwd_workspace_bytes(ctypes.byref(d))
        ws = L.workspace(nbytes, q.device) if nbytes else None
        if defer_combine and DEFER_COMBINE and not bf16 and kind == L.VDETR_ATTN_SHARED_KV and H == 4:
            _no_pending("fused_attention")
            parts = L.AttnParts()
            L.check(lib.vdetr_attn_fwd_parts_f32(ctypes.byref(d), L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(out), L.ptr(lse), L.ptr(scores),
                                                 L.ptr(ws), nbytes, ctypes.byref(parts), L.stream_ptr()), "attn_fwd_parts")
            if 2 <= parts.ksplit <= 16:
                _pending_parts[out.data_ptr()] = (parts, ws, lse, out)
            elif parts.ksplit != 1:
                raise RuntimeError(f"fused_attention: {parts.ksplit} key chunks left unmerged and no consumer takes that many")
        else:
            fwd = lib.vdetr_attn_fwd_bf16 if bf16 else lib.vdetr_attn_fwd_f32
            L.check(fwd(ctypes.byref(d), L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(out), L.ptr(lse),
                        L.ptr(scores), L.ptr(ws), nbytes, L.stream_ptr()), "attn_fwd")
        if need_grad:
            ctx.save_for_backward(q, k, v, table, vertices, xyz, cos_sin, mask, out, lse, scores, rng)
            ctx.cfg = (kind, H, scale, rpe, dropout_p if use_drop else 0.0, salt)
            ctx.kv_rec = None if (bf16 or mask is not None) else _kv_register(out, q, v, table, vertices, cos_sin, kind, B, H, nQ, nK)
            ctx.table_async = bool(table_async)
            ctx.boxes = bool(boxes)
            if table_async and table is not None and table.requires_grad and _async_wanted(q.shape[0], q.shape[1], k.shape[1]):
                _step_side[_dev_key(q.device)] = True
        if _kv_by_out and getattr(ctx, "kv_rec", None) is None:
            _kv_by_out.pop(out.data_ptr(), None)  # (a record an earlier call left under this address, never taken: not this output's)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, table, vertices, xyz, cos_sin, mask, out, lse, scores, rng = ctx.saved_tensors
        kind, H, scale, rpe, dropout_p, salt = ctx.cfg
        in_dtype = q.dtype
        if in_dtype == torch.bfloat16:  # the gradient GEMMs run in fp32 on the stored bf16 operands (1 MB each): what is
            q, k, v = q.float(), k.float().contiguous(), v.float().contiguous()  # 67 MB per layer stays fp32 anyway
        B, nQ, C = q.shape
        nK = k.shape[1]
        lib = L.lib()
        dout = dout.contiguous()
        shared = kind == L.VDETR_ATTN_SHARED_KV
        want_table = table is not None and ctx.needs_input_grad[3]
        d = _desc(kind, B, H, nQ, nK, scale, table, rpe, vertices, xyz, cos_sin, mask, dropout_p, rng, salt, 0, v.stride(1))
        if getattr(ctx, "boxes", False) and BWD_KERNEL == 0 and cos_sin is None:
            # the caller vouches for axis-aligned boxes (vertices out of the box decode: exact coordinate patterns): the box kernel
            # alone, without the general kernel's launch in front of it; a query that is not a box poisons dtable with NaN
            d.bwd_kernel = 2
        if want_table and DYNAMIC_BWD:  # norm maxima + query counters for the dynamic distribution (see vdetr_hip.h)
            aux = _take_zeros(q, (8,), torch.int32)  # 5 words used (vdetr_hip.h: bwd_aux)
            d.bwd_aux = aux.data_ptr()
        delta = torch.empty((B, nQ, H) if shared else (B, H, nQ), dtype=torch.float32, device=q.device)
        # the fused key-side pass where it is built AND inside its limits (attn_bwd_kv.hip:kv_run: a score matrix below 2 GB,
        # at most 65535 of them, 16-B aligned operands); everything else takes the library-GEMM composition as before
        rows_kv, nprob_kv = (4 * nQ, B) if shared else (nQ, B * H)
        fused = (_fused_kv_ok(want_table) and ((shared and H == 4) or (not shared and not want_table))
                 and rows_kv * nK * 4 < (1 << 31) and nprob_kv <= 65535 and v.stride(1) % 4 == 0
                 and all(t.data_ptr() % 16 == 0 for t in (v, lse)))
        if not fused:  # (the fused pass computes delta in the first workgroups of its operand-packing launch)
            L.check(lib.vdetr_attn_delta_f32(ctypes.byref(d), L.ptr(dout), L.ptr(out), L.ptr(v), L.ptr(delta), L.stream_ptr()),
                    "attn_delta")
        if fused:
            # one pass over the scores: dP~ = dO V^T, softmax / dropout backward, dV, dK; dS (unscaled) comes back for the
            # table gradient and the dQ GEMM (attn_bwd_kv.hip)
            ds = torch.empty_like(scores)  # [B, nQ, H, nK] (shared K/V) / [B, H, nQ, nK] (per head)
            dkv = torch.empty((2, B, nK, k.shape[2]), dtype=torch.float32, device=q.device)
            run_async = want_table and ctx.table_async and _async_wanted(B, nQ, nK)
            # 4 waves fit NEXT TO a table kernel that holds every CU; with CUs left free for the main chain the default shape
            d.kv_waves = 4 if (run_async or _side_keep) and (ASYNC_TABLE_GRID >= 256 or _ASYNC_KV4) else 8
            # the query self-attention's pass (256 one-per-CU workgroups at the model's size) next to a live table kernel: one
            # workgroup per key tile
            d.kv_halves = 1 if (_side_keep and (KV_ONE_WG >= 2 or (KV_ONE_WG == 1 and not shared))) else 2
            rec = getattr(ctx, "kv_rec", None)
            if (rec is not None and rec.prepared and rec.emitted == dout.data_ptr() and (rec.aux is not None) == bool(want_table and DYNAMIC_BWD)
                    and in_dtype == torch.float32):
                # the producer of dout left its images and delta behind, the step's prep launch everything else: the pass alone
                delta, dkv, aux = rec.delta, rec.dkv, rec.aux
                d.bwd_aux = aux.data_ptr() if aux is not None else None
                L.check(lib.vdetr_attn_bwd_kv_packed_f32(ctypes.byref(d), L.ptr(q), L.ptr(v), L.ptr(dout), L.ptr(scores), L.ptr(lse),
                                                         L.ptr(delta), L.ptr(ds), L.ptr(dkv[0]), L.ptr(dkv[1]), L.ptr(rec.ws), rec.nbytes,
                                                         L.stream_ptr()), "attn_bwd_kv_packed")
            else:
                nbytes = lib.vdetr_attn_bwd_kv_workspace_bytes(ctypes.byref(d))
                ws = L.workspace(nbytes, q.device)
                L.check(lib.vdetr_attn_bwd_kv_delta_f32(ctypes.byref(d), L.ptr(q), L.ptr(v), L.ptr(dout), L.ptr(out), L.ptr(scores),
                                                        L.ptr(lse), L.ptr(delta), L.ptr(ds), L.ptr(dkv[0]), L.ptr(dkv[1]), L.ptr(ws),
                                                        nbytes, L.stream_ptr()), "attn_bwd_kv")
            from .runtime import ts_mark
            ts_mark("main: key-side pass done" if shared else "main: self-attention key-side pass done")
            dtable = fork = None
            if run_async:
                # The side stream's launches depend on the key-side pass only, but are CAPTURED behind the main chain's next
                # kernel: a captured graph's executor keeps the FIRST edge of a node on the node's queue and moves every
                # further edge to the next one (round robin over 4) - the main chain must stay where it is, or it hops a
                # queue per layer and lands behind the table kernel every fourth time (measured: DESIGN.md 4.4e)
                dtable = _table_accumulator(table)  # (a fill, if any, is in front of the event)
                fork = torch.cuda.Event()
                fork.record(torch.cuda.current_stream(q.device))
            # dQ = scale dS K: the row-owner kernel (attn_bwd_dq.hip; exact fp32 products) for per-head K/V where K is as the
            # kernel reads it — rows of contiguous floats at a constant stride —, the library's batched GEMM otherwise
            kd = k if shared else k.reshape(B, nK, C)
            if (_DQ_KERNEL and (not shared or _DQ_KERNEL_SHARED) and k.dtype == torch.float32 and ds.dtype == torch.float32 and kd.stride(2) == 1
                    and kd.stride(1) % 4 == 0 and (B == 1 or kd.stride(0) == nK * kd.stride(1))
                    and kd.data_ptr() % 16 == 0 and ds.data_ptr() % 16 == 0 and B * (1 if shared else H) <= 65535):
                dq = q.new_empty((B, nQ, C))
                keep = d.k_row_stride
                d.k_row_stride = kd.stride(1)
                try:
                    L.check(lib.vdetr_attn_bwd_dq_f32(ctypes.byref(d), L.ptr(ds), L.ptr(kd), L.ptr(dq), L.stream_ptr()), "attn_bwd_dq")
                finally:
                    d.k_row_stride = keep
            elif shared:
                dq = q.new_empty((B, nQ * H, HEAD_DIM))
                torch.baddbmm(dq, ds.view(B, nQ * H, nK), k, beta=0.0, alpha=float(scale), out=dq)
                dq = dq.view(B, nQ, C)
            else:
                ds_r = ds.view(B * H, nQ, nK)
                k4 = k.reshape(B, nK, H, HEAD_DIM).permute(0, 2, 1, 3)
                if B == 1:  # the GEMM reads the per-head K views and writes the [nQ, H*64] layout in place
                    dq = q.new_empty((1, nQ, C))
                    dq_r = dq.view(nQ, H, HEAD_DIM).permute(1, 0, 2)
                    torch.baddbmm(dq_r, ds_r, k4[0], beta=0.0, alpha=float(scale), out=dq_r)
                else:
                    dq = torch.bmm(ds_r, k4.reshape(B * H, nK, HEAD_DIM)).mul_(float(scale))
                    dq = dq.view(B, H, nQ, HEAD_DIM).permute(0, 2, 1, 3).reshape(B, nQ, C)
            if want_table:
                if run_async:
                    dtable = _launch_table_async(lib, d, q, ds, table, aux, vertices, xyz, mask, fork, dtable, also=(cos_sin, rng))
                else:
                    dtable = _table_accumulator(table)
                    nbytes = lib.vdetr_attn_bwd_workspace_bytes(ctypes.byref(d))
                    ws = L.workspace(nbytes, q.device)
                    L.check(lib.vdetr_attn_bwd_table_f32(ctypes.byref(d), L.ptr(ds), L.ptr(dtable), L.ptr(ws), nbytes,
                                                         L.stream_ptr()), "attn_bwd_table")
            dk, dv = dkv[0], dkv[1]
            if in_dtype == torch.bfloat16:
                dq, dk, dv = dq.to(in_dtype), dk.to(in_dtype), dv.to(in_dtype)
            return (dq, dk, dv, dtable) + (None,) * 17
        if shared:
            # rows (b, q, h): [B, nQ*H, 64] views, K/V [B, nK, 64]
            do_r = dout.view(B, nQ * H, HEAD_DIM)
            dprob = torch.bmm(do_r, v.transpose(1, 2))  # [B, nQ*H, nK]
        else:
            # rows (b, h, q): [B*H, nQ, 64]; K/V [B*H, nK, 64].  For one scene the per-head operands are plain strided
            # views [H, L, 64] (row stride H*64) that the batched GEMM reads in place; B > 1 needs the copies.
            def heads(t, n):
                t4 = t.reshape(B, n, H, HEAD_DIM).permute(0, 2, 1, 3)
                return t4[0] if B == 1 else t4.reshape(B * H, n, HEAD_DIM)
            do_r, v_r = heads(dout, nQ), heads(v, nK)
            dprob = torch.bmm(do_r, v_r.transpose(1, 2))  # [B*H, nQ, nK]
        dtable = _take_zeros(table, tuple(table.shape), table.dtype) if want_table else None
        nbytes = lib.vdetr_attn_bwd_workspace_bytes(ctypes.byref(d)) if want_table else 0
        ws = L.workspace(nbytes, q.device) if nbytes else None
        # scores -> P~ (dropped probabilities), dprob -> dS.  In place where allowed; the table-gradient kernel
        # re-reads its inputs from several waves and needs separate outputs.
        probs = torch.empty_like(scores) if want_table else scores
        dscore = torch.empty_like(dprob) if want_table else dprob
        L.check(lib.vdetr_attn_bwd_scores_f32(ctypes.byref(d), L.ptr(scores), L.ptr(dprob), L.ptr(lse), L.ptr(delta),
                                              L.ptr(probs), L.ptr(dscore), L.ptr(dtable), L.ptr(ws), nbytes,
                                              L.stream_ptr()), "attn_bwd_scores")
        scores, dprob = probs, dscore
        if shared:
            p_r = scores.view(B, nQ * H, nK)
            ds_r = dprob
            q_r = q.view(B, nQ * H, HEAD_DIM)
            dv = torch.bmm(p_r.transpose(1, 2), do_r)
            dk = torch.bmm(ds_r.transpose(1, 2), q_r)  # ds_out already carries the q-scale
            dq = torch.bmm(ds_r, k).view(B, nQ, C)
        else:
            p_r = scores.view(B * H, nQ, nK)
            ds_r = dprob
            q_r, k_r = heads(q, nQ), heads(k, nK)
            if B == 1:  # the GEMMs write straight into the [L, H*64] layout (row stride H*64): no permute copies
                dv, dk, dq = q.new_empty((1, nK, C)), q.new_empty((1, nK, C)), q.new_empty((1, nQ, C))
                torch.bmm(p_r.transpose(1, 2), do_r, out=dv.view(nK, H, HEAD_DIM).permute(1, 0, 2))
                torch.bmm(ds_r.transpose(1, 2), q_r, out=dk.view(nK, H, HEAD_DIM).permute(1, 0, 2))
                torch.bmm(ds_r, k_r, out=dq.view(nQ, H, HEAD_DIM).permute(1, 0, 2))
            else:
                dv = torch.bmm(p_r.transpose(1, 2), do_r).view(B, H, nK, HEAD_DIM).permute(0, 2, 1, 3).reshape(B, nK, C)
                dk = torch.bmm(ds_r.transpose(1, 2), q_r).view(B, H, nK, HEAD_DIM).permute(0, 2, 1, 3).reshape(B, nK, C)
                dq = torch.bmm(ds_r, k_r).view(B, H, nQ, HEAD_DIM).permute(0, 2, 1, 3).reshape(B, nQ, C)
        if in_dtype == torch.bfloat16:
            dq, dk, dv = dq.to(in_dtype), dk.to(in_dtype), dv.to(in_dtype)
        return (dq, dk, dv, dtable) + (None,) * 17


def _kv_layout(t, B, nK):
    """K / V as the kernel can read them: rows of contiguous floats at a constant row stride (a multiple of 4 floats,
    16-B aligned base, batches nK rows apart) — e.g. a 64-wide column block of a wider projection output.  Anything
    else is made contiguous."""
    mult = 8 if t.dtype == torch.bfloat16 else 4
    ok = (t.dim() == 3 and t.stride(2) == 1 and t.stride(1) % mult == 0 and t.stride(1) >= t.shape[2] and
          t.data_ptr() % 16 == 0 and (B == 1 or t.stride(0) == nK * t.stride(1)))
    return t if ok else t.contiguous()


def fused_attention(q, k, v, *, num_heads, scale, shared_kv, table=None, rpe=None, vertices=None, xyz=None,
                    cos_sin=None, attn_mask=None, dropout_p=0.0, rng_state=None, salt=0, table_grad_async=False, kv_img=None,
                    vertices_are_boxes=False, operand_bf16=False, defer_combine=False):
    """out[B,nQ,H*64] = dropout(softmax(scale * q k^T + rpe + mask)) v.
    kv_img: this call's slice of pack_kv_images() (optional: the forward packs its own otherwise).
    operand_bf16: f32 q / k / v whose values are rounded to bf16 on the way into QK^T and PV (BASELINE config 4's arithmetic
    without bf16 tensors: no cast launches, the backward as for f32 operands — the rounding's derivative is the identity);
    pack_kv_images(parts=1) for kv_img.
    vertices_are_boxes: the caller vouches that every query's 8 vertices are an axis-aligned box (they come out of a box decode):
    the table gradient then launches its box kernel alone (vdetr_attn_desc.bwd_kernel = 2).
    defer_combine: the caller hands the result to rowblock.ffn NEXT and to nothing else: where the forward splits the keys, the merge
    of the partial results is done by that launch on its way in (vdetr_attn_fwd_parts_f32; the returned tensor is not valid before).

    q [B,nQ,H*64]; k,v [B,nK,64] (shared_kv) or [B,nK,H*64]; table [8,T,T,T,H]; vertices [B,nQ,8,3];
    xyz [B,nK,3]; cos_sin [B,nQ,2] or None; attn_mask [B,nQ,nK] bool (-100 fill) / float (additive) or None.
    """
    B, nQ, _ = q.shape
    nK = k.shape[1]
    kind = L.VDETR_ATTN_SHARED_KV if shared_kv else L.VDETR_ATTN_PER_HEAD
    if dropout_p > 0.0 and rng_state is None:
        rng_state = current_rng(q.device)
        if rng_state is None:
            rng_state = begin_step(q.device)
    mask = _prep_mask(attn_mask, B, nQ, nK)
    need_grad = torch.is_grad_enabled() and any(
        t is not None and t.requires_grad for t in (q, k, v, table))
    if table is not None:
        table = table.contiguous()
        vertices = vertices.detach().contiguous()
        xyz = xyz.detach().contiguous()
        if cos_sin is not None:
            cos_sin = cos_sin.detach().contiguous()
    k, v = _kv_layout(k, B, nK), _kv_layout(v, B, nK)
    return _FusedAttention.apply(q.contiguous(), k, v, table, vertices, xyz, cos_sin, mask,
                                 kind, num_heads, float(scale), rpe, float(dropout_p), rng_state, need_grad, int(salt),
                                 bool(table_grad_async), kv_img, bool(vertices_are_boxes), bool(operand_bf16), bool(defer_combine))


class KVImage:
    """one attention call's K / V operand image (a row of pack_kv_images' buffer) with what it was packed for"""
    __slots__ = ("data", "parts", "B", "nK")

    def __init__(self, data, parts, B, nK):
        self.data, self.parts, self.B, self.nK = data, int(parts), int(B), int(nK)


def pack_kv_images(kv, n, parts=3):
    """kv [B, nK, n * 128] f32, the joint K | V projection of n cross-attention layers (layer i: columns 128 i .. + 63 = K,
    + 64 .. + 127 = V) -> n KVImage items (rows of one uint8 [n, bytes] buffer): each layer's operand images for the persistent forward
    (fused_attention(kv_img=)), one launch for all layers.  parts: 3 = f32 accuracy (the default forward), 1 = operands rounded to bf16
    (fused_attention(operand_bf16=True)).  None where the forward would not use them."""
    if (FWD_KERNEL != 0 or not kv.is_cuda or kv.dtype != torch.float32 or not kv.is_contiguous() or kv.shape[2] != n * 2 * HEAD_DIM
            or kv.data_ptr() % 16):
        return None
    B, nK = kv.shape[0], kv.shape[1]
    lib = L.lib()
    nbytes = lib.vdetr_attn_kv_image_parts_bytes(B, nK, parts)
    img = torch.empty((n, nbytes), dtype=torch.uint8, device=kv.device)
    L.check(lib.vdetr_attn_pack_kv_parts_f32(kv.data_ptr(), kv.data_ptr() + 4 * HEAD_DIM, B, nK, kv.shape[2], kv.shape[2], n, 2 * HEAD_DIM,
                                             parts, img.data_ptr(), L.stream_ptr()), "attn_pack_kv")
    return [KVImage(img[i], parts, B, nK) for i in range(n)]


def attention_probabilities(q, k, *, num_heads, scale, shared_kv, table=None, rpe=None, vertices=None, xyz=None,
                            cos_sin=None, attn_mask=None, dropout_p=0.0, rng_state=None, salt=0):
    """The ``attn`` tensor the reference modules return next to the output (vdetr_transformer.py:751-752,758):
    dropout(softmax(scores)) as [B,H,nQ,nK].  Only materialised on request (it is 64 MB per layer at full size).
    ``rng_state`` must be the state the matching forward call used."""
    B, nQ, C = q.shape
    nK = k.shape[1]
    kind = L.VDETR_ATTN_SHARED_KV if shared_kv else L.VDETR_ATTN_PER_HEAD
    mask = _prep_mask(attn_mask, B, nQ, nK)
    lib = L.lib()
    with torch.no_grad():
        q, k = q.detach().contiguous(), k.detach().contiguous()
        if table is not None:
            table, vertices, xyz = table.detach().contiguous(), vertices.detach().contiguous(), xyz.detach().contiguous()
            cos_sin = cos_sin.detach().contiguous() if cos_sin is not None else None
        d = _desc(kind, B, num_heads, nQ, nK, scale, table, rpe, vertices, xyz, cos_sin, mask, dropout_p, rng_state, salt)
        out = torch.empty_like(q)
        rows = (B, nQ, num_heads) if shared_kv else (B, num_heads, nQ)
        lse = torch.empty(rows, dtype=torch.float32, device=q.device)
        scores = torch.empty(rows + (nK,), dtype=torch.float32, device=q.device)
        # V is irrelevant for the probabilities; K doubles as V to keep the call well-formed
        nbytes = lib.vdetr_attn_fwd_workspace_bytes(ctypes.byref(d))
        ws = L.workspace(nbytes, q.device) if nbytes else None
        L.check(lib.vdetr_attn_fwd_f32(ctypes.byref(d), L.ptr(q), L.ptr(k), L.ptr(k), L.ptr(out), L.ptr(lse),
                                       L.ptr(scores), L.ptr(ws), nbytes, L.stream_ptr()), "attn_fwd")
        L.check(lib.vdetr_attn_bwd_scores_f32(ctypes.byref(d), L.ptr(scores), None, L.ptr(lse), None, L.ptr(scores),
                                              None, None, None, 0, L.stream_ptr()), "attn_probs")
    return scores.permute(0, 2, 1, 3) if shared_kv else scores


def rpe_bias(table, vertices, xyz, rpe, cos_sin=None):
    """rpe[B,H,nQ,nK] of vdetr_transformer.py:710-731 alone (parity hook; the fused kernel never stores it)."""
    _check_inputs(table=table, vertices=vertices, xyz=xyz, cos_sin=cos_sin)
    B, nQ = vertices.shape[:2]
    nK = xyz.shape[1]
    H = table.shape[-1]
    d = _desc(L.VDETR_ATTN_SHARED_KV, B, H, nQ, nK, 1.0, table, rpe, vertices, xyz, cos_sin, None, 0.0, None)
    out = torch.empty((B, nQ, H, nK), dtype=torch.float32, device=table.device)
    L.check(L.lib().vdetr_rpe_bias_f32(ctypes.byref(d), L.ptr(out), L.stream_ptr()), "rpe_bias")
    return out.permute(0, 2, 1, 3)


def dropout_keep_mask(B, H, nQ, nK, shared_kv, dropout_p, rng_state, salt=0):
    """uint8 keep-mask the kernels draw for (dropout_p, rng_state): [B,H,nQ,nK].  Test hook."""
    kind = L.VDETR_ATTN_SHARED_KV if shared_kv else L.VDETR_ATTN_PER_HEAD
    d = _desc(kind, B, H, nQ, nK, 1.0, None, None, None, None, None, None, dropout_p, rng_state, salt)
    rows = (B, nQ, H) if shared_kv else (B, H, nQ)
    keep = torch.empty(rows + (nK,), dtype=torch.uint8, device=rng_state.device)
    L.check(L.lib().vdetr_attn_dropout_mask_u8(ctypes.byref(d), L.ptr(keep), L.stream_ptr()), "dropout_mask")
    return keep.permute(0, 2, 1, 3) if shared_kv else keep
