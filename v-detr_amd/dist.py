"""Scene-sharded data parallelism over RCCL/xGMI: one process per GPU, gradients only.

Replaces the reference's ``DistributedDataParallel(model, find_unused_parameters=True)`` wrap (main.py:515-517) and
its process-group setup (utils/dist.py:51-64).  The path shards by independent scenes (DistributedSampler,
main.py:531-532), so the only data-path collective is the gradient average:

  * all gradients live in a few contiguous fp32 BUCKETS (``p.grad`` are views into them) — fewer, larger
    all-reduces suit xGMI's point-to-point links (7 x ~153 GB/s per GPU; a ring is bound by ONE link, so message
    count and size matter more than on a switched fabric);
  * eager mode: post-accumulate hooks count the parameters of a bucket; when a bucket is complete its all-reduce
    is issued on a SIDE stream (ordered after the producing kernels by an event), overlapping the rest of
    backward; ``finish()`` joins the side stream and flushes buckets whose parameters got no gradient
    (the reference needs find_unused_parameters for those);
  * graph mode: forward+backward run as one captured hipGraph (no hooks fire inside a replay), so
    ``reduce_all()`` is called after the replay.
``bucket_views=False`` keeps ``p.grad`` ordinary tensors (autograd then hands its gradient buffers over without
the per-parameter accumulate kernel that writing into a pre-existing ``.grad`` costs — ~1100 launches per step
for this model) and packs them into the buckets with one multi-tensor copy before the all-reduce
(``pack_and_reduce()``), pointing ``p.grad`` at the reduced views afterwards.
On CPU tensors (gloo, used by the tests) the same code runs without streams.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torchrun convention).  Returns
    (rank, local_rank, world_size).  backend "nccl" is RCCL on ROCm."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        _AVG_VERDICT.clear()  # verdicts of an earlier process group of this process do not carry over
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)  # no rank -> device guessing in barrier(), eager communicator
        try:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
        except TypeError:  # a torch without the device_id argument
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


class FlatParams:
    """All trainable parameters as views of ONE contiguous fp32 buffer, their gradients packed into a second one.

    Sized for 288 GB of HBM rather than for launch count: the optimizer then updates a single 11.7 M-element tensor
    (one fused AdamW launch instead of ~30 multi-tensor launches over 796 tensors), gradient clipping is one norm over
    one tensor and rides into the optimizer launch as its ``grad_scale``, and the all-reduce buckets are plain slices of
    the gradient buffer.  ``p.data`` of every parameter is re-pointed at its slice (state_dict / load_state_dict keep
    working, they copy in place); calling ``module.to()`` afterwards would detach the views again.
    Layout = the model's adjacency groups first, then the rest: shape by shape (``group_shapes=True``, for loops that park
    their weight gradients) or in plain REVERSE parameter order (``group_shapes=False``: the hooked eager path, whose slices
    then complete roughly front-to-back during backward)."""

    def __init__(self, params, groups=None, first=None, group_shapes=True):
        """``groups``: optional list of (parameters, slot) — parameters a model wants ADJACENT in memory, in this order
        (see helpers.cat_params).  slot = None packs them back to back; slot = n gives every parameter a zero-padded
        slab of n elements (helpers.slot_stack_params).  Everything else follows in reverse parameter order.
        ``first``: parameters to lay out before all others of the ungrouped rest (e.g. the part of a model whose gradients
        are final first, so that its gradient buckets are contiguous and can be all-reduced while the rest is still in
        its backward pass).  ``group_shapes`` (default): the ungrouped matrices are laid out shape by shape — for loops that
        park their weight gradients and compute them per shape in batched GEMMs (runtime.defer_weight_grads, reduce_phased);
        False keeps plain REVERSE parameter order, the layout for the hooked eager path (GradientReducer with
        bucket_views=True), whose buckets then complete roughly back to front while the backward pass is still running;
        "first" groups the shapes of the ``first`` part only (a captured decoder in front of an eager backbone).
        (A flat optimizer state saved under one layout does not line up with the other: persist it per parameter —
        ``state_dict_per_parameter`` — not as the flat tensor.)"""
        plist = [p for p in params if p.requires_grad]
        first_ids = {id(p) for p in (first or [])}
        assert plist, "no trainable parameters"
        known = {id(p) for p in plist}
        layout, placed, off = [], set(), 0  # (param, offset)
        for plist_g, slot in (groups or []):
            plist_g = [p for p in plist_g if id(p) in known and id(p) not in placed]
            if not plist_g:
                continue
            off = (off + 63) // 64 * 64
            for p in plist_g:
                assert slot is None or p.numel() <= slot, "slot smaller than a parameter of its group"
                layout.append((p, off))
                placed.add(id(p))
                off += slot if slot is not None else p.numel()
        # the rest in reverse parameter order, matrices of one shape next to each other: the parked weight gradients are
        # computed per shape as ONE batched GEMM (helpers.DeferredParamGrads), and a gradient bucket that holds whole shape
        # groups lets reduce_phased() finish and send it without splitting such a batch
        rest = [p for p in reversed(plist) if id(p) not in placed]
        order = {}
        def shape_key(p):  # [k*C, C] stacks go with [C, C]; an empty matrix is its own kind
            return (p.shape[1], p.shape[1] > 0 and p.shape[0] % p.shape[1] == 0)
        if group_shapes:  # ("first": only the `first` part is shape-grouped, the rest keeps reverse parameter order)
            grouped = (lambda p: id(p) in first_ids) if group_shapes == "first" else (lambda p: True)
            for p in rest:
                if p.ndim == 2 and grouped(p):
                    order.setdefault(shape_key(p), len(order))
            if group_shapes == "first":
                rest = sorted((p for p in rest if grouped(p)), key=lambda p: order[shape_key(p)] if p.ndim == 2 else len(order)) + \
                    [p for p in rest if not grouped(p)]
            else:
                rest.sort(key=lambda p: order[shape_key(p)] if p.ndim == 2 else len(order))  # stable
        rest.sort(key=lambda p: 0 if id(p) in first_ids else 1)  # stable: keeps the shape runs inside each part
        for p in rest:
            if id(p) not in placed:
                off = (off + 3) // 4 * 4  # 16-B aligned slices (vectorised pack / fused optimizer)
                layout.append((p, off))
                off += p.numel()
        self.params = [p for p, _ in layout]
        dev, dtype = self.params[0].device, self.params[0].dtype
        total = (off + 3) // 4 * 4
        self.data = torch.zeros(total, dtype=dtype, device=dev)   # padding stays zero: zero gradient, zero decay
        self.grad = torch.zeros(total, dtype=dtype, device=dev)
        self.grad_views, self.offsets = [], {}
        with torch.no_grad():
            for p, o in layout:
                assert p.device == dev and p.dtype == dtype, "one device / dtype per flat buffer"
                n = p.numel()
                self.data[o:o + n].copy_(p.data.reshape(-1))
                p.data = self.data[o:o + n].view(p.shape)
                self.grad_views.append(self.grad[o:o + n].view(p.shape))
                self.offsets[id(p)] = o
        self.param = torch.nn.Parameter(self.data)  # what the optimizer sees; shares the storage
        self.param.grad = self.grad

    def pack_grads_span(self, start, end):
        """``pack_grads`` for the parameters whose slice lies in [start, end) of the flat buffer only (one gradient bucket
        of a data-parallel step: its parameters' ``.grad`` are final, the others' may still be missing)."""
        sel = [i for i, p in enumerate(self.params) if start <= self.offsets[id(p)] < end]
        src = [self.params[i].grad for i in sel]
        if self.grad.is_cuda:
            return self._pack_hip(src, span=(start, end), sel=sel)
        views = [self.grad_views[i] for i in sel]
        missing = [v for v, g in zip(views, src) if g is None]
        if missing:
            torch._foreach_zero_(missing)
        have = [(v, g) for v, g in zip(views, src) if g is not None]
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])

    def pack_grads(self, grads=None):
        """``p.grad`` (or the given tensors, in ``self.params`` order) -> the flat gradient buffer; parameters that
        received no gradient contribute zeros.  GPU: ONE launch of ``vdetr_pack_f32`` (the source-pointer table goes up
        through pinned memory; inside a hipGraph capture that upload is a captured copy node and the captured gradient
        buffers are static).  CPU tensors (gloo tests of the host logic): multi-tensor copy."""
        src = [p.grad for p in self.params] if grads is None else list(grads)
        if self.grad.is_cuda:
            return self._pack_hip(src)
        have_v = [v for v, g in zip(self.grad_views, src) if g is not None]
        have_g = [g for g in src if g is not None]
        missing = [v for v, g in zip(self.grad_views, src) if g is None]
        if missing:
            torch._foreach_zero_(missing)
        if have_g:
            torch._foreach_copy_(have_v, have_g)

    def _pack_tables(self, sel=None):
        from . import _lib as L
        chunk = L.lib().vdetr_pack_chunk_floats()
        params = self.params if sel is None else [self.params[i] for i in sel]
        n = len(params)
        tab = torch.zeros((n, 3), dtype=torch.int64)
        be, bc = [], []
        for i, p in enumerate(params):
            tab[i, 1], tab[i, 2] = self.offsets[id(p)], p.numel()
            nch = (p.numel() + chunk - 1) // chunk
            be += [i] * nch
            bc += list(range(nch))
        dev = self.grad.device
        return {"static": tab, "block_entry": torch.tensor(be, dtype=torch.int32, device=dev),
                      "block_chunk": torch.tensor(bc, dtype=torch.int32, device=dev), "nblocks": len(be),
                      "dev": torch.empty((n, 3), dtype=torch.int64, device=dev),
                      # eager launches: a ring of pinned tables, so that the host only waits for the upload of THREE steps ago
                      # (one table = the launching thread can never be more than one step ahead of the device)
                      "hosts": [tab.clone().pin_memory() for _ in range(3)], "events": [None, None, None], "tick": 0,
                      "captured": [],
                      # pinned tables for hipGraph captures, allocated up front (no host allocation while capturing)
                      "spare": [tab.clone().pin_memory() for _ in range(4)]}

    def _pack_hip(self, src, span=None, sel=None):
        from . import _lib as L
        if span is None:
            if not hasattr(self, "_pack"):
                self._pack = self._pack_tables()
            P = self._pack
        else:  # one table set per gradient bucket
            spans = self.__dict__.setdefault("_pack_spans", {})
            if span not in spans:
                spans[span] = self._pack_tables(sel)
            P = spans[span]
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing:  # a captured upload reads ITS host table at every replay: never reuse it
            assert P["spare"], "FlatParams: more than 4 graph captures of pack_grads (call it once eagerly first)"
            host = P["spare"].pop()
            P["captured"].append(host)
        else:
            slot = P["tick"] % 3
            P["tick"] += 1
            host = P["hosts"][slot]
            if P["events"][slot] is not None:
                P["events"][slot].synchronize()  # the upload that last used this pinned table has left it
        keep, ptrs = [], []
        for g in src:
            if g is None:
                ptrs.append(0)
                continue
            if g.dtype != self.grad.dtype or not g.is_contiguous():
                g = g.to(self.grad.dtype).contiguous()
                keep.append(g)
            ptrs.append(g.data_ptr())
        host[:, 0] = torch.tensor(ptrs, dtype=torch.int64)  # one strided copy (an element assignment per tensor costs ~1 us each)
        P["dev"].copy_(host, non_blocking=True)
        if span is None and getattr(self, "want_sumsq", False):
            # (optim.ClipAdamW asked) the same launch leaves the sum of squares of what each workgroup copied: the gradient norm
            # costs no pass of its own.  Valid for the WHOLE buffer only, and only until something else writes the flat gradient.
            if getattr(self, "sumsq", None) is None or self.sumsq.numel() != P["nblocks"]:
                self.sumsq = torch.empty(P["nblocks"], dtype=torch.float32, device=self.grad.device)
            L.check(L.lib().vdetr_pack_sumsq_f32(L.ptr(P["dev"]), L.ptr(P["block_entry"]), L.ptr(P["block_chunk"]), P["nblocks"],
                                                 L.ptr(self.grad), L.ptr(self.sumsq), L.stream_ptr()), "pack_sumsq")
        else:
            L.check(L.lib().vdetr_pack_f32(L.ptr(P["dev"]), L.ptr(P["block_entry"]), L.ptr(P["block_chunk"]), P["nblocks"],
                                           L.ptr(self.grad), L.stream_ptr()), "pack")
        if not capturing:
            P["events"][slot] = torch.cuda.Event()
            P["events"][slot].record()
        for g in keep:
            g.record_stream(torch.cuda.current_stream())

    # ---- optimizer interoperability (optimizer.py:6-26, utils/io.py:23-29) -----------------------------------------
    def decay_vector(self, named_params, lr, weight_decay, filter_biases_wd=True):
        """The reference's ``--filter_biases_wd`` (optimizer.py:12-15: no weight decay for 1-D parameters and ``*.bias``)
        for the one-tensor optimizer: a per-element factor ``1 - lr * wd`` (decayed) or ``1`` (exempt, and the zero
        padding).  Use with ``AdamW([flat.param], weight_decay=0)``: ``flat.data.mul_(vec)`` before ``opt.step()`` is
        AdamW's decoupled decay (``p *= 1 - lr * wd`` precedes the Adam update and the update does not read ``p``)."""
        vec = torch.ones_like(self.data)
        names = {id(p): n for n, p in named_params}
        for p in self.params:
            n = names.get(id(p), "")
            exempt = filter_biases_wd and (p.ndim == 1 or n.endswith("bias"))
            if not exempt:
                o = self.offsets[id(p)]
                vec[o:o + p.numel()] = 1.0 - lr * weight_decay
        return vec

    def per_param_optimizer_state(self, opt, params=None):
        """The one-tensor AdamW state (``exp_avg`` / ``exp_avg_sq`` of the flat buffer) as the per-parameter state
        dictionaries a per-parameter ``torch.optim.AdamW`` over ``params`` (default: this buffer's parameters, e.g. in
        ``model.parameters()`` order for a reference checkpoint, utils/io.py:23-29) would hold: a list aligned with
        ``params``, copies, independent of the flat layout."""
        st = opt.state.get(self.param, {})
        out = []
        for p in (self.params if params is None else params):
            o, n = self.offsets[id(p)], p.numel()
            d = {"step": st["step"].clone() if "step" in st else torch.tensor(0.0)}
            for k in ("exp_avg", "exp_avg_sq", "max_exp_avg_sq"):
                if k in st:
                    d[k] = st[k][o:o + n].view(p.shape).clone()
            out.append(d)
        return out

    def load_per_param_optimizer_state(self, opt, states, params=None):
        """Inverse of ``per_param_optimizer_state``: per-parameter AdamW states (a reference optimizer checkpoint, in
        ``params`` order) -> the one-tensor optimizer's state.  ``step`` must agree over the parameters."""
        plist = self.params if params is None else list(params)
        assert len(plist) == len(states)
        st = opt.state[self.param]
        steps = {float(d["step"]) for d in states if "step" in d}
        assert len(steps) <= 1, "per-parameter AdamW steps differ: not representable as one tensor"
        for k in ("exp_avg", "exp_avg_sq"):
            st.setdefault(k, torch.zeros_like(self.data))
        if steps:
            step = torch.tensor(steps.pop(), dtype=torch.float32, device=self.data.device if opt.defaults.get("capturable") else "cpu")
            st["step"] = step
        with torch.no_grad():
            for p, d in zip(plist, states):
                o, n = self.offsets[id(p)], p.numel()
                for k in ("exp_avg", "exp_avg_sq", "max_exp_avg_sq"):
                    if k in d:
                        st.setdefault(k, torch.zeros_like(self.data))[o:o + n].copy_(d[k].reshape(-1))

    def clip_scale(self, max_norm, eps=1e-6):
        """1 / clip coefficient of ``clip_grad_norm_(params, max_norm)`` as a device scalar: max(||g|| / max_norm, 1).
        Handed to a fused optimizer as ``grad_scale`` (it divides the gradient by it inside its own launch)."""
        norm = torch.linalg.vector_norm(self.grad)
        return torch.clamp((norm + eps) / max_norm, min=1.0), norm


class GradientReducer:
    def __init__(self, params, bucket_mb=25.0, overlap=True, process_group=None, bucket_views=True, flat=None, force=False,
                 break_before=()):
        """``force``: issue the collectives even on ONE rank (a 1-rank RCCL communicator: how the captured, overlapped path
        is exercised on a single GPU).  ``break_before``: parameters at which a new bucket starts whatever its size."""
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())
        self._spans = []
        self.overlap = overlap
        assert self.params, "no trainable parameters"
        dev, dtype = self.params[0].device, self.params[0].dtype
        # buckets in REVERSE parameter order: backward produces gradients roughly last-layer-first
        cap = max(int(bucket_mb * (1 << 20) / 4), 1)
        groups, cur, cur_n = [], [], 0
        breaks = {id(p) for p in break_before}
        self.flat = flat  # FlatParams: the buckets are consecutive slices of its gradient buffer
        for p in (flat.params if flat is not None else reversed(self.params)):
            assert p.device == dev and p.dtype == dtype, "one device / dtype per reducer"
            if cur and (cur_n + p.numel() > cap or id(p) in breaks):
                groups.append(cur)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            groups.append(cur)
        self.buckets, self._bucket_of, self._pending, self._launched = [], {}, [], []
        self.bucket_views = bucket_views
        self._views = []  # (param, view) in bucket order
        gview = {id(p): v for p, v in zip(flat.params, flat.grad_views)} if flat is not None else None
        for gi, g in enumerate(groups):
            n_g = sum(p.numel() for p in g)
            if self.flat is not None:  # a slice of the flat gradient buffer (alignment / slot padding included: zeros)
                start = self.flat.offsets[id(g[0])]
                end = self.flat.offsets[id(g[-1])] + g[-1].numel()
                flat = self.flat.grad[start:end]
                self._spans.append((start, end))
            else:
                flat = torch.zeros(n_g, dtype=dtype, device=dev)
            off = 0
            for p in g:
                view = gview[id(p)] if gview is not None else flat[off:off + p.numel()].view_as(p)
                if bucket_views:
                    p.grad = view  # gradients accumulate straight into the bucket
                self._views.append((p, view))
                off += p.numel()
                self._bucket_of[id(p)] = gi
            self.buckets.append(flat)
            self._pending.append(len(g))
            self._launched.append(False)
        self._counts = list(self._pending)
        self._seen = set()  # parameters whose hook has fired in the current step
        self._side = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self._handles = []
        self._hook_handles = []
        # decided now (a collective of its own), not inside the first bucket's launch — which may be under hipGraph capture
        self._use_avg = avg_reduce_supported(process_group, dev) if self.active else False
        if overlap and self.active and bucket_views:
            for p in self.params:
                self._hook_handles.append(p.register_post_accumulate_grad_hook(self._hook))

    def remove_hooks(self):
        """detach this reducer from its parameters (before another reducer takes them over)"""
        for h in self._hook_handles:
            h.remove()
        self._hook_handles = []

    # ---- hooks (eager mode) -------------------------------------------------------------------------
    def _hook(self, p):
        """Post-accumulate hook.  A parameter may receive gradient more than once per step (several uses, or the
        deferred weight-gradient flushes of runtime.defer_weight_grads() that run after loss.backward()): only its FIRST
        fire of the step counts towards its bucket, and while deferral is on no bucket is launched from a hook at all --
        its gradients are complete only after runtime.flush_weight_grads(), so finish() launches them."""
        if id(p) in self._seen:
            if self._launched[self._bucket_of[id(p)]]:
                raise RuntimeError("GradientReducer: a parameter received gradient after its bucket's all-reduce was "
                                   "launched (accumulate into it before backward ends, or use pack_and_reduce)")
            return
        self._seen.add(id(p))
        gi = self._bucket_of[id(p)]
        self._pending[gi] -= 1
        if self._pending[gi] == 0 and not self._launched[gi] and not self._hooks_deferred():
            self._launch(gi)

    @staticmethod
    def _hooks_deferred():
        from . import runtime
        return runtime.weight_grads_deferred()

    def _launch(self, gi):
        self._launched[gi] = True
        flat = self.buckets[gi]
        if self._side is not None:
            self._side.wait_stream(torch.cuda.current_stream())  # gradients of this bucket are complete
            with torch.cuda.stream(self._side):
                self._allreduce_avg(flat)
            flat.record_stream(self._side)
        else:
            self._handles.append(self._allreduce_avg(flat, async_op=True))

    # ---- public -------------------------------------------------------------------------------------
    def zero_grad(self):
        if self.bucket_views:
            for b in self.buckets:
                b.zero_()
        else:
            for p in self.params:
                p.grad = None

    def pack_and_reduce(self, grads=None):
        """bucket_views=False: one multi-tensor copy of the fresh gradients into the buckets, all-reduce, and
        ``p.grad`` -> reduced bucket views (parameters without a gradient contribute zeros).  ``grads`` (one
        tensor or None per parameter, in ``self.params`` order) overrides ``p.grad`` as the source — the static
        gradient buffers of a captured hipGraph."""
        src = {id(p): (p.grad if grads is None else g) for p, g in zip(self.params, grads or self.params)}
        if self.flat is not None:
            self.flat.pack_grads([src[id(p)] for p in self.flat.params])
            self.reduce_all()
            for p, v in self._views:
                p.grad = v
            return
        have = [(src[id(p)], v) for p, v in self._views if src[id(p)] is not None]
        missing = [v for p, v in self._views if src[id(p)] is None]
        if missing:
            torch._foreach_zero_(missing)
        if have:
            torch._foreach_copy_([v for _, v in have], [g for g, _ in have])
        self.reduce_all()
        for p, v in self._views:
            p.grad = v

    def pack_and_launch(self, buckets):
        """flat-buffer, plain-gradient mode: the parameters of these buckets hold their final ``.grad``: pack the slices and
        start the all-reduces on the side stream (join with ``finish()``).  For a step whose gradients become final part by
        part (the decoder's after its captured backward, the backbone's later)."""
        assert self.flat is not None and not self.bucket_views
        for k in buckets:
            self.flat.pack_grads_span(*self._spans[k])
            if self.active:
                self._launch(k)

    def launch_when_complete(self, buckets):
        """flat-buffer, plain-gradient mode, EAGER backward: watch the parameters of these buckets; as soon as every one of a
        bucket has received its gradient (post-accumulate hooks, whatever runtime.defer_weight_grads says: the caller vouches
        that these parameters' gradients are final when they arrive), its slice is packed and its all-reduce starts on the
        side stream — while the backward pass that is still producing the other buckets' gradients keeps running.  What the
        reference gets from DistributedDataParallel's bucket hooks (main.py:515-517).  Call ``begin_watch()`` before every
        backward pass, ``pack_and_launch(pending_watched())`` + ``finish()`` after it."""
        assert self.flat is not None and not self.bucket_views
        self._watched = {k: 0 for k in sorted(buckets)}
        self._watch_need = {k: 0 for k in sorted(buckets)}
        self._watch_order = sorted(buckets)   # the order the collectives are issued in, on EVERY rank
        self._watch_skip = None               # buckets that never complete (an unused parameter), agreed by the ranks after step 1
        self._watch_next = 0
        self._watch_seen = set()
        for p in self.flat.params:
            k = self._bucket_of.get(id(p))
            if k in self._watched:
                self._watch_need[k] += 1
                self._hook_handles.append(p.register_post_accumulate_grad_hook(self._watch_hook))

    def begin_watch(self):
        for k in self._watched:
            self._watched[k] = 0
        self._watch_seen.clear()
        self._watch_next = 0

    def _watch_hook(self, p):
        if id(p) in self._watch_seen:
            return
        self._watch_seen.add(id(p))
        k = self._bucket_of[id(p)]
        self._watched[k] += 1
        # A bucket leaves only when it and every EARLIER watched bucket are complete: the ranks then issue the same
        # collectives in the same (ascending) order on the communicator whatever order autograd produced the gradients in on
        # each of them — a rank with a parameter that got no gradient holds back the later buckets until
        # pack_and_launch(pending_watched()), it does not reorder them (DistributedDataParallel's fixed bucket order).
        order = self._watch_chain()
        while self._watch_next < len(order):
            j = order[self._watch_next]
            if self._launched[j]:
                self._watch_next += 1
                continue
            if self._watched[j] != self._watch_need[j]:
                break
            self.flat.pack_grads_span(*self._spans[j])
            if self.active:
                self._launch(j)
            else:
                self._launched[j] = True
            self._watch_next += 1

    def _watch_chain(self):
        """the fixed issue order of the watched buckets: ascending, the structurally incomplete ones (agreed set) last"""
        skip = self._watch_skip or ()
        return [k for k in self._watch_order if k not in skip] + [k for k in self._watch_order if k in skip]

    def pending_watched(self):
        """watched buckets that did not leave during the backward pass (a parameter without gradient in one of them, or in an
        earlier one of the chain), in chain order.  The FIRST call (every rank makes it once per step) agrees on the buckets that
        never complete — the union over the ranks of this step's leftovers, one small MAX all-reduce: they go to the end of the
        chain from the next step on, so that an unused parameter does not hold the buckets behind it back for good."""
        if not hasattr(self, "_watch_order"):
            return []
        chain = self._watch_chain()
        pend = [k for k in chain if not self._launched[k]]
        if self._watch_skip is None and getattr(self, "_watch_incomplete", None) is None:
            # incomplete HERE (not merely held back by an earlier bucket of the chain); agreed on in finish(), when every rank
            # has issued all of this step's bucket collectives (the ranks may have issued different numbers of them so far)
            self._watch_incomplete = [k for k in pend if self._watched[k] != self._watch_need[k]]
        return pend

    def buckets_of(self, params):
        """indices of the buckets that hold these parameters"""
        return sorted({self._bucket_of[id(p)] for p in params if id(p) in self._bucket_of})

    def reduce_phased(self):
        """The data-parallel step of a loop that parks its weight gradients (runtime.defer_weight_grads) and keeps ordinary
        ``p.grad`` tensors (bucket_views=False, flat buffer): bucket by bucket, compute the parked gradients that belong to
        it, pack its slice of the flat buffer (one launch), and start its all-reduce on the side stream while the next
        bucket's gradients are computed; join, point ``p.grad`` at the reduced views.  Every call is stream-ordered (no host
        wait), so the whole sequence — RCCL launches included — can sit inside ONE captured hipGraph together with the
        forward, the backward and the optimizer step: the fork / join of the side stream become graph dependencies.
        The reference gets this overlap from DistributedDataParallel's bucket hooks (main.py:515-517)."""
        from . import runtime
        assert self.flat is not None and not self.bucket_views, "reduce_phased: flat buffer + plain gradients"

        def after_phase(k):
            self.flat.pack_grads_span(*self._spans[k])
            if self.active:
                self._launch(k)

        runtime.flush_weight_grads_phased(lambda p: self._bucket_of.get(id(p), 0), len(self.buckets), after_phase)
        self.finish()
        for p, v in self._views:
            p.grad = v

    def finish(self):
        """End of backward: flush buckets that never completed (unused parameters), join the side stream."""
        if self.active:
            for gi in range(len(self.buckets)):
                if not self._launched[gi]:
                    self._launch(gi)
            for h in self._handles:
                h.wait()
            self._last_works = self._handles  # (kept for collectives_done(): what a capture has to see finished first)
            self._handles = []
            if self._side is not None:
                torch.cuda.current_stream().wait_stream(self._side)
        if getattr(self, "_watch_incomplete", None) is not None and self._watch_skip is None:
            mine = torch.zeros(len(self.buckets), dtype=torch.int32, device=self.buckets[0].device)
            for k in self._watch_incomplete:
                mine[k] = 1
            if self.active:  # every bucket of the step is out on every rank: the same position in everybody's sequence
                dist.all_reduce(mine, op=dist.ReduceOp.MAX, group=self.group)
            flags = mine.tolist()
            self._watch_skip = {k for k in self._watch_order if flags[k]}
            self._watch_incomplete = None
        self._pending = list(self._counts)
        self._launched = [False] * len(self.buckets)
        self._seen.clear()

    def collectives_done(self):
        """True when every collective of the last finished step has completed on the device (Work.is_completed(); a handle that
        cannot be asked counts as done after the caller's device synchronise)."""
        for h in getattr(self, "_last_works", ()):
            try:
                if not h.is_completed():
                    return False
            except RuntimeError:
                pass
        return True

    def _allreduce_avg(self, t, async_op=False):
        """Average over the ranks.  RCCL averages inside the collective (ncclAvg) where every rank's library has it — a
        COLLECTIVE decision taken once per group (avg_reduce_supported), never a per-rank try / except around a collective:
        ranks that disagreed would pair an AVG with a SUM.  Otherwise (gloo: the CPU tests) divide, then sum."""
        if self._use_avg is None:
            self._use_avg = avg_reduce_supported(self.group, t.device)
        if self._use_avg:
            return dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=async_op)
        t.div_(self.world)
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def reduce_all(self):
        """Average every bucket on the current stream (after a captured forward+backward has been replayed)."""
        if self.active:
            for b in self.buckets:
                self._allreduce_avg(b)

    def grad_bytes(self):
        return sum(b.numel() * b.element_size() for b in self.buckets)


_AVG_VERDICT = {}


def destroy_distributed():
    """dist.destroy_process_group() + the per-group caches of this module"""
    _AVG_VERDICT.clear()
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def avg_reduce_supported(group=None, device=None):
    """True iff EVERY rank of the group may use ``ReduceOp.AVG`` (ncclAvg: NCCL / RCCL >= 2.10).  Each rank reads its own
    library version (no collective is attempted to find out), the verdicts are combined with a MIN all-reduce, and the
    result is cached per group: all ranks take the same branch in GradientReducer._allreduce_avg from the first call on."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    # keyed by what the verdict depends on — the backend and the ranks of the group — not by the group object's id (which
    # a later group may reuse); init_distributed / destroy clear the cache
    try:
        ranks = tuple(dist.get_process_group_ranks(group if group is not None else dist.group.WORLD))
    except Exception:  # noqa: BLE001
        ranks = (dist.get_world_size(group),)
    key = (dist.get_backend(group), ranks)
    if key in _AVG_VERDICT:
        return _AVG_VERDICT[key]
    if dist.get_backend(group) != "nccl":
        _AVG_VERDICT[key] = False  # gloo has no AVG at all: the same answer on every rank without asking
        return False
    try:
        ver = torch.cuda.nccl.version()
        local = tuple(ver[:2]) >= (2, 10) if isinstance(ver, tuple) else int(ver) >= 21000
    except Exception:  # noqa: BLE001
        local = False
    dev = device if device is not None and torch.device(device).type == "cuda" else torch.device("cuda", torch.cuda.current_device())
    t = torch.tensor([1 if local else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    _AVG_VERDICT[key] = bool(int(t.item()))
    return _AVG_VERDICT[key]


def broadcast_parameters(module, src=0, process_group=None):
    """Same initial weights on every rank (DDP does this at construction)."""
    if dist.is_initialized() and dist.get_world_size(process_group) > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src, group=process_group)


# ---- the engine's loss reductions (engine.py:97-98; utils/dist.py:67-110) -------------------------------------------
def all_reduce_average(tensor):
    """utils/dist.py:82-84: mean over ranks (identity on one rank).  In place, as the reference's all_reduce_sum is."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() < 2:
        return tensor
    t = tensor[None] if tensor.ndim == 0 else tensor
    dist.all_reduce(t)
    t /= dist.get_world_size()
    return t[0] if tensor.ndim == 0 else t


def reduce_dict(input_dict, average=True):
    """utils/dist.py:88-110: every value averaged (or summed) over the ranks, keys sorted so that all ranks agree.  Values
    that are views of ONE tensor (the device criterion's loss_dict: v-detr_amd/criterion.py) are reduced through that
    tensor directly: one collective, no stack kernel."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() < 2:
        return input_dict
    with torch.no_grad():
        names = sorted(input_dict.keys())
        vals = [input_dict[k] for k in names]
        base = vals[0]._base if vals and vals[0]._base is not None else None
        if (base is not None and base.is_contiguous() and base.is_floating_point()
                and all(v._base is base and v.ndim == 0 and v.dtype == base.dtype for v in vals)):
            # reduce only the span of the base that the values occupy: other words of the base (the criterion's
            # 64-bit cardinality tickets, reinterpreted as floats) must not go through a float sum
            off0 = base.storage_offset()
            offs = [v.storage_offset() - off0 for v in vals]
            lo, hi = min(offs), max(offs) + 1
            red = base.detach().reshape(-1)[lo:hi].clone()
            dist.all_reduce(red)
            if average:
                red /= dist.get_world_size()
            return {k: red[o - lo] for k, o in zip(names, offs)}
        values = torch.stack([v.detach() for v in vals], dim=0)
        dist.all_reduce(values)
        if average:
            values /= dist.get_world_size()
        return dict(zip(names, values))
