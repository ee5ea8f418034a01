"""Sparse-convolution primitives of the backbone on top of lib/libvdetr_hip.so.

A sparse tensor is a sorted int64 KEY vector [N] (``pack_keys``: batch, x, y, z as biased 16-bit fields, so ascending key
order is lexicographic coordinate order) plus a point-major feature table [N, C] — the layout the hot path's FPS and row
gathers consume.  The four native entry points (kernel map, inverse map, column gather, adjoint gather: csrc/sparse_conv.hip)
are geometry / HBM-stream kernels; the contraction itself is ONE library GEMM per layer over the gathered columns
(``col [N, K*Cin] @ W [K*Cin, Cout]``), its two gradients two more.  No CPU path: every entry point raises on CPU tensors.
"""
import ctypes
import os
import threading

import numpy as np
import torch
from torch.autograd import Function

from . import _lib as L

KEY_BIAS = 32768


def pack_keys(coords, check=True):
    """[N,4] integer (batch, x, y, z) -> int64 keys (include/vdetr_hip.h: vdetr_sp_kernel_map_i32).  ``check`` validates the
    16-bit field range (a device->host sync: used on the raw input coordinates only)."""
    c = coords.to(torch.int64)
    if check and c.numel():
        lo, hi = int(c[:, 1:].min()), int(c[:, 1:].max())
        if lo < -KEY_BIAS or hi >= KEY_BIAS or int(c[:, 0].min()) < 0 or int(c[:, 0].max()) >= 32768:
            raise ValueError(f"voxel coordinates outside the 16-bit key range: [{lo}, {hi}]")
    return (c[:, 0] << 48) | ((c[:, 1] + KEY_BIAS) << 32) | ((c[:, 2] + KEY_BIAS) << 16) | (c[:, 3] + KEY_BIAS)


def unpack_keys(keys):
    """int64 keys -> [N,4] int32 (batch, x, y, z)."""
    b = keys >> 48
    x = ((keys >> 32) & 0xFFFF) - KEY_BIAS
    y = ((keys >> 16) & 0xFFFF) - KEY_BIAS
    z = (keys & 0xFFFF) - KEY_BIAS
    return torch.stack((b, x, y, z), dim=1).to(torch.int32)


def kernel_map(in_keys, out_keys, offsets):
    """nbr [K, Nout] int32: row of the (sorted) input site at out_keys[u] + offsets[k], -1 where unoccupied."""
    L.require_gpu(in_keys, "in_keys")
    assert in_keys.dtype == torch.int64 and out_keys.dtype == torch.int64 and offsets.dtype == torch.int32
    K, nout = offsets.shape[0], out_keys.shape[0]
    nbr = torch.empty((K, nout), dtype=torch.int32, device=in_keys.device)
    L.check(L.lib().vdetr_sp_kernel_map_i32(L.ptr(in_keys.contiguous()), in_keys.shape[0], L.ptr(out_keys.contiguous()), nout,
                                            L.ptr(offsets.contiguous()), K, L.ptr(nbr), L.stream_ptr()), "sp_kernel_map")
    return nbr


def inverse_map(nbr, nin):
    """inv [K, Nin] int32: the output row that reads input row i through offset k (unique on a lattice), -1 if none."""
    L.require_gpu(nbr, "nbr")
    K, nout = nbr.shape
    inv = torch.full((K, nin), -1, dtype=torch.int32, device=nbr.device)
    L.check(L.lib().vdetr_sp_inverse_map_i32(L.ptr(nbr), K, nout, nin, L.ptr(inv), L.stream_ptr()), "sp_inverse_map")
    return inv


def gather_cols(feats, nbr):
    """col [Nout, K, C] = feats[nbr[k, u]] (zeros where -1); C % 4 == 0."""
    L.require_gpu(feats, "feats")
    L.require_float(feats, "feats")
    L.require_contiguous(feats, "feats")
    K, nout = nbr.shape
    C = feats.shape[1]
    col = torch.empty((nout, K, C), dtype=torch.float32, device=feats.device)
    L.check(L.lib().vdetr_sp_gather_cols_f32(L.ptr(feats), L.ptr(nbr), K, nout, C, L.ptr(col), L.stream_ptr()), "sp_gather_cols")
    return col


def gather_sum(dcol, inv, offset_major=False, flat=False):
    """din [Nin, C] = sum_k dcol[inv[k, i], k] (the adjoint of ``gather_cols``, as a gather).  ``offset_major``: dcol is
    [K, M, C] (slice k holds the compacted rows of offset k) instead of [Nout, K, C]; ``flat``: dcol is [P, C] and ``inv``
    holds absolute rows (the pair lists of PairPlan)."""
    L.require_gpu(dcol, "dcol")
    L.require_float(dcol, "dcol")
    L.require_contiguous(dcol, "dcol")
    K, nin = inv.shape
    C = dcol.shape[-1]
    M = -1 if flat else (dcol.shape[1] if offset_major else 0)
    assert flat or dcol.shape[0 if offset_major else 1] == K
    din = torch.empty((nin, C), dtype=torch.float32, device=dcol.device)
    L.check(L.lib().vdetr_sp_gather_sum_f32(L.ptr(dcol), L.ptr(inv), K, nin, C, M, L.ptr(din), L.stream_ptr()), "sp_gather_sum")
    return din


class _SparseConvFn(Function):
    """out [Nout, Cout] = sum_k feats[nbr[k]] @ W[k]   (W [K, Cin, Cout]); gradients w.r.t. feats and W."""

    @staticmethod
    def forward(ctx, feats, weight, nbr, inv):
        K, cin, cout = weight.shape
        col = gather_cols(feats, nbr)                       # [Nout, K, Cin]
        out = col.view(col.shape[0], K * cin) @ weight.reshape(K * cin, cout)
        ctx.save_for_backward(col, weight, inv)
        return out

    @staticmethod
    def backward(ctx, dout):
        col, weight, inv = ctx.saved_tensors
        K, cin, cout = weight.shape
        dout = dout.contiguous()
        dw = dfeats = None
        if ctx.needs_input_grad[1]:
            dw = (col.view(col.shape[0], K * cin).t() @ dout).view(K, cin, cout)
        if ctx.needs_input_grad[0]:
            dcol = (dout @ weight.reshape(K * cin, cout).t()).view(-1, K, cin)
            dfeats = gather_sum(dcol, inv)
        return dfeats, dw, None, None


class ConvPlan:
    """Geometry of one sparse convolution, compacted per kernel offset (built once per scene and layer shape, no gradients).

    On indoor scans a site has 3-7 of its 27 neighbours, so the im2col operand of ``_SparseConvFn`` is ~85 % zeros.  The
    plan keeps, for every kernel offset k, only the n_k (output, input) pairs that exist; offsets with similar n_k form a
    GROUP whose lists are padded to the group's longest (M) so that a group is ONE batched GEMM  [Kg, M, Cin] x [Kg, Cin, Cout]:
      src  [Kg, M]    input row of pair j of offset k        (-1 padding)
      rows [Kg, M]    output row of that pair
      slot [Kg, Nout] pair index j of output u in offset k   (-1: u has no neighbour through k)
      islot[Kg, Nin]  pair index j of input i in offset k
    forward  out[u]  = sum_k (A[k] W[k])[slot[k][u]]   with A[k][j] = in[src[k][j]]       (gather, bmm, gather-sum)
    backward din[i]  = sum_k (dP[k] W[k]^T)[islot[k][i]], dW[k] = A[k]^T dP[k], dP[k][j] = dout[rows[k][j]].
    Both reductions are gathers in a fixed order: deterministic, no atomics.  The centre offset of a stride-1 layer is the
    identity map: no gather at all."""

    SORTED_MIN_CHANNELS = 128 * 128  # Cin * Cout from which gathering a group's weights costs less than its padding
    SORTED_SPLIT = 1.3               # hit counts within a count-sorted group differ by at most this factor
    SORTED_MAX_GROUPS = 6

    def __init__(self, nbr, nin):
        K, nout = nbr.shape
        self.nin, self.nout, self.K = nin, nout, K
        self.nbr = nbr
        self.counts = (nbr >= 0).sum(1).tolist()  # geometry phase: one sync per plan
        self.pairs = int(sum(self.counts))
        self.ident = None  # the centre offset of a stride-1 layer maps every site to itself: a plain GEMM, no gather
        if nin == nout and K % 2 == 1 and self.counts[K // 2] == nout:
            if bool((nbr[K // 2] == torch.arange(nin, dtype=torch.int32, device=nbr.device)).all()):
                self.ident = K // 2
        self._groups = {}
        self.groups = self.grouping("ranges")
        self.padded_pairs = sum(g["n"] * g["M"] for g in self.groups)

    def grouping(self, kind):
        """"ranges": groups are CONTIGUOUS offset ranges, a group's weights are the view W[k0:k1] (no copy of the parameter,
        no scatter of its gradient) — the identity offset on its own, the ranges before / after it as one batched GEMM each.
        "sorted": offsets sorted by hit count and cut wherever the count drops by SORTED_SPLIT: ~1.15x padding instead of
        ~1.6x, at the price of an index_select of W per group (worth it for wide layers)."""
        if kind not in self._groups:
            K, counts, ident = self.K, self.counts, self.ident
            sets = []
            if kind == "ranges":
                for k0, k1 in ([(0, K)] if ident is None else [(0, ident), (ident + 1, K)]):
                    ks = [k for k in range(k0, k1)]
                    if ks and max(counts[k] for k in ks) > 0:
                        sets.append(ks)
            else:
                order = sorted((k for k in range(K) if counts[k] > 0 and k != ident), key=lambda k: -counts[k])
                while order:
                    top = counts[order[0]]
                    ks = [k for k in order if counts[k] * self.SORTED_SPLIT >= top] if len(sets) < self.SORTED_MAX_GROUPS - 1 else order
                    order = [k for k in order if k not in ks]
                    sets.append(sorted(ks))
            groups = [self._build(ks) for ks in sets]
            if ident is not None:
                groups.insert(1 if kind == "ranges" and len(groups) > 1 else 0,
                              {"ks": [ident], "k0": ident, "k1": ident + 1, "n": 1, "identity": True, "M": self.nout})
            self._groups[kind] = groups
        return self._groups[kind]

    def _build(self, ks):
        nbr, dev = self.nbr, self.nbr.device
        contiguous = ks == list(range(ks[0], ks[-1] + 1))
        sel = nbr[ks[0]:ks[-1] + 1] if contiguous else nbr[torch.tensor(ks, dtype=torch.int64, device=dev)]
        M = max(self.counts[k] for k in ks)
        v = sel >= 0
        slot = torch.where(v, v.cumsum(1, dtype=torch.int32) - 1, torch.full_like(sel, -1))
        kloc, u = torch.nonzero(v, as_tuple=True)
        j = slot[kloc, u].long()
        i = sel[kloc, u]
        rows = torch.full((len(ks), M), -1, dtype=torch.int32, device=dev)
        src = torch.full((len(ks), M), -1, dtype=torch.int32, device=dev)
        islot = torch.full((len(ks), self.nin), -1, dtype=torch.int32, device=dev)
        rows[kloc, j] = u.int()
        src[kloc, j] = i
        islot[kloc, i.long()] = j.int()
        return {"ks": ks, "k0": ks[0] if contiguous else None, "k1": ks[-1] + 1 if contiguous else None, "n": len(ks),
                "kidx": None if contiguous else torch.tensor(ks, dtype=torch.int64, device=dev), "identity": False, "M": M,
                "src": src, "rows": rows, "slot": slot.contiguous(), "islot": islot}

    def select(self, cin, cout):
        return self.grouping("sorted" if cin * cout >= self.SORTED_MIN_CHANNELS and not _RANGES_ONLY else "ranges")


def _group_weight(weight, g):
    return weight[g["k0"]:g["k1"]] if g["k0"] is not None else weight.index_select(0, g["kidx"])


class _PlannedConvFn(Function):
    """out [Nout, Cout] = sum_k in[nbr[k]] @ W[k] through a ConvPlan (see there)."""

    @staticmethod
    def forward(ctx, feats, weight, plan):
        groups = plan.select(weight.shape[1], weight.shape[2])
        out, saved = None, []
        for g in groups:
            w = _group_weight(weight, g)
            if g["identity"]:
                part, a3 = feats @ w[0], feats
            else:
                kg, M = g["src"].shape
                a3 = gather_cols(feats, g["src"].view(1, kg * M)).view(kg, M, feats.shape[1])
                part = gather_sum(torch.bmm(a3, w), g["slot"], offset_major=True)
            saved.append(a3)
            out = part if out is None else out.add_(part)
        if out is None:
            out = feats.new_zeros((plan.nout, weight.shape[2]))
        ctx.plan, ctx.groups = plan, groups
        ctx.save_for_backward(weight, *saved)
        return out

    @staticmethod
    def backward(ctx, dout):
        weight, *saved = ctx.saved_tensors
        plan = ctx.plan
        dout = dout.contiguous()
        need_f, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dfeats = None
        dw = torch.zeros_like(weight) if need_w else None
        for g, a3 in zip(ctx.groups, saved):
            w = _group_weight(weight, g)
            if g["identity"]:
                if need_w:
                    torch.mm(a3.t(), dout, out=dw[g["k0"]])
                part = dout @ w[0].t() if need_f else None
            else:
                kg, M = g["rows"].shape
                dp = gather_cols(dout, g["rows"].view(1, kg * M)).view(kg, M, dout.shape[1])
                if need_w:
                    if g["k0"] is not None:
                        torch.bmm(a3.transpose(1, 2), dp, out=dw[g["k0"]:g["k1"]])
                    else:
                        dw.index_copy_(0, g["kidx"], torch.bmm(a3.transpose(1, 2), dp))
                part = gather_sum(torch.bmm(dp, w.transpose(1, 2)), g["islot"], offset_major=True) if need_f else None
            if need_f:
                dfeats = part if dfeats is None else dfeats.add_(part)
        if need_f and dfeats is None:
            dfeats = dout.new_zeros((plan.nin, weight.shape[1]))
        return dfeats, dw, None


_RANGES_ONLY = os.environ.get("VDETR_SP_RANGES_ONLY", "0") == "1"  # A/B switch: never use the count-sorted grouping
_WGRAD_WGS = int(os.environ.get("VDETR_SP_WGRAD_WGS", "768"))  # workgroups per weight-gradient launch (A/B switch)


class _PinnedCounts:
    """a ring of pinned host rows for the asynchronous copies of the plans' pair counts (one allocation per process)"""
    rows, width = 512, 64
    _buf, _next = None, 0
    _lock = threading.Lock()

    @classmethod
    def take(cls, n):
        assert n <= cls.width
        with cls._lock:
            if cls._buf is None:
                cls._buf = torch.empty((cls.rows, cls.width), dtype=torch.int32, pin_memory=True)
            row = cls._buf[cls._next % cls.rows]
            cls._next += 1
        return row[:n]


class PairPlan:
    """Geometry of one sparse convolution as a PAIR LIST sorted by kernel offset (built once per scene and layer shape):
      pin [P], pout [P]   input / output row of pair p; the pairs of offset k are the segment seg[k] .. seg[k+1]
      slot [K, Nout]      pair (absolute index) through which output u reads offset k, -1 if none
      islot [K, Nin]      pair through which input i is read with offset k
      tiles [T, 3]        (k, first pair, count <= 128): the 128-pair tiles of vdetr_sp_pairs_gemm_f32
    Nothing is padded: the fused kernels (csrc/sparse_conv.hip) gather rows straight into the matrix-core operands.

    On the GPU the lists are built by vdetr_sp_pair_plan_i32 without a host round trip; what needs the pair COUNTS on the host
    (P, seg, tiles, the weight-gradient chunks) is derived on first use from an asynchronous copy, so that a whole scene's
    plans cost one synchronisation instead of three each."""

    _LAZY = ("P", "pairs", "counts", "seg", "tiles", "ntiles", "pin", "pout")

    def __init__(self, nbr, nin):
        K, nout = nbr.shape
        self.K, self.nin, self.nout = K, nin, nout
        self._chunks = {}
        self._pending = None
        if nbr.is_cuda and K + 1 <= _PinnedCounts.width:
            self._launch(nbr)
        else:
            self._build_host(nbr)

    # ---- tensor-expression construction (CPU tensors: the host statement the device builder is tested against) --------
    def _build_host(self, nbr):
        K, nout, nin, dev = self.K, self.nout, self.nin, nbr.device
        valid = nbr >= 0
        counts = valid.sum(1).tolist()
        pk, pout = torch.nonzero(valid, as_tuple=True)      # sorted by (k, u)
        P = int(pk.shape[0])
        self.pout = pout.int().contiguous()
        self.pin = nbr[pk, pout].contiguous()
        idx = torch.arange(P, dtype=torch.int32, device=dev)
        self.slot = torch.full((K, nout), -1, dtype=torch.int32, device=dev)
        self.slot[pk, pout] = idx
        self.islot = torch.full((K, nin), -1, dtype=torch.int32, device=dev)
        self.islot[pk, self.pin.long()] = idx
        self._tables(counts)

    # ---- device construction ------------------------------------------------------------------------------------------
    def _launch(self, nbr):
        K, nout, nin, dev = self.K, self.nout, self.nin, nbr.device
        lib = L.lib()
        nbr = nbr.contiguous()
        cap = max(K * nout, 1)
        self._pin_buf = torch.empty(cap, dtype=torch.int32, device=dev)
        self._pout_buf = torch.empty(cap, dtype=torch.int32, device=dev)
        self.slot = torch.empty((K, nout), dtype=torch.int32, device=dev)
        self.islot = torch.empty((K, nin), dtype=torch.int32, device=dev)
        counts_dev = torch.empty(K + 1, dtype=torch.int32, device=dev)
        ws = torch.empty(max(lib.vdetr_sp_pair_plan_workspace_ints(K, nout), 1), dtype=torch.int32, device=dev)
        L.check(lib.vdetr_sp_pair_plan_i32(L.ptr(nbr), K, nout, nin, L.ptr(self._pin_buf), L.ptr(self._pout_buf), L.ptr(self.slot),
                                           L.ptr(self.islot) if nin else None, L.ptr(counts_dev), L.ptr(ws), L.stream_ptr()),
                "sp_pair_plan")
        host = _PinnedCounts.take(K + 1)
        host.copy_(counts_dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._pending = (host, ev, counts_dev, ws)  # device buffers kept alive until the copy has landed

    def finalize(self):
        """wait for the pair counts (no-op once done) and derive the host-side tables"""
        if self._pending is not None:
            host, ev, _, _ = self._pending
            ev.synchronize()
            counts = host.tolist()
            self._pending = None
            P = counts[self.K]
            self.pin, self.pout = self._pin_buf[:P], self._pout_buf[:P]
            self._tables(counts[:self.K])
            for cin, cout in sorted(self.__dict__.pop("_wanted", ())):
                self.wgrad_chunks(cin, cout)
        return self

    def __getattr__(self, name):  # only reached for attributes not set yet
        if name in PairPlan._LAZY and self.__dict__.get("_pending") is not None:
            self.finalize()
            return self.__dict__[name]
        raise AttributeError(name)

    @staticmethod
    def _cut(counts, seg, L):
        """every segment k (seg[k], counts[k]) cut into pieces of at most L: (k, start, length) arrays and pieces per segment;
        array arithmetic instead of a Python loop per piece (a scene has ~10^4 pieces over its 16 plans, and the loader thread
        that builds them shares the interpreter lock with the thread that launches the training step)"""
        counts = np.asarray(counts, dtype=np.int64)
        per = -(-counts // L)
        kidx = np.repeat(np.arange(counts.shape[0], dtype=np.int64), per)
        first = np.cumsum(per) - per
        within = np.arange(kidx.shape[0], dtype=np.int64) - first[kidx]
        start = np.asarray(seg[:-1], dtype=np.int64)[kidx] + within * L
        length = np.minimum(L, counts[kidx] - within * L)
        return kidx, start, length, per

    def _tables(self, counts):
        dev = self.slot.device
        self.counts = counts
        self.P = self.pairs = int(sum(counts))
        seg = [0]
        for c in counts:
            seg.append(seg[-1] + c)
        self.seg = seg
        kidx, start, length, _ = PairPlan._cut(counts, seg, 128)
        self.ntiles = int(kidx.shape[0])
        tiles = np.stack([kidx, start, length], 1).astype(np.int32) if self.ntiles else np.zeros((1, 3), np.int32)
        self.tiles = torch.from_numpy(tiles).to(dev)

    def request_wgrad(self, cin, cout):
        """a layer of cin x cout channels will ask for wgrad_chunks: build that table with the geometry (`finalize`), not inside
        the backward pass, where its upload would make the launching thread wait for the device"""
        cin, cout = cin + (-cin) % 16, cout + (-cout) % 16  # (sparse_conv pads the channels of the fused kernels to 16)
        if self.__dict__.get("_pending") is None:
            self.wgrad_chunks(cin, cout)
        else:
            self.__dict__.setdefault("_wanted", set()).add((cin, cout))

    def wgrad_chunks(self, cin, cout):
        """(chunks [n, 4] i32, cseg [K+1] i32, n): every offset's segment cut into chunks of at most L pairs, L chosen so that
        the launch has ~3 workgroups per CU.  Chunks of EQUAL length, not an equal number of chunks per offset: the centre
        offset holds a quarter of all pairs, and a launch is as slow as its longest workgroup.  Chunk c writes partial c;
        the partials of offset k are cseg[k] .. cseg[k+1]."""
        t = 128 if (cin >= 128 and cout >= 128) else 64  # channel tile of vdetr_sp_pairs_wgrad_f32
        key = (-(-cin // t)) * (-(-cout // t))
        if key not in self._chunks:
            want = max(1, _WGRAD_WGS // key)
            L = max(64, -(-(-(-self.P // want)) // 16) * 16)
            kidx, start, length, per = PairPlan._cut(self.counts, self.seg, L)
            n = int(kidx.shape[0])
            rows = np.stack([kidx, start, length, np.arange(n, dtype=np.int64)], 1).astype(np.int32) if n else np.zeros((1, 4), np.int32)
            cseg = np.concatenate([[0], np.cumsum(per)]).astype(np.int32)
            dev = self.slot.device
            self._chunks[key] = (torch.from_numpy(rows).to(dev), torch.from_numpy(cseg).to(dev), n)
        return self._chunks[key]


def pairs_gemm(x, arow, weight, plan, transposed):
    """y [P, Cout] = x[arow[p]] @ W[k(p)]  (transposed: y [P, Cin] = x[arow[p]] @ W[k(p)]^T)"""
    L.require_gpu(x, "x")
    L.require_float(x, "x")
    L.require_contiguous(x, "x")
    K, cin, cout = weight.shape
    if x.numel() * 4 >= 2 ** 32 or cin * cout * 4 >= 2 ** 32:
        raise RuntimeError("pairs_gemm: a feature table of 4 GB or more (32-bit buffer offsets in the persistent kernel): split the batch")
    y = torch.empty((plan.P, cin if transposed else cout), dtype=torch.float32, device=x.device)
    L.check(L.lib().vdetr_sp_pairs_gemm_f32(L.ptr(x), L.ptr(arow), L.ptr(weight), L.ptr(plan.tiles), plan.ntiles, cin, cout,
                                            1 if transposed else 0, L.ptr(y), L.stream_ptr()), "sp_pairs_gemm")
    return y


def pairs_wgrad(x, dy, plan, cin, cout):
    """dW [K, Cin, Cout] = sum over the pairs of offset k of x[pin[p]]^T dy[pout[p]]"""
    L.require_gpu(x, "x")
    L.require_contiguous(x, "x")
    L.require_contiguous(dy, "dy")
    chunks, cseg, n = plan.wgrad_chunks(cin, cout)
    part = torch.empty((max(n, 1), cin, cout), dtype=torch.float32, device=x.device)
    L.check(L.lib().vdetr_sp_pairs_wgrad_f32(L.ptr(x), L.ptr(dy), L.ptr(plan.pin), L.ptr(plan.pout), L.ptr(chunks),
                                             n, cin, cout, L.ptr(part), L.stream_ptr()), "sp_pairs_wgrad")
    dw = torch.empty((plan.K, cin, cout), dtype=torch.float32, device=x.device)
    L.check(L.lib().vdetr_sp_wgrad_reduce_f32(L.ptr(part), L.ptr(cseg), plan.K, cin * cout, L.ptr(dw), L.stream_ptr()),
            "sp_wgrad_reduce")
    return dw


class _PairsConvFn(Function):
    """out [Nout, Cout] = sum_k in[nbr[k]] @ W[k] through a PairPlan and the fused matrix-core kernels."""

    @staticmethod
    def forward(ctx, feats, weight, plan):
        weight = weight.contiguous()
        if plan.P == 0:  # no site of the layer has a neighbour (a 1x1x1 stride-2 layer on a tiny cloud)
            out = feats.new_zeros((plan.nout, weight.shape[2]))
        else:
            out = gather_sum(pairs_gemm(feats, plan.pin, weight, plan, False), plan.slot, flat=True)
        ctx.plan = plan
        ctx.save_for_backward(feats, weight)
        return out

    @staticmethod
    def backward(ctx, dout):
        feats, weight = ctx.saved_tensors
        plan = ctx.plan
        dout = dout.contiguous()
        K, cin, cout = weight.shape
        dfeats = dw = None
        if plan.P == 0:
            return (feats.new_zeros(feats.shape) if ctx.needs_input_grad[0] else None,
                    torch.zeros_like(weight) if ctx.needs_input_grad[1] else None, None)
        if ctx.needs_input_grad[0]:
            dfeats = gather_sum(pairs_gemm(dout, plan.pout, weight, plan, True), plan.islot, flat=True)
        if ctx.needs_input_grad[1]:
            dw = pairs_wgrad(feats, dout, plan, cin, cout)
        return dfeats, dw, None


_ACT = {None: 0, "none": 0, "relu": 1, "elu": 2}


class _SpBnActFn(Function):
    """y = act(batch_norm(x) + residual) over a point-major table [N, C] (csrc/sp_bn.hip)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, nbt, training, momentum, eps, act):
        L.require_gpu(x, "x")
        L.require_float(x, "x")
        x = x.contiguous()
        N, C = x.shape
        lib = L.lib()
        d = L.SpBnDesc()
        d.N, d.C, d.act, d.training, d.eps, d.momentum = N, C, act, 1 if training else 0, float(eps), float(momentum)
        y = torch.empty_like(x)
        save_mean = torch.empty(C, dtype=torch.float32, device=x.device)
        save_invstd = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = L.workspace(lib.vdetr_sp_bn_workspace_bytes(N, C), x.device)
        res = residual.contiguous() if residual is not None else None
        d.x, d.gamma, d.beta, d.residual = x.data_ptr(), L.ptr(gamma).value, L.ptr(beta).value, L.ptr(res).value
        d.running_mean, d.running_var = L.ptr(running_mean).value, L.ptr(running_var).value
        d.num_batches_tracked = L.ptr(nbt).value if training else None
        d.y, d.save_mean, d.save_invstd, d.workspace = y.data_ptr(), save_mean.data_ptr(), save_invstd.data_ptr(), ws.data_ptr()
        L.check(lib.vdetr_sp_bn_act_fwd_f32(ctypes.byref(d), L.stream_ptr()), "sp_bn_act_fwd")
        ctx.save_for_backward(x, gamma, y, save_mean, save_invstd)
        ctx.cfg = (act, training, eps, residual is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, y, save_mean, save_invstd = ctx.saved_tensors
        act, training, eps, has_res = ctx.cfg
        N, C = x.shape
        lib = L.lib()
        dy = dy.contiguous()
        d = L.SpBnDesc()
        d.N, d.C, d.act, d.training, d.eps = N, C, act, 1 if training else 0, float(eps)
        ws = L.workspace(lib.vdetr_sp_bn_workspace_bytes(N, C), x.device)
        d.x, d.gamma, d.y = x.data_ptr(), L.ptr(gamma).value, y.data_ptr()
        d.save_mean, d.save_invstd, d.workspace = save_mean.data_ptr(), save_invstd.data_ptr(), ws.data_ptr()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dres = torch.empty_like(x) if (has_res and ctx.needs_input_grad[3]) else None
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device) if gamma is not None else None
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device) if gamma is not None else None
        L.check(lib.vdetr_sp_bn_act_bwd_f32(ctypes.byref(d), L.ptr(dy), L.ptr(dx), L.ptr(dres), L.ptr(dgamma), L.ptr(dbeta),
                                            L.stream_ptr()), "sp_bn_act_bwd")
        return dx, dgamma, dbeta, dres, None, None, None, None, None, None, None


class _SpSyncBnActFn(Function):
    """_SpBnActFn with batch statistics over all data-parallel ranks (bn_act.set_sync; the reference converts the backbone's
    BatchNorms to SyncBatchNorm like every other, main.py:512-514).  Ranks hold different site counts: the local (mean, M2,
    N) triples are all-gathered and merged; the normalisation itself is the fused launch (its running-statistics path, fed
    with the global batch statistics); the backward's two sums are all-reduced between its reduction and its element pass,
    which therefore run as tensor expressions here."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, nbt, momentum, eps, act):
        import torch.distributed as dist
        from . import bn_act as BNA
        x = x.contiguous()
        N, C = x.shape
        grp = BNA._sync["group"]
        var_l, mean_l = torch.var_mean(x, 0, unbiased=False)
        loc = torch.stack((mean_l, var_l * N, torch.full_like(mean_l, float(N))))
        world = dist.get_world_size(grp)
        allr = torch.empty((world, 3, C), dtype=torch.float32, device=x.device)
        dist.all_gather(list(allr.unbind(0)), loc, group=grp)
        cnt = allr[:, 2]
        n = cnt.sum(0)
        mean = (allr[:, 0] * cnt).sum(0) / n
        m2 = (allr[:, 1] + cnt * (allr[:, 0] - mean) ** 2).sum(0)
        var = m2 / n
        if running_mean is not None:
            running_mean.mul_(1.0 - momentum).add_(mean, alpha=momentum)
            running_var.mul_(1.0 - momentum).add_(m2 / torch.clamp(n - 1.0, min=1.0), alpha=momentum)
        if nbt is not None:
            nbt += 1
        lib = L.lib()
        d = L.SpBnDesc()
        d.N, d.C, d.act, d.training, d.eps, d.momentum = N, C, act, 0, float(eps), float(momentum)
        y = torch.empty_like(x)
        res = residual.contiguous() if residual is not None else None
        mean_c, var_c = mean.contiguous(), var.contiguous()
        d.x, d.gamma, d.beta, d.residual = x.data_ptr(), L.ptr(gamma).value, L.ptr(beta).value, L.ptr(res).value
        d.running_mean, d.running_var, d.y = mean_c.data_ptr(), var_c.data_ptr(), y.data_ptr()
        L.check(lib.vdetr_sp_bn_act_fwd_f32(ctypes.byref(d), L.stream_ptr()), "sp_bn_act_fwd")
        invstd = torch.rsqrt(var + eps)
        ctx.save_for_backward(x, gamma, y, mean_c, invstd, (1.0 / n[:1]))
        ctx.cfg = (act, residual is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        import torch.distributed as dist
        from . import bn_act as BNA
        x, gamma, y, mean, invstd, inv_n = ctx.saved_tensors
        act, has_res = ctx.cfg
        g = dy
        if act == 1:
            g = dy * (y > 0)
        elif act == 2:
            g = dy * torch.where(y > 0, torch.ones_like(y), y + 1.0)
        xhat = (x - mean) * invstd
        loc = torch.stack((g.sum(0), (g * xhat).sum(0)))
        tot = loc.clone()
        dist.all_reduce(tot, group=BNA._sync["group"])
        k = invstd if gamma is None else gamma * invstd
        dx = k * (g - tot[0] * inv_n - xhat * (tot[1] * inv_n)) if ctx.needs_input_grad[0] else None
        return dx, (loc[1] if gamma is not None else None), (loc[0] if gamma is not None else None), \
            (g if has_res and ctx.needs_input_grad[3] else None), None, None, None, None, None, None


def bn_act(x, bn, act=None, residual=None):
    """act(bn(x) + residual) for an ``nn.BatchNorm1d`` module over the rows of x [N, C]: one fused HIP pass on the GPU; any
    other module type (SyncBatchNorm) or a CPU tensor goes through the module itself and plain torch ops."""
    # (sp_bn.hip holds one float4 of a row per thread, 256 threads: up to 1024 channels; momentum=None is a cumulative
    # average, which the kernels do not do: both go through the module)
    fused = x.is_cuda and type(bn) is torch.nn.BatchNorm1d and x.shape[1] % 4 == 0 and 0 < x.shape[1] <= 1024 and x.shape[0] > 0 \
        and not _NO_FUSED_BN and (bn.momentum is not None or not bn.training)
    if not fused:
        from . import bn_act as BNA
        # (with bn_act.set_sync this is still a cross-replica BatchNorm — as tensor expressions — and a rank whose tensor is
        # EMPTY joins its collectives with a count of zero instead of leaving the other ranks waiting)
        y = BNA.batch_norm_module(bn, x, channel_last=True)
        if residual is not None:
            y = y + residual
        return torch.relu(y) if act == "relu" else torch.nn.functional.elu(y) if act == "elu" else y
    training = bn.training or bn.running_mean is None
    momentum = 0.1 if bn.momentum is None else bn.momentum
    if training:
        from . import bn_act as BNA
        if BNA.sync_active():
            return _SpSyncBnActFn.apply(x, bn.weight, bn.bias, residual, bn.running_mean, bn.running_var,
                                        bn.num_batches_tracked if bn.track_running_stats else None, momentum, bn.eps, _ACT[act])
    return _SpBnActFn.apply(x, bn.weight, bn.bias, residual, bn.running_mean, bn.running_var,
                            bn.num_batches_tracked if bn.track_running_stats else None, training, momentum, bn.eps, _ACT[act])


_NO_FUSED_BN = os.environ.get("VDETR_SP_FUSED_BN", "1") == "0"  # A/B switch
_MODE = os.environ.get("VDETR_SP_MODE", "pairs")  # pairs (fused kernels) | plan (batched library GEMMs) | im2col
_IM2COL = _MODE == "im2col" or os.environ.get("VDETR_SP_IM2COL", "0") == "1"  # A/B switch: one dense im2col GEMM per layer instead of the plan


def sparse_conv(feats, weight, nbr, inv, plan=None):
    """feats [Nin, Cin] (any Cin: padded to a multiple of 4 here), weight [K, Cin, Cout], nbr [K, Nout], inv [K, Nin];
    ``plan``: the ConvPlan of (nbr, Nin) if the caller caches it (the coordinate manager does)."""
    cin, cout = feats.shape[1], weight.shape[2]
    if isinstance(plan, PairPlan) or (plan is None and _MODE == "pairs"):
        if plan is None:
            plan = PairPlan(nbr, feats.shape[0])
        # the fused kernels contract over multiples of 16 channels on either side (forward: Cin, input gradient: Cout)
        pi, po = (-cin) % 16, (-cout) % 16
        if pi or po:
            feats = torch.nn.functional.pad(feats, (0, pi))
            weight = torch.nn.functional.pad(weight, (0, po, 0, pi))
        out = _PairsConvFn.apply(feats.contiguous(), weight, plan)
        return out[:, :cout] if po else out
    pad = (-cin) % 4
    if pad:
        feats = torch.nn.functional.pad(feats, (0, pad))
        weight = torch.nn.functional.pad(weight, (0, 0, 0, pad))
    if _IM2COL:
        return _SparseConvFn.apply(feats.contiguous(), weight, nbr, inv)
    if plan is None:
        plan = ConvPlan(nbr, feats.shape[0])
    return _PlannedConvFn.apply(feats.contiguous(), weight, plan)
