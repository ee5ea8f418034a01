"""The five box heads of a decoder stage as three launches, and the learned query-position MLP as one (csrc/heads.hip).

Reference: get_proposal_box_predictions_refine (models/vdetr_transformer.py:244-285: five GenericMLPs on the same features,
models/helpers.py:74-141) and PositionEmbeddingLearned (helpers.py:17-33).  Host side of ``vdetr_heads_fwd_f32`` /
``vdetr_pos_mlp_fwd_f32``: training mode on the GPU only; the launches write every tensor the existing batched backward reads
(vdetr_transformer._DeferredHeads, helpers.DeferredPosEmbedGrads), with the dropout streams of bn_act.py for the same salts, so a
stage computes the same values on either path.  ``VDETR_HEADS_FUSED=0`` keeps the one-launch-per-op path (A/B runs, parity tests).
No CPU path.
"""
import ctypes
import os

import torch

from . import _lib as L
from . import attention as A
from . import bn_act as BNA

C = 256
FUSED = os.environ.get("VDETR_HEADS_FUSED", "1") != "0"
FUSED_POS = os.environ.get("VDETR_POS_FUSED", "1") != "0"

# ---- W^T images of the [256, 256] weights the forward launches read (the same transposer as rowblock.py: one launch) ---------
_tables = {}    # (source pointers) -> (device table of source pointers, [n, 256, 256] images)
_current = {}   # data_ptr of a source weight -> its image [256, 256] (rewritten by the last refresh that listed it)


def _ok_weight(w):
    return w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() and w.numel() == C * C and w.data_ptr() % 16 == 0


def refresh(weights):
    """Rewrite the W^T images of `weights` ([256, 256(, 1)] parameters) in ONE launch; consecutive entries get consecutive images.
    Weights change in place at every optimiser step: the decoder calls this once per forward for every stage's heads."""
    weights = list(weights)
    ptrs = tuple(w.data_ptr() for w in weights)
    dev = weights[0].device
    ent = _tables.get(ptrs)
    if ent is None or ent[1].device != dev:
        for w in weights:
            if not _ok_weight(w):
                raise RuntimeError("heads.refresh: weights must be contiguous 16-B aligned fp32 [256, 256] tensors on the GPU")
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("heads.refresh: the weight pointers changed inside a stream capture (run one step eagerly first)")
        table = torch.tensor(ptrs, dtype=torch.int64).to(dev)
        ent = (table, torch.empty((len(weights), C, C), dtype=torch.float32, device=dev))
        if len(_tables) > 64:
            _tables.clear()
            _current.clear()
        _tables[ptrs] = ent
    L.check(L.lib().vdetr_rb_transpose_f32(ent[0].data_ptr(), ent[1].data_ptr(), len(weights), L.stream_ptr()), "rb_transpose")
    for i, p in enumerate(ptrs):
        _current[p] = ent[1][i]
    return ent[1]


def _images(weights, fresh):
    """[n, 256, 256] images of consecutive `weights`: those a decoder-level refresh left for this forward (`fresh`), else rewritten"""
    imgs = [_current.get(w.data_ptr()) for w in weights] if fresh else None
    if imgs is None or any(i is None for i in imgs) or any(
            imgs[k + 1].data_ptr() - imgs[k].data_ptr() != C * C * 4 for k in range(len(imgs) - 1)):
        return refresh(weights)
    return imgs[0]


_fresh = {"on": False}


def decoder_refresh(decoder):
    """One transposer launch for every image the decoder's forward will read: per stage the five heads' first and second layers,
    per layer the position MLP's second convolution.  Returns False (and leaves the fused paths to rewrite their own) when the
    modules are not the shapes the launches are built for."""
    _fresh["on"] = False
    no_pending_pos("heads.decoder_refresh")
    if not (FUSED or FUSED_POS):
        return False
    ws = []
    stages = list(range(len(decoder.mlp_heads))) if decoder.mlp_sep else [0]
    try:
        if FUSED:
            for st in stages:
                heads = decoder.mlp_heads[st] if decoder.mlp_sep else decoder.mlp_heads
                if not decoder._batchable(heads):
                    continue
                Ls = [heads[n].layers for n in decoder._HEAD_NAMES]
                ws += [l[0].weight for l in Ls] + [l[4].weight for l in Ls]
        if FUSED_POS:
            ws += [m.position_embedding_head[3].weight for m in decoder.query_pos_projection]
        fl = getattr(decoder, "first_layer", None)  # (rowblock.ffn0: the FFN layer in front of the decoder reads two images too)
        if fl is not None and hasattr(fl, "linear1") and hasattr(fl, "linear2"):
            ws += [fl.linear1.weight, fl.linear2.weight]
    except (AttributeError, IndexError, KeyError):
        return False
    ws = [w for w in ws if _ok_weight(w)]
    if not ws:
        return False
    refresh(ws)
    _fresh["on"] = True
    return True


def heads_usable(feats_seq, L5, rows):
    """feats_seq [N, B, 256] (the stage's features, sequence-first); L5: the five heads' layer lists; rows: slab height"""
    if not (FUSED and feats_seq.is_cuda and feats_seq.dtype == torch.float32 and feats_seq.dim() == 3 and feats_seq.is_contiguous()
            and feats_seq.shape[2] == C and feats_seq.shape[0] % 32 == 0 and feats_seq.data_ptr() % 16 == 0):
        return False
    if len(L5) > 8 or rows > 32 or BNA.sync_active():
        return False
    bn0 = L5[0][1]
    for l in L5:
        if not (_ok_weight(l[0].weight) and _ok_weight(l[4].weight) and l[0].weight.shape[0] == C and l[4].weight.shape[0] == C):
            return False
        for bn in (l[1], l[5]):
            if bn.eps != bn0.eps or bn.momentum != bn0.momentum or bn.momentum is None:
                return False
        if l[3].p != L5[0][3].p or l[7].p != L5[0][7].p:
            return False
    return True


def heads_forward(feats_seq, L5, params, stats, salts, rows, h1_out, h2_out):
    """The three launches.  params = (g1, b1, g2, b2, w3, b3) tensors as the batched path builds them; stats = (rm1, rv1, rm2, rv2)
    aliases of the adjacent running statistics; salts of the two dropout streams; h1_out / h2_out [B, G*256, N] receive the hidden
    activations.  Returns (y [B, G, rows, N], bn1 record, bn2 record) — the records of bn_act.forward_record."""
    N, B, _ = feats_seq.shape
    G = len(L5)
    dev = feats_seq.device
    g1, b1, g2, b2, w3, b3 = (t.detach().contiguous() for t in params)
    p1, p2 = float(L5[0][3].p), float(L5[0][7].p)
    rng = None
    if p1 > 0.0 or p2 > 0.0:
        rng = A.current_rng(dev)
        if rng is None:
            rng = A.begin_step(dev)
    w_imgs = _images([l[0].weight for l in L5] + [l[4].weight for l in L5], _fresh["on"])
    pre1 = torch.empty((B, G * C, N), dtype=torch.float32, device=dev)
    pre2 = torch.empty_like(pre1)
    sm = torch.empty((4, G * C), dtype=torch.float32, device=dev)
    y = torch.empty((B, G, rows, N), dtype=torch.float32, device=dev)
    nbytes = L.lib().vdetr_heads_workspace_bytes(B, N, G)
    ws = L.workspace(nbytes, dev)
    d = L.HeadsDesc()
    d.B, d.N, d.G, d.rows = B, N, G, rows
    d.x = feats_seq.data_ptr()
    d.w1t, d.w2t = w_imgs.data_ptr(), w_imgs.data_ptr() + G * C * C * 4
    d.w3, d.b3 = w3.data_ptr(), b3.data_ptr()
    d.gamma1, d.beta1, d.gamma2, d.beta2 = g1.data_ptr(), b1.data_ptr(), g2.data_ptr(), b2.data_ptr()
    d.running_mean1, d.running_var1, d.running_mean2, d.running_var2 = (t.data_ptr() for t in stats)
    for g, l in enumerate(L5):
        d.counters1[g] = l[1].num_batches_tracked.data_ptr()
        d.counters2[g] = l[5].num_batches_tracked.data_ptr()
    bn0 = L5[0][1]
    d.eps, d.momentum, d.p1, d.p2 = float(bn0.eps), float(bn0.momentum), p1, p2
    d.salt1, d.salt2 = int(salts[0]) & 0xFFFFFFFFFFFFFFFF, int(salts[1]) & 0xFFFFFFFFFFFFFFFF
    d.rng_state = rng.data_ptr() if rng is not None else None
    assert h1_out.is_contiguous() and h2_out.is_contiguous() and h1_out.shape == pre1.shape and h2_out.shape == pre1.shape
    d.pre1, d.h1, d.pre2, d.h2 = pre1.data_ptr(), h1_out.data_ptr(), pre2.data_ptr(), h2_out.data_ptr()
    d.save_mean1, d.save_invstd1, d.save_mean2, d.save_invstd2 = (sm[i].data_ptr() for i in range(4))
    d.y = y.data_ptr()
    d.workspace = ws.data_ptr()
    L.check(L.lib().vdetr_heads_fwd_f32(ctypes.byref(d), L.stream_ptr()), "heads_fwd")
    cfg = float(bn0.eps), float(bn0.momentum)
    rec1 = (pre1, g1, b1, sm[0], sm[1], rng if p1 > 0 else None, cfg + (p1, int(salts[0])), None)
    rec2 = (pre2, g2, b2, sm[2], sm[3], rng if p2 > 0 else None, cfg + (p2, int(salts[1])), None)
    return y, rec1, rec2


def pos_mlp_usable(module, x_tok):
    """module: helpers.PositionEmbeddingLearned; x_tok [B, N, cin] coordinates"""
    head = module.position_embedding_head
    bn = head[1]
    return bool(FUSED_POS and x_tok.is_cuda and x_tok.dtype == torch.float32 and x_tok.dim() == 3 and x_tok.shape[2] <= 8
                and x_tok.shape[1] % 16 == 0 and head[0].weight.shape[0] == C and _ok_weight(head[3].weight)
                and type(bn) is torch.nn.BatchNorm1d and bn.momentum is not None and bn.track_running_stats
                and not BNA.sync_active())


# Position MLPs whose launch is left to the consumer of their output (vdetr_rb_qkv_pos_f32: the decoder layer's q / k / v projection
# computes the position rows on its way in): data_ptr of `out` -> (descriptor, tensors it points at).  The decoder turns this on for
# the layers it runs through rowblock.py; a layer that takes another path calls materialize_pos() first.  `out`, the hidden
# activations and the statistics are NOT valid until one of the two has run.
_pending_pos = {}
POS_IN_QKV = os.environ.get("VDETR_POS_IN_QKV", "1") != "0"
_lazy_pos = {"on": False}


def lazy_pos(on):
    """from here on position MLPs are left to rowblock.qkv (True) or launched at once (False); returns the previous state"""
    prev, _lazy_pos["on"] = _lazy_pos["on"], bool(on) and POS_IN_QKV
    return prev


def take_pending_pos(pos):
    """the descriptor of a position MLP that was left to the consumer of `pos` (None: `pos` is final)"""
    return _pending_pos.pop(pos.data_ptr(), None) if (_pending_pos and pos is not None) else None


def materialize_pos(pos):
    """launch the position MLP behind `pos` now, if it was left to a consumer that will not run"""
    rec = take_pending_pos(pos)
    if rec is not None:
        L.check(L.lib().vdetr_pos_mlp_fwd_f32(ctypes.byref(rec[0]), L.stream_ptr()), "pos_mlp_fwd")


def no_pending_pos(where):
    if _pending_pos:
        _pending_pos.clear()
        raise RuntimeError(f"{where}: a position MLP was left to rowblock.qkv, which never ran on its output")


def pos_mlp_forward(module, x_tok):
    """One launch — or none yet: with lazy_pos(True) the launch is left to rowblock.qkv (or materialize_pos) and only the outputs are
    allocated.  Returns (out [N, B, 256] dense, hidden activations [B, 256, N], the BatchNorm record of bn_act.forward_record)."""
    conv1, bn, _, conv2 = module.position_embedding_head
    x_tok = x_tok.detach().contiguous()
    B, N, cin = x_tok.shape
    dev = x_tok.device
    w2t = _images([conv2.weight], _fresh["on"])
    hpre = torch.empty((B, C, N), dtype=torch.float32, device=dev)
    hact = torch.empty_like(hpre)
    sm = torch.empty((2, C), dtype=torch.float32, device=dev)
    out = torch.empty((N, B, C), dtype=torch.float32, device=dev)
    g, b = bn.weight.detach().contiguous(), bn.bias.detach().contiguous()
    w1 = conv1.weight.detach().reshape(C, cin).contiguous()
    d = L.PosMlpDesc()
    d.B, d.N, d.cin = B, N, cin
    d.x, d.w1 = x_tok.data_ptr(), w1.data_ptr()
    d.b1 = conv1.bias.data_ptr() if conv1.bias is not None else None
    d.gamma, d.beta = g.data_ptr(), b.data_ptr()
    d.running_mean, d.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
    d.counter = bn.num_batches_tracked.data_ptr()
    d.eps, d.momentum = float(bn.eps), float(bn.momentum)
    d.w2t = w2t.data_ptr()
    d.b2 = conv2.bias.data_ptr() if conv2.bias is not None else None
    d.hpre, d.hact, d.save_mean, d.save_invstd, d.out = hpre.data_ptr(), hact.data_ptr(), sm[0].data_ptr(), sm[1].data_ptr(), out.data_ptr()
    if _lazy_pos["on"] and B * N % 16 == 0 and (B == 1 or N % 4 == 0):
        _pending_pos[out.data_ptr()] = (d, (x_tok, w1, g, b, w2t, hpre, hact, sm, out))
    else:
        L.check(L.lib().vdetr_pos_mlp_fwd_f32(ctypes.byref(d), L.stream_ptr()), "pos_mlp_fwd")
    rec = (hpre, g, b, sm[0], sm[1], None, (float(bn.eps), float(bn.momentum), 0.0, 0), None)
    return out, hact, rec, (x_tok, w1)
