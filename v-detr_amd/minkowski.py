"""The slice of MinkowskiEngine's module API that the reference's backbone uses (models/mink_resnet.py:4-5,38-84;
models/model_vdetr.py:8,139-185,248-280), on the MI355X sparse primitives of ``sparse_ops``.

Same class names, constructor arguments, parameter names / shapes (``MinkowskiConvolution.kernel [K, Cin, Cout]``,
``MinkowskiBatchNorm.bn.*``) so that the reference's ``pre_encoder.* / up_block_*.* / out_block_0.*`` checkpoint keys load.
MinkowskiEngine is NOT under /root/reference (un-vendored): semantics follow its published operator and documentation —
  * kernel offsets of a hypercubic region: odd sizes centred (-1, 0, 1), even sizes 0 .. k-1, in units of the INPUT tensor
    stride; kernel index = ix + k * (iy + k * iz) (first spatial dimension fastest);
  * a strided convolution writes the sites floor(c / s_out) * s_out of its input sites (tensor stride multiplied);
  * a transposed convolution with stride 2 divides the tensor stride; ``MinkowskiConvolutionTranspose`` writes the sites
    that already exist at that stride in the coordinate manager, ``MinkowskiGenerativeConvolutionTranspose`` generates
    every child site;
  * duplicated input coordinates keep one feature row (here: the first in input order), sites are held in key order.
The ORDER of the sites of a MinkowskiEngine tensor is an implementation detail of its hash map; here it is ascending
(batch, x, y, z).  Parity against the MinkowskiEngine binary is unpinned; ``oracle/sparse_oracle.py`` pins the arithmetic
against torch's dense convolutions.
"""
import math
import threading

import torch
import torch.nn as nn

from . import sparse_ops as S


# Geometry-only pass (ModelVDETR.prepare_geometry): convolutions build their sites / kernel maps / pair lists and hand on an
# UNINITIALISED feature table of the right shape; every other layer is the identity.  Nothing is computed on features.
# The switch is per THREAD: a loader thread prepares the next scene while the training thread runs the model.
_tls = threading.local()


def _geometry_only():
    return getattr(_tls, "geometry_only", False)


class geometry_only:
    def __enter__(self):
        self._prev, _tls.geometry_only = _geometry_only(), True

    def __exit__(self, *exc):
        _tls.geometry_only = self._prev


def _region_offsets(kernel_size):
    r = range(-(kernel_size // 2), kernel_size // 2 + 1) if kernel_size % 2 else range(kernel_size)
    return [(x, y, z) for z in r for y in r for x in r]  # x fastest


class CoordinateManager:
    """Sites per tensor stride + cached kernel maps of one batch (geometry only: no gradients flow through it)."""

    def __init__(self, device):
        self.device = device
        self.keys = {}    # tensor_stride -> canonical sorted int64 keys of that stride
        self.maps = {}    # (in key set, out key set, strides, kernel_size, transposed) -> (nbr [K,Nout], inv [K,Nin])

    def insert_points(self, coordinates):
        """raw integer coordinates [N,4] -> the sites of tensor stride 1 (+ ``unique_index``: the input row kept per site:
        the first occurrence in input order; MinkowskiEngine: RANDOM_SUBSAMPLE)"""
        keys_in = S.pack_keys(coordinates)
        keys, inverse = torch.unique(keys_in, return_inverse=True)
        first = torch.full((keys.shape[0],), keys_in.shape[0], dtype=torch.int64, device=keys.device)
        first.scatter_reduce_(0, inverse, torch.arange(keys_in.shape[0], device=keys.device), reduce="amin")
        self.keys[1] = keys
        self.unique_index = first

    @staticmethod
    def _strided_keys(keys, new_ts):
        c = S.unpack_keys(keys).to(torch.int64)
        c[:, 1:] = torch.div(c[:, 1:], new_ts, rounding_mode="floor") * new_ts
        return torch.unique(S.pack_keys(c, check=False))

    def strided(self, keys, ts, new_ts):
        """sites of a strided convolution's output.  The sites derived from the canonical set of stride `ts` become the
        canonical set of stride `new_ts` (what later transposed convolutions / skip connections land on)."""
        if keys is self.keys.get(ts):
            if new_ts not in self.keys:
                self.keys[new_ts] = self._strided_keys(keys, new_ts)
            return self.keys[new_ts]
        return self._strided_keys(keys, new_ts)

    @staticmethod
    def generated(keys, new_ts, kernel_size):
        """children of every site (generative transposed convolution): a NEW key set at stride `new_ts`"""
        c = S.unpack_keys(keys).to(torch.int64)
        off = torch.tensor(_region_offsets(kernel_size), dtype=torch.int64, device=c.device) * new_ts
        child = c[:, None, :].repeat(1, off.shape[0], 1)
        child[:, :, 1:] += off[None]
        return torch.unique(S.pack_keys(child.reshape(-1, 4)))

    def kernel_map(self, in_keys, out_keys, in_ts, out_ts, kernel_size, transposed):
        key = (in_keys.data_ptr(), out_keys.data_ptr(), in_ts, out_ts, kernel_size, transposed)
        if key not in self.maps:
            reg = torch.tensor(_region_offsets(kernel_size), dtype=torch.int32)
            # convolution: output site u reads u + offset * in_stride; transposed: the output (fine) site v = u + offset *
            # out_stride of the input (coarse) site u reads v - offset * out_stride
            offsets = (reg * in_ts if not transposed else -reg * out_ts).to(self.device)
            nbr = S.kernel_map(in_keys, out_keys, offsets)
            pairs = not S._IM2COL and S._MODE == "pairs"
            inv = None if pairs else S.inverse_map(nbr, in_keys.shape[0])  # (the pair lists carry their own inverse: islot)
            plan = None if S._IM2COL else (S.PairPlan if pairs else S.ConvPlan)(nbr, in_keys.shape[0])
            self.maps[key] = (nbr, inv, plan, in_keys, out_keys)  # key tensors kept alive
        return self.maps[key][:3]

    def scene_counts(self, keys):
        """sites per batch element of a key set, as a host list.  Cached per key set: it is geometry, and reading it back is a
        host synchronisation that must not sit between the backbone's forward launches and what follows them (measured: the
        host then waits for the whole forward pass, 12 ms per step at 40k points, before it can enqueue the decoder)."""
        cache = self.__dict__.setdefault("_scene_counts", {})
        key = keys.data_ptr()
        if key not in cache:
            b = keys >> 48
            nb = int(b.max()) + 1 if b.numel() else 0
            cache[key] = (torch.bincount(b, minlength=nb).tolist(), keys)  # the key tensor is kept alive with its entry
        return cache[key][0]

    def device_tensors(self):
        """every device tensor this manager holds, directly or through its cached plans (sites, kernel maps, pair lists,
        tile tables ...)"""
        seen, stack, out = set(), [self], []
        while stack:
            o = stack.pop()
            if id(o) in seen:
                continue
            seen.add(id(o))
            if torch.is_tensor(o):
                if o.is_cuda:
                    out.append(o)
            elif isinstance(o, dict):
                stack.extend(o.values())
            elif isinstance(o, (list, tuple, set)):
                stack.extend(o)
            elif hasattr(o, "__dict__") and not isinstance(o, (type, torch.nn.Module)):
                stack.extend(vars(o).values())
        return out

    def use_on(self, stream=None):
        """Tell the caching allocator that `stream` (default: the current one) reads this manager's tensors.  A manager
        built by ``prepare_geometry`` on a loader stream is consumed by forward and backward kernels of the training
        stream; without this, dropping the manager lets the allocator hand its blocks to the loader stream's next scene
        while those kernels are still queued.  Idempotent per stream; plans added later are picked up on the next call."""
        stream = stream or torch.cuda.current_stream()
        if torch.cuda.is_current_stream_capturing():
            return self
        # keyed by storage + stream (views of one buffer are one allocator block), and the memo HOLDS a tensor of every
        # recorded storage: while it does, the address cannot be handed to a new allocation that would then be skipped
        # and have its block recycled under the training stream's kernels
        done = self.__dict__.setdefault("_recorded", {})
        for t in self.device_tensors():
            key = (t.untyped_storage().data_ptr(), stream.cuda_stream)
            if key not in done:
                t.record_stream(stream)
                done[key] = t
        return self

    def finalize(self):
        """wait for the pair counts of every plan built so far (one synchronisation for the whole scene)"""
        for entry in self.maps.values():
            if hasattr(entry[2], "finalize"):
                entry[2].finalize()
        return self


class SparseTensor:
    """features [N, C] on the sites ``coordinates`` [N, 4] = (batch, x, y, z) (MinkowskiEngine's SparseTensor)."""

    def __init__(self, features, coordinates=None, tensor_stride=1, coordinate_manager=None, keys=None):
        if coordinate_manager is None:
            assert coordinates is not None and tensor_stride == 1
            coordinate_manager = CoordinateManager(features.device)
            coordinate_manager.insert_points(coordinates)
            features = features[coordinate_manager.unique_index]  # one row per site, in site (key) order
        self.F = features
        self.tensor_stride = tensor_stride
        self.coordinate_manager = coordinate_manager
        self.keys = keys if keys is not None else coordinate_manager.keys[tensor_stride]

    @property
    def C(self):
        return S.unpack_keys(self.keys)

    @property
    def device(self):
        return self.F.device

    def _like(self, features):
        return SparseTensor(features, tensor_stride=self.tensor_stride, coordinate_manager=self.coordinate_manager, keys=self.keys)

    def __add__(self, other):
        assert self.keys is other.keys or torch.equal(self.keys, other.keys), "sparse tensors on different coordinate maps"
        return self if _geometry_only() else self._like(self.F + other.F)

    def decomposed(self):
        """per batch element: (coordinates [n,3] int32, features [n,C]) — key order keeps the scenes contiguous"""
        counts = self.coordinate_manager.scene_counts(self.keys)
        coords, out, s = self.C, [], 0
        for n in counts:
            out.append((coords[s:s + n, 1:], self.F[s:s + n]))
            s += n
        return out


class MinkowskiConvolution(nn.Module):
    """ME.MinkowskiConvolution(in, out, kernel_size, stride=1, dilation=1, bias=False, dimension=3)."""
    transposed, generative = False, False

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False, dimension=3):
        super().__init__()
        assert dimension == 3 and dilation == 1 and kernel_size > 0
        self.in_channels, self.out_channels, self.kernel_size, self.stride = in_channels, out_channels, kernel_size, stride
        kv = kernel_size ** 3
        # MinkowskiEngine keeps a plain [in, out] matrix only for the 1x1x1 stride-1 case
        shape = (in_channels, out_channels) if (kv == 1 and stride == 1) else (kv, in_channels, out_channels)
        self.kernel = nn.Parameter(torch.empty(shape))
        self.bias = nn.Parameter(torch.zeros(1, out_channels)) if bias else None
        stdv = 1.0 / math.sqrt((out_channels if self.transposed else in_channels) * kv)
        with torch.no_grad():
            self.kernel.uniform_(-stdv, stdv)

    def _out_stride(self, ts):
        if not self.transposed:
            return ts * self.stride
        assert ts % self.stride == 0, "transposed convolution below tensor stride 1"
        return ts // self.stride

    def forward(self, x):
        cm, ts = x.coordinate_manager, x.tensor_stride
        out_ts = self._out_stride(ts)
        w = self.kernel if self.kernel.dim() == 3 else self.kernel[None]
        if self.generative:
            out_keys = cm.generated(x.keys, out_ts, self.kernel_size)
        elif self.transposed:
            assert out_ts in cm.keys, "MinkowskiConvolutionTranspose needs existing sites at the output stride"
            out_keys = cm.keys[out_ts]
        else:
            out_keys = x.keys if out_ts == ts else cm.strided(x.keys, ts, out_ts)
        nbr, inv, plan = cm.kernel_map(x.keys, out_keys, ts, out_ts, self.kernel_size, self.transposed)
        if _geometry_only():
            if hasattr(plan, "request_wgrad"):  # the weight-gradient chunk table of this layer's width: built with the geometry
                plan.request_wgrad(w.shape[1], w.shape[2])
            return SparseTensor(x.F.new_empty((out_keys.shape[0], self.out_channels)), tensor_stride=out_ts,
                                coordinate_manager=cm, keys=out_keys)
        f = S.sparse_conv(x.F, w, nbr, inv, plan)
        if self.bias is not None:
            f = f + self.bias
        return SparseTensor(f, tensor_stride=out_ts, coordinate_manager=cm, keys=out_keys)


class MinkowskiConvolutionTranspose(MinkowskiConvolution):
    transposed = True


class MinkowskiGenerativeConvolutionTranspose(MinkowskiConvolution):
    transposed, generative = True, True


class MinkowskiBatchNorm(nn.Module):
    """BatchNorm1d over the feature rows of all sites of the batch (ME.MinkowskiBatchNorm: parameters under ``.bn``)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine,
                                 track_running_stats=track_running_stats)

    fused_act = None  # "elu" / "relu": the activation module that follows in an nn.Sequential is folded into this layer

    def forward(self, x, act=None, residual=None):
        if _geometry_only():
            return x
        act = act or self.fused_act
        out = x._like(S.bn_act(x.F, self.bn, act, None if residual is None else residual.F))
        out.applied_act = act
        return out


class MinkowskiInstanceNorm(nn.Module):
    """per batch element, per channel normalisation over the element's sites, with affine parameters"""

    def __init__(self, num_features, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(1, num_features))
        self.bias = nn.Parameter(torch.zeros(1, num_features))
        self.eps = eps

    def forward(self, x):
        if _geometry_only():
            return x
        b = x.keys >> 48
        nb = int(b.max()) + 1
        cnt = torch.bincount(b, minlength=nb).clamp(min=1).to(x.F.dtype)[:, None]
        mean = torch.zeros((nb, x.F.shape[1]), dtype=x.F.dtype, device=x.F.device).index_add_(0, b, x.F) / cnt
        d = x.F - mean[b]
        var = torch.zeros_like(mean).index_add_(0, b, d * d) / cnt
        return x._like(d / torch.sqrt(var[b] + self.eps) * self.weight + self.bias)


class _Pointwise(nn.Module):
    kind = None

    def forward(self, x):
        if _geometry_only():
            return x
        if getattr(x, "applied_act", None) == self.kind:  # already applied by the fused BatchNorm in front (fuse_activations)
            return x._like(x.F)
        return x._like(self.fn(x.F))


class MinkowskiReLU(_Pointwise):
    kind = "relu"

    def __init__(self, inplace=False):
        super().__init__()
        self.fn = nn.ReLU()


class MinkowskiELU(_Pointwise):
    kind = "elu"

    def __init__(self, alpha=1.0, inplace=False):
        super().__init__()
        self.fn = nn.ELU(alpha)


class BasicBlock(nn.Module):
    """MinkowskiEngine.modules.resnet_block.BasicBlock (used by models/mink_resnet.py:5,22-23)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, dimension=-1):
        super().__init__()
        assert dimension > 0
        self.conv1 = MinkowskiConvolution(inplanes, planes, kernel_size=3, stride=stride, dilation=dilation, dimension=dimension)
        self.norm1 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv2 = MinkowskiConvolution(planes, planes, kernel_size=3, stride=1, dilation=dilation, dimension=dimension)
        self.norm2 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.relu = MinkowskiReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        residual = x if self.downsample is None else self.downsample(x)
        out = self.norm1(self.conv1(x), act="relu")
        return self.norm2(self.conv2(out), act="relu", residual=residual)  # relu(bn(conv) + residual) in one pass


class Bottleneck(nn.Module):
    """MinkowskiEngine.modules.resnet_block.Bottleneck (depth 50 / 101 / 152)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, dimension=-1):
        super().__init__()
        assert dimension > 0
        self.conv1 = MinkowskiConvolution(inplanes, planes, kernel_size=1, dimension=dimension)
        self.norm1 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv2 = MinkowskiConvolution(planes, planes, kernel_size=3, stride=stride, dilation=dilation, dimension=dimension)
        self.norm2 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv3 = MinkowskiConvolution(planes, planes * self.expansion, kernel_size=1, dimension=dimension)
        self.norm3 = MinkowskiBatchNorm(planes * self.expansion, momentum=bn_momentum)
        self.relu = MinkowskiReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        residual = x if self.downsample is None else self.downsample(x)
        out = self.norm1(self.conv1(x), act="relu")
        out = self.norm2(self.conv2(out), act="relu")
        return self.norm3(self.conv3(out), act="relu", residual=residual)


def fuse_activations(module):
    """In every nn.Sequential, fold a MinkowskiReLU / MinkowskiELU (alpha 1) that directly follows a MinkowskiBatchNorm into
    that layer's fused kernel (the activation module stays in place — state-dict indices are unchanged — and passes
    tensors that already carry its activation through)."""
    for m in module.modules():
        if isinstance(m, nn.Sequential):
            mods = list(m)
            for a, b in zip(mods, mods[1:]):
                if isinstance(a, MinkowskiBatchNorm) and isinstance(b, _Pointwise) and (b.kind == "relu" or b.fn.alpha == 1.0):
                    a.fused_act = b.kind
    return module


def kaiming_normal_(tensor, a=0, mode="fan_in", nonlinearity="leaky_relu"):
    """ME.utils.kaiming_normal_ on a [K, Cin, Cout] kernel: fan_in = K * Cin, fan_out = K * Cout."""
    if tensor.dim() == 3:
        fan = tensor.shape[0] * (tensor.shape[1] if mode == "fan_in" else tensor.shape[2])
    else:
        fan = tensor.shape[0] if mode == "fan_in" else tensor.shape[1]
    std = nn.init.calculate_gain(nonlinearity, a) / math.sqrt(fan)
    with torch.no_grad():
        return tensor.normal_(0, std)


def batch_sparse_collate(data):
    """ME.utils.batch_sparse_collate([(coords_i [n_i,3] float or int, feats_i [n_i,C]), ...]) -> (coordinates [N,4] int32 with
    the batch index in column 0, features [N,C]).  Float coordinates are floored (model_vdetr.py:252-259 passes p / voxel_size)."""
    coords, feats = [], []
    for b, (c, f) in enumerate(data):
        c = torch.floor(c) if c.is_floating_point() else c
        c = c.to(torch.int32)
        coords.append(torch.cat((torch.full((c.shape[0], 1), b, dtype=torch.int32, device=c.device), c), dim=1))
        feats.append(f)
    return torch.cat(coords), torch.cat(feats)
