"""Post-backbone part of the reference's ``models/model_vdetr.py``: FPS + gather of the backbone's voxel
features, projection, first-stage anchors and the decoder call, with the same ``forward(inputs)`` contract and the
same state-dict keys (``encoder_to_decoder_projection.layers.{0,1}``, ``decoder.*``).

``pre_encoder`` is either
  * a ``mink_resnet.MinkResNet``: the model then owns the reference's FPN neck (``up_block_{1,2,3}``,
    ``out_block_0``; model_vdetr.py:139-185) and runs ``run_encoder`` as the reference does (:248-280: voxelise, ResNet34,
    top-down FPN, out block), on the sparse primitives of ``vdetr_amd.minkowski`` instead of MinkowskiEngine; or
  * any callable that returns what the reference takes from the backbone at model_vdetr.py:279-280 — per scene, the voxel
    coordinates ``xyz [n,3]`` (= out.C[:,1:] * voxel_size) and features ``feats [n,C]`` (= out.F) — e.g. ``TensorBackbone``
    (benchmarks and tests of the post-backbone hot path).
"""
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import box_decode
from . import minkowski as ME
from . import pointnet2_utils
from .dataset_config import ScannetDatasetConfig
from .helpers import GenericMLP
from .mink_resnet import MinkResNet
from .position_embedding import PositionEmbeddingCoordsSine
from .vdetr_transformer import FFNLayer, GlobalDecoderLayer, TransformerDecoder


class FPSModule(nn.Module):
    """Farthest point sampling + gather of coordinates and features (model_vdetr.py:18-34)."""

    def forward(self, xyz, features, num_proposal, sample_inds=None):
        """xyz (B,K,3), features (B,C,K) -> new_xyz (B,M,3), new_features (B,C,M), sample_inds (B,M) int32.
        ``sample_inds`` may be handed in when it was computed ahead of time (``ModelVDETR.sample_indices``)."""
        if sample_inds is None:
            sample_inds = pointnet2_utils.furthest_point_sample(xyz, num_proposal)
        xyz_flipped = xyz.transpose(1, 2).contiguous()
        new_xyz = pointnet2_utils.gather_operation(xyz_flipped, sample_inds).transpose(1, 2).contiguous()
        new_features = pointnet2_utils.gather_operation(features, sample_inds).contiguous()
        return new_xyz, new_features, sample_inds


class TensorBackbone(nn.Module):
    """Backbone provider that hands back tensors stored in the batch itself: ``inputs["backbone_xyz"]`` /
    ``inputs["backbone_features"]`` are lists with one ``[n_i,3]`` / ``[n_i,C]`` tensor per scene.  Used by the
    benchmarks and tests in place of the sparse-conv backbone."""

    def forward(self, inputs):
        return list(zip(inputs["backbone_xyz"], inputs["backbone_features"]))


def convert_unnorm2norm(xyz_unnorm, point_cloud_dims, with_offset=True):
    """model_vdetr.py:383-390"""
    scene_size = point_cloud_dims[1] - point_cloud_dims[0]
    offset = point_cloud_dims[0].unsqueeze(1) if with_offset else 0
    return (xyz_unnorm - offset) / scene_size.unsqueeze(1)


class ModelVDETR(nn.Module):
    """forward(inputs) with inputs = {"point_clouds": ..., "point_cloud_dims_min": [B,3], "point_cloud_dims_max":
    [B,3]} -> {"outputs", "aux_outputs", "seed_inds", "seed_xyz", "enc_outputs"} (model_vdetr.py:328-381)."""

    def __init__(self, pre_encoder, encoder, decoder, dataset_config, encoder_dim=256, decoder_dim=256,
                 num_queries=1024, querypos_mlp=False, minkowski=False, inplane=64, num_stages=4, voxel_size=0.01,
                 npoint=2048, use_fpn=False, layer_idx=-1, proj_nohid=False, woexpand_conv=False, args=None):
        super().__init__()
        self.pre_encoder = pre_encoder
        self.encoder = encoder  # always None: V-DETR removed the 3DETR encoder (model_vdetr.py:452)
        self.querypos_mlp = querypos_mlp
        self.voxel_size = voxel_size
        self.npoint = npoint
        self.use_fpn, self.layer_idx, self.num_stages, self.woexpand_conv = use_fpn, layer_idx, num_stages, woexpand_conv
        self.use_color = bool(getattr(args, "use_color", False))
        self.xyz_color = bool(getattr(args, "xyz_color", False))
        self.sparse_backbone = isinstance(pre_encoder, MinkResNet)
        if self.sparse_backbone:
            depth = int(getattr(args, "depth", 34))
            channels = [(4 if depth > 34 else 1) * inplane * 2 ** i for i in range(num_stages)]
            self._init_fpn_layers(channels, encoder_dim)
        # reads args.random_fps with a default: the reference never defines the flag
        self.random_fps = bool(getattr(args, "random_fps", False))
        self.fps_module = FPSModule()
        hidden_dims = [] if (encoder is None and proj_nohid) else [encoder_dim]
        self.encoder_to_decoder_projection = GenericMLP(
            input_dim=encoder_dim, hidden_dims=hidden_dims, output_dim=decoder_dim, norm_fn_name="bn1d",
            activation="relu", use_conv=True, output_use_activation=True, output_use_norm=True,
            output_use_bias=False)
        if not self.querypos_mlp:
            self.pos_embedding = PositionEmbeddingCoordsSine(d_pos=decoder_dim, pos_type="fourier", normalize=True)
            self.query_projection = GenericMLP(input_dim=decoder_dim, hidden_dims=[decoder_dim],
                                               output_dim=decoder_dim, use_conv=True, output_use_activation=True,
                                               hidden_use_bias=True)
        self.decoder = decoder
        self.num_queries = num_queries
        self.dataset_config = dataset_config
        self.hard_anchor = bool(getattr(args, "hard_anchor", False))

    # ---- FPN neck of the sparse backbone (model_vdetr.py:139-185) -----------------------------------------------------
    @staticmethod
    def _make_block(in_channels, out_channels):
        return nn.Sequential(ME.MinkowskiConvolution(in_channels, out_channels, kernel_size=3, dimension=3),
                             ME.MinkowskiBatchNorm(out_channels), ME.MinkowskiELU())

    @staticmethod
    def _make_up_block(in_channels, out_channels, woexpand_conv):
        up = ME.MinkowskiConvolutionTranspose if woexpand_conv else ME.MinkowskiGenerativeConvolutionTranspose
        return nn.Sequential(up(in_channels, out_channels, kernel_size=2, stride=2, dimension=3),
                             ME.MinkowskiBatchNorm(out_channels), ME.MinkowskiELU(),
                             ME.MinkowskiConvolution(out_channels, out_channels, kernel_size=3, dimension=3),
                             ME.MinkowskiBatchNorm(out_channels), ME.MinkowskiELU())

    def _init_fpn_layers(self, in_channels, out_channels):
        if self.use_fpn:
            for i in range(self.layer_idx + 1, len(in_channels)):
                if i > 0:
                    setattr(self, f"up_block_{i}", self._make_up_block(in_channels[i], in_channels[i - 1], self.woexpand_conv))
        setattr(self, f"out_block_{self.layer_idx}", self._make_block(in_channels[self.layer_idx], out_channels))
        for m in self.modules():  # the reference re-initialises EVERY Minkowski layer of the model here (:178-185)
            if isinstance(m, ME.MinkowskiConvolution):
                ME.kaiming_normal_(m.kernel, mode="fan_out", nonlinearity="relu")
            if isinstance(m, ME.MinkowskiBatchNorm):
                nn.init.constant_(m.bn.weight, 1)
                nn.init.constant_(m.bn.bias, 0)
        ME.fuse_activations(self)  # BatchNorm -> ELU pairs of the neck run as one fused pass

    def backbone_forward(self, inputs):
        """model_vdetr.py:248-280: voxelise at ``voxel_size``, MinkResNet, top-down FPN, out block -> per scene
        (xyz [n,3] = out.C[:,1:] * voxel_size, features [n,C] = out.F).  The reference's default (no colour) path reads an
        undefined name (:259); the evident intent — xyz as the input features — is what runs here."""
        clouds = inputs["point_clouds"]
        if self.use_color:
            data = [(p[:, :3] / self.voxel_size, p[:, :] if self.xyz_color else p[:, 3:]) for p in clouds]
        else:
            data = [(p[:, :3] / self.voxel_size, p[:, :3]) for p in clouds]
        coordinates, features = ME.batch_sparse_collate(data)
        # the coordinate manager (sites per stride, kernel maps, compacted row lists) depends on the point COORDINATES only:
        # like the FPS indices it may be built ahead of time (``prepare_geometry``) and handed in as inputs["geometry"]
        geometry = inputs.get("geometry")
        if geometry is None:
            x = ME.SparseTensor(features.contiguous(), coordinates=coordinates)
        else:
            if features.is_cuda:
                geometry.use_on()  # built on another stream (loader thread / side stream): this stream reads it from here on
            x = ME.SparseTensor(features[geometry.unique_index].contiguous(), coordinate_manager=geometry)
        stages = self.pre_encoder(x)
        x = stages[-1]
        out = None
        for i in range(len(stages) - 1, self.layer_idx - 1, -1):
            if self.use_fpn:
                if i < len(stages) - 1:
                    x = getattr(self, f"up_block_{i + 1}")(x)
                    x = stages[i] + x
            else:
                x = stages[i]
            if i == self.layer_idx:
                out = getattr(self, f"out_block_{i}")(x)
        return [(c.to(out.F.dtype) * self.voxel_size, f) for c, f in out.decomposed()]

    @torch.no_grad()
    def prepare_geometry(self, inputs):
        """Everything of the backbone that depends on the point coordinates only — voxel sites of every tensor stride, the
        kernel maps of every layer shape and their compacted row lists (sparse_ops.ConvPlan) — as a coordinate manager to
        pass as ``inputs["geometry"]``.  A training loop can run this for the NEXT scene (data loader / side stream) while
        the current one trains; the maps themselves are filled in by one geometry-only pass over the layers.  The consumer
        must first wait for the producing stream (event / wait_stream); the manager's memory is then tied to the consuming
        stream by ``backbone_forward`` itself (``CoordinateManager.use_on``), so it may be dropped right after the step."""
        clouds = inputs["point_clouds"]
        coordinates, _ = ME.batch_sparse_collate([(p[:, :3] / self.voxel_size, p[:, :0]) for p in clouds])
        cm = ME.CoordinateManager(clouds[0].device)
        cm.insert_points(coordinates)
        with ME.geometry_only():  # the layers only register their sites / maps / pair lists: no feature arithmetic
            self.backbone_forward(dict(inputs, geometry=cm))
        return cm.finalize()

    def _scenes(self, inputs):
        return self.backbone_forward(inputs) if self.sparse_backbone else self.pre_encoder(inputs)

    def _anchor_sizes(self, ref):
        """per-class anchor sizes on ref's device (model_vdetr.py:348-352), cached so the forward has no H2D copy"""
        key = (ref.device, ref.dtype, self.hard_anchor)
        cache = self.__dict__.setdefault("_anchor_cache", {})
        if key not in cache:
            sizes = self.dataset_config.mean_size_arr_hard_anchor if self.hard_anchor else self.dataset_config.mean_size_arr
            cache[key] = torch.as_tensor(sizes, dtype=ref.dtype, device=ref.device)
        return cache[key]

    def get_query_embeddings(self, encoder_xyz, enc_features, point_cloud_dims):
        if self.querypos_mlp:
            return encoder_xyz, encoder_xyz, None
        pos_embed = self.pos_embedding(encoder_xyz, input_range=point_cloud_dims)
        return encoder_xyz, self.query_projection(pos_embed).permute(2, 0, 1), None

    @torch.no_grad()
    def flat_param_groups(self):
        """Adjacency wishes for dist.FlatParams (see TransformerDecoder.flat_param_groups)."""
        return self.decoder.flat_param_groups() if hasattr(self.decoder, "flat_param_groups") else []

    def sample_indices(self, inputs):
        """FPS indices [B, npoint] (int32) of a batch whose scenes have equal voxel counts.  The sampling only
        depends on the voxel COORDINATES, so a training loop can run it ahead of time — e.g. for the next batch on a
        side stream while the current batch is in the decoder — and pass the result as ``inputs["fps_inds"]``."""
        scenes = self._scenes(inputs)
        return pointnet2_utils.furthest_point_sample_varlen([s[0].contiguous() for s in scenes], self.npoint)

    def run_encoder(self, inputs):
        """Backbone output -> FPS to ``npoint`` tokens per scene (model_vdetr.py:279-326).  A caller that already holds
        the sampled tokens (a training loop that runs the variable-size backbone eagerly and the fixed-size rest as a
        captured hipGraph: bench.BackboneTrainer) hands them in as inputs["enc_xyz"] [B,m,3], ["enc_features"] [m,B,C],
        ["enc_inds"] [B,m]."""
        if "enc_features" in inputs:
            return inputs["enc_xyz"], inputs["enc_features"], inputs.get("enc_inds")
        scenes = self._scenes(inputs)
        if not self.random_fps and len(scenes) <= 32:
            # point-major tables as the backbone hands them over, scenes of any sizes: ONE FPS launch (a workgroup per
            # scene) and two row gathers for the batch; no (B,C,n) transposed copy of the features in either direction
            xyzs = [s[0].contiguous() for s in scenes]
            enc_inds = inputs.get("fps_inds")
            if enc_inds is None:
                enc_inds = pointnet2_utils.furthest_point_sample_varlen(xyzs, self.npoint)
            enc_xyz = pointnet2_utils.gather_rows(xyzs, enc_inds)                                  # B,m,3
            enc_rows = pointnet2_utils.gather_rows([s[1].contiguous() for s in scenes], enc_inds)  # B,m,C
            return enc_xyz, enc_rows.permute(1, 0, 2), enc_inds  # features: npoints x batch x channel
        else:
            out = []
            for xyz_i, feats_i in scenes:
                if self.random_fps:
                    perm = torch.randperm(xyz_i.shape[0], device=xyz_i.device)
                    xyz_i, feats_i = xyz_i[perm], feats_i[perm]
                out.append(self.fps_module(xyz_i.unsqueeze(0).contiguous(),
                                           feats_i.transpose(0, 1).unsqueeze(0).contiguous(), self.npoint))
            enc_xyz = torch.cat([o[0] for o in out])
            enc_features = torch.cat([o[1] for o in out])
            enc_inds = torch.cat([o[2] for o in out])
        return enc_xyz, enc_features.permute(2, 0, 1), enc_inds  # features: npoints x batch x channel

    def forward(self, inputs, encoder_only=False):
        point_cloud_dims = [inputs["point_cloud_dims_min"], inputs["point_cloud_dims_max"]]
        enc_xyz, enc_features, enc_inds = self.run_encoder(inputs)
        bs, npoints, _ = enc_xyz.shape
        enc_features = self.encoder_to_decoder_projection(enc_features.permute(1, 2, 0)).permute(2, 0, 1)

        point_cls_logits = self.decoder.pointcls_heads(enc_features.permute(1, 2, 0)) \
            .transpose(1, 2).reshape((bs, npoints, -1)).contiguous()
        query_xyz, query_embed, _ = self.get_query_embeddings(enc_xyz, enc_features, point_cloud_dims)
        if point_cls_logits.is_cuda and point_cls_logits.dtype == torch.float32 and query_xyz.dtype == torch.float32 \
                and type(self.dataset_config).box_parametrization_to_corners is ScannetDatasetConfig.box_parametrization_to_corners:
            # the ~10 launches below as one (box_decode.anchor_boxes: csrc/box_decode.hip); nothing of it is differentiated
            size_unnormalized, center_normalized, size_normalized, corners = box_decode.anchor_boxes(
                point_cls_logits, query_xyz, point_cloud_dims, self._anchor_sizes(enc_features))
            enc_box_predictions = {"point_cls_logits": point_cls_logits, "center_unnormalized": query_xyz,
                                   "center_normalized": center_normalized, "size_unnormalized": size_unnormalized,
                                   "size_normalized": size_normalized, "box_corners": corners}
        else:
            class_idx = point_cls_logits.sigmoid().max(dim=-1)[1]
            size_unnormalized = self._anchor_sizes(enc_features)[class_idx]
            enc_box_predictions = {
                "point_cls_logits": point_cls_logits,
                "center_unnormalized": query_xyz,
                "center_normalized": convert_unnorm2norm(query_xyz, point_cloud_dims),
                "size_unnormalized": size_unnormalized,
                "size_normalized": convert_unnorm2norm(size_unnormalized, point_cloud_dims, with_offset=False),
            }
            enc_box_predictions["box_corners"] = self.decoder.box_processor.box_parametrization_to_corners(
                query_xyz, size_unnormalized, None)  # (yaw 0, as the reference's zero tensor: pc_util.get_3d_box_batch_tensor)
        tgt = None if self.querypos_mlp else torch.zeros_like(query_embed)
        box_predictions = self.decoder(tgt, enc_features, query_xyz, enc_xyz, point_cloud_dims, query_pos=query_embed,
                                       enc_box_predictions=enc_box_predictions, enc_box_features=enc_features)[0]
        box_predictions["seed_inds"] = enc_inds
        box_predictions["seed_xyz"] = enc_xyz
        box_predictions["enc_outputs"] = enc_box_predictions
        return box_predictions


def default_args(**overrides):
    """The hot-path flags of main.py:30-216 with their defaults."""
    a = dict(enc_dim=256, dec_nlayers=9, dec_dim=256, dec_ffn_dim=256, dec_dropout=0.1, dec_nhead=4, rpe_dim=128,
             rpe_quant="bilinear_4_10", log_scale=512.0, pos_for_key=False, querypos_mlp=True, q_content="random",
             proj_nohid=True, share_selfattn=False, mlp_dropout=0.3, mlp_norm="bn1d", mlp_act="relu", mlp_sep=True,
             preenc_npoints=4096, nqueries=1024, is_bilable=True, angle_type="", hard_anchor=False,
             cls_loss="focalloss_0.25", voxel_size=0.01, random_fps=False,
             # backbone flags (main.py:55-63,88,112,166-167)
             depth=34, inplanes=64, num_stages=4, stem_bn=True, use_fpn=True, layer_idx=0, woexpand_conv=True,
             use_color=False, xyz_color=False, use_normals=False)
    a.update(overrides)
    return SimpleNamespace(**a)


def build_decoder(args, dataset_config):
    """model_vdetr.py:413-447"""
    first_layer = FFNLayer(d_model=args.dec_dim, dim_feedforward=args.dec_ffn_dim, dropout=args.dec_dropout)
    decoder_layer = GlobalDecoderLayer(d_model=args.dec_dim, nhead=args.dec_nhead,
                                       dim_feedforward=args.dec_ffn_dim, dropout=args.dec_dropout,
                                       pos_for_key=args.pos_for_key, args=args)
    return TransformerDecoder(first_layer, decoder_layer, dataset_config, num_layers=args.dec_nlayers - 1,
                              decoder_dim=args.dec_dim, mlp_dropout=args.mlp_dropout, mlp_norm=args.mlp_norm,
                              mlp_act=args.mlp_act, mlp_sep=args.mlp_sep, pos_for_key=args.pos_for_key,
                              num_queries=args.nqueries, cls_loss=args.cls_loss, is_bilable=args.is_bilable,
                              q_content=args.q_content, return_intermediate=True, args=args)


def build_backbone(args):
    """model_vdetr.py:392-410: MinkResNet on xyz (3), xyz + colour (6) [+ normals (+3)] input features."""
    if args.use_color and args.xyz_color:
        point_dim = 9 if args.use_normals else 6
    else:
        point_dim = 6 if args.use_normals else 3
    return MinkResNet(depth=args.depth, in_channels=point_dim, inplanes=args.inplanes, num_stages=args.num_stages,
                      stem_bn=args.stem_bn)


def build_vdetr(args, dataset_config, pre_encoder=None):
    """model_vdetr.py:450-474.  ``pre_encoder``: None = TensorBackbone (the post-backbone hot path on handed-in voxel
    tables), "minkowski" = the reference's sparse ResNet34 + FPN backbone (build_backbone), or any provider."""
    if isinstance(pre_encoder, str) and pre_encoder == "minkowski":
        pre_encoder = build_backbone(args)
    return ModelVDETR(pre_encoder if pre_encoder is not None else TensorBackbone(), None,
                      build_decoder(args, dataset_config), dataset_config, encoder_dim=args.enc_dim,
                      decoder_dim=args.dec_dim, num_queries=args.nqueries, querypos_mlp=args.querypos_mlp,
                      minkowski=True, inplane=getattr(args, "inplanes", 64), num_stages=getattr(args, "num_stages", 4),
                      voxel_size=args.voxel_size, npoint=args.preenc_npoints, use_fpn=getattr(args, "use_fpn", True),
                      layer_idx=getattr(args, "layer_idx", 0), proj_nohid=args.proj_nohid,
                      woexpand_conv=getattr(args, "woexpand_conv", True), args=args)
