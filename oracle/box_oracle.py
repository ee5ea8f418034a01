"""TEST INFRASTRUCTURE — CPU/torch restatement of the box decode of one decoder stage.

Follows the reference op by op (models/vdetr_transformer.py:286-333, BoxProcessor :20-90, datasets/scannet.py:168-171,
utils/box_util.py:294-352) with plain ATen calls, so autograd provides the reference gradients.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline may import this module; the product path is
v-detr_amd/box_decode.py -> vdetr_box_decode_{fwd,bwd}_f32.  Pinned through tests/golden/decoder_*.npz (outputs of the
reference's TransformerDecoder, which runs this code for every stage).
"""
import numpy as np
import torch
import torch.nn.functional as F


def _flip_axis_to_camera(pc):  # box_util.py:294-301
    return torch.stack((pc[..., 0], -pc[..., 2], pc[..., 1]), dim=-1)


def _roty(t):  # box_util.py:304-316
    c, s = torch.cos(t), torch.sin(t)
    zero, one = torch.zeros_like(c), torch.ones_like(c)
    return torch.stack((torch.stack((c, zero, s), -1), torch.stack((zero, one, zero), -1),
                        torch.stack((-s, zero, c), -1)), -2)


_SX = (1, 1, -1, -1, 1, 1, -1, -1)
_SY = (1, 1, 1, 1, -1, -1, -1, -1)
_SZ = (1, -1, -1, 1, 1, -1, -1, 1)


def _corners(size, angle, center_cam):  # box_util.py:319-352
    sx, sy, sz = (torch.tensor(v, dtype=size.dtype, device=size.device) * 0.5 for v in (_SX, _SY, _SZ))
    l, w, h = size[..., 0:1], size[..., 1:2], size[..., 2:3]
    local = torch.stack((l * sx, h * sy, w * sz), dim=-1)
    return torch.matmul(local, _roty(angle).transpose(-1, -2)) + center_cam.unsqueeze(-2)


def _angle(angle_logits, angle_residual, num_angle_bin, zero_angle=False):  # vdetr_transformer.py:48-71
    nbin = angle_logits.shape[-1]
    if nbin == 1 or zero_angle:
        if nbin == 1:
            angle = (angle_logits * 0 + angle_residual * 0).squeeze(-1).clamp(min=0)
        else:
            angle = (angle_logits.sum(-1) * 0 + angle_residual.sum(-1) * 0).squeeze(-1).clamp(min=0)
        return angle, angle
    angle_per_cls = 2 * np.pi / num_angle_bin
    angle_prob, pred = F.softmax(angle_logits, dim=-1).max(dim=-1)
    pred = pred.detach()
    angle = angle_per_cls * pred + angle_residual.gather(2, pred.unsqueeze(-1)).squeeze(-1)
    angle = torch.where(angle > np.pi, angle - 2 * np.pi, angle)
    return angle, angle_prob


def decode_boxes_reference(raw, pre_center_normalized, pre_size_normalized, point_cloud_dims, num_angle_bin,
                           cls_loss="celoss"):
    """Same signature and result dictionary as vdetr_amd.box_decode.decode_boxes."""
    batch, _, nq = raw["center_head"].shape
    dmin = point_cloud_dims[0].unsqueeze(1)
    scene_size = (point_cloud_dims[1] - point_cloud_dims[0]).unsqueeze(1)
    pre_center_unnormalized = pre_center_normalized * scene_size + dmin
    pre_size_unnormalized = pre_size_normalized * scene_size
    cls_logits = raw["sem_cls_head"].transpose(1, 2)
    center_reg = raw["center_head"].transpose(1, 2).contiguous().view(batch, nq, 3)
    center_unnormalized = center_reg * pre_size_unnormalized + pre_center_unnormalized
    center_normalized = (center_unnormalized - dmin) / scene_size
    size_reg = raw["size_head"].transpose(1, 2).contiguous().view(batch, nq, 3)
    size_unnormalized = torch.exp(size_reg) * pre_size_unnormalized
    size_normalized = size_unnormalized / scene_size
    angle_logits = raw["angle_cls_head"].transpose(1, 2)
    angle_residual_normalized = raw["angle_residual_head"].transpose(1, 2)
    angle_residual = angle_residual_normalized * (np.pi / angle_residual_normalized.shape[-1])
    angle_continuous, angle_prob = _angle(angle_logits, angle_residual, num_angle_bin)
    cam = _flip_axis_to_camera(center_unnormalized)
    box_corners = _corners(size_unnormalized, angle_continuous, cam)
    angle_zero, _ = _angle(angle_logits, angle_residual, num_angle_bin, zero_angle=True)
    box_corners_axis_align = _corners(size_unnormalized, angle_zero, cam)
    with torch.no_grad():
        if cls_loss.split("_")[0] == "focalloss":
            semcls_prob, objectness_prob = cls_logits, cls_logits.sigmoid().max(dim=-1)[0]
        else:
            cls_prob = F.softmax(cls_logits, dim=-1)
            semcls_prob, objectness_prob = cls_prob[..., :-1], 1 - cls_prob[..., -1]
    return {
        "sem_cls_logits": cls_logits,
        "center_normalized": center_normalized.contiguous(),
        "center_unnormalized": center_unnormalized,
        "size_normalized": size_normalized,
        "size_unnormalized": size_unnormalized,
        "angle_logits": angle_logits,
        "angle_prob": angle_prob,
        "angle_residual": angle_residual,
        "angle_residual_normalized": angle_residual_normalized,
        "angle_continuous": angle_continuous,
        "objectness_prob": objectness_prob,
        "sem_cls_prob": semcls_prob,
        "box_corners": box_corners,
        "box_corners_axis_align": box_corners_axis_align,
        "pre_box_center_unnormalized": pre_center_unnormalized,
        "center_reg": center_reg,
        "pre_box_size_unnormalized": pre_size_unnormalized,
        "size_reg": size_reg,
    }
