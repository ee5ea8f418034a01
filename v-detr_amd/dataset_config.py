"""Dataset constants the decoder reads (datasets/scannet.py:38-171): class count, angle bins, per-class mean
box sizes and the box -> 8 corner parametrisation.  The data loader itself is out of scope."""
import numpy as np

from .pc_util import flip_axis_to_camera_tensor, get_3d_box_batch_tensor


class ScannetDatasetConfig:
    """num_semcls=18, num_angle_bin=1 (axis-aligned boxes), max_num_obj=64 (scannet.py:40-43)."""

    def __init__(self):
        self.num_semcls = 18
        self.num_angle_bin = 1
        self.max_num_obj = 64
        # scannet.py:72-91
        self.mean_size_arr = np.array([
            [0.76966726, 0.81160211, 0.92573741], [1.876858, 1.84255952, 1.19315654],
            [0.61327999, 0.61486087, 0.71827014], [1.39550063, 1.51215451, 0.83443565],
            [0.97949596, 1.06751485, 0.63296875], [0.53166301, 0.59555772, 1.75001483],
            [0.96247056, 0.72462326, 1.14818682], [0.83221924, 1.04909355, 1.68756634],
            [0.21132214, 0.4206159, 0.53728459], [1.44400728, 1.89708334, 0.26985747],
            [1.02942616, 1.40407966, 0.87554322], [1.37664116, 0.65521793, 1.68131292],
            [0.66508189, 0.71111926, 1.29885307], [0.41999174, 0.37906947, 1.75139715],
            [0.59359559, 0.59124924, 0.73919014], [0.50867595, 0.50656087, 0.30136236],
            [1.15115265, 1.0546296, 0.49706794], [0.47535286, 0.49249493, 0.58021168]])
        self.mean_size_arr_hard_anchor = np.ones((18, 3))  # scannet.py:93-95

    def box_parametrization_to_corners(self, box_center_unnorm, box_size, box_angle):
        """scannet.py:168-171"""
        return get_3d_box_batch_tensor(box_size, box_angle, flip_axis_to_camera_tensor(box_center_unnorm))


class RotatedBoxDatasetConfig(ScannetDatasetConfig):
    """Rotated-box dataset stand-in for BASELINE config 5 (SUN RGB-D values of 3DETR: 10 classes, 12 angle bins;
    the reference's own SUN RGB-D loader is unreleased, datasets/__init__.py:2)."""

    def __init__(self, num_semcls=10, num_angle_bin=12):
        super().__init__()
        self.num_semcls = num_semcls
        self.num_angle_bin = num_angle_bin
        rng = np.random.default_rng(0)
        self.mean_size_arr = 0.4 + rng.random((num_semcls, 3)) * 1.6
        self.mean_size_arr_hard_anchor = np.ones((num_semcls, 3))
