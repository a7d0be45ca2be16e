"""KITTI raw front end (dataloaders/kitti.py)."""
import numpy as np

from .generic import DataLoaderGeneric


class DataLoaderKittiRaw(DataLoaderGeneric):
    """Dataloader for the raw Kitti dataset: intrinsics come from the csv (normalised by the image
    size), ground truth = sparse velodyne projections stored as uint16 PNG, depth = value / 256,
    resized with nearest neighbour; evaluation uses the Garg/Eigen crop (kitti.py:14-20,43-50)."""
    depth_kind = 1

    def __init__(self):
        super(DataLoaderKittiRaw, self).__init__('kitti-raw')
        self.in_size = [370, 1220]
        self.depth_type = "velodyne"

    def _set_output_size(self, out_size=[256, 768]):
        self.out_size = list(out_size)
        crop = np.array([0.40810811 * out_size[0], 0.99189189 * out_size[0],
                         0.03594771 * out_size[1], 0.96405229 * out_size[1]]).astype(np.int32)
        self.eval_crop = tuple(int(v) for v in crop)

    def _camera(self, row):
        f32 = np.float32
        return ((float(f32(row['fx'] * self.out_size[1])), float(f32(row['fy'] * self.out_size[0]))),
                (float(f32(row['cx'] * self.out_size[1])), float(f32(row['cy'] * self.out_size[0]))))

    def _depth_crop(self):
        return self.eval_crop if self.usecase == "eval" else None          # kitti.py:48-50

    def _perform_augmentation(self):
        self._augmentation_step_color(invert_color=False)
