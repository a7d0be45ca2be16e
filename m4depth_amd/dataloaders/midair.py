"""Mid-Air front end (dataloaders/midair.py)."""
import numpy as np
import torch

from .generic import DataLoaderGeneric


class DataLoaderMidAir(DataLoaderGeneric):
    """Dataloader for the Mid-Air dataset: 1024x1024 JPEG frames, ground truth = 16-bit PNG holding
    float16 stereo disparity, depth = 512 / disparity (midair.py:49-55)."""
    depth_kind = 0

    def __init__(self, out_size=[384, 384], crop=False):
        super(DataLoaderMidAir, self).__init__('midair')
        self.in_size = [1024, 1024]
        self.depth_type = "map"
        self.crop = crop

    def _set_output_size(self, out_size=[384, 384]):
        self.out_size = list(out_size)
        self.long_edge = 0 if out_size[0] >= out_size[1] else 1
        if self.crop:
            self.intermediate_size = [out_size[self.long_edge], out_size[self.long_edge]]
        else:
            self.intermediate_size = list(out_size)
        self.fx = 0.5 * self.intermediate_size[1]                    # midair.py:20-23
        self.fy = 0.5 * self.intermediate_size[0]
        self.cx = 0.5 * self.intermediate_size[1]
        self.cy = 0.5 * self.intermediate_size[0]

    def get_dataset(self, usecase, settings, batch_size=3, out_size=[384, 384], crop=False, **kw):
        self.crop = crop
        if (usecase == "eval" or usecase == "predict") and self.crop:
            raise AttributeError("Crop option should be disabled when evaluating or predicting samples")
        return super(DataLoaderMidAir, self).get_dataset(usecase, settings, batch_size=batch_size, out_size=out_size, **kw)

    def _decode_size(self):
        return self.intermediate_size

    def _depth_column(self):
        return "disp"

    def _camera(self, row):
        return (self.fx, self.fy), (self.cx, self.cy)

    def _perform_augmentation(self):
        """midair.py:59-108: flips, transposition of square frames, crop, colour."""
        if not self.usecase == "finetune":
            self._augmentation_step_flip()
            if self.intermediate_size[0] == self.intermediate_size[1] and self._uniform(0., 1.) < 0.5:
                rot, trans = self.out_data["rot"], self.out_data["trans"]
                self.out_data["RGB_im"] = self.out_data["RGB_im"].permute(0, 2, 1, 3).contiguous()
                self.out_data["depth"] = self.out_data["depth"].permute(0, 2, 1, 3).contiguous()
                self.out_data["rot"] = torch.stack([rot[:, 0], -rot[:, 2], -rot[:, 1], -rot[:, 3]], dim=1)
                self.out_data["trans"] = torch.stack([trans[:, 1], trans[:, 0], trans[:, 2]], dim=1)
        if self.crop:
            c = self.out_data['camera']['c']
            if self.long_edge == 0:
                diff = self.intermediate_size[1] - self.out_size[1]
                offset = int(self.rng.integers(0, max(diff, 1)))
                sl = (slice(None), slice(0, self.out_size[0]), slice(offset, offset + self.out_size[1]))
                self.out_data['camera']['c'] = torch.stack([c[0] - float(offset), c[1]])
            else:
                diff = self.intermediate_size[0] - self.out_size[0]
                offset = int(self.rng.integers(0, max(diff, 1)))
                sl = (slice(None), slice(offset, offset + self.out_size[0]), slice(0, self.out_size[1]))
                self.out_data['camera']['c'] = torch.stack([c[0], c[1] - float(offset)])
            self.out_data['RGB_im'] = self.out_data['RGB_im'][sl].contiguous()
            self.out_data['depth'] = self.out_data['depth'][sl].contiguous()
        self._augmentation_step_color()
