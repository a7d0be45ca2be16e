"""TartanAir front end (dataloaders/tartanair.py)."""
import numpy as np

from .generic import DataLoaderGeneric


class DataLoaderTartanAir(DataLoaderGeneric):
    """Dataloader for the TartanAir dataset: 480x640 frames, ground truth = raw float32 maps (the last
    480*640 floats of the .npy file), nearest resize, zeroed where the colour image is black
    (tartanair.py:37-45)."""
    depth_kind = 2

    def __init__(self, out_size=[384, 512]):
        super(DataLoaderTartanAir, self).__init__('tartanair')
        self.in_size = [480, 640]
        self.depth_type = "map"

    def _set_output_size(self, out_size=[384, 512]):
        self.out_size = list(out_size)
        self.fx = 0.5 * self.out_size[1]                             # tartanair.py:16-19
        self.fy = 2. / 3. * self.out_size[0]
        self.cx = 0.5 * self.out_size[1]
        self.cy = 0.5 * self.out_size[0]

    def _camera(self, row):
        return (self.fx, self.fy), (self.cx, self.cy)

    def _read_depth(self, path):
        raw = np.fromfile(path, dtype=np.float32)                    # tf.io.decode_raw(file, tf.float32)
        n = self.in_size[0] * self.in_size[1]
        return raw[-n:].reshape(self.in_size)

    def _perform_augmentation(self):
        self._augmentation_step_flip()
        self._augmentation_step_color()
