"""Superclass of the dataset front ends -- the counterpart of dataloaders/generic.py.

What the reference builds out of tf.data (csv -> from_tensor_slices -> batch(db_seq_len) ->
_cut_sequence -> shuffle -> map(_decode_samples) -> batch(seq_len) -> _build_sequence_samples ->
batch(batch_size) -> prefetch, generic.py:84-146) is here a plain Python pipeline split where the
hardware suggests:

  host   : csv records, file reads and the JPEG / PNG decompression (PIL) -- in a background
           thread, ``prefetch`` batches ahead of the consumer (tf.data's prefetch(AUTOTUNE));
  device : everything after decompression.  The decoder's raw output is uploaded as it is
           (uint8 RGB, uint16 / float32 depth: 3-4x fewer PCIe bytes than float32 frames) and ONE
           HIP kernel per modality casts, scales and resizes straight into the NHWC float32 network
           input (m4d_decode_rgb8_resize / m4d_decode_depth_resize); augmentations are elementwise
           torch ops on the device tensors.

Batches come out as the dict ``M4Depth.train_step / test_step`` take: RGB_im [b,T,H,W,3],
depth [b,T,H,W,1], rot [b,T,4], trans [b,T,3] on the device, new_traj [b,T] (host bool tensor: it steers
control flow), camera {f [b,2], c [b,2]}.  Streaming evaluation (db_seq_len None) yields single
frames without the T axis, batch 1, like the reference.
"""
from __future__ import annotations

import csv
import ctypes
import glob
import os
import queue
import threading
from collections import namedtuple

import numpy as np
import torch

from .._lib import lib, dptr, stream_ptr, check

DataloaderParameters = namedtuple('DataloaderParameters', ('db_path_config', 'records_path', 'db_seq_len', 'seq_len', 'augment'))

_NUMERIC = ("id", "qw", "qx", "qy", "qz", "tx", "ty", "tz", "fx", "fy", "cx", "cy")


def read_trajectory_csv(path):
    """One trajectory file (tab separated, header row; scripts/midair-split-generator.py:55):
    returns a list of row dicts; the pose / intrinsics columns are floats, ``id`` an int."""
    rows = []
    with open(path, newline="") as fh:
        for rec in csv.DictReader(fh, delimiter="\t"):
            row = {}
            for k, v in rec.items():
                if k == "id":
                    row[k] = int(float(v))
                elif k in _NUMERIC:
                    row[k] = float(v)
                else:
                    row[k] = v
            rows.append(row)
    return rows


class _Prefetcher:
    """Runs ``producer()`` (a generator of host-side work items) in a thread, ``depth`` items ahead."""

    def __init__(self, producer, depth):
        self.q = queue.Queue(maxsize=max(1, depth))
        self.t = threading.Thread(target=self._run, args=(producer,), daemon=True)
        self.t.start()

    def _run(self, producer):
        try:
            for item in producer():
                self.q.put(("item", item))
            self.q.put(("end", None))
        except BaseException as e:            # surfaced in the consumer, not swallowed
            self.q.put(("error", e))

    def __iter__(self):
        while True:
            kind, item = self.q.get()
            if kind == "end":
                return
            if kind == "error":
                raise item
            yield item


class SequenceDataset:
    """Iterable over batches; ``cardinality()`` = number of batches per epoch."""

    def __init__(self, loader, plan_fn, n_batches, prefetch=2):
        self.loader, self.plan_fn, self.n, self.prefetch = loader, plan_fn, n_batches, prefetch

    def cardinality(self):
        return self.n

    def __len__(self):
        return self.n

    def __iter__(self):
        loader = self.loader

        def producer():
            for batch_rows in self.plan_fn():
                yield [[loader._load_raw(r) for r in seq] for seq in batch_rows], batch_rows

        for raws, rows in _Prefetcher(producer, self.prefetch):
            yield loader._assemble(raws, rows)


class DataLoaderGeneric():
    """Superclass for other dataset dataloaders (dataloaders/generic.py:10-259)."""

    depth_kind = None           # 0 Mid-Air float16 disparity, 1 KITTI uint16/256, 2 TartanAir float32

    def __init__(self, dataset_name):
        self.build_functions = {"train": self._build_train_dataset,
                                "finetune": self._build_train_dataset,
                                "eval": self._build_eval_dataset,
                                "predict": self._build_eval_dataset}
        self.augment = None
        self.settings = None
        self.db_name = dataset_name
        self.device = None
        self.rng = np.random.default_rng(42)

    # -- to be provided by the dataset classes ---------------------------------------------
    def _set_output_size(self, out_size=None):
        raise NotImplementedError

    def _camera(self, row):
        """(fx, fy), (cx, cy) of one csv row at the decode size."""
        raise NotImplementedError

    def _perform_augmentation(self):
        raise NotImplementedError

    def _decode_size(self):
        """[h, w] the frames are resized to by _decode_samples."""
        return self.out_size

    def _depth_column(self):
        return "depth"

    def _read_depth(self, path):
        """Host part of the ground-truth decode: the file's pixels, untouched."""
        from PIL import Image
        with Image.open(path) as im:
            return np.asarray(im, dtype=np.uint16)

    # -- host side -----------------------------------------------------------------------------
    def _load_raw(self, row):
        """Decompress one sample's files: {'rgb': uint8 [h,w,3], 'depth': raw map or None}."""
        from PIL import Image
        with Image.open(os.path.join(self.db_path, row["camera_l"])) as im:
            rgb = np.asarray(im.convert("RGB"), dtype=np.uint8)
        depth = None
        col = self._depth_column()
        if col in row and row[col]:
            depth = self._read_depth(os.path.join(self.db_path, row[col]))
        return {"rgb": rgb, "depth": depth}

    # -- device side ---------------------------------------------------------------------------
    def _to_device(self, arr):
        t = torch.from_numpy(np.ascontiguousarray(arr))
        return t.pin_memory().to(self.device, non_blocking=True) if self.device.type == "cuda" else t

    def _decode_device(self, raws):
        """Raw frames of one sequence -> (RGB_im [T,h,w,3], depth [T,h,w,1] or None) on the device."""
        oh, ow = self._decode_size()
        groups = {}
        for i, r in enumerate(raws):
            groups.setdefault((r["rgb"].shape, None if r["depth"] is None else r["depth"].shape), []).append(i)
        T = len(raws)
        rgb_out = torch.empty((T, oh, ow, 3), dtype=torch.float32, device=self.device)
        has_depth = all(r["depth"] is not None for r in raws)
        dep_out = torch.empty((T, oh, ow, 1), dtype=torch.float32, device=self.device) if has_depth else None
        crop = self._depth_crop()
        crop_arg = (ctypes.c_int * 4)(*crop) if crop is not None else None
        for (rgb_shape, dep_shape), idx in groups.items():            # frames of one size go through one launch
            ih, iw = rgb_shape[:2]
            stack = self._to_device(np.stack([raws[i]["rgb"] for i in idx]))
            dst = rgb_out if len(idx) == T else torch.empty((len(idx), oh, ow, 3), dtype=torch.float32, device=self.device)
            check(lib.m4d_decode_rgb8_resize(dptr(stack, "rgb", torch.uint8), len(idx), ih, iw, oh, ow, dptr(dst),
                                             stream_ptr()), "m4d_decode_rgb8_resize")
            if dst is not rgb_out:
                rgb_out[idx] = dst
            if has_depth:
                dh, dw = dep_shape[:2]
                raw = np.stack([raws[i]["depth"] for i in idx])
                if self.depth_kind == 2:
                    dstack = self._to_device(raw.astype(np.float32, copy=False))
                    dt = torch.float32
                else:
                    dstack = self._to_device(raw.view(np.int16))       # torch has no uint16 transfers: same bits
                    dt = torch.int16
                ddst = dep_out if len(idx) == T else torch.empty((len(idx), oh, ow, 1), dtype=torch.float32, device=self.device)
                check(lib.m4d_decode_depth_resize(dptr(dstack, "depth", dt), self.depth_kind, len(idx), dh, dw, oh, ow,
                                                  dptr(dst) if self.depth_kind == 2 else None, crop_arg, dptr(ddst),
                                                  stream_ptr()), "m4d_decode_depth_resize")
                if ddst is not dep_out:
                    dep_out[idx] = ddst
        return rgb_out, dep_out

    def _depth_crop(self):
        return None

    def _decode_samples(self, data_sample):
        ''' Creates a sample to be fed to the network from a line of the dataset csv files
            (dataloaders/generic.py:23-34): camera, depth (optional), RGB_im, rot, trans, new_traj. '''
        rgb, depth = self._decode_device([self._load_raw(data_sample)])
        out = self._pose(data_sample)
        out["RGB_im"] = rgb[0]
        if depth is not None:
            out["depth"] = depth[0]
        return out

    def _pose(self, row):
        f, c = self._camera(row)
        return {"camera": {"f": torch.tensor(f, dtype=torch.float32, device=self.device),
                           "c": torch.tensor(c, dtype=torch.float32, device=self.device)},
                "rot": torch.tensor([row['qw'], row['qx'], row['qy'], row['qz']], dtype=torch.float32, device=self.device),
                "trans": torch.tensor([row['tx'], row['ty'], row['tz']], dtype=torch.float32, device=self.device),
                "new_traj": row['id'] == 0}

    def _assemble(self, raws, rows):
        """Batch of sequences: decode on the device, augment per sequence, stack."""
        seqs = []
        for seq_raw, seq_rows in zip(raws, rows):
            rgb, depth = self._decode_device(seq_raw)
            if self.streaming:
                sample = self._pose(seq_rows[0])
                sample["RGB_im"] = rgb[0]
                if depth is not None:
                    sample["depth"] = depth[0]
                seqs.append(sample)
                continue
            poses = np.array([[r['qw'], r['qx'], r['qy'], r['qz'], r['tx'], r['ty'], r['tz']] for r in seq_rows], np.float32)
            f, c = self._camera(seq_rows[0])                                  # generic.py:163-166: the first frame's
            self.out_data = {"camera": {"f": torch.tensor(f, dtype=torch.float32, device=self.device),
                                        "c": torch.tensor(c, dtype=torch.float32, device=self.device)},
                             "RGB_im": rgb, "rot": torch.from_numpy(poses[:, :4]).to(self.device),
                             "trans": torch.from_numpy(poses[:, 4:]).to(self.device),
                             "new_traj": np.array([i == 0 for i in range(len(seq_rows))])}
            if depth is not None:
                self.out_data["depth"] = depth
            if self.augment:
                self._perform_augmentation()
            seqs.append(dict(self.out_data))
        batch = {}
        for key in seqs[0]:
            if key == "camera":
                batch[key] = {k: torch.stack([s["camera"][k] for s in seqs]) for k in ("f", "c")}
            elif key == "new_traj":
                batch[key] = torch.from_numpy(np.stack([np.asarray(s[key]) for s in seqs]))     # stays on the host
            else:
                batch[key] = torch.stack([s[key] for s in seqs])
        return batch

    # -- dataset construction ---------------------------------------------------------------
    def get_dataset(self, usecase, settings, batch_size=3, out_size=None, device=None, seed=42, prefetch=2,
                    shard=(0, 1)):
        ''' Builds the dataset (dataloaders/generic.py:50-82).
            * usecase : train, finetune, eval or predict
            * settings: DataloaderParameters (db_path_config, records_path, db_seq_len, seq_len, augment)
            * shard   : (rank, world) -- build extension for data-parallel runs.  ``batch_size`` is the PER-RANK batch;
                        a global batch is ``world`` consecutive per-rank batches of the (identically seeded, identically
                        shuffled) chunk list and rank r reads, decompresses and decodes only its own slice.  Every rank
                        gets the same number of batches (the remainder that does not fill a global batch is dropped,
                        like ``batch(drop_remainder=True)``), so per-step collectives never deadlock. '''
        if out_size is None:
            self._set_output_size()
        else:
            self._set_output_size(out_size=out_size)
        self.settings = settings
        self.records_path = settings.records_path
        self.db_path = settings.db_path_config[self.db_name]
        self.db_seq_len = self.settings.db_seq_len
        self.seq_len = self.settings.seq_len
        self.batch_size = batch_size
        self.usecase = usecase
        self.device = torch.device(device) if device is not None else \
            torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        # two independent generators: the plan (shuffle, _cut_sequence offsets) is drawn in the prefetch thread, the
        # augmentation parameters in the consumer thread -- a shared Generator is neither thread-safe nor reproducible.
        # The plan generator is the same on every rank (same seed); the augmentation generator is per rank.
        self.shard_rank, self.shard_world = int(shard[0]), int(shard[1])
        if not 0 <= self.shard_rank < self.shard_world:
            raise ValueError(f"shard={shard}: rank must be in [0, world)")
        plan_seq, aug_seq = np.random.SeedSequence(seed).spawn(2)
        self.plan_rng = np.random.default_rng(plan_seq)
        self.rng = np.random.default_rng(aug_seq.spawn(self.shard_world)[self.shard_rank])
        self.prefetch = prefetch
        self.streaming = False
        if usecase == "train" and (self.db_seq_len is None or self.seq_len is None):
            raise Exception('db_seq_len and seq_len must be defined in train mode')
        if not (self.db_seq_len is None or self.seq_len is None) and self.db_seq_len < self.seq_len:
            raise Exception('db_seq_len must be larger or equal than seq_len')
        try:
            function = self.build_functions[usecase]
        except KeyError:
            raise Exception('Usecase "%s" not implemented for this dataloader' % usecase)
        self.dataset = function()
        self.length = self.dataset.cardinality()
        return self.dataset

    def _get_trajectories(self):
        csv_files = sorted(glob.glob(os.path.join(self.records_path, "**/*.csv"), recursive=True))
        trajectories = [read_trajectory_csv(f) for f in csv_files]
        if trajectories == []:
            raise Exception("No csv files found at the given path: %s" % self.records_path)
        return trajectories

    def _build_train_dataset(self):
        self.augment = self.settings.augment
        chunks = []
        for traj in self._get_trajectories():                                  # batch(db_seq_len, drop_remainder)
            for i in range(0, len(traj) - self.db_seq_len + 1, self.db_seq_len):
                chunks.append(traj[i:i + self.db_seq_len])
        rank, world, bs = self.shard_rank, self.shard_world, self.batch_size
        n_batches = len(chunks) // (bs * world)                                 # global batches: the same count on every rank

        def plan():                                                             # one epoch
            order = self.plan_rng.permutation(len(chunks))                      # shuffle(cardinality, reshuffle_each_iteration)
            for bi in range(n_batches):
                glob_ids = order[bi * bs * world:(bi + 1) * bs * world]
                offs = self.plan_rng.integers(0, self.db_seq_len - self.seq_len + 1, size=len(glob_ids))   # _cut_sequence (:148-158)
                yield [chunks[ci][int(off):int(off) + self.seq_len]
                       for ci, off in zip(glob_ids[rank * bs:(rank + 1) * bs], offs[rank * bs:(rank + 1) * bs])]

        return SequenceDataset(self, plan, n_batches, self.prefetch)

    def _build_eval_dataset(self):
        self.augment = False
        trajectories = self._get_trajectories()
        if self.db_seq_len is not None:
            print("Evaluating on subsequences of length %i" % self.db_seq_len)
            self.seq_len = self.db_seq_len
            chunks = []
            for traj in trajectories:
                for i in range(0, len(traj) - self.db_seq_len + 1, self.db_seq_len):
                    chunks.append(traj[i:i + self.db_seq_len])
            rank, world, bs = self.shard_rank, self.shard_world, self.batch_size
            n_batches = len(chunks) // (bs * world)

            def plan():
                for bi in range(n_batches):
                    lo = (bi * world + rank) * bs
                    yield chunks[lo:lo + bs]
            return SequenceDataset(self, plan, n_batches, self.prefetch)
        # streaming: one frame at a time, batch 1, trajectories back to back (:137-138)
        if self.shard_world > 1:
            raise ValueError("streaming evaluation (db_seq_len None) is sequential: it cannot be sharded")
        self.streaming = True
        frames = [row for traj in trajectories for row in traj]

        def plan():
            for row in frames:
                yield [[row]]
        return SequenceDataset(self, plan, len(frames), self.prefetch)

    # -- augmentation (device tensors; one random draw per sequence, as in the reference) --------
    def _uniform(self, lo, hi):
        return float(self.rng.uniform(lo, hi))

    def _augmentation_step_color(self, invert_color=True):
        ''' Perform data augmentation on the colors (dataloaders/generic.py:186-206) '''
        from . import color
        im = self.out_data["RGB_im"]
        if self.usecase == "finetune":
            b, c, s, h = 0.2, (0.8, 1.2), (0.8, 1.2), 0.2
        else:
            b, c, s, h = 0.2, (0.75, 1.25), (0.75, 1.25), 0.4
        im = color.adjust_brightness(im, self._uniform(-b, b))
        im = color.adjust_contrast(im, self._uniform(*c))
        im = color.adjust_saturation(im, self._uniform(*s))
        im = color.adjust_hue(im, self._uniform(-h, h))
        if invert_color and self._uniform(0., 1.) < 0.5:
            im = 1. - im
        self.out_data["RGB_im"] = im

    def _augmentation_step_flip(self):
        ''' Perform data augmentation on the orientation of the images (dataloaders/generic.py:208-259)
            WARNING : works only with quaternion rotations '''
        im_col, im_depth = self.out_data["RGB_im"], self.out_data["depth"]
        rot, trans, c = self.out_data["rot"], self.out_data["trans"], self.out_data["camera"]["c"]
        h, w = im_col.shape[1:3]
        dev = im_col.device
        if self._uniform(0., 1.) < 0.5:                                         # vertical flip
            im_col, im_depth = torch.flip(im_col, dims=[1]), torch.flip(im_depth, dims=[1])
            rot = rot * torch.tensor([[1., -1., 1., -1.]], device=dev)
            trans = trans * torch.tensor([[1., -1., 1.]], device=dev)
            c = torch.stack([c[0], h - c[1]])
        if self._uniform(0., 1.) < 0.5:                                         # horizontal flip
            im_col, im_depth = torch.flip(im_col, dims=[2]), torch.flip(im_depth, dims=[2])
            rot = rot * torch.tensor([[1., 1., -1., -1.]], device=dev)
            trans = trans * torch.tensor([[-1., 1., 1.]], device=dev)
            c = torch.stack([w - c[0], c[1]])
        self.out_data["camera"]["c"] = c
        self.out_data["depth"] = im_depth.contiguous()
        self.out_data["RGB_im"] = im_col.contiguous()
        self.out_data["rot"] = rot
        self.out_data["trans"] = trans
