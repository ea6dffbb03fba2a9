"""Seeded stand-in for the PoseTrack21 / JRDB video datasets (item contract of posetrack21.py:131-224).

Each item is the 11-tuple the active-learning loops consume:
  (idx, stacked_inp (3,3,H,W) [current, prev, next], label (J,h,w), label_mask (J,1,1), GTkpt (3J,),
   img_id, ann_id, bbox_crop (4,) xyxy, bbox_ann (4,) xyxy, isPrev, isNext)
Items are id-sorted, tracks are contiguous, and the prev/next crops of an item ARE the current crops of its
neighbours (what eval-mode cropping produces) — declared by ``ID_SORTED_STREAM``.
"""
from __future__ import annotations

import numpy as np
import torch
from torch.utils.data import Dataset

from alphapose.models.builder import DATASET


@DATASET.register_module
class SyntheticVideo(Dataset):
    EVAL_JOINTS = list(range(17))
    ID_SORTED_STREAM = True
    joint_pairs = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]

    def __init__(self, train=False, get_prenext=True, PRESET=None, NUM_ITEMS=64, TRACKS=2, SEED=166, **_):
        self.train, self.get_prenext = train, get_prenext
        self.n, self.tracks = int(NUM_ITEMS), int(TRACKS)
        preset = PRESET or {}
        self.H, self.W = preset.get("IMAGE_SIZE", [256, 192])
        self.h, self.w = preset.get("HEATMAP_SIZE", [64, 48])
        self.sigma = preset.get("SIGMA", 2)
        self.J = preset.get("NUM_JOINTS", 17)
        r = np.random.RandomState(SEED)
        wbox = r.uniform(60, 240, self.n)
        self.bbox = np.stack([np.full(self.n, 100.0), np.full(self.n, 50.0), 100 + wbox, 50 + wbox * 4 / 3], 1).astype(np.float32)
        self.joints_hm = np.stack([r.randint(3, self.w - 3, (self.n, self.J)), r.randint(3, self.h - 3, (self.n, self.J))], 2)
        self.vis = (r.random_sample((self.n, self.J)) > 0.2).astype(np.float32)
        self.seed = SEED
        self.emit_neighbour_crops = True       # False: (1,3,H,W) stacks — the consumer reads the neighbours from the stream
        per = self.n // self.tracks
        pos = np.arange(self.n) % per
        self.is_prev, self.is_next = pos != 0, pos != per - 1

    def __len__(self):
        return self.n

    def _crop(self, i):
        g = torch.Generator().manual_seed(self.seed * 100003 + int(i))
        x = torch.rand((3, self.H, self.W), generator=g)
        return x - torch.tensor([0.406, 0.457, 0.480]).view(3, 1, 1)

    def __getitem__(self, i):
        cur = self._crop(i)
        if self.emit_neighbour_crops:
            zero = torch.zeros_like(cur)
            prev = self._crop(i - 1) if (self.get_prenext and self.is_prev[i]) else zero
            nxt = self._crop(i + 1) if (self.get_prenext and self.is_next[i]) else zero
            stack = torch.stack([cur, prev, nxt])
        else:
            stack = cur[None]
        label = torch.zeros((self.J, self.h, self.w))
        rad = int(3 * self.sigma)
        gk = torch.arange(-rad, rad + 1, dtype=torch.float32)
        patch = torch.exp(-(gk[None, :] ** 2 + gk[:, None] ** 2) / (2 * self.sigma ** 2))
        for j in range(self.J):
            mx, my = int(self.joints_hm[i, j, 0]), int(self.joints_hm[i, j, 1])
            x0, x1, y0, y1 = max(0, mx - rad), min(self.w, mx + rad + 1), max(0, my - rad), min(self.h, my + rad + 1)
            label[j, y0:y1, x0:x1] = patch[y0 - (my - rad):y1 - (my - rad), x0 - (mx - rad):x1 - (mx - rad)]
        mask = torch.from_numpy(self.vis[i]).view(self.J, 1, 1)
        bb = self.bbox[i]
        scale = (bb[2] - bb[0]) / self.w
        gx = bb[0] + (bb[2] - bb[0]) * 0.5 + (self.joints_hm[i, :, 0] - self.w / 2) * scale
        gy = bb[1] + (bb[3] - bb[1]) * 0.5 + (self.joints_hm[i, :, 1] - self.h / 2) * scale
        gt = torch.from_numpy(np.stack([gx, gy, self.vis[i]], 1).reshape(-1).astype(np.float32))
        return (i, stack, label, mask, gt, 1000 + i, 100000 + i, torch.from_numpy(bb), torch.from_numpy(bb),
                bool(self.is_prev[i]), bool(self.is_next[i]))

    @staticmethod
    def my_collate_fn(batch):
        cols = list(zip(*batch))
        return (list(cols[0]), torch.stack(cols[1]), torch.stack(cols[2]), torch.stack(cols[3]), torch.stack(cols[4]),
                list(cols[5]), list(cols[6]), torch.stack(cols[7]), torch.stack(cols[8]), list(cols[9]), list(cols[10]))
