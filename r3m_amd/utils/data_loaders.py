"""Batch sources for train_representation.py.

R3MBuffer mirrors the Ego4D sampler of /root/reference/r3m/utils/data_loaders.py:38-109 (frame-index sampling at :66-79 and
the rc / rctraj crop semantics at :81-102) but decodes with PIL instead of torchvision.io and leaves the crop to the GPU
(r3m_amd/augment.py); SyntheticBuffer yields seeded random clips of the same shape for bring-up, tests and bench.
Each item: (frames [5,3,H,W] uint8 ordered (start, goal, s0, s1, s2), label:str); the training loop casts to float32 0..255
(the encoder's input convention) on the GPU."""
import random

import numpy as np
import torch
from torch.utils.data import IterableDataset


class SyntheticBuffer(IterableDataset):
    def __init__(self, seed=0, labels=("open the drawer", "pick up the cup", "")):
        self.seed = seed
        self.labels = labels

    def __iter__(self):
        info = torch.utils.data.get_worker_info()
        g = torch.Generator().manual_seed(self.seed + (info.id if info else 0) * 7919)
        while True:
            im = torch.randint(0, 256, (5, 3, 224, 224), generator=g, dtype=torch.uint8)   # uint8 over PCIe, float on the GPU
            label = self.labels[int(torch.randint(0, len(self.labels), (1,), generator=g))]
            yield im, label


def sample_indices(vidlen, alpha, rng=np.random):
    """Frame indices of one clip, exactly the draws of data_loaders.py:73-79 (1-based, s2 may equal s1)."""
    start_ind = rng.randint(1, 2 + int(alpha * vidlen))
    end_ind = rng.randint(int((1 - alpha) * vidlen) - 1, vidlen)
    s1_ind = rng.randint(2, vidlen)
    s0_ind = rng.randint(1, s1_ind)
    s2_ind = rng.randint(s1_ind, vidlen + 1)
    return start_ind, end_ind, s0_ind, s1_ind, s2_ind


class R3MBuffer(IterableDataset):
    def __init__(self, ego4dpath, num_workers, source1, source2, alpha, datasources, doaug="none"):
        import pandas as pd
        self._num_workers = max(1, num_workers)
        self.alpha = alpha
        self.data_sources = datasources
        self.doaug = doaug     # applied on the GPU by the training loop (augment.random_resized_crop), not here
        if "ego4d" not in self.data_sources:
            raise NameError('Invalid Dataset')
        self.manifest = pd.read_csv(f"{ego4dpath}manifest.csv")
        self.ego4dlen = len(self.manifest)

    @staticmethod
    def _read(vid, index):
        from PIL import Image
        with Image.open(f"{vid}/{index:06}.jpg") as im:
            a = np.asarray(im.convert("RGB"), dtype=np.uint8)
        return torch.from_numpy(a).permute(2, 0, 1)

    def _sample(self):
        random.choice(self.data_sources)
        vidid = np.random.randint(0, self.ego4dlen)
        m = self.manifest.iloc[vidid]
        vidlen, txt, vid = m["len"], m["txt"], m["path"]
        label = txt[2:]   # cuts off the "C " prefix (data_loaders.py:69)
        idx = sample_indices(vidlen, self.alpha)
        im = torch.stack([self._read(vid, i) for i in idx])   # uint8 [5,3,H,W]; cast / crop happen on the GPU
        return im, label

    def __iter__(self):
        while True:
            yield self._sample()
