"""Loader with the reference's surface (MML_ZYC/dataLoader/DataLoader.py:10-156) over (image, token_ids, mask, label)
samples: `MultimodalDataLoader(file_path, batch_size=64).load_data(test_subject_id) -> (contrastive_loader, train_loader,
test_loader)`; tuple batches `(x1, x2, x3, arousal, valence)` (DataLoader.py:152, train.py:101) and 7-tuple contrastive
batches of two views + label (DataLoader.py:133-137, train.py:60); `dict_loader()` gives the `(data_dict, labels)`
form that Trainer/Tester consume (data/Dataset.py:65-67). There is no dataset offline: `file_path=None` (or a missing
file) yields seeded synthetic pairs of the benchmark shape; an `.npz` with arrays image/token_ids/attention_mask/
arousal/valence/subject is used when given. The MAHNOB-HCI pickle format of the reference is out of scope."""
import os

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset, TensorDataset


class SyntheticPairs(Dataset):
    """Seeded synthetic (image, token_ids, attention_mask, arousal, valence) samples; BASELINE.md §3 generators."""

    def __init__(self, n=480, seq_len=128, image_size=224, vocab=30522, seed=1234, subjects=24):
        g = torch.Generator().manual_seed(seed)
        self.image = torch.randn(n, 3, image_size, image_size, generator=g)
        self.ids = torch.randint(0, vocab, (n, seq_len), generator=g)
        self.ids[:, 0] = 101 % vocab
        self.mask = torch.ones(n, seq_len)
        self.arousal = torch.randint(0, 3, (n,), generator=g)
        self.valence = torch.randint(0, 3, (n,), generator=g)
        self.subject = torch.arange(n) % subjects + 1

    def __len__(self):
        return self.image.shape[0]

    def __getitem__(self, i):
        return self.image[i], self.ids[i], self.mask[i], self.arousal[i], self.valence[i]


class _DictView(Dataset):
    def __init__(self, tensors):
        self.t = tensors

    def __len__(self):
        return self.t[0].shape[0]

    def __getitem__(self, i):
        return {"image": self.t[0][i], "text": self.t[1][i], "mask": self.t[2][i]}, self.t[3][i]


class MultimodalDataLoader:
    def __init__(self, file_path=None, batch_size=64, **synthetic_kw):
        self.batch_size = batch_size
        if file_path is not None and os.path.exists(file_path) and file_path.endswith(".npz"):
            d = np.load(file_path)
            self.data = tuple(torch.from_numpy(d[k]) for k in ("image", "token_ids", "attention_mask", "arousal",
                                                                "valence"))
            self.subject = torch.from_numpy(d["subject"])
        else:
            s = SyntheticPairs(**synthetic_kw)
            self.data = (s.image, s.ids, s.mask, s.arousal, s.valence)
            self.subject = s.subject

    def _split(self, test_subject_id):
        te = self.subject == test_subject_id
        tr = ~te
        return tuple(t[tr] for t in self.data), tuple(t[te] for t in self.data)

    def _contrastive(self, train, seed=0):
        """Two views per sample (image noise, token dropout to [MASK]=103) + the arousal label (train.py:60,69)."""
        g = torch.Generator().manual_seed(seed)
        img, ids, mask, arousal, _ = train
        img2 = img + 0.1 * torch.randn(img.shape, generator=g)
        drop = torch.rand(ids.shape, generator=g) < 0.1
        drop[:, 0] = False
        ids2 = torch.where(drop, torch.full_like(ids, 103), ids)
        return TensorDataset(img, ids, mask, img2, ids2, mask, arousal)

    def load_data(self, test_subject_id):
        train, test = self._split(test_subject_id)
        contrastive_loader = DataLoader(self._contrastive(train), batch_size=self.batch_size, shuffle=True, pin_memory=True)
        train_loader = DataLoader(TensorDataset(*train), batch_size=self.batch_size, shuffle=True, pin_memory=True)
        test_loader = DataLoader(TensorDataset(*test), batch_size=self.batch_size, shuffle=False, pin_memory=True)
        return contrastive_loader, train_loader, test_loader

    def dict_loader(self, test_subject_id, train=True):
        tr, te = self._split(test_subject_id)
        t = tr if train else te
        return DataLoader(_DictView(t[:4]), batch_size=self.batch_size, shuffle=train, pin_memory=True)


class DevicePrefetcher:
    """N4: double-buffered host -> device input pipeline. Wraps any loader of tensor tuples / (dict, labels)
    batches: the next batch is copied from pinned host memory on a side HIP stream while the current one trains, the
    consumer's stream waits on the copy's event only. The reference's loaders are 0-worker, pin_memory=True
    (DataLoader.py:144-156) and copy synchronously inside the step (Trainer.py:53-56)."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None

    def __len__(self):
        return len(self.loader)

    @property
    def dataset(self):
        return self.loader.dataset

    def _move(self, obj):
        if torch.is_tensor(obj):
            src = obj if (obj.is_pinned() or self.stream is None) else obj.pin_memory()
            return src.to(self.device, non_blocking=True)
        if isinstance(obj, dict):
            return {k: self._move(v) for k, v in obj.items()}
        if isinstance(obj, (tuple, list)):
            return type(obj)(self._move(v) for v in obj)
        return obj

    @staticmethod
    def _tensors(obj):
        if torch.is_tensor(obj):
            yield obj
        elif isinstance(obj, dict):
            for v in obj.values():
                yield from DevicePrefetcher._tensors(v)
        elif isinstance(obj, (tuple, list)):
            for v in obj:
                yield from DevicePrefetcher._tensors(v)

    def __iter__(self):
        if self.stream is None:
            yield from self.loader
            return
        it = iter(self.loader)

        def fetch():
            try:
                host = next(it)
            except StopIteration:
                return None
            with torch.cuda.stream(self.stream):
                dev = self._move(host)
                ev = torch.cuda.Event()
                ev.record(self.stream)
            return dev, ev

        nxt = fetch()
        while nxt is not None:
            batch, ev = nxt
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for t in self._tensors(batch):
                t.record_stream(cur)  # the side stream's allocation is consumed on the compute stream
            nxt = fetch()             # the copy of batch k+1 overlaps the step on batch k
            yield batch
