"""Frame sources for the extractor (reference: infer/src/dataset.py:101-153 D_vsc, and
infer/src/transform.py:37-42 vit_transform).  torchvision is not needed: Resize on a PIL image
is PIL's own bicubic resize, ToTensor is /255 and Normalize(0.5, 0.5) is 2x-1."""
from __future__ import annotations

import io
import os
from typing import List, Sequence
from zipfile import ZipFile

import numpy as np
import torch
from torch.nn.utils.rnn import pad_sequence


def vit_transform(width: int, height: int):
    from PIL import Image

    def apply(img) -> torch.Tensor:
        img = img.convert("RGB").resize((height, width), Image.BICUBIC)   # Resize([w, h]) = (rows, cols)
        x = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float() / 255.0
        return (x - 0.5) / 0.5
    return apply


def vit_transform_u8(width: int, height: int):
    """The resize of ``vit_transform`` only: uint8 [H, W, 3].  ToTensor + Normalize(0.5, 0.5) then run inside the encoder's
    patchify kernel (HipEncoder / SwinHipEncoder take uint8 [n,H,W,C]) -- same descriptors, a quarter of the bytes."""
    from PIL import Image

    def apply(img) -> torch.Tensor:
        return torch.from_numpy(np.asarray(img.convert("RGB").resize((height, width), Image.BICUBIC), dtype=np.uint8).copy())
    return apply


CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def clip_transform(size: int = 224):
    """Resize(size, bicubic) on the short side + CenterCrop(size) + ToTensor + CLIP normalisation
    (infer/extract_query_feats.py:97-105, the video-score model's input)."""
    from PIL import Image
    mean = torch.tensor(CLIP_MEAN).view(3, 1, 1)
    std = torch.tensor(CLIP_STD).view(3, 1, 1)

    def apply(img) -> torch.Tensor:
        img = img.convert("RGB")
        w, h = img.size
        if w <= h:
            nw, nh = size, int(size * h / w)     # torchvision Resize(int): short side -> size, long side truncated
        else:
            nw, nh = int(size * w / h), size
        img = img.resize((nw, nh), Image.BICUBIC)
        left, top = int(round((nw - size) / 2.0)), int(round((nh - size) / 2.0))
        img = img.crop((left, top, left + size, top + size))
        x = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float() / 255.0
        return (x - mean) / std
    return apply


def clip_transform_u8(size: int = 224):
    """Resize + CenterCrop of ``clip_transform`` only: uint8 [size, size, 3] (normalise with CLIP_MEAN / CLIP_STD on the GPU)."""
    from PIL import Image

    def apply(img) -> torch.Tensor:
        img = img.convert("RGB")
        w, h = img.size
        nw, nh = (size, int(size * h / w)) if w <= h else (int(size * w / h), size)
        img = img.resize((nw, nh), Image.BICUBIC)
        left, top = int(round((nw - size) / 2.0)), int(round((nh - size) / 2.0))
        return torch.from_numpy(np.asarray(img.crop((left, top, left + size, top + size)), dtype=np.uint8).copy())
    return apply


class ZipFrames(torch.utils.data.Dataset):
    """One item = all frames of one video, read from <prefix>/<vid[-2:]>/<vid>.zip of jpgs."""

    def __init__(self, video_ids: Sequence[str], zip_prefix: str, transform):
        self.zip_prefix, self.transform = zip_prefix, transform
        self.video_ids = [v for v in video_ids if os.path.exists(self._path(v))]

    def _path(self, vid):
        return "%s/%s/%s.zip" % (self.zip_prefix, vid[-2:], vid)

    def __len__(self):
        return len(self.video_ids)

    def __getitem__(self, i):
        from PIL import Image
        vid = self.video_ids[i]
        with ZipFile(self._path(vid), "r") as z:
            frames = [self.transform(Image.open(io.BytesIO(z.read(n)))) for n in sorted(z.namelist())]
        return torch.stack(frames), vid


class TensorFrames(torch.utils.data.Dataset):
    """Already decoded videos: a list of (frames [S,3,H,W] float32, video_id)."""

    def __init__(self, videos: List):
        self.videos = videos

    def __len__(self):
        return len(self.videos)

    def __getitem__(self, i):
        return self.videos[i]


def collate_fn(batch):
    frames, vids = zip(*batch)
    lengths = torch.tensor([f.shape[0] for f in frames])
    frames = pad_sequence(frames, batch_first=True, padding_value=0.0)
    mask = (torch.arange(frames.shape[1])[None, :] < lengths[:, None]).long()
    return frames, mask, vids
