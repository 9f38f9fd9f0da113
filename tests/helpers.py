"""Shared helpers for the parity tests (the oracle is the checker, never the thing under test)."""
import numpy as np


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same_bits(a, b):
    """Bit-exact float comparison (NaNs compare equal to NaNs)."""
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and bool(np.all((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))))


def to_dev(torch, dev, arr):
    """numpy (any dtype, incl. structured) -> device byte/typed tensor."""
    arr = np.ascontiguousarray(arr)
    if arr.dtype.fields is not None:
        return torch.from_numpy(arr.view(np.uint8).reshape(len(arr), arr.dtype.itemsize)).to(dev)
    return torch.from_numpy(arr).to(dev)


def make_pair(S, gpu, scene):
    torch, dev, ctx = gpu
    n = len(scene["sift"])
    d_sift = to_dev(torch, dev, scene["sift"])
    pair = S.ImagePair(ctx, scene["K"], scene["Kinv"], 2, n)
    pair.fillXU(d_sift)
    return pair, d_sift
