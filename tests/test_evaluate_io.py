"""Sample readers of the evaluation harness (row f-1), host side only: the `.h5` branch restates the reference's h5_loader
(dataloaders/dataloader.py:8-13: rgb = transpose(h5f['rgb'], (1, 2, 0)), depth = h5f['depth']).  h5py is not in this image: the branch is
executed through a stand-in module exposing h5py.File's read interface over in-memory arrays; with a real h5py the same test writes and
reads a real file."""
import argparse
import os
import sys
import types

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "fast-depth_amd"))


def _frames(seed=3):
    g = np.random.default_rng(seed)
    return g.integers(0, 256, (480, 640, 3), dtype=np.uint8), (g.random((480, 640), dtype=np.float32) * 9 + 0.7).astype(np.float32)


def _args(path):
    return argparse.Namespace(samples=str(path), repeat=2)


def test_h5_branch_through_a_stand_in_h5py(tmp_path, monkeypatch):
    import evaluate as fd_eval
    rgb, depth = _frames()
    store = {"rgb": np.ascontiguousarray(rgb.transpose(2, 0, 1)), "depth": depth}      # the reference's files hold rgb as [3, 480, 640]

    class File:
        def __init__(self, name, mode="r"):
            assert mode == "r" and name.endswith(".h5")

        def __enter__(self):
            return store

        def __exit__(self, *a):
            return False

    fake = types.ModuleType("h5py")
    fake.File = File
    monkeypatch.setitem(sys.modules, "h5py", fake)
    d = tmp_path / "val" / "official"
    d.mkdir(parents=True)
    (d / "00001.h5").write_bytes(b"")                                  # only the name matters to the stand-in
    np.savez(str(tmp_path / "00002.npz"), rgb=rgb, depth=depth)
    out = fd_eval.load_samples(_args(tmp_path))
    assert len(out) == 2
    for r, dd in out:                                                  # raw frames stay uint8 HWC + float32 depth (val_transform runs on the GPU)
        assert r.dtype == torch.uint8 and tuple(r.shape) == (480, 640, 3) and np.array_equal(r.numpy(), rgb)
        assert dd.dtype == torch.float32 and np.array_equal(dd.numpy(), depth)


def test_h5_without_h5py_is_an_explicit_error(tmp_path, monkeypatch):
    import evaluate as fd_eval
    monkeypatch.setitem(sys.modules, "h5py", None)                     # import h5py -> ImportError
    f = tmp_path / "00001.h5"
    f.write_bytes(b"")
    with pytest.raises(RuntimeError, match="h5py"):
        fd_eval.load_samples(_args(f))


def test_h5_branch_with_real_h5py(tmp_path):
    h5py = pytest.importorskip("h5py", reason="h5py is not installed in this image (the stand-in test above executes the branch)")
    import evaluate as fd_eval
    rgb, depth = _frames(4)
    with h5py.File(str(tmp_path / "00001.h5"), "w") as f:
        f.create_dataset("rgb", data=rgb.transpose(2, 0, 1)); f.create_dataset("depth", data=depth)
    (r, dd), = fd_eval.load_samples(_args(tmp_path))
    assert np.array_equal(r.numpy(), rgb) and np.array_equal(dd.numpy(), depth)


def test_default_and_network_resolution_samples(tmp_path):
    import evaluate as fd_eval
    out = fd_eval.load_samples(argparse.Namespace(samples="", repeat=3))
    assert len(out) == 3 and tuple(out[0][0].shape) == (3, 224, 224) and tuple(out[0][1].shape) == (1, 224, 224)
    g = np.random.default_rng(0)
    np.savez(str(tmp_path / "a.npz"), rgb=g.integers(0, 256, (224, 224, 3), dtype=np.uint8), depth=g.random((224, 224), dtype=np.float32))
    (r, dd), = fd_eval.load_samples(_args(tmp_path / "a.npz"))
    assert r.dtype == torch.float32 and tuple(r.shape) == (3, 224, 224) and float(r.max()) <= 1.0 and tuple(dd.shape) == (1, 224, 224)
