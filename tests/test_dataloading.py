"""`evfly_amd.dataloading` (host logic, no GPU) against golden G13 = the reference's own dataloader / preload run on
tests/golden/mini_dataset (make_golden.py g13): folder parsing, timestamp matching, duplicate / collision handling,
seeded shuffle, split, return layout. Conditioning configurations (resize / rescale / cutoff run on the device) are in
test_gpu_dataloading.py."""
import os

import numpy as np
import pytest
import torch

from evfly_amd import dataloading as dl

from _util import golden

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mini_dataset")
quiet = lambda *a: None


def check(tag, res, g, exact_evs=True, tol=0.0):
    for part, tup in (("train", res[0]), ("val", res[1])):
        meta, (ims, depths), lens, desvel, evs, folders, ids = tup[:7]
        assert torch.equal(meta, torch.from_numpy(g[f"{tag}_{part}_meta"])), (tag, part, "meta")
        assert meta.dtype == torch.float32
        for name, got in (("ims", ims), ("depths", depths)):
            want = torch.from_numpy(g[f"{tag}_{part}_{name}"])
            assert got.shape == want.shape and got.dtype == torch.float32, (tag, part, name)
            assert (got - want).abs().max().item() <= tol if want.numel() else True, (tag, part, name)
        assert np.array_equal(np.asarray(lens), g[f"{tag}_{part}_lens"])
        assert torch.equal(desvel, torch.from_numpy(g[f"{tag}_{part}_desvel"]))
        assert [os.path.basename(os.path.normpath(f)) for f in folders] == list(g[f"{tag}_{part}_folders"])
        assert np.array_equal(np.asarray(ids), g[f"{tag}_{part}_ids"])
        n_evs = len([k for k in g.files if k.startswith(f"{tag}_{part}_evs")])
        assert (0 if evs is None else len(evs)) == n_evs
        for k in range(n_evs):
            want = torch.from_numpy(g[f"{tag}_{part}_evs{k}"])
            got = evs[k].float()
            assert got.shape == want.shape
            if exact_evs:
                assert torch.equal(got, want), (tag, part, k)
        if len(tup) > 7:
            assert [len(u) for u in tup[7]] == list(g[f"{tag}_{part}_unmatched"])
    assert bool(res[2]) == bool(g[f"{tag}_flag"])


def test_folder_dataset_matches_reference_dataloader():
    g = golden("g13_dataloader")
    res = dl.dataloader(ROOT, val_split=0.25, short=0, seed=3, do_transform=False, events="evs_frames", logger=quiet, use_h5=False,
                        return_unmatched=True)
    check("a", res, g)
    # what the loader had to cope with in this dataset: one trajectory dropped for its collision, one image without a
    # metadata row (and one duplicated metadata timestamp) in trajectory 0001
    lens = list(res[0][2]) + list(res[1][2])
    assert sorted(lens) == [5, 5, 6]
    assert sum(sum(len(u) for u in t[7]) for t in res[:2]) == 1


def test_preload_matches_reference_rules():
    res = dl.dataloader(ROOT, val_split=0.25, seed=3, do_transform=False, events="evs_frames", logger=quiet, use_h5=False)
    meta, (ims, depths), lens, desvel, evs, _, _ = res[0]
    m2, i2, d2, v2, e2, none = dl.preload((meta, ims, depths, desvel, evs, None), "cpu")
    assert none is None and torch.equal(m2, meta) and torch.equal(i2, ims) and torch.equal(d2, depths) and torch.equal(v2, desvel)
    assert isinstance(e2, list) and len(e2) == len(evs) and all(t.dtype == torch.float32 for t in e2)
    arr = dl.preload((np.arange(6, dtype=np.float64).reshape(2, 3),), "cpu")[0]
    assert arr.dtype == torch.float64 and arr.shape == (2, 3)                      # ndarray keeps its dtype


def test_h5_groups_follow_the_same_rules():
    """utils/to_h5.py:35-43 groups (`data`, `ims`, `depths`, `evs`) through the loader body with a stand-in for the
    h5py.File mapping (h5py itself is not installed in this image)."""
    folder = dl.dataloader(ROOT, val_split=0.0, seed=-2, do_transform=False, events="evs_frames", logger=quiet, use_h5=False,
                           keep_collisions=True)

    class DS:                                   # dataset: supports [()] and np.array(...)
        def __init__(self, a): self.a = np.asarray(a)
        def __getitem__(self, k): return self.a[k] if k != () else self.a
        def __array__(self, dtype=None, copy=None): return self.a.astype(dtype) if dtype else self.a
        @property
        def shape(self): return self.a.shape

    class File(dict):
        closed = False
        def close(self): self.closed = True

    meta, (ims, depths), lens, desvel, evs, folders, ids = folder[0]
    h5, s = File(), 0
    for k, n in enumerate(lens):
        h5[os.path.basename(os.path.normpath(folders[k]))] = dict(data=DS(meta[s:s + n].numpy()), ims=DS(ims[s:s + n].numpy()),
                                                                   depths=DS(depths[s:s + n].numpy()), evs=DS(evs[k].numpy()))
        s += n
    got = dl._load(ROOT, h5, 0.0, 0, -2, None, False, "evs_frames.npy", True, False, quiet, False, None, "train-val", 0.0, 0.0,
                   None, None)
    assert h5.closed and got[2] is True
    m2, (i2, d2), l2, v2, e2, f2, id2 = got[0]
    assert torch.equal(m2, meta) and torch.equal(i2, ims) and torch.equal(d2, depths) and np.array_equal(l2, lens)
    assert torch.equal(v2, desvel) and all(torch.equal(a, b) for a, b in zip(e2, evs))


def test_missing_h5py_is_a_clear_error(tmp_path):
    d = tmp_path / "ds"
    d.mkdir()
    (tmp_path / "ds.h5").write_bytes(b"\x89HDF\r\n\x1a\n")
    with pytest.raises(RuntimeError, match="h5py"):
        dl.dataloader(str(d), seed=-2, do_transform=False, logger=quiet)


def test_learner_dataset_only_mode(tmp_path):
    """`Learner(dataset_name=..., no_model=True)` -- what utils/to_h5.py:100 constructs: fields, velocity-command columns,
    train_val_dirs.npy in the workspace."""
    from evfly_amd.learner import Learner
    ln = Learner(dataset_name=ROOT, short=0, no_model=True, val_split=0.0, events="evs_frames", do_transform=False, use_h5=False,
                 workspace=str(tmp_path / "ws"))
    assert ln.model is None and ln.num_val_steps == 0 and ln.num_training_steps == 4          # keep_collisions=True here
    assert ln.train_ims.shape == (int(ln.train_trajlength.sum()), 26, 34) and ln.train_depths.shape == ln.train_ims.shape
    assert torch.equal(ln.train_velcmd, ln.train_meta[:, 13:16]) and ln.train_desvel.shape == (ln.train_ims.shape[0],)
    assert len(ln.train_evs) == 4 and all(e.dtype == torch.float32 for e in ln.train_evs)
    tvd = np.load(os.path.join(ln.workspace, "train_val_dirs.npy"), allow_pickle=True)
    assert list(tvd[0]) == ln.train_dirs and list(tvd[2]) == ln.train_dirs_ids and len(tvd[1]) == 0
    with pytest.raises(NotImplementedError):
        ln.train()
