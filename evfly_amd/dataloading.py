"""`dataloader` / `preload` -- the dataset side of the reference's Python surface (learner/dataloading.py:30-560,
570-582), feeding `Learner.run_model` with the 7-tuples the reference returns.

Host part (this file, plain Python / numpy like the reference's): trajectory folders (`data.csv`, `*_im.png` /
`*_depth.png` or legacy `*.png` / `*.npy` images, an `evs_frames*.npy` object array written by to_events.py) or the
`<dataset>[_tf].h5` trajectory groups of utils/to_h5.py:35-43 (`data, ims, depths, evs` per group), timestamp
matching, duplicate / NaN / collision filtering, shuffling, `short`, train / val split -- same rules, same order,
same return layout.

Device part: everything the reference then does to the FRAMES runs on the MI355X through the C ABI --
  * `resize_input`: F.interpolate(bilinear, align_corners=False) of images, depths and event frames
    (dataloading.py:398-416)                                    -> evfly_resize_bilinear
  * `rescale_evs == -1`: per-frame 97th-percentile scale + clamp (dataloading.py:512-523)  -> evfly_condition_frames
    (the exact-quantile kernel of the deployment path, evfly_ros/run.py:250-253: same expression per frame)
  * `rescale_evs > 0`, `rescale_depth > 0`, `evs_min_cutoff`: elementwise, on the device tensors.
There is no CPU fallback for those: without a GPU a call that asks for them raises. Deviation, stated once: the
reference keeps event frames in float64 (to_events.py saves float64) until `preload` casts them to float32; here
they are cast to float32 when loaded and conditioned in float32 (the model consumes float32 either way), so a
conditioned value can differ from the reference's by one float32 ulp. PNGs are decoded with PIL (cv2 is not a
dependency here); for 8-bit gray PNGs -- what the reference's datasets hold -- the bytes are identical.
"""
import csv  # noqa: F401  (kept for parity with the reference's import surface)
import glob
import os
import random
import re
import shutil
import time
from os.path import join as opj

import numpy as np
import torch

CROP_HEIGHT, CROP_WIDTH = 60, 90          # dataloading.py:46-47 (reported size when do_transform)


def find_unmatched_indices(list1, list2):
    """Indices of the entries of each list that do not occur in the other (dataloading.py:21-28)."""
    s1, s2 = set(list1), set(list2)
    return [i for i, v in enumerate(list1) if v not in s2], [i for i, v in enumerate(list2) if v not in s1]


def _read_gray_png(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("L"), dtype=np.uint8)


def _read_meta_csv(path, is_png):
    """data.csv without its header row; rows with a wrong column count are dropped on the slow path (:202-222)."""
    try:
        return np.genfromtxt(path, delimiter=',', dtype=np.float64)[1:]
    except Exception:
        rows = []
        with open(path, 'r') as fh:
            for line in fh.readlines()[1:]:
                cols = line.strip().split(',')
                if is_png and len(cols) != 21:
                    continue
                if not is_png:
                    raise NotImplementedError('This try-except for data.csv reading code is not yet implemented for non-png datasets.')
                rows.append([float(c) for c in cols])
        return np.array(rows, dtype=np.float64)


def _image_timestamp(fname, is_png, logger):
    base = os.path.basename(fname)
    if not is_png:
        return float(base[6:-4])                                    # legacy npy names (:292)
    m = re.search(r'(\d+(\.\d+)?)', base)                           # first numeric component (:295-307)
    if m is None or len(m.groups()) != 2:
        logger(f'[DATALOADER] More than one numeric component in image filename {fname} with basename {base}. Exiting.')
        raise SystemExit(1)
    return float(m.group(1))


# ---------------------------------------------------------------------------------------------- device side
def _gpu():
    from . import _lib
    _lib.lib()                                                      # raises without a GPU / the HIP library


def _resize_list(tensors, size):
    """bilinear(align_corners=False) of every (n, H, W) tensor of the list, on the device; float32 CPU tensors back
    (`.squeeze()` like the reference: a one-frame trajectory loses its leading axis there too)."""
    from . import voxelizer
    out = []
    for t in tensors:
        out.append(voxelizer.resize_bilinear(t.float(), size).cpu().squeeze())
    return out


def _condition_events(evs, rescale_evs, evs_min_cutoff, logger):
    """dataloading.py:497-533 on the device: fixed rescale or per-frame q97 rescale, clamp, low-value cutoff."""
    from . import voxelizer
    _gpu()
    pct = []
    out = []
    for ev in evs:
        x = ev.to("cuda", torch.float32)
        if rescale_evs > 0.0:
            x = torch.clamp(x / rescale_evs, -1.0, 1.0)
        elif rescale_evs == -1.0:
            # A one-frame trajectory arrives 2-D (`_resize_list` squeezes like the reference does). Deviation, on purpose:
            # the reference then takes the quantile per image ROW (`ev.view(ev.shape[0], -1)`, dataloading.py:518) and its
            # `(h,1,1)` broadcast blows the frame up to (h, h, w); here the lone frame is conditioned as ONE frame.
            shape = x.shape
            h, w = shape[-2], shape[-1]
            x, q = voxelizer.condition_frames(x.reshape(-1, h, w), out_hw=(h, w), quantile=0.97, return_q=True)
            x = x.reshape(shape)
            pct.append(float(q.mean()))
        if evs_min_cutoff is not None:
            x[x.abs() < evs_min_cutoff] = 0.0
        out.append(x.cpu())
    if pct:
        logger(f'[DATALOADER] Rescaling evs by 97th percentile of each frame, first and last traj 97th mean percentile '
               f'values are {pct[0]:.2f} and {pct[-1]:.2f}')
    return out


# ---------------------------------------------------------------------------------------------- dataloader
def dataloader(data_dir, val_split=0., short=0, seed=None, train_val_dirs=None, do_transform=True, events='',
               keep_collisions=False, return_unmatched=False, logger=None, do_clean_dataset=False, use_h5=True,
               resize_input=None, split_method='train-val', rescale_depth=0.0, rescale_evs=0.0, traj_ids=None,
               evs_min_cutoff=None):
    """learner/dataloading.py:30-560. Returns (train 7-tuple, val 7-tuple, is_png or found_h5) with
    7-tuple = (meta (N, cols) f32, (ims (N, h, w), depths | None), traj_lengths, desired_vels (N,), evs object array of
    (n_i, h, w) tensors | None, traj_folders, traj_folder_ids) -- 8-tuples with the unmatched image indices when
    `return_unmatched`."""
    if not os.path.isabs(data_dir):
        data_dir = opj(os.getcwd(), data_dir)
    if logger is None:
        logger = print
    if events != '' and '.' not in events:
        events += '_tf.npy' if do_transform else '.npy'

    # ---- <data_dir>[_tf].h5 next to the folder (:56-62)
    h5 = None
    h5_filename = data_dir + ('_tf' if (do_transform and '_tf' not in data_dir) else '') + '.h5'
    if os.path.exists(h5_filename) and use_h5:
        try:
            import h5py
        except ImportError as e:
            raise RuntimeError(f'[DATALOADER] {h5_filename} exists but h5py is not installed; pass use_h5=False to read '
                               f'the trajectory folders instead') from e
        logger(f'[DATALOADER] Found {h5_filename}, loading dataset')
        h5 = h5py.File(h5_filename, 'r')
    return _load(data_dir, h5, val_split, short, seed, train_val_dirs, do_transform, events, keep_collisions, return_unmatched,
                 logger, do_clean_dataset, resize_input, split_method, rescale_depth, rescale_evs, traj_ids, evs_min_cutoff)


def _load(data_dir, h5, val_split, short, seed, train_val_dirs, do_transform, events, keep_collisions, return_unmatched,
          logger, do_clean_dataset, resize_input, split_method, rescale_depth, rescale_evs, traj_ids, evs_min_cutoff):
    """Body of `dataloader` behind the h5 open (`h5`: None or any mapping group -> {'data','ims','depths','evs'} of
    array-likes supporting `[()]`, i.e. an h5py.File)."""
    found_h5 = h5 is not None
    dataset_name = os.path.basename(data_dir)

    # ---- which trajectories (:64-152)
    tvd_ok = False
    if train_val_dirs is not None:
        tvd_ok = any(dataset_name in f for f in list(train_val_dirs[0]) + list(train_val_dirs[1]))
    logger(f'[DATALOADER] train_val_dirs_is_invalid={not tvd_ok}')
    if train_val_dirs is not None and tvd_ok:
        tvd = [el.tolist() if isinstance(el, np.ndarray) else el for el in train_val_dirs]
        keep = [[], [], [], []]
        for k in range(len(tvd[0])):
            if dataset_name in tvd[0][k]:
                keep[0].append(tvd[0][k]); keep[2].append(tvd[2][k])
        for k in range(len(tvd[1])):
            if dataset_name in tvd[1][k]:
                keep[1].append(tvd[1][k]); keep[3].append(tvd[3][k])
        traj_folders = keep[0] + keep[1]
        val_split = len(keep[1]) / len(traj_folders)
        traj_folders_ids = np.hstack((keep[2], keep[3]))
    else:
        if not found_h5:
            traj_folders = sorted(glob.glob(opj(data_dir, '*/')))
        else:
            traj_folders = [opj(data_dir, g) for g in list(h5.keys())]
        traj_folders_ids = np.arange(len(traj_folders))
        if traj_ids is not None:
            traj_folders = traj_folders[traj_ids[0]:traj_ids[1]]
            traj_folders_ids = traj_folders_ids[traj_ids[0]:traj_ids[1]]
        if seed > -2:                       # like the reference: seed=None is a TypeError here, callers pass an int
            seed = int(time.time() * 1e3) if seed == -1 else seed
            random.seed(seed); random.shuffle(traj_folders)
            random.seed(seed); random.shuffle(traj_folders_ids)

    if short > 0:
        assert short <= len(traj_folders), f"short={short} is greater than the number of folders={len(traj_folders)}"
        traj_folders = traj_folders[:short]
        traj_folders_ids = traj_folders_ids[:short]
    elif short == -1:                       # the reference's hand-picked subset of one dataset (:144-155)
        logger('[DATALOADER] short=-1; Using special case for 2-26 dataset')
        sl = (slice(0, 30), slice(70, 100), slice(115, 145))
        traj_folders_ids = np.hstack([traj_folders_ids[s] for s in sl])
        traj_folders = [f for s in sl for f in traj_folders[s]]

    # ---- event frames of a folder dataset: one object array for the whole dataset (:157-173)
    evframes = None
    if not found_h5:
        if 'frames' in events:
            evframes = np.load(opj(data_dir, events), allow_pickle=True)
            if short != -1:
                evframes = evframes[traj_folders_ids]
            logger(f'[DATALOADER] Loaded event frames of length {len(evframes)} from {events}')
        else:
            logger('[DATALOADER] No event frames loaded.')

    is_png = len(glob.glob(opj(traj_folders[0], '*.png'))) > 0
    logger(f"[DATALOADER] Image files are {'png' if is_png else 'npy'}")

    desired_vels, ims_all, depths_all, meta_all, evs_all = [], [], [], [], []
    unmatched_ids_ims, kept = [], []
    num_collision_trajs = 0
    have_depths = False
    t0 = time.time()
    n_f = len(traj_folders)
    for ti, folder in enumerate(traj_folders):
        name = os.path.basename(folder)
        if (n_f // 10 > 0 and ti % (n_f // 10) == 0) or n_f < 10:
            logger(f'[DATALOADER] Loading folder {name}, folder # {ti + 1}/{n_f}, time elapsed {time.time() - t0:.2f}s')
        group = h5[folder.split('/')[-1]] if found_h5 else None
        meta = group['data'][()] if found_h5 else _read_meta_csv(opj(folder, 'data.csv'), is_png)

        if np.isnan(meta).any():
            logger(f'[DATALOADER] NaN in {name}, skipping.')
            if do_clean_dataset and not found_h5:
                logger(f'[DATALOADER] Deleting folder {name}')
                shutil.rmtree(folder, ignore_errors=True)
            continue
        if meta[:, -1].sum() > 0:                                            # collision flag column (:236-240)
            num_collision_trajs += 1
            logger(f"[DATALOADER] {meta[:, -1].sum()} collisions in {name}, {num_collision_trajs}th so far, "
                   f"{'skipping!' if not keep_collisions else 'keeping!'}")
            if not keep_collisions:
                continue

        depths = None
        if not found_h5:
            depth_files = sorted(glob.glob(opj(folder, '*_depth.png')))
            if depth_files:
                if ti == 0:
                    logger(f'[DATALOADER] Found images and depths in {os.path.basename(data_dir)}')
                im_files = sorted(glob.glob(opj(folder, '*_im.png')))
            else:
                im_files = sorted(glob.glob(opj(folder, '*.png' if is_png else '*.npy')))
            if not im_files:
                logger(f'[DATALOADER] No images in {name}, skipping.')
                if do_clean_dataset:
                    logger(f'[DATALOADER] Deleting empty folder {name}')
                    shutil.rmtree(folder, ignore_errors=True)
                continue
            if is_png:                                                       # 8-bit gray -> [0, 1]
                ims = np.asarray([_read_gray_png(f) for f in im_files], dtype=np.float32) / 255.0
                if depth_files:
                    depths = np.asarray([_read_gray_png(f) for f in depth_files], dtype=np.float32) / 255.0
            else:                                                            # "old" datasets: fixed normaliser
                ims = np.asarray([np.load(f, allow_pickle=True) for f in im_files]) / 0.09

            # image <-> metadata timestamp matching (:282-350): duplicated metadata timestamps lose their first row,
            # then images / rows whose timestamp has no partner are dropped
            im_ts = [_image_timestamp(f, is_png, logger) for f in im_files]
            _, first_idx, counts = np.unique(meta[:, 1], return_index=True, return_counts=True)
            meta = np.delete(meta, first_idx[counts > 1], axis=0)
            t1 = time.time()
            un_im, un_meta = find_unmatched_indices(im_ts, list(meta[:, 1]))
            if un_im or un_meta:
                logger(f'[DATALOADER] Unmatched timestamps in {name}: (deleting these!)')
                logger(f'[DATALOADER]\tIdxs of images: {un_im}')
                logger(f'[DATALOADER]\tIdxs of metadata: {un_meta}')
                ims = np.delete(ims, un_im, axis=0)
                if depth_files:
                    depths = np.delete(depths, un_im, axis=0)
                if do_clean_dataset:
                    for k in un_im:
                        logger(f'[DATALOADER] Deleting image {im_files[k]}')
                        os.remove(im_files[k])
                        if depth_files:
                            logger(f'[DATALOADER] Deleting depth {depth_files[k]}')
                            os.remove(depth_files[k])
                meta = np.delete(meta, un_meta, axis=0)
                logger(f'[DATALOADER]\tTime to find and delete unmatched indices: {time.time() - t1:.3f}s')
            unmatched_ids_ims.append(un_im)
            meta[:, 1] -= meta[0, 1]                                         # trajectory-relative time
            meta = np.array(meta, dtype=np.float32)
            if events != '':
                evs_all.append(torch.from_numpy(np.asarray(evframes[ti])).float())
        else:
            depths = group['depths']
            ims = group['ims']
            if 'frames' in events:
                evframes = group['evs'][()]
                evs_all.append(torch.from_numpy(np.asarray(evframes)).float())
            else:
                evframes = None

        for k in range(meta.shape[0]):
            desired_vels.append(meta[k, 2] if (is_png or found_h5) else np.max(meta[:, 12]))
        ims_all.append(torch.from_numpy(np.array(ims, dtype=np.float32)))
        meta_all.append(torch.from_numpy(np.asarray(meta)))
        if depths is not None:
            have_depths = True
            depths_all.append(torch.from_numpy(np.array(depths, dtype=np.float32)))
        kept.append(ti)

    traj_folders = [traj_folders[k] for k in kept]
    traj_folders_ids = [traj_folders_ids[k] for k in kept]
    im_h = CROP_HEIGHT if do_transform else ims_all[-1].shape[1]
    im_w = CROP_WIDTH if do_transform else ims_all[-1].shape[2]
    logger(f'[DATALOADER] Images are of size {im_h, im_w} (do_transform={do_transform})')
    logger(f'[DATALOADER] Time to load dataset: {time.time() - t0:.3f}s')

    # ---- optional resize of every frame kind, on the device (:398-416)
    has_evs = evframes is not None
    if resize_input is not None and (ims_all[0].shape[-2:] != torch.Size(resize_input) or
                                     (have_depths and depths_all[0].shape[-2:] != torch.Size(resize_input)) or
                                     (not has_evs or evs_all[0].shape[-2:] != torch.Size(resize_input))):
        logger(f'[DATALOADER] Resizing input images to {resize_input}')
        t1 = time.time()
        _gpu()
        ims_all = _resize_list(ims_all, resize_input)
        depths_all = _resize_list(depths_all, resize_input)
        if has_evs:
            evs_all = _resize_list(evs_all, resize_input)
        im_h, im_w = resize_input
        logger(f'[DATALOADER] Time to resize input images: {time.time() - t1:.3f}s')
    else:
        logger(f'[DATALOADER] No resizing of input images to {resize_input} needed!')

    logger('[DATALOADER] Concatenating data')
    traj_lengths = np.array([t.shape[0] for t in ims_all])
    ims_full = torch.cat(ims_all).reshape(-1, im_h, im_w)
    depths_full = torch.cat(depths_all).reshape(-1, im_h, im_w) if have_depths else None
    meta_full = torch.cat(meta_all).reshape(-1, meta_all[-1].shape[-1])
    desired_vels = torch.Tensor(desired_vels)

    # ---- split (relies on the shuffle above to randomise the selection) (:440-463)
    n_traj = len(traj_lengths)
    if split_method == 'train-val':
        k = int((1. - val_split) * n_traj)
        tr, va = (0, k), (k, n_traj)
    elif split_method == 'val-train':
        k = int(val_split * n_traj)
        va, tr = (0, k), (k, n_traj)
    else:
        raise ValueError(f'split_method={split_method} not implemented!')
    starts = np.concatenate(([0], np.cumsum(traj_lengths))).astype(np.int64)
    tr_i, va_i = (int(starts[tr[0]]), int(starts[tr[1]])), (int(starts[va[0]]), int(starts[va[1]]))

    if depths_full is not None and rescale_depth > 0.0:
        logger(f'[DATALOADER] Rescaling depth by {rescale_depth}\tNOTE max/min of dataset depth is {depths_full.max()}/{depths_full.min()}.')
        _gpu()
        depths_full = torch.clamp(depths_full.to("cuda") / rescale_depth, 0, 1.0).cpu()

    evs_full = None
    if events != '':
        if has_evs:
            mx, mn = max(float(e.max()) for e in evs_all), min(float(e.min()) for e in evs_all)
            logger(f'[DATALOADER] Rescaling evs = {rescale_evs}\tNOTE max/min of dataset evs is {mx}/{mn}.')
            # (a cutoff <= 0 masks nothing -- |x| < c is never true -- so the dataset-only Learner default, 0.0, needs no device)
            if rescale_evs > 0.0 or rescale_evs == -1.0 or (evs_min_cutoff is not None and evs_min_cutoff > 0.0):
                evs_all = _condition_events(evs_all, rescale_evs, evs_min_cutoff, logger)
        evs_full = np.empty(len(evs_all), dtype=object)     # trajectories differ in length: object array of tensors
        for k, e in enumerate(evs_all):
            evs_full[k] = e

    def part(sl_i, sl_t):
        tup = (meta_full[sl_i[0]:sl_i[1]],
               (ims_full[sl_i[0]:sl_i[1]], depths_full[sl_i[0]:sl_i[1]] if depths_full is not None else None),
               traj_lengths[sl_t[0]:sl_t[1]], desired_vels[sl_i[0]:sl_i[1]],
               evs_full[sl_t[0]:sl_t[1]] if has_evs else None,
               traj_folders[sl_t[0]:sl_t[1]], traj_folders_ids[sl_t[0]:sl_t[1]])
        return tup + (unmatched_ids_ims[sl_t[0]:sl_t[1]],) if return_unmatched else tup

    if found_h5:
        h5.close()
    return part(tr_i, tr), part(va_i, va), is_png or found_h5


def preload(items, device='cpu'):
    """dataloading.py:570-582: every item onto `device`; lists / object arrays of per-trajectory arrays become lists of
    float32 tensors, None stays None."""
    out = []
    for item in items:
        if item is None:
            out.append(None)
        elif isinstance(item, list) or (isinstance(item, np.ndarray) and item.dtype == object):
            out.append([(torch.from_numpy(x) if isinstance(x, np.ndarray) else x).to(device).float() for x in item])
        else:
            out.append((torch.from_numpy(np.array(item, dtype=item.dtype)) if isinstance(item, np.ndarray) else item).to(device))
    return out
