"""The host steps either side of the adaptive top-k, with their data kept on the GPU.

* ``make_train_data``  -- dataset/dataset.py:166-201 (label assignment, shuffle, pos/neg re-balancing of the selected tiles)
* ``evaluate_tile``    -- evaluate.py:8-27 (+ metrics/metrics.py:7-16 ``calc_err``)
* ``rank``             -- the tile ranking inside train_seg.py:234-249
* ``generate_masks``   -- utils/image_processing.py:79-98 (square painting only; the HSV / small-region clean-up and the PNG
                          writing stay on the CPU as in the reference)

Index bookkeeping that is O(#images) stays in numpy on the host (as in ``inference.selection_plan``); everything that is
O(#tiles) or O(#pixels) runs in HIP kernels (`csrc/topk.hip`).  The shuffle of ``make_train_data`` is an explicit permutation:
with ``perm = np.random.permutation``-style draws from the caller's generator the result equals the reference's for the same
draws (parity is defined up to that permutation).
"""
import numpy as np
import torch

from . import kernels as K


def _runs(tile_idx):
    groups = np.asarray(tile_idx)
    if groups.ndim != 1 or len(groups) == 0:
        raise ValueError("tileIDX must be a non-empty 1-D sequence")
    if np.any(groups[1:] < groups[:-1]):
        raise ValueError("tileIDX must be non-decreasing (tiles of one image are contiguous, dataset/dataset.py:120-140)")
    starts = np.flatnonzero(np.r_[True, groups[1:] != groups[:-1]])
    offsets = np.r_[starts, len(groups)].astype(np.int64)
    return groups.astype(np.int32), offsets


def _dev_probs(probs, device):
    p = probs if torch.is_tensor(probs) else torch.from_numpy(np.ascontiguousarray(probs, dtype=np.float32))
    return p.to(device=device, dtype=torch.float32).contiguous()


def rank(tile_idx, tiles_grid, probs, threshold, device=None):
    """train_seg.py:234-249: tiles sorted by (image, probability), those above `threshold` kept.
    Returns (tiles[index], probs[index], groups[index]) as numpy arrays, like the reference's nested function."""
    if device is None:
        device = probs.device if torch.is_tensor(probs) else torch.device("cuda")
    groups, offsets = _runs(tile_idx)
    p = _dev_probs(probs, device)
    order = K.segmented_order(p, torch.from_numpy(offsets).to(device), int(np.diff(offsets).max()))
    sel, cnt = K.threshold_select(p, order, threshold)
    sel = sel[: int(cnt.item())].cpu().numpy()
    tiles = np.asarray(tiles_grid)
    return tiles[sel], p.cpu().numpy()[sel], np.asarray(tile_idx)[sel]


def generate_masks(n_images, image_size, tile_size, tile_idx, tiles_grid, selected, device=None):
    """utils/image_processing.py:90-98: uint8 [n_images, H, W] DEVICE tensor with a tile_size^2 square of ones for every selected
    tile (`selected` = indices into tile_idx / tiles_grid, e.g. the positions `rank` kept)."""
    if device is None:
        device = torch.device("cuda")
    H, W = image_size
    groups = torch.from_numpy(np.asarray(tile_idx, dtype=np.int32)).to(device)
    xy = torch.from_numpy(np.ascontiguousarray(np.asarray(tiles_grid, dtype=np.int32).reshape(-1, 2))).to(device)
    sel = selected if torch.is_tensor(selected) else torch.from_numpy(np.asarray(selected, dtype=np.int64))
    sel = sel.to(device=device, dtype=torch.int64).contiguous()
    return K.paint_tile_masks(sel, sel.numel(), groups, xy, tile_size, n_images, H, W)


def evaluate_tile(valset, probs, tiles_per_pos, threshold, device=None):
    """evaluate.py:8-27: (err, fpr, fnr) of thresholded tile predictions against the count-derived labels."""
    if device is None:
        device = probs.device if torch.is_tensor(probs) else torch.device("cuda")
    groups, offsets = _runs(valset.tileIDX)
    T = len(groups)
    counts = np.asarray([valset.labels[g] for g in groups[offsets[:-1]]], dtype=np.int64) * int(tiles_per_pos)
    ends = offsets[1:]
    starts = ends - counts
    if np.any(starts < 0):
        # the reference's slice assignment labels[i - n : i] = [1] * n fails the same way when n > i
        raise ValueError("could not broadcast input array: a count x tiles_per_pos exceeds the tiles sorted before the end of its image")
    pos_from_run = np.minimum.accumulate(starts[::-1])[::-1]            # an oversized count spills into the previous image(s)
    pos_from = np.zeros(int(groups.max()) + 1, dtype=np.int64)
    pos_from[groups[offsets[:-1]]] = pos_from_run
    p = _dev_probs(probs, device)
    order = K.segmented_order(p, torch.from_numpy(offsets).to(device), int(np.diff(offsets).max()))
    c = K.evaluate_tile_counts(p, order, torch.from_numpy(groups).to(device), torch.from_numpy(pos_from).to(device), threshold).cpu().numpy()
    neq, fp, fn, real1 = (int(v) for v in c)
    with np.errstate(divide="ignore", invalid="ignore"):               # numpy semantics of metrics.calc_err on empty classes
        err = float(neq) / T
        fpr = np.float64(fp) / np.int64(T - real1)
        fnr = np.float64(fn) / np.int64(real1)
    return err, fpr, fnr


def make_train_data(tile_idx, tiles_grid, labels, idxs, pos_neg_ratio=None, perm=None, generator=None, device=None):
    """dataset/dataset.py:166-201.  idxs: the tile indices `sample` selected (list / array / device tensor).
    Returns (train_data, pos, neg): train_data is an int array [(tileIDX, x, y, label)] in the shuffled, pruned order (the
    reference keeps the same rows as a numpy object array of (tileIDX, grid, label)).  perm: the shuffle, as a permutation of
    range(len(idxs)) (position p of the shuffled array holds entry perm[p]); drawn from `generator` (numpy Generator or
    RandomState) when omitted."""
    if device is None:
        device = torch.device("cuda")
    idx_host = idxs.cpu().numpy() if torch.is_tensor(idxs) else np.asarray(idxs, dtype=np.int64)
    n = len(idx_host)
    if n == 0:
        raise ValueError("make_train_data: no tiles selected")
    if perm is None:
        rng = generator if generator is not None else np.random
        perm = np.arange(n)
        rng.shuffle(perm)
    perm = np.asarray(perm, dtype=np.int64)
    if sorted(perm.tolist()) != list(range(n)):
        raise ValueError("perm must be a permutation of range(len(idxs))")
    tile_idx = np.asarray(tile_idx)
    grid = np.asarray(tiles_grid).reshape(-1, 2)
    shuffled = idx_host[perm]
    img = tile_idx[shuffled]
    lab_host = np.asarray([0 if labels[g] == 0 else 1 for g in img], dtype=np.int32)
    pos = int(lab_host.sum())
    neg = n - pos
    keep = np.arange(n)
    if pos_neg_ratio is not None:
        flag = None
        if pos > int(neg * pos_neg_ratio):
            flag, excess, pos = 1, pos - int(neg * pos_neg_ratio), int(neg * pos_neg_ratio)
            print('Note: Positive superpixels are pruned to meet the pos_neg_ratio. ')
        elif neg > int(pos / pos_neg_ratio):
            flag, excess, neg = 0, neg - int(pos / pos_neg_ratio), int(pos / pos_neg_ratio)
            print('Note: Negative superpixels are pruned to meet the pos_neg_ratio. ')
        if flag is not None:
            kept, cnt = K.prune_excess(torch.from_numpy(lab_host).to(device), flag, excess)
            keep = kept[: int(cnt.item())].cpu().numpy()
    rows = shuffled[keep]
    train_data = np.column_stack([tile_idx[rows], grid[rows, 0], grid[rows, 1], lab_host[keep]]).astype(np.int64)
    return train_data, pos, neg
