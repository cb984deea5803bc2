"""inference_tiles / sample / inference_image / inference_seg with the reference's signatures
(inference.py:9-153), on the HIP path.

* ``inference_tiles``: eval forward + on-device softmax prob of class 1, written into ONE device
  buffer and copied to the host once at the end (the reference copies every batch).
* ``sample``: the adaptive top-k.  Host logic only builds the per-tile k and the run offsets from
  ``trainset.tileIDX`` / ``trainset.labels``; sorting + selection run in the segmented top-k HIP
  kernel, bit-exact with ``np.lexsort`` + the reference's wrap-around predicate.
"""
import numpy as np
import torch

from . import kernels as K

try:
    from tqdm import tqdm
except ImportError:  # pragma: no cover
    def tqdm(x, **kw):
        return x


def inference_tiles(loader, model, device, epoch=None, total_epochs=None, mode='train'):
    """Forward inference to obtain instance classification probs -> np.ndarray[float32, len(dataset)]."""
    model.eval()
    probs = torch.zeros((len(loader.dataset),), dtype=torch.float32, device=device)
    with torch.no_grad():
        for i, input in enumerate(tqdm(loader, desc="tile forwarding")):
            if mode == 'train':
                input = input[0]
            output = model(input.to(device))
            n = input.size(0)
            probs[i * loader.batch_size:i * loader.batch_size + n] = K.softmax_prob1(output.contiguous())
    return probs.cpu().numpy()


def selection_plan(tile_idx, labels, tiles_per_pos, topk_neg):
    """Host half of ``sample``: per-tile k and run offsets (inference.py:38-39 semantics).
    labels may be a dict or a sequence indexed by group id."""
    groups = np.asarray(tile_idx)
    if groups.ndim != 1 or len(groups) == 0:
        raise ValueError("tileIDX must be a non-empty 1-D sequence")
    if np.any(groups[1:] < groups[:-1]):
        raise ValueError("tileIDX must be non-decreasing (tiles of one image are contiguous, dataset/dataset.py:120-140)")
    lab = np.asarray([labels[g] for g in groups], dtype=np.int64)
    k = np.where(lab == 0, topk_neg, lab * tiles_per_pos).astype(np.int32)
    starts = np.flatnonzero(np.r_[True, groups[1:] != groups[:-1]])
    offsets = np.r_[starts, len(groups)].astype(np.int64)
    return groups.astype(np.int32), k, offsets


def select_topk(probs, tile_idx, labels, tiles_per_pos, topk_neg, device=None):
    """order[index] of inference.py:34-42 as a python list of ints."""
    groups, k, offsets = selection_plan(tile_idx, labels, tiles_per_pos, topk_neg)
    if device is None:
        device = probs.device if torch.is_tensor(probs) else torch.device("cuda")
    p = probs if torch.is_tensor(probs) else torch.from_numpy(np.ascontiguousarray(probs, dtype=np.float32))
    p = p.to(device=device, dtype=torch.float32).contiguous()
    out, cnt = K.segmented_topk(p, torch.from_numpy(groups).to(device), torch.from_numpy(k).to(device),
                                torch.from_numpy(offsets).to(device), int(np.diff(offsets).max()))
    n = int(cnt.item())
    return out[:n].cpu().tolist()


def sample(trainset, probs, tiles_per_pos, topk_neg, pos_neg_ratio):
    """Select top-k tiles per image to create the instance training set (inference.py:31-43)."""
    selected = select_topk(probs, trainset.tileIDX, trainset.labels, tiles_per_pos, topk_neg)
    p, n = trainset.make_train_data(selected, pos_neg_ratio)
    print("Training data is sampled. (Pos samples: {} | Neg samples: {})".format(p, n))


def inference_image(loader, model, device, epoch=None, total_epochs=None, mode='train', cls_limit=False, return_id=False,
                    categorize=None, de_categorize=None):
    """Image-level class + rounded count (inference.py:46-101).  ``categorize``/``de_categorize`` are the dataset helpers the
    reference imports at module level (inference.py:6, dataset/dataset.py:745-780); only needed with cls_limit, and then taken
    from the caller's ``dataset`` module when not passed."""
    if cls_limit and (categorize is None or de_categorize is None):
        import dataset as _dataset                       # the reference's (or the user's) dataset module, as inference.py:6
        categorize = categorize or _dataset.categorize
        de_categorize = de_categorize or _dataset.de_categorize
    model.eval()
    ids, cats, counts = [], [], []
    with torch.no_grad():
        for i, data in enumerate(tqdm(loader, desc="image forwarding")):
            if mode == 'train':
                data = data[0]
            else:
                batch_ids, data = data
                ids.append(np.asarray(batch_ids))
            output = model(data.to(device))
            cat_labels = K.softmax_argmax(output[0].float().contiguous()).cpu().numpy()      # argmax of the PROBABILITIES (:72-76)
            output_reg = np.round(output[1][:, 0].cpu().numpy()).astype(int)
            if cls_limit:
                for j, x in enumerate(output_reg):
                    if categorize(x) > cat_labels[j]:
                        output_reg[j] = de_categorize(cat_labels[j])[1]
                    elif categorize(x) < cat_labels[j]:
                        output_reg[j] = de_categorize(cat_labels[j])[0]
            cats.append(cat_labels.astype(np.float64))
            counts.append(output_reg.astype(np.float64))
    cats = np.concatenate(cats) if cats else np.array(())
    counts = np.concatenate(counts) if counts else np.array(())
    if return_id:
        return (np.concatenate(ids) if ids else np.array(())), cats, counts
    return cats, counts


def inference_image_cls(loader, model, device, epoch=None, total_epochs=None, mode='train'):
    """Image-level class only (inference.py:104-120): argmax of the 7-way head."""
    model.eval()
    cats = []
    with torch.no_grad():
        for i, data in enumerate(tqdm(loader, desc="image forwarding")):
            if mode == 'train':
                data = data[0]
            output = model(data.to(device))
            cats.append(K.softmax_argmax(output[0].float().contiguous()).cpu().numpy().astype(np.float64))     # :118-119
    return np.concatenate(cats) if cats else np.array(())


def inference_image_reg(loader, model, device, epoch=None, total_epochs=None, mode='train'):
    """Image-level count only (inference.py:123-137): the raw regression output, float32 [len(dataset)]."""
    model.eval()
    nums = []
    with torch.no_grad():
        for i, data in enumerate(tqdm(loader, desc="image forwarding")):
            if mode == 'train':
                data = data[0]
            output = model(data.to(device))
            nums.append(output[1].detach()[:, 0].float().cpu())
    return torch.cat(nums, dim=0).numpy() if nums else torch.tensor(()).numpy()


def inference_seg(loader, model, device, mode='train'):
    """inference.py:140-153"""
    model.eval()
    masks = []
    with torch.no_grad():
        for i, data in enumerate(tqdm(loader, desc="image segmenting")):
            output = model(data.to(device))
            if mode == 'test':
                output = K.softmax_channel_fwd(output.contiguous(), 1)
            masks.append(output.cpu().numpy())
    return np.concatenate(masks)
