"""Golden vectors for the steps either side of the top-k, produced by the REFERENCE's own code:

* evaluate.evaluate_tile        -- imported (evaluate.py needs only metrics/ and train/, both importable here);
* LystoDataset.make_train_data  -- dataset/dataset.py cannot be imported (h5py / skimage / openslide are not installed), so the
  method's source text is cut out of the file with `ast` and executed as a plain function on a stand-in `self` holding
  tileIDX / tiles_grid / labels -- the reference's statements run unmodified, only numpy is needed;
* `rank` (nested inside train_seg.py's main) and `generate_masks` (utils/image_processing.py imports cv2 / skimage): executed the
  same way, from their source text.

Run here (needs /root/reference); writes tests/golden/stage_vectors.npz (inputs + expected outputs, data only).
The oracle restatement is asserted equal to the reference on every case before anything is written."""
import ast
import os
import sys
import types

import numpy as np

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cellseg_oracle as orc  # noqa: E402


def _function_source(path, name):
    src = open(path).read()
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.FunctionDef) and node.name == name:
            seg = ast.get_source_segment(src, node)
            lines = seg.split("\n")
            indent = len(lines[0]) - len(lines[0].lstrip())          # get_source_segment keeps later lines' indentation
            pad = node.col_offset
            return "\n".join([lines[0]] + [l[pad:] if l[:pad].strip() == "" else l for l in lines[1:]])
    raise KeyError(name)


def _exec_function(path, name, glob):
    ns = dict(glob)
    exec(compile(_function_source(path, name), f"{path}:{name}", "exec"), ns)
    return ns[name]


class _NumpyOfTheReference:
    """requirements.txt pins numpy 1.x, where np.array([(int, [x, y], int), ...]) quietly builds an (n, 3) OBJECT array; numpy
    2.x raises on the ragged rows instead.  Everything else is numpy itself."""

    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def array(x, *a, **k):
        try:
            return np.array(x, *a, **k)
        except ValueError:
            out = np.empty((len(x), len(x[0])), dtype=object)
            for i, row in enumerate(x):
                for j, v in enumerate(row):
                    out[i, j] = v
            return out


ref_make_train_data = _exec_function(os.path.join(REF, "dataset/dataset.py"), "make_train_data", {"np": _NumpyOfTheReference()})
ref_rank = _exec_function(os.path.join(REF, "train_seg.py"), "rank", {"np": np})
_tqdm = lambda it, **kw: it       # noqa: E731
ref_generate_masks = _exec_function(os.path.join(REF, "utils/image_processing.py"), "generate_masks",
                                    {"np": np, "os": types.SimpleNamespace(path=types.SimpleNamespace(exists=lambda p: True, join=os.path.join),
                                                                           makedirs=lambda p: None), "tqdm": _tqdm})
sys.path.insert(0, REF)
_stub = types.ModuleType("dataset")
_stub.categorize = lambda x: None
_stub.de_categorize = lambda x: None
sys.modules.setdefault("dataset", _stub)
import evaluate as ref_evaluate  # noqa: E402


def case(seed, n_img, lo, hi, tile=8, size=(40, 48)):
    rng = np.random.RandomState(seed)
    runs = rng.randint(lo, hi + 1, n_img)
    tile_idx = np.repeat(np.arange(n_img), runs)
    T = len(tile_idx)
    grid = np.stack([rng.randint(0, size[0] - tile + 1, T), rng.randint(0, size[1] - tile + 1, T)], 1)   # get_tiles never leaves the image
    labels = rng.choice([0, 0, 1, 2, 3], n_img)
    probs = np.round(rng.rand(T), 2).astype(np.float32)                                            # ties on purpose
    return tile_idx, grid, labels, probs, tile, size


out = {}
for ci, (seed, n_img, lo, hi) in enumerate([(1, 12, 5, 40), (2, 30, 8, 16), (3, 5, 20, 90)]):
    tile_idx, grid, labels, probs, tile, size = case(seed, n_img, lo, hi)
    tpp, thr = 2, 0.5
    tag = f"case{ci}"
    out[f"{tag}/tile_idx"], out[f"{tag}/grid"], out[f"{tag}/labels"], out[f"{tag}/probs"] = tile_idx, grid, labels, probs
    out[f"{tag}/meta"] = np.asarray([tile, size[0], size[1], tpp], dtype=np.int64)
    out[f"{tag}/thr"] = np.asarray([thr], dtype=np.float32)
    # evaluate_tile
    valset = types.SimpleNamespace(tileIDX=tile_idx.tolist(), labels=labels.tolist())
    e_ref = ref_evaluate.evaluate_tile(valset, probs, tpp, thr)
    e_orc = orc.evaluate_tile(tile_idx, labels, probs, tpp, thr)
    assert np.allclose(e_ref, e_orc, rtol=0, atol=0, equal_nan=True), (e_ref, e_orc)
    out[f"{tag}/evaluate_tile"] = np.asarray(e_ref, dtype=np.float64)
    # rank + generate_masks
    ds = types.SimpleNamespace(tileIDX=tile_idx.tolist(), tiles_grid=grid.tolist(), images=[None] * n_img, image_size=size, tile_size=tile)
    t_ref, p_ref, g_ref = ref_rank(ds, probs, thr)
    t_orc, p_orc, g_orc = orc.rank_tiles(tile_idx, grid, probs, thr)
    assert np.array_equal(t_ref, t_orc) and np.array_equal(p_ref, p_orc) and np.array_equal(g_ref, g_orc)
    out[f"{tag}/rank_tiles"], out[f"{tag}/rank_probs"], out[f"{tag}/rank_groups"] = t_ref, p_ref, g_ref
    m_ref = ref_generate_masks(ds, t_ref, g_ref, preprocess=False, save_masks=False)
    m_orc = orc.generate_masks(n_img, size, tile, t_orc, g_orc)
    assert np.array_equal(m_ref, m_orc)
    out[f"{tag}/masks_packed"] = np.packbits(m_ref.astype(np.uint8).ravel())
    # make_train_data on the oracle's top-k selection, several ratios, numpy's legacy RNG as the reference uses it
    idxs = orc.sample_indices(probs, tile_idx, labels, tpp, 3)
    out[f"{tag}/idxs"] = np.asarray(idxs, dtype=np.int64)
    for ri, ratio in enumerate([None, 0.5, 1.0, 2.0, 8.0]):
        self_ = types.SimpleNamespace(tileIDX=tile_idx.tolist(), tiles_grid=grid.tolist(), labels=labels.tolist())
        np.random.seed(100 + ri)
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            pos_ref, neg_ref = ref_make_train_data(self_, idxs, ratio)
        td = self_.train_data                                            # object rows (tileIDX, [x, y], label)
        rows_ref = np.asarray([[int(r[0]), int(r[1][0]), int(r[1][1]), int(r[2])] for r in td], dtype=np.int64).reshape(-1, 4)
        np.random.seed(100 + ri)
        perm = np.arange(len(idxs))
        np.random.shuffle(perm)                                          # same draws as shuffling the rows themselves
        rows_orc, pos_orc, neg_orc = orc.make_train_data(tile_idx, grid, labels, idxs, ratio, perm)
        assert (pos_ref, neg_ref) == (pos_orc, neg_orc) and np.array_equal(rows_ref, rows_orc), (tag, ratio)
        out[f"{tag}/mtd{ri}/perm"], out[f"{tag}/mtd{ri}/rows"] = perm, rows_ref
        out[f"{tag}/mtd{ri}/posneg"] = np.asarray([pos_ref, neg_ref, -1 if ratio is None else int(ratio * 1000)], dtype=np.int64)

np.savez_compressed(os.path.join(ROOT, "tests", "golden", "stage_vectors.npz"), **out)
print("stage_vectors.npz:", len(out), "arrays,", os.path.getsize(os.path.join(ROOT, "tests", "golden", "stage_vectors.npz")), "bytes")
