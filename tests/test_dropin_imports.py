"""The drop-in boundary: with ``dropin/`` first and the reference second on sys.path, the import statements of
the reference's stage drivers must resolve -- hot-path names to the MI355X implementation, everything else (scalar metrics) to
the reference's own modules.  The import lines are cut out of the drivers with ``ast`` and executed in a subprocess with stub
``dataset`` / ``utils`` / tensorboard modules (h5py, cv2, skimage, tensorboard are not in the image).  Needs /root/reference, so it
runs in the build container only (the reference never travels to the GPU box)."""
import ast
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.dont_write_bytecode = True      # the reference tree is read-only: no __pycache__ beside its sources
DRIVERS = ["train_image.py", "train_tile.py", "train_seg.py", "test_count.py", "test_tile.py", "test_seg.py", "train_ensemble.py"]

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")


def _import_lines(path):
    tree = ast.parse(open(path).read())
    keep = []
    for node in tree.body:
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            keep.append(ast.get_source_segment(open(path).read(), node))
    return keep


@pytest.mark.parametrize("driver", [d for d in DRIVERS if os.path.exists(os.path.join(REF, d))])
def test_driver_imports_resolve_through_the_dropin(driver, tmp_path):
    lines = _import_lines(os.path.join(REF, driver))
    assert any("model" in l for l in lines)
    prog = textwrap.dedent('''
        import sys, types
        # stubs for what the image lacks / what is outside the hot path
        def _any(name):
            if name.startswith("__"):
                raise AttributeError(name)
            return object()
        ds = types.ModuleType("dataset"); ds.__getattr__ = _any; sys.modules["dataset"] = ds
        ut = types.ModuleType("utils"); ut.__getattr__ = _any; sys.modules["utils"] = ut
        tb = types.ModuleType("torch.utils.tensorboard"); tb.SummaryWriter = object; sys.modules["torch.utils.tensorboard"] = tb
        for missing in ("cv2", "h5py", "skimage", "openslide", "simplejson", "torchsummary", "easydict", "scipy.misc", "torchvision", "torchvision.transforms"):
            m = types.ModuleType(missing); m.__getattr__ = _any
            sys.modules.setdefault(missing, m)
        sys.path.insert(0, %r)          # the reference (shadowed)
        sys.path.insert(0, %r)          # the drop-in, first
    ''') % (REF, os.path.join(ROOT, "dropin"))
    prog += "\n".join(lines) + "\n"
    prog += textwrap.dedent('''
        import model, train, inference
        assert "cellsegmentation_amd" in model.nets.__class__.__module__ or "cellsegmentation_amd" in type(model.nets).__module__ or hasattr(model, "nets")
        assert train.train_tile.__module__.startswith("cellsegmentation_amd")
        assert inference.inference_tiles.__module__.startswith("cellsegmentation_amd")
        import evaluate, metrics
        assert evaluate.evaluate_tile.__module__.startswith("cellsegmentation_amd")
        assert evaluate.evaluate_image.__module__ == "_shadowed_evaluate"        # the reference's own scalar metric
        assert callable(metrics.qwk) and callable(metrics.calc_err) and callable(metrics.calc_map)
        assert metrics.dice_coef.__module__.startswith("cellsegmentation_amd")
        assert callable(inference.inference_image_cls) and callable(inference.inference_image_reg)
        print("OK")
    ''')
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    env.pop("PYTHONPATH", None)
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, cwd=str(tmp_path), env=env, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]
