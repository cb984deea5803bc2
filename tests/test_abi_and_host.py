"""CPU: the C-ABI library loads and exports every symbol include/cellseg_hip.h declares; host-side
logic of the model mirror (setmode groups, error behaviour, state_dict names, selection plan)."""
import os
import re

import numpy as np
import pytest
import torch

from cellsegmentation_amd import _lib, synth
from cellsegmentation_amd import inference as I
from cellsegmentation_amd.model import nets, resnet as R
from oracle import cellseg_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"), allow_pickle=False)


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "cellseg_hip.h")).read()
    # entry points of the A/B flavour (`make AB=1`) sit in #ifdef CS_AB_SWITCHES blocks: not part of the production ABI
    ab_blocks = re.findall(r"#ifdef CS_AB_SWITCHES(.*?)#endif", header, flags=re.S)
    ab_only = sorted(set(re.findall(r"\b(cs_[a-z0-9_]+)\s*\(", "".join(ab_blocks))))
    production = re.sub(r"#ifdef CS_AB_SWITCHES.*?#endif", "", header, flags=re.S)
    declared = sorted(set(re.findall(r"\b(cs_[a-z0-9_]+)\s*\(", production)))
    assert len(declared) >= 30
    assert _lib.FLAVOUR == "", "the CPU suite checks the production library"
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/cellseg_hip.h but not exported"
    assert sorted(_lib.exported_symbols()) == declared, "ctypes signature table out of sync with the header"
    assert sorted(_lib._AB_SIGNATURES) == ab_only
    for name in ab_only:
        assert not hasattr(lib, name), f"{name} is an A/B switch: it must not be exported by the production library"
    assert lib.cs_abi_version() >= 5


def test_production_library_reads_no_environment():
    """the CELLSEG_* launch-rule knobs exist in the A/B flavour only -- the production library does not even
    import getenv."""
    import subprocess
    r = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    undefined = {line.split()[-1].split("@")[0] for line in r.stdout.splitlines() if line.strip()}
    assert "hipLaunchKernel" in undefined or any(s.startswith("hip") for s in undefined)
    assert "getenv" not in undefined and "secure_getenv" not in undefined


def test_argument_checks_without_gpu():
    """Entry points validate before launching: a bad geometry is refused with a message (no GPU needed)."""
    lib = _lib.load()
    g = _lib.CsConvGeom(1, 8, 8, 6, 8, 3, 3, 1, 1, 8, 8)      # C=6 is not a chunk multiple
    rc = lib.cs_conv2d_fwd(g, _lib.CS_BF16, 1, 1, None, None, None, 0, 1, None, None, None)
    assert rc == -1 and b"chunk" in lib.cs_last_error()
    assert lib.cs_igemm_tile(64 * 75 * 75, 64) == 128064
    assert lib.cs_segmented_topk_workspace(1000) >= 8000


def test_setmode_groups_match_reference():
    m = R.MILresnet18()
    for mode in ("tile", "image", "segment"):
        m.setmode(mode)
        got = sorted(k for k, p in m.named_parameters() if p.requires_grad)
        assert got == GOLD[f"setmode/{mode}"].tolist()
    with pytest.raises(Exception) as e:
        m.setmode("bogus")
    assert str(e.value) == str(GOLD["setmode/invalid_msg"])


def test_forward_errors():
    m = R.MILresnet18()
    with pytest.raises(RuntimeError, match="GPU only"):
        m.setmode("tile")
        m(torch.zeros(1, 3, 32, 32))           # CPU tensor: the product has no CPU fallback
    assert m.encoder_name == "resnet18"
    assert all("layer1.0.conv1.weight".startswith(p) is False for p in m.tile_module_prefix)
    assert any("layer1.0.conv1.weight".startswith(p) for p in m.encoder_prefix)
    assert any("upconv3.0.weight".startswith(p) for p in m.seg_module_prefix)


def test_efficientnet_structure():
    from cellsegmentation_amd.model import efficientnet as EN
    b0 = nets["efficientnet_b0"]
    # torchvision efficientnet_b0: 5,288,548 parameters of which 1,281,000 in the 1000-way classifier
    assert sum(p.numel() for p in b0.features.parameters()) == 4007548
    assert b0.feature_dim == 1280 and nets["efficientnet_b2"].feature_dim == 1408 and nets["efficientnet_b3"].feature_dim == 1536
    assert EN.mbconv_table(1.2, 1.4)[0][3:] == (40, 24, 2)         # B3 stem 40, first stage 24 x2
    assert [c[4] for c in EN.mbconv_table(1.1, 1.2)] == [16, 24, 48, 88, 120, 208, 352]
    keys = b0.state_dict().keys()
    assert "features.1.0.block.1.fc1.weight" in keys and "features.8.1.running_var" in keys and "fc_image_reg.2.bias" in keys
    b0.setmode("image")
    assert all(p.requires_grad for p in b0.features.parameters()) and not b0.fc_tile[1].weight.requires_grad
    with pytest.raises(Exception, match="Invalid mode"):
        b0.setmode("x")


@pytest.mark.parametrize("arch", ["resnet18", "resnet34", "resnet50", "resnext50_32x4d", "resnext101_32x8d"])
def test_state_dict_names_and_shapes(arch):
    m = nets[arch]
    want = {k: tuple(v.shape) for k, v in orc.empty_state_dict(arch).items()}
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert got == want
    assert nets[arch] is m                      # process-wide singleton like the reference's dict entries


def test_nets_keys():
    # every key of the reference's dict (model/__init__.py:5-13) + efficientnet_b3 (BASELINE config 4)
    assert {"resnet18", "resnet34", "resnet50", "efficientnet_b0", "efficientnet_b2", "resnext50_32x4d", "resnext101_32x8d",
            "efficientnet_b3"} == set(nets.keys())
    with pytest.raises(KeyError):
        nets["vgg16"]


def test_selection_plan_host_half():
    g, k, off = I.selection_plan([1] * 5 + [2] * 5 + [3] * 5, {1: 2, 2: 0, 3: 1}, 1, 3)
    assert k.tolist() == [2] * 5 + [3] * 5 + [1] * 5 and off.tolist() == [0, 5, 10, 15]
    with pytest.raises(ValueError):
        I.selection_plan([2, 1], {1: 0, 2: 0}, 1, 3)
    with pytest.raises(ValueError):
        I.selection_plan([], {}, 1, 3)


def test_loss_module_api():
    from cellsegmentation_amd.train import DiceLoss, MSELoss, WeightedMSELoss
    for cls in (DiceLoss, MSELoss, WeightedMSELoss):
        with pytest.raises(AssertionError):
            cls(reduction="none")
    with pytest.raises(RuntimeError):
        MSELoss()(torch.zeros(3), torch.zeros(3))          # CPU tensors: fails loudly


def test_synth_is_deterministic_and_name_keyed():
    a = synth.uniform("layer1.0.conv1.weight", 7, -1, 1)
    b = synth.uniform("layer1.0.conv1.weight", 7, -1, 1)
    c = synth.uniform("layer1.0.conv2.weight", 7, -1, 1)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    t1, t2 = synth.ihc_tiles(2, 64, 5), synth.ihc_tiles(2, 64, 5)
    assert t1.dtype == np.uint8 and np.array_equal(t1, t2)
    x = synth.normalise(t1)
    assert tuple(x.shape) == (2, 3, 64, 64) and -2.2 < float(x.min()) and float(x.max()) < 2.7


def test_get_tiles_host_grid_matches_reference_known_answer():
    from cellsegmentation_amd import tiles
    g = tiles.get_tiles((299, 299), 20, 32)                  # 15 x 15 = 225 coords, last origin 267
    assert len(g) == 225 and g[0] == (0, 0) and g[14] == (0, 267) and g[-1] == (267, 267)
    assert g == orc.get_tiles_coords(299, 299, 20, 32)
    assert len(tiles.get_tiles((299, 299), 5, 16)) == 3364   # train_seg.py:225-232: 58 x 58
    assert tiles.get_tiles((64, 64), 32, 32) == [(0, 0), (0, 32), (32, 0), (32, 32)]     # stride lands on the border: no extra tile
    ti, rc = tiles.tile_index(2, (64, 64), 32, 32)
    assert ti.tolist() == [0] * 4 + [1] * 4 and rc.shape == (8, 2)


def test_the_oracle_stays_test_infrastructure():
    """oracle/ may be imported by tests/, by smoke() (cellsegmentation_amd/smoke.py, called from __graft_entry__.smoke) and by
    bench.py's cpu_baseline leg only: the product path must never route through it."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    allowed = {os.path.join("cellsegmentation_amd", "smoke.py"), "bench.py"}
    offenders = []
    for base, dirs, files in os.walk(root):
        rel = os.path.relpath(base, root)
        # (_ab_*: git worktrees of older rounds that tools/ab_profile.sh compares against on one box; never committed)
        dirs[:] = [d for d in dirs if d not in (".git", "gpurun_out", "__pycache__", "oracle", "tests") and not d.startswith("_ab_")]
        for f in files:
            if not f.endswith(".py"):
                continue
            path = os.path.normpath(os.path.join(rel, f))
            text = open(os.path.join(base, f)).read()
            if re.search(r"^\s*(from\s+oracle\b|import\s+oracle\b)", text, re.M) and path not in allowed:
                offenders.append(path)
    assert not offenders, offenders
    # ... and inside bench.py only the cpu_baseline functions do
    src = open(os.path.join(root, "bench.py")).read()
    for m in re.finditer(r"^\s*from oracle import", src, re.M):
        head = src[:m.start()]
        fn = re.findall(r"^def (\w+)\(", head, re.M)[-1]
        assert fn.startswith("cpu_baseline"), fn


def test_adam_keeps_its_capturable_flag_across_load_state_dict():
    """torch's Optimizer.load_state_dict takes every hyper-parameter -- `capturable` included -- from the LOADED groups and places `step`
    accordingly.  Where the step counts live is a property of the optimizer object a driver constructed (train_tile.py:282), not of the
    checkpoint it resumes from (train_tile.py:161-176): cellsegmentation_amd.optim.Adam keeps its own flag and moves the counts."""
    import copy
    import torch
    from cellsegmentation_amd.optim import Adam
    p = [torch.zeros(3, requires_grad=True)]
    q = [torch.zeros(3, requires_grad=True)]
    ref = torch.optim.Adam(q, lr=1e-3)
    q[0].grad = torch.ones(3)
    ref.step()
    ref.step()
    a = Adam(p, lr=5e-4, capturable=True)
    a.load_state_dict(copy.deepcopy(ref.state_dict()))
    assert a.param_groups[0]["capturable"] is True
    st = a.state[p[0]]["step"]
    assert st.dtype == torch.float32 and st.dim() == 0 and float(st) == 2.0
    b = Adam(p, lr=5e-4)                                   # host step counts
    sd = copy.deepcopy(a.state_dict())
    b.load_state_dict(sd)
    assert b.param_groups[0]["capturable"] is False and b.state[p[0]]["step"].device.type == "cpu" and float(b.state[p[0]]["step"]) == 2.0
    assert a.param_groups[0]["lr"] == 1e-3                 # (the other hyper-parameters do come from the checkpoint, as in torch)
    with __import__("pytest").raises(ValueError):
        Adam(p, amsgrad=True)
