"""CPU: checkpoint files of the three stages -- structure, per-stage parameter filters, hand-off and
resume paths; and, where the reference checkout is present, a round trip through the reference's own model class."""
import os
import sys

import pytest
import torch

from cellsegmentation_amd import checkpoint as C
from cellsegmentation_amd.model import resnet as R

REF = "/root/reference"
sys.dont_write_bytecode = True      # the reference tree is read-only: no __pycache__ beside its sources


def _tiny_opt(m):
    params = [p for p in m.parameters()]
    opt = torch.optim.SGD(params, lr=0.1, momentum=0.9)
    sch = torch.optim.lr_scheduler.StepLR(opt, 3)
    return opt, sch


@pytest.mark.parametrize("stage,mode,prefix,has,has_not", [
    ("image", "image", "pt1", ["conv1.weight", "layer4.1.bn2.running_var", "fc_image_cls.4.weight"], ["fc_tile.1.weight", "upconv1.0.weight"]),
    ("tile", "tile", "pt2", ["conv1.weight", "fc_image_reg.7.bias", "fc_tile.1.weight"], ["upconv1.0.weight", "seg_out_conv.weight"]),
    ("seg", "seg", "pt3", ["conv1.weight", "fc_tile.1.bias", "upconv4.0.weight", "seg_out_conv.bias"], []),
])
def test_checkpoint_object_matches_the_drivers(tmp_path, stage, mode, prefix, has, has_not):
    m = R.MILresnet18()
    opt, sch = _tiny_opt(m)
    path = C.save_model(7, m, opt, sch, str(tmp_path), stage=stage)
    assert os.path.basename(path) == f"{prefix}_7epochs.pth"
    cp = torch.load(path, map_location="cpu")
    assert list(cp) == ["mode", "epoch", "state_dict", "encoder", "optimizer", "scheduler"]
    assert cp["mode"] == mode and cp["epoch"] == 7 and cp["encoder"] == "resnet18"
    assert cp["scheduler"]["last_epoch"] == 0 and "param_groups" in cp["optimizer"]
    for k in has:
        assert k in cp["state_dict"], k
    for k in has_not:
        assert k not in cp["state_dict"], k
    assert C.checkpoint_object(stage, 1, m, opt, None)["scheduler"] is None


def test_stage_hand_off_and_resume(tmp_path):
    nets_ = {"resnet18": None}

    class Nets(dict):             # the factory mapping builds a fresh model per lookup, like model.nets
        def __getitem__(self, k):
            return R.MILresnet18()

    src = R.MILresnet18()
    with torch.no_grad():
        for p in src.parameters():
            p.add_(0.5)
    opt, sch = _tiny_opt(src)
    sch.step(); sch.step()
    p1 = C.save_model(3, src, opt, sch, str(tmp_path), stage="image")
    # pt1 -> tile stage: encoder + image heads arrive, the tile head stays at its fresh initialisation
    m, last, last_s, cp = C.load_for_stage("tile", Nets(nets_), p1, "cpu", resume=False)
    assert (last, last_s) == (0, -1)
    assert torch.equal(m.conv1.weight, src.conv1.weight) and torch.equal(m.fc_image_cls[4].weight, src.fc_image_cls[4].weight)
    assert not torch.equal(m.fc_tile[1].weight, src.fc_tile[1].weight)
    # resume of the tile stage: everything the file holds, epoch and scheduler position restored
    p2 = C.save_model(5, src, opt, sch, str(tmp_path), stage="tile")
    m2, last, last_s, cp2 = C.load_for_stage("tile", Nets(nets_), p2, "cpu", resume=True)
    assert (last, last_s) == (5, 2) and torch.equal(m2.fc_tile[1].weight, src.fc_tile[1].weight)
    opt2, sch2 = _tiny_opt(m2)
    C.restore_optimizer(cp2, opt2, sch2)
    assert sch2.last_epoch == 2
    # pt2 -> seg stage keeps the decoder fresh; seg resume restores it
    m3, _, _, _ = C.load_for_stage("seg", Nets(nets_), p2, "cpu", resume=False)
    assert torch.equal(m3.fc_tile[1].weight, src.fc_tile[1].weight) and not torch.equal(m3.upconv1[0].weight, src.upconv1[0].weight)
    p3 = C.save_model(9, src, opt, None, str(tmp_path), stage="seg")
    m4, last, last_s, _ = C.load_for_stage("seg", Nets(nets_), p3, "cpu", resume=True)
    assert (last, last_s) == (9, -1) and torch.equal(m4.upconv1[0].weight, src.upconv1[0].weight)
    with pytest.raises(ValueError):
        C.load_for_stage("image", Nets(nets_), p1, "cpu", resume=False)


def test_torchvision_style_weights_load_by_name():
    m = R.MILresnet18()
    tv = {k: torch.full_like(v, 0.25) for k, v in m.state_dict().items()
          if k.startswith(("conv1", "bn1", "layer")) and v.is_floating_point()}
    tv["fc.weight"], tv["fc.bias"] = torch.zeros(1000, 512), torch.zeros(1000)       # torchvision's classifier: no counterpart
    missing, unexpected = C.load_torchvision_weights(m, tv)
    assert sorted(unexpected) == ["fc.bias", "fc.weight"]
    assert all(not k.startswith(("conv1.", "layer")) or "num_batches_tracked" in k for k in missing)
    assert float(m.layer3[0].conv1.weight.mean()) == 0.25


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "model")), reason="reference checkout not present (GPU box)")
def test_round_trip_through_the_reference_model(tmp_path):
    """A pt2 file written here loads in the reference's own MILresnet50 with the reference's tile-stage filter, and a state dict
    saved from the reference model loads here: identical keys, shapes and values."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_resnet_for_checkpoints", os.path.join(REF, "model", "resnet.py"))
    ref_resnet = importlib.util.module_from_spec(spec)       # by file path, as tests/golden/make_golden.py does: the package's
    spec.loader.exec_module(ref_resnet)                       # __init__ also pulls in the torchvision-dependent EfficientNet
    ours = R.MILresnet50()
    with torch.no_grad():
        for p in ours.parameters():
            p.uniform_(-0.1, 0.1)
    opt, sch = _tiny_opt(ours)
    path = C.save_model(2, ours, opt, sch, str(tmp_path), stage="tile")
    cp = torch.load(path, map_location="cpu")
    ref = ref_resnet.MILresnet50()
    keep = {k: v for k, v in cp["state_dict"].items()
            if k.startswith(ref.encoder_prefix + ref.tile_module_prefix + ref.image_module_prefix)}          # train_tile.py:249-253
    assert set(keep) == set(cp["state_dict"])                      # the reference's filter drops nothing we wrote
    report = ref.load_state_dict(keep, strict=False)
    assert not report.unexpected_keys
    assert all(k.startswith(("upconv", "seg_out_conv")) for k in report.missing_keys)
    for k, v in ref.state_dict().items():
        if k in keep:
            assert torch.equal(v, ours.state_dict()[k]), k
    back = R.MILresnet50()
    back.load_state_dict(ref.state_dict())                          # strict: same key set both ways
    assert list(back.state_dict()) == list(ref.state_dict())
