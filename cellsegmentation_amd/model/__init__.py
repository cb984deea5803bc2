"""Model factory mirroring the reference's ``model/__init__.py:5-13`` (``nets[name]``), but lazy and
offline: entries are built on first access, never download, and stay process-wide singletons like the
reference's (drivers mutate them with setmode(); ensembles deepcopy them)."""
from .resnet import MILResNet, MILresnet18, MILresnet34, MILresnet50, MILresnext50_32x4d, MILresnext101_32x8d
from .efficientnet import MILEfficientNet, MILefficientnetB0, MILefficientnetB2, MILefficientnetB3

_FACTORIES = {
    "resnet18": MILresnet18,
    "resnet34": MILresnet34,
    "resnet50": MILresnet50,
    "resnext50_32x4d": MILresnext50_32x4d,
    "resnext101_32x8d": MILresnext101_32x8d,
    "efficientnet_b0": MILefficientnetB0,
    "efficientnet_b2": MILefficientnetB2,
    "efficientnet_b3": MILefficientnetB3,     # not in the reference's dict (model/__init__.py:9-10); BASELINE config 4
}


class _LazyNets(dict):
    def __missing__(self, key):
        if key not in _FACTORIES:
            raise KeyError(key)
        self[key] = _FACTORIES[key](pretrained=False)
        return self[key]

    def __contains__(self, key):
        return key in _FACTORIES

    def keys(self):
        return _FACTORIES.keys()


nets = _LazyNets()
