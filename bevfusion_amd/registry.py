"""Minimal stand-in for the mmcv/mmdet registries the reference registers its ops in
(mmcv.cnn.CONV_LAYERS, mmcv.cnn NORM_LAYERS, mmdet BACKBONES, mmdet3d VTRANSFORMS — models/builder.py:5-41).

mmcv / mmdet are not vendored in the reference tree and are not installed here.  When they ARE importable the
classes below are ALSO registered in the real registries, so reference configs (`type: SparseEncoder`,
`type: DepthLSSTransform`, `conv_cfg=dict(type="SubMConv3d")`) resolve to the MI355X implementations.
"""
import inspect

from torch import nn


class Registry:
    def __init__(self, name):
        self.name = name
        self._modules = {}

    def register_module(self, name=None, force=False, module=None):
        def _register(cls):
            key = name or cls.__name__
            if key in self._modules and not force:
                raise KeyError(f"{key} is already registered in {self.name}")
            self._modules[key] = cls
            return cls

        if module is not None:
            return _register(module)
        return _register

    def get(self, key):
        return self._modules.get(key)

    def __contains__(self, key):
        return key in self._modules

    def build(self, cfg, **default_args):
        return build_from_cfg(cfg, self, default_args)


def build_from_cfg(cfg, registry, default_args=None):
    """mmcv.utils.build_from_cfg semantics: cfg['type'] names a registered class (or is the class)."""
    if not isinstance(cfg, dict) or "type" not in cfg:
        raise TypeError(f"cfg must be a dict with a 'type' key, got {cfg!r}")
    args = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    typ = args.pop("type")
    cls = registry.get(typ) if isinstance(typ, str) else typ
    if cls is None:
        raise KeyError(f"{typ} is not in the {registry.name} registry")
    if not (inspect.isclass(cls) or callable(cls)):
        raise TypeError(f"type must be a str or class, got {type(cls)}")
    return cls(**args)


CONV_LAYERS = Registry("conv layer")
NORM_LAYERS = Registry("norm layer")
BACKBONES = Registry("backbone")
VTRANSFORMS = Registry("vtransform")

for _name, _cls in (("BN", nn.BatchNorm2d), ("BN1d", nn.BatchNorm1d), ("BN2d", nn.BatchNorm2d),
                    ("BN3d", nn.BatchNorm3d), ("SyncBN", nn.SyncBatchNorm), ("GN", nn.GroupNorm), ("LN", nn.LayerNorm)):
    NORM_LAYERS.register_module(name=_name, module=_cls)
for _name, _cls in (("Conv1d", nn.Conv1d), ("Conv2d", nn.Conv2d), ("Conv3d", nn.Conv3d), ("Conv", nn.Conv2d)):
    CONV_LAYERS.register_module(name=_name, module=_cls)

_ABBR = {"BN": "bn", "BN1d": "bn", "BN2d": "bn", "BN3d": "bn", "SyncBN": "bn", "GN": "gn", "LN": "ln"}


def build_conv_layer(cfg, *args, **kwargs):
    """mmcv.cnn.build_conv_layer: cfg None -> Conv2d; extra cfg keys become constructor kwargs."""
    cfg = dict(type="Conv2d") if cfg is None else dict(cfg)
    typ = cfg.pop("type")
    cls = CONV_LAYERS.get(typ)
    if cls is None:
        raise KeyError(f"Unrecognized conv type {typ}")
    return cls(*args, **kwargs, **cfg)


def build_norm_layer(cfg, num_features, postfix=""):
    """mmcv.cnn.build_norm_layer -> (name, layer); requires_grad / eps / momentum from cfg."""
    cfg = dict(cfg)
    typ = cfg.pop("type")
    cls = NORM_LAYERS.get(typ)
    if cls is None:
        raise KeyError(f"Unrecognized norm type {typ}")
    requires_grad = cfg.pop("requires_grad", True)
    cfg.setdefault("eps", 1e-5)
    if typ == "GN":
        layer = cls(num_channels=num_features, **cfg)
    else:
        layer = cls(num_features, **cfg)
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return _ABBR.get(typ, "norm") + str(postfix), layer


def register_everywhere(registry_name, cls, name=None):
    """Register in our registry and, if mmcv/mmdet/mmdet3d are importable, in theirs too."""
    ours = {"conv": CONV_LAYERS, "backbone": BACKBONES, "vtransform": VTRANSFORMS}[registry_name]
    ours.register_module(name=name, module=cls, force=True)
    try:  # pragma: no cover - mmcv is not installed in this image
        if registry_name == "conv":
            from mmcv.cnn import CONV_LAYERS as REAL
        elif registry_name == "backbone":
            from mmdet.models import BACKBONES as REAL
        else:
            from mmdet3d.models.builder import VTRANSFORMS as REAL
        REAL.register_module(name=name, module=cls, force=True)
    except Exception:
        pass
    return cls
