"""YAML config loading with the reference's semantics: `torchpack.utils.config.configs.load(path, recursive=True)`
(tools/train.py:28 — torchpack is un-vendored) + `recursive_eval` (mmdet3d/utils/config.py:6-20).

  * every `default.yaml` from the configs root down to the file's directory is merged in order, then the file
    itself (dicts merge recursively, anything else is replaced);
  * any string of the form `${python expr}` is evaluated with the whole config as globals, e.g.
    `${[image_size[0] // 8, image_size[1] // 8]}` (recursively, until no `${}` is left).
"""
import copy
import os

import yaml

__all__ = ["load_config", "recursive_eval", "build_hot_path"]


class AttrDict(dict):
    """dict with attribute access, like torchpack's Config: expressions such as `${augment2d.resize[0]}`
    (configs/nuscenes/default.yaml:121) rely on it."""

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError as e:
            raise AttributeError(key) from e

    def __setattr__(self, key, value):
        self[key] = value


def _attrify(obj):
    if isinstance(obj, dict):
        return AttrDict({k: _attrify(v) for k, v in obj.items()})
    if isinstance(obj, list):
        return [_attrify(v) for v in obj]
    return obj


def _plain(obj):
    if isinstance(obj, dict):
        return {k: _plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_plain(v) for v in obj]
    return obj


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def recursive_eval(obj, globals=None):
    """utils/config.py:6-20."""
    if globals is None:
        globals = copy.deepcopy(obj)
    if isinstance(obj, dict):
        for key in obj:
            obj[key] = recursive_eval(obj[key], globals)
    elif isinstance(obj, list):
        for k, val in enumerate(obj):
            obj[k] = recursive_eval(val, globals)
    elif isinstance(obj, str) and obj.startswith("${") and obj.endswith("}"):
        obj = eval(obj[2:-1], globals)
        obj = recursive_eval(obj, globals)
    return obj


def load_config(path, root=None, evaluate=True):
    """Merge the default.yaml chain + `path` and (optionally) evaluate `${}` expressions."""
    path = os.path.abspath(path)
    if root is None:
        # the chain starts at the outermost ancestor directory named "configs"
        parts = path.split(os.sep)
        idx = max(i for i, p in enumerate(parts) if p == "configs")
        root = os.sep.join(parts[: idx + 1])
    root = os.path.abspath(root)
    rel = os.path.relpath(os.path.dirname(path), root)
    chain = [root]
    if rel != ".":
        cur = root
        for part in rel.split(os.sep):
            cur = os.path.join(cur, part)
            chain.append(cur)
    cfg = {}
    for d in chain:
        f = os.path.join(d, "default.yaml")
        if os.path.exists(f) and os.path.abspath(f) != path:
            with open(f) as fh:
                _merge(cfg, yaml.safe_load(fh) or {})
    with open(path) as fh:
        _merge(cfg, yaml.safe_load(fh) or {})
    if evaluate:
        cfg = _plain(recursive_eval(_attrify(cfg)))
    return cfg


def build_hot_path(cfg):
    """Instantiate the hot-path modules a BEVFusion config names: the camera view transform and the LiDAR
    voxelizer + sparse backbone (fusion_models/bevfusion.py:36-69).  Returns a dict with whatever is present."""
    from . import sparse_encoder, vtransforms  # noqa: F401  (registration side effects)
    from .registry import BACKBONES, VTRANSFORMS
    from .voxel import Voxelization

    out = {}
    enc = cfg.get("model", {}).get("encoders", {}) or {}
    cam = enc.get("camera")
    if cam and "vtransform" in cam and cam["vtransform"]["type"] in VTRANSFORMS:
        out["vtransform"] = VTRANSFORMS.build(cam["vtransform"])
    lid = enc.get("lidar")
    if lid:
        v = dict(lid["voxelize"])
        if v.get("max_num_points", -1) > 0 and "voxelize" in lid:
            out["voxelize"] = Voxelization(**v)
            out["voxelize_reduce"] = lid.get("voxelize_reduce", True)
        if lid["backbone"]["type"] in BACKBONES:
            out["lidar_backbone"] = BACKBONES.build(lid["backbone"])
    return out
