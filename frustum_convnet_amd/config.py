"""Global `cfg` with the reference's key tree (configs/config.py:57-192) and its yaml / CLI merge
behaviour (configs/config.py:231-353), restated compactly.  The reference's cfgs/*.yaml load unchanged:
unknown keys raise KeyError, tuples written as strings are literal_eval'd, list<->tuple are coerced,
other type mismatches raise ValueError, and the tree can be frozen.
"""
import copy
from ast import literal_eval

import yaml


class AttrDict(dict):
    _FROZEN = "__frozen__"

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.__dict__[AttrDict._FROZEN] = False

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.__dict__[AttrDict._FROZEN]:
            raise AttributeError('Attempted to set "%s" to "%s", but AttrDict is immutable' % (name, value))
        self[name] = value

    def immutable(self, flag):
        self.__dict__[AttrDict._FROZEN] = flag
        for v in self.values():
            if isinstance(v, AttrDict):
                v.immutable(flag)

    def is_immutable(self):
        return self.__dict__[AttrDict._FROZEN]


def _tree(d):
    return AttrDict({k: _tree(v) if isinstance(v, dict) else v for k, v in d.items()})


_DEFAULTS = {
    "TRAIN": {"WEIGHTS": "", "BATCH_SIZE": 32, "START_EPOCH": 0, "MAX_EPOCH": 200, "OPTIMIZER": "adam",
              "BASE_LR": 0.001, "MIN_LR": 1e-5, "LR_POLICY": "step", "GAMMA": 0.1, "LR_STEPS": [20],
              "MOMENTUM": 0.9, "WEIGHT_DECAY": 0.0, "DATASET": "train"},
    "MODEL": {"FILE": "", "NUM_CLASSES": 2},
    "TEST": {"WEIGHTS": "", "BATCH_SIZE": 32, "METHOD": "top", "THRESH": 0.1, "DATASET": "val"},
    "DATA": {"DATASET_NAME": "KITTI", "MAX_DEPTH": 70, "FILE": "", "DATA_ROOT": "kitti",
             "WITH_EXTRA_FEAT": True, "EXTRA_FEAT_DIM": 1, "NUM_SAMPLES": 1024, "NUM_SAMPLES_DET": 512,
             "CAR_ONLY": True, "PEOPLE_ONLY": False, "RTC": True, "NUM_HEADING_BIN": 12,
             "STRIDE": (0.25, 0.5, 1.0, 2.0), "HEIGHT_HALF": (0.25, 0.5, 1.0, 2.0), "EXTEND_FROM_DET": False},
    "LOSS": {"BOX_LOSS_WEIGHT": 1.0, "CORNER_LOSS_WEIGHT": 10.0, "HEAD_REG_WEIGHT": 20.0,
             "SIZE_REG_WEIGHT": 20.0},
    "RESUME": False, "NUM_GPUS": 1, "OUTPUT_DIR": "/tmp", "SAVE_SUB_DIR": "test", "OVER_WRITE_TEST_FILE": "",
    "FROM_RGB_DET": False, "NUM_WORKERS": 4, "USE_TFBOARD": False, "EVAL_MODE": False, "IOU_THRESH": 0.7,
    "disp": 50,
}

cfg = _tree(_DEFAULTS)


def reset_cfg():
    """Back to the defaults (test helper; the reference has no equivalent)."""
    cfg.immutable(False)
    fresh = _tree(copy.deepcopy(_DEFAULTS))
    cfg.clear()
    cfg.update(fresh)
    return cfg


def _decode(v):
    if isinstance(v, dict):
        return _tree(v)
    if not isinstance(v, str):
        return v
    try:
        return literal_eval(v)
    except (ValueError, SyntaxError):
        return v


def _coerce(new, old, full_key):
    if type(new) is type(old):
        return new
    if isinstance(old, str):
        return str(new)
    if isinstance(new, tuple) and isinstance(old, list):
        return list(new)
    if isinstance(new, list) and isinstance(old, tuple):
        return tuple(new)
    raise ValueError("Type mismatch ({} vs. {}) with values ({} vs. {}) for config key: {}".format(
        type(old), type(new), old, new, full_key))


def _merge(src, dst, path=()):
    for k, raw in src.items():
        full = ".".join(path + (k,))
        if k not in dst:
            raise KeyError("Non-existent config key: {}".format(full))
        v = _coerce(_decode(copy.deepcopy(raw)), dst[k], full)
        if isinstance(v, AttrDict):
            _merge(v, dst[k], path + (k,))
        else:
            dst[k] = v


def merge_cfg_from_file(filename):
    with open(filename, "r") as f:
        loaded = yaml.safe_load(f.read())
    _merge(_tree(loaded or {}), cfg)


def merge_cfg_from_cfg(other):
    _merge(other, cfg)


def merge_cfg_from_list(kv):
    assert len(kv) % 2 == 0
    for full, v in zip(kv[0::2], kv[1::2]):
        d = cfg
        parts = full.split(".")
        for p in parts[:-1]:
            assert p in d, "Non-existent key: {}".format(full)
            d = d[p]
        assert parts[-1] in d, "Non-existent key: {}".format(full)
        d[parts[-1]] = _coerce(_decode(v), d[parts[-1]], full)


def assert_and_infer_cfg(make_immutable=True):
    if make_immutable:
        cfg.immutable(True)
