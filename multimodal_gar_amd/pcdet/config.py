"""Attribute-style config container compatible with the reference's EasyDict usage
(pcdet/config.py:83-85 ``cfg = EasyDict()``; model code calls ``cfg.KEY`` and ``cfg.get("KEY")``).
easydict is not installed in this image; this is a minimal stand-in plus the YAML loader."""
import yaml


class EasyDict(dict):
    def __init__(self, d=None, **kwargs):
        super().__init__()
        d = dict(d or {}, **kwargs)
        for k, v in d.items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def cfg_from_yaml_file(cfg_file, config=None):
    """Reference pcdet/config.py:71-80 (without _BASE_CONFIG_ recursion, which mil3.yaml does not use)."""
    config = EasyDict() if config is None else config
    with open(cfg_file, 'r') as f:
        new = yaml.safe_load(f)
    for k, v in (new or {}).items():
        config[k] = v
    return config


cfg = EasyDict()
