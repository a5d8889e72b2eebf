"""Config system mirror of the reference's ``pcdet/config.py`` (reference lines 16-85): a global
``cfg`` attribute-dict, ``cfg_from_yaml_file`` with ``_BASE_CONFIG_`` include, ``cfg_from_list`` for
``--set`` overrides.  ``easydict`` is not a dependency: ``AttrDict`` below provides the same
attribute/dict dual access the reference modules rely on (``model_cfg.get('X')``, ``model_cfg.X``).
"""
from __future__ import annotations

from ast import literal_eval
from pathlib import Path

import yaml


class AttrDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            return AttrDict(v)
        if isinstance(v, (list, tuple)):
            return type(v)(AttrDict._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, AttrDict._wrap(v))

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def update(self, d=None, **kw):
        for k, v in dict(d or {}, **kw).items():
            self[k] = v


EasyDict = AttrDict  # name used by code written against the reference


def log_config_to_file(cfg, pre='cfg', logger=None):
    for key, val in cfg.items():
        if isinstance(val, AttrDict):
            logger.info('\n%s.%s = edict()' % (pre, key))
            log_config_to_file(val, pre=pre + '.' + key, logger=logger)
            continue
        logger.info('%s.%s: %s' % (pre, key, val))


def cfg_from_list(cfg_list, config):
    """``--set A.B.C value`` overrides with literal_eval + type check (reference config.py:16-48)."""
    assert len(cfg_list) % 2 == 0
    for k, v in zip(cfg_list[0::2], cfg_list[1::2]):
        keys = k.split('.')
        d = config
        for sub in keys[:-1]:
            assert sub in d, 'NotFoundKey: %s' % sub
            d = d[sub]
        sub = keys[-1]
        assert sub in d, 'NotFoundKey: %s' % sub
        try:
            value = literal_eval(v)
        except Exception:
            value = v
        cur = d[sub]
        if type(value) != type(cur) and isinstance(cur, AttrDict):
            for src in value.split(','):
                ck, cv = src.split(':')
                cur[ck] = type(cur[ck])(cv)
        elif type(value) != type(cur) and isinstance(cur, list):
            d[sub] = [type(cur[0])(x) for x in value.split(',')]
        else:
            assert type(value) == type(cur), 'type {} does not match original type {}'.format(type(value), type(cur))
            d[sub] = value


def merge_new_config(config, new_config):
    """Recursive merge; non-dict values (lists included) replace wholesale (reference :51-68)."""
    if '_BASE_CONFIG_' in new_config:
        with open(new_config['_BASE_CONFIG_'], 'r') as f:
            config.update(AttrDict(yaml.safe_load(f)))
    for key, val in new_config.items():
        if not isinstance(val, dict):
            config[key] = val
            continue
        if key not in config:
            config[key] = AttrDict()
        merge_new_config(config[key], val)
    return config


def cfg_from_yaml_file(cfg_file, config):
    with open(cfg_file, 'r') as f:
        merge_new_config(config=config, new_config=yaml.safe_load(f))
    return config


cfg = AttrDict()
cfg.ROOT_DIR = (Path(__file__).resolve().parent / '../').resolve()
cfg.LOCAL_RANK = 0
