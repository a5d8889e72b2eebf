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

    def setdefault(self, k, default=None):
        if k not in self:
            self[k] = default
        return self[k]


EasyDict = AttrDict  # name used by code written against the reference


def log_config_to_file(cfg, pre='cfg', logger=None):
    """One log line per leaf (``pre.KEY: value``), a header line per nested section - the lines ``tools/train.py`` writes at
    start-up (API of reference pcdet/config.py:7-13)."""
    stack = [(pre, cfg)]
    while stack:
        prefix, node = stack.pop()
        nested = []
        for key, val in node.items():
            path = f'{prefix}.{key}'
            if isinstance(val, AttrDict):
                nested.append((path, val))
            else:
                logger.info('%s: %s' % (path, val))
        for path, val in reversed(nested):               # depth first, sections in file order
            logger.info('\n%s = edict()' % path)
            stack.append((path, val))


def _parse_scalar(text):
    try:
        return literal_eval(text)
    except (ValueError, SyntaxError):
        return text


def _coerce(new, old, path, raw):
    """Value written by a ``--set`` override: it must keep the type of the value it replaces; strings are split into the
    existing container's shape (``a:1,b:2`` for a section, ``1,2,3`` for a list)."""
    if isinstance(new, type(old)) and isinstance(old, type(new)):
        return new
    if isinstance(old, AttrDict):
        out = AttrDict(old)
        for item in raw.split(','):
            sub, _, raw = item.partition(':')
            out[sub] = type(old[sub])(raw)
        return out
    if isinstance(old, list):
        return [type(old[0])(x) for x in raw.split(',')]
    raise AssertionError('type {} does not match original type {} at {}'.format(type(new), type(old), path))


def cfg_from_list(cfg_list, config):
    """``--set A.B.C value [A.D value ...]``: dotted paths must exist, values are python literals when they parse
    (API and semantics of reference pcdet/config.py:16-48)."""
    assert len(cfg_list) % 2 == 0, 'overrides come in (key, value) pairs'
    for dotted, text in zip(cfg_list[0::2], cfg_list[1::2]):
        *parents, leaf = dotted.split('.')
        node = config
        for name in parents:
            assert isinstance(node, dict) and name in node, 'NotFoundKey: %s' % name
            node = node[name]
        assert isinstance(node, dict) and leaf in node, 'NotFoundKey: %s' % leaf
        node[leaf] = _coerce(_parse_scalar(text), node[leaf], dotted, text)


def _load_yaml(path):
    with open(path, 'r') as f:
        return yaml.safe_load(f) or {}


def merge_new_config(config, new_config):
    """Overlay ``new_config`` on ``config``: sections merge recursively, everything else (lists included) replaces; a
    ``_BASE_CONFIG_`` entry first pulls in that yaml file at the same level (reference pcdet/config.py:51-68)."""
    base = new_config.get('_BASE_CONFIG_') if isinstance(new_config, dict) else None
    if base is not None:
        config.update(AttrDict(_load_yaml(base)))
    for key, val in new_config.items():
        if isinstance(val, dict):
            merge_new_config(config.setdefault(key, AttrDict()), val)
        else:
            config[key] = val
    return config


def cfg_from_yaml_file(cfg_file, config):
    return merge_new_config(config=config, new_config=_load_yaml(cfg_file))


cfg = AttrDict()
cfg.ROOT_DIR = (Path(__file__).resolve().parent / '../').resolve()
cfg.LOCAL_RANK = 0
