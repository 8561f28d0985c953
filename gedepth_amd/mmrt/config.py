"""Python-file configs with ``_base_`` inheritance — the config half of the drop-in boundary.

Re-implements the behaviour the reference relies on from ``mmcv.utils.Config``
(reference call sites: tools/train.py:83-88, tools/test.py:90-97; semantics in
SURVEY.md Appendix A): ``_base_`` lists merged left→right then the child on
top, recursive dict merge, ``_delete_=True`` replaces instead of merging,
attribute *and* item access on every node, ``merge_from_dict`` for
``--options a.b=c``.
"""
import ast
import copy
import os.path as osp
import types

BASE_KEY = '_base_'
DELETE_KEY = '_delete_'


class ConfigDict(dict):
    """dict with attribute access (missing attribute -> AttributeError)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(
                f"'{type(self).__name__}' object has no attribute '{name}'")

    def __setattr__(self, name, value):
        self[name] = _wrap(value)

    def __delattr__(self, name):
        del self[name]

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def copy(self):
        return ConfigDict(super().copy())


def _wrap(obj):
    if isinstance(obj, ConfigDict):
        return obj
    if isinstance(obj, dict):
        return ConfigDict({k: _wrap(v) for k, v in obj.items()})
    if isinstance(obj, list):
        return [_wrap(v) for v in obj]
    if isinstance(obj, tuple):
        return tuple(_wrap(v) for v in obj)
    return obj


def _unwrap(obj):
    if isinstance(obj, dict):
        return {k: _unwrap(v) for k, v in obj.items()}
    if isinstance(obj, list):
        return [_unwrap(v) for v in obj]
    if isinstance(obj, tuple):
        return tuple(_unwrap(v) for v in obj)
    return obj


def _merge_a_into_b(a, b):
    """Merge child ``a`` over base ``b`` (returns a new dict)."""
    b = dict(b)
    for k, v in a.items():
        if isinstance(v, dict):
            v = dict(v)
            if v.pop(DELETE_KEY, False) or not isinstance(b.get(k), dict):
                b[k] = _merge_a_into_b(v, {})
            else:
                b[k] = _merge_a_into_b(v, b[k])
        else:
            b[k] = v
    return b


def _load_py(filename):
    filename = osp.abspath(osp.expanduser(filename))
    if not osp.isfile(filename):
        raise FileNotFoundError(f'config file {filename} does not exist')
    with open(filename, 'r') as f:
        text = f.read()
    ast.parse(text)  # raise SyntaxError early with the right filename
    scope = {'__file__': filename, '__name__': '_gedepth_cfg_'}
    exec(compile(text, filename, 'exec'), scope)
    cfg = {
        k: v
        for k, v in scope.items() if not k.startswith('__')
        and not isinstance(v, (types.ModuleType, types.FunctionType, type))
    }
    texts = [text]
    if BASE_KEY in cfg:
        base = cfg.pop(BASE_KEY)
        base = base if isinstance(base, (list, tuple)) else [base]
        merged = {}
        for rel in base:
            sub, sub_text = _load_py(osp.join(osp.dirname(filename), rel))
            dup = merged.keys() & sub.keys()
            if dup:
                raise KeyError(f'Duplicate key(s) {sorted(dup)} in base configs')
            merged.update(sub)
            texts.insert(-1, sub_text)
        cfg = _merge_a_into_b(cfg, merged)
    return cfg, '\n'.join(texts)


class Config:
    """``Config.fromfile(path)`` -> attribute/item-accessible config tree."""

    def __init__(self, cfg_dict=None, filename=None, text=''):
        cfg_dict = {} if cfg_dict is None else cfg_dict
        if not isinstance(cfg_dict, dict):
            raise TypeError(f'cfg_dict must be a dict, got {type(cfg_dict)}')
        object.__setattr__(self, '_cfg_dict', _wrap(cfg_dict))
        object.__setattr__(self, '_filename', filename)
        object.__setattr__(self, '_text', text)

    @staticmethod
    def fromfile(filename):
        cfg, text = _load_py(filename)
        return Config(cfg, filename=filename, text=text)

    @property
    def filename(self):
        return self._filename

    @property
    def text(self):
        return self._text

    @property
    def pretty_text(self):
        def fmt(v, ind):
            pad = ' ' * ind
            if isinstance(v, dict):
                if not v:
                    return 'dict()'
                body = ',\n'.join(f'{pad}    {k}={fmt(x, ind + 4)}' for k, x in v.items())
                return f'dict(\n{body})'
            if isinstance(v, (list, tuple)) and any(isinstance(x, dict) for x in v):
                o, c = ('[', ']') if isinstance(v, list) else ('(', ')')
                body = ',\n'.join(f'{pad}    {fmt(x, ind + 4)}' for x in v)
                return f'{o}\n{body}\n{pad}{c}'
            return repr(v)

        return '\n'.join(f'{k} = {fmt(v, 0)}' for k, v in _unwrap(self._cfg_dict).items()) + '\n'

    def dump(self, file=None):
        text = self.pretty_text
        if file is None:
            return text
        with open(file, 'w') as f:
            f.write(text)

    def merge_from_dict(self, options):
        """``{'a.b.c': v}`` style overrides (tools/train.py:52,87-88)."""
        nested = {}
        for full_key, v in options.items():
            d = nested
            keys = full_key.split('.')
            for k in keys[:-1]:
                d = d.setdefault(k, {})
            d[keys[-1]] = v
        merged = _merge_a_into_b(nested, _unwrap(self._cfg_dict))
        object.__setattr__(self, '_cfg_dict', _wrap(merged))

    def to_dict(self):
        return _unwrap(self._cfg_dict)

    def get(self, key, default=None):
        return self._cfg_dict.get(key, default)

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setattr__(self, name, value):
        self._cfg_dict[name] = _wrap(value)

    def __setitem__(self, name, value):
        self._cfg_dict[name] = _wrap(value)

    def __contains__(self, name):
        return name in self._cfg_dict

    def __iter__(self):
        return iter(self._cfg_dict)

    def __len__(self):
        return len(self._cfg_dict)

    def __repr__(self):
        return f'Config (path: {self._filename}): {self._cfg_dict!r}'


class DictAction:
    """argparse action for ``--options k=v k2=v2`` (tools/train.py:52)."""

    @staticmethod
    def parse_value(val):
        try:
            return ast.literal_eval(val)
        except (ValueError, SyntaxError):
            pass
        if val.lower() in ('true', 'false'):
            return val.lower() == 'true'
        if ',' in val:
            return [DictAction.parse_value(v) for v in val.strip('[]()').split(',')]
        return val

    @staticmethod
    def parse(pairs):
        out = {}
        for kv in pairs or []:
            k, v = kv.split('=', maxsplit=1)
            out[k] = DictAction.parse_value(v)
        return out
