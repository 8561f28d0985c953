"""Name -> class registry with the build protocol the reference's configs use.

Mirrors ``mmcv.utils.Registry`` / ``build_from_cfg`` as the reference uses
them (depth/models/builder.py:4-44): ``cfg.pop('type')``
-> class lookup in this registry, then its parent -> ``cls(**cfg)``, with
``default_args`` applied via ``setdefault``.
"""
import inspect


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict):
        raise TypeError(f'cfg must be a dict, but got {type(cfg)}')
    if 'type' not in cfg:
        if default_args is None or 'type' not in default_args:
            raise KeyError(
                f'`cfg` or `default_args` must contain the key "type", but got {cfg}\n{default_args}')
    if not isinstance(registry, Registry):
        raise TypeError(f'registry must be a Registry object, but got {type(registry)}')
    if not (isinstance(default_args, dict) or default_args is None):
        raise TypeError(f'default_args must be a dict or None, but got {type(default_args)}')

    args = cfg.copy()
    if default_args is not None:
        for name, value in default_args.items():
            args.setdefault(name, value)

    obj_type = args.pop('type')
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError(f'{obj_type} is not in the {registry.name} registry')
    elif inspect.isclass(obj_type):
        obj_cls = obj_type
    else:
        raise TypeError(f'type must be a str or valid type, but got {type(obj_type)}')
    try:
        return obj_cls(**args)
    except Exception as e:
        raise type(e)(f'{obj_cls.__name__}: {e}')


class Registry:

    def __init__(self, name, build_func=None, parent=None):
        self._name = name
        self._module_dict = {}
        self._children = {}
        self.parent = parent
        self.build_func = build_func or (parent.build_func if parent is not None else build_from_cfg)
        if parent is not None:
            parent._children[name] = self

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return self.get(key) is not None

    def __repr__(self):
        return f'Registry(name={self._name}, items={sorted(self._module_dict)})'

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        if key in self._module_dict:
            return self._module_dict[key]
        reg = self.parent
        while reg is not None:
            if key in reg._module_dict:
                return reg._module_dict[key]
            reg = reg.parent
        return None

    def build(self, *args, **kwargs):
        return self.build_func(*args, **kwargs, registry=self)

    def _register(self, cls, name=None, force=False):
        if not inspect.isclass(cls) and not inspect.isfunction(cls):
            raise TypeError(f'module must be a class or function, but got {type(cls)}')
        names = [name] if isinstance(name, str) else (name or [cls.__name__])
        for n in names:
            if not force and n in self._module_dict:
                raise KeyError(f'{n} is already registered in {self.name}')
            self._module_dict[n] = cls

    def register_module(self, name=None, force=False, module=None):
        if module is not None:
            self._register(module, name, force)
            return module

        def _deco(cls):
            self._register(cls, name, force)
            return cls

        return _deco
