import functools, inspect


class _Cfg(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class ConfigMixin:
    config_name = "config.json"

    @property
    def config(self):
        return self._internal_dict


def register_to_config(init):
    @functools.wraps(init)
    def inner(self, *args, **kwargs):
        sig = inspect.signature(init)
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k != "self"}
        init(self, *args, **kwargs)
        self._internal_dict = _Cfg(cfg)

    return inner
