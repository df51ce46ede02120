import logging as _logging
from collections import OrderedDict
from packaging import version
import torch


class BaseOutput(OrderedDict):
    """dataclass-style output; fields set by @dataclass __init__ are mirrored as items."""

    def __post_init__(self):
        for k, v in self.__dict__.items():
            self[k] = v

    def __init_subclass__(cls):
        pass


def is_torch_version(op, v):
    import operator

    ops = {">=": operator.ge, ">": operator.gt, "<": operator.lt, "<=": operator.le, "==": operator.eq}
    return ops[op](version.parse(torch.__version__.split("+")[0]), version.parse(v))


def deprecate(*a, **k):
    return None


class logging:  # noqa: N801  (mirrors diffusers.utils.logging module API)
    @staticmethod
    def get_logger(name):
        return _logging.getLogger(name)
