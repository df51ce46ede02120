import torch.nn as nn

_ACT = {"swish": nn.SiLU, "silu": nn.SiLU, "mish": nn.Mish, "gelu": nn.GELU, "relu": nn.ReLU}


def get_activation(name):
    return _ACT[name.lower()]()
