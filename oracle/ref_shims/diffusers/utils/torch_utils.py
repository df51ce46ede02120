import torch


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    """diffusers.utils.torch_utils.randn_tensor: draw on the generator's device, move to `device`."""
    rand_device = device
    if generator is not None:
        gen_device = generator.device.type
        if gen_device != torch.device(device).type and gen_device == "cpu":
            rand_device = "cpu"
    return torch.randn(shape, generator=generator, device=rand_device, dtype=dtype).to(device)
