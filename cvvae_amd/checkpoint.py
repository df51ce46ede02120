"""Checkpoint tooling for the codec's on-disk format.

The inference scripts load a diffusers-style directory (`<path>/<subfolder>/config.json` +
`diffusion_pytorch_model.safetensors`, /root/reference/cvvae_inference_video.py:11).  Training writes Lightning
checkpoints instead: a `state_dict` whose VAE tensors live under the same `encoder.*` / `decoder.*` names next to the loss,
discriminator, constraint-decoder and EMA tensors (/root/reference/lvdm/models/autoencoder.py:68-86 `apply_ckpt`,
configs/cvvae_sd3_constraint_training.yaml:10-37).  `convert_training_checkpoint` extracts the codec from such a file
(optionally its EMA shadow, stored by sgm's LitEma under the parameter name with the dots removed) and writes the
diffusers-style directory the drop-in classes read.  No GPU is needed for any of this.
"""
import argparse
from typing import Dict, Optional

import torch


def load_any(path: str) -> Dict[str, torch.Tensor]:
    """state dict of a .ckpt / .pt / .bin (torch.save; Lightning nests it under 'state_dict') or .safetensors file"""
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    try:
        obj = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:  # pickle.UnpicklingError: Lightning checkpoints may carry non-tensor objects (hyper-parameters, callbacks)
        raise RuntimeError(
            f"{path}: torch.load(weights_only=True) refused this checkpoint ({type(e).__name__}: {str(e).splitlines()[0][:160]}). "
            "It holds pickled objects besides tensors.  If you trust the file, allow-list the reported classes with "
            "torch.serialization.add_safe_globals([...]) before calling, or re-export it once with "
            "torch.save({'state_dict': torch.load(path, weights_only=False)['state_dict']}, new_path).") from e
    if isinstance(obj, dict) and "state_dict" in obj and isinstance(obj["state_dict"], dict):
        obj = obj["state_dict"]
    if not isinstance(obj, dict):
        raise ValueError(f"{path}: not a state dict")
    return obj


def extract_codec_state(sd: Dict[str, torch.Tensor], expected: Dict[str, torch.Size], use_ema: bool = False,
                        prefix: str = "") -> Dict[str, torch.Tensor]:
    """Pick the `encoder.*` / `decoder.*` tensors `expected` names (key -> shape) out of a training state dict.
    prefix: an extra leading module path in the file (e.g. 'first_stage_model.').  use_ema: take LitEma's shadow copies
    ('model_ema.' + name without dots) where present.  Raises on missing keys or shape mismatches."""
    out, missing, bad = {}, [], []
    for k, shp in expected.items():
        src = None
        if use_ema:
            src = sd.get("model_ema." + (prefix + k).replace(".", ""))
        if src is None:
            src = sd.get(prefix + k)
        if src is None:
            missing.append(k)
            continue
        if tuple(src.shape) != tuple(shp):
            bad.append(f"{k}: file {tuple(src.shape)} vs model {tuple(shp)}")
            continue
        out[k] = src.detach().clone()
    if missing or bad:
        raise KeyError(f"checkpoint does not hold the codec: {len(missing)} missing (first: {missing[:3]}), "
                       f"{len(bad)} mismatched (first: {bad[:3]})")
    return out


def convert_training_checkpoint(src: str, dst_dir: str, family: str = "sd3", use_ema: bool = False, prefix: str = "",
                                torch_dtype: Optional[torch.dtype] = None, **config) -> str:
    """src: Lightning .ckpt / .safetensors of a CV-VAE training run -> dst_dir/{config.json, diffusion_pytorch_model.safetensors}
    loadable by CVVAEModel / CVVAESD3Model.from_pretrained(dirname(dst_dir), subfolder=basename(dst_dir))."""
    from .modeling import CVVAEModel, CVVAESD3Model

    cls = {"sd3": CVVAESD3Model, "vae3d_sd3": CVVAESD3Model, "vae3d": CVVAEModel}[family]
    model = cls(**config)
    expected = {k: v.shape for k, v in model.state_dict().items()}
    state = extract_codec_state(load_any(src), expected, use_ema=use_ema, prefix=prefix)
    model.load_state_dict(state, strict=True)
    if torch_dtype is not None:
        model = model.to(torch_dtype)
    model.save_pretrained(dst_dir)
    return dst_dir


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("src")
    ap.add_argument("dst_dir")
    ap.add_argument("--family", default="sd3", choices=["sd3", "vae3d_sd3", "vae3d"])
    ap.add_argument("--ema", action="store_true")
    ap.add_argument("--prefix", default="")
    ap.add_argument("--dtype", default=None, choices=[None, "float16", "bfloat16", "float32"])
    a = ap.parse_args(argv)
    dt = getattr(torch, a.dtype) if a.dtype else None
    print(convert_training_checkpoint(a.src, a.dst_dir, a.family, a.ema, a.prefix, dt))


if __name__ == "__main__":
    main()
