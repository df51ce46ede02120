"""Import the UNMODIFIED reference modules from /root/reference.  TEST INFRASTRUCTURE; build container only
(/root/reference does not exist on the GPU box -- nothing on the `-m gpu` / smoke / bench path imports this).

The reference's `models/*.py` need `diffusers` and (at call time) `xformers`; neither is installed, so
oracle/ref_shims/ is put on sys.path first.  The reference package is loaded under the alias
`cvvae_ref_models` so it cannot collide with this repo's own drop-in `models/` package.
"""
import importlib
import importlib.util
import os
import sys

REF_ROOT = os.environ.get("CVVAE_REFERENCE_ROOT", "/root/reference")
_SHIMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_shims")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "models", "modeling_vae.py"))


def load_reference():
    """returns the reference `models.modeling_vae` module (classes CVVAEModel, CVVAESD3Model)."""
    if "cvvae_ref_models.modeling_vae" in sys.modules:
        return sys.modules["cvvae_ref_models.modeling_vae"]
    if not reference_available():
        raise FileNotFoundError(f"reference not found under {REF_ROOT}")
    if _SHIMS not in sys.path:
        sys.path.insert(0, _SHIMS)
    pkg_dir = os.path.join(REF_ROOT, "models")
    spec = importlib.util.spec_from_file_location(
        "cvvae_ref_models", os.path.join(pkg_dir, "__init__.py"), submodule_search_locations=[pkg_dir])
    pkg = importlib.util.module_from_spec(spec)
    sys.modules["cvvae_ref_models"] = pkg
    spec.loader.exec_module(pkg)
    return importlib.import_module("cvvae_ref_models.modeling_vae")


def load_reference_constraint():
    """returns the reference module lvdm/modules/diffusionmodules/vae_models_sd3.py (classes Decoder, DecoderWith3DWrapper)
    with its sibling vae_blocks_sd3.py, loaded as the synthetic package `cvvae_ref_constraint` (the real package __init__
    chain of lvdm pulls in the training stack)."""
    name = "cvvae_ref_constraint.vae_models_sd3"
    if name in sys.modules:
        return sys.modules[name]
    d = os.path.join(REF_ROOT, "lvdm", "modules", "diffusionmodules")
    if not os.path.isfile(os.path.join(d, "vae_models_sd3.py")):
        raise FileNotFoundError(f"reference not found under {REF_ROOT}")
    if _SHIMS not in sys.path:
        sys.path.insert(0, _SHIMS)
    import types

    pkg = types.ModuleType("cvvae_ref_constraint")
    pkg.__path__ = [d]
    sys.modules["cvvae_ref_constraint"] = pkg
    return importlib.import_module(name)



def load_reference_ldm():
    """returns the reference module lvdm/modules/diffusionmodules/model.py (classes Encoder, Decoder, EncoderWith3DWrapper,
    DecoderWith3DWrapper -- the SD2.1-family 2-D halves), unmodified.  Its relative import `...modules.attention` is served by the
    reference's own lvdm/modules/attention.py, both loaded into a synthetic package tree `cvvae_ref_lvdm` whose package __init__
    files are empty (the real ones pull in the conditioners and training engines)."""
    name = "cvvae_ref_lvdm.modules.diffusionmodules.model"
    if name in sys.modules:
        return sys.modules[name]
    root = os.path.join(REF_ROOT, "lvdm")
    if not os.path.isfile(os.path.join(root, "modules", "diffusionmodules", "model.py")):
        raise FileNotFoundError(f"reference not found under {REF_ROOT}")
    if _SHIMS not in sys.path:
        sys.path.insert(0, _SHIMS)
    import types

    for pkg, path in (("cvvae_ref_lvdm", root), ("cvvae_ref_lvdm.modules", os.path.join(root, "modules")),
                      ("cvvae_ref_lvdm.modules.diffusionmodules", os.path.join(root, "modules", "diffusionmodules"))):
        m = types.ModuleType(pkg)
        m.__path__ = [path]
        sys.modules[pkg] = m
    return importlib.import_module(name)
