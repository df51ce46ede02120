"""TEST INFRASTRUCTURE ONLY -- minimal stand-in for the `diffusers` symbols that
/root/reference/models/*.py import at module top (diffusers is not installed and there
is no network).  Only the golden-vector generator (oracle/make_golden.py) and the
oracle-validation tests put this directory on sys.path.  The product never does.

The only arithmetic restated here is third-party arithmetic that is NOT under
/root/reference: diffusers `Attention` (AttnProcessor2_0 with
_from_deprecated_attn_block=True), `DiagonalGaussianDistribution`, `randn_tensor`.
diffusers version is not pinned anywhere in the reference (no requirements file).
"""
__version__ = "0.0.0-shim"
