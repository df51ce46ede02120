"""TEST INFRASTRUCTURE ONLY -- xformers stand-in (see ../diffusers/__init__.py)."""
from . import ops  # noqa: F401
