"""The boundary is a C ABI: tests/c/c_abi_conv.c uses libcvvae_hip.so from plain C (hipMalloc'd buffers, no Python objects,
no torch) -- pack a weight, run a convolution, compare with a host dot product, exercise an error return."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_library_from_plain_c(tmp_path):
    gcc = shutil.which("gcc") or "gcc"
    exe = os.path.join(tmp_path, "c_abi_conv")
    lib_dir = os.path.join(ROOT, "cvvae_amd")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    # plain gcc: the HIP runtime is only used for hipMalloc / hipMemcpy (its C header wants the platform macro)
    subprocess.run([gcc, "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(rocm, "include"), "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c", "c_abi_conv.c"), "-L" + lib_dir, "-lcvvae_hip",
                    "-L" + os.path.join(rocm, "lib"), "-lamdhip64", "-Wl,-rpath," + lib_dir,
                    "-Wl,-rpath," + os.path.join(rocm, "lib"), "-lm", "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "C_ABI_OK" in r.stdout, r.stdout + r.stderr
