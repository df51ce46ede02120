"""conv_fwd_kernel's blockIdx -> tile map (cvvae_amd/csrc/tile_map.h) is plain C++: its bijectivity and the
short-tiles-last property are checked on the host for every grid shape the launcher can produce (tests/c/tile_map_test.cpp)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_block_to_tile_map_is_a_bijection(tmp_path):
    exe = str(tmp_path / "tile_map_test")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "cvvae_amd", "csrc"),
                    os.path.join(ROOT, "tests", "c", "tile_map_test.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert "tile map ok" in out and "time-fold plan ok" in out, out
