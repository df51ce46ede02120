"""The reference's OWN half-precision noise at the BASELINE shapes.  TEST INFRASTRUCTURE; build container only.

Runs the UNMODIFIED reference modules (oracle/ref_loader.py) on the CPU in fp16 and bf16 on the same seeded weights and
inputs as the fp32 fixtures of `make_golden.py big`, and measures them against those fp32 fixtures with exactly the metric
the device model is measured with (oracle/parity.py::measure: encode(x) compared on the latent, decode() run on the
REFERENCE's fp32 latent).  The result -- how far the reference's own fp16 / bf16 run is from its fp32 run at THIS shape -- is
the yardstick the GPU tolerance bands are derived from (tests/test_gpu_baseline_shapes.py) and what bench.py prints next to
`parity`.

    python -m oracle.make_noise [case ...] [--dtypes bf16,f16]   -> tests/golden/ref_self_noise.json   (entries are merged, not replaced)
(fp16 on the CPU is 3x slower than bf16 on the vae3d cases and did not finish in 3.5 h on cfg 3: --dtypes bf16 skips it)
"""
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import parity as P  # noqa: E402
from oracle.golden_cases import BIG_CASES  # noqa: E402
from oracle.ref_loader import load_reference  # noqa: E402

OUT = os.path.join(P.GOLDEN_DIR, "ref_self_noise.json")
TAG = {torch.float16: "f16", torch.bfloat16: "bf16"}


class _CpuModel:
    """what parity.measure() needs of a model: dtype, device, encode, decode"""

    def __init__(self, m, dtype):
        self.m, self.dtype, self.device = m, dtype, torch.device("cpu")

    def encode(self, x):
        return self.m.encode(x)

    def decode(self, z):
        return self.m.decode(z)


def main(only):
    want = ("bf16", "f16")
    for a in list(only):
        if a.startswith("--dtypes"):
            only.remove(a)
            want = tuple(a.split("=", 1)[1].split(",")) if "=" in a else want
    ref = load_reference()
    torch.set_grad_enabled(False)
    table = {}
    if os.path.isfile(OUT):
        with open(OUT) as f:
            table = json.load(f)
    for name, (family, over, shape, wseed, xseed, s) in BIG_CASES.items():
        if only and name not in only:
            continue
        if not os.path.isfile(os.path.join(P.GOLDEN_DIR, name + ".npz")):
            print(f"{name}: fp32 fixture missing, skipped", flush=True)
            continue
        cls = ref.CVVAESD3Model if family == "sd3" else ref.CVVAEModel
        for dtype in (torch.bfloat16, torch.float16):
            if TAG[dtype] not in want:
                continue
            model = cls(**over).eval()
            P.load_seeded(model, wseed)
            model = model.to(dtype)
            t0 = time.time()
            r = P.measure(_CpuModel(model, dtype), name)
            r["seconds"] = round(time.time() - t0, 1)
            r["threads"] = torch.get_num_threads()
            r["torch"] = torch.__version__
            table.setdefault(name, {})[TAG[dtype]] = r
            print(P.fmt("ref-" + TAG[dtype], r), f"({r['seconds']} s)", flush=True)
            with open(OUT, "w") as f:
                json.dump(table, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1:])
