#!/usr/bin/env python
"""The reference's arithmetic in fp16 / bf16 on PyTorch-ROCm's own GPU kernels (the oracle restatement -- plain PyTorch ops, pinned to the
reference's modules in fp32 -- through MIOpen / ATen), measured against the reference's fp32 CPU fixtures with the metric the device model
is measured with (oracle/parity.py::measure).  A second yardstick for the 16-bit tolerance bands beside oracle/make_noise.py (the
reference's own modules on the CPU): the CPU fp16 run of cfg 3 did not finish in 3.5 h, this one takes seconds.  A measurement aid only.
    usage (GPU box): python tools/torch_gpu_noise.py [case ...] > gpurun_out/torch_gpu_noise.json
"""
import json, os, sys, time
os.environ.setdefault("MIOPEN_FIND_MODE", "2")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import cvvae_oracle as O
from oracle import parity as P
from oracle.seeded import seeded_state_dict
from oracle.shapes import state_dict_shapes


class _Out:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class GpuOracle:
    """what parity.measure() needs of a model: dtype, device, encode(x).latent_dist.parameters, decode(z).sample"""

    def __init__(self, family, dtype, wseed):
        self.family, self.dtype, self.device = family, dtype, torch.device("cuda")
        self.sd = {k: v.to(dtype).cuda() for k, v in seeded_state_dict(state_dict_shapes(family), wseed).items()}

    def encode(self, x):
        return _Out(latent_dist=_Out(parameters=O.encode_moments(x, self.sd, {}, self.family)))

    def decode(self, z):
        return _Out(sample=O.decode_sample(z, self.sd, {}, self.family))


def main(cases):
    out = {}
    for name in cases:
        family, over, shape, wseed, xseed, s = P.case_of(name)
        assert not over, f"{name}: only the default (one window, one tile) configurations"
        for tag, dt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
            m = GpuOracle(family, dt, wseed)
            with torch.no_grad():
                t0 = time.time()
                r = P.measure(m, name)
                torch.cuda.synchronize()
            r.update(seconds=round(time.time() - t0, 1), torch=torch.__version__,
                     source="oracle restatement on PyTorch-ROCm GPU kernels (MIOpen / ATen), tools/torch_gpu_noise.py")
            out.setdefault(name, {})[tag] = r
            print(P.fmt("gpu-" + tag, r), file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1:] or ["cfg2_vae3d_t17_256", "cfg3_sd3_t17_512"])
