#!/usr/bin/env python
"""Registers / scratch / LDS / occupancy of every kernel in the given .hip files (default: all conv instances), from
hipcc's -Rpass-analysis=kernel-resource-usage remarks.  usage: python tools/kernel_resources.py [file.hip ...]"""
import glob, os, re, subprocess, sys
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cvvae_amd", "csrc")
files = sys.argv[1:] or sorted(glob.glob(os.path.join(csrc, "conv_inst_*.hip")))
for f in files:
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-c", f,
                          "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], cwd=csrc, capture_output=True, text=True).stderr
    cur = {}
    for line in out.splitlines():
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+?):\s+(\S+) \[-Rpass", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
        cur[k] = v
        if k.startswith("LDS Size"):
            n = cur["name"]
            t = re.findall(r"L[ib](\d+)E", n)
            tag = "bf16" if "DF16b" in n else ("f16" if "DF16_" in n else "")
            print(f"{os.path.basename(f):16s} {tag:4s} {','.join(t):42s} vgpr={cur.get('VGPRs')} agpr={cur.get('AGPRs')} "
                  f"scratch={cur.get('ScratchSize [bytes/lane]')} spill={cur.get('VGPRs Spill')} occ={cur.get('Occupancy [waves/SIMD]')} lds={v}")
