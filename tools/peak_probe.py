"""On-box denominators: what dense bf16 matrix throughput does THIS MI355X sustain under its power cap?
 (1) hipBLASLt through torch.matmul (8192^3 and 16384x8192x8192, bf16), (2) tools/probes/valu_mfma_probe (register-only MFMA stream)."""
import subprocess, sys, os, time
import torch
for (m, n, k) in ((8192, 8192, 8192), (16384, 8192, 8192), (4096, 4096, 16384)):
    a = torch.randn((m, k), device="cuda", dtype=torch.bfloat16)
    b = torch.randn((k, n), device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        (a @ b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        c = a @ b
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    print(f"torch.matmul bf16 {m}x{n}x{k}: {ms:.3f} ms  {2.0 * m * n * k / ms / 1e9:.0f} TFLOP/s", flush=True)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
print(subprocess.run([os.path.join(root, "ab", "valu_probe")], capture_output=True, text=True).stdout)
