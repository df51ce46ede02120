#!/usr/bin/env python
"""Turn a rocprofv3 result (rocpd SQLite `*_results.db`, or `*_kernel_stats.csv`) into the text summary kept under
profiles/.  usage: python tools/rocpd_summary.py gpurun_out/prof/bench_results.db > profiles/NAME.txt"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    m = re.search(r"conv_fwd_kernel<(.*)>\(", name)
    if m:
        f = [x.strip() for x in m.group(1).split(",")]
        tail = f[-12:]  # ST,SH,SW? -- demangled forms vary; keep the numeric tail
        return "cvvae::conv_fwd_kernel<..." + ",".join(tail) + ">"
    return name if len(name) < 110 else name[:107] + "..."


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats  ({path})")
    print(f"# total kernel time {tot / 1e3:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    print(f"{'calls':>6} {'total_ms':>10} {'avg_us':>10} {'pct':>6}  kernel")
    for name, n, t, avg, pct in rows:
        print(f"{n:6d} {t / 1e3:10.3f} {avg:10.2f} {pct:6.2f}  {short(name)}")


if __name__ == "__main__":
    main(sys.argv[1])
