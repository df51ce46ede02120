#!/usr/bin/env python
"""Summarise rocprofv3 --pmc output (…_counter_collection.csv files under a directory): one line per kernel with dispatch
count, mean duration and the mean per-dispatch value of every counter collected; HBM traffic columns are derived as the
MI355X guide prescribes (FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read ->
doubled).  usage: python tools/pmc_summary.py DIR [name-filter]"""
import csv, glob, os, re, sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def short(n):
    """library-style instance name from rocprofv3's kernel name.  rocprofv3's demangler garbles the leading template
    arguments of conv_fwd_kernel (T, KT, KH, KW, ST, SH, SW); the tail TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS is reliable and
    identifies the instance (suffix of the name cvvae_conv_kernel_name() reports: conv_k..._s..._<suffix>)."""
    m = re.search(r"conv_fwd_kernel<(.*?)>\(", n)
    if m:
        f = [x.strip() for x in m.group(1).split(",")]
        sfx = ""
        if f[-1] in ("true", "false"):  # ABI 6-7: one trailing bool XP
            sfx = "_xp" if f[-1] == "true" else ""
            f = f[:-1]
        else:  # ABI 11: the name ends ...,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS,XP,NB,LD -- XP 0 / 1 split precision / 2, 3 fast fp32, NB
            # N-blocks per wave, LD 1 = DMA-staged.  Counted from the END: the number of (garbled) leading fields varies.
            xp, nb, ld = (int(v) for v in f[-3:])
            sfx = {0: "", 1: "_xp", 2: "_xq", 3: "_xq6"}[xp] + ("_nb2" if nb == 2 else "") + ("_dma" if ld == 1 else "")
            f = f[:-3]
        t = f[-9:]
        return f"conv_*_t{t[0]}x{t[1]}x{t[2]}_w{t[3]}x{t[4]}x{t[5]}_c{16 * int(t[6])}_pro{t[7]}_ups{t[8]}{sfx}"
    n = re.sub(r"\(.*", "", n)
    n = re.sub(r"^void ", "", n)
    return n[-60:]


def _fingerprint():
    """the kernel sources these counters were measured on (cvvae_amd._lib.source_fingerprint): bench.py compares it with the running tree"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from cvvae_amd import _lib
    return _lib.source_fingerprint()


def main(d, flt=None, json_out=None, sq_out=None, steps=0):
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0, 0.0]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if flt and flt not in k:
                continue
            a = acc[k][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("# per-dispatch means.  fetch_GB = 2 * FETCH_SIZE KiB (gfx950 correction), write_GB = WRITE_SIZE KiB, "
          "l2_hit = TCC_HIT/(HIT+MISS), clk_GHz = GRBM_GUI_ACTIVE / 8 XCDs / duration")
    print(f"{'kernel':58s} {'n':>5s} {'dur_us':>9s} {'fetch_GB':>9s} {'write_GB':>9s} {'l2_hit':>7s} {'clk_GHz':>8s}")
    rows = []
    for k, cs in acc.items():
        def mean(c):
            return cs[c][1] / cs[c][0] if c in cs and cs[c][0] else None
        n = min(v[0] for v in cs.values() if v[0])  # a counter collected in two passes sees every dispatch twice
        dur = max((v[2] / v[0] for v in cs.values() if v[0]), default=0.0)
        fe, wr = mean("FETCH_SIZE"), mean("WRITE_SIZE")
        hit, miss, ga = mean("TCC_HIT_sum"), mean("TCC_MISS_sum"), mean("GRBM_GUI_ACTIVE")
        durg = cs["GRBM_GUI_ACTIVE"][2] / cs["GRBM_GUI_ACTIVE"][0] if "GRBM_GUI_ACTIVE" in cs else None
        rows.append((dur * n, k, n, dur, fe, wr, hit, miss, ga, durg))
    if json_out:
        import json
        js = {}
        for _, k, n, dur, fe, wr, hit, miss, ga, durg in rows:
            if k.startswith("conv_*_") and fe is not None and wr is not None:
                js[k[len("conv_*_"):]] = {"fetch_bytes": round(fe * 2 * 1024), "write_bytes": round(wr * 1024), "dispatches": n,
                                          "mean_duration_us": round(dur, 1)}
        json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py; fetch = 2 x FETCH_SIZE KiB "
                             "(gfx950 correction of MI355X_MICROARCH.md), write = WRITE_SIZE KiB; mean per dispatch",
                   "kernels": js}, open(json_out, "w"), indent=1)
        # whole-step HBM traffic (every kernel of the run, conv or not) when the number of profiled steps is known
        tot = sum((fe * 2 * 1024 + wr * 1024) * n for _, k, n, dur, fe, wr, *_ in rows if fe is not None and wr is not None)
        doc = json.load(open(json_out))
        doc["library_source_fingerprint"] = _fingerprint()
        if steps:
            doc["steps_profiled"] = steps
            doc["step_total_bytes"] = round(tot / steps)
        json.dump(doc, open(json_out, "w"), indent=1)
    if sq_out:
        # MFMA-busy per kernel: SQ_VALU_MFMA_BUSY_CYCLES (cycles a SIMD's matrix pipe is busy, summed over the chip) over the SIMD
        # cycles of the dispatch = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 * 256 CUs * 4 SIMDs -- the gfx94x MfmaUtil formula
        # (ROCm 7.2 ships no gfx950 derived metrics).  The SQ wait buckets are quad-cycle counts: reported as shares of WAVE_CYCLES.
        import json
        js = {}
        for k, cs in acc.items():
            def mean(c):
                return cs[c][1] / cs[c][0] if c in cs and cs[c][0] else None
            mb, ga = mean("SQ_VALU_MFMA_BUSY_CYCLES"), mean("GRBM_GUI_ACTIVE")
            if mb is None or not ga:
                continue
            e = {"mfma_busy": round(mb / (ga / 8.0 * 1024.0), 4), "dispatches": min(v[0] for v in cs.values() if v[0])}
            durg = cs["GRBM_GUI_ACTIVE"][2] / cs["GRBM_GUI_ACTIVE"][0]  # mean duration (us) of the dispatches of that pass
            if durg:
                # shader clock during the dispatch; mfma_busy x clock / 2.4 GHz = executed share of the NOMINAL 2.5 PFLOP/s peak
                e["clk_ghz"] = round(ga / 8.0 / (durg * 1e3), 3)
                e["mfma_busy_x_clk_over_nominal"] = round(e["mfma_busy"] * e["clk_ghz"] / 2.4, 4)
            wc = mean("SQ_WAVE_CYCLES")
            for c, nm in (("SQ_WAIT_ANY", "wait_any"), ("SQ_WAIT_INST_ANY", "wait_inst_any"), ("SQ_ACTIVE_INST_ANY", "active_inst_any"),
                          ("SQ_WAIT_INST_LDS", "wait_inst_lds")):
                if mean(c) is not None and wc:
                    e[nm + "_share_of_wave_cycles"] = round(mean(c) / wc, 4)
            sb = mean("SQ_BUSY_CYCLES")
            if sb:
                e["sq_busy_cycles"] = round(sb)
            for c in cs:
                if c.startswith("SQ_INSTS_VALU_MFMA"):
                    e[c.lower()] = round(mean(c))
            js[k[len("conv_*_"):] if k.startswith("conv_*_") else k] = e
        json.dump({"source": "rocprofv3 --pmc SQ_* passes of bench.py (separate from the traffic passes); mfma_busy = "
                             "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs), mean per dispatch",
                   "library_source_fingerprint": _fingerprint(), "kernels": js}, open(sq_out, "w"), indent=1)
    fmt = lambda v, s: ("%" + s) % v if v is not None else " " * (int(s.split(".")[0]) - 1) + "-"
    for _, k, n, dur, fe, wr, hit, miss, ga, durg in sorted(rows, reverse=True):
        print(f"{k:58s} {n:5d} {dur:9.1f} {fmt(fe * 2 * 1024 / 1e9 if fe is not None else None, '9.3f')} "
              f"{fmt(wr * 1024 / 1e9 if wr is not None else None, '9.3f')} "
              f"{fmt(hit / (hit + miss) if hit is not None and hit + miss > 0 else None, '7.3f')} "
              f"{fmt(ga / 8.0 / (durg * 1e3) if ga is not None and durg else None, '8.3f')}")


if __name__ == "__main__":
    a = [x for x in sys.argv[1:] if not x.startswith("--")]
    j = [x[7:] for x in sys.argv[1:] if x.startswith("--json=")]
    q = [x[5:] for x in sys.argv[1:] if x.startswith("--sq=")]
    st = [int(x[8:]) for x in sys.argv[1:] if x.startswith("--steps=")]
    main(a[0], a[1] if len(a) > 1 else None, j[0] if j else None, q[0] if q else None, st[0] if st else 0)
