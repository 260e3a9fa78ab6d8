#!/usr/bin/env python3
"""
Sum a PMC counter per kernel family from a rocprofv3 --pmc run (counter_collection.csv), and -- with --json -- turn the
two separate passes (FETCH_SIZE, WRITE_SIZE) into the HBM bytes per launch that bench.py reports as ``roofline.traffic``.

  python tools/pmc_report.py DIR_OR_CSV                          # table per kernel family
  python tools/pmc_report.py --json profiles/r02_traffic.json --entry sweep_k32_c3 \\
      --fetch profiles/r02_pmc_fetch_sweep.csv --write profiles/r02_pmc_write_sweep.csv \\
      --kernels 'fwd_|bwd_|v1_assemble' --units 10 --sources factor.hip

Corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE (KiB) is doubled (128-byte requests
tallied at 64 bytes), WRITE_SIZE (KiB) is taken as is; the same runs hold a stream of known size (column dots) whose
ratio to its byte count is printed as the calibration.  ``--units``: launches of the unit the bench calls one launch
(10 sweeps of 32 columns in tools/pmc_sweep.py, 20 SpMVs in tools/pmc_spmv.py).  The JSON keeps the sha of the kernel
sources: bench.py reports the number only while those files are unchanged.
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import re
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find_csv(path):
    if os.path.isdir(path):
        return glob.glob(path + "/**/*counter_collection.csv", recursive=True)[0]
    return path


def family_sums(path):
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(find_csv(path))):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("eigd::", "")
        name = re.sub(r"<.*", "", name)
        tot[(name, r["Counter_Name"])] += float(r["Counter_Value"])
        cnt[(name, r["Counter_Name"])] += 1
    return tot, cnt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path", nargs="?")
    ap.add_argument("--json")
    ap.add_argument("--entry")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--kernels", default=".")
    ap.add_argument("--units", type=int, default=1)
    ap.add_argument("--sources", nargs="*", default=[])
    ap.add_argument("--calib-kernel", default="coldot")
    ap.add_argument("--calib-bytes", type=float, default=None, help="known bytes of ONE calibration launch")
    ap.add_argument("--mfma", default=None, help="pass with SQ_INSTS_VALU_MFMA_F64 and SQ_VALU_MFMA_BUSY_CYCLES over the same launches")
    args = ap.parse_args()
    if args.json is None:
        tot, cnt = family_sums(args.path)
        for (name, ctr), v in sorted(tot.items()):
            print(f"{name:28s} {ctr:12s} launches {cnt[(name, ctr)]:6d}  sum {v:16.1f}  per launch {v / cnt[(name, ctr)]:14.1f}")
        return
    pat = re.compile(args.kernels)
    ft, fc = family_sums(args.fetch)
    wt, _ = family_sums(args.write)
    fetch_kib = sum(v for (name, ctr), v in ft.items() if ctr == "FETCH_SIZE" and pat.search(name))
    write_kib = sum(v for (name, ctr), v in wt.items() if ctr == "WRITE_SIZE" and pat.search(name))
    traffic = (2.0 * fetch_kib + write_kib) * 1024.0 / args.units
    rec = {"traffic_bytes_per_launch": round(traffic), "fetch_size_kib_total": fetch_kib, "write_size_kib_total": write_kib,
           "units": args.units, "kernels": args.kernels, "rule": "2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes), gfx950",
           "files": [os.path.relpath(args.fetch, ROOT), os.path.relpath(args.write, ROOT)],
           "source_sha16": {fn: hashlib.sha256(open(os.path.join(ROOT, "eigd_amd", "csrc", fn), "rb").read()).hexdigest()[:16]
                            for fn in args.sources}}
    cal = [(name, v, fc[(name, ctr)]) for (name, ctr), v in ft.items() if ctr == "FETCH_SIZE" and args.calib_kernel in name]
    if cal and args.calib_bytes:
        kib = sum(v for _, v, _ in cal)
        launches = sum(c for _, _, c in cal)
        rec["calibration"] = {"kernel": args.calib_kernel, "launches": launches,
                              "ratio_2x_fetch_to_known_bytes": 2.0 * kib * 1024.0 / (launches * args.calib_bytes)}
    if args.mfma:
        mt, _ = family_sums(args.mfma)
        insts = sum(v for (name, ctr), v in mt.items() if ctr == "SQ_INSTS_VALU_MFMA_F64" and pat.search(name))
        busy = sum(v for (name, ctr), v in mt.items() if ctr == "SQ_VALU_MFMA_BUSY_CYCLES" and pat.search(name))
        rec["mfma_f64_insts_per_launch"] = insts / args.units            # wave instructions (v_mfma_f64_16x16x4: 2048 flop each)
        rec["mfma_busy_cycles_per_launch"] = busy / args.units
        rec["files"].append(os.path.relpath(args.mfma, ROOT))
    out = {}
    if os.path.exists(args.json):
        out = json.load(open(args.json))
    out[args.entry] = rec
    json.dump(out, open(args.json, "w"), indent=1, sort_keys=True)
    print(json.dumps({args.entry: rec}, indent=1))


if __name__ == "__main__":
    main()
