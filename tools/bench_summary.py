#!/usr/bin/env python3
"""the fields of a bench.py JSON line one looks at first.  Usage: bench_summary.py file.json"""
import json
import sys

d = json.load(open(sys.argv[1]))
for k in ("value", "ms_per_step", "design_point_s", "design_point_modes_per_s", "extras0", "scaling_model", "scaling_model_c5"):
    print(k, d.get(k))
print("roofline", {k: d["roofline"][k] for k in ("frac", "us_per_launch", "traffic")})
print("spmv", d["spmv"]["frac"], "spmm", d["spmm"]["frac"])
if d.get("numpy_api"):
    print("numpy", d["numpy_api"]["ms_per_step"], d["numpy_api"].get("host_twins"))
if d.get("cpu_baseline"):
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("per_mode_gpu_vs_cpu"))
print("eig", d["eigensolver"], d["preamble_s"])
print("accuracy", d["accuracy"])
