#!/usr/bin/env python3
"""Fill the @@…@@ placeholders of DESIGN.md from a bench.py JSON line and the PMC traffic file."""
import json, sys
b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = b["roofline"]
t = json.load(open(sys.argv[2]))["all_gemm"]["hbm_bytes_per_launch"] if len(sys.argv) > 2 else None
s = open(sys.argv[3] if len(sys.argv) > 3 else "DESIGN.md").read()
rep = {"@@PAIRS@@": f"{b['value']:.0f}", "@@MS@@": f"{b['ms_per_step']:.1f}", "@@GEMMTF@@": f"{r['achieved']:.0f}",
       "@@GEMMFRAC@@": f"{100 * r['frac']:.1f} %", "@@GEMMMS@@": f"{r['kernel_ms_per_step']:.1f}",
       "@@GEMMTFEV@@": f"{r.get('achieved_hip_events', 0):.0f}", "@@TRAFFIC@@": f"{t / 1e6:.0f}" if t else "n/a",
       "@@STEPFRAC@@": f"{100 * r['step_frac_of_peak']:.1f} %", "@@STEPTF@@": f"{r['step_algorithmic_tflops_per_gpu']:.0f}"}
for k, v in rep.items():
    s = s.replace(k, v)
open("DESIGN.md", "w").write(s)
