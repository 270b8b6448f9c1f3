#!/usr/bin/env python3
"""Aggregate one rocprofv3 SQ/GRBM --pmc pass (CSV) per kernel family: matrix-pipe busy fraction, LDS bank-conflict
fraction, wave wait buckets. Units per /opt/skills/guides/MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles
summed over the SIMDs, GRBM_GUI_ACTIVE the kernel's wall cycles; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles (their RATIOS are what is reported); SQ_LDS_BANK_CONFLICT = extra LDS cycles, SQ_LDS_IDX_ACTIVE = all LDS-array
cycles.  Calibration on this box: SQ_VALU_MFMA_BUSY_CYCLES is the device-wide total (16 cycles per
v_mfma_f32_16x16x32_bf16: the grouped weight-gradient launch reads exactly 7.77 M MFMAs x 16) and GRBM_GUI_ACTIVE is summed
over the 8 XCDs (3.34 M for 180 us = 8 x 2.32 GHz), so  mfma_busy = MFMA_BUSY / (GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs).
usage: pmc_sq.py counter_collection.csv out.json"""
import csv, json, sys, collections, re

def family(name):
    n = name.split("(")[0].replace("void ", "")
    if "gemm2_kernel" in n:
        return "gemm2_kernel" + n[n.index("<"):] if "<" in n else n
    if "gemm_stream" in n:  # the streaming convolution kernels (gemm_stream.hip): one family per instantiation
        return n[n.index("gemm_stream"):]
    n = re.sub(r"<.*", "", n)
    m = re.match(r"_Z\d+([a-z_0-9]+?)(I|P|E|v).*", n)
    return m.group(1) if m else n

per = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
dur = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    k = family(r["Kernel_Name"])
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in cnt[k]:
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3  # us
    cnt[k].add(r["Dispatch_Id"])
rows = []
for k, c in per.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    if gui <= 0:
        continue
    wave = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    lds = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
    rows.append({"kernel": k, "launches": len(cnt[k]), "gui_cycles": gui,
                 "mfma_busy": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8.0 * 256 * 4),
                 "clock_ghz": (gui / 8.0) / (dur[k] * 1e3) if dur[k] > 0 else None,
                 "lds_conflict_frac": (c.get("SQ_LDS_BANK_CONFLICT", 0.0) / lds) if lds > 0 else None,
                 "wait_any_frac": c.get("SQ_WAIT_ANY", 0.0) / wave, "wait_inst_frac": c.get("SQ_WAIT_INST_ANY", 0.0) / wave,
                 "active_inst_frac": c.get("SQ_ACTIVE_INST_ANY", 0.0) / wave})
rows.sort(key=lambda r: -r["gui_cycles"])
tot = sum(r["gui_cycles"] for r in rows)
g = [r for r in rows if r["kernel"].startswith("gemm2_kernel") or r["kernel"].startswith("gemm_stream")]
gt = sum(r["gui_cycles"] for r in g)
out = {"note": __doc__.split("usage")[0].strip(), "kernels": rows[:40],
       "all_gemm2": {"share_of_gpu_cycles": gt / tot if tot else None,
                     "mfma_busy": sum(r["mfma_busy"] * r["gui_cycles"] for r in g) / gt if gt else None,
                     "lds_conflict_frac": sum((r["lds_conflict_frac"] or 0) * r["gui_cycles"] for r in g) / gt if gt else None}}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out["all_gemm2"]))
for r in rows[:12]:
    print(f"{r['kernel'][:60]:60s} n={r['launches']:4d} mfma_busy={r['mfma_busy']:.3f} lds_conf={r['lds_conflict_frac'] if r['lds_conflict_frac'] is None else round(r['lds_conflict_frac'],3)} wait={r['wait_any_frac']:.2f} stall={r['wait_inst_frac']:.2f} active={r['active_inst_frac']:.2f}")
