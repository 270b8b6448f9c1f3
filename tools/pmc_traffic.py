#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc passes (FETCH_SIZE pass, WRITE_SIZE pass; CSV output) into HBM bytes per launch for the
MFMA GEMM kernels. gfx950 corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE reports half
of the bytes of wide coalesced reads -> doubled; WRITE_SIZE is exact for 16-byte-per-lane stores. Both counters are in
KiB in rocprofv3's derived-metric definition.
usage: pmc_traffic.py FETCH_counter_collection.csv WRITE_counter_collection.csv out.json"""
import csv, json, sys, collections

def agg(path, counter):
    per = collections.defaultdict(lambda: [0, 0.0])
    seen = set()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        if "gemm2_kernel" not in name and "gemm_bf16_kernel" not in name and "gemm_stream" not in name:
            continue
        key = (r["Dispatch_Id"], counter)
        if key in seen:
            continue
        seen.add(key)
        k = name.split("(")[0].replace("void ", "")
        per[k][0] += 1
        per[k][1] += float(r["Counter_Value"])
    return per

fetch = agg(sys.argv[1], "FETCH_SIZE")
write = agg(sys.argv[2], "WRITE_SIZE")
out = {"unit": "bytes per launch", "correction": "FETCH_SIZE x2 (gfx950), KiB -> bytes", "kernels": {}}
tot_n = tot_b = 0
for k in sorted(set(fetch) | set(write)):
    n = max(fetch[k][0], write[k][0])
    fb = 2.0 * 1024.0 * fetch[k][1] / max(fetch[k][0], 1)
    wb = 1024.0 * write[k][1] / max(write[k][0], 1)
    out["kernels"][k] = {"launches": n, "fetch_bytes": round(fb), "write_bytes": round(wb), "hbm_bytes": round(fb + wb)}
    tot_n += n
    tot_b += n * (fb + wb)
out["all_gemm"] = {"launches": tot_n, "hbm_bytes_per_launch": round(tot_b / max(tot_n, 1))}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["all_gemm"]))
