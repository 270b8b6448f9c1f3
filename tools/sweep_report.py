"""Compare the per-launch GEMM tables of an in-situ tile sweep (tools/sweep_tiles.sh): per distinct launch signature,
the planner's choice and time against every forced tile shape. usage: sweep_report.py <tag> [min_gain_us]"""
import csv, sys, collections
tag = sys.argv[1]; thr = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
STEPS = 3  # bench.py dumps 3 stamped steps
cfgs = ["auto", "4", "3", "2", "2_2", "2_1"]
rows = {c: list(csv.DictReader(open(f"gpurun_out/{tag}_sweep_{c}.csv"))) for c in cfgs}
n = len(rows["auto"])
assert all(len(rows[c]) == n for c in cfgs), {c: len(rows[c]) for c in cfgs}
agg = collections.OrderedDict()
for i in range(n):
    a = rows["auto"][i]
    key = (a["M"], a["N"], a["K"], a["a_kmajor"], a["b_kmajor"], a["gather"], a["epilogue"])
    e = agg.setdefault(key, {"cnt": 0, "plan": a["tile"] + "/k" + a["ksplit"], **{c: 0.0 for c in cfgs}, **{c + "_plan": "" for c in cfgs}})
    e["cnt"] += 1
    for c in cfgs:
        e[c] += float(rows[c][i]["us"]); e[c + "_plan"] = rows[c][i]["tile"] + "/k" + rows[c][i]["ksplit"]
tot_auto = tot_best = 0.0
print(f"{'M,N,K,akm,bkm,gather,epi':46s} cnt  auto(plan)            " + " ".join(f"{c:>8s}" for c in cfgs[1:]) + "   gain/step")
for key, e in agg.items():
    best = min(cfgs, key=lambda c: e[c])
    gain = (e["auto"] - e[best]) / STEPS
    tot_auto += e["auto"] / STEPS; tot_best += e[best] / STEPS
    if gain >= thr:
        print(f"{','.join(key):46s} {e['cnt']//STEPS:3d} {e['auto']/e['cnt']:7.1f} {e['plan']:12s} " +
              " ".join(f"{e[c]/e['cnt']:8.1f}" for c in cfgs[1:]) + f"   {gain:7.1f} -> {best} {e[best+'_plan']}")
print(f"total auto {tot_auto/1e3:.3f} ms/step, per-signature best {tot_best/1e3:.3f} ms/step")
