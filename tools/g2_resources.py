#!/usr/bin/env python3
"""Per-instantiation register report of gemm2_kernel from `hipcc -Rpass-analysis=kernel-resource-usage` output.
usage: g2_resources.py build.log [filter]   (filter: substring of 'WM,NJ,A_KM,B_KM,GATHER,FP8,GEN')"""
import re, sys
t = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b in re.split(r"remark: Function Name: ", t)[1:]:
    name = b.split(" ")[0]
    m = re.match(r"_Z12gemm2_kernelILi(\d)ELi(\d)ELb(\d)ELb(\d)ELi(\d)ELb(\d)EL[bi](\d)E", name)
    if not m:
        continue
    key = ",".join(m.groups())
    if flt and flt not in key:
        continue
    d = dict(re.findall(r"remark:\s+([A-Za-z \[\]/]+): (\w+)", b))
    print(f"{key}  VGPR {d.get('VGPRs'):>3} scratch {d.get('ScratchSize [bytes/lane]'):>4} vspill {d.get('VGPRs Spill'):>3} "
          f"sgpr {d.get('TotalSGPRs'):>3} sspill {d.get('SGPRs Spill'):>3}")
