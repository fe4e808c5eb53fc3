"""Per-kernel register / scratch / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage 2> remarks.txt; python tools/kernel_resources.py remarks.txt [filter]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
K = {"vgpr": r"VGPRs: (\d+)", "agpr": r"AGPRs: (\d+)", "sgpr": r"SGPRs: (\d+)", "scratch": r"ScratchSize \[bytes/lane\]: (\d+)",
     "occ": r"Occupancy \[waves/SIMD\]: (\d+)", "lds": r"LDS Size \[bytes/block\]: (\d+)"}
for b in blocks:
    name = b.split("\n")[0].strip()
    try:
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    except Exception:
        pass
    if flt and flt not in name:
        continue
    vals = {k: (re.search(p, b).group(1) if re.search(p, b) else "?") for k, p in K.items()}
    print(f"{name[:100]:100s} " + " ".join(f"{k} {v:>5}" for k, v in vals.items()))
