"""table of (kernel, VGPRs, scratch, occupancy, LDS) from `make -C thaler-study_amd/csrc resource-usage` output on stdin"""
import re
import subprocess
import sys

txt = sys.stdin.read()
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for b in txt.split("remark: Function Name: ")[1:]:
    name = b.split(" ")[0]
    d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if pat and not re.search(pat, d):
        continue
    g = lambda k: re.search(re.escape(k) + r": (\d+)", b)
    vals = [g(k) for k in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]")]
    print("%-120s vgpr %4s agpr %4s scratch %5s occ %2s lds %6s" % tuple([re.sub(r"\(.*", "", d)[:120]] + [v.group(1) if v else "-" for v in vals]))
