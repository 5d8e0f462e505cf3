#!/usr/bin/env python3
"""MFMA-busy per kernel from a rocprofv3 PMC pass over bench.py:

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d OUT -- python3 bench.py ...
    python3 tools/pmc_mfma.py OUT [steps]

MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per v_mfma_f32_32x32x16_bf16), summed over the SIMDs
that rocprofv3 samples; GRBM_GUI_ACTIVE is the sum over the 8 XCDs of the busy clocks.  MFMA-busy fraction of a kernel =
sum(MFMA_BUSY) / (256 CUs x 4 SIMDs x sum(GUI_ACTIVE) / 8).  Counter passes serialise the two streams of the step, so
these are isolated-kernel figures.  "CUs used" = min(256, workgroups) averaged over the kernel's dispatches and
"busy on them" = the busy fraction of THOSE CUs: what a kernel that deliberately runs on few CUs (gemm_tn_pp.hip beside the BPTT
chain) should be judged by -- its share of the chip is small by design.  Also printed: bf16 TFLOP/s implied by the busy cycles (16384 MAC-pairs... i.e.
2*32*32*16 flop per 32 busy cycles) against the 2.5 PFLOP/s dense peak.
"""
import collections
import csv
import glob
import re
import sys

d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(lambda: dict(mfma=0.0, gui=0.0, n=0, dur=0.0, cus=0.0))
seen = set()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        a = agg[k]
        v = float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
            a["mfma"] += v
        elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            a["gui"] += v
        key = (k, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            a["n"] += 1
            a["dur"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            # CUs a dispatch can occupy at most: one per workgroup up to the chip's 256 (a 56-workgroup launch of the 256 x 256
            # weight-gradient kernel cannot be busier than 22 % of the CHIP however well its CUs run), weighted by its duration
            try:
                wgs = max(1, int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))
            except (KeyError, ValueError):
                wgs = 256
            a["cus"] += min(256, wgs) * (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", n))
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9A-Z]+?)I", n)
    if m:
        return m.group(1)
    if "uic_gemm_kernel<" in n:
        return "uic_gemm_kernel<" + re.sub(r"\s+", "", n.split("uic_gemm_kernel<")[1].split(">")[0])[-24:] + ">"
    return n[:48]


rows = sorted(agg.items(), key=lambda kv: -kv[1]["dur"])
print("%-50s %8s %10s %9s %10s %9s %8s %12s" % ("kernel", "calls/st", "us/call", "ms/step", "MFMA busy", "TFLOP/s", "CUs used", "busy on them"))
for k, a in rows[:16]:
    if a["gui"] <= 0:
        continue
    frac = a["mfma"] / (1024.0 * a["gui"] / 8.0)
    tf = a["mfma"] / 32.0 * (2 * 32 * 32 * 16) / (a["dur"] * 1e-6) / 1e12 if a["dur"] else 0.0
    cus = a["cus"] / a["dur"] if a["dur"] else 256.0
    print("%-50s %8.1f %10.2f %9.3f %9.1f%% %9.1f %8.0f %11.1f%%" % (short(k), a["n"] / steps, a["dur"] / a["n"], a["dur"] / 1e3 / steps, 100 * frac, tf,
                                                                    cus, 100 * frac * 256.0 / cus))
