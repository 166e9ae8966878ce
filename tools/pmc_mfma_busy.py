"""MFMA-busy fraction per kernel from one rocprofv3 --pmc pass with GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES.
    python tools/pmc_mfma_busy.py out.json pmc_dir "<command that was profiled>"
GRBM_GUI_ACTIVE is summed over the 8 XCDs: cycles = value / 8.  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs)
(the counter counts cycles per SIMD with the matrix pipe busy: MI355X_MICROARCH.md, cycle constants table)."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

out, d, cmd = sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else ""
per = defaultdict(lambda: defaultdict(float))
meta = {}
for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            meta[r["Dispatch_Id"]] = (r["Kernel_Name"], r["Grid_Size"])
acc = defaultdict(lambda: [0, 0.0, 0.0])
for disp, c in per.items():
    name, grid = meta[disp]
    short = re.sub(r"^void ", "", name)
    short = re.sub(r"\(anonymous namespace\)::", "", short).split("(")[0]
    a = acc[f"{short} grid={grid}"]
    a[0] += 1
    a[1] += c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    a[2] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
rows = [{"kernel": k, "launches": n, "gpu_cycles": round(cyc / n), "mfma_busy": round(busy / (cyc * 1024.0), 4) if cyc else None}
        for k, (n, cyc, busy) in sorted(acc.items(), key=lambda kv: -kv[1][1]) if n]
json.dump({"command": cmd, "note": "GRBM_GUI_ACTIVE is summed over the 8 XCDs; cycles = value / 8.  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / "
           "(cycles * 1024 SIMDs), per kernel over all its launches", "kernels": rows[:40]}, open(out, "w"), indent=1)
for r in rows[:16]:
    print(f"{r['kernel'][:80]:80s} launches {r['launches']:5d} cycles {r['gpu_cycles']:9d} mfma_busy {r['mfma_busy']}")
