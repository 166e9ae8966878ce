"""Mean PMC counter values per kernel launch from rocprofv3 --pmc CSV output (one directory per counter pass).
python tools/pmc_summarize.py out.json dir1 [dir2 ...]
Kernels are keyed by their short name (template arguments kept) and grid size, as in profiles/r01_pmc_per_launch*.json."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

out, dirs = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(list))
for d in dirs:
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per_dispatch = defaultdict(float)
        meta = {}
        with open(path) as f:
            for r in csv.DictReader(f):
                key = (r["Dispatch_Id"], r["Counter_Name"])
                per_dispatch[key] += float(r["Counter_Value"])      # counters come per XCD / instance: sum them
                meta[r["Dispatch_Id"]] = (r["Kernel_Name"], r["Grid_Size"])
        for (disp, counter), v in per_dispatch.items():
            name, grid = meta[disp]
            short = re.sub(r"^void ", "", name)
            short = re.sub(r"\(anonymous namespace\)::", "", short).split("(")[0]
            acc[f"{short} grid={grid}"][counter].append(v)
res = {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"launches": max(len(v) for v in cs.values())} for k, cs in acc.items()}
json.dump({"per_launch": res}, open(out, "w"), indent=1, sort_keys=True)
for k, cs in sorted(res.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0))[:14]:
    print(f"{k[:90]:90s} " + " ".join(f"{c}={v:.1f}" for c, v in cs.items()))
