#!/bin/bash
# per-launch durations of the narrow convolution kernel in one HRNet pass, persistent vs one tile per workgroup (GPU box)
export TMPDIR=/tmp
for m in 1 0; do
  (cd /tmp && VSC_CONV_PERSIST=$m rocprofv3 --kernel-trace --output-format csv -d /tmp/ct$m -o t -- python $GRAFT_REPO_ROOT/tools/cnn_bench.py hrnet > /dev/null 2>&1)
done
python - <<'PY'
import csv, glob
def load(m):
    f = glob.glob(f"/tmp/ct{m}/**/*kernel_trace.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "conv_gemm_narrow" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return [((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", "?")) for r in rows]
a, b = load(1), load(0)
print(len(a), len(b), "persistent total %.1f ms, one-tile total %.1f ms" % (sum(x[0] for x in a) / 1e3, sum(x[0] for x in b) / 1e3))
n = min(len(a), len(b))
worst = sorted(range(n), key=lambda i: a[i][0] - b[i][0], reverse=True)[:12]
for i in worst:
    print(i, "persistent %.1f us grid %s | one-tile %.1f us grid %s" % (a[i][0], a[i][1], b[i][0], b[i][1]))
PY
