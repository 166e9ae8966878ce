#!/bin/bash
# end-to-end ensemble rate against the number of frames per group of videos (src.query_pipeline.run_query_videos group_frames): 208 videos x 40 frames
for g in 1024 2048 4096 8192; do
  python tools/ensemble_bench.py 208 40 --group=$g 2>/dev/null > /tmp/ens_$g.json
  python - $g <<'PY'
import json, sys
l = json.load(open(f"/tmp/ens_{sys.argv[1]}.json"))
print("group_frames", sys.argv[1], "frames/s", l["value"], "of encoder-bound", l["fraction_of_encoder_bound"], "ragged", l.get("ragged_lengths", {}).get("fraction_of_encoder_bound"), flush=True)
PY
done
