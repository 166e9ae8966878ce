#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root): the default bench line, the rocprofv3 kernel-trace
# summary of the same command at 3 steps, and separate --pmc passes (FETCH_SIZE / WRITE_SIZE) for the encoder GEMMs and the
# similarity sweep.  Outputs under gpurun_out/<tag>/; copy what is to be judged into profiles/.
set -u
TAG=${1:-r06}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# --lanes 1: under the profiler the two chunks of a step run back to back, so a kernel's average duration is its own (with the
# default two lanes the launches of the two chunks overlap and stretch each other: rocprofv3 then reports 300 us where the
# kernel alone takes 207) -- the same serialisation bench.py applies to its per-launch event loop
SHORT="--steps 3 --warmup 1 --profile-steps 3 --no-cpu-baseline --no-search --no-swin --no-matching --no-ensemble --no-fp16 --lanes 1"   # the ViT step and nothing else: every row of the summary is a ViT kernel
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
tail -c 600 "$OUT/bench_default.json"
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o enc -- python $OLDPWD/bench.py $SHORT > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err")
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/stats_knn" -o knn -- python $OLDPWD/tools/knn_bench.py 65536 1000000 100 2 > "$OUT/knn_under_rocprof.txt" 2> "$OUT/stats_knn.err")
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/stats_swin" -o swin -- python $OLDPWD/tools/swin_bench.py 256 3 256 > "$OUT/swin_under_rocprof.txt" 2> "$OUT/stats_swin.err")
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/stats_cnn" -o cnn -- python $OLDPWD/tools/cnn_bench.py > "$OUT/cnn_under_rocprof.txt" 2> "$OUT/stats_cnn.err")
# the reference's real workload end to end, with the synchronising breakdown (where the host-side time goes)
python tools/ensemble_bench.py 52 40 --breakdown > "$OUT/ensemble.json" 2> "$OUT/ensemble.err"
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --pmc $c -d "$OUT/pmc_$c" -o enc --output-format csv -- python $OLDPWD/bench.py $SHORT > /dev/null 2> "$OUT/pmc_$c.err")
  (cd /tmp && timeout 600 rocprofv3 --pmc $c -d "$OUT/pmcknn_$c" -o knn --output-format csv -- python $OLDPWD/tools/knn_bench.py 8192 1000000 100 1 > /dev/null 2> "$OUT/pmcknn_$c.err")
done
# Swin: memory-side bytes per launch (swin.roofline.traffic of the bench line) -- the same command as the kernel-trace pass
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $c -d "$OUT/pmcswin_$c" -o swin --output-format csv -- python $OLDPWD/tools/swin_bench.py 256 3 256 > /dev/null 2> "$OUT/pmcswin_$c.err")
done
python tools/pmc_summarize.py "$OUT/pmc_swin.json" "$OUT/pmcswin_FETCH_SIZE" "$OUT/pmcswin_WRITE_SIZE" > "$OUT/pmc_swin_summary.txt" 2>&1
# matrix-pipe busy share per kernel (attention / window attention / GEMMs): one pass of three SQ / GRBM counters each
MF="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
(cd /tmp && timeout 600 rocprofv3 --pmc $MF -d "$OUT/pmc_mfma" -o enc --output-format csv -- python $OLDPWD/bench.py $SHORT > /dev/null 2> "$OUT/pmc_mfma.err")
python tools/pmc_mfma_busy.py "$OUT/pmc_mfma_busy.json" "$OUT/pmc_mfma" "rocprofv3 --pmc $MF -- python bench.py $SHORT" > "$OUT/pmc_mfma_busy.txt" 2>&1
(cd /tmp && timeout 600 rocprofv3 --pmc $MF -d "$OUT/pmc_mfma_swin" -o swin --output-format csv -- python $OLDPWD/tools/swin_bench.py 256 3 256 > /dev/null 2> "$OUT/pmc_mfma_swin.err")
python tools/pmc_mfma_busy.py "$OUT/pmc_mfma_busy_swin.json" "$OUT/pmc_mfma_swin" "rocprofv3 --pmc $MF -- python tools/swin_bench.py 256 3 256" > "$OUT/pmc_mfma_busy_swin.txt" 2>&1
python tools/pmc_summarize.py "$OUT/pmc_per_launch.json" "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE" > "$OUT/pmc_summary.txt" 2>&1
python tools/pmc_summarize.py "$OUT/pmc_knn.json" "$OUT/pmcknn_FETCH_SIZE" "$OUT/pmcknn_WRITE_SIZE" > "$OUT/pmc_knn_summary.txt" 2>&1
find "$OUT" -name "*.db" | while read f; do python profiles/summarize_rocpd.py "$f" > "${f%.db}_summary.txt" 2>&1; done
find "$OUT" -name "*_kernel_stats.csv" -o -name "*summary.txt" | head
# keep the merge-back small: drop raw traces
find "$OUT" -name "*.db" -size +8M -delete
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
find "$OUT" -name "*counter_collection.csv" -size +8M -delete
du -sh "$OUT"
