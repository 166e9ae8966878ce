"""bench.py prints ONE JSON line with the keys the driver reads (short run: 2 steps, tiny secondaries)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_bench_json_contract():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "64", "--max-batch", "64",
           "--search-nq", "1024", "--search-nr", "50000", "--search-steps", "1", "--swin-batch", "8", "--no-cpu-baseline", "--ensemble-videos", "2"]
    # (8-frame Swin chunks would keep the GEMM launches in the 512-wide stage: VSC_SWIN_MLP512=1 takes the fused kernel as the full-size
    # bench does, so that the line's accounting of that kernel -- the qkv Linear inside it -- is what is checked here)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, VSC_SWIN_MLP512="1"))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["unit"] == "frames/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "bf16" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert "gfx950" in d["library"] and " src " in d["library"]      # the line says which build it measured (vsc_version())
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert d["value"] > 0 and abs(d["value"] - 64 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 0.01
    assert d["swin"]["value"] > 0 and d["search"]["value"] > 0 and d["search"]["roofline"]["bound"] == "mfma"
    # Swin: per-kernel-class HIP-event times and the roofline of its GEMM launches
    sw = d["swin"]
    assert sw["roofline"]["bound"] == "mfma" and sw["roofline"]["achieved"] > 0 and "s3.fc1" in sw["kernels"] and "s2.fc1" not in sw["kernels"] and "s3.proj_ln" in sw["kernels"] and "s0.proj_ln" not in sw["kernels"] and "s2.proj_ln" not in sw["kernels"]   # (stages 0-2: proj + LayerNorm + MLP in one kernel, booked under fc2_ln)
    # (stage 2: only the first block launches a qkv GEMM, the other 17 are computed by the previous block's kernel; two chunks per step)
    assert sw["kernels"]["s2.qkv"]["launches_per_step"] == 2 and sw["kernels"]["s2.qkv"]["tflops"] > 0 and sw["kernels"]["s1.qkv"]["launches_per_step"] == 4
    assert sw["roofline"]["algorithmic_bytes_per_launch"] > 0 and sw["kernels"]["s2.fc2_ln"]["algorithmic_bytes_per_launch"] > 0
    # ViT attention: priced against the HBM roof (its bytes), next to the MFMA numbers
    att = d["kernels"]["attention"]
    assert att["bound"] == "hbm" and 0 < att["frac_hbm"] < 1 and att["tb_per_s"] > 0 and att["tflops"] > 0
    # matching-track networks: every layer priced against the pipe that runs it
    mt = d["matching"]
    assert mt["classifier"]["maps_per_s"] > 0 and mt["refiner"]["maps_per_s"] > 0 and 0 < mt["refiner"]["frac_of_f32_mfma_peak"] < 1
    assert 0.3 < mt["refiner"]["flop_share_on_split_bf16_pipe"] < 1 and 0 < mt["refiner"]["frac_of_per_pipe_roof"] < 1
    # the reference's real workload end to end (3 x Swin-V2-B + vit_v68 + CLIP gate from uint8 host frames)
    en = d["ensemble"]
    assert en["operands"] == "fp16" and en["value_bf16_operands"] > 0      # measured at the entry points' default operand type, bf16 beside it
    f16 = d["fp16_operands"]                                                 # the ViT / Swin steps through libvsc_hip_f16.so
    assert "error" not in f16 and f16["vit"]["value"] > 0 and f16["swin"]["value"] > 0 and f16["vit"]["gemm_tflops"] > 0
    assert en["value"] > 0 and en["encoder_bound_frames_per_s"] > en["value"] and set(en["models_frames_per_s"]) == {"swinv2_base_256", "vit_v68", "clip_vit_l14_224"}


def test_bench_stdout_is_one_line_with_a_process_group():
    """The N > 1 launch initialises RCCL, which prints a version banner on the C stdout at exit: stdout must still
    hold the JSON line and nothing else (exercised with world size 1 and the sharded-search leg forced on)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-sharded-search", "--steps", "2", "--warmup", "1",
           "--batch", "64", "--max-batch", "64", "--search-nq", "1024", "--search-nr", "50000", "--search-steps", "1",
           "--no-swin", "--no-matching", "--no-cpu-baseline", "--no-ensemble"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["search"]["value"] > 0
    # the N > 1 search leg ran on a real RCCL communicator: bank all_gather (all_gather_into_tensor) + local sweep
    assert d["search"]["n_gpus"] == 1 and d["search"]["scaling"] == "weak" and d["search"]["all_gather_bytes_per_rank"] == 0
    assert "all_gather" in d["search"]["metric"] and d["search"]["nr_total"] == 50000
    # configs[3]: score normalisation is in the sharded leg, and the pipelined (shard-by-shard) gather is timed beside the one-gather form
    assert "score normalisation" in d["search"]["metric"] and d["search"]["ms_pipelined"] > 0 and d["search"]["ms_one_gather"] > 0


def test_bench_runs_with_world_size_two_on_one_gpu():
    """`bench.py --gpus 2` under torch.distributed.run with BOTH ranks on the one GPU of the box (--share-device; RCCL refuses two ranks on
    one device, so the backend is gloo on device tensors): the N > 1 code path -- per-rank frame shards, barrier + max-over-ranks timing, the
    sharded score-normalised search in its pipelined and one-gather forms, rank 0 alone printing -- executes end to end.  A plumbing check: no
    scaling claim can be drawn from two ranks time-slicing one GPU (profiles/r06_two_ranks_one_gpu.txt holds a full-size run)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29577",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "64", "--max-batch", "64",
           "--search-nq", "2048", "--search-nr", "100000", "--search-steps", "1", "--no-cpu-baseline", "--share-device", "--backend", "gloo"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and "shared_device_plumbing_run" in d
    assert abs(d["value"] - 2 * 64 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 0.01          # whole-job frames over the max-over-ranks time
    s = d["search"]
    assert s["n_gpus"] == 2 and s["nr_total"] == 100000 and s["ms_pipelined"] > 0 and s["ms_one_gather"] > 0 and s["form"].startswith("pipelined")
    assert s["all_gather_bytes_per_rank"] == 50000 * 513 * 4
    assert "swin" not in d and "fp16_operands" not in d          # secondaries are N = 1 only
