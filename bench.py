#!/usr/bin/env python
"""bench.py -- throughput of the descriptor-track hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Primary workload (BASELINE.json configs[1]): ViT-B/16 224x224 bf16 encode, synthetic
frames resident in HBM, random-init weights.  One step = one batch of --batch frames
per GPU through vsc_encoder_forward (patchify .. L2-normalised 512-d descriptors).
Frames shard across ranks with no data-path collective ("weak" scaling: per-GPU work
fixed); value = frames of all ranks / max-over-ranks time.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      dominant kernel = the bf16 GEMM (gemm_bf16_v4_kernel: qkv/proj/fc1/fc2; _v3_ for the patch GEMM);
                achieved = algorithmic FLOPs of those launches / their HIP-event time (events on the launch
                stream, in a separate loop of --profile-steps steps right after the timed region: the headline
                loop carries no events); peak = 2500 TFLOP/s dense bf16 MFMA.
  cpu_baseline  the fp32 oracle (oracle/vit_oracle.py, a port) on the host cores, on a
                bounded sample of the same frames.
  search        secondary metric (BASELINE.json configs[2]): exact 512-d inner-product top-100 over 1M x 1M pairs
                (vsc_knn_ip_f32: bf16 pre-filter sweep + exact fp32 re-scoring), Mpairs/s, with the sweep kernel's
                own roofline (HIP events around the phases of the call) and its PMC traffic.  With N > 1 ranks: the sharded form
                (each rank holds nr / N references, one RCCL all_gather assembles the bank, every rank sweeps its
                own nq queries; weak scaling, all_gather inside the timed region).
  swin          secondary metric: Swin-V2-B 256 encode (vsc_swin_forward), frames/s, with per-kernel-class HIP-event times
                (kernels{}) and the roofline of its GEMM launches (gemm_ln_kernel + gemm_bf16_v4 / v3 / v2).
  matching      secondary: the matching track's fp32 networks (pair classifier, HRNet refinement net), maps/s and the
                fraction of the fp32 MFMA peak.
  fp16_operands secondary (round 6): the same ViT step and the Swin-V2-B step through libvsc_hip_f16.so -- the build whose MFMA
                operands are IEEE fp16 instead of bf16 (same kernels; what the infer/ entry points default to because the end-to-end
                uAP parity needs it).  `value` above stays the bf16 configuration BASELINE.json names.
  ensemble      secondary: the reference's whole query-side workload from uint8 host frames, at the entry points' default operand
                type (fp16), `value_bf16_operands` beside it.
  search.cpu_baseline.oracle_check_of_the_timed_result: rows of the timed 1M x 1M result compared with oracle/knn_oracle.c, bit for bit.
With --share-device --backend gloo the N > 1 path runs with several ranks on one GPU (plumbing check; RCCL refuses that).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "vsc22-submission_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch

BF16_PEAK_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md), quoted at the 2.4 GHz peak clock
F32_MFMA_PEAK_TFLOPS = 157.3
HBM_PEAK_TBS = 8.0          # HBM3E (MI355X_MICROARCH.md: 8 TB/s peak, ~6.3 achievable by a copy)
PEAK_SCLK_MHZ = 2400.0


class ClockSampler:
    """rocm-smi shader clock / package power sampled in a thread while the timed region runs (rank 0).
    The bf16 GEMMs are power-limited on this part (profiles/r01_clock_power_probe.txt): the sustained
    clock, not 2.4 GHz, sets the matrix roof the kernels actually run under."""

    def __init__(self, period=0.25):
        import threading
        self.period, self.samples, self._stop = period, [], False
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        import re
        import subprocess
        while not self._stop:
            try:
                out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True,
                                     timeout=5).stdout.strip().splitlines()
                hdr, row = out[0].split(","), out[-1].split(",")
                sclk = float(re.sub(r"[^0-9.]", "", row[hdr.index("sclk clock speed:")]))
                power = float(row[-1])
                self.samples.append((sclk, power))
            except Exception:  # noqa: BLE001 -- no rocm-smi / other layout: report nothing rather than fail the bench
                return
            time.sleep(self.period)

    def __enter__(self):
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        self._thread.join(timeout=10)

    def summary(self):
        if not self.samples:
            return None
        s = self.samples[1:] or self.samples   # the first sample can predate the load
        sclk = sum(x[0] for x in s) / len(s)
        return {"sclk_mhz": round(sclk), "package_power_w": round(sum(x[1] for x in s) / len(s)), "samples": len(s)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=151,
                    help="timed steps; the default 151 x 664 = 100 264 frames is BASELINE.json configs[1] (100k frames)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=664, help="frames per step per GPU")
    ap.add_argument("--max-batch", type=int, default=332,
                    help="frames per internal encoder chunk (332 x 197 tokens = 255.5 -> 256 GEMM row "
                         "tiles: every GEMM grid is then a whole number of 256-CU rounds)")
    ap.add_argument("--lanes", type=int, default=2,
                    help="internal streams the chunks of a step alternate over (the encoder's default: 2 -- the two 332-frame "
                         "chunks of a step are independent, and on two HIP streams the tail of one chunk's kernel overlaps the "
                         "head of the other's: +5 %% frames/s over 1; the per-launch event loop behind kernels{} / roofline "
                         "always runs the chunks back to back on one stream)")
    ap.add_argument("--fuse-ln", action="store_true", help="LayerNorm folded into the neighbouring GEMM epilogues (DESIGN 4.1b)")
    ap.add_argument("--preset", default="vit_b16_224")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-search", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=8,
                    help="steps of the separate loop that brackets every launch with HIP events (kernels{} / roofline); "
                         "the headline loop runs without them")
    ap.add_argument("--search-nq", type=int, default=1_000_000, help="BASELINE.json configs[2]: 1M queries x 1M refs")
    ap.add_argument("--search-nr", type=int, default=1_000_000)
    ap.add_argument("--search-k", type=int, default=100)
    ap.add_argument("--search-steps", type=int, default=1)
    ap.add_argument("--no-swin", action="store_true")
    ap.add_argument("--no-fp16", action="store_true", help="skip the fp16-operand secondary (libvsc_hip_f16.so)")
    ap.add_argument("--no-matching", action="store_true")
    ap.add_argument("--no-ensemble", action="store_true")
    ap.add_argument("--ensemble-videos", type=int, default=208, help="query videos (40 frames each; then once more with 10 .. 70 frames each: ensemble.ragged_lengths) of the end-to-end ensemble secondary")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend of the N > 1 launch (nccl = RCCL; gloo only for the "
                    "--share-device plumbing run)")
    ap.add_argument("--share-device", action="store_true",
                    help="plumbing run of the N > 1 code on a box with fewer GPUs than ranks: rank r uses device r %% device_count "
                         "(RCCL refuses two ranks on one device, so this needs --backend gloo); no scaling claim can be drawn from it")
    ap.add_argument("--force-sharded-search", action="store_true",
                    help="run the N > 1 search leg (RCCL all_gather + sweep) even with one rank; needs torchrun's env")
    ap.add_argument("--swin-batch", type=int, default=256)
    return ap.parse_args()


def gemm_flops_per_frame(cfg):
    t, d, m = cfg.tokens, cfg.width, cfg.mlp_dim
    return {
        "gemm_patch": 2 * (t - 1) * cfg.patch_dim * d,
        "gemm_qkv": cfg.layers * 2 * t * d * 3 * d,
        "gemm_proj": cfg.layers * 2 * t * d * d,
        "gemm_fc1": cfg.layers * 2 * t * d * m,
        "gemm_fc2": cfg.layers * 2 * t * d * m,
    }


def gemm_traffic(cfg, chunk):
    """(measured HBM bytes per GEMM launch from the committed PMC profile, algorithmic bytes per launch).

    PMC counters need rocprofv3, so they are not collected live: profiles/r02_pmc_per_launch.json (v4 / v3 kernels; r01_pmc_per_launch_v2.json for v2) holds
    FETCH_SIZE / WRITE_SIZE per launch of this same configuration (separate --pmc passes).  Correction as
    /opt/skills/guides/MI355X_MICROARCH.md (HBM) prescribes for gfx950: FETCH_SIZE reports half of a 16-byte-per-lane
    stream, so reads = 2 x FETCH_SIZE -- confirmed in the same run on kernels with known byte counts (layernorm:
    196,212 KiB read, FETCH_SIZE 98,163; attention: 294,318 KiB, 147,246); WRITE_SIZE needs none.  These are the L2's
    memory-side requests, Infinity-Cache hits included."""
    t, d, m = cfg.tokens, cfg.width, cfg.mlp_dim
    rows = chunk * t
    launches = {"0": cfg.layers, "1": cfg.layers, "3": 2 * cfg.layers, "4": 1}   # EPI class -> launches per chunk
    algo = {"0": rows * d * 2 + 3 * d * d * 2 + rows * 3 * d * 2,
            "1": rows * d * 2 + m * d * 2 + rows * m * 2,
            "3": ((rows * d * 2 + d * d * 2 + 2 * rows * d * 4) + (rows * m * 2 + m * d * 2 + 2 * rows * d * 4)) / 2,
            "4": chunk * (t - 1) * cfg.patch_dim * 2 + cfg.patch_dim * d * 2 + chunk * (t - 1) * d * 4}
    n = sum(launches.values())
    algorithmic = sum(launches[k] * algo[k] for k in launches) / n
    for name, prefix in (("r06_pmc_per_launch.json", ("gemm_bf16_v4_kernel<", "gemm_bf16_v3_kernel<")), ("r05_pmc_per_launch.json", ("gemm_bf16_v4_kernel<", "gemm_bf16_v3_kernel<")), ("r04_pmc_per_launch.json", ("gemm_bf16_v4_kernel<", "gemm_bf16_v3_kernel<")), ("r03_pmc_per_launch.json", ("gemm_bf16_v4_kernel<", "gemm_bf16_v3_kernel<")), ("r02_pmc_per_launch.json", ("gemm_bf16_v4_kernel<", "gemm_bf16_v3_kernel<")),
                         ("r01_pmc_per_launch_v2.json", ("gemm_bf16_v2_kernel<",))):
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", name)))["per_launch"]
            meas = 0.0
            seen = 0
            for key, v in prof.items():
                if any(pf in key for pf in prefix):
                    epi = key.split("<")[1].split(",")[0].split(">")[0]
                    meas += launches.get(epi, 0) * (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
                    seen += launches.get(epi, 0)
            if seen == n:
                return meas / n, algorithmic, name
        except (OSError, KeyError, ValueError):
            pass
    return None, algorithmic, None


def cpu_baseline(cfg, weights, frames_np, budget_s=15.0):
    from oracle import vit_oracle
    # threads this process may really use: affinity mask, capped by the cgroup CPU quota
    ncpu = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            ncpu = max(1, min(ncpu, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    torch.set_num_threads(ncpu)
    w = {k: torch.from_numpy(v) for k, v in weights.items()}
    x = torch.from_numpy(frames_np)
    with torch.no_grad():
        vit_oracle.descriptors(w, cfg, x[:2])  # warm the thread pool / allocator
        t0 = time.perf_counter()
        vit_oracle.descriptors(w, cfg, x[:4])
        per4 = time.perf_counter() - t0
        n = int(max(4, 4 * (budget_s / max(per4, 1e-3)) // 4 * 4))  # ~budget_s of CPU work
        t0 = time.perf_counter()
        for i in range(0, n, 4):
            j = i % (len(x) - 3)
            vit_oracle.descriptors(w, cfg, x[j:j + 4])
        dt = time.perf_counter() - t0
    return {"value": round(n / dt, 3), "unit": "frames/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"{n} frames (bench frames, cycled), fp32 torch oracle, batches of 4, {dt:.1f} s"}


def attention_mfma_busy(kernel_prefix, path_hint=""):
    """MFMA-busy fraction of a kernel family's launches from the newest committed PMC pass (tools/pmc_mfma_busy.py: SQ_VALU_MFMA_BUSY_CYCLES /
    (GPU cycles x 1024 SIMDs); PMC needs rocprofv3, so it is not collected live).  Every kernel whose name starts with `kernel_prefix` counts
    (e.g. "window_attention": window_attention_stream_kernel, ..._wide_stream_kernel, ..._kernel<NT>), weighted by its time.
    -> {"value", "source", "kernels"}; {"value": None, "reason"} when no committed pass holds such a row (never the nearest other kernel)."""
    tried = []
    for name in (f"r06_pmc_mfma_busy{path_hint}.json", f"r05_pmc_mfma_busy{path_hint}.json", f"r04_pmc_mfma_busy{path_hint}.json"):
        tried.append(name)
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", name)))
            rows = [r for r in prof["kernels"] if r["kernel"].startswith(kernel_prefix) and r.get("mfma_busy") is not None]
            if rows:
                cyc = sum(r["gpu_cycles"] * r["launches"] for r in rows)
                return {"value": round(sum(r["mfma_busy"] * r["gpu_cycles"] * r["launches"] for r in rows) / cyc, 4),
                        "source": f"profiles/{name}", "kernels": sorted({r["kernel"].split("(")[0][:60] for r in rows}),
                        "launches": int(sum(r["launches"] for r in rows))}
        except (OSError, ValueError, KeyError, TypeError, ZeroDivisionError):
            pass
    return {"value": None, "reason": f"no kernel named {kernel_prefix}* in profiles/{{{', '.join(tried)}}}"}


def search_cpu_baseline(budget_s=12.0, check=None):
    """The search leg's CPU baseline: oracle/knn_oracle.c (the restated faiss Flat index, OpenMP over queries) on the host
    threads -- BASELINE.json configs[0]'s plumbing case (64 x 1k, top-10) and a bounded sample of configs[2]
    (n queries x 1M refs x 512, top-100, n sized to ~budget_s).  check = (q_rows, r, D_rows, I_rows, k) on the host: rows of the
    timed GPU result, compared with the oracle here (the oracle as the checker, outside every timed region)."""
    from oracle import knn_oracle
    verdict = None
    if check is not None:
        qh, rh, Dg, Ig, kk = check
        Dr, Ir = knn_oracle.knn_ip(qh, rh, kk)
        verdict = {"rows": int(len(qh)), "ids_equal": bool(np.array_equal(Ig, Ir)), "scores_bit_equal": bool(np.array_equal(Dg.view(np.uint32), Dr.view(np.uint32)))}
        del rh
    ncpu = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            ncpu = max(1, min(ncpu, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    os.environ["OMP_NUM_THREADS"] = str(ncpu)   # read by libgomp when the oracle library is first loaded
    rng = np.random.default_rng(3)
    d = 512
    r = rng.standard_normal((1000000, d), dtype=np.float32)
    r /= np.linalg.norm(r, axis=1, keepdims=True)
    q = rng.standard_normal((4096, d), dtype=np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    knn_oracle.knn_ip(q[:8], r[:1000], 10)          # load + warm
    t0 = time.perf_counter()
    reps = 200
    for _ in range(reps):
        knn_oracle.knn_ip(q[:64], r[:1000], 10)
    small = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    knn_oracle.knn_ip(q[:2 * ncpu], r, 100)
    probe = (time.perf_counter() - t0) / (2 * ncpu)
    n = int(min(len(q), max(2 * ncpu, budget_s / max(probe, 1e-6)) // ncpu * ncpu))
    t0 = time.perf_counter()
    knn_oracle.knn_ip(q[:n], r, 100)
    dt = time.perf_counter() - t0
    return {"oracle_check_of_the_timed_result": verdict,
            "value": round(n * 1e6 / dt / 1e6, 3), "unit": "Mpairs/s", "cores": ncpu, "kind": "port",
            "sample": f"{n} queries x 1,000,000 refs x 512, top-100, oracle/knn_oracle.c (fmaf chains, OpenMP over queries), {dt:.1f} s",
            "configs0_64x1k_top10": {"ms": round(small * 1e3, 3), "mpairs_per_s": round(64 * 1000 / small / 1e6, 2)}}


def search_traffic():
    """HBM-side bytes per sweep-kernel launch from the committed PMC profile (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate passes over tools/knn_bench.py; reads = 2 x FETCH_SIZE on gfx950, see gemm_traffic).  -> (bytes, nq, nr) of
    the profiled launch, or None."""
    try:
        name = next(n for n in ("r06_pmc_knn.json", "r05_pmc_knn.json", "r04_pmc_knn.json", "r03_pmc_knn.json", "r02_pmc_knn.json") if os.path.exists(os.path.join(ROOT, "profiles", n)))
        prof = json.load(open(os.path.join(ROOT, "profiles", name)))
        key = prof.get("sweep_kernel_key") or next(k for k in prof["per_launch"] if k.startswith("knn_sweep_bf16_kernel"))
        v = prof["per_launch"][key]
        # tools/profile_round.sh profiles `tools/knn_bench.py 8192 1000000 100 1`
        return (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024, prof.get("nq", 8192), prof.get("nr", 1000000), name
    except (OSError, KeyError, ValueError, StopIteration):
        return None


def bench_search(dev, args):
    import ctypes
    from vsc_hip import _lib, ops
    lib = _lib.require_device()
    d = 512
    nq, nr, k = args.search_nq, args.search_nr, args.search_k
    g = torch.Generator(device=dev).manual_seed(1)
    r = torch.randn(nr, d, generator=g, device=dev)
    q = torch.randn(nq, d, generator=g, device=dev)
    ops.l2_normalize_(r)
    ops.l2_normalize_(q)
    ops.knn_ip(q[:256], r[:4096], k)  # allocate + warm
    ops.knn_ip(q, r, k)
    torch.cuda.synchronize()
    lib.vsc_knn_set_profiling(1)
    t0 = time.perf_counter()
    for _ in range(args.search_steps):
        D, I = ops.knn_ip(q, r, k)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / args.search_steps
    phases = (ctypes.c_float * 4)()
    _lib.check(lib.vsc_knn_last_profile(phases))
    lib.vsc_knn_set_profiling(0)
    path = lib.vsc_knn_last_path()
    # cheap self-checks on the full-size result (the parity tests proper are tests/test_gpu_knn.py)
    assert bool((D[:, :-1] >= D[:, 1:]).all()) and int(I.min()) >= 0 and int(I.max()) < nr
    pairs = nq * nr
    sweep_ms = float(phases[1])
    if path == 1:   # exact fp32 MFMA sweep
        peak, kern, dtype = F32_MFMA_PEAK_TFLOPS, "knn_kernel (v_mfma_f32_32x32x2_f32)", "f32"
    else:           # bf16 pre-filter sweep (selection) + exact fp32 re-scoring of the survivors
        peak, kern, dtype = BF16_PEAK_TFLOPS, "knn_sweep_bf16_kernel (v_mfma_f32_16x16x32_bf16, 256x256x64 main loop)", "bf16 sweep / f32 re-score"
    tflops = 2.0 * pairs * d / (sweep_ms * 1e-3) / 1e12
    algo_bytes = (nq + nr) * d * 2   # both bf16 banks once; every query block re-reads the reference bank through L2 / Infinity Cache
    traffic = search_traffic()
    # the two other sweeps of the path on the same bank, 8192 queries (secondary; both share the pre-filter since round 2):
    # range search (faiss range_search: CandidateGeneration's global-threshold fallback) and the matching track's video-pair maxima
    others = {}
    try:
        from statistics import NormalDist
        nq2 = min(8192, nq)
        thr = NormalDist().inv_cdf(1.0 - 1e-4) / d ** 0.5     # ~0.01 % of the pairs of random unit vectors
        q2 = q[:nq2]
        qv = (torch.arange(nq2, device=dev) // 32).int()
        rv = (torch.arange(nr, device=dev) // 32).int()
        for name, fn, last in (("range_search", lambda: ops.range_search_ip(q2, r, thr, capacity=1 << 22), lib.vsc_range_search_last_path),
                               ("video_pair_max", lambda: ops.video_pair_max(q2, qv, int(qv[-1]) + 1, r, rv, int(rv[-1]) + 1, thr, capacity=1 << 22),
                                lib.vsc_video_pair_max_last_path)):
            fn()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            others[name] = {"nq": nq2, "nr": nr, "threshold": round(thr, 4), "ms": round((time.perf_counter() - t1) * 1e3, 3),
                            "hits": int(out[0][-1]), "path": {1: "exact fp32", 2: "bf16 pre-filter + exact re-score", 3: "pre-filter overflowed -> exact"}[last()]}
    except Exception as exc:  # noqa: BLE001 -- secondary numbers must not cost the primary line
        others["error"] = f"{type(exc).__name__}: {exc}"
    cpu = None
    if not args.no_cpu_baseline:
        try:
            # a handful of the timed call's own rows (first / middle / last query blocks, the balanced tail) go to the oracle with the bank
            rows = torch.tensor(sorted({0, 1, 255, 256, nq // 3, nq // 2, nq - 257, nq - 2, nq - 1} | {min(nq - 1, (nq // 256 // 256) * 65536 + j) for j in (0, 100)}), device=dev)
            chk = (q[rows].cpu().numpy(), r.cpu().numpy(), D[rows].cpu().numpy(), I[rows].cpu().numpy(), k) if nr * d * 4 <= (4 << 30) else None
            del D, I
            cpu = search_cpu_baseline(check=chk)
            if chk is not None and not (cpu["oracle_check_of_the_timed_result"]["ids_equal"] and cpu["oracle_check_of_the_timed_result"]["scores_bit_equal"]):
                raise AssertionError(f"the timed search result differs from the oracle: {cpu['oracle_check_of_the_timed_result']}")
        except Exception as exc:  # noqa: BLE001 -- a baseline must not cost the line
            cpu = {"error": f"{type(exc).__name__}: {exc}"}
    return {"other_sweeps": others, "cpu_baseline": cpu, "metric": "Mpairs/s (512-d exact inner-product top-k sweep, BASELINE.json configs[2])",
            "value": round(pairs / (ms * 1e-3) / 1e6, 1), "unit": "Mpairs/s", "nq": nq, "nr": nr,
            "k": k, "dtype": dtype, "ms_per_sweep": round(ms, 3), "path": {1: "exact fp32 sweep", 2: "bf16 pre-filter + exact re-score", 3: "pre-filter, some blocks redone exactly"}[path],
            "phases_ms": {"pack": round(float(phases[0]), 3), "sweep": round(sweep_ms, 3), "rescore": round(float(phases[2]), 3),
                          "merge": round(float(phases[3]), 3)},
            "roofline": {"bound": "mfma", "kernel": kern,
                         "achieved": round(tflops, 2), "peak": peak,
                         "unit": "TFLOP/s", "frac": round(tflops / peak, 4),
                         "traffic": None if traffic is None else round(traffic[0]),
                         "traffic_note": None if traffic is None else
                         f"memory-side bytes of one sweep launch at nq={traffic[1]}, nr={traffic[2]} (profiles/{traffic[3]}); "
                         f"algorithmic operand bytes of that launch: {(traffic[1] + traffic[2]) * d * 2}",
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "note": "achieved = 2 nq nr d / HIP-event time of the sweep kernel alone (vsc_knn_last_profile); value covers the whole call "
                                 "(pack + sweep + exact re-scoring + merge + one host sync)"}}


def bench_search_sharded(dev, args, dist, rank, world):
    """N > 1: the path of BASELINE.json configs[3] at bench size -- "global top-k + score-norm": every rank holds nr / N reference
    descriptors and its own nq queries; every rank normalises ITS queries against the (replicated, 100k-row) noise bank, the bank is
    assembled source shard by source shard over RCCL and every shard is swept as it lands (vsc_hip.distributed.
    sharded_knn_score_normalized(pipelined=True)); the one-gather form is timed beside it.  Weak scaling: per-GPU sweep work is fixed.
    Collective: runs on every rank."""
    from vsc_hip import ops
    from vsc_hip.distributed import sharded_knn_score_normalized
    d, k = 512, args.search_k
    nq, nr_local = args.search_nq, args.search_nr // world
    g = torch.Generator(device=dev).manual_seed(100 + rank)
    r = torch.randn(nr_local, d, generator=g, device=dev)
    q = torch.randn(nq, d, generator=g, device=dev)
    gn = torch.Generator(device=dev).manual_seed(99)          # the same noise bank on every rank
    noise = torch.randn(min(100_000, max(args.search_nr // 10, 1024)), d, generator=gn, device=dev)
    for t in (r, q, noise):
        ops.l2_normalize_(t)
    res = {}
    for name, pipelined in (("pipelined", True), ("one_gather", False)):
        sharded_knn_score_normalized(q[:256], r, noise, k, gather_to=None, pipelined=pipelined)   # warm: RCCL communicator, scratch
        times = []
        for _ in range(max(args.search_steps, 1)):
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            D, I = sharded_knn_score_normalized(q, r, noise, k, gather_to=None, pipelined=pipelined)
            torch.cuda.synchronize()
            dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            times.append(float(dt.item()))
        res[name] = sum(times) / len(times)
    t = res["pipelined"]            # `value` is the form the metric names, whichever of the two is faster (both times are in the object)
    pairs = float(world) * nq * (nr_local * world)
    return {"metric": "Mpairs/s (513-d exact top-k with score normalisation, bank all_gathered over RCCL shard by shard, every rank sweeps its own queries)",
            "value": round(pairs / t / 1e6, 1), "unit": "Mpairs/s", "form": "pipelined: one broadcast per source shard, swept as it lands",
            "value_one_gather": round(pairs / res["one_gather"] / 1e6, 1), "n_gpus": world, "nq_per_gpu": nq,
            "nr_total": nr_local * world, "k": k, "dtype": "bf16 sweep / f32 re-score", "ms_per_sweep_incl_all_gather": round(t * 1e3, 3),
            "ms_pipelined": round(res["pipelined"] * 1e3, 3), "ms_one_gather": round(res["one_gather"] * 1e3, 3), "score_norm_noise_rows": int(noise.shape[0]),
            "scaling": "weak", "all_gather_bytes_per_rank": nr_local * (d + 1) * 4 * (world - 1)}


def swin_traffic():
    """Mean memory-side bytes per GEMM-class launch of the Swin step (gemm_ln_kernel, gemm_bf16_v*, swin_mlp_kernel) from the
    committed PMC passes over `tools/swin_bench.py 256 3 256` (profiles/r05_pmc_swin.json; 2 x FETCH_SIZE + WRITE_SIZE as in
    gemm_traffic) -> (bytes, launches, source) or None."""
    try:
        src = next(n for n in ("r06_pmc_swin.json", "r05_pmc_swin.json") if os.path.exists(os.path.join(ROOT, "profiles", n)))
        prof = json.load(open(os.path.join(ROOT, "profiles", src)))["per_launch"]
        tot, n = 0.0, 0
        for key, v in prof.items():
            if key.startswith(("gemm_ln_kernel", "gemm_bf16_v", "swin_mlp_kernel", "swin_mlp512_kernel")) and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                tot += v["launches"] * (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
                n += v["launches"]
        return (tot / n, n, f"profiles/{src}") if n else None
    except (OSError, KeyError, ValueError, StopIteration):
        return None


def bench_swin(dev, args):
    """Secondary: the reference's other backbone family (swinv2_v1xx), same contract as the ViT step."""
    from tools import synth
    from vsc_hip.swin_config import get_swin_config
    from vsc_hip.swin_encoder import SwinHipEncoder
    cfg = get_swin_config("swinv2_base_256")
    enc = SwinHipEncoder(cfg, synth.swin_weights(5, cfg), max_batch=args.swin_batch, l2_normalize=True)
    b = 2 * args.swin_batch   # two encoder chunks per step: they run on the Swin encoder's two lanes, as the ViT step's do
    x = torch.from_numpy(synth.swin_frames(1, 8, cfg)).to(dev).repeat((b + 7) // 8, 1, 1, 1)[:b].contiguous()
    for _ in range(2):
        enc(x)
    torch.cuda.synchronize()
    steps = 5
    t0 = time.perf_counter()
    for _ in range(steps):
        out = enc(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    assert torch.isfinite(out).all()
    # per-launch HIP events (vsc_swin_set_profiling): the chunks run back to back on one stream, every kernel alone
    psteps = 2
    enc.set_profiling(True)
    for _ in range(psteps):
        enc(x)
    prof = enc.profile()
    enc.set_profiling(False)
    enc.close()
    kernels, gemm_ms, gemm_flop, all_ms = {}, 0.0, 0.0, 0.0
    gemm_bytes, gemm_launches = 0.0, 0
    frames_prof = b * psteps
    for name, (ms, cnt) in prof.items():
        flop, is_gemm = 0.0, False
        if name == "patch_embed":
            flop, is_gemm = 2.0 * (cfg.image_size // cfg.patch_size) ** 2 * cfg.embed_dim * 64, True
        elif name[0] == "s" and "." in name:
            st, kind = int(name[1]), name.split(".")[1]
            C, R = cfg.dim(st), cfg.resolution(st)
            T, N = R * R, min(cfg.window_size, R) ** 2
            per_block = {"qkv": 2.0 * T * 3 * C * C, "proj_ln": 2.0 * T * C * C, "fc1": 2.0 * T * 4 * C * C, "fc2_ln": 2.0 * T * 4 * C * C,
                         "attention": 4.0 * T * N * C}.get(kind)
            # blocks whose qkv Linear ran inside the previous block's fused kernel (stage 2: csrc/swin_mlp512.hip, variant 9), as a
            # share of the stage's blocks: the qkv class has that many launches fewer than the fc2_ln class
            qkv_inside = 1.0 - prof[f"s{st}.qkv"][1] / prof[f"s{st}.fc2_ln"][1]
            if kind == "merge":
                flop = 2.0 * (T // 4) * (2 * C) * (4 * C)
            else:
                flop = per_block * cfg.depths[st]
                if kind == "qkv":
                    flop *= 1.0 - qkv_inside
                if kind == "fc2_ln" and f"s{st}.fc1" not in prof:   # fused MLP kernel (swin_mlp.hip): both Linears in this class
                    flop *= 2
                    if f"s{st}.proj_ln" not in prof:                # ... and the projection in front of them (PROJ form)
                        flop += 2.0 * T * C * C * cfg.depths[st]
                    flop += 2.0 * T * 3 * C * C * cfg.depths[st] * qkv_inside   # ... and the next block's qkv Linear behind them
            is_gemm = kind != "attention"
            # algorithmic bytes of ONE launch of this class (args.swin_batch frames): operands in + results out, weights once.
            # x fp32 4 B, shadow / qkv / attention output / hidden bf16 2 B per element.
            M, fused_mlp, fused_proj = float(args.swin_batch * T), f"s{st}.fc1" not in prof, f"s{st}.proj_ln" not in prof
            per_launch = {"qkv": M * C * 2 + M * 3 * C * 2 + 3 * C * C * 2,
                          "proj_ln": M * C * 2 + M * C * (4 + 4 + 2) + C * C * 2,
                          "fc1": M * C * 2 + M * 4 * C * 2 + 4 * C * C * 2,
                          # (fused, PROJ form: att in, x in and out, shadow out, W1 W2 Wp; with the next qkv inside: qkv out instead of the shadow, + Wqkv)
                          "fc2_ln": (M * C * 2 + M * C * (4 + 4 + 2) + 8 * C * C * 2 + (M * C * 2 + C * C * 2 if fused_proj else 0)
                                     + qkv_inside * (M * 3 * C * 2 - M * C * 2 + 3 * C * C * 2)) if fused_mlp
                                    else M * 4 * C * 2 + M * C * (4 + 4 + 2) + 4 * C * C * 2,
                          "merge": M * C * 2 + (M / 4) * 2 * C * (4 + 2) + 8 * C * C * 2}.get(kind)
            if per_launch and is_gemm:
                kernels.setdefault(name, {})["algorithmic_bytes_per_launch"] = round(per_launch)
                gemm_bytes += per_launch * cnt
                gemm_launches += cnt
        tf = flop * frames_prof / (ms * 1e-3) / 1e12 if flop and ms else None
        kernels.setdefault(name, {}).update({"ms_per_step": round(ms / psteps, 4), "launches_per_step": cnt // psteps, "avg_launch_us": round(ms / cnt * 1e3, 2)})
        if "algorithmic_bytes_per_launch" in kernels[name]:
            kernels[name]["tb_per_s"] = round(kernels[name]["algorithmic_bytes_per_launch"] / (ms / cnt * 1e-3) / 1e12, 2)
        if tf:
            kernels[name]["tflops"] = round(tf, 1)
            kernels[name]["frac"] = round(tf / BF16_PEAK_TFLOPS, 4)
        all_ms += ms
        if is_gemm:
            gemm_ms += ms
            gemm_flop += flop * frames_prof
    gemm_tf = gemm_flop / (gemm_ms * 1e-3) / 1e12 if gemm_ms else 0.0
    traffic = swin_traffic()
    wbusy = attention_mfma_busy("window_attention", "_swin")
    return {"metric": "frames/s (Swin-V2-B 256x256 window-16 encode -> L2-normalised 512-d descriptors)",
            "value": round(b / dt, 1), "unit": "frames/s", "frames_per_step": b, "encoder_chunk": args.swin_batch, "lanes": 2,
            "ms_per_step": round(dt * 1e3, 3),
            "dtype": "bf16", "gflop_per_frame": round(cfg.flops_per_frame() / 1e9, 2),
            "model_tflops": round(cfg.flops_per_frame() * b / dt / 1e12, 1),
            "roofline": {"bound": "mfma", "kernel": "all GEMM-class launches of the Swin-V2-B step (swin_mlp_kernel / swin_mlp512_kernel: the whole MLP of a block of stages "
                                                    "0-2; gemm_ln_kernel / gemm_bf16_v4<LN_RES>: proj / merge / patch embedding with their LayerNorms; gemm_bf16_v4 / v3 / v2: qkv, stage-3 fc1)",
                         "achieved": round(gemm_tf, 1), "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(gemm_tf / BF16_PEAK_TFLOPS, 4), "traffic": None if traffic is None else round(traffic[0]),
                         "algorithmic_bytes_per_launch": round(gemm_bytes / gemm_launches) if gemm_launches else None,
                         "traffic_over_algorithmic": None if traffic is None or not gemm_launches else round(traffic[0] / (gemm_bytes / gemm_launches), 2),
                         "traffic_unit": None if traffic is None else f"memory-side bytes per GEMM-class launch, mean over {traffic[1]} launches "
                                                                        f"({traffic[2]}: 2 x FETCH_SIZE + WRITE_SIZE, separate --pmc passes)",
                         "window_attention_mfma_busy": wbusy,
                         "gemm_ms_per_step": round(gemm_ms / psteps, 3), "all_kernels_ms_per_step": round(all_ms / psteps, 3),
                         "whole_model_frac": round(cfg.flops_per_frame() * b / dt / 1e12 / BF16_PEAK_TFLOPS, 4),
                         "note": "achieved = sum 2 M N K of the GEMM launches / sum of their HIP-event durations, taken in a separate loop of "
                                 f"{psteps} steps on one stream (vsc_swin_set_profiling); the headline loop runs the two chunks on two lanes"},
            "kernels": kernels}


def bench_fp16_operands(dev, args, cfg, weights, frames):
    """Secondary: the SAME ViT step and the Swin-V2-B step with fp16 instead of bf16 as the 16-bit MFMA operand type (libvsc_hip_f16.so;
    what the infer/ entry points default to, because the end-to-end uAP parity needs it: tests/test_gpu_uap_e2e.py).
    Same kernels, same MFMA rate (v_mfma_f32_16x16x32_f16), same bytes: this object exists to show the rate is the same."""
    from tools import synth
    from vsc_hip.encoder import HipEncoder
    from vsc_hip.swin_config import get_swin_config
    from vsc_hip.swin_encoder import SwinHipEncoder
    from vsc_hip import _lib
    out = {"operands": "fp16", "library": _lib.require_device("fp16").vsc_version().decode()}
    enc = HipEncoder(cfg, weights, max_batch=args.max_batch, l2_normalize=True, lanes=args.lanes, precision="fp16")
    steps = max(1, min(args.steps, 40))
    for _ in range(2):
        enc(frames)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        d = enc(frames)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.isfinite(d).all()
    enc.set_profiling(True)
    for _ in range(4):
        enc(frames)
    torch.cuda.synchronize()
    prof = enc.get_profile()
    enc.close()
    fpf = gemm_flops_per_frame(cfg)
    gemm_ms = sum(prof[k][0] for k in fpf)
    tf = sum(fpf.values()) * 4 * args.batch / (gemm_ms * 1e-3) / 1e12
    out["vit"] = {"value": round(steps * args.batch / dt, 1), "unit": "frames/s", "steps": steps, "ms_per_step": round(1e3 * dt / steps, 3),
                  "gemm_tflops": round(tf, 1), "gemm_frac": round(tf / BF16_PEAK_TFLOPS, 4),
                  "avg_launch_us": {k: round(1e3 * v[0] / max(v[1], 1), 2) for k, v in prof.items() if v[1]}}
    if not args.no_swin:
        scfg = get_swin_config("swinv2_base_256")
        senc = SwinHipEncoder(scfg, synth.swin_weights(5, scfg), max_batch=args.swin_batch, l2_normalize=True, precision="fp16")
        b = 2 * args.swin_batch
        x = torch.from_numpy(synth.swin_frames(1, 8, scfg)).to(dev).repeat((b + 7) // 8, 1, 1, 1)[:b].contiguous()
        for _ in range(2):
            senc(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            d = senc(x)
        torch.cuda.synchronize()
        out["swin"] = {"value": round(5 * b / (time.perf_counter() - t0), 1), "unit": "frames/s",
                       "note": "q-hat / k-hat and every Linear on fp16 operands; the P . V product of the window attention stays on bf16 (bounded softmax: csrc/swin.hip)"}
        assert torch.isfinite(d).all()
        senc.close()
    return out


def bench_matching(dev, args):
    """Secondary: the matching track's two fp32 networks (infer_matching.py:158-204) at the reference's batch sizes --
    mobilenetv3_small_100 pair classifier on 2048 x 3 x 160 x 160 similarity maps, hrnet_w18 refinement net on
    16 x 3 x 224 x 224 -- synthetic timm-named weights (tests/cnn_synth.py).  fp32 MFMA roof: 157.3 TF/s."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cnn_synth
    from vsc_hip import cnn

    def run(model, x, iters):
        cnn.FLOPS = [0.0, 0.0]
        model(x)
        flop = (cnn.FLOPS[0], cnn.FLOPS[1])    # (all convolution calls, the share on the bf16 pipe with split operands)
        cnn.FLOPS = None
        model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            out = model(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        assert torch.isfinite(out).all()
        return dt, flop

    cls = cnn.MobileNetV3SmallHip(cnn_synth.mobilenetv3_small_state(1), dev)
    xc = cnn_synth.similarity_maps(2, 8, 160, 160).to(dev).repeat(256, 1, 1, 1)
    dtc, (fc, _) = run(cls, xc, 5)
    ref = cnn.HRNetRefineHip(cnn_synth.hrnet_refine_state(3), dev)
    xr = cnn_synth.similarity_maps(4, 16, 224, 224).to(dev)
    dtr, (fr, fr_x3) = run(ref, xr, 5)
    # time floor of the pass if every layer ran at the roof of the pipe that executes it: fp32 matrix pipe 157.3 TF/s; split-bf16 layers
    # issue six bf16 MFMA products per logical product: 2500 / 6 = 417 logical TF/s
    x3_roof = BF16_PEAK_TFLOPS / 6.0
    floor_s = (fr - fr_x3) / (F32_MFMA_PEAK_TFLOPS * 1e12) + fr_x3 / (x3_roof * 1e12)
    return {"dtype": "f32 (HRNet's thin 3 x 3 layers: bf16 pipe on split operands x = x1 + x2 + x3, six products, fp32 accumulation -- fp32-level error)",
            "peak_tflops": F32_MFMA_PEAK_TFLOPS,
            "classifier": {"model": "mobilenetv3_small_100, 2 classes", "batch": list(xc.shape), "ms": round(dtc * 1e3, 2),
                           "maps_per_s": round(xc.shape[0] / dtc, 0), "gflop_per_map": round(fc / xc.shape[0] / 1e9, 4),
                           "tflops": round(fc / dtc / 1e12, 2), "frac_of_f32_mfma_peak": round(fc / dtc / 1e12 / F32_MFMA_PEAK_TFLOPS, 4)},
            "refiner": {"model": "hrnet_w18 features + fuse head", "batch": list(xr.shape), "ms_per_pass": round(dtr * 1e3, 2),
                        "maps_per_s": round(xr.shape[0] / dtr, 1), "gflop_per_map": round(fr / xr.shape[0] / 1e9, 3),
                        "tflops": round(fr / dtr / 1e12, 2), "frac_of_f32_mfma_peak": round(fr / dtr / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                        "flop_share_on_split_bf16_pipe": round(fr_x3 / fr, 4), "split_bf16_logical_roof_tflops": round(x3_roof, 1),
                        "frac_of_per_pipe_roof": round(floor_s / dtr, 4),
                        "per_pipe_note": "frac_of_per_pipe_roof = (fp32-pipe FLOPs / 157.3 TF/s + split-bf16 logical FLOPs / 417 TF/s) / measured time: "
                                         "every layer priced against the pipe that runs it (vsc_conv_last_pipe); frac_of_f32_mfma_peak prices all of "
                                         "them against the fp32 pipe and is not a roofline fraction for the split layers"},
            "note": "whole-network wall time per batch, inputs resident in HBM; FLOPs = 2 x MACs of every convolution call "
                    "(depthwise included), each counted once whatever pipe ran it: frac_of_f32_mfma_peak prices the network against the fp32 matrix pipe; "
                    "parity of these networks is unpinned (DESIGN.md section 3)"}


def main():
    args = parse()
    # The contract is ONE line on stdout.  Libraries write there too (RCCL prints its version banner to the C stdout,
    # flushed at exit, i.e. after the JSON line): keep a private handle on the real stdout for the result and point
    # fd 1 at stderr for everything else.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
        args.gpus = world

    from tools import synth
    from vsc_hip import _lib
    from vsc_hip.config import get_config
    from vsc_hip.encoder import HipEncoder

    _lib.require_device()
    dev_index = local_rank % torch.cuda.device_count() if args.share_device else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1 or args.force_sharded_search:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.backend)

    cfg = get_config(args.preset)
    weights = synth.encoder_weights(7, cfg)
    enc = HipEncoder(cfg, weights, max_batch=args.max_batch, l2_normalize=True, lanes=args.lanes, fuse_ln=args.fuse_ln)
    # every rank encodes its own shard of the (synthetic) frame set
    base = synth.frames(1000 + rank, 32, cfg)
    frames = torch.from_numpy(base).to(dev).repeat((args.batch + 31) // 32, 1, 1, 1)[: args.batch]
    frames = (frames + 0.001 * torch.arange(args.batch, device=dev).view(-1, 1, 1, 1)).contiguous()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    tw = time.perf_counter()
    for _ in range(args.warmup):
        enc(frames)
    barrier()
    step_est = (time.perf_counter() - tw) / max(args.warmup, 1)   # sizes the clock probe below
    sampler = ClockSampler() if rank == 0 else None
    spin = None
    if sampler:
        sampler.__enter__()
        # cycle-counted shader clock under this load: one wave on a side stream spins for a fixed number of
        # s_memtime ticks (= shader cycles) while the timed steps run; ticks / event time = the clock the
        # kernels really ran at (rocm-smi's sclk is a smoothed reading)
        side = torch.cuda.Stream()
        spin = {"ticks": torch.zeros(1, dtype=torch.int64, device=dev), "e0": torch.cuda.Event(enable_timing=True),
                "e1": torch.cuda.Event(enable_timing=True)}
    # ---- the headline: EXACTLY args.steps steps, no per-launch events, barrier + synchronize on both sides
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = enc(frames)
        if spin is not None and i == 0:
            with torch.cuda.stream(side):
                spin["e0"].record()
                _lib.check(_lib.require_device().vsc_debug_spin_ticks(max(int(0.4 * args.steps * step_est * 1.2e9), 100000), spin["ticks"].data_ptr(),
                                                                      side.cuda_stream))
                spin["e1"].record()
    barrier()
    dt = time.perf_counter() - t0
    if sampler:
        sampler.__exit__()
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(out).all()
    # ---- a separate short loop with HIP events around every launch (on the launch stream): kernels{} and roofline
    psteps = max(1, min(args.profile_steps, args.steps))
    enc.set_profiling(True)
    for _ in range(psteps):
        enc(frames)
    torch.cuda.synchronize()
    prof = enc.get_profile()
    enc.set_profiling(False)

    search_multi = None
    if (world > 1 or args.force_sharded_search) and not args.no_search:
        try:
            search_multi = bench_search_sharded(dev, args, dist, rank, world)
        except Exception as exc:  # noqa: BLE001 -- the primary line must still be printed
            search_multi = {"error": f"{type(exc).__name__}: {exc}"}

    if rank == 0:
        total_frames = args.steps * args.batch * world
        fpf = gemm_flops_per_frame(cfg)
        gemm_ms = sum(prof[k][0] for k in fpf)
        gemm_flops = sum(fpf.values()) * psteps * args.batch
        achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        per_class = {k: {"ms_per_step": round(v[0] / psteps, 4), "launches_per_step": v[1] // psteps,
                         "avg_launch_us": round(1e3 * v[0] / max(v[1], 1), 2)}
                     for k, v in prof.items() if v[1]}
        for k in fpf:
            per_class[k]["tflops"] = round(fpf[k] * args.batch * psteps / (prof[k][0] * 1e-3) / 1e12, 1)
            per_class[k]["frac"] = round(per_class[k]["tflops"] / BF16_PEAK_TFLOPS, 4)
        if prof.get("attention", (0, 0))[0] > 0:
            # the two attention GEMMs (Q K^T and P V): 4 T^2 head_dim per head and frame; MFMA-busy share of the launch from the
            # committed PMC pass (profiles/r05_pmc_mfma_busy.json, else the round-2 file)
            att_flop = cfg.layers * 4.0 * cfg.tokens * cfg.tokens * cfg.width
            tf = att_flop * args.batch * psteps / (prof["attention"][0] * 1e-3) / 1e12
            # its bytes: q, k, v in (3 D bf16 per token) and the context out (D bf16 per token) -- the kernel is bound by them, not by the pipe
            att_bytes = 8.0 * cfg.tokens * cfg.width * min(args.max_batch, args.batch)
            att_us = 1e3 * prof["attention"][0] / max(prof["attention"][1], 1)
            tbs = att_bytes / (att_us * 1e-6) / 1e12
            per_class["attention"].update({"tflops": round(tf, 1), "frac": round(tf / BF16_PEAK_TFLOPS, 4),
                                           "mfma_busy": attention_mfma_busy("attention_kernel"),
                                           "bound": "hbm", "algorithmic_bytes_per_launch": round(att_bytes), "tb_per_s": round(tbs, 2),
                                           "frac_hbm": round(tbs / HBM_PEAK_TBS, 4), "hbm_peak_tb_per_s": HBM_PEAK_TBS})
        traffic, algo_bytes, traffic_src = gemm_traffic(cfg, min(args.max_batch, args.batch))
        line = {
            "metric": "frames/s (ViT-B/16 224x224 encode -> L2-normalised 512-d descriptors)",
            "value": round(total_frames / dt, 1), "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            **({"shared_device_plumbing_run": f"{world} ranks on {torch.cuda.device_count()} device(s), backend {args.backend}: exercises the N > 1 code path, "
                                              "says nothing about scaling"} if args.share_device else {}),
            "library": _lib.require_device().vsc_version().decode(),     # "... src <hash of csrc/* + include/vsc_hip.h>": which build was measured
            "config": {"workload": f"{cfg.name} bf16 encode of {args.steps * args.batch} synthetic "
                                   f"{cfg.image_size}x{cfg.image_size} frames per GPU in {args.steps} steps of {args.batch} "
                                   "(BASELINE.json configs[1]: 100k frames)",
                       "frames_per_gpu": args.steps * args.batch,
                       "frames_per_step_per_gpu": args.batch, "encoder_chunk": args.max_batch,
                       "lanes": args.lanes, "tokens": cfg.tokens,
                       "gflop_per_frame": round(cfg.flops_per_frame() / 1e9, 2),
                       "parallelism": f"frames sharded over {world} rank(s), no data-path collective"},
            "model_tflops": round(cfg.flops_per_frame() * total_frames / dt / 1e12 / world, 1),
            "roofline": {"bound": "mfma", "kernel": "gemm_bf16_v4_kernel (qkv/proj/fc1/fc2; the patch GEMM runs gemm_bf16_v3_kernel: same K loop, one tile per workgroup)",
                         "achieved": round(achieved, 1), "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / BF16_PEAK_TFLOPS, 4),
                         "traffic": None if traffic is None else round(traffic),
                         "traffic_unit": "HBM bytes per GEMM launch (mean over the 49 launches of one 332-frame chunk; "
                                         f"2 x FETCH_SIZE + WRITE_SIZE from profiles/{traffic_src}: memory-side L2 requests, Infinity-Cache hits included)",
                         "algorithmic_bytes_per_launch": round(algo_bytes),
                         "gemm_ms_per_step": round(gemm_ms / psteps, 3),
                         "profiled_steps": psteps,
                         "note": "per-launch HIP-event time on the launch stream, taken in a separate loop of profiled_steps steps right "
                                 "after the timed region (the headline loop carries no events).  The headline loop runs the two chunks of a "
                                 "step on two streams (config.lanes); the event loop runs them back to back on one, so its kernel times "
                                 "are per kernel alone and their sum exceeds ms_per_step"},
            "kernels": per_class,
        }
        clk = sampler.summary() if sampler else None
        if clk is None and spin is not None:
            clk = {}
        if spin is not None:
            torch.cuda.synchronize()
            clk["sclk_mhz_smi"] = clk.pop("sclk_mhz", None)
            clk["sclk_mhz"] = round(int(spin["ticks"].item()) / (spin["e0"].elapsed_time(spin["e1"]) * 1e3))
            clk["sclk_source"] = "s_memtime cycles of a one-wave spin on a side stream / its event time, during the timed steps"
        if clk:
            # the roof these launches ran under: peak scaled by the clock the power limit allowed
            clk["mfma_roof_at_sustained_clock_tflops"] = round(BF16_PEAK_TFLOPS * clk["sclk_mhz"] / PEAK_SCLK_MHZ, 1)
            clk["frac_of_sustained_roof"] = round(achieved / clk["mfma_roof_at_sustained_clock_tflops"], 4)
            line["roofline"]["sustained"] = clk
        secondary = world == 1   # cpu baseline and the secondary metrics are rank-0, N = 1 only
        if not args.no_cpu_baseline and secondary:
            line["cpu_baseline"] = cpu_baseline(cfg, weights, base)
        enc.close()
        if secondary and not args.no_fp16:
            try:
                line["fp16_operands"] = bench_fp16_operands(dev, args, cfg, weights, frames)
            except Exception as exc:  # noqa: BLE001 -- a secondary: the primary line must still be printed
                line["fp16_operands"] = {"error": f"{type(exc).__name__}: {exc}"}
        del frames
        torch.cuda.empty_cache()
        if not args.no_swin and secondary:
            line["swin"] = bench_swin(dev, args)
            torch.cuda.empty_cache()
        if not args.no_matching and secondary:
            line["matching"] = bench_matching(dev, args)
            torch.cuda.empty_cache()
        if not args.no_ensemble and secondary:
            try:
                from tools import ensemble_bench
                line["ensemble"] = ensemble_bench.measure(dev, args.ensemble_videos, 40)      # at the entry points' default operand type (fp16)
                line["ensemble"]["value_bf16_operands"] = ensemble_bench.measure(dev, args.ensemble_videos, 40, precision="bf16")["value"]
            except Exception as exc:  # noqa: BLE001 -- a secondary: the primary line must still be printed
                line["ensemble"] = {"error": f"{type(exc).__name__}: {exc}"}
            torch.cuda.empty_cache()
        if not args.no_search and secondary:
            line["search"] = bench_search(dev, args)
        if search_multi is not None:
            line["search"] = search_multi
        os.write(result_fd, (json.dumps(line) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
