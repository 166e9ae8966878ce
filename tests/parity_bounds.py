"""Mean-error bounds of the end-to-end descriptor parity tests (GPU).

`DESC_L2_ATOL = 1e-3` on the maximum cannot see a small systematic error: a 1 % scale error in ONE weight matrix moves ViT-B/16
descriptors by 3.5e-4, and the bf16 pipeline's own maximum error is 2.4-4e-4.  Rounding noise averages out over a descriptor, a
bias does not -- so every case also bounds
    mean |d|   <= 1.15 x the value measured for this case      (d = HIP - reference over all frames and dimensions)
    |mean d|   <= max(2 x measured, 1e-5)
The measured values (tools/parity_stats.py on the GPU, profiles/r05_parity_stats.txt) are per case because depth, width and the
descriptor's own statistics set the noise floor.  The kernels are deterministic, so the statistics are exact for a given build;
the 15 % covers rounding-order changes of future kernels (they move mean |d| by a few percent), not an error.
What the bounds catch on ViT-B/16 (same file): blocks.5.fc1.weight x 1.01 -> mean |d| 1.34e-4 against the bound 1.07e-4;
blocks.0.qkv.bias x 1.05 -> 1.9e-4; blocks.11.proj.weight x 1.01 -> 1.09e-4 (tests/test_gpu_encoder.py::test_mean_bound_sees_a_one_percent_scale_error).
What they cannot: Swin-V2's res-post-norm blocks put a LayerNorm behind every Linear pair, which removes a weight-scale error
before it reaches the residual stream -- the same three perturbations move Swin-V2-B's statistics by less than their noise."""
import numpy as np

# case -> (mean |d|, |mean d|) measured
MEASURED = {
    "vit/tiny": (6.58e-5, 9.4e-7), "vit/tiny_clip": (3.15e-5, 2.8e-8), "vit/vit_b16_224": (9.26e-5, 3.1e-6), "vit/vit_v68": (7.05e-5, 1.6e-7),
    "vit/fresh": (8.81e-5, 5.05e-6), "vit/benchmarked": (8.95e-5, 5.98e-6),
    "swin/tiny_swin": (1.276e-4, 3.30e-5), "swin/tiny_swin_w8": (1.579e-4, 3.84e-5), "swin/swinv2_base_256": (1.106e-4, 1.55e-5),
    # the second fixtures (frames that differ from one another, synth.structured_frames): maxima 6.4e-4 (ViT-B/16), 8.8e-4 / 9.4e-4 (Swin-V2-B
    # through the GEMM launches / the fused kernel of its 512-wide stage) -- nearer the 1e-3 bound than the noise-frame fixtures' 2.4e-4
    "vit/vit_b16_224_structured": (1.476e-4, 1.48e-6), "swin/swinv2_base_256_structured": (1.580e-4, 7.9e-6),
    "swin/tiny_swin_w24": (1.869e-4, 2.44e-5), "swin/swinv2_large_384": (1.237e-4, 1.14e-5), "swin/benchmarked": (1.201e-4, 1.52e-6),
}
MEAN_SLACK, BIAS_SLACK, BIAS_FLOOR = 1.15, 2.0, 1e-5


def bounds(case):
    m, b = MEASURED[case]
    return MEAN_SLACK * m, max(BIAS_SLACK * b, BIAS_FLOOR)


def stats(out, ref):
    d = np.asarray(out, np.float64) - np.asarray(ref, np.float64)
    return float(np.abs(d).mean()), float(abs(d.mean()))


def check(case, out, ref):
    mean_abs, bias = stats(out, ref)
    mb, bb = bounds(case)
    assert mean_abs <= mb, f"{case}: mean |HIP - reference| = {mean_abs:.3e} > {mb:.3e} (measured {MEASURED[case][0]:.3e})"
    assert bias <= bb, f"{case}: |mean (HIP - reference)| = {bias:.3e} > {bb:.3e} (measured {MEASURED[case][1]:.3e})"
