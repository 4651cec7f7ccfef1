#!/usr/bin/env python3
"""The per-kernel audit table of DESIGN.md section 5: one row per `__global__` kernel of
beyond_deep_ensembles_amd/csrc/*.hip -- where it is (file:line, looked up in the sources so it cannot go stale), what it
replaces in the reference, its algorithmic bytes / flops, the LAST device measurement with round and the file under
profiles/ it comes from, and whether the kernel source at HEAD has run on an MI355X (and through which test).

    python tools/kernel_table.py            # prints the table (markdown)
    python tools/kernel_table.py --write    # replaces the block between the KERNEL-TABLE markers in DESIGN.md

tests/test_abi.py::test_design_kernel_table_is_current checks that DESIGN.md holds exactly this output, i.e. that every
kernel in the sources has a row and every line number is current.
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "beyond_deep_ensembles_amd", "csrc")
BEGIN, END = "<!-- KERNEL-TABLE:BEGIN (tools/kernel_table.py --write) -->", "<!-- KERNEL-TABLE:END -->"

# device status keys
R3 = "yes -- green in the driver's r03 suite (GPUTEST_r03), source unchanged since"
R4B = "yes -- green in builder calls r4b/r4c (profiles/r04_pytest_gpu_call_{b,c}.log) after its last change"
NEVER = ("**no** -- changed after the last GPU call (pool closed since round 4); runs on the CPU model only; NOT a default path "
         "until `device_verified.json` records a green device run of these sources (§0)")

# kernel -> (replaces, algorithmic work per launch, last device measurement, device status, test)
ROWS = {
    # ---- SVGD, streaming path (headline)
    "svgd_gram_kernel": ("`cdist**2` (`svgd.py:15`), mean-centred Gram on `v_mfma_f32_16x16x4_f32`", "`4·M·D` B, `2·M²·D` flop",
                         "0.131 ms = 5.84 TB/s (0.73) alone, M=8 D=23.9M; r4 `r04_bench_torchrun_one_rank_rccl.json`", R4B,
                         "`test_svgd_step_golden`, `test_svgd_fullsize_*`"),
    "svgd_kstats_kernel": ("`quantile`, `exp`, row sums, coefficient matrices (`svgd.py:18-23,31,86,89`)", "one workgroup, latency",
                           "6–8 µs; r2 `r02_kstats_ab.txt`", R3, "`test_svgd_step_golden` (h, K vs fp64 anchor)"),
    "svgd_combine_kernel": ("`phi = K@(-(G+l2/2 P)) + c·gradK/N` (`svgd.py:23,31,86,89`): 2 matmuls + ~6 passes", "`12·M·D` B, `4·M²·D` flop",
                            "**409.3 µs avg of 128 calls = 5.60 TB/s (0.70)** rocprofv3 r3 `r03_trace_main_kernel_stats.csv`; PMC traffic 1.000× "
                            "algorithmic `r03_pmc_summary.json`; HIP events r4: 364.5 µs (0.786), 0.988 of same-shape probe", R3,
                            "`test_svgd_step_golden`, `test_svgd_fullsize_vs_oracle_and_fp64`"),
    "svgd_combine_seg_kernel": ("the same with `_store_grads` (`svgd.py:129-133`) gone: gradients read where autograd left them", "`12·M·D` B",
                                "0.403–0.408 ms vs 0.397 flat; r3 `r03_seg_bench.txt`", R3, "`test_svgd_segmented_gradients_equal_flat_rows`"),
    "svgd_gather_seg_kernel": ("packs segmented gradients into flat rows (collectives, small-model kernel, M > 16)", "`8·rows·D` B",
                               "6.0 TB/s (0.75); r3 `r03_seg_bench.txt`", R3, "`test_svgd_segmented_gradients_equal_flat_rows`"),
    "svgd_fused_kernel": ("`svgd.py:86-103` incl. the M `base_optimizer.step()` calls (SGD / Adam, shared state Q5) + next step's Gram",
                          "`(12M+8)·D` (SGD) / `(12M+16)·D` (Adam) B", "0.4725 ms = 5.26 TB/s (0.657) full step, M=8; r4 bench line", R3,
                          "`test_svgd_fused_optimizers_match_torch_shared_state`, `test_svgd_trajectory[hip-*]`"),
    "svgd_apply_sgd_kernel": ("the M shared-state SGD applications alone (`svgd.py:92-103`), 17 ≤ M ≤ 64", "`(12M+8)·D` B",
                              "0.995 ms step+apply (0.70); r4 bench line", R3, "`test_svgd_many_particles_one_apply_launch_equals_the_optimizer_loop`"),
    "svgd_apply_adam_kernel": ("the same for Adam", "`(12M+16)·D` B", "—", R3, "same test"),
    "svgd_pairs_to_d2_kernel": ("blocked path M > 16: pair partials → d²", "latency", "—", R3, "`test_svgd_blocked_path_for_more_than_16_particles`"),
    "svgd_kstats_generic_kernel": ("statistics for 17 ≤ M ≤ 64", "latency", "—", R3, "same test"),
    "svgd_combine_generic_kernel": ("combine for 17 ≤ M ≤ 64 (16 rows per pass)", "`(8M + 4M·⌈M/16⌉)·D` B", "—", R3, "same test"),
    "svgd_gram_finish_kernel": ("dimension-sharded exchange: slice Gram partials → fp64 block", "257 doubles / rank", "latency", R3,
                                "`test_svgd_sharded_hip_two_ranks_one_device[alltoall*]`"),
    "svgd_kstats_gmat_kernel": ("the ranks' fp64 blocks summed in rank order → identical statistics on every rank", "latency", "—", R3, "same test"),
    "sum_scalars_kernel": ("the returned loss (`svgd.py:66,72,105`): M−1 torch adds and the division by M → one launch (round 5; round 6: rounds like torch's GPU division, sum · fl(1/M))", "M scalars", "unmeasured",
                           "**no** -- new in round 5; CPU model green; behind the `mean_scalars` gate (default: torch's adds)", "`test_r5_sum_scalars_is_the_sequential_fp32_sum`"),
    # ---- SVGD, small models (BASELINE configs[1])
    "svgd_step_small_kernel": ("the whole of `svgd.py:86-103` for M ≤ 8, D ≤ 524,288: two launches of one kernel", "`12·M·D` B (one pass over P)",
                               "14.6 µs fused SGD at D=273,610 (two launches), r4 bench line -- measured BEFORE the protocol deletion (−223 lines)", NEVER,
                               "`test_svgd_small_model_kernel`, `test_svgd_small_model_fused_step`, `test_svgd_every_particle_count_small_model_kernel`, `test_svgd_trajectory[hip-*-small*]`"),
    # ---- SWAG
    "swag_update_kernel": ("`swag.py:100-104` (mean, second moment, ring row)", "`24·D` B", "0.0865 ms = 6.62 TB/s (0.83); r4 bench line", R4B,
                           "`test_swag_update_bit_exact` (bit-exact)"),
    "swag_sample_kernel": ("`swag.py:57,112-114`: `mean + W@eps_W + sqrt(diag)·eps_D`", "`4·D·(K+3)` B (+4D with supplied eps_D)",
                           "0.3666 ms = 5.99 TB/s (0.749), K=20; r4 bench line", R4B, "`test_swag_sample_golden_and_oracle`"),
    "swag_sample_batched_dma_kernel": ("S samples per pass, `[S,K]×[K,D]` on `v_mfma_f32_32x32x2_f32`, K ≤ 20, next tile via LDS-DMA", "`4·D·(K+2+S)` B",
                                       "0.918–0.952 ms for S=30 = 5.2–5.4 TB/s (0.65–0.68); 0.995–1.003 of its R22 W30 probe `r04_swag_batched_vs_probe.txt`", R4B,
                                       "`test_swag_batched_sampler_both_kernels_equal_single_samples`"),
    "swag_sample_batched_kernel": ("the same, register kernel (K > 20)", "`4·D·(K+2+S)` B", "0.712 of 8 TB/s at K=20 `r04_swag_batched_ab.txt`", R4B, "same test"),
    "philox_normal_kernel": ("noise hook (tests, small activations)", "`4·D` B", "—", R3, "`test_kernel_normals_equal_the_checker_transform`"),
    "philox_bits_kernel": ("Random123 known-answer hook", "—", "—", R3, "`test_kernel_words_equal_the_known_answers_and_the_checker`"),
    # ---- BBB element-wise
    "gauss_draw_fwd_kernel": ("`w = mu + softplus(rho)·eps` (`util.py:170-171,183`)", "`12·D` B", "0.0484 ms = 5.92 TB/s (0.74); r4 bench line", R3, "`test_gauss_draw_kl_golden`"),
    "gauss_draw_bwd_kernel": ("its autograd backward", "`24·D` B", "0.0867 ms = 6.61 TB/s (0.826); r4", R3, "`test_gauss_draw_kl_golden`"),
    "gauss_draw_fwd_scalar_kernel": ("unaligned per-tensor views", "`12·D` B", "—", R3, "`test_accumulating_unaligned_and_value_only_variants`"),
    "gauss_draw_bwd_scalar_kernel": ("unaligned per-tensor views", "`24·D` B", "—", R3, "same test"),
    "gauss_kl_kernel": ("closed-form KL + both gradients (`bbb.py:18-21,71-74`)", "`24·D` B accumulate / `16·D` overwrite", "0.092 ms = 6.23 TB/s (0.779); r4", R3, "`test_gauss_draw_kl_golden`"),
    "mixture_nll_kernel": ("`MixturePrior.kl_divergence` (`bbb.py:31-37`)", "`8·D` B", "—", R3, "`test_mixture_prior_kernel`"),
    "l2_kernel": ("`l2_scale/2·‖p‖²` for plain parameters (`bbb.py:75-76`)", "`8·D` B", "—", R3, "`test_gauss_kl_large_and_l2`"),
    "reduce_finish_kernel": ("fixed-order fp64 finish of the workgroup partials", "latency", "—", R3, "all KL tests"),
    "local_reparam_fwd_kernel": ("`mean + sqrt(var)·eps` (`bbb_layers.py:70-80`)", "`12·D` B", "0.0477 ms = 6.01 TB/s (0.751); r4", R3, "`test_local_reparam_epilogue`"),
    "local_reparam_bwd_kernel": ("its backward (g_var = g·eps/(2√var))", "`12·D` B", "—", R3, "same test"),
    "var_operand_kernel": ("`clamp(x²)`, `clamp(softplus(rho)²)`, `softplus(rho)²` (+ backward)", "`8·n` / `12·n` B", "BBBConv2d 16→16 b128 fwd+bwd 0.39 vs 0.52 ms; r2", R3, "`test_var_operand_kernels`"),
    # ---- BBBLinear
    "lrt_partial_kernel": ("whole `BBBLinear.forward` (`bbb_layers.py:61-80`), narrow layers, batch ≤ 128", "`8·O·I` B + split-K partials",
                           "17–21 µs at the iWildCam head; r3 `r03_lrt_kernel_stats.csv`", R3, "`test_lrt_linear_forward`"),
    "lrt_wide_kernel": ("the same for `O·I ≥ 2²⁰`", "`8·O·I` B, `4·B·O·I` flop", "51.5 µs at 4096²/64 = 0.53 of fp32-MFMA peak; r3 `r03_lrt_kernel_stats.csv`", R3, "`test_lrt_linear_forward`"),
    "lrt_finish_kernel": ("split-K finish + bias + `sqrt(var)·eps`", "latency", "—", R3, "same"),
    "lrt_sigma_cache_kernel": ("σ², dσ²/dρ once per weight version", "`12·O·I` B", "33 µs at 4096²; r3 `r03_lrt_bench_sigma_cache.txt`", R3, "`test_lrt_sigma_cache_is_bit_identical`"),
    "lrt_bwd_prep_kernel": ("g_var, transposed copies, bias gradients", "latency", "—", R3, "`test_lrt_linear_backward`"),
    "lrt_bwd_w_kernel": ("both weight gradients of a wide layer", "`12·O·I` B, `4·B·O·I` flop", "0.36 of fp32-MFMA peak at 4096²/64; r3 `r03_lrt_kernel_stats.csv`", R3, "same"),
    "lrt_bwd_x_kernel": ("input gradient (general shapes)", "`8·O·I` B", "79 µs at 4096²; r3", R3, "same"),
    "lrt_bwd_x4_kernel": ("input gradient, float4 rows, mask-free FULL form", "`8·O·I` B", "64 µs at 4096² (backward 140 µs); r3 `r03_lrt_limiter_experiments.txt`", R3, "same"),
    "lrt_bwd_fused_kernel": ("all three matrix gradients in one pass, layers ≤ 2²⁰ weights", "`20·O·I` B", "18–19 µs at the reference's sizes; r3", R3, "same"),
    "lrt_bwd_x_finish_kernel": ("O-slice partials finish", "latency", "—", R3, "same"),
    # ---- BBBConv2d
    "conv_lrt_kernel": ("whole `BBBConv2d.forward` (`bbb_layers.py:146-154`) / its input gradient: dual-accumulator implicit GEMM", "`4·N·O·Ho·Wo·C·K²` flop",
                        "version 1 only: 40.7 µs vs 102.3 µs torch at 16→16 32×32 b128, 0.8× at 64→64 `r04_conv_lrt_fwd_v1_bench.txt`; HEAD (round-4 version 2 + round 5: flat 8-deep staging, pair layout, alternating operand sets, XCD mapping): **unmeasured**",
                        NEVER + "; `fused_conv=\"auto\"` keeps it OFF until `conv_profit.json` holds a device measurement", "`test_conv_lrt_forward`, `test_conv_lrt_backward`"),
    "conv_lrt_prep_kernel": ("σ², dσ²/dρ, weight matrices in staging order, per-phase matrices", "`~40·O·C·K²` B", "unmeasured", NEVER, "same"),
    "conv_lrt_wgrad_kernel": ("both weight-gradient convolutions, reduction over pixels", "same flops as forward; partials ≤ 512 blocks (round 5: −1/3 … −1/2 of round 4's bytes)",
                              "**unmeasured**", NEVER, "`test_conv_lrt_backward`"),
    "conv_lrt_gvar_kernel": ("first pass of the layer's backward: g_var = g·eps/(2√var) + the channel sums the bias gradients need (round 5; replaces `local_reparam_bwd` + two torch reductions)",
                             "`16·N·O·Ho·Wo` B", "unmeasured", NEVER, "`test_r5_conv_gvar_and_bias_gradients_in_one_pass`"),
    "conv_lrt_bias_finish_kernel": ("channel partials in order + ρ chain rule of the bias variance (`bbb_layers.py:147`)", "latency", "unmeasured", NEVER, "same test"),
    "conv_lrt_wgrad_finish_kernel": ("shares summed in order + ρ chain rule", "partials once", "unmeasured", NEVER, "same"),
    # ---- iVON
    "ivon_sample_kernel": ("`ivorn.py:102-115`", "`20·D` B", "0.0737 ms = 6.48 TB/s (0.81); r4", R3, "`test_ivon_golden_bit_exact`"),
    "ivon_update_kernel": ("`ivorn.py:66-89`", "`32·D` B", "0.1158 ms = 6.60 TB/s (0.825); r4", R3, "`test_ivon_golden_bit_exact` (bit-exact)"),
}


def kernels_in_sources():
    found = {}
    for fn in sorted(os.listdir(CSRC)):
        if not fn.endswith(".hip"):
            continue
        lines = open(os.path.join(CSRC, fn)).read().split("\n")
        for i, line in enumerate(lines):
            if "__global__" not in line:
                continue
            text = line + " " + (lines[i + 1] if i + 1 < len(lines) else "")
            m = re.search(r"void\s+([A-Za-z0-9_]+)\s*\(", text)
            if m:
                found[m.group(1)] = (fn, i + 1)
    return found


def table() -> str:
    src = kernels_in_sources()
    missing = sorted(set(src) - set(ROWS))
    stale = sorted(set(ROWS) - set(src))
    if missing or stale:
        raise SystemExit(f"tools/kernel_table.py: kernels without a row {missing}, rows without a kernel {stale}")
    out = ["| Kernel | Where | Replaces | Algorithmic work / launch | Last device measurement (round, file under `profiles/`) | On an MI355X at HEAD? | Parity test |",
           "|---|---|---|---|---|---|---|"]
    for name in ROWS:                                   # the dict's order: by family
        fn, line = src[name]
        rep, work, meas, status, test = ROWS[name]
        out.append(f"| `{name}` | `csrc/{fn}:{line}` | {rep} | {work} | {meas} | {status} | {test} |")
    return "\n".join(out)


def main():
    t = table()
    if "--write" in sys.argv:
        path = os.path.join(ROOT, "DESIGN.md")
        s = open(path).read()
        a, b = s.index(BEGIN), s.index(END)
        open(path, "w").write(s[:a] + BEGIN + "\n" + t + "\n" + s[b:])
        print("DESIGN.md updated")
    else:
        print(t)


if __name__ == "__main__":
    main()
