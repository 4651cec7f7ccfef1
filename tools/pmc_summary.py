#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter CSVs (one directory per pass) into one JSON: mean counter value per kernel
launch, HBM bytes corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE
tallies half the bytes of a wide coalesced read stream -> x2), next to the algorithmic bytes of the launch.

    python tools/pmc_summary.py OUT.json PASS_DIR [PASS_DIR ...]
"""
import csv, glob, json, os, re, sys
from collections import defaultdict

D50, D20, M, K, S = 23_880_950, 273_610, 8, 20, 30


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("bde::", "")


LRT = os.environ.get("BDE_PMC_LRT_SHAPE")          # "BxIxO" when the passes ran tools/lrt_bench.py on ONE shape


def algorithmic(kernel, grid):
    """Algorithmic bytes of one launch at ResNet-50 size; None when the launch is not that size."""
    if LRT and kernel.startswith("lrt_"):
        b, i, o = (int(v) for v in LRT.split("x"))
        w = 4 * i * o
        if kernel.startswith("lrt_wide_kernel"):
            return 2 * w + 4 * b * i                                      # W_mu, W_rho (or the cached sigma^2) once + x
        if kernel.startswith("lrt_bwd_w_kernel"):
            return 3 * w + 4 * b * (i + 2 * o)                            # rho (or d sigma^2) read, two gradients written
        if kernel.startswith("lrt_bwd_x_kernel"):
            return 2 * w + 8 * o * 64                                     # W_mu, W_rho (or sigma^2) once + the transposed g copies
        if kernel.startswith("lrt_bwd_x4_kernel"):
            return 2 * w + 8 * b * o                                      # W_mu, W_rho (or sigma^2) once + g, gvar
        if kernel.startswith("lrt_sigma_cache_kernel"):
            return 3 * w
        return None
    big = grid >= 200_000          # every ResNet-50-size launch of this library uses >= 1024 workgroups of 256
    if not big:
        return None
    table = {"svgd_gram_kernel<2>": 4 * M * D50, "svgd_combine_kernel<8, true>": 12 * M * D50,
             "svgd_combine_seg_kernel<8>": 12 * M * D50,
             "svgd_fused_kernel<8, 0, true, false>": (12 * M + 8) * D50, "svgd_fused_kernel<8, 0, false, false>": (12 * M + 8) * D50,
             "svgd_fused_kernel<8, 0, true, true>": (12 * M + 8) * D50, "svgd_fused_kernel<8, 0, false, true>": (12 * M + 8) * D50,
             "svgd_gather_seg_kernel": 8 * M * D50,
             "svgd_apply_sgd_kernel": (12 * M + 8) * D50,
             "swag_update_kernel": 24 * D50, "swag_update_kernel<0>": 24 * D50, "swag_update_kernel<12>": 24 * D50,   # <..>: round 3's names
             "swag_sample_kernel<true>": 4 * D50 * (K + 3), "swag_sample_kernel<true, 0>": 4 * D50 * (K + 3),
             "swag_sample_kernel<true, 12>": 4 * D50 * (K + 3),
             "swag_sample_batched_kernel<true>": 4 * D50 * (K + 2 + S),
             "swag_sample_batched_dma_kernel<true, 7>": 4 * D50 * (K + 2 + S),
             "gauss_draw_fwd_kernel<true>": 12 * D50, "gauss_draw_bwd_kernel<true, true>": 24 * D50,
             "gauss_kl_kernel<true, true>": 24 * D50, "gauss_kl_kernel<true, false>": 16 * D50,
             "local_reparam_fwd_kernel<true>": 12 * D50, "ivon_sample_kernel<true>": 20 * D50, "ivon_update_kernel": 32 * D50}
    return table.get(kernel)


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(list))
    grids = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = short(row["Kernel_Name"])
                    if not ("svgd" in k or "swag" in k or "gauss" in k or "ivon" in k or "local_reparam" in k or "philox" in k
                            or "lrt_" in k):
                        continue
                    g = int(row["Grid_Size"])
                    key = (k, g >= 200_000)
                    acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
                    grids[key] = g
    res = {"note": "rocprofv3 --pmc, one pass per counter group; means per launch. FETCH_SIZE/WRITE_SIZE in KiB; read bytes = "
                   "2 * FETCH_SIZE * 1024 (gfx950 correction, MI355X_MICROARCH.md section HBM); write bytes = WRITE_SIZE * 1024.",
           "kernels": {}}
    for (k, big), ctrs in sorted(acc.items()):
        e = {c: sum(v) / len(v) for c, v in ctrs.items()}
        e["launches_seen"] = max(len(v) for v in ctrs.values())
        e["grid_threads"] = grids[(k, big)]
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_read_bytes_corrected"] = 2 * e["FETCH_SIZE"] * 1024
            e["hbm_write_bytes"] = e["WRITE_SIZE"] * 1024
            e["hbm_traffic_bytes_per_launch"] = e["hbm_read_bytes_corrected"] + e["hbm_write_bytes"]
            alg = algorithmic(k, grids[(k, big)])
            if alg:
                e["algorithmic_bytes_per_launch"] = alg
                e["traffic_over_algorithmic"] = round(e["hbm_traffic_bytes_per_launch"] / alg, 4)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "GRBM_GUI_ACTIVE" in e:
            simd_cycles = e["GRBM_GUI_ACTIVE"] / 8 * 8 * 32 * 4          # cycles per XCD x 1024 SIMDs
            e["mfma_busy_fraction_of_simd_cycles"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles, 4)
            if "SQ_ACTIVE_INST_VALU" in e:
                e["valu_active_fraction_of_simd_cycles"] = round(4 * e["SQ_ACTIVE_INST_VALU"] / simd_cycles, 4)
        res["kernels"][k + ("" if big else " (small launch)")] = e
    json.dump(res, open(out, "w"), indent=1)
    print("wrote", out, "with", len(res["kernels"]), "kernels")


if __name__ == "__main__":
    main()
