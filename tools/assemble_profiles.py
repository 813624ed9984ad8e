"""Copy the judged summaries of a tools/collect_profiles.sh run from gpurun_out/prof_<tag>/ into profiles/<tag>_* and build the two
PMC digests (traffic per launch with the gfx950 FETCH_SIZE correction; MFMA / LDS / wave-time counters of the MFMA kernels alone)."""
import json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", f"prof_{tag}"), os.path.join(root, "profiles")
for f in ("bench_c2_bf16.json", "bench_c2_fp8.json", "bench_c3_batch4_bf16.json", "bench_c5shape_1536_bf16.json", "bench_c5_1536_fp8.json",
          "bench_under_rocprof_bf16.json", "bench_under_rocprof_fp8.json", "kernel_microbench.txt", "bench_c2_fp8mx.json", "bench_c5_1536_fp8mx.json",
          "bench_under_rocprof_fp8mx.json", "bench_c4_inpaint.json", "mx_gemm_microbench.txt"):
    if os.path.isfile(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f"{tag}_{f}"))
for f, g in (("stats_bf16_kernel_stats.csv", "bench_kernel_stats_bf16.csv"), ("stats_fp8_kernel_stats.csv", "bench_kernel_stats_fp8.csv"), ("stats_fp8mx_kernel_stats.csv", "bench_kernel_stats_fp8mx.csv"),
             ("stats_vae_kernel_stats.csv", "vae_decode_kernel_stats.csv")):
    if os.path.isfile(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f"{tag}_{g}"))
load = lambda f: json.load(open(os.path.join(src, f))) if os.path.isfile(os.path.join(src, f)) else {}


def traffic(fetch, write, names):
    out = {}
    for key, pat in names.items():
        fk = next((k for k in fetch if pat in k), None)
        wk = next((k for k in write if pat in k), None)
        if fk is None or wk is None:
            continue
        F, W = fetch[fk]["FETCH_SIZE"], write[wk]["WRITE_SIZE"]
        out[key] = {"kernel": fk, "launches": fetch[fk]["launches"], "FETCH_SIZE_KiB_avg": round(F, 1), "WRITE_SIZE_KiB_avg": round(W, 1),
                    "hbm_bytes_per_launch_corrected": int((2 * F + W) * 1024)}
    return out


names_bf16 = {"gemm": "gemm_pp_kernel", "attn": "attention_v3_kernel", "attn_short_sequences": "attention_fwd_kernel", "ln": "layernorm_mod_kernel", "rope": "qk_rmsnorm_rope_kernel"}
names_fp8 = {"gemm_pp_kernel": "gemm_pp_kernel", "attention_fp8_kernel": "attention_fp8_kernel", "prep_qk_kernel": "prep_qk_kernel",
             "prep_vt_kernel": "prep_vt_kernel", "quantize_rows": "quantize_rows_fp8_reg_kernel<30>", "ln": "layernorm_mod_kernel"}
t = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/collect_profiles.sh) over `python bench.py --steps 1 --warmup 0 --no-cpu-baseline "
               "--no-roofline-pass --no-graph`; per-kernel means over all launches of one image; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); "
               f"hbm_bytes = (2*FETCH + WRITE) KiB * 1024 (final kernels of round {tag})",
     "kernels": traffic(load("fetch_bf16.json"), load("write_bf16.json"), names_bf16),
     "fp8_run": {"source": "same with --precision fp8 (gemm_pp_kernel: e4m3 and the 7 % bf16 launches together, the counter records drop template arguments)", "kernels": traffic(load("fetch_fp8.json"), load("write_fp8.json"), names_fp8)},
     "fp8mx_run": {"source": "same with --precision fp8-mx (no quantize_rows launches; block-scaled operands)",
                   "kernels": traffic(load("fetch_fp8mx.json"), load("write_fp8mx.json"), {k: v for k, v in names_fp8.items() if k != "quantize_rows"})},
     "alone": {"source": "tools/prof_one.py (5 launches of one kernel, S = 4608 x 24 heads / 4608x21504x3072): attention_v3 (default) and attention.hip (RT_ATTN_V3=0) side by side",
               "kernels": {**traffic(load("pmc_attn_p4.json"), load("pmc_attn_p5.json"), {"attention_v3_kernel": "attention_v3_kernel"}),
                           **traffic(load("pmc_attnold_p4.json"), load("pmc_attnold_p5.json"), {"attention_fwd_kernel": "attention_fwd_kernel"}),
                           **traffic(load("pmc_gemm_p4.json"), load("pmc_gemm_p5.json"), {"gemm_pp_kernel<bf16>": "gemm_pp_kernel"})}}}
json.dump(t, open(os.path.join(dst, f"{tag}_traffic_pmc.json"), "w"), indent=1)


def counters(prefix, pat):
    acc = {}
    for i in (1, 2, 3):
        d = load(f"pmc_{prefix}_p{i}.json")
        k = next((k for k in d if pat in k), None)
        if k:
            acc.update(d[k])
    if acc:
        w, busy = acc.get("SQ_WAVE_CYCLES"), acc.get("SQ_VALU_MFMA_BUSY_CYCLES")
        gui = acc.get("GRBM_GUI_ACTIVE")
        if w and busy and gui:
            # SQ_*_CYCLES count quad-cycles per wave; MFMA_BUSY counts cycles summed over SIMDs; GRBM_GUI_ACTIVE sums the 8 XCDs
            acc["derived"] = {"mfma_pipe_busy_frac_of_simd_cycles": round(busy / (gui / 8 * 256 * 4), 4),
                              "wave_time_waiting_frac": round(acc.get("SQ_WAIT_ANY", 0) / w, 4), "wave_time_issue_stalled_frac": round(acc.get("SQ_WAIT_INST_ANY", 0) / w, 4),
                              "wave_time_issuing_frac": round(acc.get("SQ_ACTIVE_INST_ANY", 0) / w, 4), "gpu_cycles_per_launch": round(gui / 8)}
    return acc


m = {"source": "rocprofv3 --pmc (three passes per kernel, tools/collect_profiles.sh / tools/prof_one.py): per-launch means; attention S=4608 H=24 batch 1 with the key-split "
               "(attention_v3 = the default kernel for this shape, attention_fwd = csrc/attention.hip under RT_ATTN_V3=0); GEMM 4608x21504x3072",
     "kernels": {"attention_v3_kernel": counters("attn", "attention_v3_kernel"), "attention_fwd_kernel": counters("attnold", "attention_fwd_kernel"),
                 "gemm_pp_kernel<bf16>": counters("gemm", "gemm_pp_kernel")}}
json.dump(m, open(os.path.join(dst, f"{tag}_mfma_lds_pmc.json"), "w"), indent=1)
print(json.dumps({k: v.get("derived") for k, v in m["kernels"].items()}, indent=1))
print(json.dumps({k: v["hbm_bytes_per_launch_corrected"] for k, v in t["kernels"].items()}), json.dumps({k: v["hbm_bytes_per_launch_corrected"] for k, v in t["alone"]["kernels"].items()}))
