#!/usr/bin/env python3
"""gpurun_out/*_<tag>.* (written by tools/gpu_round_check.sh on the GPU box) -> the committed summaries under profiles/.
usage: python tools/assemble_profiles.py [tag]"""
import json
import os
import subprocess
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r6"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = os.path.join(root, "gpurun_out")
P = os.path.join(root, "profiles")
commit = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], text=True).strip()


def read(name):
    with open(os.path.join(g, name)) as f:
        return f.read()


def table(txt):
    lines = [l for l in txt.splitlines() if l.strip()]
    hdr = lines[0].split()
    rows = {}
    for l in lines[1:]:
        parts = l.split()
        n = len(hdr) - 1
        rows[" ".join(parts[:-n])] = dict(zip(hdr[1:], parts[-n:]))
    return hdr, rows


def find(rows, key):
    return next(v for k, v in rows.items() if key in k)


pmc1, pmc2, pmc3, pmc3c3 = (read(f"pmc{n}_{tag}.txt") for n in ("1", "2", "3", "3c3"))
h1, r1 = table(pmc1)
h3, r3 = table(pmc3)
busy = {}
for k in ("model_kernel", "layer_tail", "qkv_attention2"):
    try:
        m = float(find(r1, k)[next(c for c in h1 if "MFMA_BUSY" in c)])
        gui = float(find(r3, k)[next(c for c in h3 if "GRBM" in c)]) / 8          # summed over the 8 XCDs
        busy[k] = (m, gui, m / (1024 * gui))
    except StopIteration:
        pass
bench_text = [l for l in read(f"bench_{tag}.log").splitlines() if l.startswith("{")][-1]      # (RCCL prints its banner to stdout too)
line = json.loads(bench_text)                  # the ONE compact line (what the driver parses)
assert len(bench_text) < 6144, len(bench_text)
bench = json.loads(read(f"bench_detail_{tag}.json"))      # the full detail the same run wrote beside it (bench_detail.json)
hdr = (f"# rocprofv3 --kernel-trace --stats -- python bench.py --steps 100 --warmup 10 --repeat 2 --no-cpu-baseline --also   (MI355X, round {tag[1:]}, commit {commit}; tools/gpu_round_check.sh)\n"
       f"# config: all-MiniLM-L6-v2 dims f16, 256 x 128 tokens per step, 6 layers: ONE launch for all layers (model_kernel: a workgroup per window, qkv_attention2 + layer_tail as phases)\n"
       f"# same box, un-profiled default bench line: {bench['value'] / 1e3:.1f} k sentences/s (device-resident; host to host {bench['host_api']['value'] / 1e3:.1f} k; bert_eval_batch {bench['eval_batch_api']['value'] / 1e3:.1f} k), its own time of {bench['roofline']['kernel']} ({bench['roofline']['timing']}): {bench['roofline']['avg_launch_us']:.1f} us per launch\n"
       "# MFMA busy share = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): " +
       ", ".join(f"{k} {v[0] / 1e6:.2f} M / (1024 x {v[1] / 1e3:.1f} k) = {v[2]:.2f}" for k, v in busy.items()) + " (two launches per layer, earlier this round: layer_tail 0.41, qkv_attention2 0.34)\n")
c3 = bench["also"]["config3"]
with open(os.path.join(P, f"{tag}_kernel_stats.txt"), "w") as f:
    f.write(hdr + read(f"stats_{tag}.txt") +
            f"\n# --config 3 (bert-base dims q4_1 expanded to f16 at load, 512 x 512 tokens per step, 12 layers): rocprofv3 --kernel-trace --stats -- python bench.py --config 3 --steps 3 --warmup 1 --repeat 1 --no-cpu-baseline --also\n"
            f"# gemm256_kernel<EPI, WT, LN>: EPI 0 = QKV (bias), 1 = FFN up (bias + GELU), 2 = attention output and FFN down (bias + residual); WT 0 = f16 image, 2 = q4_1 planes (layer 0's QKV); LN = the LayerNorm fold: 1 the input rows' statistics ride in (QKV of layers >= 1, FFN up), 4 / 6 = 4 the rows' partial sums go out + 2 the residual is rebuilt from the un-normalised rows (attention output, FFN down; 4 alone = layer 0's attention output); same box un-profiled: {c3['value']:.0f} sentences/s = {c3['path_mfma_frac']:.3f} of the MFMA peak\n" +
            read(f"stats_config3_{tag}.txt") +
            (f"\n# --config 33: the same model and batch with BERT_HIP_Q4=fused — the matrices stay 4-bit in HBM, gemm256_kernel<EPI, 2> (q4_1 planes) dequantises the blocks in its tile load; same box un-profiled: "
             f"{bench['also']['config3_fused']['value']:.0f} sentences/s\n" + read(f"stats_config33_{tag}.txt") if os.path.exists(os.path.join(g, f"stats_config33_{tag}.txt")) and "config3_fused" in bench["also"] else ""))
with open(os.path.join(P, f"{tag}_pmc.txt"), "w") as f:
    f.write(f"# rocprofv3 --pmc <set> --kernel-trace -- python bench.py --steps 2 --warmup 1 --repeat 1 --no-cpu-baseline --also   (separate runs per counter set; mean per dispatch; commit {commit})\n"
            "# SQ_* in quad-cycles except SQ_VALU_MFMA_BUSY_CYCLES (cycles); FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE must be doubled on gfx950 (MI355X_MICROARCH.md, HBM section)\n" +
            pmc1 + "\n" + pmc2 + "\n" + pmc3 + "\n# --config 3 (bert-base dims q4_1, 512 x 512 tokens): FETCH_SIZE / GRBM_GUI_ACTIVE / WRITE_SIZE\n" + pmc3c3)
with open(os.path.join(P, "traffic.json"), "w") as f:
    f.write(read(f"traffic_{tag}.json"))
with open(os.path.join(P, f"{tag}_bench_line.json"), "w") as f:
    f.write(bench_text + "\n")
with open(os.path.join(P, f"{tag}_bench_detail.json"), "w") as f:
    json.dump(bench, f, indent=1)
    f.write("\n")
# the line the bench printed UNDER rocprofv3 (the process whose kernel trace is {tag}_kernel_stats.txt): like with like
prof_line = read(f"bench_prof_line_{tag}.txt").strip()
if prof_line.startswith("{"):
    with open(os.path.join(P, f"{tag}_bench_line_profiled.json"), "w") as f:
        f.write(prof_line + "\n")
    pl = json.loads(prof_line)
    print("under rocprofv3:", round(pl["value"]), "sentences/s,", pl["roofline"]["kernel"], round(pl["roofline"]["avg_launch_us"], 1), "us per launch")
cal = [l for l in read(f"gemm_calibration_{tag}.txt").splitlines() if "amdgpu.ids" not in l]
with open(os.path.join(P, f"{tag}_gemm_calibration.txt"), "w") as f:
    f.write(f"# python tools/gemm_calibration.py on the box of this round check (commit {commit}): torch.matmul = hipBLASLt (Custom_Cijk_..._MT256x256x64_MI16x16x1 on the bert-base shapes), f16 in, f16 out, NO bias / GELU / residual;\n"
            f"# beside it on the same box: profiles/{tag}_kernel_stats.txt, config 3: gemm256_kernel<0> QKV + bias, <1> FFN up + bias + GELU, <2> attention-out / FFN down + bias + residual (average of the two)\n" +
            "\n".join(cal) + "\n")
print("profiles/ refreshed at", commit, {k: round(v[2], 3) for k, v in busy.items()})
for k, v in [("config1", bench)] + [kv for kv in bench["also"].items() if "value" in kv[1] and "ms_per_step" in kv[1]]:
    print(f"{k:22s} {v['value']:12.0f} /s  {v['ms_per_step']:8.3f} ms  frac {v.get('path_mfma_frac', 0):.3f}  host {(v.get('host_api') or {}).get('value', 0):10.0f}  cpu {(v.get('cpu_baseline') or {}).get('value', 0):8.2f}  x{v.get('speedup_vs_cpu', 0):.0f}  cos {v.get('mean_cosine_vs_cpu')}")
enc = bench["also"].get("encode_batch_text")
if enc:
    print(f"encode_batch_text      {enc['value']:12.0f} texts/s  {enc['ms_per_call']:8.3f} ms per call of {enc['n_texts']} texts, {enc['mean_tokens_per_text']:.1f} tokens per text, {enc['host_threads']} host threads")
lat = bench["also"].get("latency_b1", {})
for k, v in lat.items():
    if isinstance(v, dict):
        print(f"latency {k:12s} median {v['median_us']:7.1f} us (p10 {v['p10_us']:.1f}, p90 {v['p90_us']:.1f})", f"{v['launches']} launches, kernels {v['kernel_us']}" if "launches" in v else f"{v['sentences']} sentences, {v['tokens']} tokens")
