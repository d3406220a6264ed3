#!/usr/bin/env python3
"""Registers / spills / scratch per kernel of one translation unit, with the Makefile's per-object flags:
tools/resource_usage.py attention [extra flags]   (hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = {"qkv_attention2": ["-fno-slp-vectorize"], "attention": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-mllvm", "-amdgpu-kernarg-preload-count=16"],
         "skinny": ["-mllvm", "-amdgpu-kernarg-preload-count=16"], "misc_kernels": ["-mllvm", "-amdgpu-kernarg-preload-count=16"],
         "layer_tail": ["-mllvm", "-structurizecfg-skip-uniform-regions=true"],
         "model_kernel": ["-fno-slp-vectorize", "-mllvm", "-structurizecfg-skip-uniform-regions=true"]}
unit = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "--cuda-device-only",
       "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(ROOT, "bert.cpp_amd", "csrc", unit + ".hip"), "-o", "/dev/null"] + FLAGS.get(unit, []) + sys.argv[2:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
for line in err.splitlines():
    m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        if "error" in line: print(line)
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0]}
    else:
        cur[k.split(" [")[0]] = v
        if k.startswith("LDS"):
            print(f"{cur['name'][:90]:90s} vgpr {cur.get('VGPRs'):>4s} agpr {cur.get('AGPRs'):>3s} spill {cur.get('VGPRs Spill'):>3s} scratch {cur.get('ScratchSize'):>4s} occ {cur.get('Occupancy')}")
