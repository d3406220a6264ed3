#!/usr/bin/env python3
"""One line per bench JSON: ms per step and the GEMM kernels' ms (used by A/B runs of gemm256 variants: `python bench.py --config 3
--also ... > x.json; python tools/c3k.py x.json`)."""
import json,sys
d=json.load(open(sys.argv[1])); k=d["kernel_ms_per_step"]
print(sys.argv[1].split('/')[-1], round(d["ms_per_step"],2), {x:round(k[x],2) for x in k if x.startswith("gemm")})
