#!/usr/bin/env python3
"""Per-kernel HIP-event times of one bench config (quick A/B helper for kernel work)."""
import json, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
class A: steps=int(os.environ.get("STEPS","5")); warmup=2; repeat=int(os.environ.get("REPEAT","3")); also=False; config=1; gpus=1; inproc=False; no_cpu_baseline=True
cfg=int(sys.argv[1]) if len(sys.argv)>1 else 1
torch.cuda.set_device(0); dev=torch.device("cuda",0)
with tempfile.TemporaryDirectory() as d:
    r=bench.run_config(cfg,A,0,1,dev,None,torch,d)
    roof,bd=bench.kernel_roofline(r,torch,dev,steps=5)
    print(f"cfg{cfg} {r['value']:.0f} sent/s {r['ms_per_step']:.3f} ms/step", json.dumps(bd))
