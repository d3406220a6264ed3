import json,sys
d=json.load(open(sys.argv[1])); k=d["kernel_ms_per_step"]
print(sys.argv[1].split('/')[-1], round(d["ms_per_step"],2), {x:round(k[x],2) for x in k if x.startswith("gemm")})
