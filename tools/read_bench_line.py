import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d["roofline"]; a=dict(r["also"]); a[r["kernel"]]=r
print(d["value"], d["ms_per_step"], "conv_tiles", a["k_conv3x3_tiles"]["total_us_per_step"], a["k_conv3x3_tiles"]["frac"])
