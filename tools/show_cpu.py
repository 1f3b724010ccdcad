import json, sys
d = json.load(open(sys.argv[1])); c = d["cpu_baseline"]
print(d["value"], d["ms_per_step"]); print({k: v for k, v in c.items() if k != "flavours"}); print(c["flavours"])
