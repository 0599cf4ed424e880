import json, sys
d = json.load(open(sys.argv[1]))
print(d["value"], d["unit"], d["ms_per_step"], "ms/step")
print(d.get("roofline"))
print(d.get("stages"))
for k in d.get("kernels", [])[:12]: print(k)
for l in d.get("conv_layers", []): print(l)
