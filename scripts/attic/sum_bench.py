import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("streamed_1e6_chunks",{}).get("frac"))
for r in d["configs"]:
    print(r["name"][:72].ljust(72), r["kernel"][:20].ljust(20), r["kernel_ms"], r["wall_ms"], r["frac"], r["frac_wall"], round(r["wall_ms"]/max(r["kernel_ms"],1e-9),2))
print(list(d.keys())[-3:])
