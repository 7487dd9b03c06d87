mkdir -p gpurun_out/r06/ab2
for i in 1 2 3; do
  timeout -k 10 300 python scripts/bench_configs.py c4 > gpurun_out/r06/ab2/inline_$i.jsonl 2>/dev/null
  MRHIP_SCHED_BESIDE_LANE=1 timeout -k 10 300 python scripts/bench_configs.py c4 > gpurun_out/r06/ab2/beside_$i.jsonl 2>/dev/null
done
python3 - <<'P'
import json,glob
for fn in sorted(glob.glob("gpurun_out/r06/ab2/*.jsonl")):
    for ln in open(fn):
        if ln.startswith("{") and '"C4 ' in ln:
            d=json.loads(ln); print(fn.split("/")[-1], d["kernel"], "kernel_ms", d["kernel_ms_per_pass"], "wall", d["wall_ms_per_pass_incl_host"], "memo_kernel", d.get("kernel_ms_with_schedule_memo"), "memo_wall", d.get("wall_ms_with_schedule_memo"))
P
