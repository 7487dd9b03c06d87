# C3b (fir_stream_kernel, 1//4, 128 taps, ComplexF32, 256 ch x 1e6): compute waves per workgroup x steps per tile x workgroups per CU
for w in 2 3 4 5 7; do for j in 0 2 4 8 16; do for b in 0 4; do
  r=$(MRHIP_STREAM_WAVES=$w MRHIP_STREAM_J=$j MRHIP_STREAM_BPC=$b python scripts/bench_configs.py c3b 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print(d.get('kernel_ms_per_pass'), d.get('frac_of_8TBps'))")
  echo "waves=$w J=$j bpc=$b: $r"
done; done; done
