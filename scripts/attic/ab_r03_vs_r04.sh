# A/B on one box: the round-3 library (ab_r03/, built from commit 3297886) against the current tree, alternating
for i in 1 2 3; do
  (cd ab_r03 && python scripts/bench_configs.py c3b 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('r03', d['config'][:30], d['kernel_ms_per_pass'], d['frac_of_8TBps'])")
  python scripts/bench_configs.py c3b 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('r04', d['config'][:30], d['kernel_ms_per_pass'], d['frac_of_8TBps'])"
done
