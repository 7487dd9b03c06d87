#!/bin/bash
# A/B on ONE box: the round-4 library and harness (ab_r04/, built from commit 632c30f) against the current tree, alternating.
R="${GRAFT_REPO_ROOT:-.}"; cd "$R"
for i in 1 2 3; do
  (cd ab_r04 && python bench.py --no-configs --no-cpu-baseline --no-streamed --no-check 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r04 headline', d['roofline']['avg_launch_ms'], d['roofline']['frac'])")
  python bench.py --no-configs --no-cpu-baseline --no-streamed --no-check 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r05 headline', d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done
for i in 1 2; do
  (cd ab_r04 && python scripts/bench_configs.py c1 c2 c3a c5 xmix64 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('r04', d['config'][:40].ljust(40), d['kernel_ms_per_pass'], d['frac_of_8TBps'])")
  python scripts/bench_configs.py c1 c2 c3a c5 xmix64 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('r05', d['config'][:40].ljust(40), d['kernel_ms_per_pass'], d['frac_of_8TBps'])"
done
