R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
run() { echo "== $* $EXTRA"; env MRHIP_DEBUG=1 "$@" python bench.py --steps 3 --warmup 1 --samples 50000000 --no-cpu-baseline --no-check $EXTRA 2>&1 | grep -E "mrhip\]|metric|rror" | sed -e 's/.*"achieved": \([0-9.]*\).*"avg_launch_ms": \([0-9.]*\).*/   GBps=\1 ms=\2/' | sed -e 's/.*occ.CU=\([0-9]*\) regs=\([0-9]*\).*J=\([0-9]*\).*/   occ=\1 regs=\2 J=\3/' | cut -c1-220; }
for r in 2 3 4; do for b in 0 3 4; do run MRHIP_PAIR_C=2 MRHIP_PAIR_ROUNDS=$r MRHIP_PAIR_BPC=$b; done; done
run MRHIP_PAIR_C=4 MRHIP_PAIR_ROUNDS=2 MRHIP_PAIR_BPC=2
run MRHIP_PAIR_C=3 MRHIP_PAIR_ROUNDS=2
run MRHIP_PAIR_C=1 MRHIP_PAIR_ROUNDS=4
