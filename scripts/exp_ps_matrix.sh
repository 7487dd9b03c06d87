R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
run() { echo "== $* $EXTRA"; env MRHIP_DEBUG=1 "$@" python bench.py --steps 3 --warmup 1 --samples 50000000 --no-cpu-baseline $EXTRA 2>&1 | grep -E "mrhip\]|metric|rror" | sed -e 's/.*"achieved": \([0-9.]*\).*"avg_launch_ms": \([0-9.]*\).*/   GBps=\1 ms=\2/' | sed -e 's/.*occ.CU=\([0-9]*\) regs=\([0-9]*\).*J=\([0-9]*\).*/   occ=\1 regs=\2 J=\3/' | cut -c1-220; }
for rep in 1 2; do for st in 0 1; do run MRHIP_PAIR_STRIP=$st; done; done
for st in 0 1; do run MRHIP_PAIR_STRIP=$st MRHIP_PAIR_C=4 MRHIP_PAIR_ROUNDS=2; done
