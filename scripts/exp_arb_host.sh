R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
{
/opt/rocm/lib/llvm/bin/clang++ -O3 -fPIC -std=c++17 scripts/ubench/arb_recurrence_core.cpp -o /tmp/arb_core; /tmp/arb_core 32 | tail -1; /tmp/arb_core 10 | tail -1; /tmp/arb_core 7 0.7 | tail -1
timeout 600 python -m pytest tests -x -q -m gpu -k "arbitrary or farrow or arb or cascade" 2>&1 | tail -3
} > gpurun_out/exp_arb_host.log 2>&1
