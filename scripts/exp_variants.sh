R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
{
for i in 1 2; do
echo "== base"; python scripts/bench_configs.py c3b xstd 2>/dev/null | cut -c1-60,150-330
echo "== scalar taps"; MRHIP_LIB_PATH=$R/build_exp/lib_scalar_taps.so python scripts/bench_configs.py c3b xstd 2>/dev/null | cut -c1-60,150-330
done
MRHIP_LIB_PATH=$R/build_exp/lib_scalar_taps.so timeout 600 python -m pytest tests -x -q -m gpu -k "tuned or config3 or sweep" 2>&1 | tail -2
} > gpurun_out/exp_scalar_taps.log 2>&1
