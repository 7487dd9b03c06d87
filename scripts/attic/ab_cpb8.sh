# A/B: fir_stream_kernel with 4 (product) vs 8 reads per block (libmrhip_cpb8.so, built with -DMRHIP_STREAM_CPB_MIN=8)
for i in 1 2; do
  python scripts/bench_configs.py c3b xstd 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('cpb4', d.get('name'), d.get('kernel_ms'), d.get('frac'), d.get('kernel'))"
  MRHIP_LIB_PATH=$GRAFT_REPO_ROOT/multirate.jl_amd/libmrhip_cpb8.so python scripts/bench_configs.py c3b xstd 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('cpb8', d.get('name'), d.get('kernel_ms'), d.get('frac'), d.get('kernel'))"
done
MRHIP_ENV_DYNAMIC=1 python scripts/exp_stream_rt.py --small-m > gpurun_out/small_m_cpb4.txt 2>&1
MRHIP_LIB_PATH=$GRAFT_REPO_ROOT/multirate.jl_amd/libmrhip_cpb8.so MRHIP_ENV_DYNAMIC=1 python scripts/exp_stream_rt.py --small-m > gpurun_out/small_m_cpb8.txt 2>&1
