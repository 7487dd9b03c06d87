#!/bin/bash
# rocm-smi clocks / power sampled while the headline launch runs in a loop (and once after it).  usage: bash scripts/exp_smi_sample.sh
set -u
R="${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}"
cd "$R"
python3 scripts/exp_one.py --long 100000000 --reps 4000 > gpurun_out/smi_run.log 2>&1 &
PID=$!
sleep 30
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "sclk\|mclk\|power\|junction" | head -8 | tr '\n' ';'; echo; sleep 1; done > gpurun_out/smi_samples.txt
wait $PID
echo idle >> gpurun_out/smi_samples.txt
rocm-smi --showpower --showclocks 2>/dev/null | grep -i "sclk\|power" | head -4 | tr '\n' ';' >> gpurun_out/smi_samples.txt
cat gpurun_out/smi_samples.txt; tail -1 gpurun_out/smi_run.log
