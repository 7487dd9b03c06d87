#!/bin/bash
# The table in profiles/<round>/ratio_survey.txt: which kernel serves which ratio / decimation / filter length / dtype and how
# fast (scripts/exp_ratio_survey.py rows), with the kernels of round 1 beside where a switch exists.
# usage: bash scripts/survey_all.sh > gpurun_out/ratio_survey.txt      (on the GPU box, from the repo root)
S="python3 scripts/exp_ratio_survey.py"
f() { grep -v amdgpu.ids; }
echo "# scripts/exp_ratio_survey.py, product library, 64 ch x 2e6 samples per launch; % of the 8 TB/s HBM roofline (algorithmic bytes)"
echo "# Float32, 24 taps per phase"; $S 2>&1 | f
echo "# the same with the output-pair kernel switched off (MRHIP_OPAIR=0): the kernels these ratios ran on before"
MRHIP_OPAIR=0 $S 441/160 160/441 80/441 3/17 7/3 3/7 2/1 3/1 5/2 2/5 25/12 12/25 2>&1 | f
echo "# Float32, 4 / 56 / 64 taps per phase"; $S --taps-per-phase 4 147/160 2/1 2>&1 | f; $S --taps-per-phase 56 147/160 160/147 2/1 2>&1 | f; $S --taps-per-phase 64 147/160 160/147 2/1 2>&1 | f
echo "# decimators, Float32, 96 taps"; $S --taps-per-phase 96 1/2 1/3 1/4 1/5 1/6 1/7 1/8 1/10 1/12 1/16 2>&1 | f
echo "# the same with the streaming kernel switched off (MRHIP_STREAM=0)"; MRHIP_STREAM=0 $S --taps-per-phase 96 1/2 1/3 1/4 1/5 1/6 1/7 1/8 1/10 1/12 1/16 2>&1 | f
echo "# decimators, Float32, 128 taps (1//64: 512 taps too)"; $S --taps-per-phase 128 1/17 1/20 1/25 1/32 1/50 1/64 2>&1 | f; $S --taps-per-phase 512 1/64 2>&1 | f
echo "# short filters, Float32 (2, 5, 15 taps), and with the streaming kernel switched off"; for t in 2 5 15; do $S --taps-per-phase $t 1/1 1/2 1/4 2>&1 | f; done; MRHIP_STREAM=0 $S --taps-per-phase 2 1/1 1/2 2>&1 | f
echo "# long filters (1024 taps), and with the streaming kernel switched off"; $S --taps-per-phase 1024 1/1 1/4 2>&1 | f; MRHIP_STREAM=0 $S --taps-per-phase 1024 1/1 1/4 2>&1 | f
echo "# Float64 (samples and taps): rational 24 / 36 / 48 taps per phase, decimators 96 taps"
$S --dtype float64 147/160 160/147 441/160 2/1 4/1 2>&1 | f; $S --dtype float64 --taps-per-phase 36 147/160 2/1 2>&1 | f; $S --dtype float64 --taps-per-phase 48 147/160 2/1 2>&1 | f
$S --dtype float64 --taps-per-phase 96 1/1 1/2 1/3 1/4 1/8 2>&1 | f; MRHIP_STREAM=0 $S --dtype float64 --taps-per-phase 96 1/1 1/4 2>&1 | f
echo "# ComplexF32"; $S --dtype complex64 147/160 160/441 441/160 4/1 2>&1 | f; $S --dtype complex64 --taps-per-phase 56 147/160 2>&1 | f
echo "# ComplexF64 (rational and interpolating: output-pair kernel; MRHIP_OPAIR=0: poly_phase_stationary_kernel)"
$S --dtype complex128 147/160 160/147 2/1 441/160 2>&1 | f; MRHIP_OPAIR=0 $S --dtype complex128 147/160 2/1 2>&1 | f; $S --dtype complex128 --taps-per-phase 96 1/1 1/2 1/4 2>&1 | f
echo "# interpolators, Float32, 32 taps per phase"; $S --taps-per-phase 32 2/1 3/1 4/1 8/1 16/1 2>&1 | f
