#!/bin/bash
# Round profile of the headline bench on the GPU box.  Everything is written under gpurun_out/<tag>/; the summaries to be
# judged are copied into profiles/ by hand afterwards.
#   1. rocprofv3 --kernel-trace --stats of the headline (python3 bench.py --no-streamed --no-configs): per-kernel time, duration agreement
#   2. the same for the chunked stream (bench.py --chunk 1000000) and for the rows bench.py reports as `configs`
#   3. PMC passes (never combined with trace domains, separate runs): FETCH_SIZE, WRITE_SIZE, two SQ sets, stall set
# usage: bash scripts/profile_round.sh <tag> [stats|pmc|all]
set -u
R="${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}"
TAG="${1:?usage: profile_round.sh <tag> [stats|pmc|all]}"
WHAT="${2:-all}"
[ -f "$R/bench.py" ] || { echo "no bench.py under $R" >&2; exit 1; }
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
if [ "$WHAT" = stats ] || [ "$WHAT" = all ]; then
  rm -rf "$OUT/prof_full" "$OUT/prof_chunk"
  # One launch size per kernel name and run, so that AverageNs in the stats IS the per-launch duration the JSON line quotes:
  # the headline alone (--no-streamed --no-configs), the chunked stream alone, the `configs` rows in a run of their own
  # (C2 apart: it runs the headline's instantiation at another size)
  rm -rf "$OUT/prof_cfg" "$OUT/prof_c2"
  rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/prof_full" -o full -- python3 "$R/bench.py" --no-streamed --no-configs > "$OUT/bench_under_rocprof.json" 2> "$OUT/prof_full.log"
  rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/prof_chunk" -o chunk -- python3 "$R/bench.py" --chunk 1000000 --no-cpu-baseline > "$OUT/bench_chunked_under_rocprof.json" 2> "$OUT/prof_chunk.log"
  rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/prof_cfg" -o cfg -- python3 "$R/scripts/bench_configs.py" c3a c3b c4 c4f c5 xmix64 af ms xdec xlarge > "$OUT/configs_under_rocprof.jsonl" 2> "$OUT/prof_cfg.log"
  rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/prof_c2" -o c2 -- python3 "$R/scripts/bench_configs.py" c2 > "$OUT/c2_under_rocprof.jsonl" 2> "$OUT/prof_c2.log"
  # BASELINE configs 1 and 2 as stated: one 1e6-sample call; one launch per arriving chunk; the ring (ONE resident launch per pass: its
  # duration in the trace is the whole pass)
  rm -rf "$OUT/prof_c1" "$OUT/prof_c2s" "$OUT/prof_c2r"
  rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/prof_c1" -o c1 -- python3 "$R/scripts/bench_configs.py" c1 > "$OUT/c1_under_rocprof.jsonl" 2> "$OUT/prof_c1.log"
  rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/prof_c2s" -o c2s -- python3 "$R/scripts/bench_configs.py" c2s > "$OUT/c2s_under_rocprof.jsonl" 2> "$OUT/prof_c2s.log"
  rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/prof_c2r" -o c2r -- python3 "$R/scripts/bench_configs.py" c2r > "$OUT/c2r_under_rocprof.jsonl" 2> "$OUT/prof_c2r.log"
fi
if [ "$WHAT" = pmc ] || [ "$WHAT" = all ]; then
  for p in fetch write fetch_chunk write_chunk sq sq2 stall; do rm -rf "$OUT/prof_$p"; done
  # one-call launches (49.12 GB per launch) and 1e6-sample launches (491 MB) of the same kernel, each in its own pass
  rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/prof_fetch" -o fetch -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-streamed --no-configs > "$OUT/prof_fetch.log" 2>&1
  rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/prof_write" -o write -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-streamed --no-configs > "$OUT/prof_write.log" 2>&1
  rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/prof_fetch_chunk" -o fetch -- python3 "$R/bench.py" --chunk 1000000 --steps 1 --warmup 1 --no-cpu-baseline > "$OUT/prof_fetch_chunk.log" 2>&1
  rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/prof_write_chunk" -o write -- python3 "$R/bench.py" --chunk 1000000 --steps 1 --warmup 1 --no-cpu-baseline > "$OUT/prof_write_chunk.log" 2>&1
  rocprofv3 --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d "$OUT/prof_sq" -o sq -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-streamed --no-configs > "$OUT/prof_sq.log" 2>&1
  rocprofv3 --output-format csv --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -d "$OUT/prof_sq2" -o sq2 -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-streamed --no-configs > "$OUT/prof_sq2.log" 2>&1
  rocprofv3 --output-format csv --pmc SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_WAIT_INST_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS -d "$OUT/prof_stall" -o stall -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-streamed --no-configs > "$OUT/prof_stall.log" 2>&1
fi
cd "$R" && python3 profiles/summarize_rocprof.py gpurun_out/"$TAG"/prof_* > "gpurun_out/$TAG/summary.json" 2> "gpurun_out/$TAG/summary.err"
# keep only the small files (the merge back is capped at 64 MiB)
find "$OUT" -name "*.csv" -size +3M -delete
find "$OUT" -name "*.db" -delete
