# Round profile of the headline bench on the GPU box: rocprofv3 kernel stats of the DEFAULT command, then PMC passes
# (never combined with trace domains) on a shortened run of the same launch shape, all under gpurun_out/; summaries are
# copied into profiles/ by hand afterwards.   usage: bash scripts/profile_headline.sh [extra bench.py args, e.g. --chunk 1000000]
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out
EXTRA="$@"
rm -rf $OUT/prof_full $OUT/prof_fetch $OUT/prof_write $OUT/prof_sq $OUT/prof_sq2
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/prof_full -o full -- python3 $R/bench.py --no-streamed $EXTRA > $OUT/prof_full_bench.json 2> $OUT/prof_full.log
CMD="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-streamed $EXTRA"
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/prof_fetch -o fetch -- $CMD > $OUT/prof_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/prof_write -o write -- $CMD > $OUT/prof_write.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/prof_sq -o sq -- $CMD > $OUT/prof_sq.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -d $OUT/prof_sq2 -o sq2 -- $CMD > $OUT/prof_sq2.log 2>&1
cd $R && python3 profiles/summarize_rocprof.py gpurun_out/prof_full gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_sq gpurun_out/prof_sq2 > gpurun_out/prof_summary.json 2> gpurun_out/prof_summary.err
# keep only the small files (the merge back is capped at 64 MiB)
find gpurun_out/prof_* -name "*.csv" -size +3M -delete
