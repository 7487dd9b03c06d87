# Round profile of the headline bench on the GPU box: kernel stats, then two PMC passes (never combined
# with trace domains), all under gpurun_out/; summaries are copied into profiles/ by hand afterwards.
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out
CMD="python3 $R/bench.py --steps 2 --warmup 1 --samples 20000000 --no-cpu-baseline"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/prof_stats -o stats -- $CMD > $OUT/prof_stats.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/prof_fetch -o fetch -- $CMD > $OUT/prof_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/prof_write -o write -- $CMD > $OUT/prof_write.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/prof_sq -o sq -- $CMD > $OUT/prof_sq.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -d $OUT/prof_sq2 -o sq2 -- $CMD > $OUT/prof_sq2.log 2>&1
cd $R && python3 profiles/summarize_rocprof.py gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_sq gpurun_out/prof_sq2 > gpurun_out/prof_summary.json 2> gpurun_out/prof_summary.err
# keep only the small files (the merge back is capped at 64 MiB)
find gpurun_out/prof_* -name "*.csv" -size +3M -delete
