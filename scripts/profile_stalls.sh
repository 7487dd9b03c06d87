# PMC passes on the stall / FIFO counters of the headline kernel (never combined with trace domains)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out
CMD="python3 $R/bench.py --steps 2 --warmup 1 --samples 20000000 --no-cpu-baseline"
rm -rf $OUT/prof_st1 $OUT/prof_st2 $OUT/prof_st3
rocprofv3 --output-format csv --pmc SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/prof_st1 -o st1 -- $CMD > $OUT/prof_st1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS -d $OUT/prof_st2 -o st2 -- $CMD > $OUT/prof_st2.log 2>&1
rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_INSTS -d $OUT/prof_st3 -o st3 -- $CMD > $OUT/prof_st3.log 2>&1
cd $R && python3 profiles/summarize_rocprof.py gpurun_out/prof_st1 gpurun_out/prof_st2 gpurun_out/prof_st3 > gpurun_out/prof_stalls.json 2> gpurun_out/prof_stalls.err
find gpurun_out/prof_st* -name "*.csv" -size +3M -delete
