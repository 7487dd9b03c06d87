#!/bin/bash
# A/B counter passes on the headline shape (developer build, 64 ch x 2e7 samples per launch): arm 0 / arm 1 = the
# environment variable named by AB_VAR (default MRHIP_OPAIR_C) set to AB_0 / AB_1.
# usage: AB_VAR=MRHIP_OPAIR_C AB_0=6 AB_1=4 bash scripts/exp_pmc_ab.sh <tag>    (GRAFT_REPO_ROOT must be set)
set -u
R="${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}"
TAG="${1:?usage: exp_pmc_ab.sh <tag>}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export MRHIP_LIB_PATH="$R/multirate.jl_amd/libmultirate_hip_fast.so"
cd /tmp && export TMPDIR=/tmp
for arm in 0 1; do
  v=AB_$arm; export "${AB_VAR:-MRHIP_OPAIR_C}=${!v:-6}"
  rocprofv3 --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d "$OUT/sq_$arm" -o sq -- python3 "$R/scripts/exp_one.py" > "$OUT/sq_$arm.log" 2>&1
  rocprofv3 --output-format csv --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -d "$OUT/sq2_$arm" -o sq2 -- python3 "$R/scripts/exp_one.py" > "$OUT/sq2_$arm.log" 2>&1
  rocprofv3 --output-format csv --pmc SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_WAIT_INST_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS -d "$OUT/stall_$arm" -o stall -- python3 "$R/scripts/exp_one.py" > "$OUT/stall_$arm.log" 2>&1
  rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/fetch_$arm" -o fetch -- python3 "$R/scripts/exp_one.py" > "$OUT/fetch_$arm.log" 2>&1
  rocprofv3 --output-format csv --pmc SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC -d "$OUT/misc_$arm" -o misc -- python3 "$R/scripts/exp_one.py" > "$OUT/misc_$arm.log" 2>&1
done
cd "$R" && python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(os.path.join(out, "*_[01]"))):
    tot = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "rational_o" not in row["Kernel_Name"]: continue
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
    print(os.path.basename(d), {k: round(v / max(n[k], 1)) for k, v in sorted(tot.items())}, "dispatches", max(n.values()) if n else 0)
PY
find "$OUT" -name "*.csv" -size +3M -delete; find "$OUT" -name "*.db" -delete
