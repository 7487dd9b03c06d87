set -u
R="${GRAFT_REPO_ROOT:?}"; OUT="$R/gpurun_out/pmc160"; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
for lm in "160 147" "147 160"; do
  tag=$(echo $lm | tr ' ' '_')
  python3 "$R/scripts/exp_160_147_long.py" $lm 2>/dev/null | grep -v amdgpu
  rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU -d "$OUT/$tag" -o c -- python3 "$R/scripts/exp_160_147_long.py" $lm > "$OUT/$tag.log" 2>&1
  python3 - "$OUT/$tag" <<'PY'
import csv, glob, os, sys, collections
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "rational_opair" not in row["Kernel_Name"]: continue
        tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
c = {k: v / n[k] for k, v in tot.items()}
cyc = c["GRBM_GUI_ACTIVE"] / 8
print(sys.argv[1].split("/")[-1], "VALU busy %.3f" % (c["SQ_INSTS_VALU"] * 2 / 1024 / cyc), "LDS active %.3f" % (c["SQ_LDS_IDX_ACTIVE"] / 256 / cyc), "conflict share %.3f" % (c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]), "cycles/XCD %.0f" % cyc, "VALU instr %.3g" % c["SQ_INSTS_VALU"])
PY
done
find "$OUT" -name "*.csv" -size +2M -delete
