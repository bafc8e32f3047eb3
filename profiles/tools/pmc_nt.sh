#!/bin/bash
export DGTTA_LIB="$(dirname "$(readlink -f "$0")")/libdgtta_hip_diag.so"      # laboratory build (python -m dg_tta_amd.build --diag): the product library has no *_ABL / ROWS_VAR switches
# FETCH_SIZE / WRITE_SIZE of the row-reuse conv at 128^3 32->32 for the cache-policy experiments (DGTTA_ROWS_ABL 0 / 8 / 9)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for abl in 0 8; do
  export DGTTA_ROWS_ABL=$abl
  python3 $R/profiles/tools/kbench.py conv bf16 32 32 128 20 > $R/gpurun_out/nt_time_$abl.log 2>&1
  for ctr in FETCH_SIZE WRITE_SIZE; do
    KB_STATS=1 rocprofv3 --pmc $ctr --kernel-trace -d $R/gpurun_out/nt_${abl}_${ctr} -o pmc --output-format csv -- python3 $R/profiles/tools/kbench.py conv bf16 32 32 128 12 > $R/gpurun_out/nt_${abl}_${ctr}.log 2>&1 || exit 1
  done
done
cd $R
python3 - <<'PY'
import csv, glob
for abl in (0, 8):
    print("ABL", abl, open(f"gpurun_out/nt_time_{abl}.log").read().strip().splitlines()[-1])
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(f"gpurun_out/nt_{abl}_{ctr}/**/*counter_collection.csv", recursive=True)
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "conv3_rows" in r["Kernel_Name"] and r["Counter_Name"] == ctr]
        print("   ", ctr, "avg KiB", sum(vals) / max(len(vals), 1), "n", len(vals))
PY
