#!/bin/bash
# PMC passes (one counter group per run, kernel-trace only) over profiles/tools/warpbench.py: where do the logit warps spend their time?
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum" "TA_BUSY_avr TA_TA_BUSY_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $R/gpurun_out/pmc_warp_$i -o pmc --output-format csv -- python3 $R/profiles/tools/warpbench.py > $R/gpurun_out/pmc_warp_$i.log 2>&1 || echo "group $i ($grp) failed"
done
cd $R
python - <<'PY'
import csv, glob, statistics, collections
out = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/pmc_warp_*/**/*counter_collection.csv", recursive=True):
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "warp" in k:
            vals[(k.split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in vals.items():
        out[k][c] = statistics.median(v)
for k, d in out.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} {v:16.0f}")
PY
