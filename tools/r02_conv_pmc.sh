#!/bin/bash
# PMC passes over one gather-GEMM layer (each counter set in its own run, --kernel-trace only)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/conv_pmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVES" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" \
         "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES"; do
  # (a fourth set with TA_* / TCP_*_sum / GRBM_GUI_ACTIVE made rocprofv3 abort and hang on this pool: not collected)
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p$i -o r -- python $R/tools/conv_pmc_probe.py $1 $2 > $O/p$i.log 2>&1
done
python - $O <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "spconv" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k} | {c} | {sum(v)/len(v):.1f} | n={len(v)}")
PY
find $O -name "*kernel_trace.csv" -delete
