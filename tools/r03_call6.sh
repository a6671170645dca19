#!/bin/bash
# PMC passes over the dense-grid 32->32 layer (each counter set in its own run, --kernel-trace only)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_c6
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
         "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_MISC" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  for mode in dense_bias gather_bias; do
    EPRECON_D3_WV=4 timeout 100 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p${i}_$mode -o r -- python $R/tools/conv3d_probe.py 32 32 6 $mode > $O/p${i}_$mode.log 2>&1
    find $O/p${i}_$mode -name "*kernel_trace.csv" -delete
  done
done
python - $O <<'PY' | tee $O/summary.txt
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    mode = "dense" if "dense" in f else "gather"
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        big = int(r["Grid_Size"]) > 100000
        if (("conv3d_tile_kernel" in n and mode == "dense") or ("spconv_resident" in n and mode == "gather")) and big:
            acc[(mode, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (m, c), v in sorted(acc.items()):
    v = sorted(v)[-8:]   # the probe's launches (the largest grid repeated)
    print(f"{m} | {c} | {sum(v)/len(v):.1f} | n={len(v)}")
PY
