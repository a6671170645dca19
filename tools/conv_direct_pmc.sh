#!/bin/bash
# rocprofv3 --pmc passes over tools/conv_shapes_ab.py (the 3x3x3 shapes of a cfg4 fragment): clock, MFMA pipe busy cycles, instruction
# mix and waits per launch of the direct gather kernel -> gpurun_out/r03_pmc/ (summary on stdout; profiles/r03/conv_direct_pmc.txt)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_pmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
i=0
for c in "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAVES" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p$i -o r -- python tools/conv_shapes_ab.py pmc > $O/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python - <<'PY'
import csv, glob, collections, os
O=os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/r03_pmc'
for d in sorted(glob.glob(O+'/p*/')):
    f=glob.glob(d+'**/*counter_collection.csv',recursive=True)
    if not f: print('no counters in',d); continue
    rows=list(csv.DictReader(open(f[0])))
    tr=glob.glob(d+'**/*kernel_trace.csv',recursive=True)
    dur={}
    if tr:
        for r in csv.DictReader(open(tr[0])): dur[r['Dispatch_Id']]=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        n=r['Kernel_Name']
        if 'direct16' not in n and 'tile16' not in n and 'resident' not in n and 'wide' not in n and 'splitk' not in n: continue
        key=(n.split('(')[0][-45:], r['Grid_Size'])
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
        agg[key]['dur_us'].append(dur.get(r['Dispatch_Id'],0))
    for k,v in agg.items():
        if len(v['dur_us'])<20: continue
        print(k, {c:round(sum(x)/len(x),1) for c,x in v.items()})
PY
