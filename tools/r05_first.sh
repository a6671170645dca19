#!/bin/bash
# round 5, first GPU call: the suite on the current build, the driver's bench line, the vector-L1 probe, the -fno-honor-nans A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05a
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
timeout 120 tools/probes/l1_line_rate > $O/l1_line_rate.txt 2>&1; cat $O/l1_line_rate.txt
timeout 600 python bench.py --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$O/bench.json'))
print({k: d[k] for k in ('value','ms_per_step','cfg4_ms_per_fragment','cfg4_launches_per_fragment','cfg4_blocking_reads_per_fragment','cfg4_roofline_conv_alone_frac','e2e_ms_per_fragment')})
print(d['roofline']); print(d.get('roofline_conv')); print({k:v for k,v in d['extra'].items() if 'ms' in k or 'error' in k})"
for lib in "" eprecon_amd/libeprecon_hip_plain.so; do
  for aff in 1 0; do
    EPRECON_LIB_PATH=${lib:+$R/$lib} EPRECON_AB_IN_AFFINE=$aff timeout 300 python tools/conv_shapes_ab.py "lib=${lib:-shipped} affine=$aff" > $O/conv_shapes_${aff}_$(basename ${lib:-shipped} .so).txt 2>&1
    tail -1 $O/conv_shapes_${aff}_$(basename ${lib:-shipped} .so).txt
  done
done
