for v in default t16mix1 t16mix2 default t16mix1 t16mix2; do
  if [ $v = default ]; then unset EPRECON_LIB_PATH; else export EPRECON_LIB_PATH=$PWD/build/variants/$v/libeprecon_hip.so; fi
  python bench.py --no-extra --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],4), d['roofline_conv']['frac'] if d.get('roofline_conv') else None, d['roofline_conv'].get('avg_launch_ms') if d.get('roofline_conv') else None)"
done
python tools/profile_cfg2_stages.py 2>/dev/null | grep -E "sparse stack|SubM k3"
EPRECON_LIB_PATH=$PWD/build/variants/t16mix1/libeprecon_hip.so python tools/profile_cfg2_stages.py 2>/dev/null | grep -E "sparse stack|SubM k3"
EPRECON_LIB_PATH=$PWD/build/variants/t16mix2/libeprecon_hip.so python tools/profile_cfg2_stages.py 2>/dev/null | grep -E "sparse stack|SubM k3"
