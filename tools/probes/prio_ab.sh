for v in default prio1 prio2 default prio1 prio2; do
  if [ $v = default ]; then unset EPRECON_LIB_PATH; else export EPRECON_LIB_PATH=$PWD/build/variants/$v/libeprecon_hip.so; fi
  echo "== $v"; python tools/conv_shapes_ab.py $v 2>/dev/null | grep -E "ConvGRU (voxel s1|s2|img s1)|SPVCNN2 up2|mask|sum"
  EPRECON_CONV_BF16X3=1 python tools/conv_shapes_ab.py $v 2>/dev/null | grep -E "ConvGRU (voxel s1 |s2)|sum" | sed 's/^/bf: /'
done
