for v in default flat1 flat2 default flat1 flat2; do
  if [ $v = default ]; then unset EPRECON_LIB_PATH; else export EPRECON_LIB_PATH=$PWD/build/variants/$v/libeprecon_hip.so; fi
  echo "== $v"; python tools/conv_shapes_ab.py $v 2>/dev/null | grep -E "ConvGRU (voxel s1|s2)|SPVCNN1 up2|SPVCNN2 up2 |SPVCNN2 stage1|mask"
done
