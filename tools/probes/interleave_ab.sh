# A/B of the interleaved schedule of the direct gather kernel (EPRECON_CONV_INTERLEAVE=0: the fenced one), two rounds in one call;
# without and with a pending BatchNorm + ReLU of the input (EPRECON_AB_IN_AFFINE=1)
for aff in 0 1; do for v in 0 1 0 1; do
  echo "== interleave=$v in_affine=$aff"; EPRECON_AB_IN_AFFINE=$aff EPRECON_CONV_INTERLEAVE=$v python tools/conv_shapes_ab.py "interleave=$v" 2>/dev/null | grep -v "^#"
done; done
