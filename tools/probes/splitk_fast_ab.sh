# A/B of the branch-free instantiations of the split-K kernel (EPRECON_CONV_SPLITK_FAST=0: the general form), two rounds in one call;
# EPRECON_CONV_WIDEK=0 sends the medium lists with wide channels to the split-K kernel as well; without / with a pending BatchNorm
for wk in 1 0; do for aff in 0 1; do for v in 0 1 0 1; do
  echo "== splitk_fast=$v widek=$wk in_affine=$aff"; EPRECON_CONV_WIDEK=$wk EPRECON_AB_IN_AFFINE=$aff EPRECON_CONV_SPLITK_FAST=$v python tools/conv_shapes_ab.py "fast=$v" 2>/dev/null | grep -v "^#"
done; done; done
