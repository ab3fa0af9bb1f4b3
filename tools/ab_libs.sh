for rep in 1 2; do for v in nrow1 nrow2; do echo "== $v"; SEGGROUP_HIP_LIB=$PWD/build_micro/lib_$v.so bash tools/quick_prof.sh ab_$v 12 2>&1 | grep "k_edgeconv_hb"; done; done
