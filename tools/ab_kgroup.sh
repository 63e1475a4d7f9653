mkdir -p gpurun_out; export TMPDIR=/tmp
for cfg in "0 0 0" "4 0 4" "2 0 2" "6 0 6" "4 1 4" "4 2 4"; do
  set -- $cfg
  echo "== OG_KGROUP_S1=$1 OG_KGROUP_S2=$2 OG_KGROUP_PH=$3"
  OG_KGROUP_S1=$1 OG_KGROUP_S2=$2 OG_KGROUP_PH=$3 timeout 120 tools/conv_bench "" 8 2 | grep -E "res1_128|res2_128|res1_64|up_256|objd_l2|objd_l3|d_l4 |joint|incep" | cut -c1-175
done
