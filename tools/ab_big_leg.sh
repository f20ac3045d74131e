cd /root/repo
for tag in "" ntxy ntout ntboth ""; do
  lib=optimization_amd/libmi355opt${tag:+_$tag}.so
  echo "== ${tag:-base}"
  MI355OPT_LIB=$PWD/$lib python tools/big_leg.py 200 100 2>/dev/null | tail -1
done
