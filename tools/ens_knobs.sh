#!/bin/bash
# A/B of the launch-shape knobs for one (critics, batch) configuration: tools/ens_knobs.sh <out> [E B]
OUT=$1; E=${2:-4}; B=${3:-1024}
run() { echo "== $*" >> $OUT; env "$@" timeout -k 10 120 python tools/ens_run.py $E $B 3000 2>/dev/null >> $OUT || exit 1; }
run X=0
for mt in 1 2 4; do for pw in 1 2; do run IQLHIP_FWD_MT=$mt IQLHIP_FWD_PW=$pw; done; done
for pw in 1 2 4; do run IQLHIP_BWD_PW=$pw; done
run IQLHIP_BWD_PRE=1
run IQLHIP_UPD_LAT=1
run IQLHIP_ITEM_BALANCE=0
run X=0
