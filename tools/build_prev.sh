#!/bin/bash
# build the kernels of git revision $1 (default HEAD) as iqlpref_amd/libiqlhip_prev.so for tools/ab.sh
REV=${1:-HEAD}
D=$(mktemp -d) && mkdir -p $D/iqlpref_amd/csrc $D/include || exit 1
for f in $(git ls-tree --name-only $REV iqlpref_amd/csrc/) include/iqlhip.h; do git show $REV:$f > $D/$f; done
(cd $D/iqlpref_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -mllvm -amdgpu-kernarg-preload-count=16 \
   -o $OLDPWD/iqlpref_amd/libiqlhip_prev.so api.hip iql_step.hip buffer.hip mlp_f32.hip cvar.hip pt.hip prep.hip)
rm -rf $D
