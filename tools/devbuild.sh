#!/bin/bash
# development aid: tools/devbuild.sh NAME [ilp|noilp] [-DFLAG ...]  ->  moshpp_amd/libmoshii_NAME_prof.so with only the NBLK=4
# chain kernel (seconds to build); the other objects are taken from the last full profile build.
set -e
name=$1; shift
PROF=-DMOSHII_PROFILE; [ -n "$NOPROF" ] && PROF=
sched="-mllvm -amdgpu-sched-strategy=iterative-ilp"
if [ "$1" = "noilp" ]; then sched=""; shift; elif [ "$1" = "ilp" ]; then shift; fi
cd "$(dirname "$0")/.."
C=moshpp_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value $PROF "$@" -c $C/moshii_api.hip -o /tmp/t/v/api_$name.o &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value -DMOSHII_DEV_ONLY_NBLK4 $PROF "$@" -c $C/chain_solve.hip -o /tmp/t/v/cs_$name.o $sched
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o moshpp_amd/libmoshii_${name}_prof.so /tmp/t/v/api_$name.o /tmp/t/v/cs_$name.o $C/lbs_forward_prof.o $C/stagei_prof.o
echo built moshpp_amd/libmoshii_${name}_prof.so
