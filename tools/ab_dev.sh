#!/bin/bash
# A/B of development builds (make dev DEVNAME=...: libdsg_hip_<name>.so + dsg_kernels_<name>.hsaco) on ONE box, two interleaved rounds:
#   gpurun --timeout 600 -- 'bash tools/ab_dev.sh TAG "devA devB" block:1x16,block:4x4 [steps] [extra sweep.py args]'
# sha1 = checksum of the samples (bit-identity of two variants shows as the same checksum).
TAG=$1; NAMES=$2; SPEC=$3; STEPS=${4:-200}; shift 4
O=gpurun_out; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for n in $NAMES; do
    DSG_LIB=$GRAFT_REPO_ROOT/diffusestylegesture_amd/csrc/libdsg_hip_$n.so python tools/sweep.py --spec $SPEC --steps $STEPS --reps 3 "$@" 2>&1 | grep -v amdgpu.ids | sed "s/^/$n r$round: /" | tee -a $O/${TAG}_ab_$n.log
  done
done
