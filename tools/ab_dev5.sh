TAG=$1; NAMES=$2; SPEC=$3; STEPS=${4:-300}
O=gpurun_out; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for round in 1 2 3 4 5; do
  for n in $NAMES; do
    DSG_LIB=$GRAFT_REPO_ROOT/diffusestylegesture_amd/csrc/libdsg_hip_$n.so python tools/sweep.py --spec $SPEC --steps $STEPS --reps 5 2>&1 | grep -v amdgpu.ids | sed "s/^/$n r$round: /" | tee -a $O/${TAG}_ab_$n.log
  done
done
