ROOT=${GRAFT_REPO_ROOT:-/root/repo}
L=$ROOT/curobo_amd/lib
cp $L/libcurobo_hip.so /tmp/orig.so; cp $L/variants/libcurobo_hip_stampterms.so $L/libcurobo_hip.so
python $ROOT/tools/r05/terms_stamps.py --seeds 8 2>&1 | grep -v amdgpu
CUROBO_HIP_FUSED_NO_POSE_STAGE=1 python $ROOT/tools/r05/terms_stamps.py --seeds 8 2>&1 | grep -v amdgpu
python $ROOT/tools/r05/terms_stamps.py --seeds 256 2>&1 | grep -v amdgpu
CUROBO_HIP_FUSED_NO_POSE_STAGE=1 python $ROOT/tools/r05/terms_stamps.py --seeds 256 2>&1 | grep -v amdgpu
cp /tmp/orig.so $L/libcurobo_hip.so
