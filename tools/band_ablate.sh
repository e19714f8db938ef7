# 3x3 band kernels: what bounds a launch?  Single launches at ResNet50's conv2 shapes with the probe build's ablation bits
# (results wrong on purpose): 4 = no LDS fragment reads in the K loop, 1 = no LDS-DMA issue in the K loop, 5 = both (MFMA + barriers only)
cd $GRAFT_REPO_ROOT
export HAWQ_LIB=$GRAFT_REPO_ROOT/hawq_amd/lib/libhawq_mi355_ablate.so
for d in 0 4 1 5; do echo "== HAWQ_DBG=$d"; HAWQ_DBG=$d timeout 300 python tools/bandprobe.py 64 2>&1 | grep -v amdgpu.ids; done
