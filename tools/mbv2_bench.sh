set -e
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_network.py -x -q -k "depthwise or im2col or mobilenet" 2>&1 | tail -3
timeout 300 python - <<'PY'
import torch, json, bench
print(json.dumps(bench.mobilenet_line(128, torch.device('cuda:0'), 50)))
PY
mkdir -p gpurun_out/mb
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/mb -o mb -- python - <<'PY' > /dev/null 2>&1
import sys, os, torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import bench
bench.mobilenet_line(128, torch.device('cuda:0'), 30)
PY
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $(find gpurun_out/mb -name "*.db" | head -1) gpurun_out/mb_kernel_trace.md | head -16
