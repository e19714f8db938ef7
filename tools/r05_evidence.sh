#!/bin/bash
# Round-5 kernel evidence, run ON the GPU box: gpurun -- 'bash tools/r05_evidence.sh'   (needs the probe build: make -C hawq_amd/csrc ABLATE=1)
# Writes gpurun_out/r05_*: single launches of every 3x3 kernel (rounds 1-4 band kernels, persistent kernel, band_v2) and of the streaming
# 1x1 kernels against the best general tile at ResNet50's shapes, the s_memtime stamps and ablations of the probe build (HAWQ_DBG bits of
# band_v2.hip / gemm_v2.hip: 1 no LDS-DMA in the K loop, 2 no MFMA, 4 no fragment reads, 32 no weight DMA, 64 no band DMA, 128 stamps),
# two passes of SQ counters over the 3x3 probe, and the three LDS-DMA micro-benchmarks of this round.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 300 python tools/band2probe.py 64 128 2>&1 | grep -v amdgpu.ids > $O/r05_band2probe.txt
timeout 300 python tools/gemm2probe.py 64 128 2>&1 | grep -v amdgpu.ids > $O/r05_gemm2probe.txt
export HAWQ_LIB=$R/hawq_amd/lib/libhawq_mi355_ablate.so
for d in 128 129 130 132 134 160 192 135; do
  echo "== HAWQ_DBG=$d"
  HAWQ_DBG=$d SHAPES=2,3 timeout 200 python tools/band2probe.py 64 2>&1 | grep -v amdgpu.ids | grep "v2\|band-v2" | sort | uniq -c | sort -k2 | head -40
done > $O/r05_band2_stamps_ablations.txt
for d in 128 129 130 132; do
  echo "== HAWQ_DBG=$d"
  HAWQ_DBG=$d timeout 200 python tools/gemm2probe.py 64 2>&1 | grep -v amdgpu.ids | grep "g2\|gemm-v2" | sort | uniq -c | sort -k2 | head -40
done > $O/r05_gemm2_stamps_ablations.txt
unset HAWQ_LIB
bash tools/band2pmc.sh 64 > $O/r05_band2_pmc.md 2>&1
for u in dma_issue dma_l2 dma_exec; do
  [ -x tools/ubench/bin/$u ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o tools/ubench/bin/$u tools/ubench/$u.hip
  timeout 200 tools/ubench/bin/$u > $O/r05_ubench_$u.txt 2>&1
done
