#!/bin/bash
# rocprofv3 counter passes over the bottleneck convolution kernel (ConvLSTM shape), GPU box.  Usage: tools/pmc_bottleneck.sh <out-dir>
# One pass per counter group (PMC only: never combined with tracing), then a kernel trace; summary in <out-dir>/summary.txt.
out="$1"
mkdir -p "$out"
export TMPDIR=/tmp
root="$(pwd)"
cat > "$out/driver.py" <<'PY'
import os, sys
ROOT = os.environ["DVMVS_ROOT"]
sys.path.insert(0, os.path.join(ROOT, "deep-video-mvs_amd"))
import torch
from dvmvs.hip import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
for (ci, co, h, w, st) in ((1024, 2048, 8, 10, 1), (512, 512, 8, 10, 1), (512, 256, 16, 20, 1)):
    x = torch.randn(1, ci, h, w, generator=g).to(dev)
    wt = (torch.randn(co, ci, 3, 3, generator=g) / 96).to(dev)
    packed = ops.bottleneck_conv_pack(wt)
    S = ops.bottleneck_conv_splits(1, co, ci, h, w, st)
    parts = torch.empty(S * co * (h // st) * (w // st), device=dev)
    for _ in range(5):
        ops.bottleneck_conv_into(x, packed, co, st, parts)
    torch.cuda.synchronize()
PY
groups=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"
 "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE"
 "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
 "FETCH_SIZE"
)
i=0
for g in "${groups[@]}"; do
  (cd /tmp && DVMVS_ROOT="$root" timeout 120 rocprofv3 --pmc $g --kernel-include-regex "bottleneck_conv_kernel" -d "$root/$out/pass$i" --output-format csv -- \
     python "$root/$out/driver.py" > "$root/$out/pass$i.log" 2>&1)
  i=$((i+1))
done
(cd /tmp && DVMVS_ROOT="$root" timeout 120 rocprofv3 --kernel-trace --stats -d "$root/$out/trace" --output-format csv -- python "$root/$out/driver.py" > "$root/$out/trace.log" 2>&1)
cp "$out"/trace/*/*kernel_stats.csv "$out/kernel_stats.csv" 2>/dev/null
python tools/pmc_summary.py "$out" bottleneck > "$out/summary.txt" 2>&1
