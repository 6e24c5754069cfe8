#!/bin/bash
# rocprofv3 counter passes over the cost-volume microbench (GPU box).  Usage:
#   tools/pmc_sweep.sh <out-dir> <microbench args...>
# One pass per counter group (PMC only: never combined with tracing), summaries in <out-dir>/summary.txt.
out="$1"; shift
mkdir -p "$out"
export TMPDIR=/tmp
root="$(pwd)"
groups=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"
 "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
# PMC_ONLY="3 4" restricts the run to those counter groups (indices into the list above), e.g. the two HBM-traffic passes
i=0
for g in "${groups[@]}"; do
  if [ -n "$PMC_ONLY" ] && ! [[ " $PMC_ONLY " == *" $i "* ]]; then i=$((i+1)); continue; fi
  (cd /tmp && timeout 120 rocprofv3 --pmc $g --kernel-include-regex "sweep_|cost_volume_tiled|cost_volume_spill" -d "$root/$out/pass$i" --output-format csv -- \
     python "$root/tools/cv_microbench.py" "$@" > "$root/$out/pass$i.log" 2>&1)
  i=$((i+1))
done
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d "$root/$out/trace" --output-format csv -- \
   python "$root/tools/cv_microbench.py" "$@" > "$root/$out/trace.log" 2>&1)
cp "$out"/trace/*/*kernel_stats.csv "$out/kernel_stats.csv" 2>/dev/null
python tools/pmc_summary.py "$out" > "$out/summary.txt" 2>&1
