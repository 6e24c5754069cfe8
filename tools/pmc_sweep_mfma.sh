out=${1:-gpurun_out/r05_pmc3}; mkdir -p $out; export TMPDIR=/tmp; root=$(pwd)
(cd /tmp && rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $root/$out/sq_counters.txt)
groups=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"
 "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_ACTIVE_INST_FLAT"
 "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_IFETCH GRBM_GUI_ACTIVE"
)
i=0
for g in "${groups[@]}"; do
  (cd /tmp && timeout 120 rocprofv3 --pmc $g --kernel-include-regex "sweep_mfma" -d "$root/$out/pass$i" --output-format csv -- python "$root/tools/cv_microbench.py" --variants 6 --layouts nhwc --lines 0 --reps 5 > "$root/$out/pass$i.log" 2>&1)
  i=$((i+1))
done
python tools/pmc_summary.py $out > $out/summary.txt 2>&1
cat $out/summary.txt; grep -c "" $out/sq_counters.txt; grep -i "error\|invalid\|not " $out/pass2.log $out/pass3.log | head
