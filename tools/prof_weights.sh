#!/bin/bash
# PMC passes on the weights kernel (C2).  gpurun -- 'bash tools/prof_weights.sh r05b'
tag=$1
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_${tag}_w
mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/prof_kernel.py --weights --reps 5 > $out/trace.log 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_F32" \
           "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $out/pmc$i -- python3 tools/prof_kernel.py --weights --reps 3 > $out/pmc$i.log 2>&1
done
python3 tools/summarize_pmc.py $out letkf_tile2w_kernel "C2 weights: letkf_tile2_kernel + letkf_tile2w_kernel on tile lists (tools/prof_kernel.py --weights --reps 3)" > $out/pmc_summary.json
grep -h "letkf_tile2w_kernel" $out/trace/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
head -c 2600 $out/pmc_summary.json
