#!/bin/bash
# PMC passes (one per counter set; FETCH_SIZE and WRITE_SIZE alone) on tools/prof_kernel.py, summarised per kernel.
#   gpurun -- 'bash tools/prof_pmc.sh r03_v1 letkf_tile2_kernel [prof_kernel args]'
tag=${1:-rXX}; kern=${2:-letkf_tile2_kernel}; shift; shift
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_F32" \
           "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $out/pmc$i -- python3 tools/prof_kernel.py --reps 3 "$@" > $out/pmc$i.log 2>&1
  echo "pass $i done"
done
for kn in $kern localize_tiles_kernel letkf_tile2w_kernel; do
  python3 tools/summarize_pmc.py $out $kn > $out/pmc_summary_$kn.json
  cat $out/pmc_summary_$kn.json
done
rm -rf $out/pmc[0-9]*/
