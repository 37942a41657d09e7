#!/bin/bash
# kernel trace + HBM traffic counters of the per-point weight transform (csrc/apply_local.hip), 16 state rows of 1e5 points, k = 40
#   gpurun -- 'bash tools/prof_apply.sh r04_apply'
tag=${1:-rXX}
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/apply_sweep.py 100000 40 16 > $out/trace.log 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_F32" "SQ_WAVES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $out/pmc$i -- python3 tools/apply_sweep.py 100000 40 16 > $out/pmc$i.log 2>&1
done
for k in apply_local_tile_kernel apply_global_tile_kernel; do
  python3 tools/summarize_pmc.py $out $k "k = 40, 1e5 grid points, 16 state rows (tools/apply_sweep.py 100000 40 16)" > $out/pmc_$k.json
  head -60 $out/pmc_$k.json
done
