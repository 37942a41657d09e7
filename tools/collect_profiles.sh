#!/bin/bash
# Round-end evidence on the GPU box: kernel-trace stats of the default bench command, then PMC passes on the
# dominant kernel (one pass per counter set; FETCH_SIZE and WRITE_SIZE alone, as MI355X_MICROARCH.md prescribes).
#   gpurun -- 'bash tools/collect_profiles.sh r01_v6'
tag=${1:-rXX}
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p $out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py > $out/bench.log 2>&1
grep -h '^{"metric"' $out/bench.log > $out/bench.json
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_F32" \
           "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $out/pmc$i -- python3 tools/prof_kernel.py --reps 3 > $out/pmc$i.log 2>&1
done
python3 tools/summarize_pmc.py $out > $out/pmc_summary.json
cat $out/bench.json | cut -c1-400
grep -h "cheb\|localize_kernel\|index_\|letkf_sys" $out/trace/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
cat $out/pmc_summary.json | head -50
