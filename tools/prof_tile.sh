#!/bin/bash
# kernel-trace stats of the default bench + PMC passes on the dominant kernel.  gpurun -- 'bash tools/prof_tile.sh r02_vN [kernel-name]'
tag=${1:-rXX}; kern=${2:-letkf_tile_kernel}
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline --steps ${PROF_STEPS:-500} > $out/bench.log 2>&1
grep -h '^{"metric"' $out/bench.log > $out/bench.json
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_F32" \
           "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $out/pmc$i -- python3 tools/prof_kernel.py --reps 3 ${PROF_ARGS} > $out/pmc$i.log 2>&1
done
python3 tools/summarize_pmc.py $out $kern "${PROF_WORKLOAD:-C2: 1e5 grid points, k=40, <=20 local obs, m=1 (tools/prof_kernel.py --reps 3)}" > $out/pmc_summary.json
cut -c1-300 $out/bench.json
grep -h "tile\|cheb\|localize_kernel\|index_\|letkf_sys" $out/trace/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
cat $out/pmc_summary.json | head -60
