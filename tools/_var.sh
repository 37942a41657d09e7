export PROF_STEPS=20
PROF_ARGS="--config c5" PROF_WORKLOAD="C5: 1e5 grid points, k=40, <=20 local obs, RBF gamma 0.5, m=1 (tools/prof_kernel.py --config c5 --reps 3)" bash tools/prof_tile.sh r04_c5 lketkf_tile_kernel > gpurun_out/prof_r04_c5.log 2>&1
PROF_ARGS="--weights" PROF_WORKLOAD="C2 weights: 1e5 grid points, k=40, (G,k,k) weights (tools/prof_kernel.py --weights --reps 3)" bash tools/prof_tile.sh r04_w letkf_tile2w_kernel > gpurun_out/prof_r04_w.log 2>&1
PROF_ARGS="--config c4" PROF_WORKLOAD="C4: 1e5 grid points, k=80, <=63 local obs, m=1 (tools/prof_kernel.py --config c4 --reps 3)" bash tools/prof_tile.sh r04_c4 letkf_tile2_kernel > gpurun_out/prof_r04_c4.log 2>&1
tail -3 gpurun_out/prof_r04_c5.log
