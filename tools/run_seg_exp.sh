cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for x in 0 64 128 192; do
  MIA_EXPERIMENT_SKIP=$x timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/exp$x -- python3 tools/native_timeline.py 4 30 > gpurun_out/exp$x.log 2>&1
  echo "xskip=$x"; grep -h "cheb_seg\|segment_wait" gpurun_out/exp$x/*/*kernel_stats.csv | cut -d, -f1-5 | cut -c1-120
done
