cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_coexec; mkdir -p $out
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_VALU --output-format csv -d $out/p1 -- python3 tools/prof_kernel.py --reps 3 > $out/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/p2 -- python3 tools/prof_kernel.py --reps 3 > $out/p2.log 2>&1
python3 - <<'PY'
import csv,glob,collections
for d in ("p1","p2"):
    acc=collections.defaultdict(list)
    for f in glob.glob("gpurun_out/prof_coexec/%s/**/*counter_collection.csv"%d, recursive=True):
        per=collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if "letkf_tile" in r["Kernel_Name"]:
                per[r["Dispatch_Id"]][r["Counter_Name"]]+=float(r["Counter_Value"])
        for disp in per.values():
            for k,v in disp.items(): acc[k].append(v)
    print(d, {k: sum(v)/len(v) for k,v in acc.items()})
PY
tail -3 $out/p1.log $out/p2.log
