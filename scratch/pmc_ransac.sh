#!/bin/bash
# PMC counters of ransac_kernel / bf_match_kernel on the C4-like pair stage (separate passes; no trace domains)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_ransac; rm -rf $O; mkdir -p $O
RX="ransac_kernel|bf_match_kernel"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-include-regex "$RX" --output-format csv -d $O/p1 -o p -- python scratch/ransac_time.py 200 > $O/log1.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_WAVES --kernel-include-regex "$RX" --output-format csv -d $O/p2 -o p -- python scratch/ransac_time.py 200 > $O/log2.txt 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_WAIT_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --kernel-include-regex "$RX" --output-format csv -d $O/p3 -o p -- python scratch/ransac_time.py 200 > $O/log3.txt 2>&1
python - <<'PY'
import csv, glob, collections
for d in ("p1", "p2", "p3"):
    f = glob.glob("gpurun_out/pmc_ransac/%s/**/*counter_collection.csv" % d, recursive=True)
    if not f: print(d, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].replace("(anonymous namespace)::","")[:34] + " grid=" + r.get("Grid_Size", "?")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
    for k, v in sorted(acc.items()):
        print(d, k, {c: "%.4g" % (x / cnt[(k, c)]) for c, x in v.items()})
PY
tail -3 $O/log2.txt
rm -rf $O/p1 $O/p2 $O/p3
