#!/bin/bash
# PMC counters of the streaming SIFT kernels (separate passes; no trace domains)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_sift; rm -rf $O; mkdir -p $O
RX="blur16_stream|extrema_stream"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-include-regex "$RX" --output-format csv -d $O/p1 -o p -- python scratch/sift_time.py 16 4000 3000 8 > $O/log1.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_VMEM --kernel-include-regex "$RX" --output-format csv -d $O/p2 -o p -- python scratch/sift_time.py 16 4000 3000 8 > $O/log2.txt 2>&1
python - <<'PY'
import csv, glob, collections
for d in ("p1", "p2"):
    f = glob.glob("gpurun_out/pmc_sift/%s/**/*counter_collection.csv" % d, recursive=True)
    if not f: print(d, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].replace("(anonymous namespace)::","")[:34] + " grid=" + r.get("Grid_Size", "?")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
    for k, v in sorted(acc.items()):
        print(d, k, {c: "%.4g" % (x / cnt[(k, c)]) for c, x in v.items()})
PY
rm -rf $O/p1 $O/p2
