#!/bin/bash
# PMC counters of the blur lab kernels (separate passes; no trace domains).  usage: scratch/pmc_lab.sh <only-filter>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_lab; rm -rf $O; mkdir -p $O
rocprofv3 -L 2>&1 | grep -E "Counter_Name" | grep -oE "(SQ|SQC|TCP|TCC|TA|TD|GRBM|LDS)_[A-Za-z0-9_]+" | sort -u | tr "\n" " " > $O/counter_names.txt
F="${1:-v2 px4}"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/p1 -o p -- scratch/blur_lab 4000 3000 16 "$F" > $O/log1.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_VMEM --output-format csv -d $O/p2 -o p -- scratch/blur_lab 4000 3000 16 "$F" > $O/log2.txt 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_WAIT_IFETCH SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC --output-format csv -d $O/p3 -o p -- scratch/blur_lab 4000 3000 16 "$F" > $O/log3.txt 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("p1", "p2", "p3"):
    f = glob.glob("gpurun_out/pmc_lab/%s/**/*counter_collection.csv" % d, recursive=True)
    if not f: print(d, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"][:60] + " grid=" + r.get("Grid_Size", "?")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
    for k, v in sorted(acc.items()):
        print(d, k, {c: "%.4g" % (x / cnt[(k, c)]) for c, x in v.items()})
PY
rm -rf $O/p1 $O/p2 $O/p3
