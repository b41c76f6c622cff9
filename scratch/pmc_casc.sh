#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_casc; rm -rf $O; mkdir -p $O
VARIANT=x rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-include-regex "pyr_cascade|blur_stream" --output-format csv -d $O/p1 -o p -- python scratch/casc_time.py > $O/log1.txt 2>&1
VARIANT=x rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-include-regex "pyr_cascade|blur_stream" --output-format csv -d $O/p2 -o p -- python scratch/casc_time.py > $O/log2.txt 2>&1
python - <<'PY'
import csv, glob, collections
for d in ("p1", "p2"):
    f = glob.glob("gpurun_out/pmc_casc/%s/**/*counter_collection.csv" % d, recursive=True)
    if not f: print(d, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"][:40] + " grid=" + r.get("Grid_Size", "?")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
    for k, v in acc.items():
        print(d, k, {c: "%.4g" % (x / cnt[(k, c)]) for c, x in v.items()})
PY
