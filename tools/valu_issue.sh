#!/bin/bash
# tools/valu_issue.sh <tag>: the VALU issue-rate microbenchmark (tools/microbench/valu_issue.hip) and one PMC pass over it that says what
# SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES read for a known instruction count -> gpurun_out/<tag>/valu_issue.{txt,json}, valu_issue_pmc.csv
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
[ -x tools/microbench/valu_issue ] || hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_issue.hip -o tools/microbench/valu_issue
tools/microbench/valu_issue $O/valu_issue.json > $O/valu_issue.txt 2>&1
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/pmc_valu
timeout -k 5 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d /tmp/pmc_valu -o valu --output-format csv -- $R/tools/microbench/valu_issue /tmp/valu_pmc.json 400 quick > $O/valu_issue_under_pmc.txt 2>&1
F=$(find /tmp/pmc_valu -name "*counter_collection.csv" | head -1)
python3 - "$F" > $O/valu_issue_pmc.csv <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
d=collections.OrderedDict()
for r in rows:
    d.setdefault(r['Dispatch_Id'],{'kernel':r['Kernel_Name'].split('(')[0],'grid':r.get('Grid_Size','')})[r['Counter_Name']]=float(r['Counter_Value'])
names=sorted({r['Counter_Name'] for r in rows})
print(','.join(['dispatch','kernel','grid']+names))
for k,v in d.items():
    print(','.join([k,'"%s"'%v['kernel'],str(v['grid'])]+['%.6g'%v.get(n,0) for n in names]))
PY
