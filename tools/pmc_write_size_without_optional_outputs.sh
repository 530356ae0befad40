export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmcx
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmcx/noopt -- python3 tools/kernel_probe.py 4096 > gpurun_out/pmcx/noopt.log 2>&1
python3 - <<'PY'
import sys; sys.path.insert(0,'tools')
from pmc_summary import counter_values
import statistics
for d in ('gpurun_out/pmcx/noopt',):
    v=counter_values(d,'WRITE_SIZE'); print(d, len(v), statistics.median(v))
PY
find gpurun_out/pmcx -name "*.csv" -size +4M -delete
