#!/bin/bash
# which engine kernels surround the torch glue kernels inside one step (dev tool; run via gpurun)
root=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pn
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pn -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/pn.log 2>&1
f=$(find /tmp/pn -name "*kernel_trace.csv" | head -1)
python $root/tools/step_neighbors.py "$f"
