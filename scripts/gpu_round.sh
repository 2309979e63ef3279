#!/bin/bash
# Full measurement pass on the GPU box: bench (N=1), rocprofv3 kernel stats, PMC FETCH/WRITE passes.
# Outputs under gpurun_out/round/; copy what should be judged into profiles/.
set -o pipefail
R=/root/repo; O=$R/gpurun_out/round; mkdir -p $O
cd $R && python bench.py > $O/bench_n1.json 2> $O/bench_n1.err && tail -c 3000 $O/bench_n1.json &&
cd /tmp && export TMPDIR=/tmp &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 3 --queries 200 --no-cpu-baseline --dropin-files 0 --aniso-rows 0 --other-rows 0 > $O/stats.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 1 --warmup 0 --queries 20 --no-cpu-baseline --dropin-files 0 --aniso-rows 0 --other-rows 0 > $O/pmc_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 1 --warmup 0 --queries 20 --no-cpu-baseline --dropin-files 0 --aniso-rows 0 --other-rows 0 > $O/pmc_write.log 2>&1 &&
python $R/scripts/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_fetch_write_summary.json &&
rm -rf $O/pmc_fetch $O/pmc_write && find $O/stats -name "*kernel_trace.csv" -delete && ls -la $O $O/stats/*
