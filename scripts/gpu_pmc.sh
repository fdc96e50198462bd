#!/bin/bash
# PMC passes (separate runs, --pmc only) over a short run of the bench workload
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --prof-steps 0 ${BENCH_ARGS}"
rm -rf $R/gpurun_out/pmc; mkdir -p $R/gpurun_out/pmc
timeout -k 10 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc/p1 -o p1 -- $CMD > $R/gpurun_out/pmc1.log 2>&1 || { tail -n 20 $R/gpurun_out/pmc1.log; exit 1; }
echo pass1 done
timeout -k 10 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc/p2 -o p2 -- $CMD > $R/gpurun_out/pmc2.log 2>&1 || { tail -n 20 $R/gpurun_out/pmc2.log; exit 1; }
echo pass2 done
timeout -k 10 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc/p3 -o p3 -- $CMD > $R/gpurun_out/pmc3.log 2>&1 || { tail -n 20 $R/gpurun_out/pmc3.log; exit 1; }
echo pass3 done
head -n 2 $(find $R/gpurun_out/pmc/p1 -name "*counter_collection.csv" | head -1) | cut -c1-600
python3 $R/scripts/pmc_summary.py $R/gpurun_out/pmc_summary.csv $R/gpurun_out/pmc/p1 $R/gpurun_out/pmc/p2 $R/gpurun_out/pmc/p3
find $R/gpurun_out/pmc -name "*.csv" -size +8M -delete
