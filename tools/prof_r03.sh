cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r03
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-genome --no-pmc --no-overlap > /tmp/ks.log 2>&1
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r03/r03_kernel_stats.csv
cd $R
bash tools/cli_prof.sh 50 > gpurun_out/r03/cli_prof.txt 2>&1; cp gpurun_out/cli_kernel_stats.csv gpurun_out/r03/r03_cli_kernel_stats.csv
bash tools/pmc_inflate.sh 10 > gpurun_out/r03/r03_inflate_pmc.txt 2>&1
bash tools/pmc_traffic.sh > gpurun_out/r03/r03_pmc_all_kernels.txt 2>&1
BDX_KZ_PROF=/tmp/kzprof.bin python tools/bamdec_probe.py --mbp 50 --inflate-only > gpurun_out/r03/r03_inflate_probe.txt 2>&1
head -5 gpurun_out/r03/r03_kernel_stats.csv | cut -c1-150; tail -3 gpurun_out/r03/r03_inflate_probe.txt
