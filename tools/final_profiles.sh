export TMPDIR=/tmp
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/rp -o rp -- python3 bench.py > gpurun_out/final/bench_under_rocprof.json 2>/dev/null
python tools/profile_ops.py > gpurun_out/final/ops.txt 2>&1
export DVITS_NO_GRAPH=1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/final/pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/final/pmc_write -o w -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > /dev/null 2>&1
unset DVITS_NO_GRAPH
python tools/pmc_traffic.py gpurun_out/final/pmc_fetch/f_results.db gpurun_out/final/pmc_write/w_results.db > gpurun_out/final/traffic.json 2> gpurun_out/final/traffic.err
timeout 200 python tools/attn_trace.py > gpurun_out/final/attn_trace.txt 2>&1
timeout 200 python tools/gemm_trace.py > gpurun_out/final/gemm_trace.txt 2>&1
rm -rf gpurun_out/final/pmc_fetch gpurun_out/final/pmc_write
rm -f gpurun_out/final/rp/*kernel_trace.csv
ls -la gpurun_out/final gpurun_out/final/rp
cat gpurun_out/final/bench.json | cut -c1-400
