set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ev
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/ev/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/ev/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/ev/pytest_gpu.log
timeout -k 10 400 python bench.py > gpurun_out/ev/bench.json 2> gpurun_out/ev/bench.err
echo bench done
timeout -k 10 300 python tools/bench_encode.py --texts > gpurun_out/ev/bench_encode.json 2> gpurun_out/ev/bench_encode.err
echo encode done
timeout -k 10 300 python tools/bench_latency.py > gpurun_out/ev/latency.json 2> gpurun_out/ev/latency.err
echo latency done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ev/prof_s1 -o s1 -- python3 bench.py --steps 100 --warmup 10 --streams 1 --no-cpu-baseline --no-check > gpurun_out/ev/bench_s1_profiled.json 2> gpurun_out/ev/bench_s1.err
echo prof done
