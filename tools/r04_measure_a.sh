#!/bin/bash
# Round-4 measurement set A (run on the GPU box through gpurun): bench line, rocprofv3 summary of the same command, PMC traffic,
# collector timings, emit tiles, the CLI end to end with its per-phase lines, `query` under rocprofv3.
O=$GRAFT_REPO_ROOT/gpurun_out/r04; mkdir -p $O
cd $GRAFT_REPO_ROOT
# (the PMC passes first: the bench line quotes roofline.traffic from profiles/pmc_latest.json, and only when that file is from THIS kernel build)
python tools/pmc_traffic.py --out $O/pmc_latest.json > $O/pmc_traffic.log 2>&1; echo "pmc rc $?"; tail -c 400 $O/pmc_traffic.log
cp $O/pmc_latest.json profiles/pmc_latest.json
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc $?"; tail -c 600 $O/bench_n1.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_torchrun_n1.json 2> $O/bench_torchrun_n1.err; echo "torchrun rc $?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_bench -o b --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_profiled.json 2> $O/bench_profiled.err; echo "rocprof bench rc $?"
cd $GRAFT_REPO_ROOT
python tools/config_timings.py > $O/collector_timings.log 2>&1; echo "timings rc $?"
bash tools/r04_emit_tiles.sh > $O/emit_tiles.out 2>&1; echo "emit rc $?"
timeout -k 10 600 python tests/manual/cli_e2e.py > $O/cli_e2e.log 2>&1; echo "cli e2e rc $?"; grep -v "^    " $O/cli_e2e.log | cut -c1-200 | head -8
bash tests/manual/rocprof_query.sh > /dev/null 2>&1; echo "rocprof query rc $?"; tail -3 $O/rocprof_query.log
