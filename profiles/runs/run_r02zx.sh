# r02zx: bench.py's process-group path (RCCL init, barriers, max over ranks, size all-gather, payload all-gatherv) launched the way the driver
# launches N > 1 -- with one rank, the most a single-GPU box allows (ZHIP_BENCH_FORCE_DIST=1)
mkdir -p gpurun_out
export ZHIP_BENCH_FORCE_DIST=1
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 2 --warmup 1 --frames 2048 --no-cpu-baseline > gpurun_out/r02zx_decode.json 2> gpurun_out/r02zx_decode.err; tail -c 400 gpurun_out/r02zx_decode.json; tail -2 gpurun_out/r02zx_decode.err
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 2 --warmup 1 --frames 2048 --no-cpu-baseline --config roundtrip > gpurun_out/r02zx_roundtrip.json 2> gpurun_out/r02zx_roundtrip.err; tail -c 600 gpurun_out/r02zx_roundtrip.json; tail -2 gpurun_out/r02zx_roundtrip.err
