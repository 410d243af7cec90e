cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 600 --timeout-method=thread -k "nbody_bench_c or auto_lands or two_real_ranks or survives or shard_leg or frame_loop or zero_copy" > $O/r06_pytest_b.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/r06_pytest_b.txt
HSA_ENABLE_IPC_MODE_LEGACY=1 timeout -k 10 300 python bench.py --gpus 2 --steps 4 --warmup 1 --particles 65536 --extra-particles 131072 --no-extras > $O/r06_legacy_ipc_2ranks.json 2> $O/r06_legacy_ipc_2ranks.err; echo "legacy-ipc bench rc=$?"
HSA_ENABLE_IPC_MODE_LEGACY=1 timeout -k 10 300 ./nbody_amd/lib/nbody-bench --gpus 2 --n 65536 --steps 5 --warmup 1 --dt 0.01 > $O/r06_legacy_ipc_cbench.txt 2>&1; echo "legacy-ipc nbody-bench rc=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 3 > $O/r06_bench_b.json 2> $O/r06_bench_b.err; echo "bench rc=$?"
