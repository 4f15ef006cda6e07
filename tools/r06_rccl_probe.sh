# One gpurun call (round 6): the multi-GPU half on ONE MI355X.
#  (1) tests/test_gpu_rccl.py + tests/test_gpu_dist.py;
#  (2) the forced one-rank RCCL record with its stream probes -> gpurun_out/r06_rccl_world1.json;
#  (3) does RCCL accept TWO ranks on one device?  (IIC_RCCL_SHARED_DEVICE=1; expected: no -- recorded either way, under
#      a timeout, so that a hang cannot take the box down);
#  (4) python bench.py --gpus 2 on this 1-GPU box: exit code + message.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
S="--pairs 66 --steps 3 --warmup 1 --no-roofline --no-cpu-baseline --no-secondary --no-reference-api"
timeout 1500 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_dist.py -m gpu -x -q > gpurun_out/r06_rccl_tests.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r06_rccl_tests.log
IIC_DIST_FORCE=1 IIC_STREAM_PROBE_LOG=1 timeout 300 python bench.py --gpus 1 $S > gpurun_out/r06_rccl_world1.json 2> gpurun_out/r06_rccl_world1.err; echo "world1 rc=$?"; grep "stream probe" gpurun_out/r06_rccl_world1.err | head -40
python - <<'P'
import json
d = json.loads([l for l in open("gpurun_out/r06_rccl_world1.json") if l.startswith("{")][-1])
print(json.dumps(d["config"]["data_parallel"], indent=1)[:3000]); print(d["config"]["launch"]); print(d["value"], d["ms_per_step"])
P
( IIC_RCCL_SHARED_DEVICE=1 NCCL_DEBUG=WARN timeout -s KILL 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29731 bench.py --gpus 2 $S > gpurun_out/r06_rccl_two_ranks_one_device.out 2> gpurun_out/r06_rccl_two_ranks_one_device.err; echo "two ranks on one device through RCCL: rc=$?" ) 2>&1 | tee gpurun_out/r06_rccl_two_ranks_one_device.rc
grep -i "duplicate\|error\|invalid" gpurun_out/r06_rccl_two_ranks_one_device.err | head -8; tail -2 gpurun_out/r06_rccl_two_ranks_one_device.out | cut -c1-300
env -u WORLD_SIZE timeout 200 python bench.py --gpus 2 $S > gpurun_out/r06_gpus2_one_box.out 2> gpurun_out/r06_gpus2_one_box.err; echo "python bench.py --gpus 2 on this box: rc=$?"; tail -2 gpurun_out/r06_gpus2_one_box.err
