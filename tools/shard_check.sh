for mode in 0 1; do
GF_FRAME_THREE_STEP_DAF=$mode GF_BENCH_SHARED_GPU=1 GF_BENCH_CHECK=1 MASTER_ADDR=127.0.0.1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); fs=d['frame_sharded']
print('three_step=$mode', {c:{h:fs[c][h].get('labels_equal_single_gpu_fraction') for h in ('slab','allreduce')} for c in ('nuscenes_gs25600_solid','nuscenes_gs144000')})"
done
