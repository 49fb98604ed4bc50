"""The multi-GPU code path on the REAL backend, as far as a 1-GPU box can take it: `torch.distributed.run --nproc-per-node 1`
(the driver's launcher) + `init_process_group('nccl')` (= RCCL on ROCm) with world size 1, then every collective of
neuray_amd/parallel.py - the fused all-gather of tiles, the view-sharded encoder all-gather, the flat gradient all-reduce with its
grad-None flags, the ft-mode sparse exchange (int64 MAX all-reduce, all_gather of ids, fp32 all-reduce) - on device tensors produced
by the HIP kernels.  World size 2 ... 8 is covered on CPU (gloo, tests/test_sharding_gloo.py, tests/test_bench_launcher.py); what
gloo cannot show is whether RCCL accepts these calls (dtypes, views, in-place buffers, device binding), which this does."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
ROOT = %r
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from neuray_amd import parallel, synthetic
from neuray_amd.network.renderer import NeuralRayBaseRenderer

rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ['LOCAL_RANK'])
dev = torch.device('cuda', local)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', device_id=dev)
assert dist.get_world_size() == world == 1
seen = torch.tensor([1.0], device=dev); dist.all_reduce(seen); assert int(seen.item()) == 1

cfg = {'use_hierarchical_sampling': True, 'dist_decoder_cfg': {'use_vis': False}, 'depth_sample_num': 16, 'fine_depth_sample_num': 16,
       'agg_net_cfg': {'sample_num': 16}, 'fine_agg_net_cfg': {'sample_num': 16}, 'ray_batch_num': 64}
torch.manual_seed(0)
r = NeuralRayBaseRenderer(cfg).eval().to(dev)
que, ref = synthetic.make_scene(48, 64, 3, seed=5)
que['coords'] = synthetic.meshgrid_coords(48, 64)[:, :200]
tq = {k: torch.from_numpy(v).to(dev) for k, v in que.items()}
tr = {k: torch.from_numpy(v).to(dev) for k, v in ref.items()}
with torch.no_grad():
    single = r.render(dict(tq), dict(tr), False)
    shard = parallel.render_image_sharded(r, dict(tq), dict(tr))
for k, v in single.items():
    assert torch.equal(shard[k], v), k                       # one fused all_gather_into_tensor, bools through fp32

# flat gradient all-reduce with grad-None flags + the ft-mode sparse exchange, on gradients the backward kernels wrote
r.train()
tr2 = {k: v.clone() for k, v in tr.items()}
tr2['ray_feats'].requires_grad_(True)
q2 = dict(tq, coords=tq['coords'][:, :32].contiguous())
out = r.render(q2, tr2, True)
(out['pixel_colors_nr'].sum() + out['pixel_colors_nr_fine'].sum()).backward()
params = list(r.parameters())
before = [None if p.grad is None else p.grad.clone() for p in params]
parallel.allreduce_gradients(params)
for p, b in zip(params, before):
    assert (p.grad is None) == (b is None) and (b is None or torch.equal(p.grad, b))
# steady state: the second and third call take the previous union of flags and copy their own flags to pinned memory without
# blocking (parallel._Pending on a CUDA tensor); the deferred check passes
for _ in range(2):
    parallel.allreduce_gradients(params)
    for p, b in zip(params, before):
        assert (p.grad is None) == (b is None) and (b is None or torch.equal(p.grad, b))
assert any('pending' in st for st in parallel._FLAG_STATE.values())
parallel.check_deferred_flags()
maps = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(1, 32, 12, 16, device=dev)) for _ in range(5)])
maps[1].grad = torch.ones_like(maps[1]); maps[3].grad = torch.full_like(maps[3], 2.0)
union = parallel.allreduce_scene_feature_gradients(maps, [3, 1])
assert union == [1, 3] and maps[0].grad is None and float(maps[3].grad.mean()) == 2.0
dist.barrier()
dist.destroy_process_group()
print('RCCL_WORLD1_OK')
''' % ROOT


@pytest.mark.gpu
def test_parallel_collectives_through_rccl_with_one_rank(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    port = 29600 + os.getpid() % 300
    res = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
                          '--master-port', str(port), str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0 and 'RCCL_WORLD1_OK' in res.stdout, (res.stdout[-2000:], res.stderr[-4000:])
