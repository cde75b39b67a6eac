import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'dynamic-video-depth_amd')); sys.path.insert(0, ROOT)
from dvd_hip import ops, synthetic
import bench
opt = bench.make_opt(depth_chunk=2, depth_graphs=False)
model = bench.build_model(opt, torch.device('cuda'), seed=0)
batch = synthetic.make_batch(2, 96, 160, gap=1, seed=1, device='cuda')
f0 = ops.executed_flops()
model._train_on_batch(6, 0, synthetic.with_loader_dim(batch))
torch.cuda.synchronize()
f1 = ops.executed_flops()
print({k: f1[k] - f0[k] for k in f1})
print('expected mlp per class', 2 * 2 * 96 * 160 * 593408.0)
