import os, sys
from types import SimpleNamespace
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'dynamic-video-depth_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import warnings; warnings.simplefilter('ignore')
import helpers
from dvd_hip import synthetic, ops
from dvd_hip.models.scene_flow_motion_field import Model

def model(seed, **over):
    o = dict(helpers.FULL_STEP_OPT); o.update(midas=False, full_logdir='/tmp', lr=1e-4); o.update(over)
    m = Model(SimpleNamespace(**o), None)
    helpers.seeded_fill_(m.net_depth, seed); helpers.seeded_fill_(m.net_sceneflow, seed + 1)
    m.to(torch.device('cuda')); return m

batch = synthetic.make_batch(2, 32, 48, gap=1, seed=9)
step = lambda m, i: m._train_on_batch(6, i, helpers.loader_batch(dict(batch)))
def info(tag, m, log):
    d1, d2 = m._last['depth_1'], m._last['depth_2']
    print('%-40s loss %.6f  depth_1 mean %.6f max %.4f  depth_2 mean %.6f  kept %s  capstate %s' % (
        tag, log['loss'], float(d1.mean()), float(d1.max()), float(d2.mean()),
        sorted((k[0], k[1]) for k, v in m._depth_graphs.items() if v is not None), ops._capture_state), flush=True)
a = model(51); info('a step0', a, step(a, 0))
a.save_state_dict('/tmp/ck.pt', save_optimizer=True)
e = model(77, depth_graphs=0); e.load_state_dict('/tmp/ck.pt'); info('eager from ckpt step1', e, step(e, 1))
b = model(77); b.load_state_dict('/tmp/ck.pt'); info('b graphs from ckpt step1 (after eager model)', b, step(b, 1))
c = model(77); c.load_state_dict('/tmp/ck.pt'); info('c graphs from ckpt step1 (right after b)', c, step(c, 1))
info('a replay step1', a, step(a, 1))
d = model(77); d.load_state_dict('/tmp/ck.pt'); info('d graphs from ckpt step1 (after a replay)', d, step(d, 1))
e2 = model(77, depth_graphs=0); e2.load_state_dict('/tmp/ck.pt'); info('eager again', e2, step(e2, 1))
f = model(77); f.load_state_dict('/tmp/ck.pt'); info('f graphs (after eager)', f, step(f, 1))
