"""Debug: which of (replay after a weight update, first replay after capture, eager) disagree in the checkpoint test."""
import os, sys, copy
from types import SimpleNamespace
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'dynamic-video-depth_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import warnings; warnings.simplefilter('ignore')
import helpers
from dvd_hip import synthetic
from dvd_hip.models.scene_flow_motion_field import Model

def model(seed, **over):
    o = dict(helpers.FULL_STEP_OPT); o.update(midas=False, full_logdir='/tmp', lr=1e-4); o.update(over)
    m = Model(SimpleNamespace(**o), None)
    helpers.seeded_fill_(m.net_depth, seed); helpers.seeded_fill_(m.net_sceneflow, seed + 1)
    m.to(torch.device('cuda')); return m

batch = synthetic.make_batch(2, 32, 48, gap=1, seed=9)
step = lambda m, i: m._train_on_batch(6, i, helpers.loader_batch(dict(batch)))
a = model(51); l0 = step(a, 0); print('a step0', l0['loss'])
a.save_state_dict('/tmp/ck.pt', save_optimizer=True)
e = model(77, depth_graphs=0); e.load_state_dict('/tmp/ck.pt'); print('eager from ckpt, step1', step(e, 1)['loss'])
b = model(77); b.load_state_dict('/tmp/ck.pt'); print('graphs (first capture) from ckpt, step1', step(b, 1)['loss'])
print('a replay, step1', step(a, 1)['loss'])
b2 = model(78); b2.load_state_dict('/tmp/ck.pt'); print('graphs again (third model) step1', step(b2, 1)['loss'])
e0 = model(51, depth_graphs=0); print('eager fresh seed51 step0', step(e0, 0)['loss'], 'step1', step(e0, 1)['loss'])
