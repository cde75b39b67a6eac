import os, sys
from types import SimpleNamespace
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'dynamic-video-depth_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import warnings; warnings.simplefilter('ignore')
import helpers
from dvd_hip import synthetic, ops
from dvd_hip.models.scene_flow_motion_field import Model
graphs = int(os.environ.get('DBG_GRAPHS', '1'))
def model(seed, **over):
    o = dict(helpers.FULL_STEP_OPT); o.update(midas=False, full_logdir='/tmp', lr=1e-4, depth_graphs=graphs); o.update(over)
    m = Model(SimpleNamespace(**o), None)
    helpers.seeded_fill_(m.net_depth, seed); helpers.seeded_fill_(m.net_sceneflow, seed + 1)
    m.to(torch.device('cuda')); return m
batch = synthetic.make_batch(2, 32, 48, gap=1, seed=9)
step = lambda m, i: m._train_on_batch(6, i, helpers.loader_batch(dict(batch)))
def info(tag, m, log):
    d1 = m._last['depth_1']
    print('  %-28s loss %.6f  depth_1 mean %.6f' % (tag, log['loss'], float(d1.mean())), flush=True)
print('DVD_AB=%s graphs=%d' % (os.environ.get('DVD_AB', ''), graphs))
a = model(51); info('a step0', a, step(a, 0))
a.save_state_dict('/tmp/ck.pt', save_optimizer=True)
b = model(77); b.load_state_dict('/tmp/ck.pt')
info('a step1', a, step(a, 1))
info('b step1 (want = a step1)', b, step(b, 1))
