"""Shared helpers for the parity tests (test infrastructure)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CAM_KEYS = ('R_1', 'R_2', 'R_1_T', 'R_2_T', 't_1', 't_2', 'K', 'K_inv')


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    return {k: z[k] for k in z.files}


def t(a, device='cpu'):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def golden_batch(gd, device='cpu'):
    b = {}
    for k, v in gd.items():
        if k.startswith('in_') and k not in ('in_depth_1', 'in_depth_2', 'in_sf_1_2'):
            name = k[3:]
            b[name] = t(v) if name == 'time_step' else t(v, device)
    return b


def golden_opt(gd):
    from oracle.losses import default_opt
    o = {str(k): float(v) for k, v in zip(gd['opt_keys'], gd['opt_vals'])}
    for k in ('midas', 'use_disp', 'use_disp_ratio', 'time_dependent', 'use_cnn', 'warm_reg', 'weight_steps',
              'use_motion_seg'):
        if k in o:
            o[k] = bool(o[k])
    for k in ('interp_steps', 'n_freq_xyz', 'n_freq_t'):
        o[k] = int(o[k])
    return default_opt(**o)


def golden_mlp_sd(gd, device='cpu', prefix='sd_'):
    return {k[len(prefix):]: t(v, device) for k, v in gd.items() if k.startswith(prefix)}


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
