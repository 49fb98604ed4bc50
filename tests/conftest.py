import ast
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "fused_abn: tests/test_costreg_kernels.py - the test drives mvsnet.FUSED_ABN itself")


def load_case(name):
    """-> (cfg, que, ref, out, mid, extra) from tests/golden/case_<name>.npz"""
    z = np.load(os.path.join(GOLDEN_DIR, 'case_%s.npz' % name), allow_pickle=False)
    cfg = ast.literal_eval(str(z['cfg_json']))  # repr() of a plain dict written by make_golden.py
    que = {k[4:]: z[k] for k in z.files if k.startswith('que.')}
    ref = {k[4:]: z[k] for k in z.files if k.startswith('ref.')}
    out = {k[4:]: z[k] for k in z.files if k.startswith('out.')}
    mid = {k[4:]: z[k] for k in z.files if k.startswith('mid.')}
    extra = {'is_train': bool(z['is_train']), 'u': z['u'] if 'u' in z.files else None}
    return cfg, que, ref, out, mid, extra


def load_weights(vis=False):
    z = np.load(os.path.join(GOLDEN_DIR, 'weights_seed0_vis.npz' if vis else 'weights_seed0.npz'))
    return {k: z[k] for k in z.files}


def oracle_cfg(cfg):
    """reference yaml-style cfg -> oracle cfg (flattens the decoder use_vis flags)"""
    c = dict(cfg)
    c['coarse_use_vis'] = cfg.get('dist_decoder_cfg', {}).get('use_vis', True)
    c['fine_use_vis'] = cfg.get('fine_dist_decoder_cfg', {}).get('use_vis', True)
    return c


CASES = ['a_small', 'b_default', 'c_adversarial', 'd_train_vis', 'e_use_all']


def case_uses_vis_weights(name):
    return name == 'd_train_vis'
