"""Stage-II driver + AMASS export (moshpp_amd/mosh_head.py) against the behaviour of the reference's
MoSh.mosh_stageii / MoSh.load_as_amass_npz (src/moshpp/mosh_head.py:268-301, 444-541). CPU only: the solver is
replaced by a stub callable, which is exactly the injection point the reference offers."""
import os
import pickle

import numpy as np
import pytest

from moshpp_amd import cfg as mcfg
from moshpp_amd import mosh_head


def _stagei(M=5, with_vt=False):
    d = dict(markers_latent=np.arange(M * 3, dtype=float).reshape(M, 3), latent_labels=[f'L{i}' for i in range(M)],
             betas=np.linspace(0, 1, 16), marker_meta={'marker_type': {}, 'marker_type_mask': {}},
             markers_latent_vids={f'L{i}': i for i in range(M)}, stagei_debug_details={})
    if with_vt:
        d['stagei_debug_details']['v_template'] = np.zeros((7, 3))
        d['v_template_fname'] = '/somewhere/vt.ply'
    return d


def _stub_solver(calls, T=4, P=156):
    def f(**kw):
        calls.append(kw)
        return {'fullpose': np.arange(T * P, dtype=float).reshape(T, P), 'trans': np.ones((T, 3)),
                'stageii_debug_details': {'mocap_frame_rate': 120.0, 'mocap_time_length': T / 120.0,
                                          'markers_orig': np.zeros((T, 6, 3)), 'labels_orig': list('abcdef'),
                                          'markers_obs': [np.zeros((5, 3))] * T, 'labels_obs': [['L0']] * T,
                                          'markers_sim': [np.zeros((5, 3))] * T}}
    return f


def _cfg(tmp_path, **dot):
    c = mcfg.make_cfg(**{'mocap.fname': str(tmp_path / 'ds' / 'subj' / 'walk.c3d'), 'surface_model.type': 'smplh',
                         'surface_model.gender': 'female', 'surface_model.fname': '/models/smplh/female/model.pkl', **dot})
    return c


def test_run_stageii_calls_the_injected_solver_like_the_reference_and_merges(tmp_path):
    calls = []
    c = _cfg(tmp_path)
    st1 = _stagei(with_vt=True)
    out_fname = str(tmp_path / 'out' / 'walk_stageii.pkl')
    res = mosh_head.run_stageii(st1, c, out_fname, mosh_stageii_func=_stub_solver(calls))
    assert len(calls) == 1
    assert set(calls[0]) == {'mocap_fname', 'cfg', 'markers_latent', 'latent_labels', 'betas', 'marker_meta',
                             'v_template_fname'}                       # mosh_head.py:280-286
    assert calls[0]['mocap_fname'] == c.mocap.fname and calls[0]['v_template_fname'] == '/somewhere/vt.ply'
    for k in st1:                                                        # stagei data merged in (:289)
        assert k in res
    dbg = res['stageii_debug_details']
    assert dbg['stageii_elapsed_time'] >= 0 and isinstance(dbg['cfg'], dict) and not isinstance(dbg['cfg'], mcfg.Cfg)
    assert dbg['cfg']['surface_model']['gender'] == 'female'
    with open(out_fname, 'rb') as fh:
        again = pickle.load(fh)
    assert np.array_equal(again['fullpose'], res['fullpose'])
    # a second call loads the pickle instead of solving (:272-274)
    res2 = mosh_head.run_stageii(st1, c, out_fname, mosh_stageii_func=_stub_solver(calls))
    assert len(calls) == 1 and np.array_equal(res2['trans'], res['trans'])


def test_run_stageii_requires_stagei_results(tmp_path):
    with pytest.raises(ValueError, match='please run stagei first'):
        mosh_head.run_stageii(str(tmp_path / 'missing_stagei.pkl'), _cfg(tmp_path), None, mosh_stageii_func=lambda **k: {})


@pytest.mark.parametrize('include_markers', [False, True])
def test_amass_npz_keys_and_side_files(tmp_path, include_markers):
    calls = []
    c = _cfg(tmp_path)
    res = mosh_head.run_stageii(_stagei(with_vt=True), c, None, mosh_stageii_func=_stub_solver(calls))
    npz_fname = str(tmp_path / 'amass' / 'walk_stageii.npz')
    d = mosh_head.load_as_amass_npz(res, npz_fname, include_markers=include_markers, include_extra_details=True)
    base = {'gender', 'surface_model_type', 'mocap_frame_rate', 'mocap_time_length', 'markers_latent', 'latent_labels',
            'markers_latent_vids', 'trans', 'poses', 'surface_model_fname', 'v_template', 'betas', 'num_betas',
            'root_orient', 'pose_body', 'pose_hand'}
    mk = {'markers', 'labels', 'markers_obs', 'labels_obs', 'markers_sim', 'marker_meta', 'num_markers'}
    assert set(d) == (base | mk if include_markers else base)
    assert d['poses'].shape == (4, 156) and d['pose_body'].shape == (4, 63) and d['pose_hand'].shape == (4, 90)
    assert np.array_equal(np.hstack([d['root_orient'], d['pose_body'], d['pose_hand']]), d['poses'])
    assert d['betas'].shape == (c.surface_model.num_betas,)
    if include_markers:
        assert d['num_markers'] == 6
    z = np.load(npz_fname, allow_pickle=True)
    assert set(z.files) == set(d)
    s1 = np.load(os.path.join(os.path.dirname(npz_fname), 'female_stagei.npz'), allow_pickle=True)   # :525-539
    assert set(s1.files) == {'gender', 'surface_model_type', 'markers_latent', 'latent_labels', 'markers_latent_vids',
                             'betas', 'v_template'}
    # existing files are left alone (:521, 528)
    mt = os.path.getmtime(npz_fname)
    mosh_head.load_as_amass_npz(res, npz_fname)
    assert os.path.getmtime(npz_fname) == mt


def test_amass_npz_from_pickle_file_and_smplx_parts(tmp_path):
    c = _cfg(tmp_path, **{'surface_model.type': 'smplx', 'moshpp.optimize_betas': False})
    res = mosh_head.run_stageii(_stagei(), c, str(tmp_path / 'x_stageii.pkl'),
                                mosh_stageii_func=_stub_solver([], P=165))
    d = mosh_head.load_as_amass_npz(str(tmp_path / 'x_stageii.pkl'))
    assert 'betas' not in d and 'v_template' not in d
    assert d['pose_jaw'].shape == (4, 3) and d['pose_eye'].shape == (4, 6) and d['pose_hand'].shape == (4, 90)
    assert np.array_equal(d['pose_jaw'], res['fullpose'][:, 66:69])
