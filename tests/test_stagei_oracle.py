"""Stage-I oracle self-checks (oracle/stagei_oracle.py) and the arithmetic of the Stage-I kernels, run through the g++ emulation build
of moshpp_amd/csrc/stagei.hip (tests/emu): same source as the GPU library, one sequential "thread" per block."""
import numpy as np
import pytest

from oracle import stageii_oracle as so
from oracle import stagei_oracle as s1
from tests import helpers


@pytest.fixture(scope='module')
def case():
    return helpers.stagei_case()


def _objective(case, fingers=True):
    m, M, nb = case['m'], case['M'], case['nb']
    m2b = np.ones(M) * 0.0095
    can = so.verts_forward(m, so.fullpose_from_pose(m, np.zeros(m['NP'])), np.zeros(3), None, shp=np.zeros(nb))
    ml0 = s1.markers_latent_init(can, case['faces'], case['vids'], m2b)
    root, body, finger, step1, step2 = so.pose_id_sets('smplh', m['NP'], optimize_fingers=fingers)
    obj = s1.StageIObjective(m, case['faces'], case['prior'], body, case['frames'], ml0, m2b, [(np.arange(M), 300.0)], nb)
    rng = np.random.default_rng(3)
    obj.pose[:, step2] = rng.normal(0, 0.2, (obj.F, len(step2)))
    obj.trans[:] = rng.normal(0, 0.3, (obj.F, 3))
    obj.betas = rng.normal(0, 0.5, nb)
    obj.ml = ml0 + rng.normal(0, 0.003, ml0.shape)
    a = 0.5
    obj.set_round(step2, finger, dict(anneal=a, data=(75 / a) * (46.0 / M), poseB=3 * a, poseH=3 * a, beta=10 * a, surf=1e4,
                                      init_head=300 * a))
    return obj


def test_markers_start_at_the_skin_distance(case):
    m = case['m']
    can = so.verts_forward(m, so.fullpose_from_pose(m, np.zeros(m['NP'])), np.zeros(3), None, shp=np.zeros(case['nb']))
    ml = s1.markers_latent_init(can, case['faces'], case['vids'], np.ones(case['M']) * 0.0095)
    d, tri, part = s1.signed_surface_distance(ml, can, case['faces'])
    assert (d <= 0.0095 + 1e-12).all() and (d > 0.009).all()   # a vertex pushed out along its normal: at most 9.5 mm from the surface


def test_surface_distance_gradients_match_finite_differences(case):
    m = case['m']
    can = so.verts_forward(m, so.fullpose_from_pose(m, np.zeros(m['NP'])), np.zeros(3), None, shp=np.zeros(case['nb']))
    rng = np.random.default_rng(1)
    ml = s1.markers_latent_init(can, case['faces'], case['vids'], np.ones(case['M']) * 0.0095)
    pts = ml + rng.normal(0, 0.01, ml.shape)                  # both sides of the surface, all part types
    d, tri, part, dp, dabc, fv = s1.signed_surface_distance(pts, can, case['faces'], want_jac=True)
    assert len(np.unique(part)) >= 3 and (d < 0).any() and (d > 0).any()
    h = 1e-7
    for k in range(3):
        e = np.zeros(3); e[k] = h
        fd = (s1.signed_surface_distance(pts + e, can, case['faces'])[0] - s1.signed_surface_distance(pts - e, can, case['faces'])[0]) / (2 * h)
        assert np.abs(fd - dp[:, k]).max() < 1e-6
    dv = rng.normal(0, 1, can.shape)
    fd = (s1.signed_surface_distance(pts, can + h * dv, case['faces'])[0] - s1.signed_surface_distance(pts, can - h * dv, case['faces'])[0]) / (2 * h)
    assert np.abs(fd - np.einsum('mva,mva->m', dabc, dv[fv])).max() < 1e-6


def test_objective_jacobian_matches_finite_differences(case):
    obj = _objective(case)
    x = obj.x()
    res, jac = obj.evaluate(x, want_J=True)
    F, M, npid = obj.F, obj.M, len(obj.pose_ids)
    blocks = {'trans': (0, 3 * F), 'ml': (3 * F, 3 * F + 3 * M), 'pose': (3 * F + 3 * M, 3 * F + 3 * M + F * npid),
              'beta': (3 * F + 3 * M + F * npid, len(x))}
    rng = np.random.default_rng(5)
    h = 1e-6
    for s, e in blocks.values():
        d = np.zeros(len(x)); d[s:e] = rng.normal(0, 1, e - s)
        rp, rm = obj.evaluate(x + h * d), obj.evaluate(x - h * d)
        for name in res:
            fd = (rp[name] - rm[name]) / (2 * h)
            an = jac[name].dot(d)
            assert np.abs(fd - an).max() < 1e-6 * max(1.0, np.abs(an).max()) + 2e-3 * (name == 'surf'), (name, s, e)


def test_stagei_fits_the_frames(case):
    out = s1.stagei_solve(case['m'], case['faces'], case['prior'], 'smplh', case['frames'], case['vids'], case['mask'], case['m2b'],
                          case['nb'])
    obj = out['objective']
    for f, (ids, obs) in enumerate(obj.frames):
        assert np.sqrt(((obj.markers_sim(f)[ids] - obs) ** 2).sum(1).mean()) < 5e-3
    d = s1.signed_surface_distance(out['markers_latent'], obj.can_verts(out['betas']), case['faces'])[0]
    assert np.abs(d - 0.0095).max() < 2e-3                    # the surface term keeps the markers near their skin distance
    assert np.linalg.norm(out['markers_latent'] - case['ml_gt'], axis=1).mean() < 0.012


@pytest.mark.parametrize('fingers', [False, True])
def test_kernel_arithmetic_matches_oracle_in_emulation(case, fingers):
    """stagei.hip compiled by g++ (tests/emu): same kernels and host dogleg as the GPU library, bit-for-bit the same control flow."""
    from tests.emu import emu_stagei
    c = case if not fingers else helpers.stagei_case(finger_markers=True, M=36, seed=2)
    kw = helpers.stagei_kwargs(c, optimize_fingers=fingers)
    out = emu_stagei.solve(c['m'], c['prior'], **kw)
    ref = s1.stagei_solve(c['m'], c['faces'], c['prior'], 'smplh', c['frames'], c['vids'], c['mask'], c['m2b'], c['nb'],
                          optimize_fingers=fingers)
    assert np.abs(out['betas'] - ref['betas']).max() < 1e-8
    assert np.abs(out['markers_latent'] - ref['markers_latent']).max() < 1e-9
    assert np.abs(out['pose'] - ref['pose']).max() < 1e-8 and np.abs(out['trans'] - ref['trans']).max() < 1e-9
    assert (out['markers_latent_vids'] == ref['markers_latent_vids']).all()
    e = ref['errs']
    want = [e['data'], e['poseB'], e['init_0'], e['beta'], e['surf'], e.get('poseH', 0.0)]
    assert np.allclose(out['errs'][:6], want, rtol=1e-7, atol=1e-12) and out['errs'][6] == 0.0


@pytest.mark.parametrize('name', ['mano', 'fixed_betas', 'head_corr', 'face', 'extra_rigid', 'collinear'])
def test_kernel_arithmetic_variants_in_emulation(name):
    """Other model families / options through the emulated kernels (the GPU tests run the full list)."""
    from tests.emu import emu_stagei
    from tests.test_gpu_stagei import _variant
    c, fingers, extra = _variant(name)
    out = emu_stagei.solve(c['m'], c['prior'], **helpers.stagei_kwargs(c, optimize_fingers=fingers, **extra))
    ref = s1.stagei_solve(c['m'], c['faces'], c['prior'], c['model_type'], c['frames'], c['vids'], c['mask'], c['m2b'], c['nb'],
                          optimize_fingers=fingers, **helpers.stagei_oracle_extra(extra))
    if name == 'face':
        assert np.abs(out['expression'] - ref['expression']).max() < 1e-8
    assert np.abs(out['markers_latent'] - ref['markers_latent']).max() < 1e-9
    assert np.abs(out['pose'] - ref['pose']).max() < 1e-8 and np.abs(out['trans'] - ref['trans']).max() < 1e-9
    if c['nb']:
        assert np.abs(out['betas'] - ref['betas']).max() < 1e-8


def test_mosh_stagei_host_path_on_cpu(tmp_path, monkeypatch):
    """The drop-in `mosh_stagei` (files in, dict out) with libmoshii's entry points replaced by the emulation build + oracle LBS:
    exercises the host code (layout, cfg switches, frame packing, weights, output dict) without a GPU."""
    import json
    import os
    import pickle
    from moshpp_amd import capi, chmosh, synth
    from moshpp_amd.cfg import make_cfg
    from tests.emu import emu_stagei

    class FakeModel:
        def __init__(self, v_template, shapedirs, posedirs, weights, J_regressor, parents, body_dof, hand_dof, hands_mean, comps):
            self.model = dict(v_template=v_template, shapedirs=shapedirs, posedirs=posedirs, weights=weights, J_regressor=J_regressor,
                              parents=parents, body_dof=body_dof, hand_dof=hand_dof, hands_mean=hands_mean, selected_components=comps)
            self.m = so.prepare_model(self.model)
            self.NP = self.m['NP']

        def set_betas(self, betas):
            self.m = so.prepare_model(self.model, betas)

        def lbs_forward(self, pose, trans):
            return np.array([so.verts_forward(self.m, so.fullpose_from_pose(self.m, p), t) for p, t in zip(pose, trans)])

    class FakePrior:
        def __init__(self, means, chols, weights):
            self.p = dict(means=means, chols=chols, weights=weights, npose=means.shape[1])

    def fake_solve(dev, prior, **kw):
        so.set_free_shape(dev.m, 0, kw['nb'])
        out = dict(emu_stagei.solve(dev.m, prior.p if prior is not None else None, **kw))
        out['errs'] = dict(zip(capi.STAGEI_ERR_NAMES, out['errs'].tolist()))
        out['iters'] = int(out['iters'][0])
        return out
    monkeypatch.setattr(capi, 'Model', FakeModel)
    monkeypatch.setattr(capi, 'Prior', FakePrior)
    monkeypatch.setattr(capi, 'stagei_solve_host', fake_solve)
    pb = synth.make_stagei_problem('smplh', n_verts=1500, nb=4, M=20, F=3, seed=21)
    raw = {k: v for k, v in pb['dd'].items() if not k.startswith('_')}
    with open(tmp_path / 'model.pkl', 'wb') as f:
        pickle.dump(raw, f)
    with open(tmp_path / 'prior.pkl', 'wb') as f:
        pickle.dump(pb['gmm'], f)
    np.savez(tmp_path / 'hands.npz', **synth.synth_hand_prior(21))
    labels = [f'MK{i:02d}' for i in range(20)]
    with open(tmp_path / 'layout.json', 'w') as f:
        json.dump({'surface_model_type': 'smplh', 'markersets': [
            {'type': 'body', 'indices': {l: int(v) for l, v in zip(labels, pb['vids'])}}]}, f)
    frames = [{labels[i]: xyz for i, xyz in zip(ids, obs)} for ids, obs in pb['frames']]
    frames[0]['UNKNOWN'] = np.zeros(3)                     # a label the layout does not know: ignored (chmosh.py:199-206)
    frames[1][labels[0]] = np.full(3, np.nan)              # NaN observation: dropped from that frame
    cfg = make_cfg(**{'surface_model.type': 'smplh', 'surface_model.fname': str(tmp_path / 'model.pkl'), 'surface_model.num_betas': 4,
                      'surface_model.dof_per_hand': 12, 'surface_model.use_hands_mean': False,
                      'moshpp.pose_body_prior_fname': str(tmp_path / 'prior.pkl'), 'moshpp.pose_hand_prior_fname': str(tmp_path / 'hands.npz'),
                      'moshpp.optimize_fingers': True,    # the layout has no finger markers -> switched off (:128-139)
                      'dirs.marker_layout.fname': str(tmp_path / 'layout.json')})
    res = chmosh.mosh_stagei(frames, cfg)
    assert cfg.moshpp.optimize_fingers is False
    assert set(res) == {'betas', 'markers_latent', 'latent_labels', 'marker_meta', 'markers_latent_vids', 'stagei_debug_details'}
    assert res['latent_labels'] == labels and res['betas'].shape == (10,) and np.all(res['betas'][4:] == 0)
    dbg = res['stagei_debug_details']
    assert set(dbg['stagei_errs']) == {'data', 'poseB', 'init_body', 'beta', 'surf'}
    assert [len(l) for l in dbg['stagei_labels_obs']] == [len(pb['frames'][0][0]), len(pb['frames'][1][0]) - (0 in pb['frames'][1][0]),
                                                          len(pb['frames'][2][0])]
    assert labels[0] not in dbg['stagei_labels_obs'][1]
    fit = np.sqrt(np.mean([((a - b) ** 2).sum(1).mean() for a, b in zip(dbg['stagei_markers_sim'], dbg['stagei_markers_obs'])]))
    assert fit < 5e-3
    assert set(res['markers_latent_vids']) == set(labels) and set(dbg['markers_latent_all_vids']) <= set(labels)
    pickle.dumps(res)
    # the pipeline driver: Stage-I frames picked from a capture, pickled, re-loaded on the second call
    from moshpp_amd.mosh_head import run_moshpp_once
    F, N = 30, 20
    cap = np.zeros((F, N, 3))
    rng = np.random.default_rng(2)
    for t in range(F):
        ids, obs = pb['frames'][t % 3]
        cap[t] = np.nan
        cap[t, ids] = obs + rng.normal(0, 1e-4, obs.shape)
    np.savez(tmp_path / 'capture.npz', markers=cap * 1000.0, labels=np.array(labels), frame_rate=120.0)
    cfg.mocap.fname = str(tmp_path / 'capture.npz')
    cfg.dirs.stagei_fname = str(tmp_path / 'res' / 'stagei.pkl')
    cfg.runtime.stagei_only = True
    cfg.moshpp.stagei_frame_picker.num_frames = 3
    cfg.moshpp.stagei_frame_picker.least_avail_markers = 0.5
    st1, st2 = run_moshpp_once(cfg)
    assert st2 is None and os.path.exists(cfg.dirs.stagei_fname) and len(st1['stagei_debug_details']['stagei_fnames']) == 3
    st1b, _ = run_moshpp_once(cfg)
    from moshpp_amd.mosh_head import dump_stagei_marker_layout
    from moshpp_amd.marker_layout import marker_layout_load
    opt = marker_layout_load(dump_stagei_marker_layout(cfg.dirs.stagei_fname))
    assert list(opt['marker_vids'].keys()) == labels
    assert [opt['marker_vids'][l] for l in labels] == [st1['markers_latent_vids'][l] for l in labels]      # optimised vertex ids
    assert np.array_equal(st1b['markers_latent'], st1['markers_latent'])       # loaded, not re-solved
    # optimize_betas = false with given betas: they are kept, no beta term
    np.savez(tmp_path / 'betas.npz', betas=np.array([0.5, -0.3, 0.2, 0.1]))
    cfg.moshpp.optimize_betas = False
    res2 = chmosh.mosh_stagei(frames, cfg, betas_fname=str(tmp_path / 'betas.npz'))
    assert np.allclose(res2['betas'][:4], [0.5, -0.3, 0.2, 0.1]) and 'beta' not in res2['stagei_debug_details']['stagei_errs']


@pytest.mark.parametrize('name', ['body', 'fingers'])
def test_oracle_reproduces_committed_stagei_golden(name):
    """tests/golden/stagei_golden.npz (tests/golden/make_stagei_golden.py): the oracle's Stage-I solution is pinned against drift."""
    import os
    from tests.golden.make_stagei_golden import CASES
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'stagei_golden.npz'))
    kw, fingers = CASES[name]
    c = helpers.stagei_case(**kw)
    ref = s1.stagei_solve(c['m'], c['faces'], c['prior'], 'smplh', c['frames'], c['vids'], c['mask'], c['m2b'], c['nb'],
                          optimize_fingers=fingers)
    assert np.abs(ref['betas'] - G[f'{name}_betas']).max() < 1e-9 and np.abs(ref['markers_latent'] - G[f'{name}_markers_latent']).max() < 1e-10
    assert np.abs(ref['pose'] - G[f'{name}_pose']).max() < 1e-9 and (ref['markers_latent_vids'] == G[f'{name}_markers_latent_vids']).all()


def test_closest_point_on_triangle_properties():
    """The exhaustive nearest-triangle search the oracle restates (psbody's aabbtree_nearest): the returned point lies in the triangle,
    no sampled point of the triangle is closer, and the part code agrees with where the point sits (interior / edge / vertex)."""
    rng = np.random.default_rng(0)
    n = 400
    a, b, c = rng.normal(0, 1, (3, n, 3))
    p = rng.normal(0, 1.5, (n, 3))
    q, part = s1._closest_on_triangles(p, a, b, c)
    # barycentric coordinates of q
    T = np.stack([b - a, c - a], axis=2)                      # n,3,2
    uv = np.array([np.linalg.lstsq(T[i], (q - a)[i], rcond=None)[0] for i in range(n)])
    u, v = uv[:, 0], uv[:, 1]
    w = 1 - u - v
    assert (u > -1e-9).all() and (v > -1e-9).all() and (w > -1e-9).all()
    assert np.abs(a + u[:, None] * (b - a) + v[:, None] * (c - a) - q).max() < 1e-9
    tol = 1e-9
    zero = np.stack([w < tol, u < tol, v < tol], axis=1)      # weight of a, b, c vanishes
    # part codes: 0 interior; 1 ab (c-weight 0), 2 bc (a-weight 0), 3 ca (b-weight 0); 4 a, 5 b, 6 c
    expect = {0: (False, False, False), 1: (False, False, True), 2: (True, False, False), 3: (False, True, False),
              4: (False, True, True), 5: (True, False, True), 6: (True, True, False)}
    for i in range(n):
        assert tuple(zero[i]) == expect[int(part[i])], (i, part[i], zero[i])
    assert set(part.tolist()) == set(range(7))
    # no point of a dense barycentric sampling is closer
    g = np.linspace(0, 1, 41)
    uu, vv = np.meshgrid(g, g)
    keep = uu + vv <= 1
    uu, vv = uu[keep], vv[keep]
    samples = a[:, None, :] + uu[None, :, None] * (b - a)[:, None, :] + vv[None, :, None] * (c - a)[:, None, :]
    dmin = np.sqrt(((samples - p[:, None, :]) ** 2).sum(-1)).min(1)
    assert (np.sqrt(((q - p) ** 2).sum(1)) <= dmin + 1e-12).all()


def test_nearest_on_mesh_equals_unpruned_search(case):
    """The candidate pruning of nearest_on_mesh (triangles with a vertex within d_nearest_vertex + longest edge) never changes the result."""
    m = case['m']
    can = so.verts_forward(m, so.fullpose_from_pose(m, np.zeros(m['NP'])), np.zeros(3), None, shp=np.zeros(case['nb']))
    f = case['faces']
    rng = np.random.default_rng(4)
    pts = can[rng.integers(0, len(can), 25)] + rng.normal(0, 0.03, (25, 3))
    tri, part, near = s1.nearest_on_mesh(pts, can, f)
    a, b, c = can[f[:, 0]], can[f[:, 1]], can[f[:, 2]]
    for i, p in enumerate(pts):
        q, pc = s1._closest_on_triangles(np.broadcast_to(p, (len(f), 3)), a, b, c)
        k = int(np.argmin(((q - p) ** 2).sum(1)))
        assert k == tri[i] and pc[k] == part[i] and np.array_equal(q[k], near[i])


def test_stagei_dogleg_keeps_descending_and_scipy_agrees_on_descent(case):
    """From the Stage-I solution (stopped by the reference's 1e-3 relative-improvement rule) both the dogleg and scipy's trust-region
    least squares still find lower cost with the same residual / Jacobian functions -- i.e. the Jacobian is a descent-consistent
    derivative of the residual.  (They need not end in the same place: the objective is only piecewise smooth, because the marker
    attachment and the nearest triangle are re-evaluated at every point, as in the reference; measured on this case: dogleg to
    convergence 1266.97, scipy 1230.32, start 1287.67.)"""
    import scipy.optimize
    out = s1.stagei_solve(case['m'], case['faces'], case['prior'], 'smplh', case['frames'], case['vids'], case['mask'], case['m2b'],
                          case['nb'])
    obj = out['objective']                       # still set up with the last round's weights and free variables
    x0 = obj.x()
    c0 = (obj.r(x0) ** 2).sum()
    xd = so.minimize_dogleg(obj, x0, e_3=0.0, delta_0=0.5, maxiter=15)
    sol = scipy.optimize.least_squares(obj.r, x0, jac=obj.J, method='trf', max_nfev=15)
    cd, cs = (obj.r(xd) ** 2).sum(), (sol.fun ** 2).sum()
    assert cd < c0 and cs < c0
    assert (c0 - cd) / c0 < 0.05 and (c0 - cs) / c0 < 0.10            # the solution was already close to a local minimum


@pytest.mark.parametrize('fingers', [False, True])
def test_schur_solver_equals_dense_solver_in_emulation(case, fingers, monkeypatch):
    """The arrow-structured solver (per-frame elimination + Schur complement on the shared block; the default) takes the same
    Gauss-Newton steps as the dense blocked Cholesky (MOSHII_S1_SOLVER=dense)."""
    from tests.emu import emu_stagei
    c = case if not fingers else helpers.stagei_case(finger_markers=True, M=36, seed=2)
    kw = helpers.stagei_kwargs(c, optimize_fingers=fingers)
    monkeypatch.setenv('MOSHII_S1_SOLVER', 'dense')
    a = emu_stagei.solve(c['m'], c['prior'], **kw)
    monkeypatch.setenv('MOSHII_S1_SOLVER', 'schur')
    b = emu_stagei.solve(c['m'], c['prior'], **kw)
    assert int(a['iters'][0]) == int(b['iters'][0])
    assert np.abs(a['betas'] - b['betas']).max() < 1e-10 and np.abs(a['markers_latent'] - b['markers_latent']).max() < 1e-11
    assert np.abs(a['pose'] - b['pose']).max() < 1e-10


@pytest.mark.parametrize('name', ['face', 'mano', 'fixed_betas'])
def test_schur_solver_variants_in_emulation(name, monkeypatch):
    """The arrow-structured solver with per-frame expression columns in the frame blocks (face), without a prior / body block (mano)
    and without shared betas (fixed_betas): same steps as the dense solver."""
    from tests.emu import emu_stagei
    from tests.test_gpu_stagei import _variant
    c, fingers, extra = _variant(name)
    kw = helpers.stagei_kwargs(c, optimize_fingers=fingers, **extra)
    monkeypatch.setenv('MOSHII_S1_SOLVER', 'dense')
    a = emu_stagei.solve(c['m'], c['prior'], **kw)
    monkeypatch.setenv('MOSHII_S1_SOLVER', 'schur')
    b = emu_stagei.solve(c['m'], c['prior'], **kw)
    assert int(a['iters'][0]) == int(b['iters'][0])
    assert np.abs(a['markers_latent'] - b['markers_latent']).max() < 1e-11 and np.abs(a['pose'] - b['pose']).max() < 1e-9
    if a['expression'].size:
        assert np.abs(a['expression'] - b['expression']).max() < 1e-10
