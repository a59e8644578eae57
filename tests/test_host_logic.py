"""CPU tests of the host side (ingest, model loading, attachment, prior, cfg, free-variable sets)."""
import os
import pickle

import numpy as np
import pytest

from moshpp_amd import synth
from moshpp_amd.c3d_io import read_c3d, write_c3d
from moshpp_amd.cfg import make_cfg
from moshpp_amd.chmosh import stageii_pose_ids
from moshpp_amd.mocap_interface import MocapSession, read_mocap, write_mocap_c3d
from moshpp_amd.models import load_surface_model
from moshpp_amd.prior import create_gmm_body_prior
from moshpp_amd.transformed_lm import TransformedCoeffs
from oracle import stageii_oracle as so
from tests.helpers import oracle_case

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def test_c3d_reader_on_file_from_reference_writer():
    """The fixture was written by the reference's vendored py-c3d Writer (tests/golden/make_c3d_fixture.py)."""
    c = read_c3d(os.path.join(GOLDEN, 'refwriter_6x41.c3d'))
    e = np.load(os.path.join(GOLDEN, 'refwriter_6x41_expected.npz'))
    assert c['points'].shape == (6, 41, 3) and c['frame_rate'] == 100.0
    assert [l for l in c['labels']] == [str(l) for l in e['labels']]
    inv = np.isnan(c['points']).any(-1)
    np.testing.assert_array_equal(inv, e['invalid'])
    np.testing.assert_array_equal(c['points'][~inv], e['points'][~inv].astype(np.float64))


def test_c3d_roundtrip_and_mocap_session_rules(tmp_path):
    rng = np.random.default_rng(0)
    F, N = 7, 6
    m = rng.normal(0, 0.5, (F, N, 3))            # metres
    m[2, 1] = np.nan                              # NaN sample -> invalid
    m[3, 2] = 0.0                                 # exact zero sample -> invalid
    labels = ['LFHD', 'subj:RFHD', 'HEAD_TOP', '*7', 'L SHO', 'LFHD']   # prefix, alias, star, space, duplicate
    fn = str(tmp_path / 'a.c3d')
    write_mocap_c3d(m, labels, fn, frame_rate=60)
    rec = read_mocap(fn)
    assert rec['frame_rate'] == 60.0 and rec['markers'].shape == (F, N, 3)
    assert set(rec['subject_mask']) == {'null', 'subj'}
    ms = MocapSession(fn, 'mm', labels_map={'HEAD_TOP': 'ARIEL'})
    assert ms.labels == ['LFHD', 'RFHD', 'ARIEL', 'LSHO', 'LFHD'] and ms.read_status
    assert ms.markers.shape == (F, 5, 3) and ms.frame_rate == 60.0
    good = ~np.isnan(m).any(-1) & ~(m == 0).all(-1)
    keep = [0, 1, 2, 4, 5]
    np.testing.assert_allclose(ms.markers[good[:, keep]], m[:, keep][good[:, keep]], atol=1e-6)   # float32 mm in the file
    assert (ms.markers[~good[:, keep]] == 0).all()
    d = ms.markers_asdict()
    assert 'RFHD' not in d[2] and 'ARIEL' not in d[3] and len(d[0]) == 4   # duplicate label collapses in the dict
    obs, vis = ms.markers_aslabeled_arrays(['ARIEL', 'LFHD', 'NOPE', 'RFHD'])
    for t in range(F):
        for j, l in enumerate(['ARIEL', 'LFHD', 'NOPE', 'RFHD']):
            assert vis[t, j] == (l in d[t])
            if vis[t, j]:
                np.testing.assert_array_equal(obs[t, j], d[t][l])
    # units, rotation, npz / pkl containers
    ms2 = MocapSession(fn, 'm', mocap_rotate=[90, 0, 0], ignore_stared_labels=False)
    assert len(ms2.labels) == 6
    np.testing.assert_allclose(ms2.markers[0, 0], 1000 * np.array([m[0, 0, 0], -m[0, 0, 2], m[0, 0, 1]]), atol=1e-3)
    np.savez(tmp_path / 'b.npz', markers=m * 1000, labels=np.array(labels), frame_rate=90.)
    ms3 = MocapSession(str(tmp_path / 'b.npz'), 'mm')
    assert ms3.frame_rate == 90.0 and ms3.labels == ['LFHD', 'RFHD', 'HEAD_TOP', 'LSHO', 'LFHD']
    with open(tmp_path / 'c.pkl', 'wb') as f:
        pickle.dump({'markers': m, 'labels': labels, 'frame_rate': 30}, f)
    ms4 = MocapSession(str(tmp_path / 'c.pkl'), 'm', only_subjects=['subj'])
    assert ms4.labels == ['RFHD'] and ms4.markers.shape == (F, 1, 3) and not ms4.multi_subject
    assert abs(ms4.time_length() - F / 30) < 1e-12
    with pytest.raises(ValueError):
        read_mocap('nope.xyz')


def test_large_c3d_many_labels(tmp_path):
    rng = np.random.default_rng(1)
    pts = rng.normal(0, 100, (3, 300, 3))
    labels = [f'L{i:03d}' for i in range(300)]
    fn = str(tmp_path / 'big.c3d')
    write_c3d(fn, pts, labels, frame_rate=120.0)
    c = read_c3d(fn)
    assert c['labels'] == labels and c['points'].shape == (3, 300, 3)
    np.testing.assert_allclose(c['points'], pts.astype(np.float32), rtol=0, atol=0)


class _FakeCh:   # pickles as chumpy.ch.Ch would: class path 'chumpy.ch.Ch', state dict with 'x'
    pass


def test_model_loader_types_and_chumpy_free_pickle(tmp_path):
    import sys
    import types
    dd = synth.synth_model('smplh', seed=3)
    hp = synth.synth_hand_prior(3)
    raw = {k: v for k, v in dd.items() if not k.startswith('_')}
    # emulate a real SMPL pickle: shapedirs stored as a chumpy object
    mod = types.ModuleType('chumpy'); sub = types.ModuleType('chumpy.ch')
    Ch = type('Ch', (), {'__module__': 'chumpy.ch'})
    sub.Ch = Ch; mod.ch = sub
    sys.modules['chumpy'] = mod; sys.modules['chumpy.ch'] = sub
    try:
        obj = Ch(); obj.__dict__['x'] = raw['shapedirs']
        raw_ch = dict(raw); raw_ch['shapedirs'] = obj
        fn = str(tmp_path / 'model.pkl')
        with open(fn, 'wb') as f:
            pickle.dump(raw_ch, f)
    finally:
        del sys.modules['chumpy'], sys.modules['chumpy.ch']
    np.savez(tmp_path / 'hand.npz', **hp)
    sm = load_surface_model(fn, pose_hand_prior_fname=str(tmp_path / 'hand.npz'), use_hands_mean=True, dof_per_hand=24)
    assert sm.model_type == 'smplh' and sm.body_dof == 66 and sm.hand_dof == 48 and sm.NP == 114 and sm.K == 52
    np.testing.assert_array_equal(sm.shapedirs, dd['shapedirs'])
    np.testing.assert_array_equal(sm.parents, synth.kintree_parents('smplh'))
    assert sm.selected_components.shape == (48, 90) and (sm.selected_components[:24, 45:] == 0).all()
    fp = sm.fullpose(np.zeros(114))
    np.testing.assert_allclose(fp[66:], np.r_[hp['hands_meanl'], hp['hands_meanr']])
    sm0 = load_surface_model(raw, pose_hand_prior_fname=hp, use_hands_mean=False, dof_per_hand=12)
    assert sm0.hand_dof == 24 and (sm0.hands_mean == 0).all()
    for mt, bd, hd in (('smpl', 72, 0), ('smplx', 75, 48), ('mano', 3, 24)):
        d2 = {k: v for k, v in synth.synth_model(mt, seed=1, num_betas=4).items() if not k.startswith('_')}
        s2 = load_surface_model(d2, pose_hand_prior_fname=hp, use_hands_mean=True, dof_per_hand=24)
        assert (s2.model_type, s2.body_dof, s2.hand_dof) == (mt, bd, hd)
    mano = {k: v for k, v in synth.synth_model('mano', seed=1, num_betas=4).items() if not k.startswith('_')}
    assert (load_surface_model(mano, use_hands_mean=True, dof_per_hand=24).hands_mean == 0).all()   # inverted flag (:114)
    with pytest.raises(AssertionError):
        load_surface_model(raw, pose_hand_prior_fname=None)


def test_attachment_and_prior_match_oracle():
    case = oracle_case('smplh', F=2, M=53, seed=5)
    tc = TransformedCoeffs(case['can'], case['s']['markers_latent'])
    np.testing.assert_array_equal(tc.closest, case['closest'])
    np.testing.assert_allclose(tc.coef, case['coef'], atol=1e-15)
    pr = create_gmm_body_prior(case['s']['gmm'], exclude_hands=True)
    for k in ('means', 'chols', 'weights'):
        np.testing.assert_allclose(pr[k], case['prior'][k], rtol=1e-13)
    assert pr['npose'] == 63 and create_gmm_body_prior(case['s']['gmm'])['npose'] == 69


def test_attachment_collinear_fallback_and_eyeballs():
    # three collinear nearest neighbours for one marker -> the 3rd neighbour is swapped for ALL markers (:94-101)
    body = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0], [0, 1.5, 0], [5, 5, 5], [5, 6, 5], [6, 5, 5], [9, 9, 9.]] + [[20 + i, 0, 0] for i in range(3)])
    mk = np.array([[0.1, 0.05, 0.0], [5.1, 5.1, 5.2]])
    tc = TransformedCoeffs(body, mk)
    ref_closest, ref_coef = so.transformed_coeffs(body, mk)
    np.testing.assert_array_equal(tc.closest, ref_closest)
    np.testing.assert_allclose(tc.coef, ref_coef)
    assert tc.closest[0, 2] == 3          # not the collinear vertex 2
    big = np.random.default_rng(0).normal(0, 1, (10475, 3))
    big[9383:] = 0.0                       # eyeballs sit exactly on the marker...
    tcb = TransformedCoeffs(big, np.zeros((1, 3)))
    assert tcb.closest.max() < 9383        # ...but are never used (:48-50, 67-69)


def test_pose_id_sets_and_cfg():
    for mt, NP in (('smpl', 72), ('smplh', 114), ('smplx', 123), ('mano', 27)):
        for fingers in (False, True):
            for toes in (False, True):
                a = stageii_pose_ids(mt, NP, fingers, toes)
                root, body, finger, st1, st2 = so.pose_id_sets(mt, NP, fingers, toes)
                assert a['step1'] == list(st1) and a['step2'] == list(st2) and a['body'] == list(body)
    ids = stageii_pose_ids('smplh', 114, False, False)
    assert len(ids['step1']) == 60 and 30 not in ids['step1'] and 35 not in ids['step1'] and len(ids['body']) == 63
    assert len(stageii_pose_ids('smplx', 123, True, False)['step2']) == 60 + 48
    cfg = make_cfg(**{'surface_model.type': 'smplh', 'mocap.fname': '/a/b/c.c3d', 'moshpp.optimize_fingers': True})
    assert cfg.opt_settings.weights.stageii_wt_data == 400 and cfg['moshpp']['optimize_fingers'] is True
    cfg.moshpp['optimize_fingers'] = False
    assert cfg.moshpp.optimize_fingers is False and cfg.opt_settings.maxiter == 100
    with pytest.raises(KeyError):
        make_cfg(**{'surface_model.type': 'smpl'})   # the reference yaml has no smpl weights table either
    assert make_cfg(**{'surface_model.type': 'smpl', 'opt_settings.weights_type': 'smplh'}).opt_settings.weights.stageii_wt_velo == 2.5


def test_partition_units():
    from moshpp_amd.parallel import partition_units
    parts = partition_units([4000] * 32, 8)
    assert sorted(sum(parts, [])) == list(range(32)) and all(len(p) == 4 for p in parts)
    parts = partition_units([10, 1, 1, 1, 7, 3], 2)
    loads = [sum([10, 1, 1, 1, 7, 3][i] for i in p) for p in parts]
    assert sorted(sum(parts, [])) == list(range(6)) and abs(loads[0] - loads[1]) <= 1


@pytest.mark.parametrize('variant,tol', [('mips_float', 0.0), ('dec_float', 0.0), ('intel_int', 1e-4), ('mips_int', 1e-4)])
def test_c3d_reader_on_processor_and_int_variants(variant, tol):
    """The other on-disk forms of C3D (big-endian MIPS, DEC floats, scaled int16 points): files written by our writer,
    parsed by the reference's vendored reader when the fixture was made (tests/golden/make_c3d_variants.py); our reader must
    return the same coordinates and the same invalid mask, and both must match what was written to format precision."""
    from moshpp_amd import c3d_io
    g = np.load(os.path.join(GOLDEN, 'c3d_variants_expected.npz'))
    d = c3d_io.read_c3d(os.path.join(GOLDEN, f'variant_{variant}.c3d'))
    inv_ref = g[f'{variant}_invalid']
    assert np.array_equal(np.isnan(d['points']).any(-1), inv_ref)
    assert d['labels'] == [str(x) for x in g['labels']] and d['frame_rate'] == 100.0 == float(g[f'{variant}_rate'])
    ours = d['points'][~inv_ref]
    # the two readers agree: exactly on float files; to float32 rounding on int files (the reference scales in float32)
    assert np.abs(ours - g[f'{variant}_xyz'][~inv_ref]).max() <= tol
    written = g['points_written']
    assert np.array_equal(np.isnan(written).any(-1), inv_ref)
    assert np.abs(ours - written[~inv_ref]).max() < (0.051 if variant.endswith('int') else 1e-4)


def test_c3d_writer_rejects_out_of_range_int_scale(tmp_path):
    from moshpp_amd import c3d_io
    with pytest.raises(ValueError):
        c3d_io.write_c3d(str(tmp_path / 'x.c3d'), np.full((1, 1, 3), 5000.0), ['A'], int_scale=0.1)


def test_face_and_dmpl_host_pieces(tmp_path):
    """optimize_face id sets (chmosh.py:560-563, 685-689) agree with the oracle's; DMPL directions load from pkl / npz / dict
    (chmosh.py:511: `pickle.load(f)['eigvec']`)."""
    from moshpp_amd.chmosh import read_dmpl_pcs
    a = stageii_pose_ids('smplx', 123, True, False, optimize_face=True)
    _, _, _, st1, st2 = so.pose_id_sets('smplx', 123, True, False, optimize_face=True)
    assert a['face'] == [66, 67, 68] == so.face_pose_ids('smplx', True)
    assert a['step2'] == list(st2) and a['step1'] == list(st1) and len(a['step2']) == 60 + 48 + 3
    assert stageii_pose_ids('smplh', 114, False, False, optimize_face=True)['face'] == []      # only SMPL-X has a jaw
    assert so.face_pose_ids('smplh', True) == []
    ev = np.random.default_rng(0).normal(0, 1, (7, 3, 9))
    with open(tmp_path / 'd.pkl', 'wb') as f:
        pickle.dump({'eigvec': ev}, f, protocol=2)
    np.savez(tmp_path / 'd.npz', eigvec=ev)
    for src in (str(tmp_path / 'd.pkl'), str(tmp_path / 'd.npz'), {'eigvec': ev}):
        assert np.array_equal(read_dmpl_pcs(src), ev)


def test_c3d_roundtrip_property_over_every_variant(tmp_path):
    """Property test (hypothesis): any small point cloud with NaN drop-outs survives write -> read in every on-disk variant
    (Intel / MIPS / DEC floats, Intel / MIPS scaled int16): same invalid mask, coordinates to format precision, labels."""
    from hypothesis import given, settings, strategies as st
    from moshpp_amd import c3d_io
    counter = [0]

    @settings(max_examples=40, deadline=None)
    @given(F=st.integers(1, 4), N=st.integers(1, 9), seed=st.integers(0, 10 ** 6),
           proc=st.sampled_from([c3d_io.PROC_INTEL, c3d_io.PROC_MIPS, c3d_io.PROC_DEC]),
           int_scale=st.sampled_from([None, None, 0.05, 0.25]), rate=st.sampled_from([60.0, 100.0, 120.0, 239.76]))
    def run(F, N, seed, proc, int_scale, rate):
        if proc == c3d_io.PROC_DEC and int_scale is not None:
            int_scale = None                                    # (DEC + int is legal; the int path is endian-only)
        rng = np.random.default_rng(seed)
        pts = np.clip(rng.normal(0, 500, (F, N, 3)), -1500, 1500)    # inside the int16 range at the smallest scale (0.05 mm)
        pts[rng.random((F, N)) < 0.2] = np.nan
        labels = [f'L{seed % 7}_{i}' for i in range(N)]
        counter[0] += 1
        fn = str(tmp_path / f'p{counter[0]}.c3d')
        c3d_io.write_c3d(fn, pts, labels, frame_rate=rate, processor=proc, int_scale=int_scale)
        d = c3d_io.read_c3d(fn)
        assert d['labels'] == labels and abs(d['frame_rate'] - rate) < 1e-3
        inv = np.isnan(pts).any(-1)
        assert np.array_equal(np.isnan(d['points']).any(-1), inv)
        tol = (int_scale * 0.5 + 1e-3) if int_scale else 1e-3
        assert np.abs(d['points'][~inv] - pts[~inv]).max(initial=0.0) <= tol

    run()


def test_frame_ranges_and_single_rank_sharded_solve():
    """frame_ranges partitions [0, F) into contiguous near-equal ranges; without a process group solve_sequence_sharded is one call
    over the whole sequence with the first-frame schedule."""
    from hypothesis import given, settings, strategies as st
    from moshpp_amd.parallel import frame_ranges, solve_sequence_sharded

    @settings(max_examples=60, deadline=None)
    @given(F=st.integers(1, 100000), world=st.integers(1, 16))
    def prop(F, world):
        r = frame_ranges(F, world)
        assert len(r) == world and r[0][0] == 0 and r[-1][1] == F
        assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
        sizes = [b - a for a, b in r]
        assert max(sizes) - min(sizes) <= 1 and all(s >= 0 for s in sizes)

    prop()
    calls = []

    def solve_range(a, b, init):
        calls.append((a, b, init))
        return dict(pose=np.zeros((b - a, 5)), trans=np.zeros((b - a, 3)), status=np.zeros(b - a, dtype=np.int32))

    out, info = solve_sequence_sharded(solve_range, 37, dist=None)
    assert calls == [(0, 37, None)] and out['pose'].shape == (37, 5)
    assert info == dict(range=(0, 37), rounds=0, repaired=[], max_handoff_dev=0.0)


def test_host_level_chunk_scheme_reproduces_the_sequential_chain():
    """parallel.solve_sequence_chunked_host with a stand-in chain that carries an extra per-frame state ('shape', like the expression /
    DMPL coefficients of the extended kernel): a fresh start that lands in another basin is detected at the hand-off and repaired from
    the left neighbour's end state, a repair that changes a chunk's end state cascades to the right neighbour in the next round, and
    the stitched result equals the sequential chain."""
    from moshpp_amd.parallel import solve_sequence_chunked_host
    F = 400

    def tgt(g):
        return np.array([np.sin(0.05 * g), np.cos(0.03 * g), 0.01 * g, 1.0])

    def chain(a, b, init):
        n = b - a
        pose = np.zeros((n, 4)); shape = np.zeros((n, 2))
        fresh = init is None
        p1 = np.zeros(4) if fresh else np.array(init['pose'])
        s1 = np.zeros(2) if fresh else np.array(init.get('shape', np.zeros(2)))
        wrong = False
        for t in range(n):
            g = a + t
            if fresh and t == 0 and 180 <= g <= 260:
                wrong = True                                  # a fresh start in this stretch converges to another basin ...
            if g > 300:
                wrong = False                                 # ... which only merges with the right one at frame 301
            off = np.array([0.4, 0, 0, 0]) if wrong else 0.0
            p = tgt(g) + off + (0.02 if (fresh and t == 0) else 0.45 * (p1 - tgt(g - 1) - off))
            s = 0.9 * s1 + 0.1 * np.array([np.sin(0.02 * g), 1.0]) if not (fresh and t == 0) else np.array([0.3, 0.3])
            pose[t], shape[t] = p, s
            p1, s1 = p, s
        return dict(pose=pose, trans=pose[:, :3] * 0.1, shape=shape, status=np.zeros(n, dtype=np.int32))

    calls = []

    def solve_ranges(items):
        calls.append([(a, b, init is not None) for a, b, init in items])
        return [chain(a, b, init) for a, b, init in items]

    # the slowly forgetting 'shape' state needs a long warm-up to verify at 1e-9 (0.9^n): with 32 frames EVERY hand-off misses on it
    out, info = solve_sequence_chunked_host(solve_ranges, F, n_chunks=8, warmup=32, verify_tol=1e-9, state_keys=('pose', 'trans', 'shape'))
    ref = chain(0, F, None)
    assert np.abs(out['pose'] - ref['pose']).max() < 1e-8 and np.abs(out['shape'] - ref['shape']).max() < 1e-8
    assert info['n_chunks'] == 8 and info['repaired'][0] == list(range(1, 8)) and info['rounds'] == 4     # the cascade dies out once
    # the remaining difference of the shape state has decayed below the tolerance (0.9^n)
    # a state that forgets quickly: only the wrong-basin chunk (start 200 - 32 = 168 ... fresh at 168 is outside 180..260; chunk 5
    # starts fresh at 250 - 32 = 218, inside) is repaired
    calls.clear()
    out2, info2 = solve_sequence_chunked_host(solve_ranges, F, n_chunks=8, warmup=32, verify_tol=1e-9, state_keys=('pose', 'trans'))
    assert np.abs(out2['pose'] - ref['pose']).max() < 1e-8
    assert info2['repaired'] == [[5, 6], [6]], info2
    assert all(not has_init for _, _, has_init in calls[0]) and all(has_init for _, _, has_init in calls[1])


def test_marker_layout_roundtrip_filter_and_pose_id_sets(tmp_path):
    """marker_layout_write -> marker_layout_load round trip, marker_meta_filter, and the Stage-I free-variable sets of every model
    family against the oracle's (chmosh.py:281-310, 383-388)."""
    from moshpp_amd.chmosh import stagei_pose_ids
    from moshpp_amd.marker_layout import marker_layout_load, marker_layout_write, marker_meta_filter
    from oracle import stageii_oracle as so
    meta = marker_layout_load({'surface_model_type': 'smplx', 'markersets': [
        {'type': 'face', 'distance_from_skin': 0.0002, 'indices': {'CHIN': 8800, 'NOSE': 8970}},
        {'type': 'body', 'indices': {'C7': 3832, 'T10': 5623, 'STRN': 5532}}]}, labels_map={})
    fn = str(tmp_path / 'sub' / 'layout.json')
    marker_layout_write(meta, fn)
    back = marker_layout_load(fn, labels_map={})
    assert list(back['marker_vids'].items()) == list(meta['marker_vids'].items())
    assert {k: v.tolist() for k, v in back['marker_type_mask'].items()} == {k: v.tolist() for k, v in meta['marker_type_mask'].items()}
    assert dict(back['m2b_distance']) == {'body': 0.0095, 'face': 0.0002} and back['surface_model_type'] == 'smplx'
    sub = marker_meta_filter(meta, ['C7', 'NOSE'])
    assert list(sub['marker_vids']) == ['C7', 'NOSE'] and sub['marker_type_mask']['body'] == [True, False]
    assert set(sub['marker_colors']) == {'C7', 'NOSE', 'nan'}
    for mt, NP in (('smpl', 72), ('smplh', 66 + 24), ('smplx', 75 + 48), ('mano', 3 + 24)):
        for fingers in (False, True):
            pose_ids, body, finger = stagei_pose_ids(mt, NP, fingers, False)
            root, obody, ofinger, step1, _ = so.pose_id_sets(mt, NP, optimize_fingers=fingers)
            assert pose_ids == step1 and body == obody
            assert finger == (ofinger if (fingers or mt == 'mano') else [])


def test_prepare_stagei_frames_dispatch(tmp_path):
    """mosh_head.prepare_stagei_frames: picker type and arguments come from cfg (mosh_head.py:156-197)."""
    from moshpp_amd.cfg import make_cfg
    from moshpp_amd.mosh_head import prepare_stagei_frames
    rng = np.random.default_rng(0)
    mk = rng.normal(0, 500, (20, 6, 3))
    mk[3, 2] = np.nan
    fn = str(tmp_path / 'cap.npz')
    np.savez(fn, markers=mk, labels=np.array([f'L{i}' for i in range(6)]), frame_rate=100.0)
    cfg = make_cfg(**{'surface_model.type': 'smplh', 'mocap.fname': fn, 'moshpp.stagei_frame_picker.type': 'manual'})
    frames, names = prepare_stagei_frames(cfg, [f'{fn}_3', f'{fn}_7'])
    assert [os.path.basename(n) for n in names] == ['cap.npz_000003', 'cap.npz_000007'] and 'L2' not in frames[0] and len(frames[1]) == 6
    assert np.allclose(frames[1]['L0'], mk[7, 0] / 1000.0)
    cfg.moshpp.stagei_frame_picker.type = 'random_strict'
    cfg.moshpp.stagei_frame_picker.num_frames = 4
    frames, names = prepare_stagei_frames(cfg, [fn])
    assert len(frames) == 4 and all(len(f) == 6 for f in frames)              # least_avail_markers = 1.0: the NaN frame is never picked
    cfg.moshpp.stagei_frame_picker.type = 'nope'
    with pytest.raises(ValueError):
        prepare_stagei_frames(cfg, [fn])


def test_the_default_chain_mode_is_chunked_only_for_long_body_only_solves():
    """mosh_stageii's default ('auto', chmosh.StageIISolver.choose_chain_mode): a body-only solve of a long capture is cut into
    verified chunks; free finger / face / DMPL / shape coefficients (long memory: every fresh chunk start misses) and short captures
    stay on the sequential chain; an explicit request is never overridden."""
    from types import SimpleNamespace
    from moshpp_amd.chmosh import StageIISolver
    from moshpp_amd.cfg import default_cfg
    body = SimpleNamespace(optimize_fingers=False, optimize_face=False, optimize_dynamics=False, n_shape=0, AUTO_MIN_FRAMES=StageIISolver.AUTO_MIN_FRAMES)
    pick = StageIISolver.choose_chain_mode
    assert pick(body, 4000) == 'chunked' and pick(body, StageIISolver.AUTO_MIN_FRAMES) == 'chunked'
    assert pick(body, StageIISolver.AUTO_MIN_FRAMES - 1) == 'sequential' and pick(body, 0) == 'sequential'
    for k in ('optimize_fingers', 'optimize_face', 'optimize_dynamics'):
        s = SimpleNamespace(**{**vars(body), k: True})
        assert pick(s, 4000) == 'sequential'
    assert pick(SimpleNamespace(**{**vars(body), 'n_shape': 8}), 4000) == 'sequential'
    for m in ('sequential', 'chunked', 'chunked_host'):
        assert pick(body, 10, m) == m and pick(body, 4000, m) == m
    assert default_cfg().moshpp_amd.chain_mode == 'auto'


def test_bench_takes_counter_numbers_only_from_the_build_it_runs(tmp_path):
    """bench.py --pmc-file: the committed PMC summary (profiles/r05_pmc.json, tools/r05_collect.sh) carries the source hash of the
    library it was collected on; a file of another build -- or none -- gives (None, why), never numbers.  The committed file should be
    the one of THIS tree's native sources (then the driver's bench line prints `traffic`, not null); if it is not, the test says so and skips."""
    import json
    import os
    import bench
    from moshpp_amd import build
    good = tmp_path / 'pmc.json'
    good.write_text(json.dumps({'source_hash': 'abc', 'collected': 'now', 'chain': {}, 'lbs': {}}))
    d, note = bench.load_pmc(str(good), 'abc')
    assert d is not None and d['source_hash'] == 'abc' and 'abc' in note
    d, note = bench.load_pmc(str(good), 'def')
    assert d is None and 'not used' in note
    d, note = bench.load_pmc(str(tmp_path / 'missing.json'), 'abc')
    assert d is None and 'no PMC file' in note
    committed = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'r05_pmc.json')
    d, note = bench.load_pmc(committed, build.source_hash())
    if d is None:   # (native sources edited since the last collection: stale evidence, not a broken product)
        import pytest
        pytest.skip(f'{note} -- re-run tools/r05_collect.sh on a GPU box and copy gpurun_out/r05/r05_pmc.json to profiles/')
    assert d['chain']['pass1_bytes_per_solved_frame'] > 0 and d['lbs']['mesh_order']['bytes_per_call_at_4000_frames'] > 0
