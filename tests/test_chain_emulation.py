"""The product's Stage-II kernels on the CPU: moshii_api.hip + chain_solve.hip compiled UNCHANGED by g++ against a stand-in for
<hip/hip_runtime.h> (tests/emu/fakehip) and executed one fiber per GPU thread with real barrier / wavefront-rendezvous semantics
(tests/emu/hip_emu_runtime.cpp).  Same parity checks as tests/test_gpu_parity.py, at sizes the emulation finishes in seconds.  This
checks the kernels' arithmetic and control flow; the GPU tests check the hipcc build on the device."""
import numpy as np
import pytest

from oracle import stageii_oracle as so
from tests.emu.emu_moshii import emulated_libmoshii
from tests.helpers import device_case, oracle_case


@pytest.mark.parametrize('model_type,fingers,F', [('smplh', False, 4), ('smpl', False, 3), ('smplh', True, 3), ('smplx', True, 3),
                                                  ('mano', True, 4)])
def test_chain_kernel_matches_oracle_in_emulation(model_type, fingers, F):
    M = {'smpl': 41, 'smplh': 53, 'smplx': 89, 'mano': 33}[model_type]
    case = oracle_case(model_type, F=F, M=M, seed=1, body_only_markers=not fingers)
    with emulated_libmoshii() as capi:
        dev = device_case(case, optimize_fingers=fingers)
        out = capi.chain_solve_host(dev['model'], dev['prior'], dev['opts'],
                                    [dict(attach=dev['attach'], obs=case['obs'], vis=case['vis'], first=True)])[0]
        kernel = capi.last_launch_info()[0]
    ref = so.stageii_chain(case['m'], case['prior'], case['closest'], case['coef'], case['obs'], case['vis'], model_type,
                           optimize_fingers=fingers)
    assert kernel.startswith('k_chain_solve<')
    solved = np.flatnonzero(out['status'] == 0)
    assert list(solved) == list(ref['frame_ids'])
    assert np.abs(out['fullpose'][solved] - ref['fullpose']).max() < 1e-9 and np.abs(out['trans'][solved] - ref['trans']).max() < 1e-10
    np.testing.assert_array_equal(out['iters'][solved, 0], ref['iters'])


def test_extended_kernel_matches_oracle_in_emulation():
    """The xt variant (jaw + expression coefficients free, chmosh.py:562-567, 685-699)."""
    from tests.helpers import shape_case
    case = shape_case('smplx', F=3, M=40, E=6, seed=3, kind='expr')
    with emulated_libmoshii() as capi:
        dev = device_case(case, optimize_face=True, shape_kind='expr')
        out = capi.chain_solve_host(dev['model'], dev['prior'], dev['opts'],
                                    [dict(attach=dev['attach'], obs=case['obs'], vis=case['vis'], first=True)])[0]
        assert capi.last_launch_info()[0].endswith(',xt>')
    ref = so.stageii_chain(case['m'], case['prior'], case['closest'], case['coef'], case['obs'], case['vis'], 'smplx',
                           optimize_face=True, free_shape='expr')
    assert np.abs(out['fullpose'] - ref['fullpose']).max() < 1e-8 and np.abs(out['shape'] - ref['shape']).max() < 1e-8
    np.testing.assert_array_equal(out['iters'][:, 0], ref['iters'])


def test_chunked_sequence_solve_equals_sequential_chain_in_emulation():
    """moshii_sequence_solve (chunks, on-device hand-off verification, repairs) against moshii_chain_solve on the same sequence."""
    case = oracle_case('smplh', F=32, M=53, seed=2)
    with emulated_libmoshii() as capi:
        dev = device_case(case)
        seq = capi.chain_solve_host(dev['model'], dev['prior'], dev['opts'],
                                    [dict(attach=dev['attach'], obs=case['obs'], vis=case['vis'], first=True)])[0]
        outs, report = capi.sequence_solve_host(dev['model'], dev['prior'], dev['opts'],
                                                [dict(attach=dev['attach'], obs=case['obs'], vis=case['vis'])],
                                                num_chunks=4, warmup=6, verify_tol=1e-9)
    assert report['n_chunks'] == 4
    assert np.abs(outs[0]['fullpose'] - seq['fullpose']).max() < 1e-7 and np.abs(outs[0]['trans'] - seq['trans']).max() < 1e-8


@pytest.mark.parametrize('kind,model_type', [('expr', 'smplx'), ('dmpl', 'smplh')])
def test_chunked_sequence_solve_carries_the_free_shape_block_in_emulation(kind, model_type):
    """moshii_sequence_solve with n_shape > 0: the expression / DMPL coefficients travel in the chunk hand-off states (verified with
    pose and trans, re-solved from the predecessor's end state, rejoin test on the shape rows too) -- equal to the sequential chain."""
    from tests.helpers import shape_case
    F = 24
    case = shape_case(model_type, F=F, M=40, E=4, seed=9, kind=kind)
    with emulated_libmoshii() as capi:
        dev = device_case(case, optimize_face=(kind == 'expr'), shape_kind=kind)
        seq = capi.chain_solve_host(dev['model'], dev['prior'], dev['opts'],
                                    [dict(attach=dev['attach'], obs=case['obs'], vis=case['vis'], first=True)])[0]
        outs, rep = capi.sequence_solve_host(dev['model'], dev['prior'], dev['opts'],
                                             [dict(attach=dev['attach'], obs=case['obs'], vis=case['vis'])],
                                             num_chunks=3, warmup=6, verify_tol=1e-9)
    assert rep['n_chunks'] == 3 and np.abs(seq['shape']).max() > 0.2
    assert np.abs(outs[0]['fullpose'] - seq['fullpose']).max() < 1e-7 and np.abs(outs[0]['shape'] - seq['shape']).max() < 1e-7
    assert np.array_equal(outs[0]['status'], seq['status'])


def test_empty_frames_and_chain_continuation_in_emulation():
    """Frames without any visible marker are skipped (status 1, chmosh.py:586-588) and the chain carries on; a chain continued from an
    earlier chain's state (init_pose / init_trans / init_pose_prev, no first-frame schedule) reproduces the uninterrupted chain."""
    case = oracle_case('smplh', F=12, M=53, seed=4, empty_frames=(5,))
    with emulated_libmoshii() as capi:
        dev = device_case(case)
        args = (dev['model'], dev['prior'], dev['opts'])
        full = capi.chain_solve_host(*args, [dict(attach=dev['attach'], obs=case['obs'], vis=case['vis'], first=True)])[0]
        head = capi.chain_solve_host(*args, [dict(attach=dev['attach'], obs=case['obs'][:8], vis=case['vis'][:8], first=True)])[0]
        tail = capi.chain_solve_host(*args, [dict(attach=dev['attach'], obs=case['obs'][8:], vis=case['vis'][8:], first=False,
                                                  init_pose=head['pose'][7], init_trans=head['trans'][7], init_pose_prev=head['pose'][6])])[0]
    ref = so.stageii_chain(case['m'], case['prior'], case['closest'], case['coef'], case['obs'], case['vis'], 'smplh')
    assert full['status'].tolist() == [0] * 5 + [1] + [0] * 6 and list(ref['frame_ids']) == [t for t in range(12) if t != 5]
    solved = np.flatnonzero(full['status'] == 0)
    assert np.abs(full['fullpose'][solved] - ref['fullpose']).max() < 1e-9
    assert np.abs(np.vstack([head['fullpose'], tail['fullpose']])[solved] - full['fullpose'][solved]).max() < 1e-12


def test_bench_call_path_in_emulation():
    """bench.py's device-resident path (workload.DeviceSequence: moshii_sequence_solve / moshii_chain_solve with MOSHII_BUFFERS_DEVICE,
    outputs in caller-owned buffers; workload.solve_many_chunked with several sequence descriptors) run against the emulated library with
    CPU tensors standing in for HBM: chunked == sequential, struct layouts of the ctypes descriptors match the header."""
    from moshpp_amd import workload
    with emulated_libmoshii():
        job = workload.make_job('smplh', n_frames=20, n_markers=53, seed=3)
        solver = workload.make_solver(job)
        ds = workload.DeviceSequence(job, solver, 'cpu')
        rep = ds.solve_chunked(None, num_chunks=3, warmup=5, verify_tol=1e-9)
        chunked = ds.results()
        ds.solve_sequential(None)
        seq = ds.results()
        two = [workload.DeviceSequence(job, solver, 'cpu') for _ in range(2)]
        rep2 = workload.solve_many_chunked(two, None, num_chunks=3, warmup=5, verify_tol=1e-9)
        many = [d.results() for d in two]
    assert rep['n_chunks'] == 3 and rep2['n_chunks'] == 6
    assert np.abs(chunked['fullpose'] - seq['fullpose']).max() < 1e-7 and np.array_equal(chunked['status'], seq['status'])
    for r in many:
        assert np.abs(r['fullpose'] - seq['fullpose']).max() < 1e-7


def test_sequence_solve_continues_a_chain_in_emulation():
    """moshii_sequence_desc.init_*: a chunked solve that starts from another chain's end state (how one sequence is spread over ranks)."""
    case = oracle_case('smplh', F=30, M=53, seed=6)
    with emulated_libmoshii() as capi:
        dev = device_case(case)
        args = (dev['model'], dev['prior'], dev['opts'])
        full = capi.chain_solve_host(*args, [dict(attach=dev['attach'], obs=case['obs'], vis=case['vis'], first=True)])[0]
        outs, rep = capi.sequence_solve_host(*args, [dict(attach=dev['attach'], obs=case['obs'][12:], vis=case['vis'][12:],
                                                         init_pose=full['pose'][11], init_trans=full['trans'][11],
                                                         init_pose_prev=full['pose'][10])], num_chunks=3, warmup=5, verify_tol=1e-9)
    assert rep['n_chunks'] == 3
    assert np.abs(outs[0]['fullpose'] - full['fullpose'][12:]).max() < 1e-7


def test_underdetermined_frames_are_flagged_and_still_fit_their_data_in_emulation():
    """A frame whose normal matrix is singular (MANO has no pose prior; with ONE visible marker and no velocity term yet, 3 data rows face
    6 root / translation unknowns): the solution is not unique.  chumpy hands the singular system to a dense solver and takes whatever
    comes back (the oracle restates that with LAPACK: solve, lstsq on LinAlgError); the kernel's LDL^T stops at the non-positive pivot,
    takes the Cauchy step instead and reports status -1 for the frame.  Both drive the data term to zero -- at different points of the
    solution set (here 1.2 rad apart), which is all the reference's algorithm defines -- and the chain re-converges once the frames are
    determined again (weak memory through the velocity term).  This is the one place where the kernel knowingly departs from a literal
    restatement; DESIGN.md section 3."""
    case = oracle_case('mano', F=4, M=33, seed=9)
    vis = case['vis'].copy()
    vis[:2] = 0
    vis[:2, 5] = 1
    with emulated_libmoshii() as capi:
        dev = device_case(case, optimize_fingers=True)
        out = capi.chain_solve_host(dev['model'], dev['prior'], dev['opts'],
                                    [dict(attach=dev['attach'], obs=case['obs'], vis=vis, first=True)])[0]
    ref = so.stageii_chain(case['m'], case['prior'], case['closest'], case['coef'], case['obs'], vis, 'mano', optimize_fingers=True)
    assert out['status'].tolist() == [-1, -1, 0, 0]
    assert out['errs'][:2, 0].max() < 1e-12 and np.asarray(ref['errs']['data'])[:2].max() < 1e-12          # both fit the lone marker
    assert np.abs(out['fullpose'][:2] - ref['fullpose'][:2]).max() > 0.1                                     # ... at different poses
    assert np.abs(out['fullpose'][2:] - ref['fullpose'][2:]).max() < 1e-2                                    # and meet again afterwards
    assert np.allclose(out['errs'][2:, 0], np.asarray(ref['errs']['data'])[2:], rtol=1e-2)


@pytest.mark.parametrize('model_type,F,order,still', [('mano', 20, 'shuffled', None), ('smpl', 17, 'shuffled', None), ('mano', 140, 'mesh', None),
                                                      ('smpl', 19, 'shuffled', 30), ('mano', 21, 'mesh', 3), ('smpl', 18, 'mesh', 0)])
def test_lbs_export_kernel_matches_f64_in_emulation(model_type, F, order, still):
    """The f16-MFMA full-mesh export (lbs_forward.hip: k_lbs_prep + k_lbs_export, compiled unchanged by the host clang++) against the
    f64 kernel of the same emulated library: 16x16x32 MFMA fragment layouts, the feature ring, the vertex groups and their
    joint lists (several blend rounds per group on the shuffled bodies, mostly one on the mesh-ordered one), the blend on the f32
    matrix instruction with its operands straight from the transform array, the result exchange's un-permutation, partial vertex /
    frame tiles, and a workgroup's second tile (F = 140: two frame tiles).  `still`: the pose variables from that index on are the
    same in every frame -- the joints behind them do not move, k_lbs_prep leaves them unmarked and the export evaluates their k-steps
    for one frame block (still = 30 on SMPL: 3 of 7 k-steps stay; 3 on MANO: every finger still, no k-step stays; 0: a still body)."""
    from moshpp_amd import synth
    M = {'mano': 24, 'smpl': 41}[model_type]
    case = oracle_case(model_type, F=4, M=M, seed=61, dd=synth.synth_model(model_type, seed=61, vertex_order=order))
    rng = np.random.default_rng(5)
    pose = rng.normal(0, 0.35, (F, case['m']['NP']))
    trans = rng.normal(0, 1, (F, 3))
    if still is not None:
        pose[:, still:] = pose[0, still:]
    with emulated_libmoshii():
        dev = device_case(case)
        ref = dev['model'].lbs_forward(pose, trans)
        got = dev['model'].lbs_forward(pose, trans, dtype=np.float32)
        got2 = dev['model'].lbs_forward(pose, trans, dtype=np.float32)
        if still is not None:      # ... and a following call in which everything moves again (the marks carry the call's number)
            pose2 = rng.normal(0, 0.35, pose.shape)
            ref3 = dev['model'].lbs_forward(pose2, trans)
            got3 = dev['model'].lbs_forward(pose2, trans, dtype=np.float32)
            assert np.abs(got3 - ref3).max() < 2e-5
    assert np.abs(got - ref).max() < 2e-5
    np.testing.assert_array_equal(got, got2)


def test_the_librarys_own_choice_of_a_cooperative_group_in_emulation(monkeypatch):
    """flags without a MOSHII_COOP_GROUP word and MOSHII_COOP=auto: the LIBRARY picks the group (prepare_launch, coop_g == -1: one rank
    per 256 (marker, joint) Jacobian items + one for the prior, if every workgroup can be resident).  On a device that is the default of
    every drop-in call; the emulation build answers "plain chains" unless the environment spells the choice out, so this is the only
    CPU coverage of the selection path.  SMPL-H / 53 markers on the emulated 8-CU device: six ranks, same iteration counts as the
    plain chain; MANO: the library stays with plain chains (the exchanges cost what the split saves)."""
    monkeypatch.setenv('HIPEMU_CONCURRENT', '1')
    monkeypatch.setenv('MOSHII_COOP', 'auto')
    case = oracle_case('smplh', F=3, M=53, seed=4, body_only_markers=True)
    with emulated_libmoshii() as capi:
        dev = device_case(case)
        ch = [dict(attach=dev['attach'], obs=case['obs'], vis=case['vis'], first=True)]
        out = capi.chain_solve_host(dev['model'], dev['prior'], dev['opts'], ch)[0]
        kernel = capi.last_launch_info()[0]
        monkeypatch.setenv('MOSHII_COOP', '1')
        plain = capi.chain_solve_host(dev['model'], dev['prior'], dev['opts'], ch)[0]
        assert capi.last_launch_info()[0] == 'k_chain_solve<4,1>'
    assert kernel == 'k_chain_solve<4,1,coop6>', kernel
    np.testing.assert_array_equal(out['iters'], plain['iters'])
    assert np.abs(out['fullpose'] - plain['fullpose']).max() < 1e-9
    monkeypatch.setenv('MOSHII_COOP', 'auto')
    case = oracle_case('mano', F=2, M=33, seed=4)
    with emulated_libmoshii() as capi:
        dev = device_case(case, optimize_fingers=True)
        capi.chain_solve_host(dev['model'], dev['prior'], dev['opts'], [dict(attach=dev['attach'], obs=case['obs'], vis=case['vis'], first=True)])
        assert ',coop' not in capi.last_launch_info()[0], capi.last_launch_info()


@pytest.mark.parametrize('model_type,fingers,G,F', [('smplh', False, 5, 5), ('smplh', True, 4, 3), ('mano', True, 3, 4), ('smpl', False, 2, 3)])
def test_cooperative_chain_matches_oracle_in_emulation(monkeypatch, model_type, fingers, G, F):
    """One chain solved by G workgroups (MOSHII_COOP_GROUP; chain_solve.hip, COOP variant): the ranks split markers / vertices / Jacobian
    rows and the prior, meet in two exchanges per dogleg iteration, and every rank takes the same decisions.  The emulation runs the G
    workgroups on G OS threads (HIPEMU_CONCURRENT=1) with real atomics.  Held to the oracle like the plain chain (identical iteration
    counts), with an empty frame and occluded markers in the sequence."""
    monkeypatch.setenv('HIPEMU_CONCURRENT', '1')
    M = {'smpl': 41, 'smplh': 53, 'smplx': 89, 'mano': 33}[model_type]
    case = oracle_case(model_type, F=F, M=M, seed=4, body_only_markers=not fingers, empty_frames=(2,))
    with emulated_libmoshii() as capi:
        dev = device_case(case, optimize_fingers=fingers)
        ch = [dict(attach=dev['attach'], obs=case['obs'], vis=case['vis'], first=True)]
        plain = capi.chain_solve_host(dev['model'], dev['prior'], dev['opts'], ch)[0]
        out = capi.chain_solve_host(dev['model'], dev['prior'], dev['opts'], ch, coop=G)[0]
        kernel = capi.last_launch_info()[0]
        # the ranks' arrival order at every exchange randomised (MOSHII_COOP_SKEW, chain_solve.hip: coop_skew): not a bit may move
        monkeypatch.setenv('MOSHII_COOP_SKEW', '5')
        skewed = capi.chain_solve_host(dev['model'], dev['prior'], dev['opts'], ch, coop=G)[0]
        monkeypatch.delenv('MOSHII_COOP_SKEW')
    assert kernel.endswith(f',coop{G}>'), kernel
    for k in ('pose', 'fullpose', 'trans', 'markers_sim', 'errs', 'iters', 'status'):
        np.testing.assert_array_equal(skewed[k], out[k], err_msg=k)
    ref = so.stageii_chain(case['m'], case['prior'], case['closest'], case['coef'], case['obs'], case['vis'], model_type,
                           optimize_fingers=fingers)
    solved = np.flatnonzero(out['status'] == 0)
    assert list(solved) == list(ref['frame_ids']) and out['status'][2] == 1
    assert np.abs(out['fullpose'][solved] - ref['fullpose']).max() < 1e-9 and np.abs(out['trans'][solved] - ref['trans']).max() < 1e-10
    np.testing.assert_array_equal(out['iters'][solved, 0], ref['iters'])
    # against the plain chain: every output row (the ranks write their own simulated markers; rank 0 the rest)
    np.testing.assert_array_equal(out['iters'], plain['iters'])
    for k in ('pose', 'fullpose', 'trans', 'markers_sim'):
        assert np.abs(out[k] - plain[k]).max() < 1e-9, k
    assert np.abs(out['errs'] - plain['errs']).max() < 1e-7 * max(1.0, np.abs(plain['errs']).max())


def test_cooperative_chains_several_groups_in_one_launch_in_emulation(monkeypatch):
    """Two chains x 3 workgroups in one launch (block -> (chain, rank) mapping, separate exchange buffers), one of them continuing a chain."""
    monkeypatch.setenv('HIPEMU_CONCURRENT', '1')
    case = oracle_case('smplh', F=6, M=53, seed=6)
    with emulated_libmoshii() as capi:
        dev = device_case(case)
        first = dict(attach=dev['attach'], obs=case['obs'][:3], vis=case['vis'][:3], first=True)
        a = capi.chain_solve_host(dev['model'], dev['prior'], dev['opts'], [first])[0]
        cont = dict(attach=dev['attach'], obs=case['obs'][3:], vis=case['vis'][3:], first=False, init_pose=a['pose'][2], init_trans=a['trans'][2],
                    init_pose_prev=a['pose'][1])
        plain = capi.chain_solve_host(dev['model'], dev['prior'], dev['opts'], [first, cont])
        outs = capi.chain_solve_host(dev['model'], dev['prior'], dev['opts'], [first, cont], coop=3)
        assert capi.last_launch_info()[0].endswith(',coop3>')
    for o, p_ in zip(outs, plain):
        np.testing.assert_array_equal(o['iters'], p_['iters'])
        assert np.abs(o['fullpose'] - p_['fullpose']).max() < 1e-9 and np.abs(o['markers_sim'] - p_['markers_sim']).max() < 1e-9


def test_chunked_solve_with_cooperative_repair_chains_in_emulation(monkeypatch):
    """moshii_sequence_solve with MOSHII_COOP_GROUP(g): no carry-on inside the first launch; the host's repair rounds launch every
    repair chain as g workgroups -- rank 0 takes the decisions that depend on other chains' memory (boundary negotiation, stop
    requests, re-joining the stored rows) and the other ranks follow its word.  Short warm-up + tight tolerance: most hand-offs miss,
    sweeps run through several chunks, meet other chains' territories and re-join.  Result: the sequential chain's."""
    monkeypatch.setenv('HIPEMU_CONCURRENT', '1')
    case = oracle_case('smplh', F=40, M=53, seed=52, empty_frames=(9, 10, 20))
    with emulated_libmoshii() as capi:
        dev = device_case(case)
        seq = capi.chain_solve_host(dev['model'], dev['prior'], dev['opts'],
                                    [dict(attach=dev['attach'], obs=case['obs'], vis=case['vis'], first=True)])[0]
        outs, rep = capi.sequence_solve_host(dev['model'], dev['prior'], dev['opts'],
                                             [dict(attach=dev['attach'], obs=case['obs'], vis=case['vis'])],
                                             num_chunks=5, warmup=2, verify_tol=1e-12, coop=3)
        kernel = capi.last_launch_info()[0]
    print(rep, kernel)
    assert rep['n_chunks'] == 5 and rep['n_repaired'] >= 1 and rep['repair_rounds'] >= 1
    assert ',coop' in kernel
    np.testing.assert_array_equal(outs[0]['status'], seq['status'])
    assert np.abs(outs[0]['fullpose'] - seq['fullpose']).max() < 1e-9 and np.abs(outs[0]['markers_sim'] - seq['markers_sim']).max() < 1e-9
    np.testing.assert_array_equal(outs[0]['iters'], seq['iters'])


@pytest.mark.parametrize('G', [3, 5])
def test_cooperative_extended_kernel_matches_oracle_in_emulation(monkeypatch, G):
    """The xt variant (jaw + expression coefficients free) as a cooperative chain: every rank re-shapes its own rest vertices, keeps its
    own shape-derivative arrays, builds the shape columns of its own markers' rows; the regulariser columns are every rank's alike."""
    from tests.helpers import shape_case
    monkeypatch.setenv('HIPEMU_CONCURRENT', '1')
    case = shape_case('smplx', F=3, M=40, E=6, seed=3, kind='expr')
    with emulated_libmoshii() as capi:
        dev = device_case(case, optimize_face=True, shape_kind='expr')
        ch = [dict(attach=dev['attach'], obs=case['obs'], vis=case['vis'], first=True)]
        plain = capi.chain_solve_host(dev['model'], dev['prior'], dev['opts'], ch)[0]
        out = capi.chain_solve_host(dev['model'], dev['prior'], dev['opts'], ch, coop=G)[0]
        assert capi.last_launch_info()[0].endswith(f',xt,coop{G}>'), capi.last_launch_info()
    ref = so.stageii_chain(case['m'], case['prior'], case['closest'], case['coef'], case['obs'], case['vis'], 'smplx',
                           optimize_face=True, free_shape='expr')
    assert np.abs(out['fullpose'] - ref['fullpose']).max() < 1e-8 and np.abs(out['shape'] - ref['shape']).max() < 1e-8
    np.testing.assert_array_equal(out['iters'][:, 0], ref['iters'])
    np.testing.assert_array_equal(out['iters'], plain['iters'])
    assert np.abs(out['shape'] - plain['shape']).max() < 1e-9 and np.abs(out['markers_sim'] - plain['markers_sim']).max() < 1e-9


@pytest.mark.parametrize('E,G', [(20, 0), (20, 3)])
def test_extended_kernel_with_more_than_127_unknowns_in_emulation(monkeypatch, E, G):
    """3 + 111 + E unknowns (fingers, jaw and E expression coefficients free): beyond eight register blocks the factor lives in global
    memory and the solve is ldl_big -- 16-column panels on wavefront 0, the left-looking products on the matrix pipe (v_mfma_f64_16x16x4)
    by the other wavefronts, a staged back-substitution; as a cooperative chain the assembly's exchange is the reduce-scatter +
    all-gather form (14 tiles per wavefront at ten blocks).  Same iteration counts as the oracle, the first frame from a cold start."""
    from tests.helpers import shape_case
    if G:
        monkeypatch.setenv('HIPEMU_CONCURRENT', '1')
    case = shape_case('smplx', F=2, M=60, E=E, seed=5, kind='expr')
    with emulated_libmoshii() as capi:
        dev = device_case(case, optimize_fingers=True, optimize_face=True, shape_kind='expr')
        out = capi.chain_solve_host(dev['model'], dev['prior'], dev['opts'],
                                    [dict(attach=dev['attach'], obs=case['obs'], vis=case['vis'], first=True)], coop=G or 1)[0]
        name = capi.last_launch_info()[0]
        assert name == ('k_chain_solve<10,1,xt,coop3>' if G else 'k_chain_solve<10,1,xt>'), name
    ref = so.stageii_chain(case['m'], case['prior'], case['closest'], case['coef'], case['obs'], case['vis'], 'smplx',
                           optimize_fingers=True, optimize_face=True, free_shape='expr')
    assert np.abs(out['fullpose'] - ref['fullpose']).max() < 2e-8 and np.abs(out['shape'] - ref['shape']).max() < 2e-8
    np.testing.assert_array_equal(out['iters'][:, 0], ref['iters'])
    assert np.all(out['status'] == 0)

