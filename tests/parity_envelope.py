"""TEST INFRASTRUCTURE (also read by bench.py's `parity` leg): the parity criterion for a whole Stage-II trajectory.

The north-star tolerance is 1e-4 rad on the pose and 1e-3 m marker RMSE against the reference solver "on the same c3d".  Two
float64 implementations of the same formulas agree to ~1e-9 rad frame after frame -- except where the reference's own algorithm
sits on a knife edge: a dogleg step accepted or rejected on the last bit, a max-mixture component switch (chmosh.py:665-705 through
chumpy's dogleg).  There ANY perturbation of the chain's state -- 1e-13 m on the observations is enough -- sends the next frames to
a different nearby solution of the same data for a stretch, after which the chains re-converge.  tests/golden/oracle_traj_*.npz
(make_oracle_trajectories.py: config 2's bench seeds; make_oracle_trajectories_configs.py: configs 3, 4, 5 at their stated lengths)
hold, for every frame of a sequence, the oracle's trajectory AND the spread of K perturbed oracle runs around it.  The criterion,
with no frame numbers in it (round 6: a ceiling where round 5 had a floor, a limit on how long a trajectory may stay away, the free
shape block and the simulated markers inside the check):

    spread[f] <= WELL = 3e-9 (the chain is well conditioned at f)   ->  |state - oracle|[f] <= TIGHT = 1e-7, same iteration count,
                                                                       simulated markers within MARKER_TIGHT = 1e-6 m RMSE of the oracle's
    spread[f] >  WELL  (the perturbed oracle runs themselves part)  ->  the trajectory may PART from the oracle here (the device is one
                       more perturbed run) and then follows another local solution of the same frames until the two re-converge
                       (deviation back under TIGHT).  While parted: |state - oracle| <= max(PARTED_FLOOR = 1e-3, FACTOR = 30 x the
                       stretch's largest spread), and the trajectory must be back within TAIL = 64 frames of the stretch's end
                       (measured re-convergence: 0.45 x per frame, i.e. 1e-1 -> 1e-7 in 18 frames);
    a deviation above TIGHT that BEGINS on a well-conditioned frame, one above the parted bound, and every frame of a trajectory that
    stays away longer than TAIL are outside the tolerance;
    the whole sequence: marker RMSE against the ORACLE's simulated markers <= 1e-3 m (north star), parted frames included.

`state` = pose variables, translation and -- where the solve has one -- the free shape block (expression / DMPL coefficients).
The spread is dilated by DILATE frames to both sides: a run that parts from the oracle does so a frame or two earlier or later
than the K sampled runs did.  Frames that violate the criterion are COUNTED (frames_outside_tolerance), never waved through."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TIGHT = 1e-7          # rad / m / coefficient: what two float64 implementations of the same formulas deliver on a well-conditioned frame
WELL = 3e-9           # spread of the perturbed oracle runs below which a frame counts as well conditioned (FACTOR x WELL < TIGHT)
FACTOR = 30.0         # a parted trajectory may deviate by this multiple of its stretch's largest spread ...
PARTED_FLOOR = 1e-3   # ... or by this much (three perturbed runs under-sample a knife edge: a stretch they barely open -- spread 1e-8 --
                      #     can send a fourth run further), whichever is larger.  Round 5 allowed max(0.2, 30 x spread) on every stretch.
TAIL = 64             # frames a trajectory may stay parted behind the end of the ill-conditioned stretch it parted on
DILATE = 12           # frames
POSE_TOL = 1e-4       # rad, north star (reported beside the criterion: frames_over_1e-4_rad)
MARKER_TOL = 1e-3     # m RMSE, north star, whole sequence against the oracle's simulated markers
MARKER_TIGHT = 1e-6   # m RMSE per frame on well-conditioned frames


def _fname(case):
    return os.path.join(GOLDEN, f'oracle_traj_seed{case}.npz' if isinstance(case, (int, np.integer)) or str(case).isdigit() else f'oracle_traj_{case}.npz')


def have(case):
    return os.path.exists(_fname(case))


def load(case):
    """case: a bench seed (int: config 2's 4000-frame SMPL-H sequence of that seed) or a named case (`config3_7000`, `mano_72`, ...)."""
    g = np.load(_fname(case))
    return {k: g[k] for k in g.files}


def dilated(spread, w=DILATE):
    """running maximum of `spread` over [f - w, f + w]"""
    n = len(spread)
    out = spread.astype(np.float64).copy()
    for s in range(1, w + 1):
        out[s:] = np.maximum(out[s:], spread[:n - s])
        out[:n - s] = np.maximum(out[:n - s], spread[s:])
    return out


def _stretch_max(d):
    """for every frame inside an ill-conditioned stretch (dilated spread > WELL): the stretch's largest spread; behind a stretch: the last
    stretch's value (a trajectory that parted there is still held to it while it re-converges); 0 ahead of the first."""
    out = np.zeros(len(d))
    ill = d > WELL
    f = 0
    while f < len(d):
        if ill[f]:
            e = f
            while e < len(d) and ill[e]:
                e += 1
            out[f:e] = d[f:e].max()
            f = e
        else:
            f += 1
    last = 0.0
    for f in range(len(d)):
        if out[f] > 0:
            last = out[f]
        else:
            out[f] = last
    return out


def _judge(dev, d, stretch_max):
    """Walks the frames in order.  Returns (outside[F] bool, parted[F] bool).  See the module text."""
    n = len(dev)
    outside = np.zeros(n, bool)
    parted = np.zeros(n, bool)
    away = False
    behind = 0          # consecutive parted frames on well-conditioned ground (= frames behind the stretch's end)
    for f in range(n):
        ill = d[f] > WELL
        if dev[f] <= TIGHT:
            away, behind = False, 0
            continue
        if ill:
            away, behind = True, 0
        elif away:
            behind += 1
        if away:
            parted[f] = True
            if dev[f] > max(PARTED_FLOOR, FACTOR * stretch_max[f]) or behind > TAIL:
                outside[f] = True
        else:
            outside[f] = True
    return outside, parted


def oracle_markers(m, closest, coef, pose, trans, shape=None):
    """The ORACLE's simulated markers at given states (oracle forward: transformed_lm.py:138-159 through stageii_oracle.verts_forward /
    markers_from_verts) -- what `markers_sim` of the oracle's chain holds for its own states; recomputed, so the fixtures need not store it."""
    from oracle import stageii_oracle as so
    flat = np.asarray(closest).reshape(-1)
    M = np.asarray(closest).shape[0]
    out = np.zeros((len(pose), M, 3))
    for i in range(len(pose)):
        v = so.verts_forward(m, so.fullpose_from_pose(m, pose[i]), trans[i], flat, shp=None if shape is None else shape[i]).reshape(M, 3, 3)
        out[i] = so.markers_from_verts(coef, v[:, 0], v[:, 1], v[:, 2])
    return out


def check(case, pose, trans, iters=None, frames=None, shape=None, markers_sim=None, vis=None, oracle_model=None):
    """`pose`[F', NP], `trans`[F', 3] (pose VARIABLES, as the oracle stores them) [, `shape`[F', E]] of the solved frames `frames` (default:
    all the oracle solved) against the committed oracle trajectory of `case`.  With `markers_sim`[F', M, 3], `vis`[F', M] and
    `oracle_model` = (m, closest, coef) the simulated markers are held to the oracle's (computed from the stored oracle states).
    Returns the report dict; nothing is asserted here -- `ok(report)` says whether the criterion holds."""
    g = load(case)
    fid = g['frame_ids']
    sel = np.arange(len(fid)) if frames is None else np.searchsorted(fid, np.asarray(frames))
    dev = np.maximum(np.abs(np.asarray(pose) - g['pose'][sel]).max(1), np.abs(np.asarray(trans) - g['trans'][sel]).max(1))
    if 'shape' in g and shape is not None:
        dev = np.maximum(dev, np.abs(np.asarray(shape) - g['shape'][sel]).max(1))
    q = float(g['quantum']) if 'quantum' in g else 0.0          # (the config fixtures store their states rounded to 2^-36)
    dev = np.maximum(dev - q, 0.0)
    d = dilated(g['spread'])
    well = d[sel] <= WELL
    out, parted = _judge(dev, d[sel], _stretch_max(d)[sel])
    ok_well = well & ~parted
    rep = {'frames': int(len(sel)), 'frames_outside_tolerance': int(out.sum()), 'frames_parted_on_a_knife_edge': int(parted.sum()),
           'well_conditioned_frames': int(well.sum()), 'max_dev_on_well_conditioned_frames_rad': float(dev[ok_well].max()) if ok_well.any() else 0.0,
           'ill_conditioned_frames': int((~well).sum()), 'max_dev_on_parted_frames_rad': float(dev[parted].max()) if parted.any() else 0.0,
           'max_oracle_spread_rad': float(g['spread'].max()), 'frames_over_1e-4_rad': int((dev > POSE_TOL).sum()),
           'max_abs_pose_diff_rad': float(dev.max()), 'first_frames_outside': [int(x) for x in fid[sel][out][:5]]}
    if iters is not None:
        same = np.asarray(iters).reshape(len(sel), -1)[:, 0] == g['iters'][sel]
        rep['iteration_counts_equal_on_well_conditioned_frames'] = bool(same[ok_well].all())
        rep['frames_with_other_iteration_count'] = int((~same).sum())
    if markers_sim is not None:
        m, closest, coef = oracle_model
        om = oracle_markers(m, closest, coef, g['pose'][sel], g['trans'][sel], g['shape'][sel] if 'shape' in g else None)
        v = np.asarray(vis, bool)
        sq = ((np.asarray(markers_sim) - om) ** 2).sum(-1) * v
        per = np.sqrt(sq.sum(1) / np.maximum(v.sum(1), 1))
        rep['marker_rmse_vs_oracle_m'] = float(np.sqrt(sq.sum() / max(v.sum(), 1)))
        rep['worst_frame_marker_rmse_vs_oracle_m'] = float(per.max())
        rep['worst_well_conditioned_frame_marker_rmse_vs_oracle_m'] = float(per[ok_well].max()) if ok_well.any() else 0.0
    return rep


def ok(rep, iters=True):
    """The criterion on a report of `check`."""
    good = rep['frames_outside_tolerance'] == 0 and rep['max_dev_on_well_conditioned_frames_rad'] <= TIGHT
    if iters and 'iteration_counts_equal_on_well_conditioned_frames' in rep:
        good = good and rep['iteration_counts_equal_on_well_conditioned_frames']
    if 'marker_rmse_vs_oracle_m' in rep:
        good = good and rep['marker_rmse_vs_oracle_m'] <= MARKER_TOL and rep['worst_well_conditioned_frame_marker_rmse_vs_oracle_m'] <= MARKER_TIGHT
    return bool(good)


def compare(case, a, b):
    """two DEVICE trajectories of the same sequence (e.g. chunked vs sequential): per-frame deviation against the same envelope.
    a, b: dicts with 'pose' [F, NP] and 'trans' [F, 3] over all frames; frames the oracle did not solve are skipped."""
    g = load(case)
    fid = g['frame_ids']
    dev = np.maximum(np.abs(a['pose'][fid] - b['pose'][fid]).max(1), np.abs(a['trans'][fid] - b['trans'][fid]).max(1))
    d = dilated(g['spread'])
    well = d <= WELL
    out, parted = _judge(dev, d, _stretch_max(d))
    return {'frames': int(len(fid)), 'frames_outside_tolerance': int(out.sum()), 'frames_parted_on_a_knife_edge': int(parted.sum()),
            'max_dev_on_well_conditioned_frames_rad': float(dev[well & ~parted].max()) if (well & ~parted).any() else 0.0,
            'max_dev_on_parted_frames_rad': float(dev[parted].max()) if parted.any() else 0.0,
            'ill_conditioned_frames': int((~well).sum()), 'frames_over_1e-4_rad': int((dev > POSE_TOL).sum()),
            'first_frames_outside': [int(x) for x in fid[out][:5]]}
