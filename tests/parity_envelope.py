"""TEST INFRASTRUCTURE (also read by bench.py's `parity` leg): the parity criterion for a whole Stage-II trajectory.

The north-star tolerance is 1e-4 rad on the pose and 1e-3 m marker RMSE against the reference solver "on the same c3d".  Two
float64 implementations of the same formulas agree to ~1e-9 rad frame after frame -- except where the reference's own algorithm
sits on a knife edge: a dogleg step accepted or rejected on the last bit, a max-mixture component switch (chmosh.py:665-705 through
chumpy's dogleg).  There ANY perturbation of the chain's state -- 1e-13 m on the observations is enough -- sends the next frames to
a different nearby solution of the same data for a stretch, after which the chains re-converge.  tests/golden/oracle_traj_seed*.npz
(make_oracle_trajectories.py) hold, for every frame of a sequence, the oracle's trajectory AND the spread of K perturbed oracle
runs around it.  The criterion, with no frame numbers in it:

    spread[f] <= WELL = 3e-9 (the chain is well conditioned at f)      ->  |pose - oracle|[f] <= TIGHT = 1e-7 rad, same iteration count
    spread[f] >  WELL  (the perturbed oracle runs themselves part)  ->  the trajectory may PART from the oracle here (the device is one
                       more perturbed run) and then follows another local solution of the same frames until the two re-converge
                       (deviation back under TIGHT); while parted: |pose - oracle| <= max(PARTED_MAX, FACTOR x the stretch's spread),
    a deviation above TIGHT that BEGINS on a well-conditioned frame is outside the tolerance, and in any case
    every frame: the marker RMSE of the device's fit stays within the north-star 1e-3 m of the other trajectory's (checked by the callers).

The spread is dilated by DILATE frames to both sides: a run that parts from the oracle does so a frame or two earlier or later
than the K sampled runs did.  Frames that violate the criterion are COUNTED (frames_outside_tolerance), never waved through."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TIGHT = 1e-7        # rad: what two float64 implementations of the same formulas deliver on a well-conditioned frame
WELL = 3e-9         # rad: spread of the perturbed oracle runs below which a frame counts as well conditioned (FACTOR x WELL < TIGHT)
FACTOR = 30.0       # an ill-conditioned frame may deviate by this multiple of the (dilated) spread of the perturbed oracle runs
DILATE = 12         # frames
PARTED_MAX = 0.2    # rad: ceiling of a deviation on a parted stretch (another local solution of the same frames), whatever the spread
POSE_TOL = 1e-4     # rad, north star (reported beside the criterion: frames_over_1e-4_rad)


def have(seed):
    return os.path.exists(os.path.join(GOLDEN, f'oracle_traj_seed{seed}.npz'))


def load(seed):
    g = np.load(os.path.join(GOLDEN, f'oracle_traj_seed{seed}.npz'))
    return {k: g[k] for k in g.files}


def dilated(spread, w=DILATE):
    """running maximum of `spread` over [f - w, f + w]"""
    n = len(spread)
    out = spread.astype(np.float64).copy()
    for s in range(1, w + 1):
        out[s:] = np.maximum(out[s:], spread[:n - s])
        out[:n - s] = np.maximum(out[:n - s], spread[s:])
    return out


def _judge(dev, spread_sel_dilated, stretch_max):
    """Walks the frames in order.  Returns (outside[F] bool, parted[F] bool).
    A trajectory may PART from the oracle only on a frame the envelope marks ill conditioned; from there on it follows another local
    solution of the same data until the two re-converge (deviation back under TIGHT) -- possibly long after the perturbed oracle runs,
    which took the oracle's branch, have: those frames are `parted`, held to PARTED_MAX and (by the callers) to the marker bound.
    A deviation above TIGHT that begins on a well-conditioned frame is `outside`."""
    n = len(dev)
    outside = np.zeros(n, bool)
    parted = np.zeros(n, bool)
    away = False
    for f in range(n):
        ill = spread_sel_dilated[f] > WELL
        if dev[f] <= TIGHT:
            away = False
            continue
        if ill:
            away = True
        if away:
            parted[f] = True
            if dev[f] > max(PARTED_MAX, FACTOR * stretch_max[f]):
                outside[f] = True
        else:
            outside[f] = True
    return outside, parted


def _stretch_max(d):
    """for every frame inside an ill-conditioned stretch (dilated spread > WELL): the stretch's largest spread; 0 elsewhere"""
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
    # frames behind a stretch inherit its bound while a trajectory is still `parted` (see _judge): carry the last value forward
    last = 0.0
    for f in range(len(d)):
        if out[f] > 0:
            last = out[f]
        else:
            out[f] = last
    return out


def check(seed, pose, trans, iters=None, frames=None, envelope_seed=None):
    """`pose`[F', NP], `trans`[F', 3] (pose VARIABLES, as the oracle stores them) of the solved frames `frames` (default: all the
    oracle solved) against the committed oracle trajectory of `seed`.  Returns the report dict; nothing is asserted here."""
    g = load(seed if envelope_seed is None else envelope_seed)
    fid = g['frame_ids']
    sel = np.arange(len(fid)) if frames is None else np.searchsorted(fid, np.asarray(frames))
    dev = np.maximum(np.abs(np.asarray(pose) - g['pose'][sel]).max(1), np.abs(np.asarray(trans) - g['trans'][sel]).max(1))
    d = dilated(g['spread'])
    well = d[sel] <= WELL
    out, parted = _judge(dev, d[sel], _stretch_max(d)[sel])
    rep = {'frames': int(len(sel)), 'frames_outside_tolerance': int(out.sum()), 'frames_parted_on_a_knife_edge': int(parted.sum()),
           'well_conditioned_frames': int(well.sum()), 'max_dev_on_well_conditioned_frames_rad': float(dev[well & ~parted].max()) if (well & ~parted).any() else 0.0,
           'ill_conditioned_frames': int((~well).sum()), 'max_dev_on_parted_frames_rad': float(dev[parted].max()) if parted.any() else 0.0,
           'max_oracle_spread_rad': float(g['spread'].max()), 'frames_over_1e-4_rad': int((dev > POSE_TOL).sum()),
           'max_abs_pose_diff_rad': float(dev.max()),
           'first_frames_outside': [int(x) for x in fid[sel][out][:5]]}
    if iters is not None:
        same = np.asarray(iters).reshape(len(sel), -1)[:, 0] == g['iters'][sel]
        ok = well & ~parted
        rep['iteration_counts_equal_on_well_conditioned_frames'] = bool(same[ok].all())
        rep['frames_with_other_iteration_count'] = int((~same).sum())
    return rep


def compare(seed, a, b):
    """two DEVICE trajectories of the same sequence (e.g. chunked vs sequential): per-frame deviation against the same envelope.
    a, b: dicts with 'pose' [F, NP] and 'trans' [F, 3] over all frames; frames the oracle did not solve are skipped."""
    g = load(seed)
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
