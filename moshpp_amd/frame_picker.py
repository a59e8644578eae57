"""Stage-I frame pickers: src/moshpp/frame_picker.py:43-213 with the same NumPy legacy-RNG call sequence, so that a seed picks the
same frames as the reference does.  Each returns (frames, names): an object array of per-frame `label -> xyz` dicts
(`MocapSession.markers_asdict()` entries) and the `<file>_<frame:06d>` keys."""
from __future__ import annotations

import os.path as osp

import numpy as np

from .mocap_interface import MocapSession


def _obj_array(items):
    a = np.empty(len(items), dtype=object)
    for i, it in enumerate(items):
        a[i] = it
    return a


def load_marker_sessions_manual(mocap_fnames, mocap_unit, mocap_rotate=None, only_subjects=None, only_markers=None,
                                exclude_markers=None, labels_map={}):
    """names are /path/to/mocap_<frame number>.<ext> split at the last underscore (:48-66)."""
    frames, names = [], []
    for frame in mocap_fnames:
        parts = frame.split('_')
        fname, fid = '_'.join(parts[:-1]), int(parts[-1])
        assert osp.exists(fname), FileNotFoundError(fname)
        names.append(f'{fname}_{fid:06d}')
        frames.append(MocapSession(mocap_fname=fname, mocap_unit=mocap_unit, only_subjects=only_subjects, mocap_rotate=mocap_rotate,
                                   only_markers=only_markers, exclude_markers=exclude_markers,
                                   labels_map=labels_map).markers_asdict()[fid])
    return _obj_array(frames), np.array(names)


def load_marker_sessions_random(mocap_fnames, mocap_unit, mocap_rotate=None, num_frames=12, only_subjects=None, seed=None,
                                least_avail_markers=.1, only_markers=None, exclude_markers=None, labels_map={}):
    """:71-146.  Per file `num_frames` frames drawn with replacement BEFORE the seed is applied (the seed only fixes the
    shuffle), keys numbered by draw order; frames with too few usable markers are skipped; the threshold is lowered by 0.01 and
    the whole procedure repeated (without `exclude_markers`, as the reference's recursive call does) until enough are found."""
    pool = {}
    for fname in mocap_fnames:
        mocap = MocapSession(mocap_fname=fname, mocap_unit=mocap_unit, mocap_rotate=mocap_rotate, only_subjects=only_subjects,
                             only_markers=only_markers, exclude_markers=exclude_markers, labels_map=labels_map)
        fmd = mocap.markers_asdict()
        fmd = [fmd[i] for i in np.random.choice(len(mocap), num_frames)]
        for fidx in range(len(fmd)):
            pool[f'{fname}_{fidx:06d}'] = fmd[fidx]
        if len(pool) > 100:
            break
    idxs = list(range(len(pool)))
    if seed is not None:
        np.random.seed(seed=seed)
    np.random.shuffle(idxs)
    all_frames, all_names = list(pool.values()), list(pool.keys())
    frames, names = [], []
    for idx in idxs:
        frame = all_frames[idx]
        nonans = [k for k in frame.keys() if ~np.any(np.isnan(frame[k])) and ('*' not in k)]
        if len(nonans) >= (least_avail_markers * len(frame)):
            names.append(all_names[idx]); frames.append(frame)
        if len(frames) >= num_frames:
            break
    if len(frames) < num_frames:
        least_avail_markers = least_avail_markers - 0.01
        if least_avail_markers < 0.01:
            raise ValueError(f'Not enough frames were found that have at least %{least_avail_markers * 100.:.1f} of the markers.\n')
        return load_marker_sessions_random(mocap_fnames, mocap_unit=mocap_unit, mocap_rotate=mocap_rotate, seed=seed,
                                           num_frames=num_frames, only_subjects=only_subjects,
                                           least_avail_markers=least_avail_markers, only_markers=only_markers, labels_map=labels_map)
    return _obj_array(frames), np.array(names)


def load_marker_sessions_random_strict(mocap_fnames, mocap_unit, mocap_rotate=None, num_frames=12, only_subjects=None, seed=None,
                                       least_avail_markers=.1, only_markers=None, exclude_markers=None, labels_map={}):
    """:149-213.  Seeded first; per file a random permutation of the frames, taking up to `num_frames` whose share of valid
    markers reaches the threshold; then `num_frames` of the pool without replacement.  Raises when the pool is too small."""
    np.random.seed(seed=seed)
    assert 0.1 <= least_avail_markers <= 1.0
    pool = {}
    for fname in mocap_fnames:
        mocap = MocapSession(mocap_fname=fname, mocap_unit=mocap_unit, mocap_rotate=mocap_rotate, only_markers=only_markers,
                             only_subjects=only_subjects, exclude_markers=exclude_markers, labels_map=labels_map)
        if not mocap.read_status:
            continue
        avail = MocapSession.marker_availability_mask(mocap.markers)
        avail = avail.sum(-1) / avail.shape[1]
        frames = mocap.markers_asdict()
        n_picks = 0
        for fidx in np.random.choice(len(frames), len(frames), replace=False):
            if avail[fidx] >= least_avail_markers:
                pool[f'{fname}_{fidx:06d}'] = frames[fidx]
                n_picks += 1
            if n_picks >= num_frames:
                break
        if len(pool) > 100:
            break
    if len(pool) < num_frames:
        raise ValueError(f'Not enough frames were found that have at least {least_avail_markers * 100.:.1f}% of the markers.\n'
                         f'either try moshpp.stagei_frame_picker.type: random or set '
                         f'moshpp.stagei_frame_picker.least_avail_markers to lower number in ange [0.1,1.0].')
    ids = np.random.choice(len(pool), num_frames, replace=False)
    return _obj_array(list(pool.values()))[ids], np.array(list(pool.keys()))[ids]
