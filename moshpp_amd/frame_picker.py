"""Stage-I frame pickers.

Behavioural mirror of the reference's three pickers (src/moshpp/frame_picker.py:43-213).  What has to agree is WHICH frames a seed
picks, so each picker consumes NumPy's legacy global RNG in the reference's order (`choice` per file, then `seed` + `shuffle` for
"random"; `seed` first, then a permutation per file and a final `choice` for "random_strict"); the tests replay the reference's own
functions on the same files and compare the picks (tests/test_ref_golden.py).

All pickers return `(frames, keys)`: an object array of per-frame `{label: xyz}` dicts (entries of `MocapSession.markers_asdict()`)
and the matching `<file>_<index:06d>` keys.
"""
from __future__ import annotations

import os

import numpy as np

from .mocap_interface import MocapSession

_POOL_LIMIT = 100      # the reference stops opening further files once the pool holds more frames than this


def _as_object_array(seq):
    arr = np.empty(len(seq), dtype=object)
    arr[:] = list(seq) if len(seq) != 1 else [seq[0]]
    return arr


def _session(fname, unit, rotate, subjects, only, exclude, aliases):
    return MocapSession(mocap_fname=fname, mocap_unit=unit, mocap_rotate=rotate, only_subjects=subjects, only_markers=only,
                        exclude_markers=exclude, labels_map=aliases)


def _named_markers_present(frame):
    """Labels of a frame dict that carry a finite sample and are not anonymous ('*...')."""
    return sum(1 for label, xyz in frame.items() if '*' not in label and not np.any(np.isnan(xyz)))


def load_marker_sessions_manual(mocap_fnames, mocap_unit, mocap_rotate=None, only_subjects=None, only_markers=None,
                                exclude_markers=None, labels_map={}):
    """Explicit picks: every entry is `<capture path>_<frame number>` (split at the LAST underscore)."""
    keys, picked = [], []
    for entry in mocap_fnames:
        path, _, number = entry.rpartition('_')
        if not os.path.exists(path):
            raise AssertionError(FileNotFoundError(path))
        index = int(number)
        keys.append('%s_%06d' % (path, index))
        picked.append(_session(path, mocap_unit, mocap_rotate, only_subjects, only_markers, exclude_markers,
                               labels_map).markers_asdict()[index])
    return _as_object_array(picked), np.array(keys)


def load_marker_sessions_random(mocap_fnames, mocap_unit, mocap_rotate=None, num_frames=12, only_subjects=None, seed=None,
                                least_avail_markers=.1, only_markers=None, exclude_markers=None, labels_map={}):
    """`num_frames` draws WITH replacement from every capture (made before the seed is applied -- the seed only fixes the shuffle),
    keyed by draw order; the shuffled pool is scanned for frames whose named, finite markers make up at least
    `least_avail_markers` of the frame.  If too few qualify the threshold drops by 0.01 and everything is redone (the retry does
    not pass `exclude_markers` on, as in the reference); below 0.01 it gives up."""
    threshold = least_avail_markers
    drop_excluded = exclude_markers
    while True:
        pool = {}
        for fname in mocap_fnames:
            session = _session(fname, mocap_unit, mocap_rotate, only_subjects, only_markers, drop_excluded, labels_map)
            per_frame = session.markers_asdict()
            for slot, src in enumerate(np.random.choice(len(session), num_frames)):
                pool['%s_%06d' % (fname, slot)] = per_frame[src]
            if len(pool) > _POOL_LIMIT:
                break
        order = list(range(len(pool)))
        if seed is not None:
            np.random.seed(seed=seed)
        np.random.shuffle(order)
        names, frames = list(pool.keys()), list(pool.values())
        keep = []
        for idx in order:
            if _named_markers_present(frames[idx]) >= threshold * len(frames[idx]):
                keep.append(idx)
            if len(keep) >= num_frames:
                break
        if len(keep) >= num_frames:
            return _as_object_array([frames[i] for i in keep]), np.array([names[i] for i in keep])
        threshold = threshold - 0.01
        drop_excluded = None
        if threshold < 0.01:
            raise ValueError(f'Not enough frames were found that have at least %{threshold * 100.:.1f} of the markers.\n')


def load_marker_sessions_random_strict(mocap_fnames, mocap_unit, mocap_rotate=None, num_frames=12, only_subjects=None, seed=None,
                                       least_avail_markers=.1, only_markers=None, exclude_markers=None, labels_map={}):
    """Seeded from the start.  Per capture: walk a random permutation of its frames and take up to `num_frames` whose share of
    valid samples reaches `least_avail_markers` (never lowered); finally `num_frames` of the pool, drawn without replacement."""
    np.random.seed(seed=seed)
    if not 0.1 <= least_avail_markers <= 1.0:
        raise AssertionError(least_avail_markers)
    pool = {}
    for fname in mocap_fnames:
        session = _session(fname, mocap_unit, mocap_rotate, only_subjects, only_markers, exclude_markers, labels_map)
        if not session.read_status:
            continue
        valid = MocapSession.marker_availability_mask(session.markers)
        share = valid.sum(-1) / valid.shape[1]
        per_frame = session.markers_asdict()
        taken = 0
        for src in np.random.choice(len(per_frame), len(per_frame), replace=False):
            if share[src] >= least_avail_markers:
                pool['%s_%06d' % (fname, src)] = per_frame[src]
                taken += 1
            if taken >= num_frames:
                break
        if len(pool) > _POOL_LIMIT:
            break
    if len(pool) < num_frames:
        raise ValueError(f'Not enough frames were found that have at least {least_avail_markers * 100.:.1f}% of the markers.\n'
                         f'Use moshpp.stagei_frame_picker.type: random, or lower moshpp.stagei_frame_picker.least_avail_markers '
                         f'(range [0.1, 1.0]).')
    chosen = np.random.choice(len(pool), num_frames, replace=False)
    return _as_object_array(list(pool.values()))[chosen], np.array(list(pool.keys()))[chosen]
