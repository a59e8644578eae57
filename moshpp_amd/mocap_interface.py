"""Mocap ingestion for the Stage-II path (no ezc3d / psbody / human_body_prior).

API-compatible with the reference's `read_mocap` / `MocapSession`
(src/moshpp/tools/mocap_interface.py:87-279): same constructor arguments and attributes
(`markers` [metres], `labels`, `frame_rate`, `subject_mask`, `subject_names`, `multi_subject`,
`read_status`) and the same rules, re-implemented on NumPy masks:

  * labels: spaces removed, text before ':' is the subject and is stripped, then `labels_map` (:195-201);
  * `*`-labels / excluded / not-in-`only_markers` columns dropped (:203-217);
  * a sample is invalid iff it contains NaN or is exactly (0,0,0) (:275-279); invalid samples are zeroed (:223-225);
  * optional XYZ-Euler rotation in degrees, then division by the unit scale (:227-228, 245); 120 fps default.

`markers_aslabeled_arrays` is the array form of `markers_asdict` that the GPU solver consumes.
"""
from __future__ import annotations

import os
import pickle
from collections import OrderedDict
from pathlib import Path
from typing import Dict, List, Union

import numpy as np

from .c3d_io import read_c3d, write_c3d

UNIT_SCALE = {'mm': 1000., 'cm': 100., 'm': 1.}


def rotate_points_xyz(points, rxyz_deg):
    """points[..., 3] rotated by XYZ Euler angles given in degrees (R = Rz Ry Rx)."""
    ax, ay, az = np.radians(np.asarray(rxyz_deg, dtype=np.float64).ravel()[:3])
    cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
    R = np.array([[cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx],
                  [sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx],
                  [-sy, cy * sx, cy * cx]])
    return np.asarray(points).dot(R.T)


def write_mocap_c3d(markers: np.ndarray, labels: list, out_mocap_fname: str, frame_rate: int = 120) -> None:
    """markers[F,N,3] in metres -> .c3d in millimetres; NaN / all-zero samples become invalid points."""
    if not str(out_mocap_fname).endswith('.c3d'):
        raise AssertionError(out_mocap_fname)
    write_c3d(str(out_mocap_fname), np.asarray(markers, dtype=np.float64) * 1000., list(labels),
              frame_rate=frame_rate, units='mm')


# ---- per-format loaders: each returns (markers, labels | None, frame_rate | None, raw container) ------------
def _load_c3d(fname):
    c = read_c3d(fname)
    return c['points'], list(c['labels']), c['frame_rate'], c


def _load_npz(fname):
    z = np.load(fname, allow_pickle=True)
    keys = set(z.keys())
    rate = None
    if 'frame_rate' in keys:
        rate = z['frame_rate']
    elif 'required_parameters' in keys:
        rp = z['required_parameters']
        rp = rp.item() if isinstance(rp, np.ndarray) and rp.dtype == object else rp
        if 'frame_rate' in rp:
            rate = rp['frame_rate']
    labels = z['labels'].tolist() if 'labels' in keys else None
    return z['markers'], labels, rate, z


def _load_pkl(fname):
    with open(fname, 'rb') as f:
        d = pickle.load(f, encoding='latin-1')
    rate = None
    if 'required_parameters' in d:
        rate = d['required_parameters']['frame_rate']
    elif 'frame_rate' in d:
        rate = d['frame_rate']
    labels = d.get('labels', None)
    if isinstance(labels, np.ndarray):
        labels = labels.tolist()
    if labels is not None:   # non-string entries (a BMLmovi quirk) become anonymous labels
        labels = [f'*{i}' if isinstance(l, np.ndarray) else l for i, l in enumerate(labels)]
    return d['markers'], labels, rate, d


def _load_mat(fname):
    import scipy.io
    d = scipy.io.loadmat(fname)
    for key in ('Markers', 'MoCaps'):
        if key in d:
            labels = np.vstack(d['Labels'][0]).ravel().tolist() if 'Labels' in d else None
            return d[key], labels, None, d
    raise ValueError("The .mat file do not have the expected field for marker data! "
                     "Expected fields are ['MoCaps', 'Markers']")


_LOADERS = {'.c3d': _load_c3d, '.npz': _load_npz, '.pkl': _load_pkl, '.mat': _load_mat}


def read_mocap(mocap_fname):
    """-> {'markers' [F,N,3] file units, 'labels', 'frame_rate' | None, '_marker_data', 'subject_mask'}."""
    fname = str(mocap_fname)
    ext = os.path.splitext(fname)[1].lower()
    if ext not in _LOADERS:
        raise ValueError(f"Error! Could not recognize file format for {fname}")
    markers, labels, rate, raw = _LOADERS[ext](fname)
    markers = np.array(markers, dtype=np.float64)
    n = markers.shape[1]
    labels = [] if labels is None else [l.decode() if isinstance(l, bytes) else str(l) for l in labels]
    # unnamed trailing points: '*k' -- as the reference numbers them: a c3d file continues at the point index
    # (mocap_interface.py:126-127), every other format restarts at 0 (:145-146); a missing label list names all points '*i' (:143-144)
    if len(labels) < n:
        first = len(labels) if (ext == '.c3d' or not labels) else 0
        labels += [f'*{first + i}' for i in range(n - len(labels))]
    subjects = [l.split(':')[0] if ':' in l else 'null' for l in labels]
    subject_mask = OrderedDict()
    for s in subjects:
        if s not in subject_mask:
            subject_mask[s] = np.array([x == s for x in subjects], dtype=bool)
    if rate is not None:
        rate = float(np.asarray(rate).ravel()[0])
    return {'markers': markers, 'labels': labels, 'frame_rate': rate, '_marker_data': raw,
            'subject_mask': dict(subject_mask)}


def _load_label_aliases():
    """The reference's marker-label alias table (vendor / lab spellings -> canonical layout labels), applied on ingest via
    `labels_map=general_labels_map` (chmosh.py:466, mocap_interface.py:195-201).  Data file made by tools/make_label_aliases.py."""
    import json
    import os
    fn = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'label_aliases.json')
    with open(fn) as fh:
        return json.load(fh)


general_labels_map = _load_label_aliases()


class MocapSession(object):
    """One labelled optical capture: markers[F,N,3] in metres plus labels."""

    def __init__(self, mocap_fname: Union[str, Path], mocap_unit: str, mocap_rotate: list = None,
                 exclude_markers: List[str] = None, only_subjects: List[str] = None,
                 only_markers: List[str] = None, labels_map: dict = None,
                 ignore_stared_labels: bool = True, remove_label_before_colon: bool = True):
        self.mocap_fname = mocap_fname
        self.read_status = False
        if only_subjects and not isinstance(only_subjects, list):
            raise AssertionError(f'attribute only_subjects should be a list of strings as subject names: {only_subjects}')
        rec = read_mocap(mocap_fname)
        self._marker_data = rec['_marker_data']   # SOMA evaluation reads per-frame labels from here

        names = [l.replace(' ', '') for l in rec['labels']]
        if remove_label_before_colon:
            names = [l.rsplit(':', 1)[-1] for l in names]
        if labels_map is not None:
            names = [labels_map.get(l, l) for l in names]
        names = np.array(names, dtype=object)
        if only_markers is not None:
            keep = np.array([l in only_markers for l in names], dtype=bool)
        else:
            keep = np.ones(len(names), dtype=bool)
            if ignore_stared_labels:
                keep &= np.array([not l.startswith('*') for l in names], dtype=bool)
            if exclude_markers is not None:
                keep &= np.array([l not in exclude_markers for l in names], dtype=bool)
        names = names[keep]
        subject_mask = {s: m[keep] for s, m in rec['subject_mask'].items()}
        subject_names = sorted(subject_mask)
        markers = rec['markers'][:, keep]
        markers[~self.marker_availability_mask(markers)] = 0.
        if mocap_rotate is not None:
            markers = rotate_points_xyz(markers, mocap_rotate).reshape(markers.shape)
        if only_subjects:
            missing = [s for s in only_subjects if s not in subject_names]
            if missing:
                import logging
                logging.getLogger('moshpp_amd').error(
                    f'subject names {only_subjects} not available in mocap. available subjects: {subject_names}')
                return
            sel = np.zeros(markers.shape[1], dtype=bool)
            for s in only_subjects:
                sel |= subject_mask[s]
            subject_mask = {s: subject_mask[s][sel] for s in only_subjects}
            subject_names = only_subjects
            markers = markers[:, sel]
            names = names[sel]
        self.markers = markers / UNIT_SCALE[mocap_unit]
        self.labels = [str(l) for l in names]
        self.subject_mask = subject_mask
        self.subject_names = subject_names
        self.multi_subject = sum(1 for s in subject_names if s != 'null') > 1
        self.frame_rate = 120. if rec['frame_rate'] is None else rec['frame_rate']
        self.read_status = True

    @staticmethod
    def marker_availability_mask(markers):
        markers = np.asarray(markers)
        return ~np.isnan(markers).any(-1) & ~(markers == 0).all(-1)

    def markers_asdict(self) -> List[Dict[str, np.ndarray]]:
        """One OrderedDict label -> xyz per frame holding only the valid samples, in label order."""
        ok = self.marker_availability_mask(self.markers)
        return [OrderedDict((self.labels[i], self.markers[t, i]) for i in np.flatnonzero(ok[t]))
                for t in range(self.markers.shape[0])]

    def markers_aslabeled_arrays(self, latent_labels, frame_ids=None):
        """obs[F,M,3], vis[F,M]: `markers_asdict` restricted to and ordered by `latent_labels`
        (the row selection of chmosh.py:591-594).  With duplicate labels the last valid column wins,
        exactly as repeated dict assignment does."""
        frames = np.arange(len(self)) if frame_ids is None else np.asarray(list(frame_ids), dtype=np.int64)
        mk = self.markers[frames]
        ok = self.marker_availability_mask(mk)
        obs = np.zeros((len(frames), len(latent_labels), 3))
        vis = np.zeros((len(frames), len(latent_labels)), dtype=bool)
        col_of = {}
        for i, l in enumerate(self.labels):
            col_of.setdefault(l, []).append(i)
        for j, l in enumerate(latent_labels):
            for i in col_of.get(l, ()):
                obs[ok[:, i], j] = mk[ok[:, i], i]
                vis[:, j] |= ok[:, i]
        return obs, vis

    def __len__(self):
        return self.markers.shape[0]

    def __getitem__(self, given):
        return self.markers[given]

    def time_length(self):
        """seconds"""
        if self.frame_rate is None:
            raise AssertionError(f'mocap frame_rate is unknown: {self.mocap_fname}')
        return self.markers.shape[0] / self.frame_rate

    def write_as_c3d(self, out_c3d_fname: Union[str, Path]):
        write_mocap_c3d(self.markers, self.labels, str(out_c3d_fname), frame_rate=self.frame_rate)

    def write_as_npz(self, out_npz_fname: Union[str, Path]):
        if not str(out_npz_fname).endswith('.npz'):
            raise AssertionError(out_npz_fname)
        np.savez(out_npz_fname, markers=self.markers, labels=self.labels, frame_rate=self.frame_rate)
