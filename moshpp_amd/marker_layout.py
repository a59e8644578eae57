"""Marker-layout files (the json the reference keeps next to a dataset) -- the part of
src/moshpp/marker_layout/edit_tools.py that Stage-I reads: `marker_layout_load` (:83-183), `marker_meta_filter` (:186-200),
`marker_layout_write` (:203-220).  Same arguments, same returned dict, same ordering rules (marker sets sorted by type, labels
sorted inside a set, alias table applied before sorting).  The mesh / visualisation helpers of that file are out of scope.
"""
from __future__ import annotations

import colorsys
import copy
import json
import logging
import os
from collections import OrderedDict

import numpy as np

from .mocap_interface import general_labels_map

logger = logging.getLogger('moshpp_amd')


def _red_to_blue(n):
    """`Color('red').range_to(Color('blue'), n)` of the `colour` package: a linear ramp in HSL hue from 0 to 2/3."""
    if n <= 1:
        return [(1.0, 0.0, 0.0)] * n
    return [colorsys.hls_to_rgb((2.0 / 3.0) * i / (n - 1), 0.5, 1.0) for i in range(n)]


def marker_layout_load(marker_layout_fname, labels_map=general_labels_map, include_nan=True, exclude_marker_types=None,
                       exclude_markers=None, only_markers=None, verbosity=1):
    """-> dict(marker_vids, marker_colors, marker_type, marker_type_mask, m2b_distance, surface_model_type, marker_layout_fname).
    `marker_layout_fname` may also be the already parsed json dict (tests)."""
    if isinstance(marker_layout_fname, dict):
        d = marker_layout_fname
        marker_layout_fname = d.get('marker_layout_fname')
    else:
        assert str(marker_layout_fname).endswith('.json')
        assert os.path.exists(marker_layout_fname), FileNotFoundError(marker_layout_fname)
        with open(marker_layout_fname) as f:
            d = json.load(f)
    only_markers = only_markers or []
    exclude_markers = exclude_markers or []
    exclude_marker_types = exclude_marker_types or []
    marker_vids, marker_types, m2b_distance = OrderedDict(), OrderedDict(), OrderedDict()
    if 'surface_model_type' not in d:
        logger.debug(f'Assuming SMPLx for marker layout since surface_model_type field was not available in: {marker_layout_fname}')
        surface_model_type = 'smplx'
    else:
        surface_model_type = d['surface_model_type']
    for markerset in sorted(d['markersets'], key=lambda a: a['type']):
        marker_type = markerset['type']
        if marker_type in exclude_marker_types:
            logger.debug(f'excluding marker_type {marker_type}')
            continue
        if marker_type in m2b_distance:
            raise ValueError(f'Marker type appears in multiple occasions: {markerset["type"]}!')
        m2b_distance[marker_type] = markerset.get('distance_from_skin', 0.0095)
        cur = markerset['indices']
        if labels_map:
            cur = {labels_map.get(k, k): cur[k] for k in cur}
        for label in sorted(cur):
            if only_markers and label not in only_markers:
                continue
            if label in exclude_markers:     # the reference only logs here (:150-151); the marker stays in the layout and
                logger.debug(f'excluding label {label}')   # is dropped on the mocap side by MocapSession(exclude_markers=...)
            if label in marker_vids:
                raise ValueError(f'Label ({label}) is present in multiple occasions.')
            marker_vids[label] = cur[label]
            marker_types.setdefault(marker_type, []).append(labels_map.get(label, label) if labels_map else label)
    marker_type_mask = OrderedDict((k, np.array([l in marker_types[k] for l in marker_vids])) for k in marker_types)
    marker_colors = OrderedDict(zip(marker_vids, _red_to_blue(len(marker_vids))))
    if include_nan:
        marker_colors['nan'] = [0.83, 1, 0]
    marker_type = OrderedDict()
    for lid, l in enumerate(marker_vids):
        for cur_type, mask in marker_type_mask.items():
            if mask[lid]:
                marker_type[l] = cur_type
    return {'marker_vids': marker_vids, 'marker_colors': marker_colors, 'marker_type': marker_type,
            'marker_type_mask': marker_type_mask, 'm2b_distance': m2b_distance, 'surface_model_type': surface_model_type,
            'marker_layout_fname': marker_layout_fname}


def marker_meta_filter(marker_meta, interested_labels):
    new_meta = copy.deepcopy(marker_meta)
    available = [l in interested_labels for l in marker_meta['marker_vids'].keys()]
    for marker_type, mask in new_meta['marker_type_mask'].items():
        new_meta['marker_type_mask'][marker_type] = (np.array(mask)[available]).tolist()
    new_meta['marker_vids'] = OrderedDict((k, v) for k, v in marker_meta['marker_vids'].items() if k in interested_labels)
    new_meta['marker_colors'] = OrderedDict((k, v) for k, v in marker_meta['marker_colors'].items()
                                            if k in list(interested_labels) + ['nan'])
    return new_meta


def marker_layout_write(marker_meta, marker_layout_fname):
    assert str(marker_layout_fname).endswith('.json')
    os.makedirs(os.path.dirname(os.path.abspath(marker_layout_fname)), exist_ok=True)
    labels = np.array(list(marker_meta['marker_vids'].keys()))
    layout = {'surface_model_type': marker_meta['surface_model_type'], 'markersets': []}
    for marker_type, mask in marker_meta['marker_type_mask'].items():
        vids = marker_meta['marker_vids']
        layout['markersets'].append({
            'indices': {str(l): [int(v) for v in vids[l]] if isinstance(vids[l], list) else int(vids[l])
                        for l in labels[np.asarray(mask, dtype=bool)]},
            'distance_from_skin': marker_meta['m2b_distance'][marker_type], 'type': marker_type})
    with open(marker_layout_fname, 'w') as f:
        json.dump(layout, f, sort_keys=True, indent=2, separators=(',', ': '))


def _marker_vid_tables():
    """(all_marker_vids, marker_type_labels) of the reference's marker_vids.py, shipped as data (tools/make_marker_vids.py)."""
    fn = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'marker_vids.json')
    with open(fn) as fh:
        d = json.load(fh)
    return d['all_marker_vids'], d['marker_type_labels']


def marker_labels_to_marker_layout(chosen_markers, marker_layout_fname, surface_model_type, labels_map=general_labels_map,
                                   wrist_markers_on_stick=False, separate_types=None) -> bool:
    """Create a marker-layout json for the labels seen in a capture (create_marker_layout_for_mocaps.py:44-131): known labels get
    their table vertex id, are sorted, typed (face / finger_left / finger_right / wrist-on-stick / body) and written with the default
    skin distances.  Unknown labels are skipped with an error message."""
    all_marker_vids, marker_type_labels = _marker_vid_tables()
    if separate_types is None:
        separate_types = ['body', 'face', 'finger']
    assert surface_model_type in all_marker_vids, \
        ValueError(f'No suitable database of labels found for surface_model_type: {surface_model_type}')
    mean_dist_from_skin = {'wrist': 0.039, 'body': 0.0095, 'face': 0.0002, 'finger_right': 0.0002, 'finger_left': 0.0002}
    has_face = surface_model_type in ['smplx', 'flame'] and 'face' in separate_types
    has_finger = surface_model_type in ['smplh', 'smplx', 'mano'] and 'finger' in separate_types
    has_body = surface_model_type not in ['mano', 'flame']
    unique_labels = list(set(labels_map.get(l, l) for l in chosen_markers))
    marker_vids, unknown = {}, []
    for l in sorted(unique_labels):
        if l not in all_marker_vids[surface_model_type]:
            unknown.append(l)
            continue
        marker_vids[l] = all_marker_vids[surface_model_type][l]
    if unknown:
        logger.error(f'Unknown marker label(s) for surface_model_type {surface_model_type} skipped: {unknown}.')
    n = len(marker_vids)
    mask = {}
    if has_face:
        mask['face'] = np.zeros(n, dtype=bool)
    if has_finger:
        mask['finger_left'] = np.zeros(n, dtype=bool)
        mask['finger_right'] = np.zeros(n, dtype=bool)
    if has_body:
        mask['body'] = np.zeros(n, dtype=bool)
    if wrist_markers_on_stick:
        mask['wrist'] = np.zeros(n, dtype=bool)
    for i, l in enumerate(marker_vids):
        if has_face and l in marker_type_labels['face']:
            mask['face'][i] = True
        elif has_finger and l in marker_type_labels['finger_left']:
            mask['finger_left'][i] = True
        elif has_finger and l in marker_type_labels['finger_right']:
            mask['finger_right'][i] = True
        elif wrist_markers_on_stick and l in marker_type_labels['wrist']:
            mask['wrist'][i] = True
        elif has_body:
            mask['body'][i] = True
        else:
            raise ValueError(f'Marker {l} could not be assigned to any marker type.')
    marker_layout_write({'marker_vids': marker_vids, 'marker_type_mask': {k: v for k, v in mask.items() if v.sum() != 0},
                         'm2b_distance': {k: mean_dist_from_skin[k] for k, v in mask.items() if v.sum() != 0},
                         'surface_model_type': surface_model_type}, marker_layout_fname)
    logger.info(f'Created marker layout: {marker_layout_fname}')
    return True
