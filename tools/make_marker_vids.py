"""Writes moshpp_amd/data/marker_vids.json: the reference's label -> vertex-id tables and the label lists per marker type
(src/moshpp/marker_layout/marker_vids.py:36-360; the SMPL-X table is completed from the SMPL one through
support_data/smplx_fit2_smplh.npz, :330-332).  `marker_labels_to_marker_layout` needs them to create a layout for a capture whose
labels it recognises.  Data, extracted mechanically.  Run in the build container only (needs /root/reference)."""
import importlib.util
import json
import os
import sys
import types

REF = '/root/reference/src/moshpp'
hbp = types.ModuleType('human_body_prior'); tools = types.ModuleType('human_body_prior.tools')
omni = types.ModuleType('human_body_prior.tools.omni_tools')
omni.get_support_data_dir = lambda f: '/root/reference/support_data'
sys.modules.update({'human_body_prior': hbp, 'human_body_prior.tools': tools, 'human_body_prior.tools.omni_tools': omni})
for name in ('moshpp', 'moshpp.marker_layout'):
    sys.modules[name] = types.ModuleType(name)


def load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


load('moshpp.marker_layout.markerset_smplh2smplx', 'marker_layout/markerset_smplh2smplx.py')
mv = load('moshpp.marker_layout.marker_vids', 'marker_layout/marker_vids.py')
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'moshpp_amd', 'data', 'marker_vids.json')
with open(out, 'w') as fh:
    json.dump({'all_marker_vids': {k: dict(sorted(v.items())) for k, v in mv.all_marker_vids.items()},
               'marker_type_labels': mv.marker_type_labels}, fh, separators=(',', ':'))
print(out, {k: len(v) for k, v in mv.all_marker_vids.items()})
