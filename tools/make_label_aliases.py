"""Writes moshpp_amd/data/label_aliases.json: the marker-label alias table the reference applies while ingesting a capture
(`labels_map=general_labels_map`, src/moshpp/chmosh.py:466 -> tools/mocap_interface.py:195-201; the table itself is data in
src/moshpp/marker_layout/labels_map.py:34-231: vendor / lab spellings -> the canonical label names of the marker layouts).
It is data the drop-in needs for ingest parity, so it is extracted mechanically here instead of being retyped.
Run in the build container only (needs /root/reference)."""
import importlib.util
import json
import os

spec = importlib.util.spec_from_file_location('ref_labels_map', '/root/reference/src/moshpp/marker_layout/labels_map.py')
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'moshpp_amd', 'data', 'label_aliases.json')
with open(out, 'w') as fh:
    json.dump(dict(sorted(mod.general_labels_map.items())), fh, separators=(',', ':'))
print(out, len(mod.general_labels_map), 'aliases')
