"""bench.py's ONE stdout line: the driver keeps the tail of stdout, and round 5's 20 KB line came back unparsed (BENCH_r05.parsed = null).
The line builder is run here on canned results -- the whole round-5 result (profiles/r05_bench_line.json, 20 KB) and inflated / broken
variants of it -- and must give strict JSON under bench.LINE_LIMIT with every key of the contract."""
import copy
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
            'data', 'config', 'roofline', 'cpu_baseline')


@pytest.fixture(scope='module')
def full():
    with open(os.path.join(ROOT, 'profiles', 'r05_bench_line.json')) as fh:
        return json.load(fh)


def _strict(line):
    def no_const(x):
        raise ValueError(x)
    return json.loads(line, parse_constant=no_const)


def test_line_is_short_strict_json_with_the_contract_keys(full):
    line = bench.compact_line(bench._finite(full))
    assert len(line.encode()) < bench.LINE_LIMIT <= 3600 and '\n' not in line
    d = _strict(line)
    for k in CONTRACT:
        assert k in d, k
    assert d['value'] == full['value'] and d['ms_per_step'] == full['ms_per_step'] and d['dtype'] == 'f64'
    assert set(d['roofline']) >= {'kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'}
    assert set(d['cpu_baseline']) >= {'value', 'unit', 'cores', 'kind', 'sample'}
    assert d['roofline_lbs']['frac'] == full['roofline_lbs']['frac'] and 'workload' in d['config'] and 'model' not in d['config']
    assert d['parity']['frames_outside_tolerance'] == 0 and d['config3']['frames_per_s'] and d['many_sequences']['frames_per_s']


def test_line_stays_short_when_the_result_grows_and_survives_broken_legs(full):
    big = copy.deepcopy(full)
    big['parity_every_frame']['by_seed'] = {str(i): big['parity_every_frame']['by_seed'] for i in range(40)}   # ~300 KB
    big['seeds'] = {str(i): {'frames_per_s': 1.0 * i, 'ms_per_step': 2.0} for i in range(400)}
    big['config']['workload'] = 'x' * 5000
    big['cpu_baseline']['sample'] = 'y' * 5000
    big['roofline_lbs'] = {'error': 'RuntimeError(' + 'z' * 300 + ')'}
    big['config3'] = {'error': 'boom'}
    big['stagei'] = {'seconds': float('nan'), 'dogleg_iterations': 3}
    big['roofline']['traffic'] = None
    fin = bench._finite(big)
    assert 'non_finite' in fin and fin['stagei']['seconds'] is None
    line = bench.compact_line(fin)
    assert len(line.encode()) < bench.LINE_LIMIT
    d = _strict(line)
    for k in CONTRACT:
        assert k in d, k
    assert d['roofline']['traffic'] is None and d['value'] == full['value']


def test_the_strong_scaling_headline_keeps_its_keys(full):
    st = copy.deepcopy(full)
    st.update(scaling='strong', n_gpus=8, rccl={'backend': 'nccl (RCCL)', 'ranks_seen_by_an_all_reduce': 8, 'distinct_devices': True},
              one_gpu_same_job={'frames_per_s': 1.0, 'ms': 2.0, 'speedup': 6.6, 'note': 'n' * 500}, speedup_vs_one_gpu_same_job=6.6)
    st['config'] = {'workload': 'w' * 900, 'mode': 'chunked', 'markers': 53, 'parallelism': '8 ranks'}
    d = _strict(bench.compact_line(bench._finite(st)))
    assert d['scaling'] == 'strong' and d['n_gpus'] == 8 and d['rccl']['ranks_seen_by_an_all_reduce'] == 8
    assert d['one_gpu_same_job']['speedup'] == 6.6 and d['config']['parallelism'] == '8 ranks'
