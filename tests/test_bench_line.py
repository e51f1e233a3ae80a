"""The ONE stdout line of bench.py: bounded size, valid JSON, contract keys (round 5's 20.8 KB line
outgrew the driver's capture and the round went unmeasured).  No GPU needed: the compaction is a pure function of the
full record; it is run on the real round-5 record (tests/golden/bench_record_r05.json = the line bench.py printed on an
MI355X in round 5) and on the record a --dry run fabricates with every table populated."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location('bench_line_mod', os.path.join(ROOT, 'bench.py'))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b


def _check(b, line):
    assert len(line) <= b.LINE_LIMIT <= 6000 and '\n' not in line
    c = json.loads(line)
    for k in b.REQUIRED_KEYS:
        assert k in c, k
    for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'algorithmic_bytes_per_launch', 'avg_launch_ms'):
        assert k in c['roofline'], k
    assert set(c['cpu_baseline']) >= {'value', 'unit', 'cores', 'kind', 'sample'}
    assert 'workload' in c['config'] and 'whole_step_frac' in c['end_to_end']
    return c


def test_real_round5_record_fits_and_keeps_the_contract():
    b = _bench()
    with open(os.path.join(ROOT, 'tests', 'golden', 'bench_record_r05.json')) as f:
        full = json.load(f)
    assert len(json.dumps(full)) > 20000          # the record that broke the driver's capture
    c = _check(b, b.compact_line(full))
    assert c['value'] == full['value'] and c['ms_per_step'] == full['ms_per_step'] and c['dtype'] == 'f16x3'
    assert c['roofline']['frac'] == full['roofline']['frac'] and c['cpu_baseline']['value'] == full['cpu_baseline']['value']
    # one row per other BASELINE config: pairs/s + whole-step fraction
    w = c['workloads']
    assert set(w) == {'columns', 'cfg2_b32', 'cfg4_b32_per_gpu', 'cfg5_image_only', 'cfg5_lidar_only'}
    assert w['cfg4_b32_per_gpu'][0] == full['extra']['workloads']['cfg4_b32_per_gpu']['value']
    assert 'dropped_for_size' not in c
    # the launch classes furthest below their roofline survive the compaction
    rows = {r[0]: r for r in c['kernels']['cfg4_b32_per_gpu']['rows']}
    assert 'rows GEMM 512->1024 (pair prologue)' in rows


def test_oversized_record_drops_optional_blocks_never_contract_fields():
    b = _bench()
    with open(os.path.join(ROOT, 'tests', 'golden', 'bench_record_r05.json')) as f:
        full = json.load(f)
    # 40 legs with long tables: far more than fits
    for i in range(40):
        full['extra']['kernels']['leg%d' % i] = full['extra']['kernels']['cfg5_lidar_only']
        full['extra']['workloads']['wl%d' % i] = full['extra']['workloads']['cfg2_b32']
    c = _check(b, b.compact_line(full))
    assert c['dropped_for_size'] and c['value'] == full['value']
    # contract strings of absurd length are clipped, not allowed to push the line over
    full['config']['workload'] = 'w' * 5000
    full['roofline']['kernel'] = 'k' * 5000
    full['cpu_baseline']['sample'] = 's' * 5000
    _check(b, b.compact_line(full))
    with pytest.raises(RuntimeError):
        b.compact_line(full, limit=500)


def test_dry_run_prints_one_bounded_line_and_the_detail_record(tmp_path):
    b = _bench()
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--dry', '--workload', 'tiny', '--steps', '2',
                        '--warmup', '1', '--pairs', '3'], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = p.stdout.splitlines()
    assert len(lines) == 1                       # stdout carries exactly one line
    c = _check(b, lines[0])
    assert c['n_gpus'] == 1 and c['steps'] == 2 and c['warmup'] == 1
    det = [l for l in p.stderr.splitlines() if l.startswith('BENCH_DETAIL ')]
    assert len(det) == 1
    full = json.loads(det[0][len('BENCH_DETAIL '):])
    # the fabricated record has every table of the GPU run: 5 legs x 25 launch classes
    assert len(full['extra']['kernels']) == 5 + 2 and all(len(t['rows']) == 25 for k, t in full['extra']['kernels'].items()
                                                          if isinstance(t, dict))
    assert len(json.dumps(full)) > 3 * b.LINE_LIMIT
    with open(os.path.join(ROOT, c['detail'])) as f:
        assert json.load(f)['extra'].keys() == full['extra'].keys()
    # --detail stdout: the record as an EARLIER stdout line, the compact line still last
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--dry', '--workload', 'tiny', '--steps', '1',
                        '--warmup', '0', '--pairs', '2', '--detail', 'stdout'], capture_output=True, text=True, timeout=300,
                       env=env, cwd=ROOT)
    lines = p.stdout.splitlines()
    assert len(lines) == 2 and lines[0].startswith('BENCH_DETAIL ')
    _check(b, lines[1])
