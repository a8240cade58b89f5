"""CPU tests of the single-command multi-GPU launch (wavenet_vocoder/launch.py): `python bench.py --gpus N` and
`python train.py --hparams wavenet_num_gpus=N` start their own ranks -- the reference's one-command UX (hparams.py:37,
wavenet.py:227-239) on one process per GPU.  The rehearsal runs over gloo with a stand-in engine (bench.py --dry-run): it proves
the launch, the rendezvous on 127.0.0.1 and the product's bucketed tower-mean walk, never a number."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'WN_SELF_LAUNCHED'):
        env.pop(k, None)
    return env


def test_bench_gpus2_starts_two_ranks_that_rendezvous():
    """`python bench.py --gpus 2` with no launcher must not silently run one rank."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run', '--steps', '3', '--warmup', '1'],
                       capture_output=True, text=True, timeout=240, env=_clean_env())
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout                       # ONE line, from rank 0
    d = json.loads(lines[0])
    assert d['dry_run'] is True and d['value'] is None and d['ms_per_step'] is None        # never mistaken for a measurement
    assert d['n_gpus'] == 2 and d['collective'] == {'backend': 'gloo', 'world_size': 2, 'ranks_counted_by_allreduce': 2, 'self_launched': True}
    assert d['tower_mean_correct'] and d['replicas_identical']
    assert 'starting 2 ranks' in r.stderr
    # the N-rank line explains itself -- every rank's own step time and, per gradient bucket, the span from its
    # ready event to the end of its all-reduce, plus the caller stream's wait at the join (bench.py gather_per_rank / parallel.ExchangeTimer)
    pr = d['per_rank']
    assert [p['rank'] for p in pr] == [0, 1] and all(p['ms_per_step'] > 0 for p in pr)
    for p in pr:
        x = p['exchange']
        assert x['calls'] == 4 and len(x['bucket_ready_to_allreduce_end_ms']) == 3 and all(v >= 0 for v in x['bucket_ready_to_allreduce_end_ms'])
        assert x['caller_stream_wait_at_join_ms'] >= 0 and x['first_bucket_ready_to_last_allreduce_end_ms'] >= max(x['bucket_ready_to_allreduce_end_ms']) - 1e-6


def test_bench_accepts_the_c5_workload_under_n_ranks():
    """`--workload c5_stress --gpus N` (BASELINE configs[4]: the Gaussian / raw-PCM 24 kHz model on 8 GPUs) parses and launches."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--workload', 'c5_stress', '--dry-run', '--steps', '2', '--warmup', '1'],
                       capture_output=True, text=True, timeout=240, env=_clean_env())
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert d['n_gpus'] == 2 and d['config']['workload_key'] == 'c5_stress' and len(d['per_rank']) == 2
    # the fields the driver's 8-GPU run of this workload will carry exist and are filled by both ranks
    assert [p['rank'] for p in d['per_rank']] == [0, 1] and all(p['exchange']['calls'] == 3 for p in d['per_rank'])
    assert d['scaling_vs_n1'] is not None and d['scaling_vs_n1'] > 0 and d['n1_reference']['wall_s'] > 0
    assert d['dry_run'] is True and d['value'] is None


def test_rccl_topology_excerpt_reads_the_debug_files(tmp_path):
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    (tmp_path / 'rccl.h.1.log').write_text('h:1:1 [0] NCCL INFO Channel 00/08 : 0 1 2 3\nh:1:1 [0] NCCL INFO something else\nh:1:1 [0] NCCL INFO Channel 00 : 0[0] -> 1[1] via P2P/IPC\n')
    t = b.rccl_topology_excerpt(str(tmp_path))
    assert t['files'] == 1 and t['lines_matched'] == 2 and 'via P2P' in t['excerpt'][1]


def test_bench_refuses_more_gpus_than_the_node_has():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '64'], capture_output=True, text=True, timeout=240, env=_clean_env())
    assert r.returncode != 0 and 'asked for 64 GPUs' in r.stderr and not r.stdout.strip()


def test_bench_rejects_a_launcher_with_the_wrong_rank_count():
    env = _clean_env(); env.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run'], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode != 0 and '--gpus 2 but WORLD_SIZE=1' in r.stderr and not r.stdout.strip()


def test_spawn_ranks_propagates_failure_and_stops_the_other_ranks(tmp_path):
    from wavenet_vocoder import launch
    script = tmp_path / 'w.py'
    script.write_text('import os, sys, time\nr = int(os.environ["RANK"])\nassert os.environ["WORLD_SIZE"] == "3" and os.environ["MASTER_ADDR"] == "127.0.0.1"\n'
                      'if r == 1: sys.exit(7)\ntime.sleep(60)\n')
    import time
    t0 = time.time()
    assert launch.spawn_ranks([str(script)], 3) == 7
    assert time.time() - t0 < 30                           # ranks 0 and 2 were terminated, not waited for


def test_spawn_ranks_forwards_sigterm_and_retries_a_lost_port_race(tmp_path):
    """(1) a SIGTERM to the launching process must reach the ranks (they would keep the GPUs): run a launcher in a child
    interpreter, kill it with SIGTERM, and require that its ranks are gone; (2) free_port() releases the port before the ranks bind it --
    a rank that loses that race exits with launch.EADDRINUSE_RC and spawn_ranks starts the ranks again on a fresh port."""
    import signal
    import time
    from wavenet_vocoder import launch
    pidfile = tmp_path / 'pids'
    worker = tmp_path / 'w.py'
    worker.write_text('import os, time\nopen(%r, "a").write(str(os.getpid()) + "\\n")\ntime.sleep(120)\n' % str(pidfile))
    parent = tmp_path / 'p.py'
    parent.write_text('import sys\nsys.path.insert(0, %r)\nfrom wavenet_vocoder import launch\nsys.exit(launch.spawn_ranks([%r], 2))\n'
                      % (os.path.join(ROOT, 'tacotron-2_amd'), str(worker)))
    p = subprocess.Popen([sys.executable, str(parent)])
    t0 = time.time()
    while time.time() - t0 < 60 and (not pidfile.exists() or len(pidfile.read_text().split()) < 2):
        time.sleep(0.1)
    pids = [int(x) for x in pidfile.read_text().split()]
    assert len(pids) == 2
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=30) == 128 + signal.SIGTERM
    time.sleep(0.5)
    for pid in pids:                                       # the ranks were children of p: reaped by it, so the pid must be gone
        try:
            os.kill(pid, 0)
            alive = open('/proc/%d/stat' % pid).read().split()[2] != 'Z'
        except (ProcessLookupError, FileNotFoundError):
            alive = False
        assert not alive, pid
    # (2) first attempt: every rank reports the lost race; second attempt: success
    marker = tmp_path / 'attempt'
    flaky = tmp_path / 'f.py'
    flaky.write_text('import os, sys\nm = %r\nr = os.environ["RANK"]\n'
                     'if not os.path.exists(m + r):\n    open(m + r, "w").write(os.environ["MASTER_PORT"])\n    sys.exit(%d)\n'
                     'assert open(m + r).read() != "" \nsys.exit(0)\n' % (str(marker), launch.EADDRINUSE_RC))
    assert launch.spawn_ranks([str(flaky)], 2) == 0
    assert os.path.exists(str(marker) + '0') and os.path.exists(str(marker) + '1')


def test_train_cli_self_launches_when_wavenet_num_gpus_is_set(monkeypatch):
    spec = importlib.util.spec_from_file_location('train_cli', os.path.join(ROOT, 'tacotron-2_amd', 'train.py'))
    cli = importlib.util.module_from_spec(spec); spec.loader.exec_module(cli)
    from wavenet_vocoder import launch
    import types
    calls = []
    monkeypatch.setattr(launch, 'require_gpus', lambda n: calls.append(('need', n)))
    monkeypatch.setattr(launch, 'spawn_ranks', lambda argv, n, **k: calls.append(('spawn', n, argv[0])) or 0)
    for k in ('RANK', 'WORLD_SIZE'):
        monkeypatch.delenv(k, raising=False)
    assert cli._self_launch_if_asked(types.SimpleNamespace(hparams='wavenet_num_gpus=4,wavenet_batch_size=32')) == 0
    assert calls == [('need', 4), ('spawn', 4, os.path.join(ROOT, 'tacotron-2_amd', 'train.py'))]
    calls.clear()
    assert cli._self_launch_if_asked(types.SimpleNamespace(hparams='wavenet_num_gpus=1')) is None and not calls
    # already one of the ranks: no second launch; a rank-count mismatch is refused
    monkeypatch.setenv('RANK', '2'); monkeypatch.setenv('WORLD_SIZE', '4')
    assert cli._self_launch_if_asked(types.SimpleNamespace(hparams='wavenet_num_gpus=4')) is None and not calls
    monkeypatch.setenv('WORLD_SIZE', '8')
    with pytest.raises(SystemExit, match='launcher started 8 ranks'):
        cli._self_launch_if_asked(types.SimpleNamespace(hparams='wavenet_num_gpus=4'))
    from hparams import hparams
    hparams.parse('wavenet_num_gpus=1,wavenet_batch_size=8')
