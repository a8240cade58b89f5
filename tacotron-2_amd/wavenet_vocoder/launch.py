"""Single-command multi-GPU launch: one process per GPU on this node.

The reference trains N towers inside ONE process (``python train.py --hparams wavenet_num_gpus=N``, hparams.py:37,
wavenet.py:227-239, 553-581).  Here every GPU is its own rank (torch.distributed, backend "nccl" == RCCL over xGMI), so the
single command re-executes itself N times with the rendezvous environment ``torch.distributed.run`` would set (RANK,
LOCAL_RANK, WORLD_SIZE, MASTER_ADDR = 127.0.0.1, MASTER_PORT = a free port).  Rank 0 inherits this process's stdout (bench.py's
one JSON line), the other ranks' stdout goes to stderr; the first rank to fail takes the others down (by PID, never by pattern)
and its exit code is returned.  A process that already runs under a launcher (WORLD_SIZE set) never calls this.
"""
import os
import socket
import subprocess
import sys
import time


def launched():
    """True when a launcher (torch.distributed.run or spawn_ranks) already set up this process as one rank."""
    return 'WORLD_SIZE' in os.environ and 'RANK' in os.environ


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def require_gpus(n):
    """Fail loudly -- before any rank starts -- when this node has fewer GPUs than ranks were asked for."""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n:
        raise SystemExit('asked for %d GPUs but this node shows %d (torch.cuda.device_count()); refusing to run %d ranks on fewer devices'
                         % (n, have, n))


def spawn_ranks(argv, n, env_extra=None, poll_s=0.2):
    """Run ``sys.executable argv...`` as ranks 0..n-1 and wait.  Returns the exit code (0 = every rank returned 0)."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({'RANK': str(r), 'LOCAL_RANK': str(r), 'WORLD_SIZE': str(n), 'LOCAL_WORLD_SIZE': str(n),
                    'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'WN_SELF_LAUNCHED': '1'})
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC only on this driver (RCCL needs it across processes)
        env.setdefault('GPU_MAX_HW_QUEUES', '8')
        if env_extra:
            env.update(env_extra)
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=env, stdout=None if r == 0 else sys.stderr))
    rc = 0
    try:
        live = list(procs)
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in live:            # a rank died: the others would hang in their next collective
                        q.terminate()
            time.sleep(poll_s)
    except BaseException:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        raise
    finally:
        deadline = time.time() + 10
        for p in procs:
            if p.poll() is None:
                try:
                    p.wait(timeout=max(0.1, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
    return rc
