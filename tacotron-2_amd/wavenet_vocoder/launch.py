"""Single-command multi-GPU launch: one process per GPU on this node.

The reference trains N towers inside ONE process (``python train.py --hparams wavenet_num_gpus=N``, hparams.py:37,
wavenet.py:227-239, 553-581).  Here every GPU is its own rank (torch.distributed, backend "nccl" == RCCL over xGMI), so the
single command re-executes itself N times with the rendezvous environment ``torch.distributed.run`` would set (RANK,
LOCAL_RANK, WORLD_SIZE, MASTER_ADDR = 127.0.0.1, MASTER_PORT = a free port).  Rank 0 inherits this process's stdout (bench.py's
one JSON line), the other ranks' stdout goes to stderr; the first rank to fail takes the others down (by PID, never by pattern)
and its exit code is returned.  A process that already runs under a launcher (WORLD_SIZE set) never calls this.
"""
import ctypes
import os
import signal
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


def init_process_group(**kw):
    """torch.distributed.init_process_group for a rank started by spawn_ranks: a rendezvous that fails because the port picked by
    free_port() was taken in the meantime ends this rank with EADDRINUSE_RC, which spawn_ranks answers with a fresh port."""
    import torch.distributed as dist
    try:
        return dist.init_process_group(**kw)
    except Exception as e:      # noqa: BLE001 -- DistNetworkError / RuntimeError, by message: torch has no stable exception type for it
        msg = str(e).lower()
        if os.environ.get('WN_SELF_LAUNCHED') == '1' and ('address already in use' in msg or 'eaddrinuse' in msg or 'errno: 98' in msg):
            sys.stderr.write('launch: rendezvous on port %s failed (%s)\n' % (os.environ.get('MASTER_PORT'), str(e).splitlines()[0][:160]))
            sys.exit(EADDRINUSE_RC)
        raise


def require_gpus(n):
    """Fail loudly -- before any rank starts -- when this node has fewer GPUs than ranks were asked for."""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n:
        raise SystemExit('asked for %d GPUs but this node shows %d (torch.cuda.device_count()); refusing to run %d ranks on fewer devices'
                         % (n, have, n))


# libc's prctl is resolved ONCE, at import: the preexec_fn below runs in the forked child of a multi-threaded parent (torch is already
# imported by then), where dlopen / a Python-level allocation can deadlock on a lock some other thread held at fork() (subprocess docs;
# ADVICE round 5).  The child then only calls the pre-bound C function.
try:
    _PRCTL = ctypes.CDLL('libc.so.6', use_errno=True).prctl
    _PRCTL.argtypes = [ctypes.c_int, ctypes.c_ulong, ctypes.c_ulong, ctypes.c_ulong, ctypes.c_ulong]
    _PRCTL.restype = ctypes.c_int
except (OSError, AttributeError):      # not Linux / no libc.so.6: the SIGTERM forwarding of spawn_ranks is the portable path
    _PRCTL = None
_PR_SET_PDEATHSIG = 1


def _die_with_parent():
    """preexec_fn of every rank (Linux): SIGTERM when the launching THREAD dies, however it dies (a SIGKILLed parent cannot forward
    anything).  PR_SET_PDEATHSIG is tied to the thread that forked: spawn_ranks must therefore be called from a thread that outlives the
    ranks -- the main thread in bench.py / train.py (spawn_ranks says so when it is not)."""
    if _PRCTL is not None:
        _PRCTL(_PR_SET_PDEATHSIG, int(signal.SIGTERM), 0, 0, 0)


class _Terminated(BaseException):
    pass


def spawn_ranks(argv, n, env_extra=None, poll_s=0.2, attempts=3):
    """Run ``sys.executable argv...`` as ranks 0..n-1 and wait.  Returns the exit code (0 = every rank returned 0).
    A SIGTERM / SIGINT to this process is forwarded to every rank (they would otherwise keep the GPUs); the port is picked by binding
    port 0 and released before the ranks bind it, so a lost race (exit code EADDRINUSE_RC of a rank within the first seconds: another
    process took the port) is retried on a fresh port up to `attempts` times."""
    rc = 0
    for attempt in range(attempts):
        t0 = time.time()
        rc, stderr_tail_hint = _spawn_once(argv, n, env_extra, poll_s)
        if rc == EADDRINUSE_RC and time.time() - t0 < 60 and attempt + 1 < attempts:
            sys.stderr.write('launch: rendezvous port was taken by another process, retrying on a fresh port (%d/%d)\n' % (attempt + 2, attempts))
            continue
        break
    return rc


EADDRINUSE_RC = 98      # a rank whose rendezvous failed with "address already in use" exits with errno EADDRINUSE (bench.py / train.py map it)


def _spawn_once(argv, n, env_extra, poll_s):
    import threading
    port = free_port()
    procs = []

    def on_term(signum, frame):
        raise _Terminated(signum)
    old = {}
    main_thread = threading.current_thread() is threading.main_thread()
    if not main_thread:
        sys.stderr.write('launch: spawn_ranks called from a non-main thread: signals are not forwarded by this call, and the ranks receive '
                         'SIGTERM when THIS thread exits (PR_SET_PDEATHSIG follows the forking thread) -- keep it alive until they are done\n')
    rc = 0
    # everything from the first handler installed to the last rank reaped is inside ONE try: a signal that arrives while the ranks are
    # still being started terminates the ones already running and restores the handlers (the Popen loop used to sit
    # in front of the try block)
    try:
        for sg in (signal.SIGTERM, signal.SIGINT):
            try:
                old[sg] = signal.signal(sg, on_term)
            except ValueError:          # not the main thread: the caller owns signal handling
                pass
        for r in range(n):
            env = dict(os.environ)
            env.update({'RANK': str(r), 'LOCAL_RANK': str(r), 'WORLD_SIZE': str(n), 'LOCAL_WORLD_SIZE': str(n),
                        'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'WN_SELF_LAUNCHED': '1'})
            env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC only on this driver (RCCL needs it across processes)
            env.setdefault('GPU_MAX_HW_QUEUES', '8')
            if env_extra:
                env.update(env_extra)
            procs.append(subprocess.Popen([sys.executable] + list(argv), env=env, stdout=None if r == 0 else sys.stderr, preexec_fn=_die_with_parent))
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
    except _Terminated as t:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        rc = 128 + int(t.args[0])
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
        for sg, h in old.items():
            signal.signal(sg, h)
    return rc, None
