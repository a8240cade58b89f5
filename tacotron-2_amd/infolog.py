"""Run log with the reference's surface (``infolog.init(path, run_name, slack_url)``, ``infolog.log(msg, end, slack)``; the
line format of ``Terminal_train_log`` is part of the drop-in surface), built for one process per GPU: rank 0 owns the terminal
and the log file, the other ranks are silent unless a message is marked ``all_ranks``.  The Slack webhook is accepted and
ignored (no network)."""
import atexit
import sys
from datetime import datetime


class _RunLog(object):
    banner = '-' * 65

    def __init__(self):
        self.rank = 0
        self.run_name = None
        self.fh = None

    def open(self, filename, run_name):
        self.close()
        self.run_name = run_name
        if self.rank != 0:
            return
        self.fh = open(filename, 'a', encoding='utf-8')
        self.fh.write('\n%s\nStarting new %s training run\n%s\n' % (self.banner, run_name, self.banner))

    def write(self, msg, end, all_ranks):
        if self.rank != 0:
            if all_ranks:
                sys.stdout.write('[rank %d] %s%s' % (self.rank, msg, end))
                sys.stdout.flush()
            return
        sys.stdout.write('%s%s' % (msg, end))
        if self.fh is not None:
            stamp = datetime.now().strftime('%Y-%m-%d %H:%M:%S.%f')[:-3]
            self.fh.write('[%s]  %s\n' % (stamp, msg))
            self.fh.flush()

    def close(self):
        if self.fh is not None:
            self.fh.close()
            self.fh = None


_LOG = _RunLog()
atexit.register(_LOG.close)


def set_rank(rank):
    """Call before ``init`` in a multi-rank job (train.py does, from RANK)."""
    _LOG.rank = int(rank)


def init(filename, run_name, slack_url=None):
    _LOG.open(filename, run_name)


def log(msg, end='\n', slack=False, all_ranks=False):
    _LOG.write(msg, end, all_ranks)
