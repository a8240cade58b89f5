"""Run log: stdout + append-to-file (reference infolog.py).  The Slack webhook of the reference is accepted
and ignored (no network in this environment)."""
import atexit
from datetime import datetime

_format = '%Y-%m-%d %H:%M:%S.%f'
_file = None
_run_name = None


def init(filename, run_name, slack_url=None):
    global _file, _run_name
    _close_logfile()
    _file = open(filename, 'a', encoding='utf-8')
    _file.write('\n-----------------------------------------------------------------\n')
    _file.write('Starting new {} training run\n'.format(run_name))
    _file.write('-----------------------------------------------------------------\n')
    _run_name = run_name


def log(msg, end='\n', slack=False):
    print(msg, end=end)
    if _file is not None:
        _file.write('[%s]  %s\n' % (datetime.now().strftime(_format)[:-3], msg))
        _file.flush()


def _close_logfile():
    global _file
    if _file is not None:
        _file.close()
        _file = None


atexit.register(_close_logfile)
