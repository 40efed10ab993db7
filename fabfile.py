# coding: utf-8
"""Remote launch tasks (reference: fabfile.py:4-23). Same task names and arguments; they shell out to ./run.sh.
Works with fabric3's `fab` CLI when it is installed; every task is also callable as
`python fabfile.py <task> [key=value ...]`, which prints (or, with --run, executes over ssh) the command."""
import shlex
import subprocess
import sys

try:  # optional: fabric3 is not a dependency of the framework
    from fabric.api import cd, local, parallel, run, settings  # type: ignore
    HAVE_FABRIC = True
except Exception:  # pragma: no cover
    HAVE_FABRIC = False

    def parallel(f):
        return f


def _cmd(num_proc, conf=None, host=None):
    parts = ["./run.sh", str(num_proc)]
    if conf:
        parts.append(conf)
    if host:
        parts.append(host)
    return " ".join(parts)


def local_run(num_proc='4'):
    c = _cmd(num_proc)
    return local(c) if HAVE_FABRIC else subprocess.call(shlex.split(c))


def all_run(directory='distributed_rl', num_proc='4'):
    if not HAVE_FABRIC:
        return "cd %s && %s" % (directory, _cmd(num_proc))
    with cd(directory):
        with settings(warn_only=True):
            run(_cmd(num_proc))


@parallel
def actor_run(directory='distributed_rl', num_proc='1', learner_host='localhost'):
    print('num_proc: %s' % num_proc)
    print('learner_host: %s' % learner_host)
    if not HAVE_FABRIC:
        return "cd %s && %s" % (directory, _cmd(num_proc, "config/actor.conf", learner_host))
    with cd(directory):
        with settings(warn_only=True):
            run(_cmd(num_proc, "config/actor.conf", learner_host))


def learner_run(directory='distributed_rl'):
    if not HAVE_FABRIC:
        return "cd %s && %s" % (directory, _cmd(0, "config/learner.conf", "localhost"))
    with cd(directory):
        with settings(warn_only=True):
            run(_cmd(0, "config/learner.conf", "localhost"))


if __name__ == '__main__':
    task = sys.argv[1] if len(sys.argv) > 1 else 'local_run'
    kwargs = dict(a.split('=', 1) for a in sys.argv[2:] if '=' in a)
    print(globals()[task](**kwargs))
