"""Deployment helpers (reference: fabfile.py, aws/aws_run_instances.py, aws/aws_config.yaml) — command construction only."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, path))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_fab_tasks_build_run_sh_commands():
    fab = _load("fabfile.py", "fabfile_t")
    if fab.HAVE_FABRIC:
        return
    assert fab.actor_run(num_proc='15', learner_host='10.1.1.1') == "cd distributed_rl && ./run.sh 15 config/actor.conf 10.1.1.1"
    assert fab.learner_run() == "cd distributed_rl && ./run.sh 0 config/learner.conf localhost"
    assert fab.all_run(num_proc='2').endswith("./run.sh 2")


def test_aws_plan_and_commands():
    aws = _load("aws/aws_run_instances.py", "aws_t")
    cfg = aws.load_config(os.path.join(ROOT, "aws", "aws_config.yaml"))
    p = aws.plan(cfg)
    assert p["actors_total"] == 75                      # 5 instances x 15 processes (aws_config.yaml:3-4)
    l, a = aws.fab_commands(cfg, ["1.1.1.1", "2.2.2.2"], "3.3.3.3")
    assert "learner_run" in l and "actor_run:num_proc=15,learner_host=3.3.3.3" in a and "1.1.1.1,2.2.2.2" in a
