# -*- coding: utf-8 -*-
"""Learner process (same flags as the reference's learner_node.py:12-19, plus --transport/--engine/--resume/...)."""
import sys
from distributed_rl_amd.runtime.nodes import learner_main

if __name__ == '__main__':
    sys.exit(learner_main())
