# -*- coding: utf-8 -*-
"""Actor process (same flags as the reference's actor_node.py:9-16, plus --transport/--num-envs/...)."""
import sys
from distributed_rl_amd.runtime.nodes import actor_main

if __name__ == '__main__':
    sys.exit(actor_main())
