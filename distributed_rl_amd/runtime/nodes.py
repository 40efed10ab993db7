"""Process entrypoints behind ``actor_node.py`` / ``learner_node.py`` (reference: actor_node.py:8-37,
learner_node.py:11-47). The reference's flags are kept verbatim (``-n -e -a -t -r -v`` / ``-e -a -r -v -d -s``);
new flags select the transport (shm by default — no Redis needed on one node), the number of envs
per actor process, the execution engine and bounded runs for tests.
"""
from __future__ import annotations

import argparse
import os
import sys

import torch
import torch.optim as optim

from .. import envs
from ..config import make_config
from ..models import DuelingDQN, DuelingLSTMDQN
from ..obs import make_metrics
from ..transport import connect, shm_names


def _common(parser):
    parser.add_argument('-e', '--env', type=str, default='MultiFrameBreakout-v0', help='Environment name.')
    parser.add_argument('-a', '--algorithm', type=str, default='ape_x', choices=['ape_x', 'r2d2'], help='Select an algorithm.')
    parser.add_argument('-r', '--redisserver', type=str, default='localhost', help="Redis's server name (transport=redis).")
    parser.add_argument('-v', '--visdomserver', type=str, default='', help="Visdom's server name ('' = no visdom sink).")
    parser.add_argument('--transport', type=str, default=os.environ.get('DRL_TRANSPORT', 'shm'), choices=['shm', 'redis', 'inproc'])
    parser.add_argument('--run-id', type=str, default=os.environ.get('DRL_RUN_ID', 'default'), help='Namespace of the shm segments.')
    parser.add_argument('--max-steps', type=int, default=None, help='Stop after this many steps (default: run forever).')
    parser.add_argument('--seed', type=int, default=None)
    parser.add_argument('--log-dir', type=str, default='', help='Write JSONL metrics here.')
    parser.add_argument('--quiet', action='store_true')
    parser.add_argument('--device', type=str, default='', help="'' = cuda if available else cpu.")
    parser.add_argument('--param-channel', type=str, default='', choices=['', 'collective'],
                        help="collective = this process is a member of the launcher's torch.distributed world "
                             "(DRL_WORLD_PLAN / _RANK / _ADDR / _PORT, runtime/placement.py): actors receive parameters by "
                             "broadcast over their lane (RCCL between GPUs, gloo on CPU) instead of the transport's register.")
    parser.add_argument('--param-backend', type=str, default='gloo', choices=['gloo', 'nccl'],
                        help="what the lanes carry: nccl = device tensors over RCCL (every lane member on its own GPU), "
                             "gloo = host tensors.")


def shard_run_id(run_id: str, rank: int, world: int) -> str:
    """Data-parallel learners: every learner rank owns its OWN experience ring + parameter register (a replay shard in
    its own HBM fed by its own actors); rank 0 keeps the plain run id so single-learner tools keep working."""
    return run_id if world <= 1 or rank == 0 else "%s_r%d" % (run_id, rank)


def _flat_numel(algorithm: str, n_action: int) -> int:
    """Size of the flat fp32 parameter buffer both sides of a parameter channel agree on (models/packing.py)."""
    from ..models.packing import ParamPack, apex_layout, r2d2_layout
    lay = apex_layout(n_action) if algorithm == 'ape_x' else r2d2_layout(n_action)
    return ParamPack(lay, 'cpu').flat.numel()


def _lane_device(world, device: torch.device) -> torch.device:
    """Where a lane's tensors live: the member's GPU when the lanes carry device tensors (RCCL), else the host."""
    return device if (world.param_backend == 'nccl' and device.type == 'cuda') else torch.device('cpu')


def _actor_lane(world, numel: int, device: torch.device):
    """Actor side of the collective parameter channel: the receiver of the ONE lane this world rank belongs to."""
    from ..transport.collective_params import CollectiveParams
    mine = world.my_lanes()
    if len(mine) != 1:
        raise SystemExit("world rank %d belongs to %d parameter lanes (expected exactly one)" % (world.rank, len(mine)))
    lane = mine[0]
    return CollectiveParams(numel, src_rank=world.lanes[lane][0], group=world.lane_groups[lane], device=_lane_device(world, device))


def _learner_fanout(world, register, numel: int, device: torch.device, keep_register: bool):
    """Learner side: the shard's register (co-located / CPU actors) + every lane that starts at this learner rank."""
    from ..transport.collective_params import CollectiveParams, FanoutParams
    lanes = [CollectiveParams(numel, src_rank=world.rank, group=world.lane_groups[i], device=_lane_device(world, device))
             for i in world.my_lanes()]
    return FanoutParams(register if keep_register else None, lanes)


def _device(name: str) -> torch.device:
    """'' = this process was started by hand: cuda (the current device) if there is one. The launcher always names the
    device (runtime/placement.py), so actors of a multi-GPU node never pile up on GPU 0."""
    if name:
        return torch.device(name)
    return torch.device("cuda" if torch.cuda.is_available() else "cpu")


def _heartbeat(args, role, ident):
    if args.transport != 'shm':
        return None
    try:
        from .heartbeat import HeartbeatTable
        return HeartbeatTable(shm_names(args.run_id)[2], create=False).register(role, ident)
    except OSError:
        return None


def actor_main(argv=None) -> int:
    parser = argparse.ArgumentParser(description='Actor process for distributed reinforcement.')
    parser.add_argument('-n', '--no', type=int, default=1, help='Actor number.')
    parser.add_argument('-t', '--num_total_actors', type=int, default=4, help='Total number of actors.')
    parser.add_argument('--num-envs', type=int, default=int(os.environ.get('DRL_NUM_ENVS', '1')), help='Envs per actor process.')
    parser.add_argument('--actor-batch', type=int, default=None)
    parser.add_argument('--param-period', type=int, default=None)
    parser.add_argument('--engine', type=str, default='auto', choices=['auto', 'torch', 'hip'],
                        help='hip = batched gfx950 inference engine (Ape-X and R2D2; needs a GPU).')
    parser.add_argument('--render-every', type=int, default=0,
                        help="Send env 0's screen to the metrics sinks every N env steps (0 = never). The reference sends one "
                             "frame per step over HTTP (ape_x/actor.py:82); here it is opt-in and rate-limited.")
    parser.add_argument('--learner-ranks', type=int, default=int(os.environ.get('DRL_LEARNER_RANKS', '1')),
                        help='Number of data-parallel learner ranks; actor i feeds the shard of rank (i-1) %% ranks.')
    _common(parser)
    args = parser.parse_args(argv)
    from .actor import Actor
    device = _device(args.device)
    cfg = make_config(args.algorithm, actor_batch=args.actor_batch, actor_param_period=args.param_period)
    env = envs.make_vec(args.env, args.num_envs, seed=args.seed)
    metrics = make_metrics("actor_%d" % args.no, args.visdomserver, args.log_dir, args.quiet)
    shard = shard_run_id(args.run_id, (args.no - 1) % max(1, args.learner_ranks), args.learner_ranks)
    import time as _time
    t_end = _time.time() + float(os.environ.get("DRL_CONNECT_TIMEOUT", "120"))
    while True:                    # the learner (possibly several ranks under torch.distributed.run) creates the segments
        try:
            exp, par = connect(args.transport, shard, create=False, host=args.redisserver)
            break
        except OSError:
            if _time.time() > t_end:
                raise
            _time.sleep(0.25)
    world = None
    if args.param_channel == 'collective':
        from ..parallel.world import join_world
        world = join_world(device)
        if world is None:
            raise SystemExit("--param-channel collective without a world plan (DRL_WORLD_PLAN): start through run.sh / the launcher")
        par.close()
        par = _actor_lane(world, _flat_numel(args.algorithm, env.n_action), device)
    if device.type == 'cuda':
        # an explicit index from here on ("cuda" alone = the current device): set_device and the IPC-mapped parameter mirror
        # (transport.ShmParams) need one
        device = torch.device('cuda', device.index if device.index is not None else torch.cuda.current_device())
        torch.cuda.set_device(device)
    hb = _heartbeat(args, "actor", args.no)
    if args.algorithm == 'ape_x':
        engine = None
        use_hip = args.engine == 'hip' or (args.engine == 'auto' and device.type == 'cuda')
        net = DuelingDQN(env.n_action).to('cpu' if use_hip else device)
        if use_hip:
            from ..engine.actor import ActorEngine
            from ..utils import epsilon_ladder
            eps = epsilon_ladder(args.num_envs, (args.no - 1) * args.num_envs + 1, args.num_total_actors * args.num_envs)
            engine = ActorEngine(env.n_action, args.num_envs, device, eps=eps, seed=args.seed or args.no, init_model=net)
        actor = Actor(args.no, env, net, cfg, exp, par, metrics, args.num_total_actors, device, heartbeat=hb, seed=args.seed,
                      engine=engine, render_every=args.render_every)
    else:
        engine = None
        use_hip = args.engine == 'hip' or (args.engine == 'auto' and device.type == 'cuda')
        net = DuelingLSTMDQN(env.n_action, args.num_envs, cfg.burn_in, cfg.nstep).to('cpu' if use_hip else device)
        tgt = DuelingLSTMDQN(env.n_action, args.num_envs, cfg.burn_in, cfg.nstep).to('cpu' if use_hip else device)
        if use_hip:
            from ..engine.r2d2_actor import R2D2ActorEngine
            from ..utils import epsilon_ladder
            eps = epsilon_ladder(args.num_envs, (args.no - 1) * args.num_envs + 1, args.num_total_actors * args.num_envs)
            engine = R2D2ActorEngine(env.n_action, args.num_envs, cfg, device, eps=eps, seed=args.seed or args.no,
                                     init_model=net, done_offset=cfg.nstep if cfg.reference_nstep_order else 0)
        actor = Actor(args.no, env, net, cfg, exp, par, metrics, args.num_total_actors, 'cpu' if use_hip else device,
                      target_net=tgt, heartbeat=hb, seed=args.seed, engine=engine, render_every=args.render_every)
    actor.pull_params()
    try:
        actor.run(max_steps=args.max_steps)
    except KeyboardInterrupt:
        pass
    finally:
        if hb is not None:
            hb.close()
    return 0


def learner_main(argv=None) -> int:
    parser = argparse.ArgumentParser(description='Learner process for distributed reinforcement.')
    parser.add_argument('-d', '--actordevice', type=str, default='', help="Actor's device (kept for compatibility).")
    parser.add_argument('-s', '--replaysize', type=int, default=100000, help="Replay memory size.")
    parser.add_argument('--save-dir', type=str, default='./models')
    parser.add_argument('--resume', type=str, default='', help='model_<t>.pth to resume from (+ its .aux.pth / .replay.pth sidecars).')
    parser.add_argument('--save-replay', action='store_true',
                        help='HIP engine: checkpoints also snapshot the HBM replay (model_<t>.replay.pth; can be tens of GB).')
    parser.add_argument('--batch-size', type=int, default=None)
    parser.add_argument('--start-memory-size', type=int, default=None)
    parser.add_argument('--save-timing', type=int, default=None)
    parser.add_argument('--target-update', type=int, default=None)
    parser.add_argument('--engine', type=str, default='auto', choices=['auto', 'torch', 'hip'],
                        help='torch = ATen learner; hip = graph-captured gfx950 engine (needs a GPU).')
    parser.add_argument('--register-actors', type=int, default=-1,
                        help="number of this shard's actors that read the parameter register (-1 = unknown: keep publishing "
                             "to it); 0 lets a learner whose actors are all on broadcast lanes skip the host copy.")
    parser.add_argument('--attach', action='store_true',
                        help="take over the EXISTING experience ring / parameter register of --run-id instead of creating fresh "
                             "ones: a learner restarted by the launcher (self-healing) next to actors that keep running.")
    parser.add_argument('--ring-slots', type=int, default=64)
    parser.add_argument('--ring-slot-mb', type=int, default=8)
    _common(parser)
    args = parser.parse_args(argv)
    device = _device(args.device)
    cfg = make_config(args.algorithm, batch_size=args.batch_size, start_memory_size=args.start_memory_size,
                      save_timing=args.save_timing, target_update=args.target_update)
    n_action = envs.n_action_of(args.env)             # the reference builds a whole env just for this (learner_node.py:20)
    # data-parallel identity: from the launcher's world plan (one process per rank, started by runtime/launcher.py), or
    # from torch.distributed.run's RANK / WORLD_SIZE when the ranks were started that way
    from ..parallel.world import join_world, world_from_env
    spec = world_from_env()
    if spec is not None:
        rank, world = int(os.environ["DRL_WORLD_RANK"]), int(spec["learners"])
    else:
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        if world > 1 and device.type == "cuda" and not args.device:
            lr_ = int(os.environ.get("LOCAL_RANK", "0"))
            if os.environ.get("DRL_SHARE_GPU") == "1":         # rehearsal: more ranks than GPUs (see parallel/dist.py)
                lr_ %= torch.cuda.device_count()
            device = torch.device("cuda", lr_)
    metrics = make_metrics("learner" if rank == 0 else "learner_r%d" % rank, args.visdomserver, args.log_dir,
                           args.quiet or rank > 0)
    exp, par = connect(args.transport, shard_run_id(args.run_id, rank, world), create=not args.attach, host=args.redisserver,
                       n_slots=args.ring_slots, slot_bytes=args.ring_slot_mb << 20)
    register = None
    dist = None
    pg_world = join_world(device) if spec is not None else None
    if pg_world is not None:
        from ..parallel.dist import Dist
        if world > 1:
            dist = Dist(device=device, group=pg_world.learner_group)
        if pg_world.my_lanes():
            register = par           # stays open either way: actors attach to both segments of the transport first
            par = _learner_fanout(pg_world, register, _flat_numel(args.algorithm, n_action), device,
                                  keep_register=args.register_actors != 0)
    elif world > 1:
        from ..parallel.dist import Dist
        dist = Dist(device=device)
    hb_table, hb = None, None
    if args.transport == 'shm':
        from .heartbeat import HeartbeatTable
        try:
            hb_table = HeartbeatTable(shm_names(args.run_id)[2], create=False)
        except OSError:
            hb_table = HeartbeatTable(shm_names(args.run_id)[2], create=True)
        hb = hb_table.register("learner", rank)
    engine = args.engine
    if engine == 'auto':
        engine = 'hip' if device.type == 'cuda' else 'torch'
    try:
        if engine == 'hip':
            from ..engine.learner import EXIT_GUARD, run_hip_learner
            rc = run_hip_learner(args, cfg, n_action, exp, par, metrics, hb, device, dist=dist)
            if rc == EXIT_GUARD:
                # the launcher starts a fresh learner that ATTACHES to these segments (actors keep running): leave them in place
                for ch in (exp, par, register):
                    if ch is not None and hasattr(ch, "_owner"):
                        ch._owner = False
            return rc
        from .learner import Learner, load_checkpoint
        if args.algorithm == 'ape_x':
            model, target = DuelingDQN(n_action).to(device), DuelingDQN(n_action).to(device)
            opt = optim.RMSprop(model.parameters(), lr=cfg.lr, alpha=cfg.rms_alpha, eps=cfg.rms_eps)     # learner_node.py:28
        else:
            model = DuelingLSTMDQN(n_action, cfg.batch_size, cfg.burn_in, cfg.nstep).to(device)
            target = DuelingLSTMDQN(n_action, cfg.batch_size, cfg.burn_in, cfg.nstep).to(device)
            opt = optim.Adam(model.parameters(), lr=cfg.lr, eps=cfg.adam_eps)                             # learner_node.py:40
        start = load_checkpoint(args.resume, model, target, opt) if args.resume else 0
        if dist is not None:
            with torch.no_grad():                          # identical initial weights on every rank
                for p_ in model.parameters():
                    dist.broadcast_flat(p_.data)
                target.load_state_dict(model.state_dict())
        learner = Learner(model, target, opt, cfg, exp, par, metrics, args.replaysize, device, heartbeat=hb,
                          sync_target=not args.resume, dist=dist, reset_channels=not args.attach)
        learner.publish()
        try:
            learner.optimize_loop(max_steps=args.max_steps, save_model_dir=args.save_dir, start_step=start)
        except KeyboardInterrupt:
            pass
        except RuntimeError as e:
            if "discarded" not in str(e):
                raise
            # same contract as the GPU learner (engine/learner.py _guard_exit): state out, note, EXIT_GUARD; the segments stay
            import json
            from ..engine.learner import EXIT_GUARD, FAULT_NOTE
            from .learner import save_checkpoint
            print("[learner] %s" % e, file=sys.stderr, flush=True)
            ck = save_checkpoint(args.save_dir, learner.step, model, target, opt)
            with open(os.path.join(args.save_dir, FAULT_NOTE), "w") as fh:
                json.dump({"step": learner.step, "error": str(e)[:500], "checkpoint": ck}, fh)
            for ch in (exp, par, register):
                if ch is not None and hasattr(ch, "_owner"):
                    ch._owner = False
            return EXIT_GUARD
        finally:
            learner.close()
    finally:
        if hb is not None:
            hb.close()
        exp.close()
        par.close()
        if register is not None and getattr(par, "register", None) is not register:
            register.close()
        if hb_table is not None:
            hb_table.close()
        if dist is not None:
            dist.close()
        if pg_world is not None:
            pg_world.close()
    return 0


if __name__ == '__main__':
    sys.exit(learner_main() if 'learner' in sys.argv[0] else actor_main())
