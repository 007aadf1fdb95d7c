"""Device plumbing of bench.py: the rank launcher and the hip / emu back-ends (process group + start-up self-check)."""
import os
import sys

import torch

from benchlib.common import ROOT
from spotlight_amd import _native


def spawn_ranks(args):
    """`python bench.py --gpus N` without a torchrun environment: launch the N ranks ourselves (one process
    per GPU, torch.distributed.run on 127.0.0.1) and pass rank 0's JSON line through.  Fails loudly when the
    machine does not have N GPUs -- it never degrades to fewer ranks."""
    import socket
    import subprocess
    if args.backend == 'hip':
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            sys.stderr.write('bench.py: --gpus %d requested but %d HIP device(s) visible; refusing to run fewer ranks\n'
                             % (args.gpus, have))
            return 3
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'bench.py')] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', SLK_BENCH_SPAWNED='1')
    env.setdefault('OMP_NUM_THREADS', '4')
    return subprocess.call(cmd, env=env)


class Backend(object):
    """Device plumbing of the benchmark: 'hip' = torch-ROCm tensors + RCCL + libspotlight_hip.so (the product);
    'emu' = CPU tensors + gloo + tests/emu's build of the same kernels (test harness for the launch logic)."""

    def __init__(self, kind, local_rank):
        self.kind = kind
        if kind == 'hip':
            torch.cuda.set_device(local_rank)
            self.dev = torch.device('cuda', local_rank)
            self.engine = _native.Engine(local_rank)
            self.dist_backend = 'nccl'
            self.name = torch.cuda.get_device_name(local_rank)
        else:
            sys.path.insert(0, os.path.join(ROOT, 'tests'))
            from emu_backend import emu_lib
            self.dev = torch.device('cpu')
            self.engine = _native.Engine(0, lib=emu_lib())
            self.dist_backend = 'gloo'
            self.name = 'cpu emulator (test harness)'
        self.side = None

    def init_dist(self, rank, world, local_rank):
        """Process group + a self-check of everything the first multi-GPU run could trip over, BEFORE any table is allocated:
        every failure names the rank, the device and the variable to look at, and exits non-zero within the time-out instead of
        hanging (the driver's 8-GPU run is the first hardware run of this path: it must not be lost to a launcher problem)."""
        import datetime
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29400')
        who = 'bench.py rank %d/%d (local rank %d)' % (rank, world, local_rank)

        def die(code, msg):
            sys.stderr.write('%s: %s\n' % (who, msg))
            sys.stderr.flush()
            os._exit(code)
        timeout = datetime.timedelta(seconds=int(os.environ.get('SLK_BENCH_DIST_TIMEOUT', '180')))
        try:
            if self.kind == 'hip':
                if os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0') != '0':
                    die(4, 'HSA_ENABLE_IPC_MODE_LEGACY=%s: this host driver only supports dmabuf IPC; RCCL needs it unset or 0'
                        % os.environ['HSA_ENABLE_IPC_MODE_LEGACY'])
                if torch.cuda.device_count() <= local_rank:
                    die(4, 'LOCAL_RANK %d but only %d HIP device(s) visible (HIP_VISIBLE_DEVICES=%s)'
                        % (local_rank, torch.cuda.device_count(), os.environ.get('HIP_VISIBLE_DEVICES')))
                os.environ.setdefault('TORCH_NCCL_ASYNC_ERROR_HANDLING', '1')  # a failed collective raises instead of hanging
                dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank),
                                        timeout=timeout)
            else:
                dist.init_process_group('gloo', rank=rank, world_size=world, timeout=timeout)
        except SystemExit:
            raise
        except Exception as e:  # noqa: BLE001 -- rendezvous / RCCL initialisation
            die(4, 'init_process_group failed: %r (MASTER_ADDR=%s MASTER_PORT=%s WORLD_SIZE=%s)'
                % (e, os.environ.get('MASTER_ADDR'), os.environ.get('MASTER_PORT'), os.environ.get('WORLD_SIZE')))
        try:
            # (1) every rank sits on its own device
            ident = 'cpu:%d' % rank
            if self.kind == 'hip':
                props = torch.cuda.get_device_properties(local_rank)
                ident = '%s/%s' % (getattr(props, 'uuid', None) or props.name, local_rank)
            idents = [None] * world
            dist.all_gather_object(idents, ident)
            if self.kind == 'hip' and len(set(idents)) != world:
                die(5, 'two ranks share a device: %s' % idents)
            # (2) the collectives the sharded path issues, at their smallest: all_reduce, all_to_all_single with uneven splits
            x = torch.full((4,), float(rank + 1), device=self.dev)
            dist.all_reduce(x)
            want = world * (world + 1) / 2.0
            if abs(float(x[0].item()) - want) > 1e-3:
                die(5, 'all_reduce returned %r, expected %r' % (float(x[0].item()), want))
            send_counts = [(rank + p) % 3 + 1 for p in range(world)]
            recv_counts = [(p + rank) % 3 + 1 for p in range(world)]
            send = torch.cat([torch.full((c,), float(rank * 100 + p), device=self.dev) for p, c in enumerate(send_counts)])
            recv = torch.empty(sum(recv_counts), device=self.dev)
            dist.all_to_all_single(recv, send, recv_counts, send_counts)
            got = recv.cpu().tolist()
            exp = [float(p * 100 + rank) for p, c in enumerate(recv_counts) for _ in range(c)]
            if got != exp:
                die(5, 'all_to_all_single with uneven splits returned %r, expected %r' % (got, exp))
        except SystemExit:
            raise
        except Exception as e:  # noqa: BLE001
            die(5, 'collective self-check failed: %r' % (e,))
        return dist

    def generator(self, seed):
        gen = torch.Generator(device=self.dev)
        gen.manual_seed(seed)
        return gen

    def use_side_stream(self):
        if self.kind == 'hip':
            self.side = torch.cuda.Stream(self.dev)
            self.side.wait_stream(torch.cuda.current_stream(self.dev))
            torch.cuda.set_stream(self.side)

    def stream(self):
        return torch.cuda.current_stream(self.dev).cuda_stream if self.kind == 'hip' else 0

    def sync(self):
        if self.kind == 'hip':
            torch.cuda.synchronize(self.dev)
