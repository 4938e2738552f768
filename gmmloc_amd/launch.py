"""One process per GPU: how bench.py and tools/replay_euroc.py become N ranks and meet.

`--gpus N` with no rank environment means "start N ranks": the script re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` and relays the ranks' output
(rank 0 prints the one JSON line).  Launched by torchrun already (RANK / WORLD_SIZE set) it is a rank and just runs.
The backend is "nccl" (= RCCL over xGMI) on GPUs; "gloo" exists for the CPU tests of this plumbing only.
"""
import os
import socket
import subprocess
import sys


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


RENDEZVOUS_RETRY_S = 20.0  # spawn_ranks: a launcher that fails sooner than this is started again (port taken between probe and bind)


def is_rank():
    """True when a launcher (torchrun) already made this process one rank of a job."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def spawn_ranks(n, script, argv, need_gpus=True):
    """Start `n` ranks of `script argv` on this node and return the launcher's exit code.  Fails loudly when the node
    has fewer GPUs than ranks: a line that says n_gpus = N must have used N."""
    if need_gpus:
        import torch
        have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write("%s: --gpus %d asked for, %d GPU(s) visible on this node\n" % (os.path.basename(script), n, have))
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    # free_port() closes its socket before the launcher binds the port: another process can take it in between.  A launch that
    # dies inside the first seconds (the rendezvous) is tried again on another port; one that ran longer failed for its own reasons.
    import time
    rc = 1
    for attempt in range(3):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script] + list(argv)
        t0 = time.monotonic()
        rc = subprocess.call(cmd, env=env)
        if rc == 0 or time.monotonic() - t0 > RENDEZVOUS_RETRY_S:
            break
        sys.stderr.write("%s: the launcher exited with %d after %.1f s (attempt %d of 3)\n" % (os.path.basename(script), rc, time.monotonic() - t0, attempt + 1))
    return rc


class Ranks:
    """The rank's view of the job: rank / world / local from the environment, the process group, barrier and MAX."""

    def __init__(self, backend="nccl"):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.backend = backend
        self.dist = None
        self.device = None
        self.last_own_dt = None  # timed_steps: this rank's own elapsed seconds (the barrier-to-barrier time is the MAX over ranks)
        self.device_sync = None  # gloo stub only: stand-in for the device synchronise (tests of the timing with an asynchronous step)

    def init(self, always=False):
        """Join the process group (world > 1, or always=True to run the collectives with one rank too)."""
        import torch
        if self.backend == "nccl":
            assert torch.cuda.is_available(), "the HIP path needs a GPU (no CPU fallback exists)"
            torch.cuda.set_device(self.local)
            self.device = torch.device("cuda", self.local)
        else:
            self.device = torch.device("cpu")
        if self.world > 1 or always:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            kw = {"device_id": self.device} if self.backend == "nccl" else {}
            dist.init_process_group(self.backend, rank=self.rank, world_size=self.world, **kw)
            self.dist = dist
        return self

    def sync(self):
        """wait for everything this rank has enqueued on its device (steps are asynchronous launches)"""
        if self.backend == "nccl":
            import torch
            torch.cuda.synchronize()
        elif self.device_sync is not None:
            self.device_sync()

    def barrier(self):
        self.sync()
        if self.dist is not None:
            self.dist.barrier()
        self.sync()

    def max(self, value):
        """MAX over the ranks of one float."""
        import torch
        t = torch.tensor([value], dtype=torch.float64, device=self.device)
        if self.dist is not None:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, value):
        import torch
        t = torch.tensor([value], dtype=torch.float64, device=self.device)
        if self.dist is not None:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def gather(self, value):
        """every rank's float, in rank order (all_gather)"""
        import torch
        t = torch.tensor([value], dtype=torch.float64, device=self.device)
        if self.dist is None:
            return [float(t.item())]
        out = [torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [float(o.item()) for o in out]

    def group_size(self):
        """ranks the process group really has (what RCCL / gloo report, not what the command line said)"""
        return int(self.dist.get_world_size()) if self.dist is not None else 1

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None


def timed_steps(step, steps, warmup, ranks, before_timed=None):
    """The bench contract: `warmup` untimed steps, then EXACTLY `steps` steps bracketed by barrier + synchronize on both
    sides; returns the MAX over ranks of the elapsed seconds."""
    import time
    for _ in range(warmup):
        step()
    if before_timed:
        before_timed()
    ranks.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ranks.sync()                    # the steps only ENQUEUE work: the rank's own time ends when its device is done,
    own = time.perf_counter() - t0  # ... and before it waits for the other ranks
    ranks.barrier()
    ranks.last_own_dt = own
    return ranks.max(time.perf_counter() - t0)
