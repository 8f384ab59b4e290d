"""Single-node data parallelism over RCCL/xGMI (one process per GPU; reference: DistributedDataParallel wrap at
training/sg_trainer/sg_trainer.py:452-459, NCCL init at training/utils/distributed_training_utils.py:289-311).

MI355X-first design instead of DDP's per-parameter autograd hooks + 25 MB buckets:
  * gradients already live in ONE flat fp32 arena in reverse-backward order, so a "bucket" is a contiguous arena range
    owned by a sub-network (heads / neck / each backbone stage); when that sub-network's backward kernels have been
    enqueued, its range is all-reduced with ONE RCCL call (async: RCCL's stream waits for the compute stream at that
    point, the remaining backward keeps the CUs busy).  4-7 large collectives per step instead of hundreds of small ones -
    xGMI is point-to-point, per-link bound, so few large messages is the right shape;
  * the 32% dead parameters (QARepVGGBlock.rbr_reparam) are not in the arena: they are never communicated
    (DDP needs find_unused_parameters=True and still carries them in its buckets);
  * the mean over ranks is folded into the optimizer kernel (grad_scale = 1/world), no extra pass over the arena;
  * BN running statistics stay per-rank (the reference benchmark runs with SyncBN off); rank 0's are checkpointed.
"""
import os
from typing import List, Tuple

import torch
import torch.distributed as dist


def collectives_active() -> bool:
    """A process group exists and its collectives are to be issued: more than one rank - or ONE rank with SGX_DIST_SINGLE_RANK_COLLECTIVES=1,
    the switch of tests/test_distributed.py::test_rccl_single_rank_communicator: the whole data-parallel choreography (RCCL communicator,
    bucket all-reduces issued from the side stream, the loss's 16-byte all-reduce, synchronised BatchNorm, buffer broadcasts) on the ONE
    MI355X a test box has - every collective is then an identity, so the step must equal the non-distributed one."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("SGX_DIST_SINGLE_RANK_COLLECTIVES") == "1"


def is_distributed() -> bool:
    return collectives_active()


def get_world_size() -> int:
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def get_rank() -> int:
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def setup_device_from_env(backend: str = None) -> Tuple[int, int, torch.device]:
    """env:// rendezvous as launched by `python -m torch.distributed.run` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    Returns (rank, world_size, device).  backend defaults to nccl (= RCCL on ROCm) when a GPU is present, gloo otherwise."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    has_gpu = torch.cuda.is_available()
    if has_gpu:
        torch.cuda.set_device(local)
    device = torch.device(f"cuda:{local}") if has_gpu else torch.device("cpu")
    if (world > 1 or os.environ.get("SGX_DIST_SINGLE_RANK_COLLECTIVES") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend or ("nccl" if has_gpu else "gloo"), init_method="env://", rank=rank, world_size=world)
    return rank, world, device


def barrier():
    """dist.barrier() that names this process's GPU for the RCCL backend (without device_ids torch guesses the device from the rank and
    warns that a wrong guess can hang); a no-op without a process group."""
    if not is_distributed():
        return
    if dist.get_backend() == "nccl":
        dist.barrier(device_ids=[torch.cuda.current_device()])
    else:
        dist.barrier()


def setup_device(multi_gpu=None, num_gpus: int = None, device: str = "cuda"):
    """Reference: training/utils/distributed_training_utils.py:229-286.  On the MI355X path the process model is fixed: one
    process per GPU, launched by `python -m torch.distributed.run` (env:// rendezvous); this call joins the process group
    when WORLD_SIZE > 1 and selects the local GPU.  `multi_gpu="DDP"`/`num_gpus` are validated against the launch."""
    if device not in ("cuda", None):
        raise ValueError("the MI355X path runs on the HIP device only (device='cuda')")
    rank, world, dev = setup_device_from_env()
    if num_gpus not in (None, -1) and int(num_gpus) != world:
        raise ValueError(f"num_gpus={num_gpus} but the launch provides WORLD_SIZE={world}: start one process per GPU with "
                         "`python -m torch.distributed.run --nproc-per-node N`")
    return rank, world, dev


class GradientAllReducer:
    """Bucketed, backward-overlapped all-reduce of a network's gradient arena."""

    def __init__(self, net, bucket_prefixes: List[str]):
        self.net = net
        self.world = get_world_size()
        self.ranges = {}
        slots = net.slots
        for pref in bucket_prefixes:
            idx = [i for i, s in enumerate(slots) if s.name.startswith(pref)]
            if not idx:
                continue
            start = slots[idx[0]].start
            end = slots[idx[-1] + 1].start if idx[-1] + 1 < len(slots) else net.g_arena.size
            self.ranges[pref] = (start, end)
        covered = sorted(self.ranges.values())
        pos = 0
        for a, b in covered:
            if a != pos:
                raise RuntimeError("gradient buckets must tile the arena")
            pos = b
        if pos != net.g_arena.size and covered:
            raise RuntimeError("gradient buckets must tile the arena")
        self.pending = []
        self.issued = set()
        self.from_side = os.environ.get("SGX_ALLREDUCE_FROM_SIDE", "1") != "0"  # 0: join the side stream into the current one per bucket
        self.grad_scale = torch.full((1,), 1.0 / self.world, device=net.g_arena.buf.device)
        # False on the micro-batches of a gradient accumulation that do not end in an optimizer step (DistributedDataParallel.no_sync()
        # semantics): the arena keeps accumulating locally and is exchanged once, by the backward of the stepping micro-batch
        self.sync = True
        net._grad_ready = self.ready
        net._post_backward_hook = self.finish
        # RCCL's stream is one more HIP stream beside main + side + the branch-stream lanes, and the runtime has four hardware queues: with two
        # lanes the collectives cost 6.3 % of the step on the one-rank communicator, with one lane 0.8 % (r6ag, r6ah)
        if collectives_active() and hasattr(net, "data_parallel_streams"):
            net.data_parallel_streams()

    # (seams for the CPU tests, which have no HIP streams: tests/test_distributed.py substitutes recording stand-ins)
    @staticmethod
    def _current_stream():
        return torch.cuda.current_stream()

    @staticmethod
    def _on_stream(stream):
        return torch.cuda.stream(stream)

    def _side_stream(self):
        return getattr(self.net, "side_stream", None)

    def ready(self, prefix: str):
        """Called by the network's backward right after the kernels of sub-network `prefix` have been enqueued."""
        if not collectives_active() or not self.sync or prefix not in self.ranges or prefix in self.issued:
            return
        a, b = self.ranges[prefix]
        self.issued.add(prefix)
        buf = self.net.g_arena.buf[a:b]
        side = self._side_stream()
        if side is None or not self.from_side:
            self.net.join_side()  # the bucket's weight gradients were forked onto the side stream
            self.pending.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True))
            return
        # The bucket's weight gradients were written on the side stream, its BatchNorm / bias gradients on the current one.  Joining the
        # side stream into the current one here would stall the data-gradient chain (everything the rest of backward depends on) behind
        # the weight-gradient backlog at every bucket boundary - an overlap the single-GPU step keeps until the end of backward.  Instead
        # the SIDE stream waits for the current one and issues the collective: RCCL's stream then waits for both producers, later
        # weight gradients (other arena ranges) keep flowing on the side stream beside the collective, and the current stream waits for
        # nothing until finish().
        side.wait_stream(self._current_stream())
        with self._on_stream(side):
            self.pending.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        """End of backward: exchange whatever the network did not announce itself (a network that never calls `ready` still
        gets a correct, just un-overlapped, all-reduce), then make the compute stream wait for the collectives."""
        for prefix in self.ranges:
            self.ready(prefix)
        for w in self.pending:
            w.wait()  # makes the compute stream wait for RCCL's stream; no host block for NCCL/RCCL work objects
        self.pending = []
        self.issued = set()

    def broadcast_buffers(self, src: int = 0):
        """What DistributedDataParallel(broadcast_buffers=True) - the reference's wrapping (sg_trainer.py:1352-1357, torch default) - does
        at the start of every training forward: BatchNorm running statistics and step counters of every rank are overwritten with rank
        `src`'s.  Two small collectives over the buffer arenas."""
        if collectives_active():
            dist.broadcast(self.net.b_arena.buf, src)
            dist.broadcast(self.net.i_arena, src)

    def broadcast_parameters(self, src: int = 0):
        """DDP-constructor semantics: every rank starts from rank `src`'s parameters and buffers."""
        if collectives_active():
            dist.broadcast(self.net.p_arena.buf, src)
            dist.broadcast(self.net.b_arena.buf, src)
