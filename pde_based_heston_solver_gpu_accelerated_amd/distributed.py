"""Multi-GPU layer: one process per GPU, instances sharded, no collective on the data path.

The reference is single-device; its only reduction is J^T J / J^T r of the Levenberg-Marquardt step
(KokkosBlas gemm/gemv, src/jacobian_computation.cpp:117,154).  Option instances are independent for
the whole sweep (TeamPolicy league, src/device_solver.hpp:83-88), so every rank solves a contiguous
block of the instance list on its own GPU and the LM step all-reduces 31 doubles
(25 J^T J + 5 J^T r + 1 sum r^2) over RCCL (`backend="nccl"` on ROCm) -- a latency-bound message, so
one small all-reduce, nothing bandwidth-tuned.  Tests drive the same code over gloo on CPU.
"""
import numpy as np


def shard_range(n, world_size, rank, costs=None):
    """Contiguous block [lo, hi) of n instances for `rank`.  With `costs` (per-instance work, e.g.
    (m1+1)(m2+1)*N_k for multi-maturity batches, heston_calibration.cpp:2517) the cut points balance
    the cumulative cost; an option's rows stay on one rank so its J row is formed locally."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    if n < 0:
        raise ValueError("n < 0")
    if costs is None:
        base, rem = divmod(n, world_size)
        lo = rank * base + min(rank, rem)
        return lo, lo + base + (1 if rank < rem else 0)
    c = np.asarray(costs, dtype=np.float64)
    if c.shape != (n,) or (c < 0).any():
        raise ValueError("costs must be n non-negative numbers")
    cum = np.concatenate([[0.0], np.cumsum(c)])
    total = cum[-1]
    cuts = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        k = int(np.searchsorted(cum, target, side="left"))
        # pick the cut whose prefix cost is nearest to the target
        if k > 0 and abs(cum[k - 1] - target) <= abs(cum[min(k, n)] - target):
            k -= 1
        cuts.append(min(max(k, cuts[-1]), n))
    cuts.append(n)
    return cuts[rank], cuts[rank + 1]


class Communicator:
    """Thin wrapper over torch.distributed (RCCL on GPUs, gloo on CPU); world_size 1 needs no init."""

    def __init__(self, device=None, group=None):
        """`group`: the process group the collectives run on (None = the default group).
        `device`: where the collective's tensors live.  None = chosen from the process group's backend: RCCL
        (`nccl`) only moves GPU memory, so the tensors go to this process's current CUDA device (one process per GPU:
        the launcher has called torch.cuda.set_device(LOCAL_RANK)); gloo takes CPU tensors."""
        self.dist = None
        self.rank, self.world_size = 0, 1
        self.device = device
        self.group = group
        self.backend = None
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                self.dist = dist
                self.rank, self.world_size = dist.get_rank(), dist.get_world_size()
                self.backend = str(dist.get_backend(group))
                if self.device is None and "nccl" in self.backend:
                    import torch
                    self.device = torch.device("cuda", torch.cuda.current_device())
        except ImportError:
            pass

    def allreduce_sum(self, vec):
        """Sum of a small float64 vector over all ranks (returns a numpy array)."""
        v = np.ascontiguousarray(np.asarray(vec, dtype=np.float64))
        if self.dist is None or self.world_size == 1:
            return v.copy()
        import torch
        t = torch.from_numpy(v.copy())
        if self.device is not None:
            t = t.to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def allgather_rows(self, local, counts):
        """Concatenates per-rank row blocks (e.g. base_prices for reporting)."""
        local = np.ascontiguousarray(np.asarray(local, dtype=np.float64))
        if self.dist is None or self.world_size == 1:
            return local.copy()
        import torch
        width = int(np.prod(local.shape[1:])) if local.ndim > 1 else 1
        maxc = max(counts)
        buf = np.zeros((maxc, width))
        buf[:local.shape[0]] = local.reshape(local.shape[0], width)
        t = torch.from_numpy(buf)
        if self.device is not None:
            t = t.to(self.device)
        outs = [torch.empty_like(t) for _ in range(self.world_size)]
        self.dist.all_gather(outs, t, group=self.group)
        parts = [o.cpu().numpy()[:c] for o, c in zip(outs, counts)]
        out = np.concatenate(parts, axis=0)
        return out.reshape((-1,) + local.shape[1:]) if local.ndim > 1 else out.reshape(-1)

    def barrier(self):
        if self.dist is not None and self.world_size > 1:
            self.dist.barrier(group=self.group)
