"""Multi-GPU sharding of a factor table: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference has no multi-GPU code at all; this is new.  Factors are independent units, so the
table is partitioned across ranks with no data-path collective; the one exchange step is the north-star's "all-reduce of
the stacked 6x6 H blocks": every rank writes its factors' records into its slots of a zero-initialised stacked
[F_total x 122] f64 buffer and ONE all_reduce(SUM) leaves every rank (and, after one D2H, its host optimizer) with all
records.  Each slot is written by exactly one rank, so the sum is exact (x + 0 + ... + 0) and order-independent.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): the stack is tiny (4096 factors -> 4.0 MB f64), so the collective
is latency-bound; it is issued once per linearise on the compute stream's successor, never per factor.
"""
import numpy as np

RECORD_DOUBLES = 122  # gp_linearized6


def partition_factors(weights, world_size):
    """Contiguous partition of the factor list into `world_size` ranges balanced by sum(weights) (= source points per
    factor).  Contiguity keeps factors that share a source cloud / target map on one rank when the list is ordered by
    submap.  Returns [(begin, end)] per rank; ranges may be empty when there are fewer factors than ranks."""
    w = np.asarray(weights, dtype=np.float64)
    n = len(w)
    if world_size <= 0:
        raise ValueError("world_size must be positive")
    cum = np.concatenate([[0.0], np.cumsum(w)])
    total = cum[-1]
    bounds = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        # first index whose prefix reaches the target, but never before the previous bound
        i = int(np.searchsorted(cum, target, side="left"))
        if i > 0 and i <= n and abs(cum[i - 1] - target) <= abs(cum[min(i, n)] - target):
            i -= 1  # the boundary nearest to the ideal split
        i = min(max(i, bounds[-1]), n)
        bounds.append(i)
    bounds.append(n)
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


class ShardedLinearizer:
    """Drives one rank's shard of a global factor table.

      issue(poses_local, out_view)  -- computes this rank's records into `out_view`, a [F_local x 122] f64 view of the
                                       stacked buffer (on GPUs: gp_vgicp_batch_issue_linearize with out_dev = view pointer)
    """

    def __init__(self, total_factors, slot_range, device, issue, group=None):
        import torch

        self.total = int(total_factors)
        self.begin, self.end = int(slot_range[0]), int(slot_range[1])
        self.issue = issue
        self.group = group
        self.stacked = torch.zeros((self.total, RECORD_DOUBLES), dtype=torch.float64, device=device)

    def linearize(self, poses_local):
        """Returns the stacked [F_total x 122] tensor holding every rank's records (device-resident)."""
        import torch.distributed as dist

        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world > 1:
            self.stacked.zero_()
        if self.end > self.begin:
            self.issue(poses_local, self.stacked[self.begin : self.end])
        if world > 1:
            dist.all_reduce(self.stacked, op=dist.ReduceOp.SUM, group=self.group)
        return self.stacked
