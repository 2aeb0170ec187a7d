"""Multi-GPU sharding of a factor table: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference has no multi-GPU code at all; this is new.  Factors are independent units, so the
table is partitioned across ranks with no data-path collective; the one exchange step is the north-star's "all-reduce of
the stacked 6x6 H blocks": every rank writes its factors' records into its slots of a zero-initialised stacked
[F_total x 122] f64 buffer and ONE all_reduce(SUM) leaves every rank (and, after one D2H, its host optimizer) with all
records.  Each slot is written by exactly one rank, so the sum is exact (x + 0 + ... + 0) and order-independent.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): the stack is tiny (4096 factors -> 4.0 MB f64), so the collective
is latency-bound; it is issued once per linearise on the compute stream's successor, never per factor.

Round 5: exchange="peer" -- for small stacks (the headline: one record per GPU) every rank stores its rows straight into every peer's buffer over xGMI and flags
its arrival (gp_peer_exchange_*, csrc/gp_peer.hip): one single-workgroup kernel per step instead of a collective library's ring.  The buffers' IPC handles travel
once through torch.distributed; the form is validated against the all-gather before it is used and every rank falls back together when anything fails.
"""
import numpy as np

RECORD_DOUBLES = 122  # gp_linearized6


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


def partition_factors(weights, world_size):
    """Contiguous partition of the factor list into `world_size` ranges minimising the largest sum(weights) (= source points
    per shard) -- gp_shard_plan_create of the C-ABI, the same plan the single-process multi-GPU batch uses.  Contiguity keeps
    factors that share a source cloud / target map on one rank when the list is ordered by submap.
    Returns [(begin, end)] per rank; ranges may be empty when there are fewer factors than ranks.  Pure host code (no GPU)."""
    import ctypes as C

    from . import _capi

    if world_size <= 0:
        raise ValueError("world_size must be positive")
    lib = _capi.load()
    w = np.ascontiguousarray(weights, dtype=np.int64)
    plan = C.c_void_p()
    _capi.check(lib.gp_shard_plan_create(C.c_void_p(w.ctypes.data) if len(w) else None, len(w), int(world_size), C.byref(plan)), "gp_shard_plan_create")
    out = []
    for r in range(world_size):
        b, e = C.c_int(), C.c_int()
        _capi.check(lib.gp_shard_plan_range(plan, r, C.byref(b), C.byref(e)), "gp_shard_plan_range")
        out.append((b.value, e.value))
    lib.gp_shard_plan_destroy(plan)
    return out


def record_digests(rows):
    """SHA-256 of every [122] f64 record of `rows` ([n x 122] host array): what the ranks tell one another about the records they computed"""
    import hashlib

    rows = np.ascontiguousarray(rows, dtype=np.float64)
    return [hashlib.sha256(rows[i].tobytes()).hexdigest() for i in range(rows.shape[0])]


def verify_exchanged_stack(stack_host, own_rows_host, begin, end, group=None):
    """Is the exchanged stack, on EVERY rank, bit for bit what the ranks computed?  Collective (every rank calls it; works on any backend: objects only).
      stack_host     [F_total x 122] f64 host array: the stack this rank holds after the exchange
      own_rows_host  [(end - begin) x 122] f64: the records this rank computed by itself, NOT taken from the stack
    Every rank publishes the SHA-256 of its own records (all_gather_object), checks all rows of its stack against the digests of the rank that owns them, and the
    verdicts are gathered again so that every rank returns the same answer: (verified, {rank: [bad global rows]} for the ranks that found any).
    A row nobody claims, or claimed twice, fails.  Without an initialised process group the check is local (one rank)."""
    import torch.distributed as dist

    stack = np.ascontiguousarray(stack_host, dtype=np.float64)
    mine = (int(begin), int(end), record_digests(own_rows_host))
    if int(end) - int(begin) != len(mine[2]):
        raise ValueError("verify_exchanged_stack: own_rows_host must hold end - begin records")
    up = dist.is_available() and dist.is_initialized()
    if up:
        claims = [None] * dist.get_world_size(group)
        dist.all_gather_object(claims, mine, group=group)
        me = dist.get_rank(group)
    else:
        claims, me = [mine], 0
    want = [None] * stack.shape[0]
    bad = set()
    for b, e, digests in claims:
        for k, dg in zip(range(b, e), digests):
            if not (0 <= k < len(want)) or want[k] is not None:
                bad.add(min(max(k, 0), len(want) - 1))  # out of range, or two owners
            else:
                want[k] = dg
    got = record_digests(stack)
    bad.update(k for k in range(len(want)) if want[k] is None or want[k] != got[k])
    verdict = sorted(bad)
    if up:
        verdicts = [None] * len(claims)
        dist.all_gather_object(verdicts, verdict, group=group)
    else:
        verdicts = [verdict]
    by_rank = {r: v for r, v in enumerate(verdicts) if v}
    del me
    return (not by_rank), by_rank


class MultiDeviceBatch:
    """gp_vgicp_multi_batch_*: a factor list sharded over the GPUs of one node from ONE process (the form a C++ optimizer
    process uses; `ShardedLinearizer` below is the one-process-per-GPU form bench.py is launched in).
      factors        IntegratedVGICPFactorGPU objects, each created from arrays / a map on the device of its shard
      shard_of       None = one shard per device; or a list of shard indices (several shards may share a device)
      use_rccl       0 no collective (every shard's finalize kernel stores its records into its rows of one host-pinned stack), 1 ncclAllReduce of the zeroed
                     stack (required), 2 in-place ncclAllGather when the shards are equal contiguous ranges in rank order (else the all-reduce), -1 automatic"""

    def __init__(self, factors, shard_of=None, num_shards=0, use_rccl=-1):
        import ctypes as C

        from . import _capi

        self._lib = _capi.load()
        self._factors = list(factors)  # keep them alive
        F = len(self._factors)
        arr = (C.c_void_p * max(F, 1))(*[f._h.value for f in self._factors])
        so = None
        if shard_of is not None:
            so = np.ascontiguousarray(shard_of, dtype=np.int32)
            assert len(so) == F
        self._h = C.c_void_p()
        _capi.check(self._lib.gp_vgicp_multi_batch_create(arr, F, C.c_void_p(so.ctypes.data) if so is not None else None, int(num_shards), int(use_rccl), C.byref(self._h)),
                    "gp_vgicp_multi_batch_create")
        self.size = F

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.gp_vgicp_multi_batch_destroy(self._h)
            self._h = None

    @property
    def num_shards(self):
        return int(self._lib.gp_vgicp_multi_batch_num_shards(self._h))

    @property
    def uses_rccl(self):
        return bool(self._lib.gp_vgicp_multi_batch_uses_rccl(self._h))

    @property
    def exchange(self):
        """which exchange a pass runs: "none" (records straight into the host stack), "all_reduce", "all_gather" """
        return ("none", "all_reduce", "all_gather")[int(self._lib.gp_vgicp_multi_batch_uses_rccl(self._h))]

    def shard_info(self, shard):
        import ctypes as C

        from . import _capi

        d, n, p = C.c_int(), C.c_int(), C.c_int64()
        _capi.check(self._lib.gp_vgicp_multi_batch_shard_info(self._h, shard, C.byref(d), C.byref(n), C.byref(p)), "gp_vgicp_multi_batch_shard_info")
        return dict(device=d.value, num_factors=n.value, num_points=p.value)

    def linearize(self, deltas):
        """deltas: F 4x4 poses (T_target^-1 T_source) -> [F x 122] f64 records (gp_linearized6 layout)"""
        from . import _capi

        poses = np.stack([np.ascontiguousarray(np.asarray(d, dtype=np.float64).T).reshape(16) for d in deltas]).copy() if self.size else np.zeros((0, 16))
        out = np.zeros((self.size, RECORD_DOUBLES))
        _capi.check(self._lib.gp_vgicp_multi_batch_linearize(self._h, poses.ctypes.data, out.ctypes.data), "gp_vgicp_multi_batch_linearize")
        return out

    def linearize_flat(self, poses_flat, out):
        """the call without per-call staging: poses_flat [F x 16] f64 (each pose column-major), out [F x 122] f64, both C-contiguous numpy arrays"""
        from . import _capi

        _capi.check(self._lib.gp_vgicp_multi_batch_linearize(self._h, poses_flat.ctypes.data, out.ctypes.data), "gp_vgicp_multi_batch_linearize")
        return out

    def compute_error(self, deltas_lin, deltas_eval):
        from . import _capi

        pl = np.stack([np.ascontiguousarray(np.asarray(d, dtype=np.float64).T).reshape(16) for d in deltas_lin]).copy()
        pe = np.stack([np.ascontiguousarray(np.asarray(d, dtype=np.float64).T).reshape(16) for d in deltas_eval]).copy()
        out = np.zeros(self.size)
        _capi.check(self._lib.gp_vgicp_multi_batch_compute_error(self._h, pl.ctypes.data, pe.ctypes.data, out.ctypes.data), "gp_vgicp_multi_batch_compute_error")
        return out

    def last_timing(self):
        import ctypes as C

        a, b = C.c_float(), C.c_float()
        self._lib.gp_vgicp_multi_batch_last_timing(self._h, C.byref(a), C.byref(b))
        return dict(ms_compute=a.value, ms_exchange=b.value)


class ShardedLinearizer:
    """Drives one rank's shard of a global factor table.

      issue(poses_local, out_view)  -- computes this rank's records into `out_view`, a [F_local x 122] f64 view of the
                                       stacked buffer (on GPUs: gp_vgicp_batch_issue_linearize with out_dev = view pointer)
    """

    def __init__(self, total_factors, slot_range, device, issue, group=None, stream=None, always_exchange=False, exchange="all_reduce", host_out=None, peer_timeout_ms=None):
        """stream: the torch.cuda.Stream the `issue` callback launches its kernels on (a torch.cuda.ExternalStream around the
        batch's hipStream_t when the batch owns its stream).  The zeroing of the stack, the kernels and the collective are then
        all ordered on that one stream; None = torch's current stream (CPU / gloo, or a batch created on torch's stream).
        exchange: "all_reduce" (the north star's: sum over the zeroed stack, any partition) or "all_gather" (in place, no zeroing, half the bytes: needs EQUAL
        contiguous shards in rank order -- every rank passes the same total and its own [rank * n, (rank + 1) * n) -- and falls back to the all-reduce otherwise) or
        "peer" (direct stores into every peer's buffer over xGMI, csrc/gp_peer.hip: the same plan as the all-gather, at most 8192 doubles per rank, 16 ranks; validated
        against the all-gather once, falls back to it -- all ranks together -- when the buffers cannot be shared or the validation fails).
        host_out: optional pinned [F_total x 122] f64 tensor; with the peer exchange the exchange kernel itself fills it (no D2H copy), see `delivers_to_host`.
        peer_timeout_ms: the time box a rank's exchange kernel waits for its peers (peer form; default 2000 ms -- a rank skew above it is an ERROR there, where a
        collective would wait: raise it under a debugger).  A rank that gives up poisons its peers' arrival words: all ranks fail in the same step, check() raises, and
        the exchange stays broken (build a new ShardedLinearizer with exchange="all_gather").
        The peer form maps the peers' buffers: call close() on every rank (collective) when done; an object dropped without it unmaps on its own, without the barrier."""
        import torch

        if host_out is not None:  # the peer form's kernel stores F_total x 122 doubles through its raw pointer: refuse anything that is not exactly that
            if not (isinstance(host_out, torch.Tensor) and host_out.dtype == torch.float64 and tuple(host_out.shape) == (int(total_factors), RECORD_DOUBLES)
                    and host_out.is_contiguous() and host_out.device.type == "cpu"):
                raise ValueError(f"host_out must be a contiguous float64 host tensor of shape ({int(total_factors)}, {RECORD_DOUBLES})")
            if exchange == "peer" and torch.device(device).type == "cuda" and not host_out.is_pinned():
                raise ValueError("host_out must be pinned memory for the peer exchange (torch.Tensor.pin_memory())")
        self.peer_timeout_ms = float(peer_timeout_ms) if peer_timeout_ms else None

        self.total = int(total_factors)
        self.begin, self.end = int(slot_range[0]), int(slot_range[1])
        self.issue = issue
        self.group = group
        self.stream = stream
        self.always_exchange = bool(always_exchange)  # run the collective with ONE rank as well (a 1-rank communicator is valid: RCCL smoke on a 1-GPU box)
        self.stacked = torch.zeros((self.total, RECORD_DOUBLES), dtype=torch.float64, device=device)
        # (a step is tens of microseconds: what does not change from call to call is looked up once)
        self.own_rows = self.stacked[self.begin : self.end] if self.end > self.begin else None
        self._want = exchange
        self.host_out = host_out
        self._px = None            # gp_peer_exchange handle when the peer form runs
        self._px_stack = None      # its two generations as torch views [2][F_total x 122] into the library's buffer
        self._px_own = None        # ... and this rank's rows of each
        self.peer_note = None      # why the peer form was not taken (None: not asked for, or taken)
        self._exchange = None  # None = not decided yet: torch.distributed may be initialised after this object; decided by the first pass that finds it up
        self.exchange = "none"

    def _decide(self):
        """Which exchange the passes run: decided ONCE, by all ranks together (a fallback taken by the rank that saw an exception alone would leave the
        ranks issuing different collectives).  all_gather needs the same plan on every rank (MIN over the ranks' own checks) AND a backend that accepts the in-place
        form: the form is probed here, once, and the ranks agree on the outcome (MIN again) before the first real pass."""
        import torch
        import torch.distributed as dist

        if not dist.is_initialized():
            return False  # (not cached: asked again by the next pass)
        world = dist.get_world_size(self.group)
        rank = dist.get_rank(self.group)
        self._exchange = world > 1 or self.always_exchange
        rows = self.end - self.begin
        want_gather = self._want in ("all_gather", "peer")  # (the peer form needs the all-gather's plan and falls back to it)
        gather_ok = want_gather and rows > 0 and rows * world == self.total and self.begin == rank * rows
        if self._exchange and self._want == "peer":
            self._setup_peer(world, rank, rows, gather_ok)
            if self._px is not None:
                self.exchange = "peer"
                return self._exchange
        if self._exchange and want_gather:

            def agreed(ok):
                flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.stacked.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
                return bool(flag.item())

            gather_ok = agreed(gather_ok)
            if gather_ok:  # (every rank is here, or none)
                try:
                    dist.all_gather_into_tensor(self.stacked, self.own_rows, group=self.group)  # the probe: the stack's contents are overwritten by the pass anyway
                    accepted = True
                except (RuntimeError, ValueError, NotImplementedError):  # an argument check that rejects overlapping buffers: nothing was issued
                    accepted = False
                gather_ok = agreed(accepted)
        self.exchange = ("all_gather" if gather_ok else "all_reduce") if self._exchange else "none"
        return self._exchange

    def _run(self, poses_local):
        import torch.distributed as dist

        exchange = self._exchange if self._exchange is not None else self._decide()
        if exchange and self.exchange == "peer":
            return self._run_peer(poses_local)
        if exchange and self.exchange == "all_reduce":
            self.stacked.zero_()
        if self.own_rows is not None:
            self.issue(poses_local, self.own_rows)
        if exchange:
            if self.exchange == "all_gather":
                dist.all_gather_into_tensor(self.stacked, self.own_rows, group=self.group)  # in place: the input is this rank's slot of the output
            else:
                dist.all_reduce(self.stacked, op=dist.ReduceOp.SUM, group=self.group)
        return self.stacked

    # ---- the peer form (gp_peer_exchange_*) ----
    @property
    def delivers_to_host(self):
        """True when a pass leaves the complete stack in `host_out` by itself (the peer exchange with a host_out tensor): synchronise the stream, call check(), read host_out"""
        return self._px is not None and self.host_out is not None

    def check(self):
        """after the stream's synchronisation: raises when a peer did not arrive within the exchange kernel's time box (peer form; a no-op otherwise)"""
        if self._px is not None:
            from . import _capi

            _capi.check(_capi.load().gp_peer_exchange_check(self._px), "gp_peer_exchange_check")

    def _stream_ptr(self):
        import ctypes as C

        import torch

        st = self.stream if self.stream is not None else torch.cuda.current_stream(self.stacked.device)
        return C.c_void_p(st.cuda_stream)

    def _setup_peer(self, world, rank, rows, plan_ok):
        """creates the buffers, exchanges their IPC handles, maps the peers, validates three exchanges against known rows; every decision is taken by all ranks together"""
        import ctypes as C

        import torch
        import torch.distributed as dist

        from . import _capi

        lib = _capi.load()
        dev = self.stacked.device

        def agreed(ok):
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            return bool(flag.item())

        row_doubles = rows * RECORD_DOUBLES
        if not agreed(plan_ok and dev.type == "cuda" and 0 < row_doubles <= 8192 and world <= 16):
            self.peer_note = "plan not eligible (equal contiguous shards in rank order, <= 8192 doubles per rank, <= 16 ranks, CUDA device)"
            return
        px, ok, why = C.c_void_p(), True, None
        nbytes = int(lib.gp_peer_exchange_handle_bytes())
        handle = (C.c_char * nbytes)()
        try:
            _capi.check(lib.gp_peer_exchange_create(world, rank, row_doubles, C.byref(px), handle), "gp_peer_exchange_create")
        except Exception as exc:  # (no IPC on this stack, no fine-grained memory ...)
            ok, why = False, f"create: {exc}"
        gathered = [None] * world
        dist.all_gather_object(gathered, (ok, bytes(handle.raw)), group=self.group)  # (always: every rank takes part, whatever its own outcome)
        ok_all = all(g[0] for g in gathered)
        if ok_all:
            try:
                blob = b"".join(g[1] for g in gathered)
                _capi.check(lib.gp_peer_exchange_connect(px, blob), "gp_peer_exchange_connect")
            except Exception as exc:
                ok, why = False, f"connect: {exc}"
        ok_all = agreed(ok_all and ok)
        if ok_all:
            # the two generations of the library's stack as torch tensors (the batch's kernels write this rank's rows through them)
            views = []
            for gen in (0, 1):
                ptr = int(lib.gp_peer_exchange_rows(px, gen))

                class _Raw:
                    __cuda_array_interface__ = dict(shape=(self.total, RECORD_DOUBLES), typestr="<f8", data=(ptr, False), version=2, strides=None)

                views.append(torch.as_tensor(_Raw(), device=dev))
            # validation: three exchanges of rows every rank can predict, through a pinned host stack
            probe_host = torch.zeros((self.total, RECORD_DOUBLES), dtype=torch.float64).pin_memory()
            good = True
            try:
                for step in range(3):
                    gen = int(lib.gp_peer_exchange_begin(px))
                    mine = torch.arange(self.begin, self.end, dtype=torch.float64, device=dev)[:, None] * 1000.0 + torch.arange(RECORD_DOUBLES, dtype=torch.float64, device=dev)[None, :] + 0.5 * step
                    with torch.cuda.stream(self.stream) if self.stream is not None else _nullcontext():
                        views[gen][self.begin : self.end].copy_(mine)
                        _capi.check(lib.gp_peer_exchange_finish(px, self._stream_ptr(), C.c_void_p(probe_host.data_ptr())), "gp_peer_exchange_finish")
                    torch.cuda.synchronize(dev)
                    _capi.check(lib.gp_peer_exchange_check(px), "gp_peer_exchange_check")
                    want = torch.arange(self.total, dtype=torch.float64)[:, None] * 1000.0 + torch.arange(RECORD_DOUBLES, dtype=torch.float64)[None, :] + 0.5 * step
                    good = good and bool(torch.equal(probe_host, want)) and bool(torch.equal(views[gen].cpu(), want))
            except Exception as exc:
                good, why = False, f"validation: {exc}"
            ok_all = agreed(good)
            if ok_all:
                if self.peer_timeout_ms:
                    _capi.check(lib.gp_peer_exchange_set_timeout_ms(px, self.peer_timeout_ms), "gp_peer_exchange_set_timeout_ms")
                self._px, self._px_stack = px, views
                self._px_own = [v[self.begin : self.end] for v in views]  # (the same two objects every step: callers may key on them)
                return
            why = why or "validation: the stack did not carry every rank's rows"
        self.peer_note = why or "a peer could not share its buffer"
        # (every rank is here -- the decisions above were collective -- including those whose own buffer was never created: the barrier is everybody's;
        # nobody unmaps a buffer a peer's validation kernel may still write)
        torch.cuda.synchronize(dev)
        try:
            dist.barrier(group=self.group)
        finally:
            if px:
                lib.gp_peer_exchange_destroy(px)

    def _run_peer(self, poses_local):
        import ctypes as C

        from . import _capi

        lib = _capi.load()
        gen = int(lib.gp_peer_exchange_begin(self._px))  # (names the generation; the step advances in finish, once the exchange kernel is launched: an `issue` that
        stack = self._px_stack[gen]                      #  raises leaves this rank in step with its peers)
        if self.end > self.begin:
            self.issue(poses_local, self._px_own[gen])
        _capi.check(lib.gp_peer_exchange_finish(self._px, self._stream_ptr(), C.c_void_p(self.host_out.data_ptr()) if self.host_out is not None else None), "gp_peer_exchange_finish")
        return stack

    def close(self):
        """unmaps the peers' buffers (collective: every rank calls it)"""
        if self._px is not None:
            import torch
            import torch.distributed as dist

            from . import _capi

            torch.cuda.synchronize(self.stacked.device)
            if dist.is_initialized():
                dist.barrier(group=self.group)
            _capi.load().gp_peer_exchange_destroy(self._px)
            self._px, self._px_stack, self._px_own = None, None, None

    def __del__(self):  # (last resort: no barrier here -- close() is the collective form)
        px = getattr(self, "_px", None)
        if px is not None:
            try:
                from . import _capi

                _capi.load().gp_peer_exchange_destroy(px)
            except Exception:
                pass
            self._px = None

    def linearize(self, poses_local):
        """Returns the stacked [F_total x 122] tensor holding every rank's records (device-resident)."""
        if self.stream is None:
            return self._run(poses_local)
        import torch

        with torch.cuda.stream(self.stream):  # zero_ / all_reduce follow the stream the kernels are issued on
            return self._run(poses_local)
