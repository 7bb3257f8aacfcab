"""
Row-sharded evaluation: one process per GPU, each holding a contiguous block of the
training rows; three sums over ranks per NLML+grad evaluation.

The reference has no multi-device code at all; what makes sharding exact is that
every N-dependent quantity of SCFGP/SCFGP.py:104-126 is a sum over rows of per-row
terms (Phi^T Phi, Phi^T y, y^T y; then the expected-NLL sum and the weighted Gram of
the backward pass; then X^T Zbar).  The K x K stages run replicated on every rank.

`engine` is anything with the staged interface of scfgp_amd.engine.HipEngine
(pass1/factor/pass2/adjoint/pass3/finish/exchange); `allreduce(buf)` sums the buffer
in place across ranks.  With torch.distributed (backend "nccl" == RCCL on ROCm) the
buffers are CUDA tensors aliasing the library's device memory, so the collective runs
over xGMI with no host copy.
"""
import numpy as np


def torch_allreduce(group=None):
    """All-reduce (sum, in place) through torch.distributed; accepts torch tensors or
    numpy arrays (the latter for CPU/gloo rehearsals)."""
    import torch
    import torch.distributed as dist

    def _ar(buf):
        backend = dist.get_backend(group)
        if isinstance(buf, np.ndarray):
            if backend == 'nccl':                          # host scalars (e.g. the global N)
                t = torch.from_numpy(buf).cuda()
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                buf[...] = t.cpu().numpy()
            else:
                dist.all_reduce(torch.from_numpy(buf), op=dist.ReduceOp.SUM, group=group)
        elif buf.is_cuda and backend == 'gloo':            # CPU-backend rehearsal of the GPU path
            h = buf.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            buf.copy_(h)
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return _ar


def attach_native_comm(engine, group=None):
    """The sums inside the library instead (include/scfgp_hip.h: scfgp_comm_init): rank 0's 128-byte RCCL id travels over the
    process group that is there anyway (any transport would do), every rank joins, and from then on the engine's pass1 / pass2 /
    pass3 end in their own ncclAllReduce on the library's stream.  Use the engine with ShardedEvaluator(engine, None) or call
    engine.eval() directly."""
    import torch.distributed as dist
    box = [engine.comm_unique_id() if dist.get_rank(group) == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    engine.comm_init(dist.get_world_size(group), dist.get_rank(group), box[0])


def shard_rows(N, rank, world):
    """Contiguous block [lo, hi) of rank `rank` out of `world` (sizes differ by at most 1)."""
    base, rem = divmod(N, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ShardedEvaluator(object):

    def __init__(self, engine, allreduce=None, time_exchanges=False):
        self.engine = engine
        self.allreduce = allreduce
        # time_exchanges: bracket every sum over ranks with events on the stream it runs on (inside the two fences), so
        # that a scaling record can say how much of a step the three exchanges -- transfer AND the wait for the slowest
        # rank -- took (bench.py: stages_ms exchange1..3, comm_share); exchange_ms() reads them after the evaluation
        self.time_exchanges = bool(time_exchanges)
        self._ex_events = {}
        self._ex_wall = {}

    def _sum(self, stage):
        """Sum exchange buffer `stage` over the ranks, in place.  The collective runs on torch's current stream,
        the library on its own: both sides are fenced (include/scfgp_hip.h, scfgp_stream_fence), so the order
        holds whatever stream the caller's torch code is on."""
        if self.allreduce is None:
            return
        e = self.engine
        peer = self._peer_stream()
        fenced = peer is not False and hasattr(e, 'stream_fence')
        if fenced:
            e.stream_fence(peer, 0)
        buf = e.exchange(stage)
        timed = self.time_exchanges
        if timed:
            import time
            if fenced:
                import torch
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            t0 = time.perf_counter()
        self.allreduce(buf)
        if timed:
            self._ex_wall[stage] = (time.perf_counter() - t0) * 1e3
            if fenced:
                ev[1].record()
                self._ex_events[stage] = ev
        if fenced:
            e.stream_fence(peer, 1)

    def exchange_ms(self):
        """{stage: milliseconds} of the sums of the last evaluation: stream time between the two events that bracket the
        collective (GPU side; it starts when the rank's own sweep is done and ends when the summed buffer is back, so it
        contains the wait for the slowest rank), host wall time of the call where there is no GPU side.  Call after
        finish() -- the events have completed by then."""
        out = {}
        for stage, ms in self._ex_wall.items():
            ev = self._ex_events.get(stage)
            if ev is not None:
                ev[1].synchronize()
                ms = ev[0].elapsed_time(ev[1])
            out[stage] = float(ms)
        self._ex_events.clear(); self._ex_wall.clear()
        return out

    @staticmethod
    def _peer_stream():
        """Raw handle of torch's current CUDA stream; False when there is no GPU side (CPU fakes in the tests)."""
        try:
            import torch
            if not torch.cuda.is_available():
                return False
            return torch.cuda.current_stream().cuda_stream
        except ImportError:
            return False

    def _fail_rest(self, owed, want_grad):
        """This rank's evaluation has failed with the exchanges `owed`.. still to come: mark each as failed and run its sum
        anyway (engine.fail_stage does the sum itself when the communicator lives inside the library), so that the peers reach
        their finish() -- which then raises PeerFailed on every one of them -- instead of waiting in a collective."""
        e = self.engine
        if not hasattr(e, 'fail_stage'):
            return
        for stage in range(owed, 4 if want_grad else 3):
            e.fail_stage(stage, want_grad)
            self._sum(stage)

    def eval(self, want_grad=True):
        e = self.engine
        for _ in range(5):
            owed = 1                        # the next exchange this rank owes its peers
            try:
                e.pass1(); self._sum(1); owed = 2
                # False: the summed status word of exchange 1 says some rank cannot run the precision level others ran
                # pass 1 at -- every rank reads the same word and starts again at the common level
                if e.factor() is False:
                    continue
                e.pass2(want_grad); self._sum(2); owed = 3
                if want_grad:
                    e.adjoint()
                    e.pass3(); self._sum(3); owed = 4
            except Exception:
                self._fail_rest(owed, want_grad)
                raise
            if hasattr(e, 'fetch_factors'):
                e.fetch_factors()           # everything is queued: alpha / Li reach the host beside the remaining sweeps
            out = e.finish(want_grad)       # raises engine.PeerFailed on every rank when any rank failed
            # None: the library raised its precision level (condition estimate of A too high for an fp32 Gram) and wants
            # the stages again.  Every rank factors the same summed matrix, so every rank takes the same decision.
            if out is not None:
                return out
        raise RuntimeError('precision escalation did not settle')
