"""Data-parallel layer: one process per GPU, bucketed gradient all-reduce on RCCL
(`torch.distributed` backend "nccl" is RCCL on ROCm) overlapped with backward.

Replaces the reference's single-process `torch.nn.DataParallel` (train.py:220-228,
test.py:97-105: per-step parameter broadcast, scatter, gather, reduce-to-device-0).

Design for an 8-GPU MI355X node (xGMI full mesh, 7 links x ~153 GB/s per GPU):
  * gradients live in a few large flat buffers (`param.grad` are views), so every
    collective moves one contiguous message - fewer, larger transfers;
  * buckets are filled in REVERSE parameter order (= the order backward produces
    gradients).  For BCNN the classifier weight (209.7 MB of the 268.6 MB total)
    is produced first and forms bucket 0 on its own: its all-reduce is in flight
    during the whole VGG backward;
  * buckets are 16 MiB (a parameter above that is a bucket of its own): measured on the MI355X with a forced
    single-rank RCCL group (bench.py --force-pg), the classifier bucket is issued 0.3 ms into a 140 ms backward; with
    round 1's 64 MiB limit all 26 VGG tensors (56 MiB) formed ONE bucket that was only complete at the very end of
    backward - nothing left to hide it behind;
  * each bucket's all-reduce is issued from a post-accumulate-grad hook the moment
    its last gradient lands (async on RCCL's stream, overlapping the rest of
    backward); `finish()` joins before `optimizer.step()`;
  * parameters and buffers are broadcast from rank 0 once at construction
    (BatchNorm statistics stay local, as under the reference's DataParallel).

The pooling heads are per-sample independent, so the data path itself needs no
collective: the batch is sharded and this gradient all-reduce is the only exchange.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, force=False):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, local_rank).
    `force=True` creates the group even for a single rank, so that the collective path (RCCL on a GPU box) executes
    on a one-GPU machine - used by bench.py --force-pg and the GPU tests."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
            _arm_rccl_log(rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


_RCCL_LOG = None


def _arm_rccl_log(rank):
    """Before the communicator exists: have RCCL write its INIT / GRAPH / TUNING lines to a per-process file (the caller's
    own NCCL_DEBUG* settings win), so that the first run on more than one GPU records by itself what RCCL built over the
    xGMI mesh - channel count, rings / trees, transport per peer, the algorithm / protocol picked per message size - without
    a code change (SURVEY section 5's ring-vs-direct arithmetic; replaces nothing in the reference, which has no log of
    DataParallel's copies either).  HAWKEYE_RCCL_LOG=0 turns it off.  Side effect worth having: RCCL's version banner, which
    it otherwise leaves in the C stdio buffer of STDOUT until the process exits (measured on the MI355X box: five lines BEHIND
    everything python printed), goes to the file as well."""
    global _RCCL_LOG
    # (an image-wide NCCL_DEBUG=VERSION / WARN is not a choice of the caller's: only INFO / TRACE or an explicit file are)
    if os.environ.get('HAWKEYE_RCCL_LOG', '1') == '0' or os.environ.get('NCCL_DEBUG', '').upper() in ('INFO', 'TRACE') or \
            'NCCL_DEBUG_FILE' in os.environ:
        return
    import tempfile
    _RCCL_LOG = os.path.join(tempfile.gettempdir(), f'hk_rccl_{os.getpid()}_r{rank}.log')
    os.environ['NCCL_DEBUG'] = 'INFO'
    os.environ['NCCL_DEBUG_SUBSYS'] = 'INIT,GRAPH,TUNING,ENV'
    os.environ['NCCL_DEBUG_FILE'] = _RCCL_LOG


def rccl_summary(max_lines=6):
    """What the armed log says about this rank's communicator, boiled down for a bench line: library version, channels,
    how many peers are reached over which transport (P2P/IPC = xGMI or PCIe peer access, SHM, NET), the ring / tree lines'
    count and the first `max_lines` tuning decisions.  None when the log was not armed or is empty."""
    if _RCCL_LOG is None or not os.path.isfile(_RCCL_LOG):
        return None
    import re
    out = {'version': None, 'channels': None, 'rings': 0, 'trees': 0, 'transports': {}, 'tuning': [], 'env': []}
    with open(_RCCL_LOG, errors='replace') as f:
        for line in f:
            body = line.split('NCCL INFO', 1)[-1].strip()
            m = re.search(r'(RCCL|NCCL) version ([^\s]+)', line)
            if m and out['version'] is None:
                out['version'] = f'{m.group(1)} {m.group(2)}'
            m = re.search(r'Channel (\d+)/(\d+)\s*:', body)
            if m:
                out['channels'] = int(m.group(2))
            m = re.search(r'(\d+) coll channels', body)
            if m:
                out['channels'] = int(m.group(1))
            if re.match(r'Ring \d+', body):
                out['rings'] += 1
            if re.match(r'Trees? ', body):
                out['trees'] += 1
            m = re.search(r'\[(send|receive)\] via ([A-Za-z0-9/_]+)', body)
            if m:
                out['transports'][m.group(2)] = out['transports'].get(m.group(2), 0) + 1
            if ('Algo' in body or 'algorithm' in body.lower()) and 'AllReduce' in body and len(out['tuning']) < max_lines:
                out['tuning'].append(body[:120])
            if body.startswith(('NCCL_', 'RCCL_')) and 'set by environment' in body and len(out['env']) < max_lines:
                out['env'].append(body[:80])
    return out if (out['version'] or out['channels'] or out['transports']) else None


def _is_dense_permutation(t):
    """True when t's strides are a permutation of the contiguous strides (no gaps, no overlap), e.g. channels_last."""
    dims = sorted(range(t.dim()), key=lambda d: (t.stride(d), t.shape[d]), reverse=True)
    expect = 1
    for d in reversed(dims):
        if t.shape[d] != 1 and t.stride(d) != expect:
            return False
        expect *= t.shape[d]
    return True


class _Bucket:
    __slots__ = ('params', 'flat', 'views', 'pending', 'work', 'nbytes', 'issued')

    def __init__(self, params):
        self.params = params
        total = sum(p.numel() for p in params)
        self.flat = torch.zeros(total, dtype=params[0].dtype, device=params[0].device)
        self.views, off = [], 0
        for p in params:
            seg = self.flat[off:off + p.numel()]
            # same memory layout as the parameter (conv weights of a channels_last model are NHWC-strided): the
            # gradient then accumulates and the optimiser steps on matching strides instead of strided fall-backs
            dense = p.is_contiguous() or not _is_dense_permutation(p)
            self.views.append(seg.view_as(p) if dense else seg.as_strided(p.shape, p.stride()))
            off += p.numel()
        self.pending = len(params)
        self.work = None
        self.issued = None                               # trace mode: event recorded where the all-reduce was issued
        self.nbytes = total * self.flat.element_size()


class GradientAllReducer:
    """Wrap-free DDP: `model` stays the plain module (trainers keep reading
    `.backbone` / `.classifier` / `.pool`, cf. Examples/CBCNN.py:14,21 and
    Examples/MPN.py:15-17 which break under a DataParallel wrapper)."""

    def __init__(self, module, bucket_mb=16.0, process_group=None, broadcast=True, trace=False):
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        # collectives are issued whenever a process group exists - also a single-rank one (an all-reduce over one
        # rank is the identity): that is how the RCCL path is exercised on a one-GPU box
        self.collective = dist.is_initialized()
        self.trace = bool(trace) and torch.cuda.is_available()
        self._t_start = self._t_bwd_end = self._t_joined = None
        self.buckets = []
        self._index = {}
        self._handles = []
        if broadcast and self.collective:
            self.broadcast_state()
        self._build(bucket_mb)

    # ------------------------------------------------------------------ setup
    def broadcast_state(self):
        with torch.no_grad():
            for t in list(self.module.parameters()) + list(self.module.buffers()):
                dist.broadcast(t.data, src=0, group=self.group)

    def _build(self, bucket_mb):
        limit = int(bucket_mb * 1024 * 1024)
        params = [p for p in self.module.parameters() if p.requires_grad]
        cur, cur_bytes, groups = [], 0, []
        for p in reversed(params):                      # reverse registration ~ backward production order
            nbytes = p.numel() * p.element_size()
            same = (not cur) or (cur[0].dtype == p.dtype and cur[0].device == p.device)
            # a bucket is closed when the next parameter would overflow it - unless it is still tiny (< 1 MiB and < a quarter of the limit): a bias must
            # not become a collective of its own in front of its 210 MB weight (BCNN: classifier.bias + classifier.weight
            # form bucket 0 together, the first thing backward produces)
            if cur and (not same or (cur_bytes + nbytes > limit and cur_bytes >= min(1 << 20, limit // 4))):
                groups.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            groups.append(cur)
        for g in groups:
            b = _Bucket(g)
            for i, p in enumerate(g):
                self._index[p] = (b, i)
                p.grad = b.views[i]                     # gradients accumulate straight into the flat buffer
                self._handles.append(p.register_post_accumulate_grad_hook(self._hook))
            self.buckets.append(b)

    # ------------------------------------------------------------------ per step
    def zero_grad(self):
        """Use instead of optimizer.zero_grad().  Nothing is cleared: `.grad` is dropped, so that the step's first
        accumulation WRITES (autograd hands the fresh gradient over instead of adding it to zeros) and the post-accumulate
        hook moves it into the flat buffer - one read + one write per gradient byte where a memset of the buffers followed by
        autograd's read-modify-write cost a write + two reads + a write (BCNN: 268.6 MB of gradients per step).  A second
        backward before the step accumulates in place into the views, as before."""
        for b in self.buckets:
            b.pending = len(b.params)
            b.work = None
            b.issued = None
            for p in b.params:
                p.grad = None
        if self.trace:
            self._t_start = torch.cuda.Event(enable_timing=True)
            self._t_start.record()

    def _hook(self, p):
        b, i = self._index[p]
        if p.grad.data_ptr() != b.views[i].data_ptr():  # the step's first gradient (or somebody reset .grad)
            b.views[i].copy_(p.grad)
            p.grad = b.views[i]
        b.pending -= 1
        if b.pending == 0 and self.collective:
            self._issue(b)

    def _fill_missing(self, b):
        """A parameter that produced no gradient this step: its slice of the flat buffer still holds the previous step's
        values - clear it and hand it out as the gradient (what the memset gave), so that every rank reduces the same thing."""
        for p, v in zip(b.params, b.views):
            if p.grad is None:
                v.zero_()
                p.grad = v

    def _issue(self, b):
        if self.trace:
            b.issued = torch.cuda.Event(enable_timing=True)
            b.issued.record()                           # position in the backward stream where the bucket was complete
        b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        """Join all in-flight all-reduces and turn sums into means.  Call after backward, before step."""
        if self.trace:
            self._t_bwd_end = torch.cuda.Event(enable_timing=True)
            self._t_bwd_end.record()
        for b in self.buckets:
            if self.collective:
                if b.pending != 0:                      # a parameter got no gradient this step: reduce zeros for it
                    self._fill_missing(b)
                    self._issue(b)
                if b.work is not None:
                    b.work.wait()
                    if self.world > 1:
                        b.flat.div_(self.world)
            b.pending = len(b.params)
            b.work = None
        if self.trace:
            self._t_joined = torch.cuda.Event(enable_timing=True)
            self._t_joined.record()

    def timeline(self):
        """Trace mode, after a step: where in the backward each bucket's all-reduce was issued.  Milliseconds on the
        compute stream since zero_grad(): {'backward_end_ms', 'joined_ms', 'buckets': [(params, MiB, issued_ms)]}.
        A bucket issued long before `backward_end_ms` has that much backward compute to hide behind."""
        if not self.trace or self._t_start is None or self._t_bwd_end is None:
            return None
        torch.cuda.synchronize()
        t0 = self._t_start
        return {'backward_end_ms': round(t0.elapsed_time(self._t_bwd_end), 3),
                'joined_ms': round(t0.elapsed_time(self._t_joined), 3),
                'buckets': [(len(b.params), round(b.nbytes / 2 ** 20, 1),
                             round(t0.elapsed_time(b.issued), 3) if b.issued is not None else None)
                            for b in self.buckets]}

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []

    def describe(self):
        return [(len(b.params), round(b.nbytes / 2 ** 20, 1)) for b in self.buckets]
