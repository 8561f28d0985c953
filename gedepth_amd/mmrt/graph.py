"""A whole training step as one hipGraph.

The reference launches every op of a step from Python (``IterBasedRunner.train`` -> ``model.train_step`` -> autograd ->
``OptimizerHook``; depth/apis/train.py:28-121), ~1250 launches per step for Swin-T and ~2400 for Swin-L.  At 2 images per GPU
(configs/depthformer/depthformer_a.py: the 8-GPU configuration) most of them run for < 20 us and the GPU waits for the host: the
same binary measured 34.8 - 44.1 ms per step depending on the host, for 21 ms of kernel time.  ``GraphedTrainStep``
captures forward + losses + backward + gradient collection (+ the RCCL all-reduces FlatDDP launches from its hooks) + clip + AdamW
once — after a few eager warm-up steps, when MIOpen / hipBLASLt have chosen their kernels, the DropPath bank knows its layers and
FlatDDP has put its arena into arrival order — and replays it with ONE launch per step.

What makes the step capturable (all of it lives elsewhere, this file only orchestrates):
  * inputs are static buffers (a new batch is copied INTO them: ``static[k].copy_(batch[k])``; bench.py trains on the resident batch itself);
  * the optimizer's per-step scalars live in a device buffer refreshed BEFORE each replay (``FusedAdamW.prepare`` / ``launch``);
  * dropout masks come from (seed argument, DEVICE counter): the counter is incremented inside the graph (``ge_rng_salt``), DropPath
    draws through torch's graph-safe generator;
  * logged scalars stay on the device (``DeferredLogVars``), nothing in the step calls ``.item()``;
  * every workspace is a torch allocation (graph-private pool), the native entry points only enqueue kernels and async memsets.
Host-side heuristics evaluated at capture time are frozen into the graph (the cross-attention's query order, the d_value kernel
choice): results do not depend on them, and ``recapture()`` refreshes them.
"""
import os

import torch

from .. import hip


_CAPTURED_RECORDS = []      # [(first, last)] flight-recorder record ids of collectives recorded DURING a capture: the watchdog never sees those works


def _fr_entries():
    import pickle
    from torch._C._distributed_c10d import _dump_nccl_trace
    trace = pickle.loads(_dump_nccl_trace(includeCollectives=True, includeStackTraces=False, onlyActive=False))
    return trace.get('entries') if isinstance(trace, dict) else None


def _fr_last_record_id():
    try:
        ent = _fr_entries()
        return max(int(e.get('record_id', -1)) for e in ent) if ent else -1
    except Exception:
        return -1


def quiesce_collectives(timeout=10.0):
    """Block until the process-group watchdog has RETIRED every collective issued so far, i.e. has seen it complete and dropped it from the list
    whose end events it keeps querying.  Observed through the c10d flight recorder: every recorded entry carries ``retired`` (set by the watchdog
    when it erases the work; ``state`` / ``onlyActive`` are refreshed by the dump itself and say nothing about the watchdog).  The recorder is off
    unless ``TORCH_FR_BUFFER_SIZE`` (older builds: ``TORCH_NCCL_TRACE_BUFFER_SIZE``) is set before the process group is created —
    ``mmrt.ddp.init_dist`` sets it.  Call with the device idle.  Returns the number of polls (>= 1) when the condition was observed; if the
    recorder is off / empty / unavailable, or the timeout passes, falls back to the timed pause of round 5 (``GE_GRAPH_DDP_SETTLE`` seconds,
    default 0.5) and returns 0."""
    import time
    polls, t0 = 0, time.monotonic()
    try:
        while True:
            polls += 1
            entries = _fr_entries()
            if not entries:
                raise RuntimeError('flight recorder off or empty: nothing to observe')
            # collectives recorded inside an earlier capture are never handed to the watchdog (they execute at replay): not waited for
            live = [e for e in entries if not any(a <= int(e.get('record_id', -1)) <= b for a, b in _CAPTURED_RECORDS)]
            if all(e.get('retired', False) for e in live):
                return polls
            if time.monotonic() - t0 > timeout:
                raise TimeoutError('collectives still not retired')
            time.sleep(0.002)
    except Exception:
        time.sleep(float(os.environ.get('GE_GRAPH_DDP_SETTLE', '0.5')))
        return 0


class GraphedTrainStep:

    def __init__(self, model, optimizer, batch, amp_dtype=None, ddp=None, warmup=3, lr_updater=None):
        """``model``: the depther (or its FlatDDP wrapper as ``ddp``); ``batch``: dict of device tensors (+ non-tensor entries) whose
        tensors become the static input buffers; ``warmup`` eager steps run first (they are real training steps)."""
        self.model, self.optimizer, self.ddp = model, optimizer, ddp
        self.amp_dtype = amp_dtype
        self.static = {k: v for k, v in batch.items()}
        self.graph, self.out, self._log = None, None, None
        self.replays = 0
        dev = optimizer.arena.flat_param.device
        self.salt = torch.zeros(1, device=dev, dtype=torch.int64)       # dropout counter, advanced inside the graph
        self._warmup_left = int(warmup)
        self.disabled = False                    # set when a capture failed: the object then runs every step eagerly
        # RCCL collectives inside the captured step (FlatDDP's bucket all-reduces, the stacked loss all-reduce): see capture() for the one
        # precaution they need.  GE_GRAPH_DDP=0 keeps a step with an active gradient exchange eager.
        if ddp is not None and getattr(ddp, 'active', False) and os.environ.get('GE_GRAPH_DDP') == '0':
            self.disabled = True
        self.stream = torch.cuda.Stream(dev)                             # warm-up steps and the capture share this side stream

    # ---- one step, eager or being captured
    def _step_body(self):
        self.optimizer.zero_grad()
        wrapper = self.ddp if self.ddp is not None else self.model
        with torch.autocast('cuda', dtype=self.amp_dtype or torch.bfloat16, enabled=self.amp_dtype is not None):
            out = wrapper.train_step(self.static, self.optimizer)
        out['loss'].backward()
        # nothing may keep this step's autograd graph alive: its AccumulateGrad nodes remember the stream they were created on, and a node
        # that survives from an eager step (default stream) into the capture would run outside the capturing stream
        out['loss'] = out['loss'].detach()
        if self.ddp is not None:
            self.ddp.finish()
        self.optimizer.launch()
        return out

    def _load(self, batch):
        if batch is None:
            return
        for k, v in batch.items():
            if torch.is_tensor(v):
                if v is not self.static[k]:
                    self.static[k].copy_(v, non_blocking=True)
            else:
                self.static[k] = v

    def capture(self):
        dev = self.optimizer.arena.flat_param.device
        lib = hip.lib()
        torch.cuda.synchronize(dev)
        if self.ddp is not None and getattr(self.ddp, 'active', False):
            # The process group's watchdog thread polls the end events of the collectives of the last EAGER steps until it has seen them
            # complete.  Once the RCCL stream has joined the capture, hipEventQuery on such an event fails with hipErrorCapturedEvent and
            # the watchdog aborts the process (2 - 3 of 5 sessions when the capture followed the last eager step directly).  The device is
            # idle here (synchronize above); what is awaited is the watchdog's BOOK-KEEPING, observed through the process group's flight
            # recorder: every recorded collective carries ``retired`` (``quiesce_collectives``; a condition, not a pause — the timed pause
            # remains the fallback when the recorder is off).
            quiesce_collectives()
        g = torch.cuda.CUDAGraph()
        # the dropout kernels read the registered counter address at LAUNCH time, so it is baked into the captured kernel arguments: it only
        # has to be registered while this capture runs (a process-wide slot that outlived the capture could be cleared under a newer object,
        # or point at another device's counter)
        hip.check(lib.ge_rng_salt(self.salt.data_ptr()), 'ge_rng_salt')
        ddp_on = self.ddp is not None and getattr(self.ddp, 'active', False)
        rec0 = _fr_last_record_id() if ddp_on else -1
        try:
            with torch.cuda.graph(g, stream=self.stream):
                self.salt.add_(1)
                self.out = self._step_body()
        finally:
            lib.ge_rng_salt(None)
            if ddp_on:
                rec1 = _fr_last_record_id()
                if rec1 > rec0:
                    _CAPTURED_RECORDS.append((rec0 + 1, rec1))
        self.graph = g
        lv = self.out.get('log_vars') if isinstance(self.out, dict) else None
        self._log = (list(lv.keys()), lv.tensor()) if hasattr(lv, 'tensor') and lv.tensor() is not None else None
        return self

    def _capture_collectively(self):
        """capture(), with the outcome agreed between the ranks: a rank whose capture failed must not run eager collectives against ranks
        replaying graphs that contain theirs (hang / mismatched pairs).  One MIN all-reduce of a success flag — outside any capture, on every
        rank, whatever happened locally; if any rank failed, all drop their graphs and train eagerly.  Returns the exception or None."""
        err = None
        try:
            self.capture()
        except Exception as e:       # an op that cannot be captured (a new code path, a library call that synchronises): train eagerly
            err = e
            self.graph = None
            torch.cuda.synchronize()
            # hooks that fired during the aborted capture have already counted buckets down and queued captured work handles; gradients
            # of the aborted step may sit half-written in the arena: start the eager step from a clean slate
            if self.ddp is not None and hasattr(self.ddp, '_reset'):
                self.ddp._reset()
            self.optimizer.zero_grad()
        if self.ddp is not None and getattr(self.ddp, 'active', False):
            import torch.distributed as dist
            ok = torch.tensor([0 if err is not None else 1], device=self.optimizer.arena.flat_param.device, dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.ddp.group)
            if int(ok.item()) == 0 and err is None:
                err = RuntimeError('the capture failed on another rank')
                self.graph = None
        return err

    def recapture(self):
        """Throw the graph away and capture again (after a change of shapes, or to refresh frozen host-side heuristics)."""
        self.graph = None
        return self.capture()

    def __call__(self, batch=None):
        self._load(batch)
        cur = torch.cuda.current_stream()
        # Everything of a step — the scalar upload, the eager warm-up steps, the capture and the replays — runs on ONE side stream.
        # (Replays launched on the default stream with the optimizer's H2D copy in between faulted on ROCm 7.2 — "illegal memory
        # access" a few replays after a device synchronisation; the same sequence on a side stream does not: tools/ubench/graph/step_trace.py)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.optimizer.prepare()
            if self.graph is None and self._warmup_left > 0:
                self._warmup_left -= 1
                out = self._step_body()
            elif self.disabled:
                out = self._step_body()
            else:
                if self.graph is None:
                    err = self._capture_collectively()      # the capture does not execute: on success replay it for this step
                    if err is not None:
                        import warnings
                        warnings.warn(f'GraphedTrainStep: capture failed ({type(err).__name__}: {err}); continuing with eager steps')
                        self.disabled = True
                        out = self._step_body()
                        cur.wait_stream(self.stream)
                        return out
                self.graph.replay()
                self.replays += 1
                out = self.out
                if self._log is not None:        # the logged scalars sit in a static device tensor: a fresh lazy view per replay
                    from ..depth.models.depther.base import DeferredLogVars
                    out = dict(self.out, log_vars=DeferredLogVars(*self._log))
        cur.wait_stream(self.stream)
        return out

    def release(self):
        """Drop the graph (its private memory pool).  The dropout counter stays alive with the object; its address is registered with the
        library only for the duration of ``capture``."""
        self.graph, self.out = None, None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass
