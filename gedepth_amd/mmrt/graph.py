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
            # The process group's watchdog thread polls the end events of the collectives of the last EAGER step (every 100 ms) until it has
            # seen them complete.  Once the RCCL stream has joined the capture, hipEventQuery on such an event fails with
            # hipErrorCapturedEvent and the watchdog aborts the process (2 - 3 of 5 sessions when the capture followed the last eager step
            # directly; 0 of 8 with this pause, tools/final_round_run.sh's forced-exchange runs): let it retire them first.
            import time
            time.sleep(float(os.environ.get('GE_GRAPH_DDP_SETTLE', '0.5')))
        g = torch.cuda.CUDAGraph()
        hip.check(lib.ge_rng_salt(self.salt.data_ptr()), 'ge_rng_salt')
        try:
            with torch.cuda.graph(g, stream=self.stream):
                self.salt.add_(1)
                self.out = self._step_body()
        except Exception:
            lib.ge_rng_salt(None)
            raise
        self.graph = g
        lv = self.out.get('log_vars') if isinstance(self.out, dict) else None
        self._log = (list(lv.keys()), lv.tensor()) if hasattr(lv, 'tensor') and lv.tensor() is not None else None
        return self

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
                    try:
                        self.capture()           # the capture does not execute: replay it for this step
                    except Exception as e:       # an op that cannot be captured (a new code path, a library call that synchronises): train eagerly
                        import warnings
                        warnings.warn(f'GraphedTrainStep: capture failed ({type(e).__name__}: {e}); continuing with eager steps')
                        self.disabled, self.graph = True, None
                        torch.cuda.synchronize()
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
        """Drop the graph (its private memory pool) and unregister the dropout counter.  Call before discarding the object: the
        library keeps the counter's DEVICE ADDRESS."""
        if getattr(self, 'salt', None) is not None:
            hip.lib().ge_rng_salt(None)
        self.graph, self.out = None, None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass
