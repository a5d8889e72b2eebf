"""Flat-buffer optimizer for the pre-training step: one-cycle (lr, beta1) schedule, global-norm clip,
decoupled weight decay and Adam fused into two HIP launches over ONE contiguous fp32 buffer, and the
data-parallel gradient exchange as a single RCCL all-reduce of that same buffer.

Replaces the reference's per-parameter python loops (build_optimizer 'adam_onecycle',
tools/train_utils/optimization/__init__.py:19-32; OptimWrapper.step, fastai_optim.py:135-152; OneCycle,
learning_schedules_fastai.py:60-77; clip_grad_norm_, train_utils.py:52) and DDP's bucketed reducer
(tools/train.py:146) for this model: 8.09 M parameters = 32.4 MB, i.e. one 32 MB collective per step -
on the xGMI full mesh RCCL splits it across all 7 links, and the buffer needs no flatten/unflatten
copies because parameters and gradients are views into the flat buffers.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.distributed as dist

from . import lib as L


def one_cycle(step: int, total_step: int, lr_max: float, moms, div_factor: float, pct_start: float):
    """(lr, beta1) of the reference OneCycle at ``step`` (cosine anneal, two phases, later phase wins)."""
    def cos(a, b, pct):
        return b + (a - b) / 2 * (math.cos(math.pi * pct) + 1)
    a1 = int(total_step * pct_start)
    low = lr_max / div_factor
    lr, mom = low, moms[0]
    if step >= 0:
        pct = step / a1 if a1 > 0 else 1.0
        lr, mom = cos(low, lr_max, pct), cos(moms[0], moms[1], pct)
    if step >= a1:
        pct = (step - a1) / (total_step - a1)
        lr, mom = cos(lr_max, low / 1e4, pct), cos(moms[1], moms[0], pct)
    return lr, mom


def default_bucket_of(name: str) -> str:
    """Gradient-exchange bucket of a parameter name = the unit whose backward finishes together: every SST stage
    (``backbone_3d.sst_blocks.<i>``) is one bucket, everything else goes by its top-level module, with the decoder
    (``backbone_3d.decoder_*``) as one bucket of its own."""
    parts = name.split(".")
    if len(parts) >= 3 and parts[1] == "sst_blocks":
        return ".".join(parts[:3])
    if len(parts) >= 2 and parts[1].startswith("decoder"):
        return parts[0] + ".decoder"
    return parts[0]


class FlatAdamOneCycle:
    """Owns flat fp32 parameter / gradient / moment buffers; model parameters become views into them.

    Layout: parameters are grouped into BUCKETS (``bucket_of(name)``, registration order = forward order, so the backward
    completes them last-to-first); inside a bucket the parameters the reference optimises come first, the ones it leaves
    untouched (see ``reference_layer_groups``) after them.  Each bucket is one contiguous range of the flat gradient =
    one all-reduce, which ``GradSync`` launches on a communication stream as soon as the backward has passed the
    bucket (DDP's bucketed, overlapped reducer - reference tools/train.py:146 - without flatten / unflatten copies)."""

    def __init__(self, model: torch.nn.Module, optim_cfg, total_steps: int, process_group=None,
                 reference_layer_groups: bool = True, bucket_of=default_bucket_of):
        """``reference_layer_groups`` (default, = what the reference trains): build_optimizer('adam_onecycle') collects
        the optimizer's parameters with ``flatten_model`` = the LEAF modules of the model
        (tools/train_utils/optimization/__init__.py:20-31, fastai_optim.py:16-27,115-122), so parameters registered
        directly on a module that also has children are never updated: for this model the 36 tensors
        ``win_attn.self_attn.{in_proj_weight, in_proj_bias, tau}`` (1.78 M of 8.09 M parameters; verified against the
        imported reference, tests/golden/optimizer_params.json).  Their gradients still enter the global clip norm
        (``clip_grad_norm_(model.parameters())``, train_utils.py:52) and the all-reduce.  The Adam launch covers only the
        optimised range of every bucket.  False: every parameter is optimised."""
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        assert named and all(p.dtype == torch.float32 for _, p in named)
        if reference_layer_groups:
            leaf_owned = {id(p) for m in model.modules() if not any(True for _ in m.children())
                          for p in m.parameters(recurse=False)}
        else:
            leaf_owned = {id(p) for _, p in named}
        order, groups = [], {}
        for n, p in named:
            b = bucket_of(n)
            if b not in groups:
                groups[b] = ([], [])
                order.append(b)
            groups[b][0 if id(p) in leaf_owned else 1].append(p)
        params, frozen, self.buckets, self.segments = [], [], [], []
        off = 0
        for b in order:
            opt_p, frz_p = groups[b]
            n_opt = sum(p.numel() for p in opt_p)
            n_all = n_opt + sum(p.numel() for p in frz_p)
            self.buckets.append((b, off, off + n_all))
            self.segments += [off, off + n_opt]
            params += opt_p + frz_p
            frozen += frz_p
            off += n_all
        # adjacent optimised ranges (a bucket without frozen parameters followed by the next bucket) are one Adam segment;
        # gdmae_adam_step takes up to 64 of them
        merged = []
        for lo, hi in zip(self.segments[0::2], self.segments[1::2]):
            if hi == lo:
                continue
            if merged and merged[-1][1] == lo:
                merged[-1][1] = hi
            else:
                merged.append([lo, hi])
        self.segments = [v for seg in merged for v in seg]
        if len(merged) > 64:
            raise ValueError(f"{len(merged)} disjoint optimised ranges: gdmae_adam_step takes at most 64 segments - pass a coarser "
                             "bucket_of (e.g. one bucket per top-level module)")
        self.frozen = frozen
        self.model = model
        dev = params[0].device
        n = off
        self.n = n
        self.n_opt = n - sum(p.numel() for p in frozen)
        if frozen:
            import logging
            logging.getLogger(__name__).info(
                "FlatAdamOneCycle: %d of %d parameters (%d tensors registered on non-leaf modules) are NOT optimised, as in the "
                "reference's adam_onecycle (pass reference_layer_groups=False to train them)", n - self.n_opt, n, len(frozen))
        self.flat_param = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in params:
            k = p.numel()
            self.flat_param[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat_param[off:off + k].view_as(p.data)
            p.grad = self.flat_grad[off:off + k].view_as(p.data)     # autograd accumulates in place
            p._gd_flat_grad = p.grad                                  # hand-written backwards write here directly
            off += k
        self.params = params
        self.flat_param_bf16 = torch.empty(n, dtype=torch.bfloat16, device=dev)
        self._refresh_shadows()
        self.cfg = optim_cfg
        self.total_steps = max(int(total_steps), 1)
        self.t = 0
        self.pg = process_group
        self._part = torch.empty(1024, dtype=torch.float32, device=dev)
        self._sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._grad_scale = 1.0
        self.sync = GradSync(self)

    def _refresh_shadows(self, cast=True):
        """One cast of the flat buffer -> bf16 views per parameter (consumed by gdmae_hip.ops.shadow).  ``cast=False``: the shadow
        buffer is current (written by the optimizer launch itself)."""
        if not cast:
            # fast path of the training loop: no parameter changed on the torch side since the views were stamped (one pass over the
            # version counters instead of two passes with pointer comparisons: 0.27 -> 0.05 ms of host time per step)
            stamp = getattr(self, "_shadow_stamp", None)
            if stamp is not None and stamp[0] == self.flat_param_bf16.data_ptr() and stamp[1] == [p._version for p in self.params]:
                if self.flat_param.is_cuda:
                    from . import packing
                    packing.repack_registered()
                return
            # a torch-side weight change since the last stamp (load_state_dict between two steps): the optimizer launch only rewrote
            # the elements it updates, so re-derive everything
            cast = any(getattr(p, "_gd_shadow", None) is None or p._gd_shadow[1] != p._version for p in self.params)
        if cast:
            self.flat_param_bf16.copy_(self.flat_param)
        off = 0
        for p in self.params:
            k = p.numel()
            sh = getattr(p, "_gd_shadow", None)
            # the view is persistent; only a torch-side in-place change of the parameter (version bump) needs a new stamp - the
            # optimizer kernels write the flat buffer through raw pointers (190 slice + view constructions per step were 0.7 ms)
            if sh is None or sh[1] != p._version or sh[0].data_ptr() != self.flat_param_bf16.data_ptr() + 2 * off:
                p._gd_shadow = (self.flat_param_bf16[off:off + k].view(p.shape), p._version)
            off += k
        self._shadow_stamp = (self.flat_param_bf16.data_ptr(), [p._version for p in self.params])
        if self.flat_param.is_cuda:
            from . import packing
            packing.repack_registered()          # fragment-ordered images of the encoder weights (one launch)

    def refresh_weights(self):
        """Call after changing parameter VALUES behind autograd's back (``p.data.copy_``, an EMA swap through ``.data``, a raw
        pointer write): the bf16 shadows and the packed MFMA weight images are re-derived from the fp32 masters.  Changes through
        torch that bump the parameter's version counter (``load_state_dict``, ``p.copy_`` under no_grad) are noticed without it."""
        for p in self.params:
            p._gd_shadow = None
        self._refresh_shadows()

    def _check_views(self):
        for p in self.params:
            if p.grad is not p._gd_flat_grad:
                raise RuntimeError("a parameter lost its flat gradient view (model.zero_grad() / set_to_none=True?): its "
                                   "gradient would not reach the optimizer - use FlatAdamOneCycle.zero_grad()")

    def zero_grad(self):
        self._check_views()
        self.flat_grad.zero_()
        self.sync.begin_step()

    def world_size(self):
        return dist.get_world_size(self.pg) if (dist.is_available() and dist.is_initialized()) else 1

    def all_reduce_grads(self):
        """Finish the data-parallel gradient exchange of this step: buckets that ``GradSync`` already launched from inside
        the backward are waited for, the others are all-reduced now.  The buffer then holds the SUM over ranks; the
        division by the world size is folded into the Adam launch (DDP averaging semantics without a pass of its own)."""
        self.sync.finish()
        self._grad_scale = 1.0 / self.world_size()

    def broadcast_buffers(self, src: int = 0):
        """DDP's ``broadcast_buffers`` (reference tools/train.py:146 wraps the model with the default True: BatchNorm running
        statistics and ``global_step`` of every rank are overwritten with rank ``src``'s before each forward).  The training
        arithmetic never reads those buffers and rank 0 - the rank that writes checkpoints - only ever sees its own, so the
        step itself skips the per-step broadcast; call this before evaluating / saving on a rank other than ``src``."""
        if self.world_size() <= 1:
            return
        for b in self.model.buffers():
            dist.broadcast(b, src=src, group=self.pg)

    def step(self, accumulated_iter: int | None = None):
        self._check_views()
        it = self.t if accumulated_iter is None else accumulated_iter
        c = self.cfg
        lr, beta1 = one_cycle(it, self.total_steps, c.LR, list(c.MOMS), c.DIV_FACTOR, c.PCT_START)
        self.t += 1
        if self.flat_param.is_cuda:
            st = L.stream()
            L.call("gdmae_grad_sq_norm", L.ptr(self.flat_grad), self.n, L.ptr(self._part), L.ptr(self._sq), st)
            L.call("gdmae_adam_step_shadow", L.ptr(self.flat_param), L.ptr(self.flat_grad), L.ptr(self.exp_avg),
                   L.ptr(self.exp_avg_sq), L.host_i64(self.segments), len(self.segments) // 2, float(lr), float(beta1), 0.99, 1e-8,
                   float(c.WEIGHT_DECAY), self.t, float(c.GRAD_NORM_CLIP), float(self._grad_scale), L.ptr(self._sq),
                   L.ptr(self.flat_param_bf16), st)
        else:
            raise RuntimeError("FlatAdamOneCycle.step needs the HIP library and device buffers (no CPU fallback)")
        self._grad_scale = 1.0
        self._refresh_shadows(cast=False)       # the Adam launch wrote the bf16 copy of every element it updated
        return lr, beta1

    # ---- checkpoint wire format: the reference stores OptimWrapper.opt.state_dict(), i.e. the state_dict of a
    #      torch.optim.Adam with two param groups [non-BatchNorm leaf parameters, BatchNorm parameters] in flatten_model
    #      order (tools/train_utils/train_utils.py:147-163; fastai_optim.py:16-27,115-122)
    def _reference_groups(self):
        bn = (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d, torch.nn.BatchNorm3d, torch.nn.SyncBatchNorm)
        leaves = [m for m in self.model.modules() if not any(True for _ in m.children())]
        g0 = [p for m in leaves if not isinstance(m, bn) for p in m.parameters(recurse=False) if p.requires_grad]
        g1 = [p for m in leaves if isinstance(m, bn) for p in m.parameters(recurse=False) if p.requires_grad]
        return [g0, g1]

    def _offsets(self):
        off, table = 0, {}
        for p in self.params:
            table[id(p)] = (off, p.numel())
            off += p.numel()
        return table

    def state_dict(self):
        """torch.optim.Adam-compatible state of the parameters the reference optimises (loadable by the reference's
        ``optimizer.load_state_dict`` and vice versa)."""
        groups = self._reference_groups()
        table = self._offsets()
        lr, beta1 = one_cycle(max(self.t - 1, 0), self.total_steps, self.cfg.LR, list(self.cfg.MOMS), self.cfg.DIV_FACTOR,
                              self.cfg.PCT_START)
        state, pgs, idx = {}, [], 0
        for g in groups:
            ids = []
            for p in g:
                o, k = table[id(p)]
                if self.t > 0:
                    state[idx] = {"step": self.t, "exp_avg": self.exp_avg[o:o + k].view(p.shape).clone(),
                                  "exp_avg_sq": self.exp_avg_sq[o:o + k].view(p.shape).clone()}
                ids.append(idx)
                idx += 1
            pgs.append({"lr": lr, "betas": (beta1, 0.99), "eps": 1e-8, "weight_decay": 0, "amsgrad": False, "params": ids})
        return {"state": state, "param_groups": pgs}

    def load_state_dict(self, sd):
        if "param_groups" not in sd:
            # flat buffers written by early versions of this class: the buffer layout has changed since (bucket order,
            # optimised parameters first), so the moments would land on the wrong parameters without any size mismatch
            raise ValueError("optimizer state in the retired flat layout ('t' / 'exp_avg' / 'exp_avg_sq'): not loadable - "
                             "save with FlatAdamOneCycle.state_dict() (torch.optim.Adam wire format, keyed by parameter order)")
        groups = self._reference_groups()
        table = self._offsets()
        assert [len(g["params"]) for g in sd["param_groups"]] == [len(g) for g in groups], "optimizer state: group sizes differ"
        flat = [p for g in groups for p in g]
        steps = set()
        for idx, p in enumerate(flat):
            st = sd["state"].get(idx)
            if st is None:
                continue
            o, k = table[id(p)]
            assert tuple(st["exp_avg"].shape) == tuple(p.shape), (idx, st["exp_avg"].shape, p.shape)
            self.exp_avg[o:o + k].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(st["step"]))
        assert len(steps) <= 1, f"optimizer state with different step counts per tensor: {steps}"
        self.t = steps.pop() if steps else 0


class GradSync:
    """Bucketed gradient all-reduce overlapped with the backward (north_star: 'RCCL all-reduce of grads over xGMI overlapped
    with backward'; reference = DDP's reducer, tools/train.py:146).

    The hand-written backwards of this path write most weight gradients straight into the flat buffer, so parameter
    hooks never fire; instead the model marks ACTIVATIONS: ``mark(tensor, bucket_names)`` registers a tensor hook that
    runs when autograd delivers the gradient of that tensor, i.e. when every backward node created after it in the forward
    has been enqueued - at that point the named buckets (the stages after that tensor, the decoder) are complete on the
    compute stream.  The hook records an event there, makes the communication stream wait for it and launches one
    asynchronous all-reduce per bucket on it; ``finish()`` (from ``all_reduce_grads``) reduces whatever was not marked and
    orders the compute stream behind the communication stream.  With one rank nothing is launched."""

    def __init__(self, opt: "FlatAdamOneCycle"):
        self.opt = opt
        self.launched = set()
        self.works = []
        self._comm = None
        self.log = []            # (bucket name, 'overlapped' | 'tail') of the last step: what the tests look at
        # mode: 'overlap' (default) - buckets reduced from the tensor hooks inside backward();
        #       'tail'    - every bucket reduced in finish(), after backward() (the A/B reference of the tests);
        #       'check'   - hooks only SNAPSHOT what they would have handed to the collective; finish() compares every
        #                   snapshot with the final local gradient (a bucket that was marked complete too early raises)
        #                   and then reduces at the tail.  GDMAE_SYNC_MODE selects it without code changes;
        #       'off'     - no collective at all (local gradients; single-rank reference inside multi-rank tests)
        import os
        self.mode = os.environ.get("GDMAE_SYNC_MODE", "overlap")
        self._snap = {}
        # force: run the exchange even in a one-rank group (the sum over one rank is the identity): how the RCCL call pattern -
        # collectives issued from tensor hooks on the autograd thread, on the communication stream - is exercised on a box
        # with a single GPU (RCCL refuses two ranks per device)
        self.force = os.environ.get("GDMAE_SYNC_FORCE", "0") == "1"
        # The overlap rests on autograd delivering a marked tensor's gradient only after every node recorded behind it has run.
        # The FIRST step of any job that really exchanges gradients therefore runs as a 'check' step on its own (snapshots at the
        # hooks, compared with the final local gradients, reduction at the tail): a model change that breaks the marks raises at
        # iteration 0 instead of training on partial sums.  GDMAE_SYNC_AUTOCHECK=0 switches it off.
        self.autocheck = os.environ.get("GDMAE_SYNC_AUTOCHECK", "1") != "0"
        self.checked_steps = 0
        self._step_mode = self.mode
        # measure: keep (compute-stream, communication-stream) event pairs of the finish() hand-over; exposed_ms() = how long the
        # compute stream had to wait for the collectives after its own backward work was done (bench.py --gpus N)
        self.measure = False
        self._pairs = []

    def _active(self):
        return self.mode != "off" and (self.opt.world_size() > 1 or (self.force and dist.is_initialized()))

    def begin_step(self):
        assert self.mode in ("overlap", "tail", "check", "off"), self.mode
        self.launched, self.works, self.log, self._snap = set(), [], [], {}
        self._step_mode = self.mode
        if self.mode == "overlap" and self.autocheck and self.checked_steps == 0 and self._active():
            self._step_mode = "check"

    def bucket_names(self):
        return [b for b, _, _ in self.opt.buckets]

    def mark(self, tensor: torch.Tensor, done_buckets):
        """Call in the forward: when the backward reaches ``tensor``, the buckets in ``done_buckets`` are complete."""
        if not (self._active() and tensor.requires_grad):
            return
        names = list(done_buckets)

        def hook(_g):
            if self._step_mode == "overlap":
                self.reduce(names, "overlapped")
            elif self._step_mode == "check":
                for b, lo, hi in self.opt.buckets:
                    if b in names and b not in self._snap:
                        self._snap[b] = self.opt.flat_grad[lo:hi].clone()
            return None
        tensor.register_hook(hook)

    def reduce(self, names, how):
        opt = self.opt
        cuda = opt.flat_grad.is_cuda
        if cuda:
            if self._comm is None:
                self._comm = torch.cuda.Stream(device=opt.flat_grad.device)
            ev = torch.cuda.Event()
            ev.record()                                   # everything enqueued so far on the compute stream
            self._comm.wait_event(ev)
        for b, lo, hi in opt.buckets:
            if b not in names or b in self.launched or hi == lo:
                continue
            self.launched.add(b)
            self.log.append((b, how))
            view = opt.flat_grad[lo:hi]
            if cuda:
                with torch.cuda.stream(self._comm):
                    self.works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=opt.pg, async_op=True))
            else:
                self.works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=opt.pg, async_op=True))

    def finish(self):
        if not self._active():
            return
        if self._step_mode == "check":
            self.checked_steps += 1
            for b, lo, hi in self.opt.buckets:
                snap = self._snap.get(b)
                if snap is not None and not torch.equal(snap, self.opt.flat_grad[lo:hi]):
                    raise RuntimeError(f"GradSync: bucket '{b}' changed after the hook that marks it complete fired - a parameter "
                                       "of this bucket is used before the marked tensor (its all-reduce would have been launched "
                                       "on an unfinished gradient)")
        self.reduce(self.bucket_names(), "tail")
        pair = None
        if self.measure and self.opt.flat_grad.is_cuda and self._comm is not None:
            pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            pair[0].record()                              # the compute stream has nothing left but to wait ...
            pair[1].record(self._comm)                    # ... for the last collective
            self._pairs.append(pair)
        for w in self.works:
            w.wait()                                      # NCCL/RCCL: orders the CURRENT stream behind the collective
        if self.opt.flat_grad.is_cuda and self._comm is not None:
            torch.cuda.current_stream().wait_stream(self._comm)
        self.works = []

    def exposed_ms(self):
        """Mean time per measured step the compute stream waited for the gradient exchange (0 when the collectives finished under
        the backward); synchronises with the recorded events."""
        if not self._pairs:
            return None
        tot = 0.0
        for a, b in self._pairs:
            b.synchronize()
            tot += max(0.0, a.elapsed_time(b))
        n = len(self._pairs)
        self._pairs = []
        return tot / n
