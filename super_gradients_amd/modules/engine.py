"""Execution substrate of the MI355X train-step: flat HBM arenas + the forward/backward protocol of the blocks.

Why not torch.autograd per op: the reference executes ~1 800 ATen ops per YOLO-NAS-S step; here every block owns a
hand-written forward and backward that enqueue libsgx_hip kernels (include/sgx_hip.h) on the current HIP stream, the
whole network is ONE node in torch's autograd graph (modules/engine.py:NetFunction), and parameters / gradients / BN
buffers live in three flat fp32 arenas so that the optimizer step, EMA, zero_grad and the data-parallel all-reduce are
each a single launch / a handful of large collectives over contiguous HBM.

Protocol (all tensors NHWC, fp32, on the HIP device):
    y  = block.fwd(x, out=None)         out: optional preallocated NHWC view (e.g. a channel slice of a concat buffer)
    dx = block.bwd(dy, dx_out=None, accumulate=False, addend=None, need_dx=True)
         dx_out/accumulate: write (or add) the input gradient into an existing buffer - used where a tensor has
         several consumers; addend: extra tensor summed into dx inside the data-gradient epilogue.
State-dict compatibility: parameter/buffer names and logical shapes equal the reference's;
the physical layout (OHWI, channel padding) is hidden behind strided views into the arena.
"""
from typing import List, Optional

import torch
from torch import nn

ALIGN = 64  # floats: every arena segment starts 256-byte aligned


def _round_up(v, m):
    return (v + m - 1) // m * m


class SgxBlock(nn.Module):
    """Base of every block that runs on libsgx_hip kernels."""

    def __setattr__(self, name, value):
        # per-call state (`self._ctx = ...`, `self._x = x`: ~250 assignments per train step) bypasses nn.Module's parameter / buffer /
        # sub-module bookkeeping, which only matters for public names and for Parameter / Module values
        if name[0] == "_" and not isinstance(value, nn.Module) and not getattr(value, "_is_param", False):
            object.__setattr__(self, name, value)
        else:
            self.__dict__.pop(name, None)  # a registry mirror (SgxNetwork._build_runtime) must not outlive a re-assignment of the name
            super().__setattr__(name, value)

    def fwd(self, x, out=None):
        raise NotImplementedError

    def bwd(self, dy, dx_out=None, accumulate=False, addend=None, need_dx=True):
        raise NotImplementedError

    def forward(self, *a, **k):  # nn.Module protocol is only exposed on whole networks (SgxNetwork)
        raise RuntimeError(f"{type(self).__name__} runs inside an SgxNetwork; call the network, not the block")


class Arena:
    """A flat fp32 buffer with named, aligned segments."""

    def __init__(self):
        self.segments = []  # (name, start, numel_physical)
        self.size = 0
        self.buf: Optional[torch.Tensor] = None

    def reserve(self, name, numel):
        start = self.size
        self.segments.append((name, start, numel))
        self.size = _round_up(start + numel, ALIGN)
        return start

    def allocate(self, device, dtype=torch.float32):
        self.buf = torch.zeros(max(self.size, ALIGN), device=device, dtype=dtype)
        return self.buf


def _conv_weight_view(flat, K, C, R, S):
    """OHWI storage with C padded to a multiple of 4; logical [K,C,R,S] view (state_dict shape)."""
    Cp = _round_up(C, 4)
    return flat[: K * R * S * Cp].view(K, R, S, Cp)[..., :C].permute(0, 3, 1, 2), flat[: K * R * S * Cp].view(K, R, S, Cp).permute(0, 3, 1, 2)


class ParamSlot:
    """Book-keeping for one parameter inside the arenas."""

    __slots__ = ("name", "param", "start", "numel", "kind", "kernel_view", "grad_kernel_view", "no_wd")

    def __init__(self, name, param, kind):
        self.name, self.param, self.kind = name, param, kind
        self.start = self.numel = 0
        self.kernel_view = self.grad_kernel_view = None
        self.no_wd = False


class SgxNetwork(nn.Module):
    """A whole model (YoloNAS, ResNet) executing on the kernels.  Subclasses implement `_fwd(x_nhwc)` and `_bwd(*grads)`.

    materialize(device) moves every parameter into the parameter arena (p.data becomes a strided view, p.grad a view
    of the gradient arena) and every BN buffer into the buffer arena.  It is idempotent and is called automatically on
    the first forward; `.to()/.cuda()` before that are fine, after that they are rejected (the arenas are the model)."""

    _materialized = False

    def __setattr__(self, name, value):
        d = self.__dict__
        if name in d and (name in d.get("_modules", ()) or name in d.get("_parameters", ()) or name in d.get("_buffers", ())):
            del d[name]  # a registry mirror (_build_runtime) must not outlive a re-assignment of the name
        super().__setattr__(name, value)

    # ----------------------------------------------------------------------------------------- arenas
    def dead_parameter(self, name: str) -> bool:
        """Parameters that never receive a gradient (reference: QARepVGGBlock.rbr_reparam):
        kept in state_dict, excluded from arenas / optimizer / all-reduce / EMA."""
        return ".rbr_reparam." in name or name.startswith("rbr_reparam.")

    def materialize(self, device=None):
        if self._materialized:
            return self
        from .. import kernels as K  # noqa: F401  (fails loudly if the HIP library is missing)

        device = torch.device(device if device is not None else "cuda")
        self.p_arena, self.g_arena, self.b_arena = Arena(), Arena(), Arena()
        self.slots: List[ParamSlot] = []
        mods = dict(self.named_modules())
        for name, p in self.named_parameters():
            if self.dead_parameter(name):
                p.data = p.data.to(device)
                p.requires_grad_(False)
                continue
            owner = mods[name.rsplit(".", 1)[0]] if "." in name else self
            kind = getattr(owner, "_param_kinds", {}).get(name.rsplit(".", 1)[-1], "flat")
            slot = ParamSlot(name, p, kind)
            if kind == "conv":
                K_, C_, R_, S_ = p.shape
                slot.numel = K_ * R_ * S_ * _round_up(C_, 4)
            else:
                slot.numel = p.numel()
            slot.no_wd = p.dim() <= 1  # biases, BN affine, scalars (optimizer_utils.py:32-59 zero-WD group)
            slot.start = self.p_arena.reserve(name, slot.numel)
            self.g_arena.reserve(name, slot.numel)
            self.slots.append(slot)
        self.p_arena.allocate(device)
        self.g_arena.allocate(device)
        self._bind_slots(copy_in=True)
        # buffers: BN running stats -> buffer arena (fp32); num_batches_tracked -> one int64 arena
        fbufs, ibufs = [], []
        for mname, m in self.named_modules():
            for bname, b in list(m._buffers.items()):
                if b is None:
                    continue
                (fbufs if b.dtype == torch.float32 else ibufs).append((m, bname, b))
        for m, bname, b in fbufs:
            self.b_arena.reserve(bname, b.numel())
        bbuf = self.b_arena.allocate(device)
        for (m, bname, b), (_, start, n) in zip(fbufs, self.b_arena.segments):
            v = bbuf[start: start + n].view(b.shape)
            v.copy_(b.to(device))
            m._buffers[bname] = v
        self.i_arena = torch.zeros(max(len(ibufs), 1), dtype=torch.int64, device=device)
        for i, (m, bname, b) in enumerate(ibufs):
            self.i_arena[i] = int(b)
            m._buffers[bname] = self.i_arena[i]
        self._device = device
        self._materialized = True
        for m in self.modules():
            if isinstance(m, SgxBlock):
                object.__setattr__(m, "_net", self)  # plain attribute: must not register the network as a child module
        self._build_runtime()
        return self

    def _bind_slots(self, copy_in: bool):
        """Make every live parameter a strided view of the parameter arena (its .grad a view of the gradient arena) and rebuild the kernel
        views.  copy_in: the parameter's current values move into the arena (materialize); otherwise the arena already holds them (a
        deep copy: nn.Parameter.__deepcopy__ clones .data and drops .grad, so the copy's parameters must be re-attached to ITS arenas)."""
        pbuf, gbuf = self.p_arena.buf, self.g_arena.buf
        for s in self.slots:
            flat, gflat = pbuf[s.start: s.start + s.numel], gbuf[s.start: s.start + s.numel]
            old = s.param.data
            if s.kind == "conv":
                K_, C_, R_, S_ = old.shape
                view, s.kernel_view = _conv_weight_view(flat, K_, C_, R_, S_)
                gview, s.grad_kernel_view = _conv_weight_view(gflat, K_, C_, R_, S_)
            elif s.kind == "convT":  # logical [C,K,2,2], stored [C][2][2][K]
                C_, K_ = old.shape[:2]
                view = flat.view(C_, 2, 2, K_).permute(0, 3, 1, 2)
                gview = gflat.view(C_, 2, 2, K_).permute(0, 3, 1, 2)
                s.kernel_view, s.grad_kernel_view = view, gview
            else:
                view, gview = flat.view(old.shape), gflat.view(old.shape)
                s.kernel_view, s.grad_kernel_view = view, gview
            if copy_in:
                view.copy_(old.to(pbuf.device))
            s.param.data = view
            s.param.grad = gview

    def _build_runtime(self):
        """Everything that refers to this instance's device memory by address or lives outside tensors: the HIP streams, and the per-step
        job tables (arena views never move, so they are built once).  Called by materialize() and, for a copy, by __deepcopy__."""
        import os

        from .. import kernels as K

        device = self._device
        # Weight gradients are independent of the data-gradient chain: they are forked onto a side HIP stream so that they fill
        # the CUs the (dependent) main-stream kernels leave idle in their tails (most YOLO-NAS layers are 1-2 waves of
        # workgroups).  SGX_SIDE_STREAM=0 disables the fork.
        self.side_stream = _make_side_stream(device) if (device.type == "cuda" and os.environ.get("SGX_SIDE_STREAM", "1") != "0") else None
        # Optional (SGX_AUX_STREAM=1): transpose all data-gradient weights on a third stream underneath the forward pass instead of
        # per call.  Measured neutral-to-negative on YOLO-NAS-S (r1p: 528.7 vs 533.3 images/s): the per-call transposes already hide
        # under the side-stream weight gradients, so it stays off by default.
        self.aux_stream = torch.cuda.Stream(device=device) if (self.side_stream is not None and os.environ.get("SGX_AUX_STREAM", "0") == "1") else None
        # Branch stream (SGX_BRANCH_STREAM: bit 0 forward, bit 1 backward; round 6): a sub-chain of a block that nothing inside the block
        # waits for (YoloNASCSPLayer's conv2: GEMM -> finalize -> sweep, a few dozen microseconds each and dependent on one another) is
        # enqueued on a second in-order stream and joined where its result is consumed, so that its short kernels fill the gaps between the
        # dependent kernels of the main chain instead of lengthening it.
        self.branch_mode = int(os.environ.get("SGX_BRANCH_STREAM", str(BRANCH_STREAM_DEFAULT))) if self.side_stream is not None else 0
        br_prio = int(os.environ.get("SGX_BRANCH_PRIORITY", "0"))  # (measurement switch, r6z)
        self.branch_stream = torch.cuda.Stream(device=device, priority=br_prio) if self.branch_mode else None
        # (SGX_BRANCH_LANES: further branch streams for call sites that fork several mutually independent chains - the head levels, the
        # d alpha reductions.  Two by default and no more: main + side + two lanes are four HIP streams on the runtime's four hardware queues;
        # r6ae / r6af: four lanes 2 % slower, and with GPU_MAX_HW_QUEUES=8 30 % slower)
        self.branch_lanes = [self.branch_stream] + [torch.cuda.Stream(device=device, priority=br_prio) for _ in range(int(os.environ.get("SGX_BRANCH_LANES", str(BRANCH_LANES_DEFAULT))) - 1)] \
            if self.branch_stream is not None else []
        # which call sites fork (SGX_BRANCH_SITES bits: 1 YoloNASCSPLayer conv2, 2 coarse head levels, 4 the up stages' skip branches, 8 the batch
        # re-layout beside the per-step filter preparations, 16 the ResNet blocks' projection shortcuts, 32 the bottlenecks'
        # d alpha = <x, dz> reductions) and up
        # to what size (SGX_BRANCH_MAX_TILES: 64-row x 64-column tiles of the forked chain's largest GEMM - a launch of several rounds of
        # workgroups has no gaps to fill and only contends)
        self.branch_sites = int(os.environ.get("SGX_BRANCH_SITES", str(BRANCH_SITES_DEFAULT)))
        self.branch_max_tiles = int(os.environ.get("SGX_BRANCH_MAX_TILES", str(BRANCH_MAX_TILES_DEFAULT)))
        self._wt_valid = False
        self._step_forward = False  # True while NetFunction.forward runs the network behind prefetch_dgrad_weights
        # Weight gradients are mutually independent: blocks queue them (ConvLayer.wgrad) and the network launches a stretch of backward's
        # worth together (kernels.conv2d_bwd_weight_group: one launch per tile shape, splits sized for the group, partials folded inside
        # the launch) - whenever the queued work passes SGX_WGRAD_GROUP_GFLOP (default 160, ~2 ms of chip time; measured r3i/r3j on
        # YOLO-NAS-S: 40 -> 613, 160 -> 633, 400 -> 629 images/s - every launch ends in a fold chain that nothing of its own fills), at every
        # gradient-bucket boundary and at the end of backward.  0: one call per layer.  Memory: the queue keeps the (x, dy) of its entries
        # alive until the flush - up to ~0.4 GB of operands per 160 GFLOP on YOLO-NAS-S (both would have been freed a few launches later).
        self.wg_group_flops = float(os.environ.get("SGX_WGRAD_GROUP_GFLOP", "160")) * 1e9
        self.wg_eager_rows = int(os.environ.get("SGX_WGRAD_EAGER_ROWS", str(WGRAD_EAGER_ROWS_DEFAULT)))
        self._wg_pending, self._wg_flops = [], 0.0
        # BatchNorm-backward reduce of a plain conv -> BatchNorm -> activation layer inside the data-gradient launch that finalises the layer's
        # output gradient (kernels.BnReduceRequest; round 4): the reduce sweep over (dy, saved conv output) disappears wherever a layer's
        # gradient has a single last writer that is a convolution's data gradient.  SGX_FUSE_BN_REDUCE=0: every layer runs its own sweep.
        self.fuse_bn_reduce = os.environ.get("SGX_FUSE_BN_REDUCE", "1") != "0"
        # The data-gradient weight transposes of ALL convolutions run as one launch at the start of every training forward (a job table
        # built once - the operands are arena views, their addresses never change) instead of one launch per convolution and parity class
        # inside backward (YOLO-NAS-S: 165 launches of ~10 us per step; SGX_WT_BATCH=0 restores the per-call form), and so do the
        # QARepVGG blocks' per-step filter preparations (W1 + I and its transpose, sgx_qarep_prep_batch).
        self.wt_batch = os.environ.get("SGX_WT_BATCH", "1") == "1" and self.aux_stream is None
        for m in self.modules():
            if isinstance(m, SgxBlock):
                m.on_materialize()
        self._dgrad_convs = [m for m in self.modules() if hasattr(m, "transpose_weights") and getattr(m, "_wt", None) is not None]
        self._wt_jobs, self._wt_njobs = None, 0
        if self.wt_batch and self._dgrad_convs:
            from .. import _lib
            import ctypes

            table = b"".join(K.conv2d_transpose_jobs(m._w, m._wt, stride=m.stride, pad=m.padding) for m in self._dgrad_convs)
            self._wt_njobs = len(table) // ctypes.sizeof(_lib.WtransJob)
            self._wt_jobs = torch.frombuffer(bytearray(table), dtype=torch.uint8).to(device)
        self._qp_jobs, self._qp_njobs = None, 0
        recs = [j for j in (m.qarep_prep_job() for m in self.modules() if hasattr(m, "qarep_prep_job")) if j is not None]
        if recs:
            self._qp_njobs = len(recs)
            self._qp_jobs = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8).to(device)
        # Pre-split filter planes (round 5; include/sgx_hip.h sgx_filter_planes_batch): every filter a bf16x3 launch of the training step reads -
        # forward filters, their data-gradient transposes, the QARepVGG blocks' prepared 1x1 filters - is split into its three bf16 pieces
        # ONCE per step, right behind the transposes, instead of once per pixel tile of every launch.  The library serves planes only inside
        # the step's scope (NetFunction.forward / backward) and only for entries made since the weights last changed.  SGX_FILTER_PLANES=0: off.
        self._fp_jobs, self._fp_dev, self._fp_buf = None, None, None
        if self._wt_jobs is not None and os.environ.get("SGX_FILTER_PLANES", "1") != "0":
            from .. import _lib
            import ctypes

            filters = []
            for r in (_lib.WtransJob * self._wt_njobs).from_buffer_copy(table):
                filters.append((r.w, r.K, r.RS, r.C))
                filters.append((r.wt, r.C, r.T, r.K))
            for raw in recs:
                r = _lib.QarepPrepJob.from_buffer_copy(raw)
                filters.append((r.w1p, r.K, 1, r.C))
                filters.append((r.w1pt, r.C, 1, r.K))
            plan, total = K.filter_planes_plan(filters)
            if plan:
                self._fp_buf = torch.empty(total, dtype=torch.uint8, device=device)
                self._fp_jobs, self._fp_dev = K.filter_planes_table(plan, self._fp_buf)
        # Sub-modules, parameters and buffers are fixed objects from here on (arena views never move, load_state_dict copies in place, the
        # network refuses to be moved): mirror them into the instance dictionaries so that `self.bn.weight` is a plain attribute read
        # instead of nn.Module.__getattr__'s three dictionary probes (~2400 of those per YOLO-NAS-S train step).  nn.Module.__setattr__
        # drops the mirror entry if a name is ever re-assigned, so the registries stay authoritative.
        for m in self.modules():
            if isinstance(m, (SgxBlock, SgxNetwork)):
                for table in (m._modules, m._parameters, m._buffers):
                    for name, v in table.items():
                        if v is not None:
                            m.__dict__[name] = v

    _RUNTIME_ATTRS = ("side_stream", "aux_stream", "branch_stream", "branch_lanes", "_wt_jobs", "_qp_jobs", "_dgrad_convs", "_wg_pending", "_fp_jobs", "_fp_dev", "_fp_buf")

    def __deepcopy__(self, memo):
        """copy.deepcopy(model) - what the reference's predict() pipeline does before fusing (pipelines.py:95-100): parameters, buffers and
        arenas are copied (views stay views of the copied arenas); streams and the address-bearing job tables are rebuilt for the copy."""
        import copy

        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k not in self._RUNTIME_ATTRS:
                new.__dict__[k] = copy.deepcopy(v, memo)
        if self._materialized:
            # the copied arenas hold the values; the copied Parameters are detached clones without .grad: re-attach them (and the BatchNorm
            # buffers, which deepcopy also cloned out of the buffer arena), then rebuild everything that refers to device addresses
            new._bind_slots(copy_in=False)
            fb, ib = 0, 0
            for _, m in new.named_modules():
                for bname, b in list(m._buffers.items()):
                    if b is None:
                        continue
                    if b.dtype == torch.float32:
                        _, start, n = new.b_arena.segments[fb]
                        fb += 1
                        m._buffers[bname] = new.b_arena.buf[start: start + n].view(b.shape)
                    else:
                        m._buffers[bname] = new.i_arena[ib]
                        ib += 1
                    m.__dict__.pop(bname, None)
            for m in new.modules():
                if isinstance(m, SgxBlock):
                    object.__setattr__(m, "_net", new)
            new._build_runtime()
        return new

    def prefetch_dgrad_weights(self):
        aux = getattr(self, "aux_stream", None)
        if getattr(self, "_wt_jobs", None) is not None:
            from .. import kernels as K

            K.wtrans_batch(self._wt_jobs, self._wt_njobs)  # current stream: ordered after the optimizer step, before backward
            if getattr(self, "_qp_jobs", None) is not None:
                K.qarep_prep_batch(self._qp_jobs, self._qp_njobs)
            if getattr(self, "_fp_jobs", None) is not None:
                K.filter_planes_invalidate(None)  # whatever an earlier step (of any network) left valid is not this step's
                K.filter_planes_batch(self._fp_jobs, self._fp_dev)
            self._wt_valid = True
            return
        if aux is None:
            return
        aux.wait_stream(torch.cuda.current_stream())  # after the optimizer step that produced the current weights
        with torch.cuda.stream(aux):
            for m in self._dgrad_convs:
                m.transpose_weights()
        self._wt_valid = True

    def join_aux(self):
        aux = getattr(self, "aux_stream", None)
        if aux is not None and self._wt_valid:
            torch.cuda.current_stream().wait_stream(aux)

    def _apply(self, fn, recurse=True):
        if self._materialized:
            probe = fn(torch.empty(0, device=self._device))
            if probe.device != self._device or probe.dtype != torch.float32:
                raise RuntimeError("this model is materialized in HBM arenas; move it before the first forward, not after")
            return self
        return super()._apply(fn, recurse)

    def fork_side(self, fn, *tensors):
        """Run fn() on the side stream after everything enqueued so far on the current stream; `tensors` are what fn reads
        (their memory must not be recycled by the caching allocator before the side stream is done with it)."""
        side = getattr(self, "side_stream", None)
        if side is None:
            return fn()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            out = fn()
        for t in tensors:
            if t is not None:
                t.record_stream(side)
        return out

    def branches(self, site: int, rows: int, cols: int, backward: bool = False) -> bool:
        """Does call site `site` fork a chain whose largest GEMM has `rows` output pixels x `cols` filters?"""
        if getattr(self, "branch_stream", None) is None or not (self.branch_mode & (2 if backward else 1)) or not (self.branch_sites & site):
            return False
        return ((rows + 63) // 64) * ((cols + 63) // 64) <= self.branch_max_tiles

    def data_parallel_streams(self):
        """Stream budget of a data-parallel run (called by GradientAllReducer): the collectives' stream takes the hardware queue of the second
        branch lane.  Measured on the one-rank RCCL communicator (bucket all-reduces really issued; `profiles/r6ah_collectives_lanes.txt`):
        two lanes 743 against 790 images/s without collectives (-6 %), one lane and no d alpha site 783 against 784 (-0.2 %).  Explicit
        SGX_BRANCH_LANES / SGX_BRANCH_SITES settings are left alone."""
        import os

        if getattr(self, "branch_stream", None) is None:
            return
        if "SGX_BRANCH_LANES" not in os.environ:
            self.branch_lanes = self.branch_lanes[:1]
        if "SGX_BRANCH_SITES" not in os.environ:
            self.branch_sites &= ~32

    def fork_branch(self, fn, backward: bool = False, lane: int = 0, queues_wgrads: bool = True):
        """Run fn() on the branch stream, ordered after everything enqueued so far on the current stream; returns (result, joined) where
        `joined` is a callable the caller invokes before the first consumer of what fn wrote (a no-op when the branch stream is off).
        Memory rule (torch's caching allocator keeps one pool per stream): what fn allocates comes from the branch stream's pool and is
        recycled only into later branch allocations, which are ordered behind a later fork's wait; what fn reads or writes of the caller's
        must stay referenced until `joined` was called.  Weight gradients queued inside fn are flushed inside it: the side stream must wait
        for THIS stream's producers (queues_wgrads=False: fn queues none and is joined before the next gradient-bucket boundary)."""
        if getattr(self, "branch_stream", None) is None or not (self.branch_mode & (2 if backward else 1)):
            return fn(), _nothing
        br = self.branch_lanes[lane % len(self.branch_lanes)]
        main = torch.cuda.current_stream()
        br.wait_stream(main)
        with torch.cuda.stream(br):
            out = fn()
            if backward and queues_wgrads:
                self.flush_wgrads()
                if self.side_stream is not None:  # whatever reads this stretch's parameter gradients from the side stream (the bucket
                    self.side_stream.wait_stream(br)  # all-reduce) is ordered behind it even when the join comes later
        return out, lambda: main.wait_stream(br)

    def queue_wgrad(self, x, dy, gw, stride, pad):
        self._wg_pending.append((x, dy, gw, stride, pad))
        k, _, r, s_ = gw.shape
        rows = dy.shape[0] * dy.shape[1] * dy.shape[2]
        self._wg_flops += 2.0 * rows * k * r * s_ * x.shape[3]
        # (large maps: a layer of the main chain takes a millisecond there, and a queue that waits for 160 GFLOP leaves the side stream idle
        # for two or three of them - then the step ends in a backlog of weight gradients with nothing beside them; r6fin2's trace: 2.9 ms idle,
        # then 3.9 ms of backlog of which 0.8 ms behind the main chain's last kernel)
        if self._wg_flops >= self.wg_group_flops or rows >= self.wg_eager_rows:
            self.flush_wgrads()

    def flush_wgrads(self):
        """Launch the queued weight gradients (side stream).  The queue holds references to their operands until here."""
        pending = getattr(self, "_wg_pending", None)
        if not pending:
            return
        from .. import kernels as K

        self._wg_pending, self._wg_flops = [], 0.0
        self.fork_side(lambda: K.conv2d_bwd_weight_group(pending), *[t for e in pending for t in e[:2]])

    def _bucket_ready(self, prefix: str):
        """Backward of sub-network `prefix` is enqueued: its queued weight gradients go out, then the gradient exchange may start."""
        self.flush_wgrads()
        ready = getattr(self, "_grad_ready", None)
        if ready is not None:
            ready(prefix)

    def join_side(self):
        """Make the current stream wait for the side stream (before anything consumes the weight gradients); queued weight gradients go
        out first."""
        self.flush_wgrads()
        side = getattr(self, "side_stream", None)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)

    def set_sync_bn(self, enabled: bool = True):
        """nn.SyncBatchNorm.convert_sync_batchnorm for this network (reference: sg_trainer.py:1344-1350, recipe `sync_bn: True`):
        every BatchNorm computes its training statistics - and their gradients - over all data-parallel ranks."""
        from .layers import BatchNorm

        for m in self.modules():
            if isinstance(m, BatchNorm):
                m.sync = bool(enabled)
        return self

    def prep_model_for_conversion(self, input_size=None, **kwargs):
        """Reference: CustomizableDetector.prep_model_for_conversion (customizable_detector.py:106-118) - every sub-module that
        knows how to re-parameterise itself does so (QARepVGGBlock: branches [+ post-BN] -> one 3x3 conv); eval forward then runs
        the deployment form."""
        self.materialize()
        for m in self.modules():
            if m is not self and hasattr(m, "prep_model_for_conversion"):
                m.prep_model_for_conversion(input_size, **kwargs)
        return self

    _half_inference = False

    def half_inference(self, enabled: bool = True):
        """Run eval-mode forwards on the half-precision path (csrc/half.hip: bf16 activations and filters, fp32 accumulation) - what the
        reference's predict() does with torch.autocast (pipelines.py:76).  The model must be in its folded deployment form
        (prep_model_for_conversion(full_fusion=True)); blocks that are not raise at the first forward.  Training is unaffected."""
        if enabled and not self.supports_half_inference():
            raise NotImplementedError(f"{type(self).__name__}: no half-precision inference path for this architecture")
        self._half_inference = bool(enabled)
        return self

    def supports_half_inference(self) -> bool:
        return False

    def weights_changed(self):
        """Anything that rewrites parameters or BatchNorm statistics outside a training step (load_state_dict, an EMA swap, a broadcast)
        calls this: eval-mode caches derived from the weights - the folded conv+BN filters of prep_model_for_conversion - are dropped and
        rebuilt on demand, instead of silently serving the old weights."""
        for m in self.modules():
            if getattr(m, "_folded", None) is not None:
                m._folded = None
            if getattr(m, "_folded_half", None) is not None:
                m._folded_half = None
        self._wt_valid = False
        self.drop_filter_planes()
        from .. import kernels as K

        K.weights_written()  # bf16 operands of the half-precision path cached against live weights (kernels._half_cached)

    def drop_filter_planes(self):
        """The registry entries of this network's pre-split filter planes stop serving launches (the weights are about to change, or the
        step that made them is over)."""
        if getattr(self, "_fp_jobs", None) is not None:
            from .. import kernels as K

            K.filter_planes_invalidate(self._fp_jobs)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self.weights_changed()
        return out

    def zero_grad(self, set_to_none: bool = False):
        if self._materialized:
            from .. import kernels as K

            K.fill(self.g_arena.buf, 0.0)
        else:
            super().zero_grad(set_to_none)

    def live_parameters(self):
        return [s.param for s in self.slots]

    # ----------------------------------------------------------------------------------------- autograd bridge
    def _fwd(self, x):
        raise NotImplementedError

    def _bwd(self, *grads):
        raise NotImplementedError

    def forward(self, x):
        if not self._materialized:
            self.materialize(x.device if x.is_cuda else None)
        from .. import _lib

        if not x.is_cuda and not _lib._TEST_HOST_MODE:  # _TEST_HOST_MODE: tests/emu only (kernel-logic checks on host memory)
            raise RuntimeError("super_gradients_amd models run on the HIP device only (no CPU fallback): move the batch to cuda")
        if self.training:
            self.i_arena.add_(1)  # every BatchNorm's num_batches_tracked in one launch
        if self.training and torch.is_grad_enabled():
            flat = NetFunction.apply(self, x, self._grad_anchor())
        else:
            with torch.no_grad():
                flat = self._fwd(x)
        return self._pack(flat)

    def _pack(self, flat):
        """flat tuple of output tensors -> the structure the reference model returns."""
        return flat[0] if len(flat) == 1 else tuple(flat)

    def _differentiable_outputs(self, n):
        return [True] * n

    def _grad_anchor(self):
        # a leaf that requires grad so that autograd calls NetFunction.backward; parameter gradients themselves are
        # written straight into the gradient arena by the blocks' bwd() (p.grad are views of it).
        a = getattr(self, "_anchor", None)
        if a is None:
            a = self._anchor = torch.zeros(1, device=self._device, requires_grad=True)
        return a


# Share of the chip's CUs the weight gradients' side stream may use (SGX_SIDE_CUS, percent; 100 = an ordinary stream).
SIDE_STREAM_CU_PERCENT = 100
BRANCH_STREAM_DEFAULT = 3
WGRAD_EAGER_ROWS_DEFAULT = 800000  # (r6z: 800000 +0.4 % on YOLO-NAS-S at batch 32 - its 160 x 160 and 320 x 320 maps; 200000 -0.6 %, 50000 -1.9 %; M, L within noise)
BRANCH_SITES_DEFAULT = 63
BRANCH_LANES_DEFAULT = 2
BRANCH_MAX_TILES_DEFAULT = 1 << 30


def _nothing():
    return None


def _make_side_stream(device):
    """The side HIP stream of the weight gradients.  Below 100 % it is created with a CU mask (sgx_stream_create_partial): the
    weight-gradient workgroups live for hundreds of microseconds and otherwise occupy every CU, and the short dependent kernels of the main
    stream - the critical path of the step - then wait between them (r4t: 6.8 ms per step)."""
    import ctypes
    import os

    pct = int(os.environ.get("SGX_SIDE_CUS") or SIDE_STREAM_CU_PERCENT)
    if pct >= 100:
        # (SGX_SIDE_PRIORITY=-1: a high-priority HIP stream - measurement switch, r6z)
        return torch.cuda.Stream(device=device, priority=int(os.environ.get("SGX_SIDE_PRIORITY", "0")))
    from .._lib import check, lib

    handle = ctypes.c_void_p()
    with torch.cuda.device(device):
        check(lib().sgx_stream_create_partial(pct, ctypes.byref(handle)), "sgx_stream_create_partial")
    return torch.cuda.ExternalStream(handle.value, device=device)  # (lives as long as the process: a network's streams are never recycled)


class NetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, anchor):
        ctx.net = net
        # the batch's re-layout (NCHW -> NHWC, a network's _input_layout) beside the per-step filter preparations: branch stream, site 8
        xh, joined = None, _nothing
        if hasattr(net, "_input_layout") and net.branches(8, 0, 0):
            xh, joined = net.fork_branch(lambda: net._input_layout(x))
        net.prefetch_dgrad_weights()
        joined()
        planes = getattr(net, "_fp_jobs", None) is not None
        if planes:
            from .. import kernels as K

            K.filter_planes_scope(True)
        net._step_forward = True  # (the per-step filter preparations above are current for exactly this forward: ConvTranspose2x2.fwd)
        try:
            flat = tuple(net._fwd(x) if xh is None else net._fwd(x, xh=xh))
        finally:
            net._step_forward = False
            if planes:
                K.filter_planes_scope(False)
        ctx.mark_non_differentiable(*[t for t, d in zip(flat, net._differentiable_outputs(len(flat))) if not d])
        return flat

    @staticmethod
    def backward(ctx, *grads):
        net = ctx.net
        net.join_aux()
        planes = getattr(net, "_fp_jobs", None) is not None
        if planes:
            from .. import kernels as K

            K.filter_planes_scope(True)
        try:
            net._bwd(*grads)
        except BaseException:
            if planes:
                K.filter_planes_scope(False)
                net.drop_filter_planes()
            # the queue holds operands of THIS backward: a later one must not launch them into the gradient arena; weight
            # gradients already forked onto the side stream are joined, and the per-step transposes count as stale
            net._wg_pending, net._wg_flops = [], 0.0
            net._wt_valid = False
            try:
                net.join_side()
            except Exception:  # (the original error is the one to report)
                pass
            raise
        if planes:
            K.filter_planes_scope(False)
            net.drop_filter_planes()
        net.flush_wgrads()
        net._wt_valid = False
        net.join_side()
        hook = getattr(net, "_post_backward_hook", None)
        if hook is not None:
            hook()
        return None, None, torch.zeros_like(net._anchor)
